"""`tlc X.tla` on the GPUs of one node — the multi-GPU front door (SURVEY.md §8b `mc X.tla -gpus P`, §8e):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port 29512 \\
        -m tla_rust_amd.mc_multi X.tla [-config X.cfg] [-maxdistinct N] [-maxlevels N] [-chunk N] [-tablelog2 T] [-arena N] [-exchange F] [-fanout N]

One process per GPU (LOCAL_RANK), `torch.distributed` backend "nccl" (= RCCL over xGMI).  Every rank resolves X.tla / X.cfg
through the C ABI exactly like the one-GPU `mc` (mc_resolve_files: same lowering registry, same text verification, same
PlusCal compiler), creates its engine and runs tla_rust_amd.sharded.ShardedChecker; rank 0 prints TLC's report lines
(reference README.md:319-320, testout2:260-266).  Counters, depth and verdict are those of the one-GPU run.  A counterexample is
rebuilt ACROSS ranks: the engines keep (rank, index, action) of every state's parent — a state that moves to its owner takes
them along — and on an error the behaviour is walked back with one small broadcast per state (ShardedChecker.counterexample),
printed in TLC's layout ("State k: <Action>" + the state, README.md:270-311; the <line, col> positions of the one-GPU report are
not reproduced).  `-rerun` adds what round 1 did instead: rank 0 repeats the search on ONE GPU bounded to the error's depth and
prints that run's full TLC report — only possible while those levels fit one GPU.
Exit status of every rank: TLC's (0 / 12 safety violation / 11 deadlock); torch.distributed.run itself returns 1 when its
ranks exit non-zero.  Without a launcher (no WORLD_SIZE) it runs one rank.  `-backend gloo -device 0` puts several ranks on ONE GPU (tests)."""
import os
import sys
import time


def parse(argv):
    o = dict(tla=None, config=None, maxdistinct=0, maxlevels=0, chunk=1 << 19, tablelog2=27, arena=1 << 25, backend="nccl", device=None,
             generic=False, unverified=False, rerun=False)
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ("-config", "-backend") and i + 1 < len(argv):
            o[a[1:]] = argv[i + 1]
            i += 2
        elif a in ("-maxdistinct", "-maxlevels", "-chunk", "-tablelog2", "-arena", "-device", "-workers", "-fanout") and i + 1 < len(argv):
            if a != "-workers":      # accepted and ignored like the one-GPU CLI: the GPUs are the worker pool
                o[a[1:]] = int(argv[i + 1])
            i += 2
        elif a == "-exchange" and i + 1 < len(argv):
            if argv[i + 1] not in ("packed", "measured", "exact"):
                raise SystemExit("mc_multi: -exchange packed | measured | exact")
            o["exchange"] = argv[i + 1]
            i += 2
        elif a == "-generic":
            o["generic"] = True
            i += 1
        elif a == "-unverified":
            o["unverified"] = True
            i += 1
        elif a == "-rerun":
            o["rerun"] = True
            i += 1
        elif not a.startswith("-"):
            o["tla"] = a
            i += 1
        else:
            raise SystemExit(f"mc_multi: unknown option {a}")
    if not o["tla"]:
        raise SystemExit(__doc__)
    return o


def report(r, world, seconds, trace=None):
    """TLC's closing lines for a sharded run (format of README.md:319-320 / testout2:260-266)."""
    out = [f"Finished computing initial states: {r.levels[0] if r.levels else 0} distinct state{'' if r.levels and r.levels[0] == 1 else 's'} generated."]
    if r.verdict == "ok":
        out.append("Model checking completed. No error has been found.")
    elif r.verdict == "budget":
        out.append("Search stopped by the level/state budget; no error has been found so far.")
    else:
        what = {"invariant": "Error: Invariant is violated.", "assert": "Error: The first argument of Assert evaluated to FALSE.",
                "deadlock": "Error: Deadlock reached.", "spec-error": "Error: TLC would raise an evaluation error."}[r.verdict]
        out.append(what)
        if trace:
            out.append("Error: The behavior up to this point is:")
            for k, (action, text) in enumerate(trace):
                out.append(f"State {k + 1}: <{action}>")
                out.append(text.rstrip("\n"))
                out.append("")
        else:
            out.append("(no behavior: the engines were created without parent pointers)")
    out.append(f"{r.generated} states generated, {r.distinct} distinct states found, {r.queue_left} states left on queue.")
    out.append(f"The depth of the complete state graph search is {r.depth}.")
    out.append(f"({world} GPU{'s' if world != 1 else ''}, {seconds:.3f} s, {r.distinct / max(seconds, 1e-9):.3g} distinct states/s)")
    return "\n".join(out) + "\n"


def main(argv=None):
    o = parse(sys.argv[1:] if argv is None else argv)
    import torch
    import torch.distributed as dist

    from . import ResolvedSpec
    from .sharded import ShardedChecker

    launched = "WORLD_SIZE" in os.environ
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = o["device"] if o["device"] is not None else local
    if not torch.cuda.is_available():
        raise SystemExit("mc_multi: no HIP device visible — the checker has no CPU fallback")
    torch.cuda.set_device(device)
    if launched:
        dist.init_process_group(o["backend"])
    rank = dist.get_rank() if launched else 0
    world = dist.get_world_size() if launched else 1
    rs = ResolvedSpec(o["tla"], o["config"], generic=o["generic"], unverified=o["unverified"])
    chk = ShardedChecker(rs.spec, rs.params, device=device, chunk_states=o["chunk"], max_distinct=o["maxdistinct"], max_levels=o["maxlevels"],
                         table_capacity=1 << o["tablelog2"], arena_capacity=o["arena"], trace=True,
                         exchange=o.get("exchange", "exact"),
                         **({"packed_fanout": o["fanout"], "move_fanout": 2 * o["fanout"]} if o.get("fanout") else {}))
    if launched:  # communicator set-up (RCCL builds its rings on the first collective) stays out of the reported time
        dist.all_reduce(torch.zeros(1, device="cpu" if o["backend"] == "gloo" else f"cuda:{device}"))
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = chk.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trace = chk.counterexample() if r.verdict not in ("ok", "budget") else None   # collective: every rank walks along
    chk.close()
    rs.close()
    if launched:
        dist.destroy_process_group()  # before rank 0's (possibly long) one-GPU re-run: the other ranks are done
    if rank == 0:
        text = report(r, world, dt, trace)
        if r.verdict not in ("ok", "budget") and not o["generic"] and o["rerun"]:
            try:  # the counterexample: one GPU, the same search bounded to the error's depth
                from . import check_files
                one, rep = check_files(o["tla"], o["config"], device=device, max_levels=r.depth + 1, chunk_states=o["chunk"], unverified=o["unverified"],
                                       table_capacity=1 << o["tablelog2"], arena_capacity=o["arena"])
                if one.verdict == r.verdict:
                    text = rep + f"(counterexample rebuilt by a one-GPU run bounded to depth {r.depth + 1}; the sharded search on {world} GPU" \
                                 f"{'s' if world != 1 else ''} had generated {r.generated} states, {r.distinct} distinct)\n"
            except Exception as e:  # noqa: BLE001 — e.g. the levels do not fit one GPU's arena: keep the sharded report
                text += f"(one-GPU re-run for the counterexample failed: {e})\n"
        sys.stdout.write(text)
        sys.stdout.flush()
    return 0 if r.verdict in ("ok", "budget") else 11 if r.verdict == "deadlock" else 12     # TLC's exit codes, like `mc`


if __name__ == "__main__":
    sys.exit(main())
