"""Worker for the multi-process tests: python -m torch.distributed.run ... tests/dist_worker.py <mode> <spec> <json params> <out>
mode = shim (CPU, gloo, host lowerings) | hip (one GPU shared by all ranks, exchange staged over gloo).  Either way the level
loop that runs is the library's (tla_rust_amd/csrc/shard_loop.h) with torch.distributed's collectives handed in as callbacks."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch.distributed as dist  # noqa: E402

from tla_rust_amd.sharded import ShardedChecker  # noqa: E402


def main():
    mode, spec, params, out = sys.argv[1], sys.argv[2], json.loads(sys.argv[3]), sys.argv[4]
    opts = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    keep = None
    if spec == "pcal_file":  # a PlusCal module compiled in this process: params = {"path", "invariants", "constants"}
        if mode == "shim":
            import helpers
            keep = helpers.ShimProgram(Path(params["path"]).read_text(), params.get("invariants", []), params.get("constants", {}))
        else:
            import tla_rust_amd as amd
            cfg = "".join(f"CONSTANT {k} = {v}\n" for k, v in params.get("constants", {}).items())
            cfg += "".join(f"INVARIANT {i}\n" for i in params.get("invariants", []))
            keep = amd.Program(Path(params["path"]).read_text(), cfg)
        spec, params = "pcal", keep.params
    common = dict(chunk_states=opts.get("chunk", 1000 if mode == "shim" else 1 << 14), max_distinct=opts.get("max_distinct", 0),
                  max_levels=opts.get("max_levels", 0), stay_threshold=opts.get("stay_threshold", 1 << 16),
                  rebalance_ratio=opts.get("rebalance_ratio", 1.25), replicate_until=opts.get("replicate_until", 0),
                  packed_fanout=opts.get("packed_fanout", 16), move_fanout=opts.get("move_fanout", 64 if mode == "shim" else 32),
                  exchange=opts.get("exchange", "exact"), cap_safety_pct=opts.get("cap_safety_pct", 0))
    if mode == "shim":
        from shim_step_engine import ShimShard  # noqa: F401
        chk = ShardedChecker(spec, params, engine=ShimShard(spec, params, rank, world), **common)
    else:
        chk = ShardedChecker(spec, params, device=0, table_capacity=opts.get("table", 1 << 22), arena_capacity=opts.get("arena", 1 << 20),
                             trace=opts.get("trace", False), **common)
    def make():
        if mode == "shim":
            return ShardedChecker(spec, params, engine=ShimShard(spec, params, rank, world), **common)
        return ShardedChecker(spec, params, device=0, table_capacity=opts.get("table", 1 << 22), arena_capacity=opts.get("arena", 1 << 20),
                              trace=opts.get("trace", False), **common)
    if opts.get("expect_error"):   # every rank writes what ITS call returned: the ranks must agree (tests of the failure paths)
        import tla_rust_amd as amd
        try:
            r = chk.run()
            res = {"code": 0, "restarts": chk.stats.get("restarts", 0),
                   "distinct": r.distinct, "generated": r.generated, "levels": list(r.levels)}
        except (amd.McError, RuntimeError) as e:   # (the host build's engine raises a plain RuntimeError: "... failed: <rc> <text>")
            import re
            m = re.search(r"failed: (-?\d+)", str(e))
            res = {"code": getattr(e, "code", None) if getattr(e, "code", None) is not None else int(m.group(1)), "msg": str(e)}
        Path(f"{out}.rank{rank}").write_text(json.dumps(res))
        dist.barrier()
        dist.destroy_process_group()
        return
    r = chk.run()
    first = None
    if opts.get("checkpoint"):  # stop on the budget, write one file per rank, continue in FRESH engines (TLC -recover)
        first = dict(r)
        stem = opts["checkpoint"]
        chk.checkpoint(f"{stem}.rank{rank}")
        dist.barrier()   # every rank's file is complete before anybody reads one (restore_wrong_rank reads a neighbour's)
        chk.close()
        chk = make()
        wrong = opts.get("restore_wrong_rank") and world > 1
        try:
            chk.restore(f"{stem}.rank{(rank + 1) % world if wrong else rank}", max_distinct=0, max_levels=opts.get("resume_max_levels", 0))
            restore_error = None
        except Exception as e:  # noqa: BLE001
            restore_error = str(e)
        if opts.get("restore_only_rank0") and rank != 0:
            chk.close()
            chk = make()
            chk.opts["max_distinct"], chk.opts["max_levels"] = 0, 0
        if restore_error is not None:
            errs = [None] * world
            dist.all_gather_object(errs, restore_error)
            if rank == 0:
                Path(out).write_text(json.dumps(dict(restore_errors=errs)))
            chk.close()
            dist.destroy_process_group()
            return
        try:
            r = chk.run()
        except Exception as e:  # noqa: BLE001
            errs = [None] * world
            dist.all_gather_object(errs, str(e))
            if rank == 0:
                Path(out).write_text(json.dumps(dict(run_errors=errs, first=first)))
            chk.close()
            dist.destroy_process_group()
            return
    trace = chk.counterexample() if opts.get("trace") else None
    shares = [None] * world
    dist.all_gather_object(shares, chk.local_distinct)
    if rank == 0:
        Path(out).write_text(json.dumps(dict(r, first=first, shares=shares, trace=trace, phases={k: v for k, v in chk.stats.items() if k.endswith("levels")},
                                             stats=chk.stats)))
    chk.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
