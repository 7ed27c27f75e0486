"""Worker for the multi-process tests: python -m torch.distributed.run ... tests/dist_worker.py <mode> <spec> <json params> <out>
mode = shim (CPU, gloo, host lowerings) | hip (one GPU shared by all ranks, exchange staged over gloo)."""
import json
import os
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
    if mode == "shim":
        from shim_step_engine import ShimStepEngine
        eng = ShimStepEngine(spec, params, rank, world)
        chk = ShardedChecker(spec, params, engine=eng, chunk_states=opts.get("chunk", 1000), max_distinct=opts.get("max_distinct", 0),
                             max_levels=opts.get("max_levels", 0), fanout_cap=opts.get("fanout_cap", 64), new_cap=opts.get("new_cap", 64),
                             stay_threshold=opts.get("stay_threshold", 1 << 16), rebalance_ratio=opts.get("rebalance_ratio", 1.25),
                             replicate_until=opts.get("replicate_until", 0), packed=opts.get("packed", True))
    else:
        chk = ShardedChecker(spec, params, device=0, chunk_states=opts.get("chunk", 1 << 14), max_distinct=opts.get("max_distinct", 0),
                             max_levels=opts.get("max_levels", 0), table_capacity=opts.get("table", 1 << 22), arena_capacity=opts.get("arena", 1 << 20),
                             fanout_cap=opts.get("fanout_cap", 32), new_cap=opts.get("new_cap", 16),
                             stay_threshold=opts.get("stay_threshold", 1 << 16), rebalance_ratio=opts.get("rebalance_ratio", 1.25),
                             replicate_until=opts.get("replicate_until", 0), packed=opts.get("packed", True), trace=opts.get("trace", False))
    r = chk.run()
    trace = chk.counterexample() if opts.get("trace") else None
    _, local, _ = chk.eng.counters()
    shares = [None] * world
    dist.all_gather_object(shares, local)
    if rank == 0:
        Path(out).write_text(json.dumps(dict(r, shares=shares, trace=trace, phases={k: v for k, v in chk.phase_s.items() if k.endswith("levels")})))
    chk.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
