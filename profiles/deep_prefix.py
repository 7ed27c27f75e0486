"""Beyond the oracle's reach: the raft bench model to a 200 M budget (the 8-GPU weak-scaling size) on ONE GPU,
fused engine vs sharded step engine (world 1) — self-consistency of two code paths, capacities, memory."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
from tla_rust_amd.sharded import ShardedChecker

# deep_prefix.py [BUDGET] [raft | ssi_sym]: ssi_sym = BASELINE config 5 under cfg SYMMETRY Perms
WHAT = sys.argv[2] if len(sys.argv) > 2 else "raft"
SPEC, P, FAN, NEW, GROW = ("raft", [3, 4, 2, 3, 1, 1, 24, 2, 8], 48, 6, 1.8) if WHAT == "raft" else ("ssi", [4, 3, 127, 0, 0, 3], 16, 12, 6.0)
BUDGET = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
CAP = int(BUDGET * GROW)
t0 = time.perf_counter()
eng = amd.Engine(SPEC, P, table_capacity=1 << 30, arena_capacity=CAP, chunk_states=1 << 20, max_distinct=BUDGET, trace=False)
a = eng.run()
ta = time.perf_counter() - t0
eng.close()
print(json.dumps(dict(path="fused", distinct=a.distinct, generated=a.generated, depth=a.depth, s=round(ta, 3), levels=a.levels[-7:])), flush=True)
t0 = time.perf_counter()
chk = ShardedChecker(SPEC, P, device=0, chunk_states=1 << 19, max_distinct=BUDGET, table_capacity=1 << 30, arena_capacity=CAP,
                     fanout_cap=FAN, new_cap=NEW)
b = chk.run()
tb = time.perf_counter() - t0
print(json.dumps(dict(path="sharded(world=1)", distinct=b.distinct, generated=b.generated, depth=b.depth, s=round(tb, 3),
                      same=(a.distinct, a.generated, a.levels) == (b.distinct, b.generated, b.levels), phases={k: round(v, 3) if isinstance(v, float) else v for k, v in chk.phase_s.items()})), flush=True)
chk.close()
