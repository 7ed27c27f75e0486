"""One run of one compiled-PlusCal model on one back-end, for rocprofv3: python profiles/run_pcal_once.py {msq3|msq4|pagecache} {jit|vm}"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import tla_rust_amd as amd

which, backend = sys.argv[1], sys.argv[2]
if which.startswith("msq"):
    k = int(which[3])
    src = (ROOT / "specs" / "pluscal" / "ms_queue_counted.tla").read_text()
    cfg = f"CONSTANTS N = 3 K = {k} Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n"
    kw = dict(table_capacity=1 << (28 if k == 3 else 30), arena_capacity=(40 if k == 3 else 130) << 20, chunk_states=1 << 21)
else:
    src = (ROOT / "specs" / "pluscal" / "pagecache.tla").read_text()
    cfg = "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n"
    kw = dict(table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 21)
prog = amd.Program(src, cfg)
eng = amd.Engine("pcal", prog.params, trace=False, jit=backend == "jit", **kw)
r = eng.run()
print(which, backend, r.distinct, r.generated, r.depth, r.verdict, round(r.seconds, 4))
eng.close()
prog.close()
