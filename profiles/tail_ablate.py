"""What is the in-wave writer's time made of?  ("first split that 52 ms three ways with a write-only and a re-read-only
ablation flag".)  Needs the ablation build of the 3-server raft translation unit (-DMC_TAIL_ABLATE=1, TLAMC_LIB=.../libtlamc_tailabl.so).
The contract workload is searched normally to level L (21: 215 M states), then ONE more level (76.6 M parents -> 78.2 M new states) is timed
with an ablation bit on; its output is garbage and is never expanded (the next variant starts from Init again).
  bit 20: the writer gathers its "parent rows" from the arena's first two blocks (every gather an L1 / L2 hit): no re-read traffic
  bit 21: every workgroup's survivors start on a 64-state boundary: every column store is one whole 512-byte row of a block
  bit 22: every row is stored into the arena's last 64 states: no write traffic to speak of
  bit 23: no writer at all
python profiles/tail_ablate.py [L] [repeats]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd

L = int(sys.argv[1]) if len(sys.argv) > 1 else 21
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = amd.Engine("raft", [3, 4, 3, 3, 1, 1, 8, 2, 4, 8], table_capacity=40 << 26, arena_capacity=527_000_000, chunk_states=(1 << 24) - 256, trace=False, timing=True)
VARIANTS = [("normal", 0), ("parent rows from two hot blocks (no re-read traffic)", 1 << 20), ("64-aligned survivor runs", 1 << 21),
            ("rows stored into one hot block (no write traffic)", 1 << 22), ("hot parents + hot stores", (1 << 20) | (1 << 22)), ("no writer", 1 << 23)]
rows = []
for rep in range(REP):
    for name, bits in VARIANTS:
        eng.run()                      # a complete search: the next step starts from Init
        r0 = eng.step(L)               # levels 1 .. L, normally
        eng.debug_flags(set=bits)
        r1 = eng.step(1)               # level L -> L + 1 with the ablation bit
        ks = eng.kernel_stats()
        eng.debug_flags(clear=bits)
        rows.append(dict(variant=name, bits=bits, rep=rep, level=L, parents=r0["levels"][-1] if "levels" in r0 else None,
                         expand_ms=ks["expand"]["ms_total"], launches=ks["expand"]["launches"], new_states=r1.distinct - r0.distinct))
        print(json.dumps(rows[-1]), flush=True)
eng.close()
