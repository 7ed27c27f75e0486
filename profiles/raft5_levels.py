"""BASELINE config 4's model on one GPU: per-level counts of levels 13-18 (oracle: level 15 = 38 579 685; round 3 found round 2's
64-bit guard mask losing 216 states there) under (timing, max_levels)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd

params = [5, 6, 2, 5, 1, 1, 18, 1, 4]
for timing, ml in ((True, 15), (False, 16), (False, 18), (True, 18)):
    eng = amd.Engine("raft", params, table_capacity=3 << 29, arena_capacity=1_300_000_000, chunk_states=1 << 20, max_levels=ml, trace=False, timing=timing)
    r = eng.run()
    print(json.dumps(dict(timing=timing, max_levels=ml, levels_13_on=r.levels[12:], missing15=38579685 - r.levels[14], distinct=r.distinct, verdict=r.verdict)), flush=True)
    eng.close()
