"""How many seen-set look-ups does the wavefront's own duplicate filter answer, as a function of the ORDER in which a level's new states are laid
out in the arena?  Host simulation (tests/_shim: the device lowering compiled for the host) of the by-family kernel's candidate stream:
python profiles/probe_order_sim.py [K]   (raft, 3 servers, MaxTerm 2, MaxMsgKeys K; --t3: MaxTerm 3, first levels)"""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import helpers

args = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(args[0]) if args else 7
lib = helpers.shim_lib()
if "--t3" in sys.argv:
    d, ml, name = helpers.spec_desc("raft", [3, 4, 3, 3, 1, 1, 8, 2, 4, 8]), 19, "t3 (MaxTerm 3, MaxMsgKeys 8), 18 levels"
else:
    d, ml, name = helpers.spec_desc("raft", [3, 4, 2, 3, 1, 1, K, 1, 4, K]), 0, f"MaxTerm 2, MaxMsgKeys {K}, complete"
out = (C.c_uint64 * 8)()
for mode, group, what in ((0, 1, "parent-major (a parent's new states adjacent)"), (1, 1, "device: per 128 parents class-major, then wavefront, slot-major"),
                          (2, 8, "groups of 8 parents, class-major inside"), (2, 16, "groups of 16 parents, class-major inside"), (2, 32, "groups of 32"), (2, 64, "groups of 64 (= per wavefront)")):
    for wfilt in (128, 256, 512):
        lib.shim_probe_order_sim(C.byref(d), C.c_uint64(ml), mode, wfilt, group, out)
        cands, hits, dups, new = (int(out[i]) for i in range(4))
        print(json.dumps({"model": name, "order": what, "filter_entries": wfilt, "candidates": cands, "filter_hits": round(hits / cands, 4),
                          "table_duplicates": round(dups / cands, 4), "new": round(new / cands, 4), "states": new}), flush=True)
