"""Deeper pins of the C oracle to the REFERENCE'S TEXT, made with the product's C++ evaluator (tla_rust_amd/csrc/tlaeval.cpp through the
test door) instead of the Python one, which needs hours at these sizes:

    raft_3s_keys7   examples/raft.tla under specs/MCraft.tla, 3 servers, MaxMsgKeys = 7: 2 303 950 states (the evaluator: ~10 min)
    ssi_2x3         examples/serializableSnapshotIsolation.tla under specs/MCssi.tla, 2 txns x 3 keys: 7 910 565 states

For each model BOTH sides are run here — the evaluator on the reference's module text, the C oracle (oracle/spec_raft.c, spec_ssi.c: an
independent hand restatement) — and the fixture is written only if their per-level state SETS (canonical TLA+ text, sha256 per level)
are equal; the fixture then holds the evaluator's counters and digests.  The CPU suite re-checks the ORACLE against the fixture's
counters always and against the digests when TLAMC_SLOW=1 (tests/test_reference_text_*.py); re-evaluating the text is this script.

usage: python tests/golden/make_deep_text_pin.py raft_3s_keys7 | ssi_2x3
"""
import hashlib
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
import helpers  # noqa: E402
from make_reference_text_golden import RAFT_ORDER, SSI_INVARIANTS, SSI_ORDER, raft_cfg, ssi_cfg  # noqa: E402

DEEP = {
    "raft_3s_keys7": dict(kind="raft", params=[3, 4, 2, 3, 1, 3], mk=7),
    "ssi_2x3": dict(kind="ssi", params=[2, 3, 127, 0]),
}


def digests(path):
    """sha256 per BFS level over the sorted state lines, streamed (the dumps are several GB)"""
    by = {}
    with open(path) as f:
        for line in f:
            lv, _, text = line.rstrip("\n").partition(" ")
            by.setdefault(int(lv[1:]), []).append(text)
    out = []
    for k in sorted(by):
        out.append(hashlib.sha256("\n".join(sorted(by[k])).encode()).hexdigest()[:16])
        by[k] = None
    return out


def main(name):
    m = DEEP[name]
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        td = Path(td)
        cfg = td / "m.cfg"
        if m["kind"] == "raft":
            cfg.write_text(raft_cfg(*m["params"][:5], m["params"][5], m["mk"]))
            tla, order, spec, op = ROOT / "specs" / "MCraft.tla", RAFT_ORDER, "raft", m["params"] + [0, m["mk"]]
        else:
            cfg.write_text(ssi_cfg(m["params"][0], m["params"][1], SSI_INVARIANTS))
            tla, order, spec, op = ROOT / "specs" / "MCssi.tla", SSI_ORDER, "ssi", m["params"]
        t = time.time()
        r = helpers.tlaeval_run(tla, cfg, search=["/root/reference/examples"], dump=td / "eval.txt", order=order)
        te = time.time() - t
        assert r["rc"] == 0, r
        de = digests(td / "eval.txt")
        (td / "eval.txt").unlink()
        t = time.time()
        o = helpers.oracle_run(spec, op, dump=str(td / "oracle.txt"))
        to = time.time() - t
        do = digests(td / "oracle.txt")
    same = (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"]) and de == do
    print(name, "evaluator", round(te), "s, oracle", round(to), "s:", r["distinct"], "states,", "EQUAL" if same else "DIFFERENT")
    if not same:
        sys.exit(1)
    f = ROOT / "tests" / "golden" / ("raft_reference_text.json" if m["kind"] == "raft" else "ssi_reference_text.json")
    g = json.loads(f.read_text())
    g[name] = dict(distinct=r["distinct"], generated=r["generated"], depth=r["depth"], levels=r["levels"], verdict="ok", level_digests=de,
                   source=f"tests/golden/make_deep_text_pin.py {name}: tlaeval.cpp on the reference's module text ({round(te)} s) == the C oracle "
                          f"({round(to)} s), per-level state sets")
    f.write_text(json.dumps(g, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
