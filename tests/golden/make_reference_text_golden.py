"""Makes tests/golden/raft_reference_text.json: the reference's OWN spec text (/root/reference/examples/raft.tla, read where
it lies) evaluated by oracle/tlaplus.py under specs/MCraft.tla.  Run in the build container (the GPU box has no
/root/reference):  python tests/golden/make_reference_text_golden.py
"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
REF = Path("/root/reference/examples")
RAFT_ORDER = ["messages", "elections", "allLogs", "currentTerm", "state", "votedFor", "clientRequests", "log", "commitIndex",
              "committedLog", "committedLogDecrease", "votesSent", "votesGranted", "voterLog", "nextIndex", "matchIndex"]

# params = the lowering's {nServer, MaxClientRequests, MaxTerm, MaxLogLen, MaxMsgs, invariantMask}
RAFT_MODELS = {
    "raft_2s_mcr1": dict(params=[2, 1, 2, 9, 1, 1], clash="test"),        # 6 128 distinct
    "raft_2s_mcr2": dict(params=[2, 2, 2, 9, 1, 1], clash="test"),        # 13 634 distinct
    "raft_2s_mcr2_naive": dict(params=[2, 2, 2, 9, 1, 1], clash="ignore"),  # negative control: 15 794
    "raft_2s_mcr2_keys8": dict(params=[2, 2, 2, 9, 1, 1], clash="test", mk=8),  # the MaxMsgKeys conjunct of StateConstraint
    "raft_2s_mm2_keys6": dict(params=[2, 1, 2, 9, 2, 1], clash="test", mk=6),   # two copies in flight: DuplicateMessage
}


def oracle_params(name):
    """the C oracle's parameter vector of a model (its p[6] = naive flag, p[7] = MaxMsgKeys)"""
    m = RAFT_MODELS[name]
    return m["params"] + [0, m["mk"]] if m.get("mk") else m["params"]


def device_params(name):
    """the lowering's parameter vector (p[6..8] = capacities, 0 = default; p[9] = MaxMsgKeys)"""
    m = RAFT_MODELS[name]
    return m["params"] + [0, 0, 0, m["mk"]] if m.get("mk") else m["params"]


def raft_cfg(n, mcr, mt, mll, mm, inv=1, mk=64):
    servers = ", ".join(f"s{i + 1}" for i in range(n))
    invs = " ".join(nm for bit, nm in ((1, "NoTwoLeaders"), (2, "CommittedLogStable")) if inv & bit)
    return f"""SPECIFICATION Spec
CONSTANTS
  Server = {{{servers}}}
  Follower = Follower   Candidate = Candidate   Leader = Leader   Nil = Nil
  RequestVoteRequest = RequestVoteRequest       RequestVoteResponse = RequestVoteResponse
  AppendEntriesRequest = AppendEntriesRequest   AppendEntriesResponse = AppendEntriesResponse
  MaxClientRequests = {mcr}
  MaxTerm = {mt}   MaxLogLen = {mll}   MaxMsgs = {mm}   MaxMsgKeys = {mk}
CONSTRAINT StateConstraint
{"INVARIANT " + invs if invs else ""}
"""


def run_raft_text(name):
    import tlaplus as T
    m = RAFT_MODELS[name]
    c = T.Checker(ROOT / "specs" / "MCraft.tla", cfg_text=raft_cfg(*m["params"][:5], m["params"][5], m.get("mk", 64)), search=[REF], clash=m["clash"])
    r = c.run_levels()
    digests = [hashlib.sha256("\n".join(sorted(c.spec.state_text(s, RAFT_ORDER) for s in lvl)).encode()).hexdigest()[:16]
               for lvl in r["level_states"]]
    return dict(distinct=r["distinct"], generated=r["generated"], depth=r["depth"], levels=r["levels"], verdict=r["verdict"],
                level_digests=digests)


if __name__ == "__main__":
    out = {}
    for name in RAFT_MODELS:
        out[name] = run_raft_text(name)
        print(name, {k: v for k, v in out[name].items() if k != "level_digests"}, flush=True)
    (ROOT / "tests" / "golden" / "raft_reference_text.json").write_text(json.dumps(out, indent=1) + "\n")
