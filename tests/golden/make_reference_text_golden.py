"""Makes tests/golden/raft_reference_text.json and tests/golden/ssi_reference_text.json: the reference's OWN spec texts
(/root/reference/examples/raft.tla, serializableSnapshotIsolation.tla, textbookSnapshotIsolation.tla, read where they lie)
evaluated by oracle/tlaplus.py under the model wrappers of specs/.  Run in the build container (the GPU box has no
/root/reference):  python tests/golden/make_reference_text_golden.py [raft] [ssi]
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
    # THREE servers (majority quorums are no longer "everybody"; the bench model's MCraft.cfg with a smaller bag bound): 4 message
    # keys are what one election needs, the 5th and 6th are the first AppendEntries request / response.  58 s / 4 min / 15 min of
    # evaluation: fixtures only, not re-evaluated by the suite
    "raft_3s_keys4": dict(params=[3, 4, 2, 3, 1, 3], clash="test", mk=4, slow=True),
    "raft_3s_keys5": dict(params=[3, 4, 2, 3, 1, 3], clash="test", mk=5, slow=True),
    "raft_3s_keys6": dict(params=[3, 4, 2, 3, 1, 3], clash="test", mk=6, slow=True),
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


# ---------------------------------------------------------------------------------------------- snapshot isolation
SSI_ORDER = ["history", "holdingXLocks", "waitingForXLock", "inConflict", "outConflict", "holdingSIREADlocks"]
TEXTBOOK_ORDER = ["history", "holdingXLocks", "waitingForXLock"]
SSI_INVARIANTS = ["TypeInv", "WellFormed", "CorrectnessOfHoldingXLocks", "CorrectnessOfWaitingForXLock", "CorrectReadView",
                  "FirstCommitterWins", "CahillOK", "BernsteinOK"]
# params = the C oracle's / the lowering's {nTxn, nKey, invariant mask, find, textbook}
SSI_MODELS = {
    "ssi_2x1": dict(params=[2, 1, 127, 0], module="MCssi"),                 # 569 distinct
    "ssi_2x2": dict(params=[2, 2, 127, 0], module="MCssi"),                 # 29 629 distinct / 50 121 generated / depth 13
    "ssi_3x1": dict(params=[3, 1, 127, 0], module="MCssi"),                 # 90 430 distinct
    "textbook_2x2": dict(params=[2, 2, 31, 0, 1], module="MCtextbookSI"),   # textbookSnapshotIsolation.tla, the SI-level invariants
}


def ssi_cfg(nt, nk, invariants):
    return (f"INIT Init\nNEXT Next\nCONSTANTS\n  TxnId = {{{', '.join(f'T{i + 1}' for i in range(nt))}}}\n"
            f"  Key = {{{', '.join(f'K{i + 1}' for i in range(nk))}}}\n  NoLock = NoLock\nINVARIANTS {' '.join(invariants)}\n")


def run_ssi_text(name):
    import tlaplus as T
    m = SSI_MODELS[name]
    textbook = len(m["params"]) > 4 and m["params"][4]
    invs = [i for i in SSI_INVARIANTS if not (textbook and i in ("CahillOK", "BernsteinOK"))]  # textbook SI is not serializable
    c = T.Checker(ROOT / "specs" / f"{m['module']}.tla", cfg_text=ssi_cfg(m["params"][0], m["params"][1], invs), search=[REF])
    r = c.run_levels()
    order = TEXTBOOK_ORDER if textbook else SSI_ORDER
    digests = [hashlib.sha256("\n".join(sorted(c.spec.state_text(s, order) for s in lvl)).encode()).hexdigest()[:16]
               for lvl in r["level_states"]]
    return dict(distinct=r["distinct"], generated=r["generated"], depth=r["depth"], levels=r["levels"], verdict=r["verdict"],
                level_digests=digests)


# ---------------------------------------------------------------------------------------------- Paxos family
PAXOS_REF = REF / "Paxos"
VOTING_ORDER = ["votes", "maxBal"]
PAXOS_ORDER = ["maxBal", "maxVBal", "maxVal", "msgs"]
# params = the C oracle's / the lowering's {kind (0 Paxos, 1 Voting), nAcceptor, nValue, nBallot, invariant mask, symmetry, property}
# tla: the reference's own model module (read where it lies) or a wrapper of specs/paxos/ with the sizes MCPaxos.tla:7-9 names
PAXOS_MODELS = {
    "voting_mc": dict(params=[1, 3, 2, 2, 1, 3, 1], tla=PAXOS_REF / "MCVoting.tla", sym=True),          # MCVoting.cfg as committed
    "voting_mc_nosym": dict(params=[1, 3, 2, 2, 1, 0, 1], tla=PAXOS_REF / "MCVoting.tla", sym=False),
    "paxos_mc": dict(params=[0, 1, 1, 2, 15, 3, 1], tla=PAXOS_REF / "MCPaxos.tla", sym=True),          # MCPaxos.cfg as committed (1 x 1)
    "voting_3x2_b3": dict(params=[1, 3, 2, 3, 1, 3, 1], tla=ROOT / "specs" / "paxos" / "MCVoting3.tla", sym=True),    # ballots 0..2
    "paxos_3x2": dict(params=[0, 3, 2, 2, 15, 3, 1], tla=ROOT / "specs" / "paxos" / "MCPaxos3.tla", sym=True),
    "paxos_3x2_nosym": dict(params=[0, 3, 2, 2, 15, 0, 1], tla=ROOT / "specs" / "paxos" / "MCPaxos3.tla", sym=False, slow=True),  # 30 s
    "paxos_3x2_b3": dict(params=[0, 3, 2, 3, 15, 3, 1], tla=ROOT / "specs" / "paxos" / "MCPaxos3.tla", sym=True, max_ballot=2, slow=True),
}


# negative controls: wrappers of specs/paxos/ in which something the cfg checks must FAIL; params as above ([7..] = quorum masks)
PAXOS_NEGATIVE = {
    "voting_badquorum": dict(params=[1, 3, 2, 2, 1, 0, 1, 3, 1, 2, 4], tla=ROOT / "specs" / "paxos" / "MCVotingBadQuorum.tla"),
    "paxos_bad_phase2a": dict(params=[0, 3, 2, 2, 15, 0, 3], tla=ROOT / "specs" / "paxos" / "MCPaxosBad.tla"),
}


def run_paxos_negative(name):
    """verdict of the evaluator on the wrapper's own cfg: what is violated (an INVARIANT by its cfg position, or the PROPERTY,
    reported as index = number of invariants + its position, the convention of the oracle and the lowering) and the length of
    the shortest counterexample"""
    import tlaplus as T
    c = T.Checker(PAXOS_NEGATIVE[name]["tla"], search=[PAXOS_REF])
    r = c.run_levels(check_deadlock=False)
    idx = r["violated_invariant"] + (len(c.invs) if r["verdict"] == "property" else 0)
    return dict(verdict=r["verdict"], index=idx, trace_len=r["trace_len"],
                name=(c.cfg["properties"] if r["verdict"] == "property" else c.cfg["invariants"])[r["violated_invariant"]])


# the two alternative configurations examples/Paxos/MCVoting.cfg:7-8 names in its comments (SPECIFICATION Spec \* MCSpec,
# INVARIANT Inv \* MCInv) and MCVoting.tla:36-55 explains: MCSpec == TypeOK /\ [][FALSE]_vars with MCInv = the statements of five
# THEOREMs of Voting.tla on EVERY type-correct state; MCSpecI == Inv /\ [][Next]_vars with Inv = "Inv is inductive"
VOTING_ALTERNATIVES = {
    "voting_theorems_on_all_type_correct_states": dict(spec="MCSpec", inv="MCInv", slow=True),   # 110 592 states, 66 s
    "voting_inv_is_inductive": dict(spec="MCSpecI", inv="Inv"),                                  # 2 771 states satisfy Inv, 20 s
}


def run_voting_alternative(name):
    import tlaplus as T
    a = VOTING_ALTERNATIVES[name]
    cfg = (PAXOS_REF / "MCVoting.cfg").read_text()
    head = cfg[:cfg.index("SPECIFICATION")]     # the CONSTANTS block as committed
    c = T.Checker(PAXOS_REF / "MCVoting.tla", cfg_text=head + f"SPECIFICATION {a['spec']}\nINVARIANT {a['inv']}\n", search=[PAXOS_REF])
    r = c.run_levels(check_deadlock=False)
    return dict(distinct=r["distinct"], generated=r["generated"], depth=r["depth"], levels=r["levels"], verdict=r["verdict"])


def run_paxos_text(name):
    """deadlock checking off: Voting with a finite Ballot set ends in states without successors (every acceptor at the last
    ballot); the reference's cfg files say nothing about it and TLC would need -deadlock to finish the run"""
    import tlaplus as T
    m = PAXOS_MODELS[name]
    tla = Path(m["tla"])
    if m.get("max_ballot"):  # same wrapper, MCMaxBallot edited: evaluated from a scratch copy beside nothing, found through `search`
        import tempfile
        d = Path(tempfile.mkdtemp())
        text = tla.read_text().replace("MCMaxBallot == 1", f"MCMaxBallot == {m['max_ballot']}")
        (d / tla.name).write_text(text)
        (d / (tla.stem + ".cfg")).write_text(tla.with_suffix(".cfg").read_text())
        tla = d / tla.name
    c = T.Checker(tla, search=[PAXOS_REF], symmetry=m["sym"])
    r = c.run_levels(check_deadlock=False)
    out = dict(distinct=r["distinct"], generated=r["generated"], depth=r["depth"], levels=r["levels"], verdict=r["verdict"])
    if not m["sym"]:  # under SYMMETRY the kept representative is whichever state of the orbit is met first: counts only
        order = VOTING_ORDER if m["params"][0] == 1 else PAXOS_ORDER
        out["level_digests"] = [hashlib.sha256("\n".join(sorted(c.spec.state_text(s, order) for s in lvl)).encode()).hexdigest()[:16]
                                for lvl in r["level_states"]]
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["raft", "ssi", "paxos"]
    if "paxos" in which:
        ppath = ROOT / "tests" / "golden" / "paxos_reference_text.json"
        out = json.loads(ppath.read_text()) if ppath.exists() else {}
        only = which[1:] if which[0] == "paxos" and len(which) > 1 else None      # ... paxos name name: only the named entries
        for name in PAXOS_MODELS:
            if only and name not in only:
                continue
            out[name] = run_paxos_text(name)
            print(name, {k: v for k, v in out[name].items() if k != "level_digests"}, flush=True)
        for name in PAXOS_NEGATIVE:
            if only and name not in only:
                continue
            out[name] = run_paxos_negative(name)
            print(name, out[name], flush=True)
        for name in VOTING_ALTERNATIVES:
            if only and name not in only:
                continue
            out[name] = run_voting_alternative(name)
            print(name, out[name], flush=True)
        (ROOT / "tests" / "golden" / "paxos_reference_text.json").write_text(json.dumps(out, indent=1) + "\n")
    if "raft" in which:
        path = ROOT / "tests" / "golden" / "raft_reference_text.json"
        out = json.loads(path.read_text()) if path.exists() else {}
        for name in RAFT_MODELS:
            if len(which) > 1 and which[0] == "raft" and name not in which[1:]:
                continue      # python make_reference_text_golden.py raft raft_3s_keys5 ...: only the named models
            out[name] = run_raft_text(name)
            print(name, {k: v for k, v in out[name].items() if k != "level_digests"}, flush=True)
        (ROOT / "tests" / "golden" / "raft_reference_text.json").write_text(json.dumps(out, indent=1) + "\n")
    if "ssi" in which:
        path = ROOT / "tests" / "golden" / "ssi_reference_text.json"
        out = json.loads(path.read_text()) if path.exists() else {}
        for name in SSI_MODELS:
            if len(which) > 1 and which[0] == "ssi" and name not in which[1:]:
                continue
            out[name] = run_ssi_text(name)
            print(name, {k: v for k, v in out[name].items() if k != "level_digests"}, flush=True)
            path.write_text(json.dumps(out, indent=1) + "\n")
