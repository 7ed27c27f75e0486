----------------------------- MODULE TwoPhaseSoup -----------------------------
(***************************************************************************)
(* HAND-WRITTEN translation of specs/pluscal/two_phase_soup.tla in the      *)
(* style of pcal2tla (p-manual App. B: two process declarations, a single   *)
(* process TM = 0 and a process set R \in 1..RM, pc initialised by CASE),   *)
(* written from the ALGORITHM text and evaluated by oracle/tlaplus.py; see  *)
(* EpochGc.tla.  msgs is the set of records the algorithm declares.         *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS RM, Hasty

VARIABLES rmState, tmState, tmPrepared, msgs, pc

vars == << rmState, tmState, tmPrepared, msgs, pc >>

ProcSet == {0} \cup (1..RM)

Init == /\ rmState = [r \in 1..RM |-> "working"]
        /\ tmState = "init"
        /\ tmPrepared = {}
        /\ msgs = {}
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "T"
                                        [] self \in 1..RM -> "W"]

Msg(t, r) == [type |-> t, rm |-> r]

T == /\ pc[0] = "T"
     /\ IF tmState = "init"
           THEN /\ \/ /\ \E m \in msgs :
                           /\ m.type = "prepared" /\ m.rm \notin tmPrepared
                           /\ tmPrepared' = tmPrepared \cup {m.rm}
                      /\ UNCHANGED << tmState, msgs >>
                   \/ /\ tmPrepared = 1..RM \/ (Hasty /\ tmPrepared # {})
                      /\ tmState' = "committed"
                      /\ msgs' = msgs \cup {Msg("commit", 0)}
                      /\ UNCHANGED tmPrepared
                   \/ /\ tmState' = "aborted"
                      /\ msgs' = msgs \cup {Msg("abort", 0)}
                      /\ UNCHANGED tmPrepared
                /\ pc' = [pc EXCEPT ![0] = "T"]
           ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
                /\ UNCHANGED << tmState, tmPrepared, msgs >>
     /\ UNCHANGED rmState

TM == T

W(self) == /\ pc[self] = "W"
           /\ \/ /\ rmState' = [rmState EXCEPT ![self] = "prepared"]
                 /\ msgs' = msgs \cup {Msg("prepared", self)}
              \/ /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                 /\ msgs' = msgs
           /\ pc' = [pc EXCEPT ![self] = "D"]
           /\ UNCHANGED << tmState, tmPrepared >>

D(self) == /\ pc[self] = "D"
           /\ IF rmState[self] = "prepared"
                 THEN \/ /\ Msg("commit", 0) \in msgs
                         /\ rmState' = [rmState EXCEPT ![self] = "committed"]
                      \/ /\ Msg("abort", 0) \in msgs
                         /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                 ELSE rmState' = rmState
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << tmState, tmPrepared, msgs >>

R(self) == W(self) \/ D(self)

Next == TM \/ (\E self \in 1..RM: R(self))
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Consistent == \A a \in 1..RM : \A b \in 1..RM : ~(rmState[a] = "committed" /\ rmState[b] = "aborted")
OneDecision == ~([type |-> "commit", rm |-> 0] \in msgs /\ [type |-> "abort", rm |-> 0] \in msgs)
PreparedWereSent == \A r \in tmPrepared : [type |-> "prepared", rm |-> r] \in msgs
KnownMessages == \A m \in msgs : (m.type = "prepared" /\ m.rm \in 1..RM) \/ (m.type \in {"commit", "abort"} /\ m.rm = 0)
SoupIsSmall == Cardinality(msgs) <= RM + 1
=============================================================================
