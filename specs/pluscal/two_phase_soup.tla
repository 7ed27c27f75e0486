--------------------------- MODULE two_phase_soup ---------------------------
(***************************************************************************)
(* Two-phase commit with a MESSAGE SOUP: msgs is a SET of RECORDS, a       *)
(* message once sent stays and can be received any number of times — the   *)
(* way Lamport's TwoPhase and the reference's Paxos modules model a        *)
(* network (examples/Paxos/Paxos.tla: `msgs`), here in PlusCal, and the    *)
(* counterpart of two_phase_channels.tla.  Resource managers prepare or    *)
(* abort on their own; the transaction manager notes the "prepared"        *)
(* messages one at a time, commits once it has them all, and may abort any *)
(* time before.  Hasty = TRUE lets it commit on the first one.             *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS RM, Hasty

(* --algorithm two_phase_soup
variables rmState = [r \in 1..RM |-> "working"],
          tmState = "init",
          tmPrepared = {},
          msgs = {};

process TM = 0
begin
  T:
    while tmState = "init" do
      either
        with m \in msgs do
          await m.type = "prepared" /\ m.rm \notin tmPrepared;
          tmPrepared := tmPrepared \cup {m.rm};
        end with;
      or
        await tmPrepared = 1..RM \/ (Hasty /\ tmPrepared # {});
        tmState := "committed";
        msgs := msgs \cup {[type |-> "commit", rm |-> 0]};
      or
        tmState := "aborted";
        msgs := msgs \cup {[type |-> "abort", rm |-> 0]};
      end either;
    end while;
end process

process R \in 1..RM
begin
  W:
    either
      rmState[self] := "prepared";
      msgs := msgs \cup {[type |-> "prepared", rm |-> self]};
    or
      rmState[self] := "aborted";
    end either;
  D:
    if rmState[self] = "prepared" then
      either
        await [type |-> "commit", rm |-> 0] \in msgs;
        rmState[self] := "committed";
      or
        await [type |-> "abort", rm |-> 0] \in msgs;
        rmState[self] := "aborted";
      end either;
    end if;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES rmState, tmState, tmPrepared, msgs, pc

vars == << rmState, tmState, tmPrepared, msgs, pc >>

ProcSet == {0} \cup (1..RM)

Init == (* Global variables *)
        /\ rmState = [r \in 1..RM |-> "working"]
        /\ tmState = "init"
        /\ tmPrepared = {}
        /\ msgs = {}
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "T"
                                        [] self \in 1..RM -> "W"]

T == /\ pc[0] = "T"
     /\ IF tmState = "init"
           THEN /\ \/ /\ \E m \in msgs:
                           /\ m.type = "prepared" /\ m.rm \notin tmPrepared
                           /\ tmPrepared' = tmPrepared \cup {m.rm}
                      /\ UNCHANGED << tmState, msgs >>
                   \/ /\ tmPrepared = 1..RM \/ (Hasty /\ tmPrepared # {})
                      /\ tmState' = "committed"
                      /\ msgs' = msgs \cup {[type |-> "commit", rm |-> 0]}
                      /\ UNCHANGED tmPrepared
                   \/ /\ tmState' = "aborted"
                      /\ msgs' = msgs \cup {[type |-> "abort", rm |-> 0]}
                      /\ UNCHANGED tmPrepared
                /\ pc' = [pc EXCEPT ![0] = "T"]
           ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
                /\ UNCHANGED << tmState, tmPrepared, msgs >>
     /\ UNCHANGED rmState

TM == T

W(self) == /\ pc[self] = "W"
           /\ \/ /\ rmState' = [rmState EXCEPT ![self] = "prepared"]
                 /\ msgs' = msgs \cup {[type |-> "prepared", rm |-> self]}
              \/ /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                 /\ UNCHANGED msgs
           /\ pc' = [pc EXCEPT ![self] = "D"]
           /\ UNCHANGED << tmState, tmPrepared >>

D(self) == /\ pc[self] = "D"
           /\ IF rmState[self] = "prepared"
                 THEN /\ \/ /\ [type |-> "commit", rm |-> 0] \in msgs
                            /\ rmState' = [rmState EXCEPT ![self] = "committed"]
                         \/ /\ [type |-> "abort", rm |-> 0] \in msgs
                            /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                 ELSE /\ TRUE
                      /\ UNCHANGED rmState
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << tmState, tmPrepared, msgs >>

R(self) == W(self) \/ D(self)

Next == TM
           \/ (\E self \in 1..RM: R(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Consistent == \A a \in 1..RM : \A b \in 1..RM : ~(rmState[a] = "committed" /\ rmState[b] = "aborted")
OneDecision == ~([type |-> "commit", rm |-> 0] \in msgs /\ [type |-> "abort", rm |-> 0] \in msgs)
PreparedWereSent == \A r \in tmPrepared : [type |-> "prepared", rm |-> r] \in msgs
KnownMessages == \A m \in msgs : (m.type = "prepared" /\ m.rm \in 1..RM) \/ (m.type \in {"commit", "abort"} /\ m.rm = 0)
SoupIsSmall == Cardinality(msgs) <= RM + 1
=============================================================================
