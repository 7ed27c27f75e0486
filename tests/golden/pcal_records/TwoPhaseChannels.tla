--------------------------- MODULE TwoPhaseChannels ---------------------------
(* specs/pluscal/two_phase_channels.tla the way pcal2tla translates it: chan stays ONE variable, a function from process numbers to
   SEQUENCES of RECORDS (Append(chan[p], [type |-> ..., from |-> ...]), Head(chan[p]).type); msg and m stay record-valued.  Written by
   hand (tests/test_pcal.py compares it with the product's translation, which keeps chan as one sequence per field). *)
EXTENDS Naturals, Sequences, FiniteSets, TLC
CONSTANTS RM, Eager
VARIABLES chan, rmState, tmState, votes, pc, msg, next, m

vars == << chan, rmState, tmState, votes, pc, msg, next, m >>

ProcSet == {0} \cup (1..RM)

Init == /\ chan = [p \in 0..RM |-> <<>>]
        /\ rmState = [r \in 1..RM |-> "working"]
        /\ tmState = "init"
        /\ votes = {}
        /\ msg = [type |-> "none", from |-> 0]
        /\ next = 1
        /\ m = [self \in 1..RM |-> [type |-> "none", from |-> 0]]
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "Ask"
                                        [] self \in 1..RM -> "Wait"]

Ask == /\ pc[0] = "Ask"
       /\ IF next <= RM
             THEN /\ chan' = [chan EXCEPT ![next] = Append(chan[next], [type |-> "prepare", from |-> 0])]
                  /\ next' = next + 1
                  /\ pc' = [pc EXCEPT ![0] = "Ask"]
             ELSE /\ pc' = [pc EXCEPT ![0] = "Collect"]
                  /\ UNCHANGED << chan, next >>
       /\ UNCHANGED << rmState, tmState, votes, msg, m >>

Collect == /\ pc[0] = "Collect"
           /\ IF tmState = "init"
                 THEN /\ chan[0] # <<>>
                      /\ msg' = Head(chan[0])
                      /\ chan' = [chan EXCEPT ![0] = Tail(chan[0])]
                      /\ pc' = [pc EXCEPT ![0] = "Decide"]
                      /\ next' = next
                 ELSE /\ next' = 1
                      /\ pc' = [pc EXCEPT ![0] = "Tell"]
                      /\ UNCHANGED << chan, msg >>
           /\ UNCHANGED << rmState, tmState, votes, m >>

Decide == /\ pc[0] = "Decide"
          /\ IF msg.type = "no"
                THEN /\ tmState' = "aborted"
                     /\ votes' = votes
                ELSE /\ votes' = (votes \cup {msg.from})
                     /\ IF Eager \/ votes' = 1..RM
                           THEN /\ tmState' = "committed"
                           ELSE /\ tmState' = tmState
          /\ pc' = [pc EXCEPT ![0] = "Collect"]
          /\ UNCHANGED << chan, rmState, msg, next, m >>

Tell == /\ pc[0] = "Tell"
        /\ IF next <= RM
              THEN /\ chan' = [chan EXCEPT ![next] = Append(chan[next], [type |-> IF tmState = "committed" THEN "commit" ELSE "abort", from |-> 0])]
                   /\ next' = next + 1
                   /\ pc' = [pc EXCEPT ![0] = "Tell"]
              ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
                   /\ UNCHANGED << chan, next >>
        /\ UNCHANGED << rmState, tmState, votes, msg, m >>

TM == Ask \/ Collect \/ Decide \/ Tell

Wait(self) == /\ pc[self] = "Wait"
              /\ chan[self] # <<>>
              /\ m' = [m EXCEPT ![self] = Head(chan[self])]
              /\ chan' = [chan EXCEPT ![self] = Tail(chan[self])]
              /\ pc' = [pc EXCEPT ![self] = "Vote"]
              /\ UNCHANGED << rmState, tmState, votes, msg, next >>

Vote(self) == /\ pc[self] = "Vote"
              /\ \/ /\ rmState' = [rmState EXCEPT ![self] = "prepared"]
                    /\ chan' = [chan EXCEPT ![0] = Append(chan[0], [type |-> "yes", from |-> self])]
                 \/ /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                    /\ chan' = [chan EXCEPT ![0] = Append(chan[0], [type |-> "no", from |-> self])]
              /\ pc' = [pc EXCEPT ![self] = "Learn"]
              /\ UNCHANGED << tmState, votes, msg, next, m >>

Learn(self) == /\ pc[self] = "Learn"
               /\ chan[self] # <<>>
               /\ m' = [m EXCEPT ![self] = Head(chan[self])]
               /\ chan' = [chan EXCEPT ![self] = Tail(chan[self])]
               /\ pc' = [pc EXCEPT ![self] = "Act"]
               /\ UNCHANGED << rmState, tmState, votes, msg, next >>

Act(self) == /\ pc[self] = "Act"
             /\ IF m[self].type = "commit"
                   THEN /\ Assert(rmState[self] = "prepared", "Failure of assertion at line 74, column 7.")
                        /\ rmState' = [rmState EXCEPT ![self] = "committed"]
                   ELSE /\ IF rmState[self] = "prepared"
                              THEN /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                              ELSE /\ TRUE
                                   /\ UNCHANGED rmState
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << chan, tmState, votes, msg, next, m >>

R(self) == Wait(self) \/ Vote(self) \/ Learn(self) \/ Act(self)

Next == TM
           \/ (\E self \in 1..RM: R(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Consistent == \A a \in 1..RM : \A b \in 1..RM : ~(rmState[a] = "committed" /\ rmState[b] = "aborted")
CommitNeedsAllVotes == tmState = "committed" => votes = 1..RM
InboxHoldsVotes == \A k \in 1..Len(chan[0]) : chan[0][k].type \in {"yes", "no"} /\ chan[0][k].from \in 1..RM
FromTheCoordinator == \A r \in 1..RM : \A k \in 1..Len(chan[r]) : chan[r][k].from = 0 /\ chan[r][k].type \in {"prepare", "commit", "abort"}
AtMostTwoWaiting == \A p \in 0..RM : Len(chan[p]) <= RM
=============================================================================
