------------------------ MODULE two_phase_channels ------------------------
(***************************************************************************)
(* Two-phase commit over FIFO channels: the message-passing shape of the    *)
(* reference's roadmap ("a lock-free distributed transaction protocol",    *)
(* README.md:26-42).  chan is an ARRAY of SEQUENCES of RECORDS: chan[0] is *)
(* the coordinator's inbox, chan[r] resource manager r's; a message is a   *)
(* record [type, from].  The coordinator asks every resource manager to    *)
(* prepare, collects the votes one message at a time, decides, and tells   *)
(* everybody.  Eager = TRUE lets it commit on the FIRST yes vote: a        *)
(* manager that voted no then receives a commit (the assert in Act fails)  *)
(* and Consistent breaks.                                                  *)
(***************************************************************************)
EXTENDS Naturals, Sequences, FiniteSets
CONSTANTS RM, Eager

(* --algorithm two_phase_channels
variables chan = [p \in 0..RM |-> <<>>],
          rmState = [r \in 1..RM |-> "working"],
          tmState = "init",
          votes = {};

process TM = 0
  variables msg = [type |-> "none", from |-> 0], next = 1;
begin
  Ask:
    while next <= RM do
      chan[next] := Append(chan[next], [type |-> "prepare", from |-> 0]);
      next := next + 1;
    end while;
  Collect:
    while tmState = "init" do
      await chan[0] # <<>>;
      msg := Head(chan[0]);
      chan[0] := Tail(chan[0]);
      Decide:
        if msg.type = "no" then
          tmState := "aborted";
        else
          votes := votes \cup {msg.from};
          if Eager \/ votes = 1..RM then
            tmState := "committed";
          end if;
        end if;
    end while;
    next := 1;
  Tell:
    while next <= RM do
      chan[next] := Append(chan[next], [type |-> IF tmState = "committed" THEN "commit" ELSE "abort", from |-> 0]);
      next := next + 1;
    end while;
end process

process R \in 1..RM
  variables m = [type |-> "none", from |-> 0];
begin
  Wait:
    await chan[self] # <<>>;
    m := Head(chan[self]);
    chan[self] := Tail(chan[self]);
  Vote:
    either
      rmState[self] := "prepared";
      chan[0] := Append(chan[0], [type |-> "yes", from |-> self]);
    or
      rmState[self] := "aborted";
      chan[0] := Append(chan[0], [type |-> "no", from |-> self]);
    end either;
  Learn:
    await chan[self] # <<>>;
    m := Head(chan[self]);
    chan[self] := Tail(chan[self]);
  Act:
    if m.type = "commit" then
      assert rmState[self] = "prepared";
      rmState[self] := "committed";
    elsif rmState[self] = "prepared" then
      rmState[self] := "aborted";
    end if;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES chan_type, chan_from, rmState, tmState, votes, pc, msg_type, msg_from, next, m_type, m_from

vars == << chan_type, chan_from, rmState, tmState, votes, pc, msg_type, msg_from, next, m_type, m_from >>

(* record variables are kept field by field: r.f is r_f *)
chan == [p \in 0..RM |-> [n_ \in 1..Len(chan_type[p]) |-> [type |-> chan_type[p][n_], from |-> chan_from[p][n_]]]]
msg == [type |-> msg_type, from |-> msg_from]
m == [self \in 1..RM |-> [type |-> m_type[self], from |-> m_from[self]]]

ProcSet == {0} \cup (1..RM)

Init == (* Global variables *)
        /\ chan_type = [p \in 0..RM |-> <<>>]
        /\ chan_from = [p \in 0..RM |-> <<>>]
        /\ rmState = [r \in 1..RM |-> "working"]
        /\ tmState = "init"
        /\ votes = {}
        (* Process TM *)
        /\ msg_type = "none"
        /\ msg_from = 0
        /\ next = 1
        (* Process R *)
        /\ m_type = [self \in 1..RM |-> "none"]
        /\ m_from = [self \in 1..RM |-> 0]
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "Ask"
                                        [] self \in 1..RM -> "Wait"]

Ask == /\ pc[0] = "Ask"
       /\ IF next <= RM
             THEN /\ chan_type' = [chan_type EXCEPT ![next] = Append(chan_type[next], "prepare")]
                  /\ chan_from' = [chan_from EXCEPT ![next] = Append(chan_from[next], 0)]
                  /\ next' = next + 1
                  /\ pc' = [pc EXCEPT ![0] = "Ask"]
             ELSE /\ pc' = [pc EXCEPT ![0] = "Collect"]
                  /\ UNCHANGED << chan_type, chan_from, next >>
       /\ UNCHANGED << rmState, tmState, votes, msg_type, msg_from, m_type, 
                       m_from >>

Collect == /\ pc[0] = "Collect"
           /\ IF tmState = "init"
                 THEN /\ chan_type[0] # <<>>
                      /\ msg_type' = Head(chan_type[0])
                      /\ msg_from' = Head(chan_from[0])
                      /\ chan_type' = [chan_type EXCEPT ![0] = Tail(chan_type[0])]
                      /\ chan_from' = [chan_from EXCEPT ![0] = Tail(chan_from[0])]
                      /\ pc' = [pc EXCEPT ![0] = "Decide"]
                      /\ UNCHANGED next
                 ELSE /\ next' = 1
                      /\ pc' = [pc EXCEPT ![0] = "Tell"]
                      /\ UNCHANGED << chan_type, chan_from, msg_type, 
                                      msg_from >>
           /\ UNCHANGED << rmState, tmState, votes, m_type, m_from >>

Decide == /\ pc[0] = "Decide"
          /\ IF msg_type = "no"
                THEN /\ tmState' = "aborted"
                     /\ UNCHANGED votes
                ELSE /\ votes' = votes \cup {msg_from}
                     /\ IF Eager \/ votes' = 1..RM
                           THEN /\ tmState' = "committed"
                           ELSE /\ TRUE
                                /\ UNCHANGED tmState
          /\ pc' = [pc EXCEPT ![0] = "Collect"]
          /\ UNCHANGED << chan_type, chan_from, rmState, msg_type, msg_from, 
                          next, m_type, m_from >>

Tell == /\ pc[0] = "Tell"
        /\ IF next <= RM
              THEN /\ chan_type' = [chan_type EXCEPT ![next] = Append(chan_type[next], IF tmState = "committed" THEN "commit" ELSE "abort")]
                   /\ chan_from' = [chan_from EXCEPT ![next] = Append(chan_from[next], 0)]
                   /\ next' = next + 1
                   /\ pc' = [pc EXCEPT ![0] = "Tell"]
              ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
                   /\ UNCHANGED << chan_type, chan_from, next >>
        /\ UNCHANGED << rmState, tmState, votes, msg_type, msg_from, m_type, 
                        m_from >>

TM == Ask \/ Collect \/ Decide \/ Tell

Wait(self) == /\ pc[self] = "Wait"
              /\ chan_type[self] # <<>>
              /\ m_type' = [m_type EXCEPT ![self] = Head(chan_type[self])]
              /\ m_from' = [m_from EXCEPT ![self] = Head(chan_from[self])]
              /\ chan_type' = [chan_type EXCEPT ![self] = Tail(chan_type[self])]
              /\ chan_from' = [chan_from EXCEPT ![self] = Tail(chan_from[self])]
              /\ pc' = [pc EXCEPT ![self] = "Vote"]
              /\ UNCHANGED << rmState, tmState, votes, msg_type, msg_from, 
                              next >>

Vote(self) == /\ pc[self] = "Vote"
              /\ \/ /\ rmState' = [rmState EXCEPT ![self] = "prepared"]
                    /\ chan_type' = [chan_type EXCEPT ![0] = Append(chan_type[0], "yes")]
                    /\ chan_from' = [chan_from EXCEPT ![0] = Append(chan_from[0], self)]
                 \/ /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                    /\ chan_type' = [chan_type EXCEPT ![0] = Append(chan_type[0], "no")]
                    /\ chan_from' = [chan_from EXCEPT ![0] = Append(chan_from[0], self)]
              /\ pc' = [pc EXCEPT ![self] = "Learn"]
              /\ UNCHANGED << tmState, votes, msg_type, msg_from, next, 
                              m_type, m_from >>

Learn(self) == /\ pc[self] = "Learn"
               /\ chan_type[self] # <<>>
               /\ m_type' = [m_type EXCEPT ![self] = Head(chan_type[self])]
               /\ m_from' = [m_from EXCEPT ![self] = Head(chan_from[self])]
               /\ chan_type' = [chan_type EXCEPT ![self] = Tail(chan_type[self])]
               /\ chan_from' = [chan_from EXCEPT ![self] = Tail(chan_from[self])]
               /\ pc' = [pc EXCEPT ![self] = "Act"]
               /\ UNCHANGED << rmState, tmState, votes, msg_type, msg_from, 
                               next >>

Act(self) == /\ pc[self] = "Act"
             /\ IF m_type[self] = "commit"
                   THEN /\ Assert(rmState[self] = "prepared", 
                                  "Failure of assertion at line 74, column 7.")
                        /\ rmState' = [rmState EXCEPT ![self] = "committed"]
                   ELSE /\ IF rmState[self] = "prepared"
                              THEN /\ rmState' = [rmState EXCEPT ![self] = "aborted"]
                              ELSE /\ TRUE
                                   /\ UNCHANGED rmState
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << chan_type, chan_from, tmState, votes, msg_type, 
                             msg_from, next, m_type, m_from >>

R(self) == Wait(self) \/ Vote(self) \/ Learn(self) \/ Act(self)

Next == TM
           \/ (\E self \in 1..RM: R(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Consistent == \A a \in 1..RM : \A b \in 1..RM : ~(rmState[a] = "committed" /\ rmState[b] = "aborted")
CommitNeedsAllVotes == tmState = "committed" => votes = 1..RM
InboxHoldsVotes == \A k \in 1..Len(chan[0]) : chan[0][k].type \in {"yes", "no"} /\ chan[0][k].from \in 1..RM
FromTheCoordinator == \A r \in 1..RM : \A k \in 1..Len(chan[r]) : chan[r][k].from = 0 /\ chan[r][k].type \in {"prepare", "commit", "abort"}
AtMostTwoWaiting == \A p \in 0..RM : Len(chan[p]) <= RM
=============================================================================
