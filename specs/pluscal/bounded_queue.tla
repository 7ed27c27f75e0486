---------------------------- MODULE bounded_queue ----------------------------
(***************************************************************************)
(* One producer, one consumer and a bounded FIFO queue held in a sequence   *)
(* variable (<<>>, Append, Head, Tail, Len, q[i]).  With TWO consumers     *)
(* (specs/pluscal/bounded_queue_race.cfg) taking an item and using it are  *)
(* two steps, so items can be used out of order: the assert in Use fails   *)
(* after 10 states.                                                        *)
(***************************************************************************)
EXTENDS Naturals, Sequences
CONSTANTS Items, MaxQ, Consumers

(* --algorithm bounded_queue
variables queue = <<>>, produced = 0, taken = 0, last = 0;

process Producer = 0
begin
  P:
    while produced < Items do
      Put:
        await Len(queue) < MaxQ;
        queue := Append(queue, produced + 1);
        produced := produced + 1;
    end while;
end process

process Consumer \in 1..Consumers
  variables item = 0;
begin
  C:
    while taken < Items do
      Get:
        await queue # <<>>;
        item := Head(queue);
        queue := Tail(queue);
        taken := taken + 1;
      Use:
        assert item > last;
        last := item;
    end while;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES queue, produced, taken, last, pc, item

vars == << queue, produced, taken, last, pc, item >>

ProcSet == {0} \cup (1..Consumers)

Init == (* Global variables *)
        /\ queue = <<>>
        /\ produced = 0
        /\ taken = 0
        /\ last = 0
        (* Process Consumer *)
        /\ item = [self \in 1..Consumers |-> 0]
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "P"
                                        [] self \in 1..Consumers -> "C"]

P == /\ pc[0] = "P"
     /\ IF produced < Items
           THEN /\ pc' = [pc EXCEPT ![0] = "Put"]
           ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
     /\ UNCHANGED << queue, produced, taken, last, item >>

Put == /\ pc[0] = "Put"
       /\ Len(queue) < MaxQ
       /\ queue' = Append(queue, produced + 1)
       /\ produced' = produced + 1
       /\ pc' = [pc EXCEPT ![0] = "P"]
       /\ UNCHANGED << taken, last, item >>

Producer == P \/ Put

C(self) == /\ pc[self] = "C"
           /\ IF taken < Items
                 THEN /\ pc' = [pc EXCEPT ![self] = "Get"]
                 ELSE /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << queue, produced, taken, last, item >>

Get(self) == /\ pc[self] = "Get"
             /\ queue # <<>>
             /\ item' = [item EXCEPT ![self] = Head(queue)]
             /\ queue' = Tail(queue)
             /\ taken' = taken + 1
             /\ pc' = [pc EXCEPT ![self] = "Use"]
             /\ UNCHANGED << produced, last >>

Use(self) == /\ pc[self] = "Use"
             /\ Assert(item[self] > last, 
                       "Failure of assertion at line 37, column 9.")
             /\ last' = item[self]
             /\ pc' = [pc EXCEPT ![self] = "C"]
             /\ UNCHANGED << queue, produced, taken, item >>

Consumer(self) == C(self) \/ Get(self) \/ Use(self)

Next == Producer
           \/ (\E self \in 1..Consumers: Consumer(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Bounded == Len(queue) <= MaxQ
Fifo == \A i \in 1..Len(queue) : queue[i] = taken + i
=============================================================================
