------------------------- MODULE growing_counters -------------------------
(***************************************************************************)
(* An algorithm whose state space is INFINITE: the counters only grow.     *)
(* It can be checked only under a cfg CONSTRAINT (growing_counters.cfg:    *)
(* CONSTRAINT Small), the way the reference's model wrappers bound their   *)
(* specs (examples/SpecifyingSystems/TLC/MCAlternatingBit.cfg:8,           *)
(* FIFO/MCInnerFIFO.cfg:23-26): a state outside the constraint is          *)
(* generated and checked, but neither stored nor expanded.                 *)
(***************************************************************************)
EXTENDS Naturals
CONSTANT Bound

(* --algorithm growing_counters
variables produced = 0, consumed = 0;
define
  NeverAhead == consumed <= produced
  Small      == produced <= Bound
end define;
process producer = 1
begin
P: while TRUE do
     produced := produced + 1;
   end while;
end process;
process consumer \in 2..3
variable mine = 0;
begin
C: while TRUE do
     await consumed < produced;
     consumed := consumed + 1;
     mine := mine + 1;
   end while;
end process;
end algorithm *)
\* BEGIN TRANSLATION
VARIABLES produced, consumed, pc

(* define statement *)
NeverAhead == consumed <= produced

Small == produced <= Bound

VARIABLES mine

vars == << produced, consumed, pc, mine >>

ProcSet == {1} \cup (2..3)

Init == (* Global variables *)
        /\ produced = 0
        /\ consumed = 0
        (* Process consumer *)
        /\ mine = [self \in 2..3 |-> 0]
        /\ pc = [self \in ProcSet |-> CASE self = 1 -> "P"
                                        [] self \in 2..3 -> "C"]

P == /\ pc[1] = "P"
     /\ produced' = produced + 1
     /\ pc' = [pc EXCEPT ![1] = "P"]
     /\ UNCHANGED << consumed, mine >>

producer == P

C(self) == /\ pc[self] = "C"
           /\ consumed < produced
           /\ consumed' = consumed + 1
           /\ mine' = [mine EXCEPT ![self] = mine[self] + 1]
           /\ pc' = [pc EXCEPT ![self] = "C"]
           /\ UNCHANGED produced

consumer(self) == C(self)

Next == producer
           \/ (\E self \in 2..3: consumer(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
=============================================================================
