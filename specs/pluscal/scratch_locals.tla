--------------------------- MODULE scratch_locals ---------------------------
(***************************************************************************)
(* Variables declared WITHOUT an initial value (`variable tmp;`): the      *)
(* translation initialises them to the model value defaultInitValue        *)
(* (p-manual section 3.3; TLC's cfg then needs                             *)
(* `defaultInitValue = defaultInitValue`, implied here).  Two workers do   *)
(* read-modify-write on a shared cell through an uninitialised scratch     *)
(* local, a third variable is never assigned at all.                       *)
(***************************************************************************)
EXTENDS Naturals
CONSTANT N

(* --algorithm scratch_locals
variables cell = 0, winner, spare;
process worker \in 1..N
variables tmp, seen;
begin
R: tmp := cell;
   seen := (cell > 0);
W: cell := tmp + 1;
   winner := self;
end process;
end algorithm *)
\* BEGIN TRANSLATION
CONSTANT defaultInitValue
VARIABLES cell, winner, spare, pc, tmp, seen

vars == << cell, winner, spare, pc, tmp, seen >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ cell = 0
        /\ winner = defaultInitValue
        /\ spare = defaultInitValue
        (* Process worker *)
        /\ tmp = [self \in 1..N |-> defaultInitValue]
        /\ seen = [self \in 1..N |-> defaultInitValue]
        /\ pc = [self \in ProcSet |-> "R"]

R(self) == /\ pc[self] = "R"
           /\ tmp' = [tmp EXCEPT ![self] = cell]
           /\ seen' = [seen EXCEPT ![self] = (cell > 0)]
           /\ pc' = [pc EXCEPT ![self] = "W"]
           /\ UNCHANGED << cell, winner, spare >>

W(self) == /\ pc[self] = "W"
           /\ cell' = tmp[self] + 1
           /\ winner' = self
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << spare, tmp, seen >>

worker(self) == R(self) \/ W(self)

Next == (\E self \in 1..N: worker(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

AtMostN == cell <= N
=============================================================================
