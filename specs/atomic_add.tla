------------------------------- MODULE atomic_add -------------------------------
(***************************************************************************)
(* The atomic-counter warm-up of tla-rust (two adders, one checker) with   *)
(* the hand-written TLA+ translation (p-manual.pdf App. B) the lowering in *)
(* tla_rust_amd/csrc/spec_pluscal.h follows.                               *)
(***************************************************************************)
EXTENDS Naturals

(* --algorithm atomic_add

\* Simple warmup for ensuring that the global_counter
\* always reaches an intended state.

variables global_counter = 0

process AdderProc \in 1..2
begin
Increment:
  global_counter := global_counter + 1;
end process

process Checker = 3
begin
Check:
    await global_counter = 2;
end process

end algorithm *)

\* BEGIN TRANSLATION
VARIABLES global_counter, pc

vars == << global_counter, pc >>

ProcSet == (1..2) \cup {3}

Init == (* Global variables *)
        /\ global_counter = 0
        /\ pc = [self \in ProcSet |-> CASE self \in 1..2 -> "Increment"
                                        [] self = 3 -> "Check"]

Increment(self) == /\ pc[self] = "Increment"
                   /\ global_counter' = global_counter + 1
                   /\ pc' = [pc EXCEPT ![self] = "Done"]

AdderProc(self) == Increment(self)

Check == /\ pc[3] = "Check"
         /\ global_counter = 2
         /\ pc' = [pc EXCEPT ![3] = "Done"]
         /\ UNCHANGED global_counter

Checker == Check

Next == Checker
           \/ (\E self \in 1..2: AdderProc(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
=============================================================================
