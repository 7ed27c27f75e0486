----------------------------- MODULE atomic_add_n -----------------------------
(***************************************************************************)
(* atomic_add generalised to N adders and one checker that awaits N — the  *)
(* "synthetic N-process atomic-counter spec" used for throughput series    *)
(* (2^N + 1 distinct states, N*2^(N-1) + 3 generated, depth N + 2).        *)
(***************************************************************************)
EXTENDS Naturals
CONSTANT N

(* --algorithm atomic_add_n
variables global_counter = 0

process AdderProc \in 1..N
begin
Increment:
  global_counter := global_counter + 1;
end process

process Checker = N + 1
begin
Check:
    await global_counter = N;
end process

end algorithm *)

\* BEGIN TRANSLATION
VARIABLES global_counter, pc

vars == << global_counter, pc >>

ProcSet == (1..N) \cup {N + 1}

Init == (* Global variables *)
        /\ global_counter = 0
        /\ pc = [self \in ProcSet |-> CASE self \in 1..N -> "Increment"
                                        [] self = N + 1 -> "Check"]

Increment(self) == /\ pc[self] = "Increment"
                   /\ global_counter' = global_counter + 1
                   /\ pc' = [pc EXCEPT ![self] = "Done"]

AdderProc(self) == Increment(self)

Check == /\ pc[N + 1] = "Check"
         /\ global_counter = N
         /\ pc' = [pc EXCEPT ![N + 1] = "Done"]
         /\ UNCHANGED global_counter

Checker == Check

Next == Checker
           \/ (\E self \in 1..N: AdderProc(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
=============================================================================
