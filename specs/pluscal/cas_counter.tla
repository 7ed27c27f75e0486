----------------------------- MODULE cas_counter -----------------------------
(***************************************************************************)
(* A lock-free counter: every worker increments a shared cell N times with *)
(* a read / compare-and-swap retry loop (README.md:26-42: "lock-free       *)
(* algorithms").  Exercises CONSTANTS, process-local variables, while,     *)
(* if/else with a goto, a checker process with await, and an invariant     *)
(* over a process-local function.                                          *)
(***************************************************************************)
EXTENDS Naturals
CONSTANTS Workers, N

(* --algorithm cas_counter
variables cell = 0;

process Worker \in 1..Workers
  variables seen = 0, done = 0;
begin
  Loop:
    while done < N do
      Read: seen := cell;
      Cas:
        if cell = seen then
          cell := seen + 1;
          done := done + 1;
        else
          goto Read;
        end if;
    end while;
end process

process Checker = 0
begin
  Check:
    await \A w \in 1..Workers : done[w] = N;
    assert cell = Workers * N;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES cell, pc, seen, done

vars == << cell, pc, seen, done >>

ProcSet == (1..Workers) \cup {0}

Init == (* Global variables *)
        /\ cell = 0
        (* Process Worker *)
        /\ seen = [self \in 1..Workers |-> 0]
        /\ done = [self \in 1..Workers |-> 0]
        /\ pc = [self \in ProcSet |-> CASE self \in 1..Workers -> "Loop"
                                        [] self = 0 -> "Check"]

Loop(self) == /\ pc[self] = "Loop"
              /\ IF done[self] < N
                    THEN /\ pc' = [pc EXCEPT ![self] = "Read"]
                    ELSE /\ pc' = [pc EXCEPT ![self] = "Done"]
              /\ UNCHANGED << cell, seen, done >>

Read(self) == /\ pc[self] = "Read"
              /\ seen' = [seen EXCEPT ![self] = cell]
              /\ pc' = [pc EXCEPT ![self] = "Cas"]
              /\ UNCHANGED << cell, done >>

Cas(self) == /\ pc[self] = "Cas"
             /\ IF cell = seen[self]
                   THEN /\ cell' = seen[self] + 1
                        /\ done' = [done EXCEPT ![self] = done[self] + 1]
                        /\ pc' = [pc EXCEPT ![self] = "Loop"]
                   ELSE /\ pc' = [pc EXCEPT ![self] = "Read"]
                        /\ UNCHANGED << cell, done >>
             /\ UNCHANGED seen

Worker(self) == Loop(self) \/ Read(self) \/ Cas(self)

Check == /\ pc[0] = "Check"
         /\ \A w \in 1..Workers : done[w] = N
         /\ Assert(cell = Workers * N, 
                   "Failure of assertion at line 35, column 5.")
         /\ pc' = [pc EXCEPT ![0] = "Done"]
         /\ UNCHANGED << cell, seen, done >>

Checker == Check

Next == Checker
           \/ (\E self \in 1..Workers: Worker(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

NeverTooMany == cell <= Workers * N
SeenIsOld == \A w \in 1..Workers : seen[w] <= cell
=============================================================================
