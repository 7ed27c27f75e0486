------------------------------- MODULE euclid -------------------------------
(***************************************************************************)
(* A uniprocess algorithm (no `process`): Euclid's subtractive gcd over all *)
(* initial pairs.  Exercises uniprocess translation, \in initial values,   *)
(* while with if/else, and an invariant.                                   *)
(***************************************************************************)
EXTENDS Naturals
CONSTANT M

(* --algorithm euclid
variables x \in 1..M, y \in 1..M, x0 = x, y0 = y;
begin
  Loop:
    while x # y do
      Step:
        if x < y then
          y := y - x;
        else
          x := x - y;
        end if;
    end while;
  Fin:
    assert x0 % x = 0 /\ y0 % x = 0;
end algorithm *)
\* BEGIN TRANSLATION
VARIABLES x, y, x0, y0, pc

vars == << x, y, x0, y0, pc >>

Init == (* Global variables *)
        /\ x \in 1..M
        /\ y \in 1..M
        /\ x0 = x
        /\ y0 = y
        /\ pc = "Loop"

Loop == /\ pc = "Loop"
        /\ IF x # y
              THEN /\ pc' = "Step"
              ELSE /\ pc' = "Fin"
        /\ UNCHANGED << x, y, x0, y0 >>

Step == /\ pc = "Step"
        /\ IF x < y
              THEN /\ y' = y - x
                   /\ UNCHANGED x
              ELSE /\ x' = x - y
                   /\ UNCHANGED y
        /\ pc' = "Loop"
        /\ UNCHANGED << x0, y0 >>

Fin == /\ pc = "Fin"
       /\ Assert(x0 % x = 0 /\ y0 % x = 0, 
                 "Failure of assertion at line 23, column 5.")
       /\ pc' = "Done"
       /\ UNCHANGED << x, y, x0, y0 >>

Next == Loop \/ Step \/ Fin
           \/ (* Disjunct to prevent deadlock on termination *)
              (pc = "Done" /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(pc = "Done")

\* END TRANSLATION

Positive == x >= 1 /\ y >= 1
=============================================================================
