------------------------------- MODULE swap -------------------------------
(* simultaneous assignment `a := e || b := f`: every right-hand side reads the values before the statement *)
EXTENDS Naturals
(* --algorithm swap
variables x = 1, y = 2, a = [i \in 1..2 |-> i];
begin
  S: x := y || y := x;
  T: a[1] := a[2] || x := x + y;
  U: assert x = 3 /\ y = 1 /\ a[1] = 2;
end algorithm *)
\* BEGIN TRANSLATION
VARIABLES x, y, a, pc

vars == << x, y, a, pc >>

Init == (* Global variables *)
        /\ x = 1
        /\ y = 2
        /\ a = [i \in 1..2 |-> i]
        /\ pc = "S"

S == /\ pc = "S"
     /\ x' = y
     /\ y' = x
     /\ pc' = "T"
     /\ UNCHANGED a

T == /\ pc = "T"
     /\ a' = [a EXCEPT ![1] = a[2]]
     /\ x' = x + y
     /\ pc' = "U"
     /\ UNCHANGED y

U == /\ pc = "U"
     /\ Assert(x = 3 /\ y = 1 /\ a[1] = 2, 
               "Failure of assertion at line 9, column 6.")
     /\ pc' = "Done"
     /\ UNCHANGED << x, y, a >>

Next == S \/ T \/ U
           \/ (* Disjunct to prevent deadlock on termination *)
              (pc = "Done" /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(pc = "Done")

\* END TRANSLATION
====
