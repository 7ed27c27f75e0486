---- MODULE csyntax_mix ----
EXTENDS Naturals
CONSTANT N
(* --algorithm csyntax_mix {
  variables x = 0, lock = 0, a = [i \in 1..N |-> 0];
  define { Sum == a[1] + a[2]  Twice(v) == v + v };
  macro acquire(l) { await l = 0; l := 1; }
  process (W \in 1..N) variables t = 0, k = 0; {
    A: acquire(lock);
    B: t := x;
       if (t > 5) { goto D; } else if (t = 3) k := Twice(k); else { skip; };
    C: x := t + 1; a[self] := a[self] + 1;
       either { k := 1; } or { with (d \in {2, 3}) { k := d; } };
    D: lock := 0;
  }
} *)
\* BEGIN TRANSLATION
VARIABLES x, lock, a, pc

(* define statement *)
Sum == a[1] + a[2]

Twice(v) == v + v

VARIABLES t, k

vars == << x, lock, a, pc, t, k >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ x = 0
        /\ lock = 0
        /\ a = [i \in 1..N |-> 0]
        (* Process W *)
        /\ t = [self \in 1..N |-> 0]
        /\ k = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "A"]

A(self) == /\ pc[self] = "A"
           /\ lock = 0
           /\ lock' = 1
           /\ pc' = [pc EXCEPT ![self] = "B"]
           /\ UNCHANGED << x, a, t, k >>

B(self) == /\ pc[self] = "B"
           /\ t' = [t EXCEPT ![self] = x]
           /\ IF t'[self] > 5
                 THEN /\ pc' = [pc EXCEPT ![self] = "D"]
                      /\ UNCHANGED k
                 ELSE /\ IF t'[self] = 3
                            THEN /\ k' = [k EXCEPT ![self] = Twice(k[self])]
                            ELSE /\ TRUE
                                 /\ UNCHANGED k
                      /\ pc' = [pc EXCEPT ![self] = "C"]
           /\ UNCHANGED << x, lock, a >>

C(self) == /\ pc[self] = "C"
           /\ x' = t[self] + 1
           /\ a' = [a EXCEPT ![self] = a[self] + 1]
           /\ \/ /\ k' = [k EXCEPT ![self] = 1]
              \/ /\ \E d \in {2, 3}:
                      /\ k' = [k EXCEPT ![self] = d]
           /\ pc' = [pc EXCEPT ![self] = "D"]
           /\ UNCHANGED << lock, t >>

D(self) == /\ pc[self] = "D"
           /\ lock' = 0
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << x, a, t, k >>

W(self) == A(self) \/ B(self) \/ C(self) \/ D(self)

Next == (\E self \in 1..N: W(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
Inv == x <= N /\ Sum = x
====
