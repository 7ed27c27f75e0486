---------------------------- MODULE euclid_assert ----------------------------
(***************************************************************************)
(* examples/p-manual.pdf section 2.4 (pp.10-11): EuclidAlg with the print  *)
(* statement replaced by `assert v = gcd(24, v_ini)`, and the manual's     *)
(* definition of gcd (a bounded CHOOSE) placed before the translation.     *)
(***************************************************************************)
EXTENDS Naturals, TLC
CONSTANT N

(* --algorithm EuclidAlg
variables u = 24, v \in 1..N, v_ini = v
begin
  while u # 0 do
    if u < v then u := v || v := u; \* swap u and v.
    end if;
    u := u - v;
  end while;
  assert v = gcd(24, v_ini)
end algorithm *)

gcd(x, y) == CHOOSE i \in 1..x :
                /\ x % i = 0
                /\ y % i = 0
                /\ \A j \in 1..x : /\ x % j = 0
                                   /\ y % j = 0
                                   => i >= j

\* BEGIN TRANSLATION
VARIABLES u, v, v_ini, pc

vars == << u, v, v_ini, pc >>

Init == (* Global variables *)
        /\ u = 24
        /\ v \in 1..N
        /\ v_ini = v
        /\ pc = "Lbl_1"

Lbl_1 == /\ pc = "Lbl_1"
         /\ IF u # 0
               THEN /\ IF u < v
                          THEN /\ u' = v
                               /\ v' = u
                          ELSE /\ TRUE
                               /\ UNCHANGED << u, v >>
                    /\ pc' = "Lbl_2"
               ELSE /\ Assert(v = gcd(24, v_ini), 
                              "Failure of assertion at line 18, column 3.")
                    /\ pc' = "Done"
                    /\ UNCHANGED << u, v >>
         /\ UNCHANGED v_ini

Lbl_2 == /\ pc = "Lbl_2"
         /\ u' = u - v
         /\ pc' = "Lbl_1"
         /\ UNCHANGED << v, v_ini >>

Next == Lbl_1 \/ Lbl_2
           \/ (* Disjunct to prevent deadlock on termination *)
              (pc = "Done" /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(pc = "Done")

\* END TRANSLATION
=============================================================================
