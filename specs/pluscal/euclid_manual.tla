---------------------------- MODULE euclid_manual ----------------------------
(***************************************************************************)
(* Euclid's algorithm exactly as the PlusCal manual of the reference       *)
(* develops it (examples/p-manual.pdf sections 2.1-2.3, pp.6-10): a        *)
(* uniprocess algorithm WITHOUT labels — "the translator will              *)
(* automatically add the necessary labels" (p.9).  With N = 4 the manual's *)
(* TLC run prints <<24, 4, "have gcd", 4>>, <<24, 3, "have gcd", 3>>,      *)
(* <<24, 2, "have gcd", 2>>, <<24, 1, "have gcd", 1>> (p.10).              *)
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
  print <<24, v_ini, "have gcd", v>>;
end algorithm *)
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
               ELSE /\ PrintT(<<24, v_ini, "have gcd", v>>)
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

\* what the manual's run prints, as a state predicate: at termination v is the gcd of 24 and the initial v
Divides(p, q) == q % p = 0
ResultIsGcd == (pc = "Done") => /\ Divides(v, 24) /\ Divides(v, v_ini)
                                /\ \A d \in 1..24 : (Divides(d, 24) /\ Divides(d, v_ini)) => d <= v
=============================================================================
