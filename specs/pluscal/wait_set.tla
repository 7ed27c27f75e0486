------------------------------ MODULE wait_set ------------------------------
(***************************************************************************)
(* A lock handed out from a SET of waiting processes (no queue, so no      *)
(* fairness among waiters).  Exercises set-valued variables: {} literals,  *)
(* \cup, \ , \in, \subseteq, Cardinality, `with` and \A over a set         *)
(* variable.                                                               *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANT N

(* --algorithm wait_set
variables waiting = {}, holder = 0, served = {}, grants = 0;

process Client \in 1..N
begin
  Req:  waiting := waiting \cup {self};
  Acq:  await holder = self;
  Rel:  served := served \cup {self};
        holder := 0;
end process

process Arbiter = 0
begin
  Loop:
    while grants < N do
      Pick:
        await holder = 0 /\ waiting # {};
        with p \in waiting do
          holder := p;
          waiting := waiting \ {p};
        end with;
        grants := grants + 1;
    end while;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES waiting, holder, served, grants, pc

vars == << waiting, holder, served, grants, pc >>

ProcSet == (1..N) \cup {0}

Init == (* Global variables *)
        /\ waiting = {}
        /\ holder = 0
        /\ served = {}
        /\ grants = 0
        /\ pc = [self \in ProcSet |-> CASE self \in 1..N -> "Req"
                                        [] self = 0 -> "Loop"]

Req(self) == /\ pc[self] = "Req"
             /\ waiting' = waiting \cup {self}
             /\ pc' = [pc EXCEPT ![self] = "Acq"]
             /\ UNCHANGED << holder, served, grants >>

Acq(self) == /\ pc[self] = "Acq"
             /\ holder = self
             /\ pc' = [pc EXCEPT ![self] = "Rel"]
             /\ UNCHANGED << waiting, holder, served, grants >>

Rel(self) == /\ pc[self] = "Rel"
             /\ served' = served \cup {self}
             /\ holder' = 0
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << waiting, grants >>

Client(self) == Req(self) \/ Acq(self) \/ Rel(self)

Loop == /\ pc[0] = "Loop"
        /\ IF grants < N
              THEN /\ pc' = [pc EXCEPT ![0] = "Pick"]
              ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
        /\ UNCHANGED << waiting, holder, served, grants >>

Pick == /\ pc[0] = "Pick"
        /\ holder = 0 /\ waiting # {}
        /\ \E p \in waiting:
             /\ holder' = p
             /\ waiting' = waiting \ {p}
        /\ grants' = grants + 1
        /\ pc' = [pc EXCEPT ![0] = "Loop"]
        /\ UNCHANGED served

Arbiter == Loop \/ Pick

Next == Arbiter
           \/ (\E self \in 1..N: Client(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Disjoint == (waiting \cap served) = {} /\ served \subseteq 1..N
HolderNotWaiting == holder = 0 \/ holder \notin waiting
Counted == Cardinality(served) <= grants /\ \A p \in served : pc[p] = "Done"
=============================================================================
