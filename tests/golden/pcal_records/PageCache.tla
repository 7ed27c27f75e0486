------------------------------ MODULE PageCache ------------------------------
(***************************************************************************)
(* HAND-WRITTEN translation of specs/pluscal/pagecache.tla in the style of  *)
(* pcal2tla (p-manual App. B): `mem` stays ONE function to records, as the  *)
(* Java translator keeps it (`mem' = [mem EXCEPT ![mine[self]] = [sum |->   *)
(* ..]]`, `mem[cur[self]].sum`), where the product's front-end keeps the    *)
(* record variables field by field (mem_sum, mem_next).  Written from the   *)
(* algorithm text, not from the product's translation: tests/test_pcal.py   *)
(* evaluates it with oracle/tlaplus.py and compares the state graph with    *)
(* the product's translation and with the compiled program (VERDICT round   *)
(* 5, next 4: a golden that does not come from the product).                *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS N, Blind

VARIABLES head, mem, used, linked, pc, seen, mine, cur, acc

vars == << head, mem, used, linked, pc, seen, mine, cur, acc >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ head = 1
        /\ mem = [n \in 1..2 * N + 1 |-> [sum |-> 0, next |-> 0]]
        /\ used = {1}
        /\ linked = 0
        (* Process T *)
        /\ seen = [self \in 1..N |-> 0]
        /\ mine = [self \in 1..N |-> 0]
        /\ cur = [self \in 1..N |-> 0]
        /\ acc = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "L1"]

L1(self) == /\ pc[self] = "L1"
            /\ \E n \in 1..2 * N + 1:
                 /\ n \notin used
                 /\ used' = (used \cup {n})
                 /\ mine' = [mine EXCEPT ![self] = n]
            /\ pc' = [pc EXCEPT ![self] = "L2"]
            /\ UNCHANGED << head, mem, linked, seen, cur, acc >>

L2(self) == /\ pc[self] = "L2"
            /\ seen' = [seen EXCEPT ![self] = head]
            /\ mem' = [mem EXCEPT ![mine[self]] = [sum |-> self, next |-> seen'[self]]]
            /\ pc' = [pc EXCEPT ![self] = "L3"]
            /\ UNCHANGED << head, used, linked, mine, cur, acc >>

L3(self) == /\ pc[self] = "L3"
            /\ IF head = seen[self]
                  THEN /\ head' = mine[self]
                       /\ linked' = linked + self
                       /\ pc' = [pc EXCEPT ![self] = "R1"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "L2"]
                       /\ UNCHANGED << head, linked >>
            /\ UNCHANGED << mem, used, seen, mine, cur, acc >>

R1(self) == /\ pc[self] = "R1"
            /\ seen' = [seen EXCEPT ![self] = head]
            /\ cur' = [cur EXCEPT ![self] = seen'[self]]
            /\ acc' = [acc EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "R2"]
            /\ UNCHANGED << head, mem, used, linked, mine >>

R2(self) == /\ pc[self] = "R2"
            /\ IF cur[self] # 0
                  THEN /\ acc' = [acc EXCEPT ![self] = acc[self] + mem[cur[self]].sum]
                       /\ cur' = [cur EXCEPT ![self] = mem[cur[self]].next]
                       /\ pc' = [pc EXCEPT ![self] = "R2"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "R3"]
                       /\ UNCHANGED << cur, acc >>
            /\ UNCHANGED << head, mem, used, linked, seen, mine >>

R3(self) == /\ pc[self] = "R3"
            /\ \E n \in 1..2 * N + 1:
                 /\ n \notin used
                 /\ used' = (used \cup {n})
                 /\ mine' = [mine EXCEPT ![self] = n]
            /\ pc' = [pc EXCEPT ![self] = "R4"]
            /\ UNCHANGED << head, mem, linked, seen, cur, acc >>

R4(self) == /\ pc[self] = "R4"
            /\ mem' = [mem EXCEPT ![mine[self]] = [sum |-> acc[self], next |-> 0]]
            /\ pc' = [pc EXCEPT ![self] = "R5"]
            /\ UNCHANGED << head, used, linked, seen, mine, cur, acc >>

R5(self) == /\ pc[self] = "R5"
            /\ IF Blind \/ head = seen[self]
                  THEN /\ head' = mine[self]
                       /\ used' = used
                  ELSE /\ used' = used \ {mine[self]}
                       /\ head' = head
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << mem, linked, seen, mine, cur, acc >>

T(self) == L1(self) \/ L2(self) \/ L3(self) \/ R1(self) \/ R2(self) \/ R3(self) \/ R4(self) \/ R5(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Nx(n) == IF n = 0 THEN 0 ELSE mem[n].next
Sm(n) == IF n = 0 THEN 0 ELSE mem[n].sum
ChainSum == Sm(head) + Sm(Nx(head)) + Sm(Nx(Nx(head))) + Sm(Nx(Nx(Nx(head))))
Conservation == ChainSum = linked
HeadIsAllocated == head \in used
=============================================================================
