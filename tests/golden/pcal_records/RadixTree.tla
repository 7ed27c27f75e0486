------------------------------ MODULE RadixTree ------------------------------
(***************************************************************************)
(* HAND-WRITTEN translation of specs/pluscal/radix_tree.tla in the style of *)
(* pcal2tla (p-manual App. B), written from the ALGORITHM text and          *)
(* evaluated by oracle/tlaplus.py; see EpochGc.tla.                         *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS N, Plain

VARIABLES child, slot, used, pc, key, node, mine

vars == << child, slot, used, pc, key, node, mine >>

ProcSet == (1..N)

Init == /\ child = [h \in 0..1 |-> 0]
        /\ slot = [i \in 0..2 * N + 1 |-> 0]
        /\ used = {}
        /\ key = [self \in 1..N |-> 0]
        /\ node = [self \in 1..N |-> 0]
        /\ mine = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "Pick"]

Half(self) == key[self] \div 2
Leaf(self) == 2 * node[self] + key[self] % 2

Pick(self) == /\ pc[self] = "Pick"
              /\ \E kk \in 0..3 : key' = [key EXCEPT ![self] = kk]
              /\ pc' = [pc EXCEPT ![self] = "Walk"]
              /\ UNCHANGED << child, slot, used, node, mine >>

Walk(self) == /\ pc[self] = "Walk"
              /\ node' = [node EXCEPT ![self] = child[Half(self)]]
              /\ IF child[Half(self)] # 0
                    THEN pc' = [pc EXCEPT ![self] = "Put"]
                    ELSE pc' = [pc EXCEPT ![self] = "Alloc"]
              /\ UNCHANGED << child, slot, used, key, mine >>

Alloc(self) == /\ pc[self] = "Alloc"
               /\ \E n \in 1..N :
                    /\ n \notin used
                    /\ used' = used \cup {n}
                    /\ mine' = [mine EXCEPT ![self] = n]
               /\ pc' = [pc EXCEPT ![self] = "Install"]
               /\ UNCHANGED << child, slot, key, node >>

Install(self) == /\ pc[self] = "Install"
                 /\ IF Plain \/ child[Half(self)] = 0
                       THEN /\ child' = [child EXCEPT ![Half(self)] = mine[self]]
                            /\ node' = [node EXCEPT ![self] = mine[self]]
                            /\ UNCHANGED << used, mine >>
                       ELSE /\ used' = used \ {mine[self]}
                            /\ node' = [node EXCEPT ![self] = child[Half(self)]]
                            /\ mine' = [mine EXCEPT ![self] = 0]
                            /\ child' = child
                 /\ pc' = [pc EXCEPT ![self] = "Put"]
                 /\ UNCHANGED << slot, key >>

Put(self) == /\ pc[self] = "Put"
             /\ IF slot[Leaf(self)] = 0
                   THEN slot' = [slot EXCEPT ![Leaf(self)] = self]
                   ELSE slot' = slot
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << child, used, key, node, mine >>

T(self) == Pick(self) \/ Walk(self) \/ Alloc(self) \/ Install(self) \/ Put(self)

Next == (\E self \in 1..N: T(self))
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Linked(n) == n = child[0] \/ n = child[1]
Found(t) == child[key[t] \div 2] # 0 /\ slot[2 * child[key[t] \div 2] + key[t] % 2] # 0
AllDone == \A t \in 1..N : pc[t] = "Done"
InsertedKeysAreFound == AllDone => \A t \in 1..N : Found(t)
NoLeak == AllDone => \A n \in used : Linked(n)
ChildrenAreNodes == \A h \in 0..1 : child[h] = 0 \/ child[h] \in used
=============================================================================
