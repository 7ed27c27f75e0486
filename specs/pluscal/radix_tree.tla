----------------------------- MODULE radix_tree -----------------------------
(***************************************************************************)
(* A lock-free radix tree of two levels of fan-out two (keys 0..3; the     *)
(* roadmap's "lock-free radix tree", README.md:26-42).  An inserter walks  *)
(* from the root; where the child is missing it allocates a node and       *)
(* installs it with a compare-and-swap — the loser of a race frees its     *)
(* node and continues in the winner's; the value goes into the leaf slot   *)
(* by compare-and-swap, first writer wins.  Plain = TRUE installs the      *)
(* child with a plain store: the subtree of the thread that was first is   *)
(* unlinked, its key is no longer found and its node leaks.                *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS N, Plain

(* --algorithm radix_tree
variables child = [h \in 0..1 |-> 0],
          slot = [i \in 0..2 * N + 1 |-> 0],
          used = {};

process T \in 1..N
  variables key = 0, node = 0, mine = 0;
begin
  Pick:
    with k \in 0..3 do
      key := k;
    end with;
  Walk:
    node := child[key \div 2];
    if node # 0 then
      goto Put;
    end if;
  Alloc:
    with n \in 1..N do
      await n \notin used;
      used := used \cup {n};
      mine := n;
    end with;
  Install:
    if Plain \/ child[key \div 2] = 0 then
      child[key \div 2] := mine;
      node := mine;
    else
      used := used \ {mine};
      node := child[key \div 2];
      mine := 0;
    end if;
  Put:
    if slot[2 * node + key % 2] = 0 then
      slot[2 * node + key % 2] := self;
    end if;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES child, slot, used, pc, key, node, mine

vars == << child, slot, used, pc, key, node, mine >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ child = [h \in 0..1 |-> 0]
        /\ slot = [i \in 0..2 * N + 1 |-> 0]
        /\ used = {}
        (* Process T *)
        /\ key = [self \in 1..N |-> 0]
        /\ node = [self \in 1..N |-> 0]
        /\ mine = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "Pick"]

Pick(self) == /\ pc[self] = "Pick"
              /\ \E k \in 0..3:
                   /\ key' = [key EXCEPT ![self] = k]
              /\ pc' = [pc EXCEPT ![self] = "Walk"]
              /\ UNCHANGED << child, slot, used, node, mine >>

Walk(self) == /\ pc[self] = "Walk"
              /\ node' = [node EXCEPT ![self] = child[key[self] \div 2]]
              /\ IF node'[self] # 0
                    THEN /\ pc' = [pc EXCEPT ![self] = "Put"]
                    ELSE /\ pc' = [pc EXCEPT ![self] = "Alloc"]
              /\ UNCHANGED << child, slot, used, key, mine >>

Alloc(self) == /\ pc[self] = "Alloc"
               /\ \E n \in 1..N:
                    /\ n \notin used
                    /\ used' = used \cup {n}
                    /\ mine' = [mine EXCEPT ![self] = n]
               /\ pc' = [pc EXCEPT ![self] = "Install"]
               /\ UNCHANGED << child, slot, key, node >>

Install(self) == /\ pc[self] = "Install"
                 /\ IF Plain \/ child[key[self] \div 2] = 0
                       THEN /\ child' = [child EXCEPT ![key[self] \div 2] = mine[self]]
                            /\ node' = [node EXCEPT ![self] = mine[self]]
                            /\ UNCHANGED << used, mine >>
                       ELSE /\ used' = used \ {mine[self]}
                            /\ node' = [node EXCEPT ![self] = child[key[self] \div 2]]
                            /\ mine' = [mine EXCEPT ![self] = 0]
                            /\ UNCHANGED child
                 /\ pc' = [pc EXCEPT ![self] = "Put"]
                 /\ UNCHANGED << slot, key >>

Put(self) == /\ pc[self] = "Put"
             /\ IF slot[2 * node[self] + key[self] % 2] = 0
                   THEN /\ slot' = [slot EXCEPT ![2 * node[self] + key[self] % 2] = self]
                   ELSE /\ TRUE
                        /\ UNCHANGED slot
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << child, used, key, node, mine >>

T(self) == Pick(self) \/ Walk(self) \/ Alloc(self) \/ Install(self) \/ Put(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Linked(n) == n = child[0] \/ n = child[1]
Found(t) == child[key[t] \div 2] # 0 /\ slot[2 * child[key[t] \div 2] + key[t] % 2] # 0
AllDone == \A t \in 1..N : pc[t] = "Done"
InsertedKeysAreFound == AllDone => \A t \in 1..N : Found(t)
NoLeak == AllDone => \A n \in used : Linked(n)
ChildrenAreNodes == \A h \in 0..1 : child[h] = 0 \/ child[h] \in used
=============================================================================
