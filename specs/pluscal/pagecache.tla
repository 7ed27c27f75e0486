------------------------------ MODULE pagecache ------------------------------
(***************************************************************************)
(* A lock-free page cache entry in the style of a log-structured store     *)
(* (the roadmap's "lock-free pagecache", README.md:26-42): the page table  *)
(* slot points at a chain of fragments — deltas prepended by LINK with one *)
(* compare-and-swap on the slot, on top of a base —, and REPLACE           *)
(* consolidates the chain it read into one fresh base and installs it by   *)
(* compare-and-swap from the head it started from; if somebody linked a    *)
(* delta in between, the consolidation is dropped and its node freed.      *)
(* Each thread links one delta (its own number) and then tries one         *)
(* consolidation.  Blind = TRUE installs the consolidated base with a      *)
(* plain store: deltas linked meanwhile are lost (Conservation breaks).     *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS N, Blind

(* --algorithm pagecache
variables head = 1,
          mem = [n \in 1..2 * N + 1 |-> [sum |-> 0, next |-> 0]],
          used = {1},
          linked = 0;

process T \in 1..N
  variables seen = 0, mine = 0, cur = 0, acc = 0;
begin
  L1:
    with n \in 1..2 * N + 1 do
      await n \notin used;
      used := used \cup {n};
      mine := n;
    end with;
  L2:
    seen := head;
    mem[mine] := [sum |-> self, next |-> seen];
  L3:
    if head = seen then
      head := mine;
      linked := linked + self;
    else
      goto L2;
    end if;
  R1:
    seen := head;
    cur := seen;
    acc := 0;
  R2:
    while cur # 0 do
      acc := acc + mem[cur].sum;
      cur := mem[cur].next;
    end while;
  R3:
    with n \in 1..2 * N + 1 do
      await n \notin used;
      used := used \cup {n};
      mine := n;
    end with;
  R4:
    mem[mine] := [sum |-> acc, next |-> 0];
  R5:
    if Blind \/ head = seen then
      head := mine;
    else
      used := used \ {mine};
    end if;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES head, mem_sum, mem_next, used, linked, pc, seen, mine, cur, acc

vars == << head, mem_sum, mem_next, used, linked, pc, seen, mine, cur, acc >>

(* record variables are kept field by field: r.f is r_f *)
mem == [n \in 1..2 * N + 1 |-> [sum |-> mem_sum[n], next |-> mem_next[n]]]

ProcSet == (1..N)

Init == (* Global variables *)
        /\ head = 1
        /\ mem_sum = [n \in 1..2 * N + 1 |-> 0]
        /\ mem_next = [n \in 1..2 * N + 1 |-> 0]
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
                 /\ used' = used \cup {n}
                 /\ mine' = [mine EXCEPT ![self] = n]
            /\ pc' = [pc EXCEPT ![self] = "L2"]
            /\ UNCHANGED << head, mem_sum, mem_next, linked, seen, cur, 
                            acc >>

L2(self) == /\ pc[self] = "L2"
            /\ seen' = [seen EXCEPT ![self] = head]
            /\ mem_sum' = [mem_sum EXCEPT ![mine[self]] = self]
            /\ mem_next' = [mem_next EXCEPT ![mine[self]] = seen'[self]]
            /\ pc' = [pc EXCEPT ![self] = "L3"]
            /\ UNCHANGED << head, used, linked, mine, cur, acc >>

L3(self) == /\ pc[self] = "L3"
            /\ IF head = seen[self]
                  THEN /\ head' = mine[self]
                       /\ linked' = linked + self
                       /\ pc' = [pc EXCEPT ![self] = "R1"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "L2"]
                       /\ UNCHANGED << head, linked >>
            /\ UNCHANGED << mem_sum, mem_next, used, seen, mine, cur, acc >>

R1(self) == /\ pc[self] = "R1"
            /\ seen' = [seen EXCEPT ![self] = head]
            /\ cur' = [cur EXCEPT ![self] = seen'[self]]
            /\ acc' = [acc EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "R2"]
            /\ UNCHANGED << head, mem_sum, mem_next, used, linked, mine >>

R2(self) == /\ pc[self] = "R2"
            /\ IF cur[self] # 0
                  THEN /\ acc' = [acc EXCEPT ![self] = acc[self] + mem_sum[cur[self]]]
                       /\ cur' = [cur EXCEPT ![self] = mem_next[cur[self]]]
                       /\ pc' = [pc EXCEPT ![self] = "R2"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "R3"]
                       /\ UNCHANGED << cur, acc >>
            /\ UNCHANGED << head, mem_sum, mem_next, used, linked, seen, 
                            mine >>

R3(self) == /\ pc[self] = "R3"
            /\ \E n \in 1..2 * N + 1:
                 /\ n \notin used
                 /\ used' = used \cup {n}
                 /\ mine' = [mine EXCEPT ![self] = n]
            /\ pc' = [pc EXCEPT ![self] = "R4"]
            /\ UNCHANGED << head, mem_sum, mem_next, linked, seen, cur, 
                            acc >>

R4(self) == /\ pc[self] = "R4"
            /\ mem_sum' = [mem_sum EXCEPT ![mine[self]] = acc[self]]
            /\ mem_next' = [mem_next EXCEPT ![mine[self]] = 0]
            /\ pc' = [pc EXCEPT ![self] = "R5"]
            /\ UNCHANGED << head, used, linked, seen, mine, cur, acc >>

R5(self) == /\ pc[self] = "R5"
            /\ IF Blind \/ head = seen[self]
                  THEN /\ head' = mine[self]
                       /\ UNCHANGED used
                  ELSE /\ used' = used \ {mine[self]}
                       /\ UNCHANGED head
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << mem_sum, mem_next, linked, seen, mine, cur, 
                            acc >>

T(self) == L1(self) \/ L2(self) \/ L3(self) \/ R1(self) \/ R2(self) \/ R3(self) \/ R4(self) \/ R5(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Nx(n) == IF n = 0 THEN 0 ELSE mem[n].next
Sm(n) == IF n = 0 THEN 0 ELSE mem[n].sum
ChainSum == Sm(head) + Sm(Nx(head)) + Sm(Nx(Nx(head))) + Sm(Nx(Nx(Nx(head))))     \* (a base and at most N <= 3 deltas)
Conservation == ChainSum = linked
HeadIsAllocated == head \in used
=============================================================================
