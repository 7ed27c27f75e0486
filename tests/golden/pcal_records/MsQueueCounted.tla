--------------------------- MODULE MsQueueCounted ---------------------------
(* specs/pluscal/ms_queue_counted.tla the way pcal2tla translates it (p-manual section 3.8 / App. B): Q stays ONE record of two
   (ptr, count) records, mem a function to records whose `next` field is a record, head / tail / next functions from process ids to
   records; a nested field is assigned with EXCEPT !.Head, ![i].next.ptr.  Written by hand: the product keeps NESTED records field by
   field, one level per pass (tla_rust_amd/csrc/pcal.cpp: RecordFlattener), and tests/test_pcal.py checks that the two are the same
   state graph. *)
EXTENDS Naturals, FiniteSets, TLC
CONSTANTS N, K, Counted
VARIABLES Q, mem, free, taken, pc, head, tail, next, node, got, phase

vars == << Q, mem, free, taken, pc, head, tail, next, node, got, phase >>

ProcSet == (1..N)

Init == /\ Q = [Head |-> [ptr |-> 1, count |-> 0], Tail |-> [ptr |-> 2, count |-> 0]]
        /\ mem = [n \in 1..K |-> [value |-> IF n = 2 THEN N + 1 ELSE 0,
                                  next |-> [ptr |-> IF n = 1 THEN 2 ELSE 0, count |-> 0]]]
        /\ free = 3..K
        /\ taken = {}
        /\ head = [self \in 1..N |-> [ptr |-> 0, count |-> 0]]
        /\ tail = [self \in 1..N |-> [ptr |-> 0, count |-> 0]]
        /\ next = [self \in 1..N |-> [ptr |-> 0, count |-> 0]]
        /\ node = [self \in 1..N |-> 0]
        /\ got = [self \in 1..N |-> 0]
        /\ phase = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "Start"]

Start(self) == /\ pc[self] = "Start"
               /\ IF phase[self] = 1
                     THEN pc' = [pc EXCEPT ![self] = "E1"]
                     ELSE IF phase[self] = 3
                             THEN pc' = [pc EXCEPT ![self] = "Fin"]
                             ELSE pc' = [pc EXCEPT ![self] = "D2"]
               /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, got, phase >>

D2(self) == /\ pc[self] = "D2"
            /\ head' = [head EXCEPT ![self] = Q.Head]
            /\ pc' = [pc EXCEPT ![self] = "D3"]
            /\ UNCHANGED << Q, mem, free, taken, tail, next, node, got, phase >>

D3(self) == /\ pc[self] = "D3"
            /\ tail' = [tail EXCEPT ![self] = Q.Tail]
            /\ pc' = [pc EXCEPT ![self] = "D4"]
            /\ UNCHANGED << Q, mem, free, taken, head, next, node, got, phase >>

D4(self) == /\ pc[self] = "D4"
            /\ next' = [next EXCEPT ![self] = mem[head[self].ptr].next]
            /\ pc' = [pc EXCEPT ![self] = "D5"]
            /\ UNCHANGED << Q, mem, free, taken, head, tail, node, got, phase >>

D5(self) == /\ pc[self] = "D5"
            /\ IF head[self] # Q.Head
                  THEN pc' = [pc EXCEPT ![self] = "D2"]
                  ELSE pc' = [pc EXCEPT ![self] = "D6"]
            /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, got, phase >>

D6(self) == /\ pc[self] = "D6"
            /\ IF head[self].ptr = tail[self].ptr
                  THEN IF next[self].ptr = 0
                          THEN /\ got' = [got EXCEPT ![self] = 0]
                               /\ pc' = [pc EXCEPT ![self] = "Advance"]
                          ELSE /\ pc' = [pc EXCEPT ![self] = "D10"]
                               /\ got' = got
                  ELSE /\ pc' = [pc EXCEPT ![self] = "D12"]
                       /\ got' = got
            /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, phase >>

D12(self) == /\ pc[self] = "D12"
             /\ got' = [got EXCEPT ![self] = mem[next[self].ptr].value]
             /\ pc' = [pc EXCEPT ![self] = "D13"]
             /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, phase >>

D13(self) == /\ pc[self] = "D13"
             /\ IF (Counted /\ Q.Head = head[self]) \/ (~Counted /\ Q.Head.ptr = head[self].ptr)
                   THEN /\ Q' = [Q EXCEPT !.Head = [ptr |-> next[self].ptr, count |-> head[self].count + 1]]
                        /\ pc' = [pc EXCEPT ![self] = "D19"]
                   ELSE /\ pc' = [pc EXCEPT ![self] = "D2"]
                        /\ Q' = Q
             /\ UNCHANGED << mem, free, taken, head, tail, next, node, got, phase >>

D19(self) == /\ pc[self] = "D19"
             /\ Assert(got[self] \notin taken, "Failure of assertion at line 63, column 5.")
             /\ taken' = (taken \cup {got[self]})
             /\ free' = (free \cup {head[self].ptr})
             /\ pc' = [pc EXCEPT ![self] = "Advance"]
             /\ UNCHANGED << Q, mem, head, tail, next, node, got, phase >>

D10(self) == /\ pc[self] = "D10"
             /\ IF (Counted /\ Q.Tail = tail[self]) \/ (~Counted /\ Q.Tail.ptr = tail[self].ptr)
                   THEN Q' = [Q EXCEPT !.Tail = [ptr |-> next[self].ptr, count |-> tail[self].count + 1]]
                   ELSE Q' = Q
             /\ pc' = [pc EXCEPT ![self] = "D2"]
             /\ UNCHANGED << mem, free, taken, head, tail, next, node, got, phase >>

E1(self) == /\ pc[self] = "E1"
            /\ \E n \in free:
                 /\ node' = [node EXCEPT ![self] = n]
                 /\ free' = free \ {n}
            /\ pc' = [pc EXCEPT ![self] = "E2"]
            /\ UNCHANGED << Q, mem, taken, head, tail, next, got, phase >>

E2(self) == /\ pc[self] = "E2"
            /\ mem' = [mem EXCEPT ![node[self]].value = self, ![node[self]].next.ptr = 0]
            /\ pc' = [pc EXCEPT ![self] = "E5"]
            /\ UNCHANGED << Q, free, taken, head, tail, next, node, got, phase >>

E5(self) == /\ pc[self] = "E5"
            /\ tail' = [tail EXCEPT ![self] = Q.Tail]
            /\ pc' = [pc EXCEPT ![self] = "E6"]
            /\ UNCHANGED << Q, mem, free, taken, head, next, node, got, phase >>

E6(self) == /\ pc[self] = "E6"
            /\ next' = [next EXCEPT ![self] = mem[tail[self].ptr].next]
            /\ pc' = [pc EXCEPT ![self] = "E7"]
            /\ UNCHANGED << Q, mem, free, taken, head, tail, node, got, phase >>

E7(self) == /\ pc[self] = "E7"
            /\ IF tail[self] # Q.Tail
                  THEN pc' = [pc EXCEPT ![self] = "E5"]
                  ELSE pc' = [pc EXCEPT ![self] = "E8"]
            /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, got, phase >>

E8(self) == /\ pc[self] = "E8"
            /\ IF next[self].ptr # 0
                  THEN pc' = [pc EXCEPT ![self] = "E13"]
                  ELSE pc' = [pc EXCEPT ![self] = "E9"]
            /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, got, phase >>

E9(self) == /\ pc[self] = "E9"
            /\ IF (Counted /\ mem[tail[self].ptr].next = next[self]) \/ (~Counted /\ mem[tail[self].ptr].next.ptr = next[self].ptr)
                  THEN /\ mem' = [mem EXCEPT ![tail[self].ptr].next = [ptr |-> node[self], count |-> next[self].count + 1]]
                       /\ pc' = [pc EXCEPT ![self] = "E17"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "E5"]
                       /\ mem' = mem
            /\ UNCHANGED << Q, free, taken, head, tail, next, node, got, phase >>

E17(self) == /\ pc[self] = "E17"
             /\ IF (Counted /\ Q.Tail = tail[self]) \/ (~Counted /\ Q.Tail.ptr = tail[self].ptr)
                   THEN Q' = [Q EXCEPT !.Tail = [ptr |-> node[self], count |-> tail[self].count + 1]]
                   ELSE Q' = Q
             /\ pc' = [pc EXCEPT ![self] = "Advance"]
             /\ UNCHANGED << mem, free, taken, head, tail, next, node, got, phase >>

E13(self) == /\ pc[self] = "E13"
             /\ IF (Counted /\ Q.Tail = tail[self]) \/ (~Counted /\ Q.Tail.ptr = tail[self].ptr)
                   THEN Q' = [Q EXCEPT !.Tail = [ptr |-> next[self].ptr, count |-> tail[self].count + 1]]
                   ELSE Q' = Q
             /\ pc' = [pc EXCEPT ![self] = "E5"]
             /\ UNCHANGED << mem, free, taken, head, tail, next, node, got, phase >>

Advance(self) == /\ pc[self] = "Advance"
                 /\ phase' = [phase EXCEPT ![self] = phase[self] + 1]
                 /\ pc' = [pc EXCEPT ![self] = "Start"]
                 /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, got >>

Fin(self) == /\ pc[self] = "Fin"
             /\ TRUE
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << Q, mem, free, taken, head, tail, next, node, got, phase >>

T(self) == Start(self) \/ D2(self) \/ D3(self) \/ D4(self) \/ D5(self) \/ D6(self) \/ D12(self) \/ D13(self) \/ D19(self)
              \/ D10(self) \/ E1(self) \/ E2(self) \/ E5(self) \/ E6(self) \/ E7(self) \/ E8(self) \/ E9(self) \/ E17(self)
              \/ E13(self) \/ Advance(self) \/ Fin(self)

Next == (\E self \in 1..N: T(self))
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

HeadLive == Q.Head.ptr \notin free
TailLive == Q.Tail.ptr \notin free
PointersAreNodes == Q.Head.ptr \in 1..K /\ Q.Tail.ptr \in 1..K /\ \A n \in 1..K : mem[n].next.ptr \in 0..K
TailAtMostOneBehind == mem[Q.Tail.ptr].next.ptr = 0 \/ mem[mem[Q.Tail.ptr].next.ptr].next.ptr = 0 \/ Q.Tail.ptr \in free
CountsGrow == \A p \in 1..N : head[p].count <= Q.Head.count /\ tail[p].count <= Q.Tail.count
=============================================================================
