--------------------------- MODULE ms_queue_counted ---------------------------
(***************************************************************************)
(* The Michael-Scott queue AS PUBLISHED (PODC 1996, Figure 1): every        *)
(* pointer is a (ptr, count) pair that a compare-and-swap replaces as a     *)
(* whole, nodes are FREED by dequeue and REUSED by enqueue, and the counts  *)
(* are what keeps a delayed compare-and-swap from succeeding on a recycled  *)
(* node (the ABA problem).  The structures of the paper are NESTED records: *)
(*   pointer_t = [ptr, count]                                               *)
(*   node_t    = [value, next : pointer_t]                                  *)
(*   queue_t   = [Head : pointer_t, Tail : pointer_t]                       *)
(* (README.md:26-42 of the reference: the lock-free list / stack / epoch GC *)
(* it wants to model all hang on pointers that carry a version.)            *)
(* Nodes are 1..K, 0 is NULL.  Initially node 1 is the dummy and node 2     *)
(* holds the value N + 1; every thread dequeues, enqueues its own id and    *)
(* dequeues again.  Counted = FALSE compares the ptr halves only: a thread  *)
(* that sleeps in front of D13 then swings Head to a node that was freed    *)
(* in the meantime (HeadLive fails).  Labels = the line numbers of Figure 1.*)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS N, K, Counted

(* --algorithm ms_queue_counted
variables Q = [Head |-> [ptr |-> 1, count |-> 0], Tail |-> [ptr |-> 2, count |-> 0]],
          mem = [n \in 1..K |-> [value |-> IF n = 2 THEN N + 1 ELSE 0,
                                 next |-> [ptr |-> IF n = 1 THEN 2 ELSE 0, count |-> 0]]],
          free = 3..K,
          taken = {};

process T \in 1..N
  variables head = [ptr |-> 0, count |-> 0], tail = [ptr |-> 0, count |-> 0], next = [ptr |-> 0, count |-> 0],
            node = 0, got = 0, phase = 0;
begin
  Start:
    if phase = 1 then
      goto E1;
    elsif phase = 3 then
      goto Fin;
    end if;
  D2: head := Q.Head;
  D3: tail := Q.Tail;
  D4: next := mem[head.ptr].next;
  D5:
    if head # Q.Head then
      goto D2;
    end if;
  D6:
    if head.ptr = tail.ptr then
      if next.ptr = 0 then
        got := 0;
        goto Advance;
      else
        goto D10;
      end if;
    end if;
  D12: got := mem[next.ptr].value;
  D13:
    if (Counted /\ Q.Head = head) \/ (~Counted /\ Q.Head.ptr = head.ptr) then
      Q.Head := [ptr |-> next.ptr, count |-> head.count + 1];
    else
      goto D2;
    end if;
  D19:
    assert got \notin taken;
    taken := taken \cup {got};
    free := free \cup {head.ptr};
    goto Advance;
  D10:
    if (Counted /\ Q.Tail = tail) \/ (~Counted /\ Q.Tail.ptr = tail.ptr) then
      Q.Tail := [ptr |-> next.ptr, count |-> tail.count + 1];
    end if;
    goto D2;
  E1:
    with n \in free do
      node := n;
      free := free \ {n};
    end with;
  E2: mem[node].value := self || mem[node].next.ptr := 0;
  E5: tail := Q.Tail;
  E6: next := mem[tail.ptr].next;
  E7:
    if tail # Q.Tail then
      goto E5;
    end if;
  E8:
    if next.ptr # 0 then
      goto E13;
    end if;
  E9:
    if (Counted /\ mem[tail.ptr].next = next) \/ (~Counted /\ mem[tail.ptr].next.ptr = next.ptr) then
      mem[tail.ptr].next := [ptr |-> node, count |-> next.count + 1];
    else
      goto E5;
    end if;
  E17:
    if (Counted /\ Q.Tail = tail) \/ (~Counted /\ Q.Tail.ptr = tail.ptr) then
      Q.Tail := [ptr |-> node, count |-> tail.count + 1];
    end if;
    goto Advance;
  E13:
    if (Counted /\ Q.Tail = tail) \/ (~Counted /\ Q.Tail.ptr = tail.ptr) then
      Q.Tail := [ptr |-> next.ptr, count |-> tail.count + 1];
    end if;
    goto E5;
  Advance:
    phase := phase + 1;
    goto Start;
  Fin: skip;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES Q_Head_ptr, Q_Head_count, Q_Tail_ptr, Q_Tail_count, mem_value, mem_next_ptr, mem_next_count, free, taken, pc, head_ptr, head_count, tail_ptr, tail_count, next_ptr, next_count, node, got, phase

vars == << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, Q_Tail_count, mem_value, mem_next_ptr, mem_next_count, free, taken, pc, head_ptr, head_count, tail_ptr, tail_count, next_ptr, next_count, node, got, phase >>

(* record variables are kept field by field: r.f is r_f *)
Q_Head == [ptr |-> Q_Head_ptr, count |-> Q_Head_count]
Q_Tail == [ptr |-> Q_Tail_ptr, count |-> Q_Tail_count]
mem_next == [n \in 1..K |-> [ptr |-> mem_next_ptr[n], count |-> mem_next_count[n]]]
Q == [Head |-> Q_Head, Tail |-> Q_Tail]
mem == [n \in 1..K |-> [value |-> mem_value[n], next |-> mem_next[n]]]
head == [self \in 1..N |-> [ptr |-> head_ptr[self], count |-> head_count[self]]]
tail == [self \in 1..N |-> [ptr |-> tail_ptr[self], count |-> tail_count[self]]]
next == [self \in 1..N |-> [ptr |-> next_ptr[self], count |-> next_count[self]]]

ProcSet == (1..N)

Init == (* Global variables *)
        /\ Q_Head_ptr = 1
        /\ Q_Head_count = 0
        /\ Q_Tail_ptr = 2
        /\ Q_Tail_count = 0
        /\ mem_value = [n \in 1..K |-> IF n = 2 THEN N + 1 ELSE 0]
        /\ mem_next_ptr = [n \in 1..K |-> IF n = 1 THEN 2 ELSE 0]
        /\ mem_next_count = [n \in 1..K |-> 0]
        /\ free = 3..K
        /\ taken = {}
        (* Process T *)
        /\ head_ptr = [self \in 1..N |-> 0]
        /\ head_count = [self \in 1..N |-> 0]
        /\ tail_ptr = [self \in 1..N |-> 0]
        /\ tail_count = [self \in 1..N |-> 0]
        /\ next_ptr = [self \in 1..N |-> 0]
        /\ next_count = [self \in 1..N |-> 0]
        /\ node = [self \in 1..N |-> 0]
        /\ got = [self \in 1..N |-> 0]
        /\ phase = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "Start"]

Start(self) == /\ pc[self] = "Start"
               /\ IF phase[self] = 1
                     THEN /\ pc' = [pc EXCEPT ![self] = "E1"]
                     ELSE /\ IF phase[self] = 3
                                THEN /\ pc' = [pc EXCEPT ![self] = "Fin"]
                                ELSE /\ pc' = [pc EXCEPT ![self] = "D2"]
               /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                               Q_Tail_count, mem_value, mem_next_ptr, 
                               mem_next_count, free, taken, head_ptr, 
                               head_count, tail_ptr, tail_count, next_ptr, 
                               next_count, node, got, phase >>

D2(self) == /\ pc[self] = "D2"
            /\ head_ptr' = [head_ptr EXCEPT ![self] = Q_Head_ptr]
            /\ head_count' = [head_count EXCEPT ![self] = Q_Head_count]
            /\ pc' = [pc EXCEPT ![self] = "D3"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, tail_ptr, 
                            tail_count, next_ptr, next_count, node, got, 
                            phase >>

D3(self) == /\ pc[self] = "D3"
            /\ tail_ptr' = [tail_ptr EXCEPT ![self] = Q_Tail_ptr]
            /\ tail_count' = [tail_count EXCEPT ![self] = Q_Tail_count]
            /\ pc' = [pc EXCEPT ![self] = "D4"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, next_ptr, next_count, node, got, 
                            phase >>

D4(self) == /\ pc[self] = "D4"
            /\ next_ptr' = [next_ptr EXCEPT ![self] = mem_next_ptr[head_ptr[self]]]
            /\ next_count' = [next_count EXCEPT ![self] = mem_next_count[head_ptr[self]]]
            /\ pc' = [pc EXCEPT ![self] = "D5"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, node, got, 
                            phase >>

D5(self) == /\ pc[self] = "D5"
            /\ IF (~(head_ptr[self] = Q_Head_ptr /\ head_count[self] = Q_Head_count))
                  THEN /\ pc' = [pc EXCEPT ![self] = "D2"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "D6"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, next_ptr, 
                            next_count, node, got, phase >>

D6(self) == /\ pc[self] = "D6"
            /\ IF head_ptr[self] = tail_ptr[self]
                  THEN /\ IF next_ptr[self] = 0
                             THEN /\ got' = [got EXCEPT ![self] = 0]
                                  /\ pc' = [pc EXCEPT ![self] = "Advance"]
                             ELSE /\ pc' = [pc EXCEPT ![self] = "D10"]
                                  /\ UNCHANGED got
                  ELSE /\ pc' = [pc EXCEPT ![self] = "D12"]
                       /\ UNCHANGED got
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, next_ptr, 
                            next_count, node, phase >>

D12(self) == /\ pc[self] = "D12"
             /\ got' = [got EXCEPT ![self] = mem_value[next_ptr[self]]]
             /\ pc' = [pc EXCEPT ![self] = "D13"]
             /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                             Q_Tail_count, mem_value, mem_next_ptr, 
                             mem_next_count, free, taken, head_ptr, 
                             head_count, tail_ptr, tail_count, next_ptr, 
                             next_count, node, phase >>

D13(self) == /\ pc[self] = "D13"
             /\ IF (Counted /\ (Q_Head_ptr = head_ptr[self] /\ Q_Head_count = head_count[self])) \/ (~Counted /\ Q_Head_ptr = head_ptr[self])
                   THEN /\ Q_Head_ptr' = next_ptr[self]
                        /\ Q_Head_count' = head_count[self] + 1
                        /\ pc' = [pc EXCEPT ![self] = "D19"]
                   ELSE /\ pc' = [pc EXCEPT ![self] = "D2"]
                        /\ UNCHANGED << Q_Head_ptr, Q_Head_count >>
             /\ UNCHANGED << Q_Tail_ptr, Q_Tail_count, mem_value, 
                             mem_next_ptr, mem_next_count, free, taken, 
                             head_ptr, head_count, tail_ptr, tail_count, 
                             next_ptr, next_count, node, got, phase >>

D19(self) == /\ pc[self] = "D19"
             /\ Assert(got[self] \notin taken, 
                       "Failure of assertion at line 63, column 5.")
             /\ taken' = taken \cup {got[self]}
             /\ free' = free \cup {head_ptr[self]}
             /\ pc' = [pc EXCEPT ![self] = "Advance"]
             /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                             Q_Tail_count, mem_value, mem_next_ptr, 
                             mem_next_count, head_ptr, head_count, tail_ptr, 
                             tail_count, next_ptr, next_count, node, got, 
                             phase >>

D10(self) == /\ pc[self] = "D10"
             /\ IF (Counted /\ (Q_Tail_ptr = tail_ptr[self] /\ Q_Tail_count = tail_count[self])) \/ (~Counted /\ Q_Tail_ptr = tail_ptr[self])
                   THEN /\ Q_Tail_ptr' = next_ptr[self]
                        /\ Q_Tail_count' = tail_count[self] + 1
                   ELSE /\ TRUE
                        /\ UNCHANGED << Q_Tail_ptr, Q_Tail_count >>
             /\ pc' = [pc EXCEPT ![self] = "D2"]
             /\ UNCHANGED << Q_Head_ptr, Q_Head_count, mem_value, 
                             mem_next_ptr, mem_next_count, free, taken, 
                             head_ptr, head_count, tail_ptr, tail_count, 
                             next_ptr, next_count, node, got, phase >>

E1(self) == /\ pc[self] = "E1"
            /\ \E n \in free:
                 /\ node' = [node EXCEPT ![self] = n]
                 /\ free' = free \ {n}
            /\ pc' = [pc EXCEPT ![self] = "E2"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, taken, head_ptr, head_count, 
                            tail_ptr, tail_count, next_ptr, next_count, got, 
                            phase >>

E2(self) == /\ pc[self] = "E2"
            /\ mem_value' = [mem_value EXCEPT ![node[self]] = self]
            /\ mem_next_ptr' = [mem_next_ptr EXCEPT ![node[self]] = 0]
            /\ pc' = [pc EXCEPT ![self] = "E5"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_next_count, free, taken, 
                            head_ptr, head_count, tail_ptr, tail_count, 
                            next_ptr, next_count, node, got, phase >>

E5(self) == /\ pc[self] = "E5"
            /\ tail_ptr' = [tail_ptr EXCEPT ![self] = Q_Tail_ptr]
            /\ tail_count' = [tail_count EXCEPT ![self] = Q_Tail_count]
            /\ pc' = [pc EXCEPT ![self] = "E6"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, next_ptr, next_count, node, got, 
                            phase >>

E6(self) == /\ pc[self] = "E6"
            /\ next_ptr' = [next_ptr EXCEPT ![self] = mem_next_ptr[tail_ptr[self]]]
            /\ next_count' = [next_count EXCEPT ![self] = mem_next_count[tail_ptr[self]]]
            /\ pc' = [pc EXCEPT ![self] = "E7"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, node, got, 
                            phase >>

E7(self) == /\ pc[self] = "E7"
            /\ IF (~(tail_ptr[self] = Q_Tail_ptr /\ tail_count[self] = Q_Tail_count))
                  THEN /\ pc' = [pc EXCEPT ![self] = "E5"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "E8"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, next_ptr, 
                            next_count, node, got, phase >>

E8(self) == /\ pc[self] = "E8"
            /\ IF next_ptr[self] # 0
                  THEN /\ pc' = [pc EXCEPT ![self] = "E13"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "E9"]
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, mem_next_ptr, 
                            mem_next_count, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, next_ptr, 
                            next_count, node, got, phase >>

E9(self) == /\ pc[self] = "E9"
            /\ IF (Counted /\ (mem_next_ptr[tail_ptr[self]] = next_ptr[self] /\ mem_next_count[tail_ptr[self]] = next_count[self])) \/ (~Counted /\ mem_next_ptr[tail_ptr[self]] = next_ptr[self])
                  THEN /\ mem_next_ptr' = [mem_next_ptr EXCEPT ![tail_ptr[self]] = node[self]]
                       /\ mem_next_count' = [mem_next_count EXCEPT ![tail_ptr[self]] = next_count[self] + 1]
                       /\ pc' = [pc EXCEPT ![self] = "E17"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "E5"]
                       /\ UNCHANGED << mem_next_ptr, mem_next_count >>
            /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                            Q_Tail_count, mem_value, free, taken, head_ptr, 
                            head_count, tail_ptr, tail_count, next_ptr, 
                            next_count, node, got, phase >>

E17(self) == /\ pc[self] = "E17"
             /\ IF (Counted /\ (Q_Tail_ptr = tail_ptr[self] /\ Q_Tail_count = tail_count[self])) \/ (~Counted /\ Q_Tail_ptr = tail_ptr[self])
                   THEN /\ Q_Tail_ptr' = node[self]
                        /\ Q_Tail_count' = tail_count[self] + 1
                   ELSE /\ TRUE
                        /\ UNCHANGED << Q_Tail_ptr, Q_Tail_count >>
             /\ pc' = [pc EXCEPT ![self] = "Advance"]
             /\ UNCHANGED << Q_Head_ptr, Q_Head_count, mem_value, 
                             mem_next_ptr, mem_next_count, free, taken, 
                             head_ptr, head_count, tail_ptr, tail_count, 
                             next_ptr, next_count, node, got, phase >>

E13(self) == /\ pc[self] = "E13"
             /\ IF (Counted /\ (Q_Tail_ptr = tail_ptr[self] /\ Q_Tail_count = tail_count[self])) \/ (~Counted /\ Q_Tail_ptr = tail_ptr[self])
                   THEN /\ Q_Tail_ptr' = next_ptr[self]
                        /\ Q_Tail_count' = tail_count[self] + 1
                   ELSE /\ TRUE
                        /\ UNCHANGED << Q_Tail_ptr, Q_Tail_count >>
             /\ pc' = [pc EXCEPT ![self] = "E5"]
             /\ UNCHANGED << Q_Head_ptr, Q_Head_count, mem_value, 
                             mem_next_ptr, mem_next_count, free, taken, 
                             head_ptr, head_count, tail_ptr, tail_count, 
                             next_ptr, next_count, node, got, phase >>

Advance(self) == /\ pc[self] = "Advance"
                 /\ phase' = [phase EXCEPT ![self] = phase[self] + 1]
                 /\ pc' = [pc EXCEPT ![self] = "Start"]
                 /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                                 Q_Tail_count, mem_value, mem_next_ptr, 
                                 mem_next_count, free, taken, head_ptr, 
                                 head_count, tail_ptr, tail_count, next_ptr, 
                                 next_count, node, got >>

Fin(self) == /\ pc[self] = "Fin"
             /\ TRUE
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << Q_Head_ptr, Q_Head_count, Q_Tail_ptr, 
                             Q_Tail_count, mem_value, mem_next_ptr, 
                             mem_next_count, free, taken, head_ptr, 
                             head_count, tail_ptr, tail_count, next_ptr, 
                             next_count, node, got, phase >>

T(self) == Start(self) \/ D2(self) \/ D3(self) \/ D4(self) \/ D5(self) \/ D6(self) \/ D12(self) \/ D13(self) \/ D19(self) \/ D10(self) \/ E1(self) \/ E2(self) \/ E5(self) \/ E6(self) \/ E7(self) \/ E8(self) \/ E9(self) \/ E17(self) \/ E13(self) \/ Advance(self) \/ Fin(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

HeadLive == Q.Head.ptr \notin free
TailLive == Q.Tail.ptr \notin free
PointersAreNodes == Q.Head.ptr \in 1..K /\ Q.Tail.ptr \in 1..K /\ \A n \in 1..K : mem[n].next.ptr \in 0..K
TailAtMostOneBehind == mem[Q.Tail.ptr].next.ptr = 0 \/ mem[mem[Q.Tail.ptr].next.ptr].next.ptr = 0 \/ Q.Tail.ptr \in free
CountsGrow == \A p \in 1..N : head[p].count <= Q.Head.count /\ tail[p].count <= Q.Tail.count
=============================================================================
