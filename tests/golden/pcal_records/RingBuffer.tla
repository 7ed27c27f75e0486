------------------------------ MODULE RingBuffer ------------------------------
(* specs/pluscal/ring_buffer.tla the way pcal2tla translates it: buf stays ONE variable, a function from slot numbers to records;
   a field is assigned with EXCEPT ![i].f, a slot with EXCEPT ![i] = [full |-> ..., item |-> ...].  Written by hand
   (tests/test_pcal.py compares it with the product's field-by-field translation). *)
EXTENDS Naturals, Sequences, TLC
CONSTANTS K, Items, Torn
VARIABLES buf, got, pc, t, n, h, x

vars == << buf, got, pc, t, n, h, x >>

ProcSet == {0} \cup {1}

Init == /\ buf = [i \in 0..K-1 |-> [full |-> FALSE, item |-> 0]]
        /\ got = <<>>
        /\ t = 0
        /\ n = 1
        /\ h = 0
        /\ x = 0
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "P0"
                                        [] self = 1 -> "C0"]

P0 == /\ pc[0] = "P0"
      /\ IF n <= Items
            THEN /\ ~buf[t].full
                 /\ IF Torn
                       THEN /\ buf' = [buf EXCEPT ![t].full = TRUE]
                            /\ pc' = [pc EXCEPT ![0] = "P1"]
                       ELSE /\ buf' = [buf EXCEPT ![t].item = n]
                            /\ pc' = [pc EXCEPT ![0] = "P2"]
            ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
                 /\ buf' = buf
      /\ UNCHANGED << got, t, n, h, x >>

P1 == /\ pc[0] = "P1"
      /\ buf' = [buf EXCEPT ![t].item = n]
      /\ pc' = [pc EXCEPT ![0] = "P3"]
      /\ UNCHANGED << got, t, n, h, x >>

P2 == /\ pc[0] = "P2"
      /\ buf' = [buf EXCEPT ![t].full = TRUE]
      /\ pc' = [pc EXCEPT ![0] = "P3"]
      /\ UNCHANGED << got, t, n, h, x >>

P3 == /\ pc[0] = "P3"
      /\ /\ n' = n + 1
         /\ t' = (t + 1) % K
      /\ pc' = [pc EXCEPT ![0] = "P0"]
      /\ UNCHANGED << buf, got, h, x >>

producer == P0 \/ P1 \/ P2 \/ P3

C0 == /\ pc[1] = "C0"
      /\ IF Len(got) < Items
            THEN /\ buf[h].full
                 /\ x' = buf[h].item
                 /\ Assert(x' # 0, "Failure of assertion at line 36, column 9.")
                 /\ pc' = [pc EXCEPT ![1] = "C1"]
            ELSE /\ pc' = [pc EXCEPT ![1] = "Done"]
                 /\ x' = x
      /\ UNCHANGED << buf, got, t, n, h >>

C1 == /\ pc[1] = "C1"
      /\ buf' = [buf EXCEPT ![h] = [full |-> FALSE, item |-> 0]]
      /\ got' = Append(got, x)
      /\ h' = (h + 1) % K
      /\ pc' = [pc EXCEPT ![1] = "C0"]
      /\ UNCHANGED << t, n, x >>

consumer == C0 \/ C1

Next == producer \/ consumer
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Fifo == \A i \in 1..Len(got) : got[i] = i
FullHasItem == \A i \in 0..K-1 : buf[i].full => (Torn \/ buf[i].item # 0)
EmptyIsClean == \A i \in 0..K-1 : (~buf[i].full /\ ~Torn) => (buf[i].item = 0 \/ (i = t /\ pc[0] = "P2"))
=============================================================================
