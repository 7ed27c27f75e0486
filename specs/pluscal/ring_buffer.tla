----------------------------- MODULE ring_buffer -----------------------------
(* A single-producer / single-consumer lock-free ring buffer: K slots, each a RECORD (full flag, item); the producer fills the
   slot at its private tail index and only then raises the flag, the consumer copies the item out and gives the slot back by
   assigning the whole record.  The ring buffer of the reference's roadmap (README.md:26-42).  With `Torn = TRUE` the producer
   raises the flag BEFORE writing the item: the consumer can read a torn slot and the assert fails. *)
EXTENDS Naturals, Sequences
CONSTANTS K, Items, Torn

(* --algorithm ring_buffer
variables
  buf = [i \in 0..K-1 |-> [full |-> FALSE, item |-> 0]],
  got = <<>>;

process producer = 0
variables t = 0, n = 1;
begin
  P0: while n <= Items do
        await ~buf[t].full;
        if Torn then
          buf[t].full := TRUE;
  P1:     buf[t].item := n;
        else
          buf[t].item := n;
  P2:     buf[t].full := TRUE;
        end if;
  P3:   t := (t + 1) % K || n := n + 1;
      end while;
end process

process consumer = 1
variables h = 0, x = 0;
begin
  C0: while Len(got) < Items do
        await buf[h].full;
        x := buf[h].item;
        assert x # 0;
  C1:   buf[h] := [full |-> FALSE, item |-> 0];
        got := Append(got, x);
        h := (h + 1) % K;
      end while;
end process
end algorithm *)
\* BEGIN TRANSLATION
VARIABLES buf_full, buf_item, got, pc, t, n, h, x

vars == << buf_full, buf_item, got, pc, t, n, h, x >>

(* record variables are kept field by field: r.f is r_f *)
buf == [i \in 0..K - 1 |-> [full |-> buf_full[i], item |-> buf_item[i]]]

ProcSet == {0} \cup {1}

Init == (* Global variables *)
        /\ buf_full = [i \in 0..K - 1 |-> FALSE]
        /\ buf_item = [i \in 0..K - 1 |-> 0]
        /\ got = <<>>
        (* Process producer *)
        /\ t = 0
        /\ n = 1
        (* Process consumer *)
        /\ h = 0
        /\ x = 0
        /\ pc = [self \in ProcSet |-> CASE self = 0 -> "P0"
                                        [] self = 1 -> "C0"]

P0 == /\ pc[0] = "P0"
      /\ IF n <= Items
            THEN /\ ~buf_full[t]
                 /\ IF Torn
                       THEN /\ buf_full' = [buf_full EXCEPT ![t] = TRUE]
                            /\ pc' = [pc EXCEPT ![0] = "P1"]
                            /\ UNCHANGED buf_item
                       ELSE /\ buf_item' = [buf_item EXCEPT ![t] = n]
                            /\ pc' = [pc EXCEPT ![0] = "P2"]
                            /\ UNCHANGED buf_full
            ELSE /\ pc' = [pc EXCEPT ![0] = "Done"]
                 /\ UNCHANGED << buf_full, buf_item >>
      /\ UNCHANGED << got, t, n, h, x >>

P1 == /\ pc[0] = "P1"
      /\ buf_item' = [buf_item EXCEPT ![t] = n]
      /\ pc' = [pc EXCEPT ![0] = "P3"]
      /\ UNCHANGED << buf_full, got, t, n, h, x >>

P2 == /\ pc[0] = "P2"
      /\ buf_full' = [buf_full EXCEPT ![t] = TRUE]
      /\ pc' = [pc EXCEPT ![0] = "P3"]
      /\ UNCHANGED << buf_item, got, t, n, h, x >>

P3 == /\ pc[0] = "P3"
      /\ t' = (t + 1) % K
      /\ n' = n + 1
      /\ pc' = [pc EXCEPT ![0] = "P0"]
      /\ UNCHANGED << buf_full, buf_item, got, h, x >>

producer == P0 \/ P1 \/ P2 \/ P3

C0 == /\ pc[1] = "C0"
      /\ IF Len(got) < Items
            THEN /\ buf_full[h]
                 /\ x' = buf_item[h]
                 /\ Assert(x' # 0, 
                           "Failure of assertion at line 36, column 9.")
                 /\ pc' = [pc EXCEPT ![1] = "C1"]
            ELSE /\ pc' = [pc EXCEPT ![1] = "Done"]
                 /\ UNCHANGED x
      /\ UNCHANGED << buf_full, buf_item, got, t, n, h >>

C1 == /\ pc[1] = "C1"
      /\ buf_full' = [buf_full EXCEPT ![h] = FALSE]
      /\ buf_item' = [buf_item EXCEPT ![h] = 0]
      /\ got' = Append(got, x)
      /\ h' = (h + 1) % K
      /\ pc' = [pc EXCEPT ![1] = "C0"]
      /\ UNCHANGED << t, n, x >>

consumer == C0 \/ C1

Next == producer
           \/ consumer
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Fifo == \A i \in 1..Len(got) : got[i] = i
FullHasItem == \A i \in 0..K-1 : buf[i].full => (Torn \/ buf[i].item # 0)
EmptyIsClean == \A i \in 0..K-1 : (~buf[i].full /\ ~Torn) => (buf[i].item = 0 \/ (i = t /\ pc[0] = "P2"))
=============================================================================
