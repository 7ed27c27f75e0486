------------------------------ MODULE ms_queue ------------------------------
(***************************************************************************)
(* The Michael-Scott lock-free FIFO queue: a singly linked LIST with a      *)
(* dummy node, `head` and `tail` pointers moved by compare-and-swap, a tail *)
(* that may lag one node behind and is helped forward by whoever notices.   *)
(* Nodes are 1..N+1 (node 1 is the initial dummy, thread p enqueues node    *)
(* p+1 carrying the value p), 0 is the null pointer.  Every thread enqueues *)
(* its value and then dequeues one.  (README.md:26-42 of the reference:     *)
(* lock-free stacks, LISTS and ring buffers are what it wants to model.)    *)
(* Ghost variables: `ticket` stamps the linearisation points (the CAS that  *)
(* links a node; the CAS that swings head), `enq_at` / `deq_at` keep them.  *)
(***************************************************************************)
EXTENDS Naturals
CONSTANTS N, Racy    \* Racy = TRUE: the link is a plain store instead of a compare-and-swap (a node gets lost)

(* --algorithm ms_queue
variables head = 1, tail = 1,
          next = [n \in 1..N+1 |-> 0],
          val = [n \in 1..N+1 |-> 0],
          ticket = 0,
          enq_at = [v \in 1..N |-> 0],
          deq_at = [v \in 1..N |-> 0];

process T \in 1..N
  variables h = 0, t = 0, nx = 0, got = 0;
begin
  EnqInit: val[self + 1] := self;
  EnqTail: t := tail;
  EnqNext: nx := next[t];
  EnqCheck:
    if t # tail then
      goto EnqTail;
    elsif nx # 0 then
      goto EnqHelp;
    end if;
  EnqLink:
    if Racy \/ next[t] = 0 then
      next[t] := self + 1;
      ticket := ticket + 1;
      enq_at[self] := ticket;
    else
      goto EnqTail;
    end if;
  EnqSwing:
    if tail = t then
      tail := self + 1;
    end if;
    goto DeqHead;
  EnqHelp:
    if tail = t then
      tail := nx;
    end if;
    goto EnqTail;
  DeqHead: h := head;
  DeqTail: t := tail;
  DeqNext: nx := next[h];
  DeqCheck:
    if h # head then
      goto DeqHead;
    elsif h = t /\ nx = 0 then
      goto Fin;
    elsif h = t then
      goto DeqHelp;
    end if;
  DeqRead: got := val[nx];
  DeqSwing:
    if head = h then
      head := nx;
      ticket := ticket + 1;
      deq_at[got] := ticket;
      goto Fin;
    else
      got := 0;
      goto DeqHead;
    end if;
  DeqHelp:
    if tail = t then
      tail := nx;
    end if;
    goto DeqHead;
  Fin: skip;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES head, tail, next, val, ticket, enq_at, deq_at, pc, h, t, nx, got

vars == << head, tail, next, val, ticket, enq_at, deq_at, pc, h, t, nx, got >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ head = 1
        /\ tail = 1
        /\ next = [n \in 1..N + 1 |-> 0]
        /\ val = [n \in 1..N + 1 |-> 0]
        /\ ticket = 0
        /\ enq_at = [v \in 1..N |-> 0]
        /\ deq_at = [v \in 1..N |-> 0]
        (* Process T *)
        /\ h = [self \in 1..N |-> 0]
        /\ t = [self \in 1..N |-> 0]
        /\ nx = [self \in 1..N |-> 0]
        /\ got = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "EnqInit"]

EnqInit(self) == /\ pc[self] = "EnqInit"
                 /\ val' = [val EXCEPT ![self + 1] = self]
                 /\ pc' = [pc EXCEPT ![self] = "EnqTail"]
                 /\ UNCHANGED << head, tail, next, ticket, enq_at, deq_at, h, 
                                 t, nx, got >>

EnqTail(self) == /\ pc[self] = "EnqTail"
                 /\ t' = [t EXCEPT ![self] = tail]
                 /\ pc' = [pc EXCEPT ![self] = "EnqNext"]
                 /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                 deq_at, h, nx, got >>

EnqNext(self) == /\ pc[self] = "EnqNext"
                 /\ nx' = [nx EXCEPT ![self] = next[t[self]]]
                 /\ pc' = [pc EXCEPT ![self] = "EnqCheck"]
                 /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                 deq_at, h, t, got >>

EnqCheck(self) == /\ pc[self] = "EnqCheck"
                  /\ IF t[self] # tail
                        THEN /\ pc' = [pc EXCEPT ![self] = "EnqTail"]
                        ELSE /\ IF nx[self] # 0
                                   THEN /\ pc' = [pc EXCEPT ![self] = "EnqHelp"]
                                   ELSE /\ pc' = [pc EXCEPT ![self] = "EnqLink"]
                  /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                  deq_at, h, t, nx, got >>

EnqLink(self) == /\ pc[self] = "EnqLink"
                 /\ IF Racy \/ next[t[self]] = 0
                       THEN /\ next' = [next EXCEPT ![t[self]] = self + 1]
                            /\ ticket' = ticket + 1
                            /\ enq_at' = [enq_at EXCEPT ![self] = ticket']
                            /\ pc' = [pc EXCEPT ![self] = "EnqSwing"]
                       ELSE /\ pc' = [pc EXCEPT ![self] = "EnqTail"]
                            /\ UNCHANGED << next, ticket, enq_at >>
                 /\ UNCHANGED << head, tail, val, deq_at, h, t, nx, got >>

EnqSwing(self) == /\ pc[self] = "EnqSwing"
                  /\ IF tail = t[self]
                        THEN /\ tail' = self + 1
                        ELSE /\ TRUE
                             /\ UNCHANGED tail
                  /\ pc' = [pc EXCEPT ![self] = "DeqHead"]
                  /\ UNCHANGED << head, next, val, ticket, enq_at, deq_at, h, 
                                  t, nx, got >>

EnqHelp(self) == /\ pc[self] = "EnqHelp"
                 /\ IF tail = t[self]
                       THEN /\ tail' = nx[self]
                       ELSE /\ TRUE
                            /\ UNCHANGED tail
                 /\ pc' = [pc EXCEPT ![self] = "EnqTail"]
                 /\ UNCHANGED << head, next, val, ticket, enq_at, deq_at, h, 
                                 t, nx, got >>

DeqHead(self) == /\ pc[self] = "DeqHead"
                 /\ h' = [h EXCEPT ![self] = head]
                 /\ pc' = [pc EXCEPT ![self] = "DeqTail"]
                 /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                 deq_at, t, nx, got >>

DeqTail(self) == /\ pc[self] = "DeqTail"
                 /\ t' = [t EXCEPT ![self] = tail]
                 /\ pc' = [pc EXCEPT ![self] = "DeqNext"]
                 /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                 deq_at, h, nx, got >>

DeqNext(self) == /\ pc[self] = "DeqNext"
                 /\ nx' = [nx EXCEPT ![self] = next[h[self]]]
                 /\ pc' = [pc EXCEPT ![self] = "DeqCheck"]
                 /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                 deq_at, h, t, got >>

DeqCheck(self) == /\ pc[self] = "DeqCheck"
                  /\ IF h[self] # head
                        THEN /\ pc' = [pc EXCEPT ![self] = "DeqHead"]
                        ELSE /\ IF h[self] = t[self] /\ nx[self] = 0
                                   THEN /\ pc' = [pc EXCEPT ![self] = "Fin"]
                                   ELSE /\ IF h[self] = t[self]
                                              THEN /\ pc' = [pc EXCEPT ![self] = "DeqHelp"]
                                              ELSE /\ pc' = [pc EXCEPT ![self] = "DeqRead"]
                  /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                  deq_at, h, t, nx, got >>

DeqRead(self) == /\ pc[self] = "DeqRead"
                 /\ got' = [got EXCEPT ![self] = val[nx[self]]]
                 /\ pc' = [pc EXCEPT ![self] = "DeqSwing"]
                 /\ UNCHANGED << head, tail, next, val, ticket, enq_at, 
                                 deq_at, h, t, nx >>

DeqSwing(self) == /\ pc[self] = "DeqSwing"
                  /\ IF head = h[self]
                        THEN /\ head' = nx[self]
                             /\ ticket' = ticket + 1
                             /\ deq_at' = [deq_at EXCEPT ![got[self]] = ticket']
                             /\ pc' = [pc EXCEPT ![self] = "Fin"]
                             /\ UNCHANGED got
                        ELSE /\ got' = [got EXCEPT ![self] = 0]
                             /\ pc' = [pc EXCEPT ![self] = "DeqHead"]
                             /\ UNCHANGED << head, ticket, deq_at >>
                  /\ UNCHANGED << tail, next, val, enq_at, h, t, nx >>

DeqHelp(self) == /\ pc[self] = "DeqHelp"
                 /\ IF tail = t[self]
                       THEN /\ tail' = nx[self]
                       ELSE /\ TRUE
                            /\ UNCHANGED tail
                 /\ pc' = [pc EXCEPT ![self] = "DeqHead"]
                 /\ UNCHANGED << head, next, val, ticket, enq_at, deq_at, h, 
                                 t, nx, got >>

Fin(self) == /\ pc[self] = "Fin"
             /\ TRUE
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << head, tail, next, val, ticket, enq_at, deq_at, 
                             h, t, nx, got >>

T(self) == EnqInit(self) \/ EnqTail(self) \/ EnqNext(self) \/ EnqCheck(self) \/ EnqLink(self) \/ EnqSwing(self) \/ EnqHelp(self) \/ DeqHead(self) \/ DeqTail(self) \/ DeqNext(self) \/ DeqCheck(self) \/ DeqRead(self) \/ DeqSwing(self) \/ DeqHelp(self) \/ Fin(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

AllDone == \A p \in 1..N : pc[p] = "Done"
PointersAreNodes == head \in 1..N+1 /\ tail \in 1..N+1 /\ \A n \in 1..N+1 : next[n] \in 0..N+1
\* the tail never lags more than one node behind the last linked node, and never behind the head's node
TailLagsByOne == next[tail] = 0 \/ next[next[tail]] = 0
\* every thread's dequeue finds a value: its own enqueue came first (a linearisable "empty" answer would be wrong)
NeverEmpty == \A p \in 1..N : pc[p] = "Done" => got[p] # 0
DequeuedOnce == \A p \in 1..N : \A q \in 1..N : (p # q /\ pc[p] = "Done" /\ pc[q] = "Done") => got[p] # got[q]
\* first in, first out at the linearisation points: a value linked earlier is never unlinked later than one linked after it
Fifo == \A a \in 1..N : \A b \in 1..N : (enq_at[a] # 0 /\ enq_at[b] # 0 /\ enq_at[a] < enq_at[b] /\ deq_at[b] # 0) => (deq_at[a] # 0 /\ deq_at[a] < deq_at[b])
Conservation == AllDone => (head = tail /\ next[tail] = 0 /\ \A v \in 1..N : deq_at[v] # 0)
=============================================================================
