------------------------------- MODULE Mailboxes -------------------------------
(* specs/pluscal/mailboxes.tla the way pcal2tla translates it: box one function to sequences of records, log one sequence of records
   (an element replaced with EXCEPT ![1]), m record-valued.  Written by hand (tests/test_pcal.py compares it with the product's
   translation, which keeps box and log as one sequence per field). *)
EXTENDS Naturals, Sequences, TLC
CONSTANTS N
VARIABLES box, heard, log, sum, pc, m

vars == << box, heard, log, sum, pc, m >>

ProcSet == (1..N)

Init == /\ box = [p \in 1..N |-> <<>>]
        /\ heard = [p \in 1..N |-> <<>>]
        /\ log = << [kind |-> "start", from |-> 0, val |-> 0] >>
        /\ sum = 0
        /\ m = [self \in 1..N |-> [kind |-> "none", from |-> 0, val |-> 0]]
        /\ pc = [self \in ProcSet |-> "S"]

S(self) == /\ pc[self] = "S"
           /\ box' = [box EXCEPT ![(self % N) + 1] = Append(box[(self % N) + 1], [kind |-> "ping", from |-> self, val |-> self * 10])]
           /\ pc' = [pc EXCEPT ![self] = "R"]
           /\ UNCHANGED << heard, log, sum, m >>

R(self) == /\ pc[self] = "R"
           /\ box[self] # <<>>
           /\ m' = [m EXCEPT ![self] = Head(box[self])]
           /\ box' = [box EXCEPT ![self] = Tail(box[self])]
           /\ pc' = [pc EXCEPT ![self] = "A"]
           /\ UNCHANGED << heard, log, sum >>

A(self) == /\ pc[self] = "A"
           /\ IF m[self].kind = "ping"
                 THEN /\ box' = [box EXCEPT ![m[self].from] = box[m[self].from] \o << [kind |-> "pong", from |-> self, val |-> m[self].val + 1] >>]
                      /\ /\ heard' = [heard EXCEPT ![self] = Append(heard[self], m[self].from)]
                         /\ log' = Append(log, m[self])
                      /\ pc' = [pc EXCEPT ![self] = "R"]
                      /\ sum' = sum
                 ELSE /\ sum' = sum + m[self].val
                      /\ log' = [log EXCEPT ![1] = m[self]]
                      /\ pc' = [pc EXCEPT ![self] = "F"]
                      /\ UNCHANGED << box, heard >>
           /\ m' = m

F(self) == /\ pc[self] = "F"
           /\ Assert(Len(box[self]) = 0 \/ Head(box[self]).kind = "ping", "Failure of assertion at line 37, column 5.")
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << box, heard, log, sum, m >>

Node(self) == S(self) \/ R(self) \/ A(self) \/ F(self)

Next == (\E self \in 1..N: Node(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

LogOk == \A k \in 2..Len(log) : log[k].kind = "ping" /\ log[k].val = log[k].from * 10
Pongs == \A p \in 1..N : \A k \in 1..Len(box[p]) : box[p][k].kind = "pong" => box[p][k].val % 10 = 1
HeardTheLeft == \A p \in 1..N : Len(heard[p]) <= 1 /\ (heard[p] # <<>> => heard[p][1] = ((p + N - 2) % N) + 1)
=============================================================================
