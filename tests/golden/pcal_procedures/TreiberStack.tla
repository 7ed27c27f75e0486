----------------------------- MODULE TreiberStack -----------------------------
(* HAND-WRITTEN fixture: the translation pcal2tla gives for specs/pluscal/treiber_procs.tla (p-manual section 3.5 / App. B): the
   procedures' variables are functions on ProcSet, `stack[self]` is a sequence of frames [procedure, pc, the procedure's variables as
   they were before the call]; `call` pushes a frame, sets the parameters and (re)initialises the procedure's variables; `return` pops
   the frame and restores them.  tests/test_pcal.py evaluates THIS module with the general TLA+ evaluator and compares counters and
   per-level counts with the product's expansion of the same procedures (evaluated, compiled for the host, compiled for the GPU). *)
EXTENDS Naturals, Sequences
CONSTANT N, defaultInitValue
VARIABLES head, nxt, got, pc, stack, node, old, top, nx

vars == << head, nxt, got, pc, stack, node, old, top, nx >>

ProcSet == (1..N)

Init == /\ head = 0
        /\ nxt = [i \in 1..N |-> 0]
        /\ got = [i \in 1..N |-> 0]
        /\ node = [ self \in ProcSet |-> defaultInitValue]
        /\ old = [ self \in ProcSet |-> 0]
        /\ top = [ self \in ProcSet |-> 0]
        /\ nx = [ self \in ProcSet |-> 0]
        /\ stack = [self \in ProcSet |-> << >>]
        /\ pc = [self \in ProcSet |-> "W1"]

PU1(self) == /\ pc[self] = "PU1"
             /\ old' = [old EXCEPT ![self] = head]
             /\ pc' = [pc EXCEPT ![self] = "PU2"]
             /\ UNCHANGED << head, nxt, got, stack, node, top, nx >>

PU2(self) == /\ pc[self] = "PU2"
             /\ nxt' = [nxt EXCEPT ![node[self]] = old[self]]
             /\ pc' = [pc EXCEPT ![self] = "PU3"]
             /\ UNCHANGED << head, got, stack, node, old, top, nx >>

PU3(self) == /\ pc[self] = "PU3"
             /\ IF head = old[self]
                   THEN /\ head' = node[self]
                        /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
                        /\ old' = [old EXCEPT ![self] = Head(stack[self]).old]
                        /\ node' = [node EXCEPT ![self] = Head(stack[self]).node]
                        /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]
                   ELSE /\ pc' = [pc EXCEPT ![self] = "PU1"]
                        /\ UNCHANGED << head, stack, node, old >>
             /\ UNCHANGED << nxt, got, top, nx >>

push(self) == PU1(self) \/ PU2(self) \/ PU3(self)

PO1(self) == /\ pc[self] = "PO1"
             /\ top' = [top EXCEPT ![self] = head]
             /\ pc' = [pc EXCEPT ![self] = "PO2"]
             /\ UNCHANGED << head, nxt, got, stack, node, old, nx >>

PO2(self) == /\ pc[self] = "PO2"
             /\ IF top[self] = 0
                   THEN /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
                        /\ top' = [top EXCEPT ![self] = Head(stack[self]).top]
                        /\ nx' = [nx EXCEPT ![self] = Head(stack[self]).nx]
                        /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]
                   ELSE /\ pc' = [pc EXCEPT ![self] = "PO3"]
                        /\ UNCHANGED << stack, top, nx >>
             /\ UNCHANGED << head, nxt, got, node, old >>

PO3(self) == /\ pc[self] = "PO3"
             /\ nx' = [nx EXCEPT ![self] = nxt[top[self]]]
             /\ pc' = [pc EXCEPT ![self] = "PO4"]
             /\ UNCHANGED << head, nxt, got, stack, node, old, top >>

PO4(self) == /\ pc[self] = "PO4"
             /\ IF head = top[self]
                   THEN /\ head' = nx[self]
                        /\ got' = [got EXCEPT ![self] = top[self]]
                        /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
                        /\ top' = [top EXCEPT ![self] = Head(stack[self]).top]
                        /\ nx' = [nx EXCEPT ![self] = Head(stack[self]).nx]
                        /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]
                   ELSE /\ pc' = [pc EXCEPT ![self] = "PO1"]
                        /\ UNCHANGED << head, got, stack, top, nx >>
             /\ UNCHANGED << nxt, node, old >>

pop(self) == PO1(self) \/ PO2(self) \/ PO3(self) \/ PO4(self)

W1(self) == /\ pc[self] = "W1"
            /\ /\ node' = [node EXCEPT ![self] = self]
               /\ stack' = [stack EXCEPT ![self] = << [ procedure |->  "push",
                                                        pc        |->  "W2",
                                                        old       |->  old[self],
                                                        node      |->  node[self] ] >>
                                                    \o stack[self]]
            /\ old' = [old EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "PU1"]
            /\ UNCHANGED << head, nxt, got, top, nx >>

W2(self) == /\ pc[self] = "W2"
            /\ stack' = [stack EXCEPT ![self] = << [ procedure |->  "pop",
                                                     pc        |->  "W3",
                                                     top       |->  top[self],
                                                     nx        |->  nx[self] ] >>
                                                 \o stack[self]]
            /\ top' = [top EXCEPT ![self] = 0]
            /\ nx' = [nx EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "PO1"]
            /\ UNCHANGED << head, nxt, got, node, old >>

W3(self) == /\ pc[self] = "W3"
            /\ TRUE
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << head, nxt, got, stack, node, old, top, nx >>

w(self) == W1(self) \/ W2(self) \/ W3(self)

Next == (\E self \in ProcSet: push(self) \/ pop(self))
           \/ (\E self \in 1..N: w(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

PopsDistinct == \A i \in 1..N : \A j \in 1..N : (i # j /\ got[i] # 0) => got[i] # got[j]
=============================================================================
