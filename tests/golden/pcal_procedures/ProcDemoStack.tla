---------------------------- MODULE ProcDemoStack ----------------------------
(* HAND-WRITTEN fixture: the translation pcal2tla gives for specs/pluscal/proc_demo.tla (p-manual section 3.5 / App. B: a `stack` of
   frames [procedure, pc, the procedure's variables as they were before the call]; `call` pushes a frame, sets the parameters and
   (re)initialises the procedure's variables; `return` pops it and restores them).  tests/test_pcal.py evaluates this module with the
   general TLA+ evaluator and compares the state graph's counters with the product's EXPANSION of the same procedures. *)
EXTENDS Naturals, Sequences
CONSTANT defaultInitValue
VARIABLES total, pc, stack, n, t

vars == << total, pc, stack, n, t >>

ProcSet == (1..2)

Init == /\ total = 0
        /\ n = [ self \in ProcSet |-> defaultInitValue]
        /\ t = [ self \in ProcSet |-> 0]
        /\ stack = [self \in ProcSet |-> << >>]
        /\ pc = [self \in ProcSet |-> "P1"]

A1(self) == /\ pc[self] = "A1"
            /\ t' = [t EXCEPT ![self] = total]
            /\ pc' = [pc EXCEPT ![self] = "A2"]
            /\ UNCHANGED << total, stack, n >>

A2(self) == /\ pc[self] = "A2"
            /\ total' = t[self] + n[self]
            /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
            /\ t' = [t EXCEPT ![self] = Head(stack[self]).t]
            /\ n' = [n EXCEPT ![self] = Head(stack[self]).n]
            /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]

add(self) == A1(self) \/ A2(self)

P1(self) == /\ pc[self] = "P1"
            /\ /\ n' = [n EXCEPT ![self] = self]
               /\ stack' = [stack EXCEPT ![self] = << [ procedure |->  "add",
                                                        pc        |->  "P2",
                                                        t         |->  t[self],
                                                        n         |->  n[self] ] >>
                                                    \o stack[self]]
            /\ t' = [t EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "A1"]
            /\ total' = total

P2(self) == /\ pc[self] = "P2"
            /\ /\ n' = [n EXCEPT ![self] = 10]
               /\ stack' = [stack EXCEPT ![self] = << [ procedure |->  "add",
                                                        pc        |->  "P3",
                                                        t         |->  t[self],
                                                        n         |->  n[self] ] >>
                                                    \o stack[self]]
            /\ t' = [t EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "A1"]
            /\ total' = total

P3(self) == /\ pc[self] = "P3"
            /\ TRUE
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << total, stack, n, t >>

p(self) == P1(self) \/ P2(self) \/ P3(self)

Next == (\E self \in ProcSet: add(self))
           \/ (\E self \in 1..2: p(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars
=============================================================================
