---------------------------- MODULE RecursiveSumStack ----------------------------
(* HAND-WRITTEN fixture: the translation pcal2tla gives for specs/pluscal/recursive_sum.tla (p-manual section 3.5 / App. B: a `stack` of
   frames [procedure, pc, the procedure's variables as they were before the call]; `call` pushes a frame, sets the parameter and
   (re)initialises the local; `return` pops the frame and restores both).  The procedure is RECURSIVE: the stack grows to N + 1 frames.
   tests/test_pcal.py evaluates this module with the general TLA+ evaluator and compares the state graph's counters with the product's
   bounded-stack compilation of the same algorithm. *)
EXTENDS Naturals, Sequences, TLC
CONSTANT N, defaultInitValue
VARIABLES acc, turn, pc, stack, n, kept

vars == << acc, turn, pc, stack, n, kept >>

ProcSet == (1..2)

Init == /\ acc = [q \in 1..2 |-> 0]
        /\ turn = 0
        /\ n = [ self \in ProcSet |-> defaultInitValue]
        /\ kept = [ self \in ProcSet |-> 0]
        /\ stack = [self \in ProcSet |-> << >>]
        /\ pc = [self \in ProcSet |-> "P1"]

D1(self) == /\ pc[self] = "D1"
            /\ IF n[self] = 0
                  THEN /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
                       /\ kept' = [kept EXCEPT ![self] = Head(stack[self]).kept]
                       /\ n' = [n EXCEPT ![self] = Head(stack[self]).n]
                       /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "D2"]
                       /\ UNCHANGED << stack, n, kept >>
            /\ UNCHANGED << acc, turn >>

D2(self) == /\ pc[self] = "D2"
            /\ kept' = [kept EXCEPT ![self] = n[self]]
            /\ turn' = turn + 1
            /\ pc' = [pc EXCEPT ![self] = "D3"]
            /\ UNCHANGED << acc, stack, n >>

D3(self) == /\ pc[self] = "D3"
            /\ /\ n' = [n EXCEPT ![self] = n[self] - 1]
               /\ stack' = [stack EXCEPT ![self] = << [ procedure |->  "down",
                                                        pc        |->  "D4",
                                                        kept      |->  kept[self],
                                                        n         |->  n[self] ] >>
                                                    \o stack[self]]
            /\ kept' = [kept EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "D1"]
            /\ UNCHANGED << acc, turn >>

D4(self) == /\ pc[self] = "D4"
            /\ acc' = [acc EXCEPT ![self] = acc[self] + kept[self]]
            /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
            /\ kept' = [kept EXCEPT ![self] = Head(stack[self]).kept]
            /\ n' = [n EXCEPT ![self] = Head(stack[self]).n]
            /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]
            /\ turn' = turn

down(self) == D1(self) \/ D2(self) \/ D3(self) \/ D4(self)

P1(self) == /\ pc[self] = "P1"
            /\ /\ n' = [n EXCEPT ![self] = N]
               /\ stack' = [stack EXCEPT ![self] = << [ procedure |->  "down",
                                                        pc        |->  "P2",
                                                        kept      |->  kept[self],
                                                        n         |->  n[self] ] >>
                                                    \o stack[self]]
            /\ kept' = [kept EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "D1"]
            /\ UNCHANGED << acc, turn >>

P2(self) == /\ pc[self] = "P2"
            /\ Assert(acc[self] * 2 = N * (N + 1), "Failure of assertion at line 27, column 7.")
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << acc, turn, stack, n, kept >>

p(self) == P1(self) \/ P2(self)

Next == (\E self \in ProcSet: down(self))
           \/ (\E self \in 1..2: p(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Bounded == turn <= 2 * N
=============================================================================
