---------------------------- MODULE RecordArgsStack ----------------------------
(* specs/pluscal/record_args.tla the way pcal2tla translates it (p-manual section 3.5): ONE copy of the procedure's body, a `stack` of frames per
   process ([procedure, pc, fee, req]), the parameter `req` a record-valued variable that starts as defaultInitValue and is restored from the
   frame on `return`; `with old = biggest` is a LET.  Written by hand; tests/test_pcal.py compares it with the product's translation (the
   procedure expanded per call site, req kept field by field). *)
EXTENDS Naturals, Sequences, TLC
CONSTANTS N, defaultInitValue
VARIABLES inbox, total, biggest, pc, stack, req, fee, mine

vars == << inbox, total, biggest, pc, stack, req, fee, mine >>

ProcSet == (1..N)

Init == /\ inbox = <<>>
        /\ total = 0
        /\ biggest = [who |-> 0, amount |-> 0]
        /\ req = [self \in ProcSet |-> defaultInitValue]
        /\ fee = [self \in ProcSet |-> 1]
        /\ mine = [self \in 1..N |-> [who |-> 0, amount |-> 0]]
        /\ stack = [self \in ProcSet |-> <<>>]
        /\ pc = [self \in ProcSet |-> "C1"]

D1(self) == /\ pc[self] = "D1"
            /\ total' = total + req[self].amount - fee[self]
            /\ pc' = [pc EXCEPT ![self] = "D2"]
            /\ UNCHANGED << inbox, biggest, stack, req, fee, mine >>

D2(self) == /\ pc[self] = "D2"
            /\ LET old == biggest IN
                 IF req[self].amount > old.amount
                    THEN biggest' = req[self]
                    ELSE biggest' = [who |-> old.who, amount |-> old.amount]
            /\ pc' = [pc EXCEPT ![self] = Head(stack[self]).pc]
            /\ fee' = [fee EXCEPT ![self] = Head(stack[self]).fee]
            /\ req' = [req EXCEPT ![self] = Head(stack[self]).req]
            /\ stack' = [stack EXCEPT ![self] = Tail(stack[self])]
            /\ UNCHANGED << inbox, total, mine >>

deposit(self) == D1(self) \/ D2(self)

C1(self) == /\ pc[self] = "C1"
            /\ mine' = [mine EXCEPT ![self] = [who |-> self, amount |-> self * 5]]
            /\ inbox' = Append(inbox, mine'[self])
            /\ pc' = [pc EXCEPT ![self] = "C2"]
            /\ UNCHANGED << total, biggest, stack, req, fee >>

C2(self) == /\ pc[self] = "C2"
            /\ req' = [req EXCEPT ![self] = mine[self]]
            /\ stack' = [stack EXCEPT ![self] = << [procedure |-> "deposit", pc |-> "C3", fee |-> fee[self], req |-> req[self]] >> \o stack[self]]
            /\ fee' = [fee EXCEPT ![self] = 1]
            /\ pc' = [pc EXCEPT ![self] = "D1"]
            /\ UNCHANGED << inbox, total, biggest, mine >>

C3(self) == /\ pc[self] = "C3"
            /\ req' = [req EXCEPT ![self] = [who |-> self, amount |-> 2]]
            /\ stack' = [stack EXCEPT ![self] = << [procedure |-> "deposit", pc |-> "C4", fee |-> fee[self], req |-> req[self]] >> \o stack[self]]
            /\ fee' = [fee EXCEPT ![self] = 1]
            /\ pc' = [pc EXCEPT ![self] = "D1"]
            /\ UNCHANGED << inbox, total, biggest, mine >>

C4(self) == /\ pc[self] = "C4"
            /\ Assert(biggest.amount >= 2, "Failure of assertion at line 42, column 5.")
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << inbox, total, biggest, stack, req, fee, mine >>

Client(self) == C1(self) \/ C2(self) \/ C3(self) \/ C4(self)

Next == (\E self \in ProcSet: deposit(self))
           \/ (\E self \in 1..N: Client(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Sane == total <= 7 * N * (N + 1) /\ biggest.who \in 0..N /\ Len(inbox) <= N
=============================================================================
