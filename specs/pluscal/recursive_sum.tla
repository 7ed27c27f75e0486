---- MODULE recursive_sum ----
EXTENDS Naturals, Sequences, TLC
CONSTANT N
(* A RECURSIVE procedure (p-manual section 3.5): down(n) keeps n in a local, recurses, and adds the kept value on the way back, in two
   processes that interleave (the shared `turn` makes the interleavings distinguishable).  pcal2tla gives every process a `stack` of
   frames; here the procedure is compiled with ONE copy of its body per process and a bounded call stack kept as plain variables
   (tla_rust_amd/csrc/pcal.cpp, call_recursive: $TLAMC_PCAL_STACK levels, 4 by default — N = 3 needs exactly 4, N = 4 fails the
   assertion at the call that would need a fifth).  tests/golden/pcal_recursion/RecursiveSumStack.tla is the hand-written stack
   translation the state graph is compared with. *)
(* --algorithm RecursiveSum
variables acc = [q \in 1..2 |-> 0], turn = 0;
procedure down(n)
  variables kept = 0;
begin
  D1: if n = 0 then
        return;
      end if;
  D2: kept := n;
      turn := turn + 1;
  D3: call down(n - 1);
  D4: acc[self] := acc[self] + kept;
      return;
end procedure;
process p \in 1..2
begin
  P1: call down(N);
  P2: assert acc[self] * 2 = N * (N + 1);
end process;
end algorithm *)
\* BEGIN TRANSLATION
CONSTANT defaultInitValue
VARIABLES acc, turn, pc, down_sp, down_ret1, down_ret2, down_ret3, down_ret4, n, n_stk1, n_stk2, n_stk3, n_stk4, kept, kept_stk1, kept_stk2, kept_stk3, kept_stk4

vars == << acc, turn, pc, down_sp, down_ret1, down_ret2, down_ret3, down_ret4, n, n_stk1, n_stk2, n_stk3, n_stk4, kept, kept_stk1, kept_stk2, kept_stk3, kept_stk4 >>

ProcSet == (1..2)

Init == (* Global variables *)
        /\ acc = [q \in 1..2 |-> 0]
        /\ turn = 0
        (* Process p *)
        /\ down_sp = [self \in 1..2 |-> 0]
        /\ down_ret1 = [self \in 1..2 |-> 0]
        /\ down_ret2 = [self \in 1..2 |-> 0]
        /\ down_ret3 = [self \in 1..2 |-> 0]
        /\ down_ret4 = [self \in 1..2 |-> 0]
        /\ n = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk1 = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk2 = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk3 = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk4 = [self \in 1..2 |-> defaultInitValue]
        /\ kept = [self \in 1..2 |-> 0]
        /\ kept_stk1 = [self \in 1..2 |-> 0]
        /\ kept_stk2 = [self \in 1..2 |-> 0]
        /\ kept_stk3 = [self \in 1..2 |-> 0]
        /\ kept_stk4 = [self \in 1..2 |-> 0]
        /\ pc = [self \in ProcSet |-> "P1"]

P1(self) == /\ pc[self] = "P1"
            /\ Assert(down_sp[self] < 4, 
                      "The call at line 26, column 7 needs more than the 4 stack frames this translation reserves for procedure down: raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm).")
            /\ down_ret1' = [down_ret1 EXCEPT ![self] = (IF down_sp[self] = 0 THEN 1 ELSE down_ret1[self])]
            /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF down_sp[self] = 0 THEN n[self] ELSE n_stk1[self])]
            /\ kept_stk1' = [kept_stk1 EXCEPT ![self] = (IF down_sp[self] = 0 THEN kept[self] ELSE kept_stk1[self])]
            /\ down_ret2' = [down_ret2 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 1 ELSE down_ret2[self])]
            /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF down_sp[self] = 1 THEN n[self] ELSE n_stk2[self])]
            /\ kept_stk2' = [kept_stk2 EXCEPT ![self] = (IF down_sp[self] = 1 THEN kept[self] ELSE kept_stk2[self])]
            /\ down_ret3' = [down_ret3 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 1 ELSE down_ret3[self])]
            /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF down_sp[self] = 2 THEN n[self] ELSE n_stk3[self])]
            /\ kept_stk3' = [kept_stk3 EXCEPT ![self] = (IF down_sp[self] = 2 THEN kept[self] ELSE kept_stk3[self])]
            /\ down_ret4' = [down_ret4 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 1 ELSE down_ret4[self])]
            /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF down_sp[self] = 3 THEN n[self] ELSE n_stk4[self])]
            /\ kept_stk4' = [kept_stk4 EXCEPT ![self] = (IF down_sp[self] = 3 THEN kept[self] ELSE kept_stk4[self])]
            /\ n' = [n EXCEPT ![self] = N]
            /\ kept' = [kept EXCEPT ![self] = 0]
            /\ down_sp' = [down_sp EXCEPT ![self] = down_sp[self] + 1]
            /\ pc' = [pc EXCEPT ![self] = "D1_p1"]
            /\ UNCHANGED << acc, turn >>

P2(self) == /\ pc[self] = "P2"
            /\ Assert(acc[self] * 2 = N * (N + 1), 
                      "Failure of assertion at line 27, column 7.")
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << acc, turn, down_sp, down_ret1, down_ret2, 
                            down_ret3, down_ret4, n, n_stk1, n_stk2, n_stk3, 
                            n_stk4, kept, kept_stk1, kept_stk2, kept_stk3, 
                            kept_stk4 >>

D1_p1(self) == /\ pc[self] = "D1_p1"
               /\ IF n[self] = 0
                     THEN /\ IF (IF down_sp[self] = 1 THEN down_ret1[self] ELSE (IF down_sp[self] = 2 THEN down_ret2[self] ELSE (IF down_sp[self] = 3 THEN down_ret3[self] ELSE down_ret4[self]))) = 1
                                THEN /\ n' = [n EXCEPT ![self] = (IF down_sp[self] = 1 THEN n_stk1[self] ELSE (IF down_sp[self] = 2 THEN n_stk2[self] ELSE (IF down_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                                     /\ kept' = [kept EXCEPT ![self] = (IF down_sp[self] = 1 THEN kept_stk1[self] ELSE (IF down_sp[self] = 2 THEN kept_stk2[self] ELSE (IF down_sp[self] = 3 THEN kept_stk3[self] ELSE kept_stk4[self])))]
                                     /\ down_ret1' = [down_ret1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE down_ret1[self])]
                                     /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                                     /\ kept_stk1' = [kept_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE kept_stk1[self])]
                                     /\ down_ret2' = [down_ret2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE down_ret2[self])]
                                     /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                                     /\ kept_stk2' = [kept_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE kept_stk2[self])]
                                     /\ down_ret3' = [down_ret3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE down_ret3[self])]
                                     /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                                     /\ kept_stk3' = [kept_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE kept_stk3[self])]
                                     /\ down_ret4' = [down_ret4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE down_ret4[self])]
                                     /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                                     /\ kept_stk4' = [kept_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE kept_stk4[self])]
                                     /\ down_sp' = [down_sp EXCEPT ![self] = down_sp[self] - 1]
                                     /\ pc' = [pc EXCEPT ![self] = "P2"]
                                ELSE /\ IF (IF down_sp[self] = 1 THEN down_ret1[self] ELSE (IF down_sp[self] = 2 THEN down_ret2[self] ELSE (IF down_sp[self] = 3 THEN down_ret3[self] ELSE down_ret4[self]))) = 2
                                           THEN /\ n' = [n EXCEPT ![self] = (IF down_sp[self] = 1 THEN n_stk1[self] ELSE (IF down_sp[self] = 2 THEN n_stk2[self] ELSE (IF down_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                                                /\ kept' = [kept EXCEPT ![self] = (IF down_sp[self] = 1 THEN kept_stk1[self] ELSE (IF down_sp[self] = 2 THEN kept_stk2[self] ELSE (IF down_sp[self] = 3 THEN kept_stk3[self] ELSE kept_stk4[self])))]
                                                /\ down_ret1' = [down_ret1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE down_ret1[self])]
                                                /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                                                /\ kept_stk1' = [kept_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE kept_stk1[self])]
                                                /\ down_ret2' = [down_ret2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE down_ret2[self])]
                                                /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                                                /\ kept_stk2' = [kept_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE kept_stk2[self])]
                                                /\ down_ret3' = [down_ret3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE down_ret3[self])]
                                                /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                                                /\ kept_stk3' = [kept_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE kept_stk3[self])]
                                                /\ down_ret4' = [down_ret4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE down_ret4[self])]
                                                /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                                                /\ kept_stk4' = [kept_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE kept_stk4[self])]
                                                /\ down_sp' = [down_sp EXCEPT ![self] = down_sp[self] - 1]
                                                /\ pc' = [pc EXCEPT ![self] = "D4_p1"]
                                           ELSE /\ Assert(FALSE, 
                                                          "Failure of assertion at line 16, column 9.")
                                                /\ pc' = [pc EXCEPT ![self] = "Done"]
                                                /\ UNCHANGED << down_sp, 
                                                                down_ret1, 
                                                                down_ret2, 
                                                                down_ret3, 
                                                                down_ret4, n, 
                                                                n_stk1, 
                                                                n_stk2, 
                                                                n_stk3, 
                                                                n_stk4, kept, 
                                                                kept_stk1, 
                                                                kept_stk2, 
                                                                kept_stk3, 
                                                                kept_stk4 >>
                     ELSE /\ pc' = [pc EXCEPT ![self] = "D2_p1"]
                          /\ UNCHANGED << down_sp, down_ret1, down_ret2, 
                                          down_ret3, down_ret4, n, n_stk1, 
                                          n_stk2, n_stk3, n_stk4, kept, 
                                          kept_stk1, kept_stk2, kept_stk3, 
                                          kept_stk4 >>
               /\ UNCHANGED << acc, turn >>

D2_p1(self) == /\ pc[self] = "D2_p1"
               /\ kept' = [kept EXCEPT ![self] = n[self]]
               /\ turn' = turn + 1
               /\ pc' = [pc EXCEPT ![self] = "D3_p1"]
               /\ UNCHANGED << acc, down_sp, down_ret1, down_ret2, down_ret3, 
                               down_ret4, n, n_stk1, n_stk2, n_stk3, n_stk4, 
                               kept_stk1, kept_stk2, kept_stk3, kept_stk4 >>

D3_p1(self) == /\ pc[self] = "D3_p1"
               /\ Assert(down_sp[self] < 4, 
                         "The call at line 20, column 7 needs more than the 4 stack frames this translation reserves for procedure down: raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm).")
               /\ down_ret1' = [down_ret1 EXCEPT ![self] = (IF down_sp[self] = 0 THEN 2 ELSE down_ret1[self])]
               /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF down_sp[self] = 0 THEN n[self] ELSE n_stk1[self])]
               /\ kept_stk1' = [kept_stk1 EXCEPT ![self] = (IF down_sp[self] = 0 THEN kept[self] ELSE kept_stk1[self])]
               /\ down_ret2' = [down_ret2 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 2 ELSE down_ret2[self])]
               /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF down_sp[self] = 1 THEN n[self] ELSE n_stk2[self])]
               /\ kept_stk2' = [kept_stk2 EXCEPT ![self] = (IF down_sp[self] = 1 THEN kept[self] ELSE kept_stk2[self])]
               /\ down_ret3' = [down_ret3 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 2 ELSE down_ret3[self])]
               /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF down_sp[self] = 2 THEN n[self] ELSE n_stk3[self])]
               /\ kept_stk3' = [kept_stk3 EXCEPT ![self] = (IF down_sp[self] = 2 THEN kept[self] ELSE kept_stk3[self])]
               /\ down_ret4' = [down_ret4 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 2 ELSE down_ret4[self])]
               /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF down_sp[self] = 3 THEN n[self] ELSE n_stk4[self])]
               /\ kept_stk4' = [kept_stk4 EXCEPT ![self] = (IF down_sp[self] = 3 THEN kept[self] ELSE kept_stk4[self])]
               /\ n' = [n EXCEPT ![self] = n[self] - 1]
               /\ kept' = [kept EXCEPT ![self] = 0]
               /\ down_sp' = [down_sp EXCEPT ![self] = down_sp[self] + 1]
               /\ pc' = [pc EXCEPT ![self] = "D1_p1"]
               /\ UNCHANGED << acc, turn >>

D4_p1(self) == /\ pc[self] = "D4_p1"
               /\ acc' = [acc EXCEPT ![self] = acc[self] + kept[self]]
               /\ IF (IF down_sp[self] = 1 THEN down_ret1[self] ELSE (IF down_sp[self] = 2 THEN down_ret2[self] ELSE (IF down_sp[self] = 3 THEN down_ret3[self] ELSE down_ret4[self]))) = 1
                     THEN /\ n' = [n EXCEPT ![self] = (IF down_sp[self] = 1 THEN n_stk1[self] ELSE (IF down_sp[self] = 2 THEN n_stk2[self] ELSE (IF down_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                          /\ kept' = [kept EXCEPT ![self] = (IF down_sp[self] = 1 THEN kept_stk1[self] ELSE (IF down_sp[self] = 2 THEN kept_stk2[self] ELSE (IF down_sp[self] = 3 THEN kept_stk3[self] ELSE kept_stk4[self])))]
                          /\ down_ret1' = [down_ret1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE down_ret1[self])]
                          /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                          /\ kept_stk1' = [kept_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE kept_stk1[self])]
                          /\ down_ret2' = [down_ret2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE down_ret2[self])]
                          /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                          /\ kept_stk2' = [kept_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE kept_stk2[self])]
                          /\ down_ret3' = [down_ret3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE down_ret3[self])]
                          /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                          /\ kept_stk3' = [kept_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE kept_stk3[self])]
                          /\ down_ret4' = [down_ret4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE down_ret4[self])]
                          /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                          /\ kept_stk4' = [kept_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE kept_stk4[self])]
                          /\ down_sp' = [down_sp EXCEPT ![self] = down_sp[self] - 1]
                          /\ pc' = [pc EXCEPT ![self] = "P2"]
                     ELSE /\ IF (IF down_sp[self] = 1 THEN down_ret1[self] ELSE (IF down_sp[self] = 2 THEN down_ret2[self] ELSE (IF down_sp[self] = 3 THEN down_ret3[self] ELSE down_ret4[self]))) = 2
                                THEN /\ n' = [n EXCEPT ![self] = (IF down_sp[self] = 1 THEN n_stk1[self] ELSE (IF down_sp[self] = 2 THEN n_stk2[self] ELSE (IF down_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                                     /\ kept' = [kept EXCEPT ![self] = (IF down_sp[self] = 1 THEN kept_stk1[self] ELSE (IF down_sp[self] = 2 THEN kept_stk2[self] ELSE (IF down_sp[self] = 3 THEN kept_stk3[self] ELSE kept_stk4[self])))]
                                     /\ down_ret1' = [down_ret1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE down_ret1[self])]
                                     /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                                     /\ kept_stk1' = [kept_stk1 EXCEPT ![self] = (IF down_sp[self] = 1 THEN 0 ELSE kept_stk1[self])]
                                     /\ down_ret2' = [down_ret2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE down_ret2[self])]
                                     /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                                     /\ kept_stk2' = [kept_stk2 EXCEPT ![self] = (IF down_sp[self] = 2 THEN 0 ELSE kept_stk2[self])]
                                     /\ down_ret3' = [down_ret3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE down_ret3[self])]
                                     /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                                     /\ kept_stk3' = [kept_stk3 EXCEPT ![self] = (IF down_sp[self] = 3 THEN 0 ELSE kept_stk3[self])]
                                     /\ down_ret4' = [down_ret4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE down_ret4[self])]
                                     /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                                     /\ kept_stk4' = [kept_stk4 EXCEPT ![self] = (IF down_sp[self] = 4 THEN 0 ELSE kept_stk4[self])]
                                     /\ down_sp' = [down_sp EXCEPT ![self] = down_sp[self] - 1]
                                     /\ pc' = [pc EXCEPT ![self] = "D4_p1"]
                                ELSE /\ Assert(FALSE, 
                                               "Failure of assertion at line 22, column 7.")
                                     /\ pc' = [pc EXCEPT ![self] = "Done"]
                                     /\ UNCHANGED << down_sp, down_ret1, 
                                                     down_ret2, down_ret3, 
                                                     down_ret4, n, n_stk1, 
                                                     n_stk2, n_stk3, n_stk4, 
                                                     kept, kept_stk1, 
                                                     kept_stk2, kept_stk3, 
                                                     kept_stk4 >>
               /\ UNCHANGED turn

p(self) == P1(self) \/ P2(self) \/ D1_p1(self) \/ D2_p1(self) \/ D3_p1(self) \/ D4_p1(self)

Next == (\E self \in 1..2: p(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
Bounded == turn <= 2 * N
====
