---- MODULE even_odd ----
EXTENDS Naturals, TLC
CONSTANT N
(* MUTUAL recursion: even(n) calls odd(n - 1) calls even(n - 2) ...; each of the two procedures keeps a bounded call stack of its own
   (tla_rust_amd/csrc/pcal.cpp, call_recursive), and the return sites inside the other procedure's body say how the two interleave.
   Two processes ask about N and N + 1 at the same time. *)
(* --algorithm EvenOdd
variables res = [q \in 1..2 |-> 2];
procedure even(n)
begin
  E1: if n = 0 then
        res[self] := 1;
        return;
      end if;
  E2: call odd(n - 1);
  E3: return;
end procedure;
procedure odd(m)
begin
  O1: if m = 0 then
        res[self] := 0;
        return;
      end if;
  O2: call even(m - 1);
  O3: return;
end procedure;
process p \in 1..2
begin
  P1: call even(N + self - 1);
  P2: assert res[self] = (IF (N + self - 1) % 2 = 0 THEN 1 ELSE 0);
end process;
end algorithm *)
\* BEGIN TRANSLATION
CONSTANT defaultInitValue
VARIABLES res, pc, even_sp, even_ret1, even_ret2, even_ret3, even_ret4, n, n_stk1, n_stk2, n_stk3, n_stk4, odd_sp, odd_ret1, odd_ret2, odd_ret3, odd_ret4, m, m_stk1, m_stk2, m_stk3, m_stk4

vars == << res, pc, even_sp, even_ret1, even_ret2, even_ret3, even_ret4, n, n_stk1, n_stk2, n_stk3, n_stk4, odd_sp, odd_ret1, odd_ret2, odd_ret3, odd_ret4, m, m_stk1, m_stk2, m_stk3, m_stk4 >>

ProcSet == (1..2)

Init == (* Global variables *)
        /\ res = [q \in 1..2 |-> 2]
        (* Process p *)
        /\ even_sp = [self \in 1..2 |-> 0]
        /\ even_ret1 = [self \in 1..2 |-> 0]
        /\ even_ret2 = [self \in 1..2 |-> 0]
        /\ even_ret3 = [self \in 1..2 |-> 0]
        /\ even_ret4 = [self \in 1..2 |-> 0]
        /\ n = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk1 = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk2 = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk3 = [self \in 1..2 |-> defaultInitValue]
        /\ n_stk4 = [self \in 1..2 |-> defaultInitValue]
        /\ odd_sp = [self \in 1..2 |-> 0]
        /\ odd_ret1 = [self \in 1..2 |-> 0]
        /\ odd_ret2 = [self \in 1..2 |-> 0]
        /\ odd_ret3 = [self \in 1..2 |-> 0]
        /\ odd_ret4 = [self \in 1..2 |-> 0]
        /\ m = [self \in 1..2 |-> defaultInitValue]
        /\ m_stk1 = [self \in 1..2 |-> defaultInitValue]
        /\ m_stk2 = [self \in 1..2 |-> defaultInitValue]
        /\ m_stk3 = [self \in 1..2 |-> defaultInitValue]
        /\ m_stk4 = [self \in 1..2 |-> defaultInitValue]
        /\ pc = [self \in ProcSet |-> "P1"]

P1(self) == /\ pc[self] = "P1"
            /\ Assert(even_sp[self] < 4, 
                      "The call at line 29, column 7 needs more than the 4 stack frames this translation reserves for procedure even: raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm).")
            /\ even_ret1' = [even_ret1 EXCEPT ![self] = (IF even_sp[self] = 0 THEN 1 ELSE even_ret1[self])]
            /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF even_sp[self] = 0 THEN n[self] ELSE n_stk1[self])]
            /\ even_ret2' = [even_ret2 EXCEPT ![self] = (IF even_sp[self] = 1 THEN 1 ELSE even_ret2[self])]
            /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF even_sp[self] = 1 THEN n[self] ELSE n_stk2[self])]
            /\ even_ret3' = [even_ret3 EXCEPT ![self] = (IF even_sp[self] = 2 THEN 1 ELSE even_ret3[self])]
            /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF even_sp[self] = 2 THEN n[self] ELSE n_stk3[self])]
            /\ even_ret4' = [even_ret4 EXCEPT ![self] = (IF even_sp[self] = 3 THEN 1 ELSE even_ret4[self])]
            /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF even_sp[self] = 3 THEN n[self] ELSE n_stk4[self])]
            /\ n' = [n EXCEPT ![self] = N + self - 1]
            /\ even_sp' = [even_sp EXCEPT ![self] = even_sp[self] + 1]
            /\ pc' = [pc EXCEPT ![self] = "E1_p1"]
            /\ UNCHANGED << res, odd_sp, odd_ret1, odd_ret2, odd_ret3, 
                            odd_ret4, m, m_stk1, m_stk2, m_stk3, m_stk4 >>

P2(self) == /\ pc[self] = "P2"
            /\ Assert(res[self] = (IF (N + self - 1) % 2 = 0 THEN 1 ELSE 0), 
                      "Failure of assertion at line 30, column 7.")
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << res, even_sp, even_ret1, even_ret2, even_ret3, 
                            even_ret4, n, n_stk1, n_stk2, n_stk3, n_stk4, 
                            odd_sp, odd_ret1, odd_ret2, odd_ret3, odd_ret4, 
                            m, m_stk1, m_stk2, m_stk3, m_stk4 >>

O1_p2(self) == /\ pc[self] = "O1_p2"
               /\ IF m[self] = 0
                     THEN /\ res' = [res EXCEPT ![self] = 0]
                          /\ IF (IF odd_sp[self] = 1 THEN odd_ret1[self] ELSE (IF odd_sp[self] = 2 THEN odd_ret2[self] ELSE (IF odd_sp[self] = 3 THEN odd_ret3[self] ELSE odd_ret4[self]))) = 1
                                THEN /\ m' = [m EXCEPT ![self] = (IF odd_sp[self] = 1 THEN m_stk1[self] ELSE (IF odd_sp[self] = 2 THEN m_stk2[self] ELSE (IF odd_sp[self] = 3 THEN m_stk3[self] ELSE m_stk4[self])))]
                                     /\ odd_ret1' = [odd_ret1 EXCEPT ![self] = (IF odd_sp[self] = 1 THEN 0 ELSE odd_ret1[self])]
                                     /\ m_stk1' = [m_stk1 EXCEPT ![self] = (IF odd_sp[self] = 1 THEN defaultInitValue ELSE m_stk1[self])]
                                     /\ odd_ret2' = [odd_ret2 EXCEPT ![self] = (IF odd_sp[self] = 2 THEN 0 ELSE odd_ret2[self])]
                                     /\ m_stk2' = [m_stk2 EXCEPT ![self] = (IF odd_sp[self] = 2 THEN defaultInitValue ELSE m_stk2[self])]
                                     /\ odd_ret3' = [odd_ret3 EXCEPT ![self] = (IF odd_sp[self] = 3 THEN 0 ELSE odd_ret3[self])]
                                     /\ m_stk3' = [m_stk3 EXCEPT ![self] = (IF odd_sp[self] = 3 THEN defaultInitValue ELSE m_stk3[self])]
                                     /\ odd_ret4' = [odd_ret4 EXCEPT ![self] = (IF odd_sp[self] = 4 THEN 0 ELSE odd_ret4[self])]
                                     /\ m_stk4' = [m_stk4 EXCEPT ![self] = (IF odd_sp[self] = 4 THEN defaultInitValue ELSE m_stk4[self])]
                                     /\ odd_sp' = [odd_sp EXCEPT ![self] = odd_sp[self] - 1]
                                     /\ pc' = [pc EXCEPT ![self] = "E3_p1"]
                                ELSE /\ Assert(FALSE, 
                                               "Failure of assertion at line 22, column 9.")
                                     /\ pc' = [pc EXCEPT ![self] = "Done"]
                                     /\ UNCHANGED << odd_sp, odd_ret1, 
                                                     odd_ret2, odd_ret3, 
                                                     odd_ret4, m, m_stk1, 
                                                     m_stk2, m_stk3, 
                                                     m_stk4 >>
                     ELSE /\ pc' = [pc EXCEPT ![self] = "O2_p2"]
                          /\ UNCHANGED << res, odd_sp, odd_ret1, odd_ret2, 
                                          odd_ret3, odd_ret4, m, m_stk1, 
                                          m_stk2, m_stk3, m_stk4 >>
               /\ UNCHANGED << even_sp, even_ret1, even_ret2, even_ret3, 
                               even_ret4, n, n_stk1, n_stk2, n_stk3, 
                               n_stk4 >>

O2_p2(self) == /\ pc[self] = "O2_p2"
               /\ Assert(even_sp[self] < 4, 
                         "The call at line 24, column 7 needs more than the 4 stack frames this translation reserves for procedure even: raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm).")
               /\ even_ret1' = [even_ret1 EXCEPT ![self] = (IF even_sp[self] = 0 THEN 2 ELSE even_ret1[self])]
               /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF even_sp[self] = 0 THEN n[self] ELSE n_stk1[self])]
               /\ even_ret2' = [even_ret2 EXCEPT ![self] = (IF even_sp[self] = 1 THEN 2 ELSE even_ret2[self])]
               /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF even_sp[self] = 1 THEN n[self] ELSE n_stk2[self])]
               /\ even_ret3' = [even_ret3 EXCEPT ![self] = (IF even_sp[self] = 2 THEN 2 ELSE even_ret3[self])]
               /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF even_sp[self] = 2 THEN n[self] ELSE n_stk3[self])]
               /\ even_ret4' = [even_ret4 EXCEPT ![self] = (IF even_sp[self] = 3 THEN 2 ELSE even_ret4[self])]
               /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF even_sp[self] = 3 THEN n[self] ELSE n_stk4[self])]
               /\ n' = [n EXCEPT ![self] = m[self] - 1]
               /\ even_sp' = [even_sp EXCEPT ![self] = even_sp[self] + 1]
               /\ pc' = [pc EXCEPT ![self] = "E1_p1"]
               /\ UNCHANGED << res, odd_sp, odd_ret1, odd_ret2, odd_ret3, 
                               odd_ret4, m, m_stk1, m_stk2, m_stk3, m_stk4 >>

O3_p2(self) == /\ pc[self] = "O3_p2"
               /\ IF (IF odd_sp[self] = 1 THEN odd_ret1[self] ELSE (IF odd_sp[self] = 2 THEN odd_ret2[self] ELSE (IF odd_sp[self] = 3 THEN odd_ret3[self] ELSE odd_ret4[self]))) = 1
                     THEN /\ m' = [m EXCEPT ![self] = (IF odd_sp[self] = 1 THEN m_stk1[self] ELSE (IF odd_sp[self] = 2 THEN m_stk2[self] ELSE (IF odd_sp[self] = 3 THEN m_stk3[self] ELSE m_stk4[self])))]
                          /\ odd_ret1' = [odd_ret1 EXCEPT ![self] = (IF odd_sp[self] = 1 THEN 0 ELSE odd_ret1[self])]
                          /\ m_stk1' = [m_stk1 EXCEPT ![self] = (IF odd_sp[self] = 1 THEN defaultInitValue ELSE m_stk1[self])]
                          /\ odd_ret2' = [odd_ret2 EXCEPT ![self] = (IF odd_sp[self] = 2 THEN 0 ELSE odd_ret2[self])]
                          /\ m_stk2' = [m_stk2 EXCEPT ![self] = (IF odd_sp[self] = 2 THEN defaultInitValue ELSE m_stk2[self])]
                          /\ odd_ret3' = [odd_ret3 EXCEPT ![self] = (IF odd_sp[self] = 3 THEN 0 ELSE odd_ret3[self])]
                          /\ m_stk3' = [m_stk3 EXCEPT ![self] = (IF odd_sp[self] = 3 THEN defaultInitValue ELSE m_stk3[self])]
                          /\ odd_ret4' = [odd_ret4 EXCEPT ![self] = (IF odd_sp[self] = 4 THEN 0 ELSE odd_ret4[self])]
                          /\ m_stk4' = [m_stk4 EXCEPT ![self] = (IF odd_sp[self] = 4 THEN defaultInitValue ELSE m_stk4[self])]
                          /\ odd_sp' = [odd_sp EXCEPT ![self] = odd_sp[self] - 1]
                          /\ pc' = [pc EXCEPT ![self] = "E3_p1"]
                     ELSE /\ Assert(FALSE, 
                                    "Failure of assertion at line 25, column 7.")
                          /\ pc' = [pc EXCEPT ![self] = "Done"]
                          /\ UNCHANGED << odd_sp, odd_ret1, odd_ret2, 
                                          odd_ret3, odd_ret4, m, m_stk1, 
                                          m_stk2, m_stk3, m_stk4 >>
               /\ UNCHANGED << res, even_sp, even_ret1, even_ret2, even_ret3, 
                               even_ret4, n, n_stk1, n_stk2, n_stk3, 
                               n_stk4 >>

E1_p1(self) == /\ pc[self] = "E1_p1"
               /\ IF n[self] = 0
                     THEN /\ res' = [res EXCEPT ![self] = 1]
                          /\ IF (IF even_sp[self] = 1 THEN even_ret1[self] ELSE (IF even_sp[self] = 2 THEN even_ret2[self] ELSE (IF even_sp[self] = 3 THEN even_ret3[self] ELSE even_ret4[self]))) = 1
                                THEN /\ n' = [n EXCEPT ![self] = (IF even_sp[self] = 1 THEN n_stk1[self] ELSE (IF even_sp[self] = 2 THEN n_stk2[self] ELSE (IF even_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                                     /\ even_ret1' = [even_ret1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN 0 ELSE even_ret1[self])]
                                     /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                                     /\ even_ret2' = [even_ret2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN 0 ELSE even_ret2[self])]
                                     /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                                     /\ even_ret3' = [even_ret3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN 0 ELSE even_ret3[self])]
                                     /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                                     /\ even_ret4' = [even_ret4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN 0 ELSE even_ret4[self])]
                                     /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                                     /\ even_sp' = [even_sp EXCEPT ![self] = even_sp[self] - 1]
                                     /\ pc' = [pc EXCEPT ![self] = "P2"]
                                ELSE /\ IF (IF even_sp[self] = 1 THEN even_ret1[self] ELSE (IF even_sp[self] = 2 THEN even_ret2[self] ELSE (IF even_sp[self] = 3 THEN even_ret3[self] ELSE even_ret4[self]))) = 2
                                           THEN /\ n' = [n EXCEPT ![self] = (IF even_sp[self] = 1 THEN n_stk1[self] ELSE (IF even_sp[self] = 2 THEN n_stk2[self] ELSE (IF even_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                                                /\ even_ret1' = [even_ret1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN 0 ELSE even_ret1[self])]
                                                /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                                                /\ even_ret2' = [even_ret2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN 0 ELSE even_ret2[self])]
                                                /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                                                /\ even_ret3' = [even_ret3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN 0 ELSE even_ret3[self])]
                                                /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                                                /\ even_ret4' = [even_ret4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN 0 ELSE even_ret4[self])]
                                                /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                                                /\ even_sp' = [even_sp EXCEPT ![self] = even_sp[self] - 1]
                                                /\ pc' = [pc EXCEPT ![self] = "O3_p2"]
                                           ELSE /\ Assert(FALSE, 
                                                          "Failure of assertion at line 13, column 9.")
                                                /\ pc' = [pc EXCEPT ![self] = "Done"]
                                                /\ UNCHANGED << even_sp, 
                                                                even_ret1, 
                                                                even_ret2, 
                                                                even_ret3, 
                                                                even_ret4, n, 
                                                                n_stk1, 
                                                                n_stk2, 
                                                                n_stk3, 
                                                                n_stk4 >>
                     ELSE /\ pc' = [pc EXCEPT ![self] = "E2_p1"]
                          /\ UNCHANGED << res, even_sp, even_ret1, even_ret2, 
                                          even_ret3, even_ret4, n, n_stk1, 
                                          n_stk2, n_stk3, n_stk4 >>
               /\ UNCHANGED << odd_sp, odd_ret1, odd_ret2, odd_ret3, 
                               odd_ret4, m, m_stk1, m_stk2, m_stk3, m_stk4 >>

E2_p1(self) == /\ pc[self] = "E2_p1"
               /\ Assert(odd_sp[self] < 4, 
                         "The call at line 15, column 7 needs more than the 4 stack frames this translation reserves for procedure odd: raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm).")
               /\ odd_ret1' = [odd_ret1 EXCEPT ![self] = (IF odd_sp[self] = 0 THEN 1 ELSE odd_ret1[self])]
               /\ m_stk1' = [m_stk1 EXCEPT ![self] = (IF odd_sp[self] = 0 THEN m[self] ELSE m_stk1[self])]
               /\ odd_ret2' = [odd_ret2 EXCEPT ![self] = (IF odd_sp[self] = 1 THEN 1 ELSE odd_ret2[self])]
               /\ m_stk2' = [m_stk2 EXCEPT ![self] = (IF odd_sp[self] = 1 THEN m[self] ELSE m_stk2[self])]
               /\ odd_ret3' = [odd_ret3 EXCEPT ![self] = (IF odd_sp[self] = 2 THEN 1 ELSE odd_ret3[self])]
               /\ m_stk3' = [m_stk3 EXCEPT ![self] = (IF odd_sp[self] = 2 THEN m[self] ELSE m_stk3[self])]
               /\ odd_ret4' = [odd_ret4 EXCEPT ![self] = (IF odd_sp[self] = 3 THEN 1 ELSE odd_ret4[self])]
               /\ m_stk4' = [m_stk4 EXCEPT ![self] = (IF odd_sp[self] = 3 THEN m[self] ELSE m_stk4[self])]
               /\ m' = [m EXCEPT ![self] = n[self] - 1]
               /\ odd_sp' = [odd_sp EXCEPT ![self] = odd_sp[self] + 1]
               /\ pc' = [pc EXCEPT ![self] = "O1_p2"]
               /\ UNCHANGED << res, even_sp, even_ret1, even_ret2, even_ret3, 
                               even_ret4, n, n_stk1, n_stk2, n_stk3, 
                               n_stk4 >>

E3_p1(self) == /\ pc[self] = "E3_p1"
               /\ IF (IF even_sp[self] = 1 THEN even_ret1[self] ELSE (IF even_sp[self] = 2 THEN even_ret2[self] ELSE (IF even_sp[self] = 3 THEN even_ret3[self] ELSE even_ret4[self]))) = 1
                     THEN /\ n' = [n EXCEPT ![self] = (IF even_sp[self] = 1 THEN n_stk1[self] ELSE (IF even_sp[self] = 2 THEN n_stk2[self] ELSE (IF even_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                          /\ even_ret1' = [even_ret1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN 0 ELSE even_ret1[self])]
                          /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                          /\ even_ret2' = [even_ret2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN 0 ELSE even_ret2[self])]
                          /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                          /\ even_ret3' = [even_ret3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN 0 ELSE even_ret3[self])]
                          /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                          /\ even_ret4' = [even_ret4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN 0 ELSE even_ret4[self])]
                          /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                          /\ even_sp' = [even_sp EXCEPT ![self] = even_sp[self] - 1]
                          /\ pc' = [pc EXCEPT ![self] = "P2"]
                     ELSE /\ IF (IF even_sp[self] = 1 THEN even_ret1[self] ELSE (IF even_sp[self] = 2 THEN even_ret2[self] ELSE (IF even_sp[self] = 3 THEN even_ret3[self] ELSE even_ret4[self]))) = 2
                                THEN /\ n' = [n EXCEPT ![self] = (IF even_sp[self] = 1 THEN n_stk1[self] ELSE (IF even_sp[self] = 2 THEN n_stk2[self] ELSE (IF even_sp[self] = 3 THEN n_stk3[self] ELSE n_stk4[self])))]
                                     /\ even_ret1' = [even_ret1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN 0 ELSE even_ret1[self])]
                                     /\ n_stk1' = [n_stk1 EXCEPT ![self] = (IF even_sp[self] = 1 THEN defaultInitValue ELSE n_stk1[self])]
                                     /\ even_ret2' = [even_ret2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN 0 ELSE even_ret2[self])]
                                     /\ n_stk2' = [n_stk2 EXCEPT ![self] = (IF even_sp[self] = 2 THEN defaultInitValue ELSE n_stk2[self])]
                                     /\ even_ret3' = [even_ret3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN 0 ELSE even_ret3[self])]
                                     /\ n_stk3' = [n_stk3 EXCEPT ![self] = (IF even_sp[self] = 3 THEN defaultInitValue ELSE n_stk3[self])]
                                     /\ even_ret4' = [even_ret4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN 0 ELSE even_ret4[self])]
                                     /\ n_stk4' = [n_stk4 EXCEPT ![self] = (IF even_sp[self] = 4 THEN defaultInitValue ELSE n_stk4[self])]
                                     /\ even_sp' = [even_sp EXCEPT ![self] = even_sp[self] - 1]
                                     /\ pc' = [pc EXCEPT ![self] = "O3_p2"]
                                ELSE /\ Assert(FALSE, 
                                               "Failure of assertion at line 16, column 7.")
                                     /\ pc' = [pc EXCEPT ![self] = "Done"]
                                     /\ UNCHANGED << even_sp, even_ret1, 
                                                     even_ret2, even_ret3, 
                                                     even_ret4, n, n_stk1, 
                                                     n_stk2, n_stk3, 
                                                     n_stk4 >>
               /\ UNCHANGED << res, odd_sp, odd_ret1, odd_ret2, odd_ret3, 
                               odd_ret4, m, m_stk1, m_stk2, m_stk3, m_stk4 >>

p(self) == P1(self) \/ P2(self) \/ O1_p2(self) \/ O2_p2(self) \/ O3_p2(self) \/ E1_p1(self) \/ E2_p1(self) \/ E3_p1(self)

Next == (\E self \in 1..2: p(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
Answered == \A q \in 1..2 : res[q] \in {0, 1, 2}
====
