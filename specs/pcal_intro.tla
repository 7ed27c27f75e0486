------------------------------- MODULE pcal_intro -------------------------------
(***************************************************************************)
(* The two-process money-transfer example of the tla-rust README, in the   *)
(* committed (atomic Transfer) form, followed by the TLA+ translation that *)
(* `pcal2tla` (p-manual.pdf §3.8, App. B) would insert.  The translation   *)
(* below was written by hand for this repository; the device lowering in   *)
(* tla_rust_amd/csrc/spec_pluscal.h follows it action by action.           *)
(***************************************************************************)
EXTENDS Naturals, TLC

(* --algorithm transfer
variables alice_account = 10, bob_account = 10,
          account_total = alice_account + bob_account

process TransProc \in 1..2
  variables money \in 1..20;
begin
Transfer:
  if alice_account >= money then
     alice_account := alice_account - money;
     bob_account := bob_account + money;
  end if;
C: assert alice_account >= 0;
end process

end algorithm *)

\* BEGIN TRANSLATION
VARIABLES alice_account, bob_account, account_total, pc, money

vars == << alice_account, bob_account, account_total, pc, money >>

ProcSet == (1..2)

Init == (* Global variables *)
        /\ alice_account = 10
        /\ bob_account = 10
        /\ account_total = alice_account + bob_account
        (* Process TransProc *)
        /\ money \in [1..2 -> 1..20]
        /\ pc = [self \in ProcSet |-> "Transfer"]

Transfer(self) == /\ pc[self] = "Transfer"
                  /\ IF alice_account >= money[self]
                        THEN /\ alice_account' = alice_account - money[self]
                             /\ bob_account' = bob_account + money[self]
                        ELSE /\ TRUE
                             /\ UNCHANGED << alice_account, bob_account >>
                  /\ pc' = [pc EXCEPT ![self] = "C"]
                  /\ UNCHANGED << account_total, money >>

C(self) == /\ pc[self] = "C"
           /\ Assert(alice_account >= 0, 
                     "Failure of assertion at line 23, column 4.")
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << alice_account, bob_account, account_total, 
                           money >>

TransProc(self) == Transfer(self) \/ C(self)

Next == (\E self \in 1..2: TransProc(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

MoneyInvariant == alice_account + bob_account = account_total
=============================================================================
