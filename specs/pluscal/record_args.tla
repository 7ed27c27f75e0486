--------------------------- MODULE record_args ---------------------------
(***************************************************************************)
(* A procedure with a RECORD PARAMETER (`call deposit(mine)`, `call        *)
(* deposit([who |-> self, amount |-> 2])`) and `with old = biggest`, a     *)
(* record bound by `with`: the parameter becomes a record variable whose   *)
(* fields start as defaultInitValue (pcal2tla's `req = defaultInitValue`,  *)
(* field by field); `with v = r` binds v field by field.                   *)
(***************************************************************************)
EXTENDS Naturals, Sequences
CONSTANTS N
(* --algorithm record_args
variables inbox = <<>>, total = 0, biggest = [who |-> 0, amount |-> 0];

procedure deposit(req)
  variables fee = 1;
begin
  D1:
    total := total + req.amount - fee;
  D2:
    with old = biggest do
      if req.amount > old.amount then
        biggest := req;
      else
        biggest := [who |-> old.who, amount |-> old.amount];
      end if;
    end with;
    return;
end procedure;

process Client \in 1..N
  variables mine = [who |-> 0, amount |-> 0];
begin
  C1:
    mine := [who |-> self, amount |-> self * 5];
    inbox := Append(inbox, mine);
  C2:
    call deposit(mine);
  C3:
    call deposit([who |-> self, amount |-> 2]);
  C4:
    assert biggest.amount >= 2;
end process

end algorithm *)
\* BEGIN TRANSLATION
CONSTANT defaultInitValue
VARIABLES inbox_who, inbox_amount, total, biggest_who, biggest_amount, pc, mine_who, mine_amount, req_who, req_amount, fee

vars == << inbox_who, inbox_amount, total, biggest_who, biggest_amount, pc, mine_who, mine_amount, req_who, req_amount, fee >>

(* record variables are kept field by field: r.f is r_f *)
inbox == [n_ \in 1..Len(inbox_who) |-> [who |-> inbox_who[n_], amount |-> inbox_amount[n_]]]
biggest == [who |-> biggest_who, amount |-> biggest_amount]
mine == [self \in 1..N |-> [who |-> mine_who[self], amount |-> mine_amount[self]]]
req == [self \in 1..N |-> [who |-> req_who[self], amount |-> req_amount[self]]]

ProcSet == (1..N)

Init == (* Global variables *)
        /\ inbox_who = <<>>
        /\ inbox_amount = <<>>
        /\ total = 0
        /\ biggest_who = 0
        /\ biggest_amount = 0
        (* Process Client *)
        /\ mine_who = [self \in 1..N |-> 0]
        /\ mine_amount = [self \in 1..N |-> 0]
        /\ req_who = [self \in 1..N |-> defaultInitValue]
        /\ req_amount = [self \in 1..N |-> defaultInitValue]
        /\ fee = [self \in 1..N |-> 1]
        /\ pc = [self \in ProcSet |-> "C1"]

C1(self) == /\ pc[self] = "C1"
            /\ mine_who' = [mine_who EXCEPT ![self] = self]
            /\ mine_amount' = [mine_amount EXCEPT ![self] = self * 5]
            /\ inbox_who' = Append(inbox_who, mine_who'[self])
            /\ inbox_amount' = Append(inbox_amount, mine_amount'[self])
            /\ pc' = [pc EXCEPT ![self] = "C2"]
            /\ UNCHANGED << total, biggest_who, biggest_amount, req_who, 
                            req_amount, fee >>

C2(self) == /\ pc[self] = "C2"
            /\ req_who' = [req_who EXCEPT ![self] = mine_who[self]]
            /\ req_amount' = [req_amount EXCEPT ![self] = mine_amount[self]]
            /\ fee' = [fee EXCEPT ![self] = 1]
            /\ pc' = [pc EXCEPT ![self] = "D1_p1"]
            /\ UNCHANGED << inbox_who, inbox_amount, total, biggest_who, 
                            biggest_amount, mine_who, mine_amount >>

C3(self) == /\ pc[self] = "C3"
            /\ req_who' = [req_who EXCEPT ![self] = self]
            /\ req_amount' = [req_amount EXCEPT ![self] = 2]
            /\ fee' = [fee EXCEPT ![self] = 1]
            /\ pc' = [pc EXCEPT ![self] = "D1_p2"]
            /\ UNCHANGED << inbox_who, inbox_amount, total, biggest_who, 
                            biggest_amount, mine_who, mine_amount >>

C4(self) == /\ pc[self] = "C4"
            /\ Assert(biggest_amount >= 2, 
                      "Failure of assertion at line 41, column 5.")
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << inbox_who, inbox_amount, total, biggest_who, 
                            biggest_amount, mine_who, mine_amount, req_who, 
                            req_amount, fee >>

D1_p1(self) == /\ pc[self] = "D1_p1"
               /\ total' = total + req_amount[self] - fee[self]
               /\ pc' = [pc EXCEPT ![self] = "D2_p1"]
               /\ UNCHANGED << inbox_who, inbox_amount, biggest_who, 
                               biggest_amount, mine_who, mine_amount, 
                               req_who, req_amount, fee >>

D2_p1(self) == /\ pc[self] = "D2_p1"
               /\ LET old_who == biggest_who IN
                    /\ LET old_amount == biggest_amount IN
                         /\ IF req_amount[self] > old_amount
                               THEN /\ biggest_who' = req_who[self]
                                    /\ biggest_amount' = req_amount[self]
                               ELSE /\ biggest_who' = old_who
                                    /\ biggest_amount' = old_amount
               /\ req_who' = [req_who EXCEPT ![self] = defaultInitValue]
               /\ req_amount' = [req_amount EXCEPT ![self] = defaultInitValue]
               /\ fee' = [fee EXCEPT ![self] = 1]
               /\ pc' = [pc EXCEPT ![self] = "C3"]
               /\ UNCHANGED << inbox_who, inbox_amount, total, mine_who, 
                               mine_amount >>

D1_p2(self) == /\ pc[self] = "D1_p2"
               /\ total' = total + req_amount[self] - fee[self]
               /\ pc' = [pc EXCEPT ![self] = "D2_p2"]
               /\ UNCHANGED << inbox_who, inbox_amount, biggest_who, 
                               biggest_amount, mine_who, mine_amount, 
                               req_who, req_amount, fee >>

D2_p2(self) == /\ pc[self] = "D2_p2"
               /\ LET old_who == biggest_who IN
                    /\ LET old_amount == biggest_amount IN
                         /\ IF req_amount[self] > old_amount
                               THEN /\ biggest_who' = req_who[self]
                                    /\ biggest_amount' = req_amount[self]
                               ELSE /\ biggest_who' = old_who
                                    /\ biggest_amount' = old_amount
               /\ req_who' = [req_who EXCEPT ![self] = defaultInitValue]
               /\ req_amount' = [req_amount EXCEPT ![self] = defaultInitValue]
               /\ fee' = [fee EXCEPT ![self] = 1]
               /\ pc' = [pc EXCEPT ![self] = "C4"]
               /\ UNCHANGED << inbox_who, inbox_amount, total, mine_who, 
                               mine_amount >>

Client(self) == C1(self) \/ C2(self) \/ C3(self) \/ C4(self) \/ D1_p1(self) \/ D2_p1(self) \/ D1_p2(self) \/ D2_p2(self)

Next == (\E self \in 1..N: Client(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Sane == total <= 7 * N * (N + 1) /\ biggest.who \in 0..N /\ Len(inbox) <= N
=============================================================================
