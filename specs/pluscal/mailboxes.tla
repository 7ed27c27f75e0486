----------------------------- MODULE mailboxes -----------------------------
(***************************************************************************)
(* N nodes in a ring, one mailbox each.  box is an ARRAY of SEQUENCES of   *)
(* RECORDS [kind, from, val]; heard an array of sequences of numbers; log  *)
(* a sequence of records with an initial element; one statement appends to *)
(* two sequences at once (`||`).  Every node pings its right neighbour,    *)
(* answers the ping it receives with a pong and adds up the pong it gets.  *)
(***************************************************************************)
EXTENDS Naturals, Sequences
CONSTANTS N

(* --algorithm mailboxes
variables box = [p \in 1..N |-> <<>>],
          heard = [p \in 1..N |-> <<>>],
          log = << [kind |-> "start", from |-> 0, val |-> 0] >>,
          sum = 0;

process Node \in 1..N
  variables m = [kind |-> "none", from |-> 0, val |-> 0];
begin
  S:
    box[(self % N) + 1] := Append(box[(self % N) + 1], [kind |-> "ping", from |-> self, val |-> self * 10]);
  R:
    await box[self] # <<>>;
    m := Head(box[self]);
    box[self] := Tail(box[self]);
  A:
    if m.kind = "ping" then
      box[m.from] := box[m.from] \o << [kind |-> "pong", from |-> self, val |-> m.val + 1] >>;
      log := Append(log, m) || heard[self] := Append(heard[self], m.from);
      goto R;
    else
      sum := sum + m.val;
      log[1] := m;
    end if;
  F:
    assert Len(box[self]) = 0 \/ Head(box[self]).kind = "ping";
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES box_kind, box_from, box_val, heard, log_kind, log_from, log_val, sum, pc, m_kind, m_from, m_val

vars == << box_kind, box_from, box_val, heard, log_kind, log_from, log_val, sum, pc, m_kind, m_from, m_val >>

(* record variables are kept field by field: r.f is r_f *)
box == [p \in 1..N |-> [n_ \in 1..Len(box_kind[p]) |-> [kind |-> box_kind[p][n_], from |-> box_from[p][n_], val |-> box_val[p][n_]]]]
log == [n_ \in 1..Len(log_kind) |-> [kind |-> log_kind[n_], from |-> log_from[n_], val |-> log_val[n_]]]
m == [self \in 1..N |-> [kind |-> m_kind[self], from |-> m_from[self], val |-> m_val[self]]]

ProcSet == (1..N)

Init == (* Global variables *)
        /\ box_kind = [p \in 1..N |-> <<>>]
        /\ box_from = [p \in 1..N |-> <<>>]
        /\ box_val = [p \in 1..N |-> <<>>]
        /\ heard = [p \in 1..N |-> <<>>]
        /\ log_kind = <<"start">>
        /\ log_from = <<0>>
        /\ log_val = <<0>>
        /\ sum = 0
        (* Process Node *)
        /\ m_kind = [self \in 1..N |-> "none"]
        /\ m_from = [self \in 1..N |-> 0]
        /\ m_val = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "S"]

S(self) == /\ pc[self] = "S"
           /\ box_kind' = [box_kind EXCEPT ![(self % N) + 1] = Append(box_kind[(self % N) + 1], "ping")]
           /\ box_from' = [box_from EXCEPT ![(self % N) + 1] = Append(box_from[(self % N) + 1], self)]
           /\ box_val' = [box_val EXCEPT ![(self % N) + 1] = Append(box_val[(self % N) + 1], self * 10)]
           /\ pc' = [pc EXCEPT ![self] = "R"]
           /\ UNCHANGED << heard, log_kind, log_from, log_val, sum, m_kind, 
                           m_from, m_val >>

R(self) == /\ pc[self] = "R"
           /\ box_kind[self] # <<>>
           /\ m_kind' = [m_kind EXCEPT ![self] = Head(box_kind[self])]
           /\ m_from' = [m_from EXCEPT ![self] = Head(box_from[self])]
           /\ m_val' = [m_val EXCEPT ![self] = Head(box_val[self])]
           /\ box_kind' = [box_kind EXCEPT ![self] = Tail(box_kind[self])]
           /\ box_from' = [box_from EXCEPT ![self] = Tail(box_from[self])]
           /\ box_val' = [box_val EXCEPT ![self] = Tail(box_val[self])]
           /\ pc' = [pc EXCEPT ![self] = "A"]
           /\ UNCHANGED << heard, log_kind, log_from, log_val, sum >>

A(self) == /\ pc[self] = "A"
           /\ IF m_kind[self] = "ping"
                 THEN /\ box_kind' = [box_kind EXCEPT ![m_from[self]] = box_kind[m_from[self]] \o <<"pong">>]
                      /\ box_from' = [box_from EXCEPT ![m_from[self]] = box_from[m_from[self]] \o <<self>>]
                      /\ box_val' = [box_val EXCEPT ![m_from[self]] = box_val[m_from[self]] \o <<m_val[self] + 1>>]
                      /\ log_kind' = Append(log_kind, m_kind[self])
                      /\ log_from' = Append(log_from, m_from[self])
                      /\ log_val' = Append(log_val, m_val[self])
                      /\ heard' = [heard EXCEPT ![self] = Append(heard[self], m_from[self])]
                      /\ pc' = [pc EXCEPT ![self] = "R"]
                      /\ UNCHANGED sum
                 ELSE /\ sum' = sum + m_val[self]
                      /\ log_kind' = [log_kind EXCEPT ![1] = m_kind[self]]
                      /\ log_from' = [log_from EXCEPT ![1] = m_from[self]]
                      /\ log_val' = [log_val EXCEPT ![1] = m_val[self]]
                      /\ pc' = [pc EXCEPT ![self] = "F"]
                      /\ UNCHANGED << box_kind, box_from, box_val, heard >>
           /\ UNCHANGED << m_kind, m_from, m_val >>

F(self) == /\ pc[self] = "F"
           /\ Assert(Len(box_kind[self]) = 0 \/ Head(box_kind[self]) = "ping", 
                     "Failure of assertion at line 37, column 5.")
           /\ pc' = [pc EXCEPT ![self] = "Done"]
           /\ UNCHANGED << box_kind, box_from, box_val, heard, log_kind, 
                           log_from, log_val, sum, m_kind, m_from, m_val >>

Node(self) == S(self) \/ R(self) \/ A(self) \/ F(self)

Next == (\E self \in 1..N: Node(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

LogOk == \A k \in 2..Len(log) : log[k].kind = "ping" /\ log[k].val = log[k].from * 10
Pongs == \A p \in 1..N : \A k \in 1..Len(box[p]) : box[p][k].kind = "pong" => box[p][k].val % 10 = 1
HeardTheLeft == \A p \in 1..N : Len(heard[p]) <= 1 /\ (heard[p] # <<>> => heard[p][1] = ((p + N - 2) % N) + 1)
=============================================================================
