------------------------------ MODULE MCPaxosBad ------------------------------
(***************************************************************************)
(* NEGATIVE CONTROL for the invariants: examples/Paxos/Paxos.tla with a    *)
(* Phase2a that has lost its quorum conjunct (Paxos.tla:138-149): a leader  *)
(* proposes without having heard from anybody, so a "2a" message exists    *)
(* for which no quorum shows the value safe — Inv!3 (Paxos.tla:196-206,    *)
(* MCPaxos.tla:67 Inv3) fails one step after Init.                         *)
(***************************************************************************)
EXTENDS Paxos, TLC

CONSTANTS a1, a2, a3
CONSTANTS v1, v2

MCAcceptor == {a1, a2, a3}
MCValue == {v1, v2}
MCQuorum == {{a1, a2}, {a1, a3}, {a2, a3}}
MCBallot == 0..1

BadPhase2a(b, v) ==
  /\ ~ \E m \in msgs : m.type = "2a" /\ m.bal = b
  /\ Send([type |-> "2a", bal |-> b, val |-> v])
  /\ UNCHANGED <<maxBal, maxVBal, maxVal>>

BadNext == \/ \E b \in Ballot : \/ Phase1a(b)
                                \/ \E v \in Value : BadPhase2a(b, v)
           \/ \E a \in Acceptor : Phase1b(a) \/ Phase2b(a)

BadSpec == Init /\ [][BadNext]_vars

VotingSpecBar == V!Spec
Inv1 == Inv!1
Inv2 == Inv!2
Inv3 == Inv!3
Inv4 == Inv!4
=============================================================================
