------------------------------ MODULE MCPaxos3 ------------------------------
(***************************************************************************)
(* Model of examples/Paxos/Paxos.tla with the sizes the reference's own     *)
(* MCPaxos.tla:7-9 carries in its comments (three acceptors, two values,   *)
(* majority quorums); same shape as the reference's wrapper, which is      *)
(* committed with the one-acceptor sizes.  Paxos.tla and Voting.tla are    *)
(* read from the reference tree (mc -I / $TLA_PATH).                       *)
(***************************************************************************)
EXTENDS Paxos, TLC

CONSTANTS a1, a2, a3
CONSTANTS v1, v2

MCAcceptor == {a1, a2, a3}
MCValue == {v1, v2}
MCQuorum == {{a1, a2}, {a1, a3}, {a2, a3}}
MCMaxBallot == 1
MCBallot == 0..MCMaxBallot
MCSymmetry == Permutations(MCAcceptor) \cup Permutations(MCValue)

VotingSpecBar == V!Spec

Inv1 == Inv!1
Inv2 == Inv!2
Inv3 == Inv!3
Inv4 == Inv!4
=============================================================================
