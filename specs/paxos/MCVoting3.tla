------------------------------ MODULE MCVoting3 ------------------------------
(***************************************************************************)
(* Model of examples/Paxos/Voting.tla: three acceptors, two values, the     *)
(* majority quorums, ballots 0..2 (one ballot more than the reference's     *)
(* own MCVoting.tla:9).  Voting.tla and Consensus.tla are read from the     *)
(* reference tree (mc -I / $TLA_PATH).  Run with -deadlock: a finite       *)
(* Ballot set ends in states without successors.                           *)
(***************************************************************************)
EXTENDS Voting, TLC

CONSTANTS a1, a2, a3
CONSTANTS v1, v2

MCAcceptor == {a1, a2, a3}
MCValue == {v1, v2}
MCQuorum == {{a1, a2}, {a1, a3}, {a2, a3}}
MCBallot == 0..2
MCSymmetry == Permutations(MCAcceptor) \cup Permutations(MCValue)

ConsensusSpecBar == C!Spec
=============================================================================
