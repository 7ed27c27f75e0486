-------------------------- MODULE MCVotingBadQuorum --------------------------
(***************************************************************************)
(* NEGATIVE CONTROL for the PROPERTY check: examples/Paxos/Voting.tla under *)
(* quorums that violate QuorumAssumption (Voting.tla:16-17: any two quorums *)
(* intersect).  With singleton quorums two acceptors choose two values:    *)
(* C!Spec (MCVoting.tla:26) fails on the step that makes `chosen` a        *)
(* two-element set, while Inv still holds.                                 *)
(***************************************************************************)
EXTENDS Voting, TLC

CONSTANTS a1, a2, a3
CONSTANTS v1, v2

MCAcceptor == {a1, a2, a3}
MCValue == {v1, v2}
MCQuorum == {{a1}, {a2}, {a3}}
MCBallot == 0..1

ConsensusSpecBar == C!Spec
=============================================================================
