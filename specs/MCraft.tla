------------------------------- MODULE MCraft -------------------------------
(***************************************************************************)
(* Model wrapper for examples/raft.tla of tla-rust (which has no .cfg and  *)
(* an unbounded term counter, raft.tla:199): bounds the state space with a *)
(* CONSTRAINT and names the two properties the modified spec carries       *)
(* (raft.tla:500-507 and raft.tla:74,302) as invariants.                   *)
(* Pattern: SpecifyingSystems/TLC/MCAlternatingBit.tla + .cfg.             *)
(***************************************************************************)
EXTENDS raft
CONSTANTS MaxTerm, MaxLogLen, MaxMsgs, MaxMsgKeys   \* used only by the constraint

InFlight ==                                   \* copies of messages currently deliverable
  LET RECURSIVE Sum(_)
      Sum(T) == IF T = {} THEN 0
                ELSE LET m == CHOOSE x \in T : TRUE IN messages[m] + Sum(T \ {m})
  IN  Sum(ValidMessage(messages))

StateConstraint == /\ \A i \in Server : currentTerm[i] <= MaxTerm
                   /\ \A i \in Server : Len(log[i]) <= MaxLogLen
                   /\ InFlight <= MaxMsgs
                   \* the bag's key set only grows (raft.tla:117-129 keep a zero-count key): bound it too
                   /\ Cardinality(DOMAIN messages) <= MaxMsgKeys

NoTwoLeaders       == ~MoreThanOneLeader
CommittedLogStable == ~committedLogDecrease
=============================================================================
