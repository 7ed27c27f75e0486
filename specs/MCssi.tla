-------------------------------- MODULE MCssi --------------------------------
(***************************************************************************)
(* Model wrapper for examples/serializableSnapshotIsolation.tla of         *)
(* tla-rust.  The reference only describes Toolbox clicks (lines 26-96);   *)
(* this is the equivalent MC module + cfg.  TLC needs single identifiers   *)
(* in the cfg, so the parameterised properties get names here.             *)
(***************************************************************************)
EXTENDS serializableSnapshotIsolation

WellFormed  == WellFormedTransactionsInHistory(history)
CahillOK    == CahillSerializable(history)
BernsteinOK == BernsteinSerializable(history)

\* "EXPECTED to be violated" (lines 81-96): reachability of interesting histories
NoTwoWaiters        == ~ AtLeastNTxnsAreWaitingForLocks(2)
NoVoluntaryAbort    == ~ AtLeastNTxnsAbortedDueToReason(1, "voluntary")
NoFCWAbort          == ~ AtLeastNTxnsAbortedDueToReason(1, "forced by First Committer Wins")
NoDeadlockAbort     == ~ AtLeastNTxnsAbortedDueToReason(1, "forced by deadlock-prevention")
NoCommitAbort       == ~ AtLeastNTxnsAbortedDueToReason(1, "in attempted commit, to preserve serializability")
NoReadAbort         == ~ AtLeastNTxnsAbortedDueToReason(1, "in attempted read, to preserve serializability")
NoWriteAbort        == ~ AtLeastNTxnsAbortedDueToReason(1, "in attempted write, to preserve serializability")
\* Key and TxnId are "symmetry sets" in the spec's run-book (serializableSnapshotIsolation.tla:38-44); cfg: SYMMETRY Perms
TxnPerms == Permutations(TxnId)
KeyPerms == Permutations(Key)
Perms    == Permutations(TxnId) \cup Permutations(Key)
=============================================================================
