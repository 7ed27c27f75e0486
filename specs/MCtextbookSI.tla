----------------------------- MODULE MCtextbookSI -----------------------------
(***************************************************************************)
(* Model wrapper for examples/textbookSnapshotIsolation.tla of tla-rust    *)
(* (plain snapshot isolation: the same model as the SSI spec without       *)
(* Cahill's three variables).  Snapshot isolation alone is NOT             *)
(* serializable, so CahillOK / BernsteinOK are expected to be violated     *)
(* (write skew needs 3 transactions and 2 keys) and, as the spec's         *)
(* comments ask (lines 84-89), by the same histories.                      *)
(***************************************************************************)
EXTENDS textbookSnapshotIsolation

WellFormed  == WellFormedTransactionsInHistory(history)
CahillOK    == CahillSerializable(history)
BernsteinOK == BernsteinSerializable(history)
\* Key and TxnId are "symmetry sets" in the spec's run-book (serializableSnapshotIsolation.tla:38-44); cfg: SYMMETRY Perms
TxnPerms == Permutations(TxnId)
KeyPerms == Permutations(Key)
Perms    == Permutations(TxnId) \cup Permutations(Key)
=============================================================================
