----------------------------- MODULE paxos_soup -----------------------------
(***************************************************************************)
(* Single-decree Paxos in PlusCal over a message soup (a SET of RECORDS),  *)
(* the consensus item of the reference's roadmap (README.md:26-42) in the  *)
(* modelling style of its own examples/Paxos/Paxos.tla (`msgs`, phases 1a  *)
(* 1b 2a 2b), with one proposer process per ballot and NA acceptors.  A    *)
(* proposer sends 1a, collects 1b promises one message at a time until it  *)
(* has a majority, proposes the value of the highest-numbered vote it was  *)
(* told of (any value if none) and sends 2a; an acceptor answers any 1a /  *)
(* 2a it may.  Forgetful = TRUE: the proposer ignores what it was told —   *)
(* two values can be chosen.  Acceptors never stop: check with deadlock    *)
(* detection off (paxos_soup.cfg says CHECK_DEADLOCK FALSE).               *)
(***************************************************************************)
EXTENDS Naturals, FiniteSets
CONSTANTS NA, NB, NV, Forgetful

(* --algorithm paxos_soup
variables msgs = {},
          maxBal = [a \in 1..NA |-> 0],
          maxVBal = [a \in 1..NA |-> 0],
          maxVal = [a \in 1..NA |-> 0];

define
  Voted(a, b, v) == [type |-> "2b", bal |-> b, acc |-> a, vbal |-> 0, val |-> v] \in msgs
  ChosenAt(b, v) == 2 * Cardinality({a \in 1..NA : Voted(a, b, v)}) > NA
end define;

process Proposer \in (NA + 1)..(NA + NB)
  variables promises = {}, hb = 0, hv = 0, v = 0;
begin
  P1:
    msgs := msgs \cup {[type |-> "1a", bal |-> self - NA, acc |-> 0, vbal |-> 0, val |-> 0]};
  P2:
    while 2 * Cardinality(promises) <= NA do
      with m \in msgs do
        await m.type = "1b" /\ m.bal = self - NA /\ m.acc \notin promises;
        promises := promises \cup {m.acc};
        if m.vbal > hb then
          hb := m.vbal;
          hv := m.val;
        end if;
      end with;
    end while;
  P3:
    if hb = 0 \/ Forgetful then
      with w \in 1..NV do
        v := w;
      end with;
    else
      v := hv;
    end if;
  P4:
    msgs := msgs \cup {[type |-> "2a", bal |-> self - NA, acc |-> 0, vbal |-> 0, val |-> v]};
end process

process Acceptor \in 1..NA
begin
  A:
    while TRUE do
      with m \in msgs do
        either
          await m.type = "1a" /\ m.bal > maxBal[self];
          maxBal[self] := m.bal;
          msgs := msgs \cup {[type |-> "1b", bal |-> m.bal, acc |-> self, vbal |-> maxVBal[self], val |-> maxVal[self]]};
        or
          await m.type = "2a" /\ m.bal >= maxBal[self];
          maxBal[self] := m.bal;
          maxVBal[self] := m.bal;
          maxVal[self] := m.val;
          msgs := msgs \cup {[type |-> "2b", bal |-> m.bal, acc |-> self, vbal |-> 0, val |-> m.val]};
        end either;
      end with;
    end while;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES msgs, maxBal, maxVBal, maxVal, pc

(* define statement *)
Voted(a, b, v) == [type |-> "2b", bal |-> b, acc |-> a, vbal |-> 0, val |-> v] \in msgs

ChosenAt(b, v) == 2 * Cardinality({a \in 1..NA : Voted(a, b, v)}) > NA

VARIABLES promises, hb, hv, v

vars == << msgs, maxBal, maxVBal, maxVal, pc, promises, hb, hv, v >>

ProcSet == ((NA + 1)..(NA + NB)) \cup (1..NA)

Init == (* Global variables *)
        /\ msgs = {}
        /\ maxBal = [a \in 1..NA |-> 0]
        /\ maxVBal = [a \in 1..NA |-> 0]
        /\ maxVal = [a \in 1..NA |-> 0]
        (* Process Proposer *)
        /\ promises = [self \in (NA + 1)..(NA + NB) |-> {}]
        /\ hb = [self \in (NA + 1)..(NA + NB) |-> 0]
        /\ hv = [self \in (NA + 1)..(NA + NB) |-> 0]
        /\ v = [self \in (NA + 1)..(NA + NB) |-> 0]
        /\ pc = [self \in ProcSet |-> CASE self \in (NA + 1)..(NA + NB) -> "P1"
                                        [] self \in 1..NA -> "A"]

P1(self) == /\ pc[self] = "P1"
            /\ msgs' = msgs \cup {[type |-> "1a", bal |-> self - NA, acc |-> 0, vbal |-> 0, val |-> 0]}
            /\ pc' = [pc EXCEPT ![self] = "P2"]
            /\ UNCHANGED << maxBal, maxVBal, maxVal, promises, hb, hv, v >>

P2(self) == /\ pc[self] = "P2"
            /\ IF 2 * Cardinality(promises[self]) <= NA
                  THEN /\ \E m \in msgs:
                            /\ m.type = "1b" /\ m.bal = self - NA /\ m.acc \notin promises[self]
                            /\ promises' = [promises EXCEPT ![self] = promises[self] \cup {m.acc}]
                            /\ IF m.vbal > hb[self]
                                  THEN /\ hb' = [hb EXCEPT ![self] = m.vbal]
                                       /\ hv' = [hv EXCEPT ![self] = m.val]
                                  ELSE /\ TRUE
                                       /\ UNCHANGED << hb, hv >>
                       /\ pc' = [pc EXCEPT ![self] = "P2"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "P3"]
                       /\ UNCHANGED << promises, hb, hv >>
            /\ UNCHANGED << msgs, maxBal, maxVBal, maxVal, v >>

P3(self) == /\ pc[self] = "P3"
            /\ IF hb[self] = 0 \/ Forgetful
                  THEN /\ \E w \in 1..NV:
                            /\ v' = [v EXCEPT ![self] = w]
                  ELSE /\ v' = [v EXCEPT ![self] = hv[self]]
            /\ pc' = [pc EXCEPT ![self] = "P4"]
            /\ UNCHANGED << msgs, maxBal, maxVBal, maxVal, promises, hb, 
                            hv >>

P4(self) == /\ pc[self] = "P4"
            /\ msgs' = msgs \cup {[type |-> "2a", bal |-> self - NA, acc |-> 0, vbal |-> 0, val |-> v[self]]}
            /\ pc' = [pc EXCEPT ![self] = "Done"]
            /\ UNCHANGED << maxBal, maxVBal, maxVal, promises, hb, hv, v >>

Proposer(self) == P1(self) \/ P2(self) \/ P3(self) \/ P4(self)

A(self) == /\ pc[self] = "A"
           /\ \E m \in msgs:
                /\ \/ /\ m.type = "1a" /\ m.bal > maxBal[self]
                      /\ maxBal' = [maxBal EXCEPT ![self] = m.bal]
                      /\ msgs' = msgs \cup {[type |-> "1b", bal |-> m.bal, acc |-> self, vbal |-> maxVBal[self], val |-> maxVal[self]]}
                      /\ UNCHANGED << maxVBal, maxVal >>
                   \/ /\ m.type = "2a" /\ m.bal >= maxBal[self]
                      /\ maxBal' = [maxBal EXCEPT ![self] = m.bal]
                      /\ maxVBal' = [maxVBal EXCEPT ![self] = m.bal]
                      /\ maxVal' = [maxVal EXCEPT ![self] = m.val]
                      /\ msgs' = msgs \cup {[type |-> "2b", bal |-> m.bal, acc |-> self, vbal |-> 0, val |-> m.val]}
           /\ pc' = [pc EXCEPT ![self] = "A"]
           /\ UNCHANGED << promises, hb, hv, v >>

Acceptor(self) == A(self)

Next == (\E self \in (NA + 1)..(NA + NB): Proposer(self))
           \/ (\E self \in 1..NA: Acceptor(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Agreement == \A b1 \in 1..NB : \A b2 \in 1..NB : \A v1 \in 1..NV : \A v2 \in 1..NV : (ChosenAt(b1, v1) /\ ChosenAt(b2, v2)) => v1 = v2
VotesAreProposed == \A m \in msgs : m.type = "2b" => [type |-> "2a", bal |-> m.bal, acc |-> 0, vbal |-> 0, val |-> m.val] \in msgs
OneValuePerBallot == \A m \in msgs : \A n \in msgs : (m.type = "2a" /\ n.type = "2a" /\ m.bal = n.bal) => m.val = n.val
PromisesAreHonest == \A m \in msgs : m.type = "1b" => m.vbal < m.bal /\ maxBal[m.acc] >= m.bal
=============================================================================
