---------------------------- MODULE treiber_stack ----------------------------
(***************************************************************************)
(* Treiber's lock-free stack with compare-and-swap, pointer style: nodes    *)
(* are 1..N, 0 is the null pointer, `next` is the link field.  Every thread *)
(* pushes its own node and then pops one node.  (README.md:26-42 of the     *)
(* reference: lock-free data structures are what it wants to model.)        *)
(***************************************************************************)
EXTENDS Naturals
CONSTANT N

(* --algorithm treiber_stack
variables top = 0,
          next = [n \in 1..N |-> 0],
          popped = [n \in 1..N |-> 0];

process T \in 1..N
  variables t = 0, nx = 0, got = 0;
begin
  PushRead: t := top;
  PushLink: next[self] := t;
  PushCas:
    if top = t then
      top := self;
    else
      goto PushRead;
    end if;
  PopRead:
    t := top;
    if t = 0 then
      goto Fin;
    end if;
  PopNext: nx := next[t];
  PopCas:
    if top = t then
      top := nx;
      got := t;
    else
      goto PopRead;
    end if;
  Mark: popped[got] := popped[got] + 1;
  Fin:  skip;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES top, next, popped, pc, t, nx, got

vars == << top, next, popped, pc, t, nx, got >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ top = 0
        /\ next = [n \in 1..N |-> 0]
        /\ popped = [n \in 1..N |-> 0]
        (* Process T *)
        /\ t = [self \in 1..N |-> 0]
        /\ nx = [self \in 1..N |-> 0]
        /\ got = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "PushRead"]

PushRead(self) == /\ pc[self] = "PushRead"
                  /\ t' = [t EXCEPT ![self] = top]
                  /\ pc' = [pc EXCEPT ![self] = "PushLink"]
                  /\ UNCHANGED << top, next, popped, nx, got >>

PushLink(self) == /\ pc[self] = "PushLink"
                  /\ next' = [next EXCEPT ![self] = t[self]]
                  /\ pc' = [pc EXCEPT ![self] = "PushCas"]
                  /\ UNCHANGED << top, popped, t, nx, got >>

PushCas(self) == /\ pc[self] = "PushCas"
                 /\ IF top = t[self]
                       THEN /\ top' = self
                            /\ pc' = [pc EXCEPT ![self] = "PopRead"]
                       ELSE /\ pc' = [pc EXCEPT ![self] = "PushRead"]
                            /\ UNCHANGED top
                 /\ UNCHANGED << next, popped, t, nx, got >>

PopRead(self) == /\ pc[self] = "PopRead"
                 /\ t' = [t EXCEPT ![self] = top]
                 /\ IF t'[self] = 0
                       THEN /\ pc' = [pc EXCEPT ![self] = "Fin"]
                       ELSE /\ pc' = [pc EXCEPT ![self] = "PopNext"]
                 /\ UNCHANGED << top, next, popped, nx, got >>

PopNext(self) == /\ pc[self] = "PopNext"
                 /\ nx' = [nx EXCEPT ![self] = next[t[self]]]
                 /\ pc' = [pc EXCEPT ![self] = "PopCas"]
                 /\ UNCHANGED << top, next, popped, t, got >>

PopCas(self) == /\ pc[self] = "PopCas"
                /\ IF top = t[self]
                      THEN /\ top' = nx[self]
                           /\ got' = [got EXCEPT ![self] = t[self]]
                           /\ pc' = [pc EXCEPT ![self] = "Mark"]
                      ELSE /\ pc' = [pc EXCEPT ![self] = "PopRead"]
                           /\ UNCHANGED << top, got >>
                /\ UNCHANGED << next, popped, t, nx >>

Mark(self) == /\ pc[self] = "Mark"
              /\ popped' = [popped EXCEPT ![got[self]] = popped[got[self]] + 1]
              /\ pc' = [pc EXCEPT ![self] = "Fin"]
              /\ UNCHANGED << top, next, t, nx, got >>

Fin(self) == /\ pc[self] = "Fin"
             /\ TRUE
             /\ pc' = [pc EXCEPT ![self] = "Done"]
             /\ UNCHANGED << top, next, popped, t, nx, got >>

T(self) == PushRead(self) \/ PushLink(self) \/ PushCas(self) \/ PopRead(self) \/ PopNext(self) \/ PopCas(self) \/ Mark(self) \/ Fin(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

PoppedOnce == \A n \in 1..N : popped[n] <= 1
TopIsNode == top \in 0..N
AllDone == \A p \in 1..N : pc[p] = "Done"
Conservation == AllDone => (\A n \in 1..N : popped[n] = 1) /\ top = 0
=============================================================================
