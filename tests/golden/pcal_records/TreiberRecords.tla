--------------------------- MODULE TreiberRecords ---------------------------
(* specs/pluscal/treiber_records.tla the way pcal2tla translates it (p-manual section 3.8 / App. B): the record variables stay
   record-valued — mem a function to records, top a record, old a function from process ids to records — and a field is assigned
   with EXCEPT !.f.  Written by hand: the product keeps records field by field (tla_rust_amd/csrc/pcal.h, RECORDS), and
   tests/test_pcal.py checks that the two are the same state graph. *)
EXTENDS Naturals, TLC
CONSTANT N
VARIABLES mem, top, popped, pc, old, nxt

vars == << mem, top, popped, pc, old, nxt >>

ProcSet == (1..N)

Init == /\ mem = [a \in 1..N |-> [val |-> 0, next |-> 0]]
        /\ top = [ptr |-> 0, ver |-> 0]
        /\ popped = [p \in 1..N |-> 0]
        /\ old = [self \in 1..N |-> [ptr |-> 0, ver |-> 0]]
        /\ nxt = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "Fill"]

Fill(self) == /\ pc[self] = "Fill"
              /\ mem' = [mem EXCEPT ![self].val = 10 * self]
              /\ pc' = [pc EXCEPT ![self] = "PushRead"]
              /\ UNCHANGED << top, popped, old, nxt >>

PushRead(self) == /\ pc[self] = "PushRead"
                  /\ old' = [old EXCEPT ![self] = top]
                  /\ pc' = [pc EXCEPT ![self] = "PushLink"]
                  /\ UNCHANGED << mem, top, popped, nxt >>

PushLink(self) == /\ pc[self] = "PushLink"
                  /\ mem' = [mem EXCEPT ![self].next = old[self].ptr]
                  /\ pc' = [pc EXCEPT ![self] = "PushCas"]
                  /\ UNCHANGED << top, popped, old, nxt >>

PushCas(self) == /\ pc[self] = "PushCas"
                 /\ IF top = old[self]
                       THEN /\ top' = [ptr |-> self, ver |-> old[self].ver + 1]
                            /\ pc' = [pc EXCEPT ![self] = "PopRead"]
                       ELSE /\ pc' = [pc EXCEPT ![self] = "PushRead"]
                            /\ top' = top
                 /\ UNCHANGED << mem, popped, old, nxt >>

PopRead(self) == /\ pc[self] = "PopRead"
                 /\ old' = [old EXCEPT ![self] = top]
                 /\ Assert(old'[self].ptr # 0, "Failure of assertion at line 27, column 13.")
                 /\ pc' = [pc EXCEPT ![self] = "PopNext"]
                 /\ UNCHANGED << mem, top, popped, nxt >>

PopNext(self) == /\ pc[self] = "PopNext"
                 /\ nxt' = [nxt EXCEPT ![self] = mem[old[self].ptr].next]
                 /\ pc' = [pc EXCEPT ![self] = "PopCas"]
                 /\ UNCHANGED << mem, top, popped, old >>

PopCas(self) == /\ pc[self] = "PopCas"
                /\ IF top = old[self]
                      THEN /\ top' = [ptr |-> nxt[self], ver |-> old[self].ver + 1]
                           /\ popped' = [popped EXCEPT ![self] = mem[old[self].ptr].val]
                           /\ pc' = [pc EXCEPT ![self] = "Done"]
                      ELSE /\ pc' = [pc EXCEPT ![self] = "PopRead"]
                           /\ UNCHANGED << top, popped >>
                /\ UNCHANGED << mem, old, nxt >>

worker(self) == Fill(self) \/ PushRead(self) \/ PushLink(self) \/ PushCas(self) \/ PopRead(self) \/ PopNext(self) \/ PopCas(self)

Next == (\E self \in 1..N: worker(self))
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

PoppedOnce == \A p \in 1..N : \A q \in 1..N : (p # q /\ popped[p] # 0) => popped[p] # popped[q]
TopIsNode == top.ptr \in 0..N /\ top.ver <= 2 * N
NextIsNode == \A a \in 1..N : mem[a].next \in 0..N /\ mem[a].next # a
OldIsNode == \A p \in 1..N : old[p].ptr \in 0..N
=============================================================================
