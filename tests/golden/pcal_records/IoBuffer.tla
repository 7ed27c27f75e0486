------------------------------ MODULE IoBuffer ------------------------------
(***************************************************************************)
(* HAND-WRITTEN translation of specs/pluscal/io_buffer.tla in the style of  *)
(* pcal2tla (p-manual App. B), written from the ALGORITHM text: the header  *)
(* hdr and every writer's copy seen stay whole RECORDS ([off, writers,      *)
(* sealed], compared with `hdr = seen[self]` and replaced as a whole), as   *)
(* the Java translator keeps them, where the product's front-end keeps them *)
(* field by field.  Evaluated by oracle/tlaplus.py; see EpochGc.tla.        *)
(***************************************************************************)
EXTENDS Naturals, Sequences, TLC
CONSTANTS N, Cap, Patient

VARIABLES hdr, gen, buf, flushed, pc, seen, at, mygen, done

vars == << hdr, gen, buf, flushed, pc, seen, at, mygen, done >>

ProcSet == (1..N)

Fresh == [off |-> 0, writers |-> 0, sealed |-> FALSE]

Init == /\ hdr = Fresh
        /\ gen = 0
        /\ buf = [i \in 0..Cap - 1 |-> 0]
        /\ flushed = << >>
        /\ seen = [self \in 1..N |-> Fresh]
        /\ at = [self \in 1..N |-> 0]
        /\ mygen = [self \in 1..N |-> 0]
        /\ done = [self \in 1..N |-> FALSE]
        /\ pc = [self \in ProcSet |-> "Look"]

Goto(self, l) == pc' = [pc EXCEPT ![self] = l]

Look(self) == /\ pc[self] = "Look"
              /\ IF ~done[self]
                    THEN /\ seen' = [seen EXCEPT ![self] = hdr]
                         /\ Goto(self, "Try")
                    ELSE /\ Goto(self, "Finish")
                         /\ seen' = seen
              /\ UNCHANGED << hdr, gen, buf, flushed, at, mygen, done >>

Try(self) == /\ pc[self] = "Try"
             /\ IF seen[self].sealed
                   THEN /\ Goto(self, "Look")
                        /\ UNCHANGED << hdr, at, mygen >>
                   ELSE IF seen[self].off = Cap
                           THEN /\ IF hdr = seen[self]
                                      THEN /\ hdr' = [off |-> seen[self].off, writers |-> seen[self].writers, sealed |-> TRUE]
                                           /\ IF seen[self].writers = 0 \/ ~Patient
                                                 THEN Goto(self, "Flush")
                                                 ELSE Goto(self, "Look")
                                      ELSE /\ Goto(self, "Look")
                                           /\ hdr' = hdr
                                /\ UNCHANGED << at, mygen >>
                           ELSE IF hdr = seen[self]
                                   THEN /\ hdr' = [off |-> seen[self].off + 1, writers |-> seen[self].writers + 1, sealed |-> FALSE]
                                        /\ at' = [at EXCEPT ![self] = seen[self].off]
                                        /\ mygen' = [mygen EXCEPT ![self] = gen]
                                        /\ Goto(self, "Copy")
                                   ELSE /\ Goto(self, "Look")
                                        /\ UNCHANGED << hdr, at, mygen >>
             /\ UNCHANGED << gen, buf, flushed, seen, done >>

Copy(self) == /\ pc[self] = "Copy"
              /\ Assert(gen = mygen[self], "Failure of assertion at Copy: the buffer was flushed under a writer")
              /\ buf' = [buf EXCEPT ![at[self]] = self]
              /\ Goto(self, "Release")
              /\ UNCHANGED << hdr, gen, flushed, seen, at, mygen, done >>

Release(self) == /\ pc[self] = "Release"
                 /\ seen' = [seen EXCEPT ![self] = hdr]
                 /\ Goto(self, "Release2")
                 /\ UNCHANGED << hdr, gen, buf, flushed, at, mygen, done >>

Release2(self) == /\ pc[self] = "Release2"
                  /\ IF hdr = seen[self]
                        THEN /\ hdr' = [off |-> seen[self].off, writers |-> seen[self].writers - 1, sealed |-> seen[self].sealed]
                             /\ done' = [done EXCEPT ![self] = TRUE]
                             /\ IF seen[self].sealed /\ seen[self].writers = 1 /\ Patient
                                   THEN Goto(self, "Flush")
                                   ELSE Goto(self, "Look")
                        ELSE /\ Goto(self, "Release")
                             /\ UNCHANGED << hdr, done >>
                  /\ UNCHANGED << gen, buf, flushed, seen, at, mygen >>

Flush(self) == /\ pc[self] = "Flush"
               /\ flushed' = Append(flushed, hdr.off)
               /\ gen' = gen + 1
               /\ hdr' = Fresh
               /\ Goto(self, "Look")
               /\ UNCHANGED << buf, seen, at, mygen, done >>

Finish(self) == /\ pc[self] = "Finish"
                /\ Goto(self, "Done")
                /\ UNCHANGED << hdr, gen, buf, flushed, seen, at, mygen, done >>

W(self) == Look(self) \/ Try(self) \/ Copy(self) \/ Release(self) \/ Release2(self) \/ Flush(self) \/ Finish(self)

Next == (\E self \in 1..N: W(self))
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

HeaderInRange == hdr.off \in 0..Cap /\ hdr.writers \in 0..N
SealedIsFull == hdr.sealed => hdr.off = Cap
FlushedFull == \A k \in 1..Len(flushed) : flushed[k] = Cap
=============================================================================
