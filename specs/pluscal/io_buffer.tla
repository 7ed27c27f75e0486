----------------------------- MODULE io_buffer -----------------------------
(***************************************************************************)
(* A lock-free IO buffer (the roadmap's "lock-free IO buffer",             *)
(* README.md:26-42): writers RESERVE a slot of the current buffer with one *)
(* compare-and-swap on its header — a record [off, writers, sealed] read   *)
(* and replaced as a whole —, copy their bytes, and RELEASE; a writer that *)
(* finds no room SEALS the buffer; the writer whose release (or seal)      *)
(* leaves a sealed buffer without writers FLUSHES it: the contents go to   *)
(* `flushed` and the header is reset.  Patient = FALSE flushes a sealed    *)
(* buffer while a writer is still copying: its bytes land in the buffer    *)
(* AFTER the flush (the assert in Copy fails).  (gen is a ghost: the        *)
(* generation a reservation was made in, noted in the step of its CAS —    *)
(* the header itself may well return to an earlier value after a flush.)   *)
(***************************************************************************)
EXTENDS Naturals, Sequences
CONSTANTS N, Cap, Patient

(* --algorithm io_buffer
variables hdr = [off |-> 0, writers |-> 0, sealed |-> FALSE],
          gen = 0,
          buf = [i \in 0..Cap - 1 |-> 0],
          flushed = <<>>;

process W \in 1..N
  variables seen = [off |-> 0, writers |-> 0, sealed |-> FALSE], at = 0, mygen = 0, done = FALSE;
begin
  Look:
    while ~done do
      seen := hdr;
      Try:
        if seen.sealed then
          skip;
        elsif seen.off = Cap then
          if hdr = seen then
            hdr := [off |-> seen.off, writers |-> seen.writers, sealed |-> TRUE];
            if seen.writers = 0 \/ ~Patient then
              goto Flush;
            end if;
          end if;
        elsif hdr = seen then
          hdr := [off |-> seen.off + 1, writers |-> seen.writers + 1, sealed |-> FALSE];
          at := seen.off;
          mygen := gen;
          goto Copy;
        end if;
    end while;
    goto Finish;
  Copy:
    assert gen = mygen;
    buf[at] := self;
  Release:
    seen := hdr;
  Release2:
    if hdr = seen then
      hdr := [off |-> seen.off, writers |-> seen.writers - 1, sealed |-> seen.sealed];
      done := TRUE;
      if seen.sealed /\ seen.writers = 1 /\ Patient then
        goto Flush;
      else
        goto Look;
      end if;
    else
      goto Release;
    end if;
  Flush:
    flushed := Append(flushed, hdr.off);
    gen := gen + 1;
    hdr := [off |-> 0, writers |-> 0, sealed |-> FALSE];
    goto Look;
  Finish:
    skip;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES hdr_off, hdr_writers, hdr_sealed, gen, buf, flushed, pc, seen_off, seen_writers, seen_sealed, at, mygen, done

vars == << hdr_off, hdr_writers, hdr_sealed, gen, buf, flushed, pc, seen_off, seen_writers, seen_sealed, at, mygen, done >>

(* record variables are kept field by field: r.f is r_f *)
hdr == [off |-> hdr_off, writers |-> hdr_writers, sealed |-> hdr_sealed]
seen == [self \in 1..N |-> [off |-> seen_off[self], writers |-> seen_writers[self], sealed |-> seen_sealed[self]]]

ProcSet == (1..N)

Init == (* Global variables *)
        /\ hdr_off = 0
        /\ hdr_writers = 0
        /\ hdr_sealed = FALSE
        /\ gen = 0
        /\ buf = [i \in 0..Cap - 1 |-> 0]
        /\ flushed = <<>>
        (* Process W *)
        /\ seen_off = [self \in 1..N |-> 0]
        /\ seen_writers = [self \in 1..N |-> 0]
        /\ seen_sealed = [self \in 1..N |-> FALSE]
        /\ at = [self \in 1..N |-> 0]
        /\ mygen = [self \in 1..N |-> 0]
        /\ done = [self \in 1..N |-> FALSE]
        /\ pc = [self \in ProcSet |-> "Look"]

Look(self) == /\ pc[self] = "Look"
              /\ IF ~done[self]
                    THEN /\ seen_off' = [seen_off EXCEPT ![self] = hdr_off]
                         /\ seen_writers' = [seen_writers EXCEPT ![self] = hdr_writers]
                         /\ seen_sealed' = [seen_sealed EXCEPT ![self] = hdr_sealed]
                         /\ pc' = [pc EXCEPT ![self] = "Try"]
                    ELSE /\ pc' = [pc EXCEPT ![self] = "Finish"]
                         /\ UNCHANGED << seen_off, seen_writers, 
                                         seen_sealed >>
              /\ UNCHANGED << hdr_off, hdr_writers, hdr_sealed, gen, buf, 
                              flushed, at, mygen, done >>

Try(self) == /\ pc[self] = "Try"
             /\ IF seen_sealed[self]
                   THEN /\ TRUE
                        /\ pc' = [pc EXCEPT ![self] = "Look"]
                        /\ UNCHANGED << hdr_off, hdr_writers, hdr_sealed, at, 
                                        mygen >>
                   ELSE /\ IF seen_off[self] = Cap
                              THEN /\ IF (hdr_off = seen_off[self] /\ hdr_writers = seen_writers[self] /\ hdr_sealed = seen_sealed[self])
                                         THEN /\ hdr_off' = seen_off[self]
                                              /\ hdr_writers' = seen_writers[self]
                                              /\ hdr_sealed' = TRUE
                                              /\ IF seen_writers[self] = 0 \/ ~Patient
                                                    THEN /\ pc' = [pc EXCEPT ![self] = "Flush"]
                                                    ELSE /\ pc' = [pc EXCEPT ![self] = "Look"]
                                         ELSE /\ pc' = [pc EXCEPT ![self] = "Look"]
                                              /\ UNCHANGED << hdr_off, 
                                                              hdr_writers, 
                                                              hdr_sealed >>
                                   /\ UNCHANGED << at, mygen >>
                              ELSE /\ IF (hdr_off = seen_off[self] /\ hdr_writers = seen_writers[self] /\ hdr_sealed = seen_sealed[self])
                                         THEN /\ hdr_off' = seen_off[self] + 1
                                              /\ hdr_writers' = seen_writers[self] + 1
                                              /\ hdr_sealed' = FALSE
                                              /\ at' = [at EXCEPT ![self] = seen_off[self]]
                                              /\ mygen' = [mygen EXCEPT ![self] = gen]
                                              /\ pc' = [pc EXCEPT ![self] = "Copy"]
                                         ELSE /\ pc' = [pc EXCEPT ![self] = "Look"]
                                              /\ UNCHANGED << hdr_off, 
                                                              hdr_writers, 
                                                              hdr_sealed, at, 
                                                              mygen >>
             /\ UNCHANGED << gen, buf, flushed, seen_off, seen_writers, 
                             seen_sealed, done >>

Copy(self) == /\ pc[self] = "Copy"
              /\ Assert(gen = mygen[self], 
                        "Failure of assertion at line 49, column 5.")
              /\ buf' = [buf EXCEPT ![at[self]] = self]
              /\ pc' = [pc EXCEPT ![self] = "Release"]
              /\ UNCHANGED << hdr_off, hdr_writers, hdr_sealed, gen, flushed, 
                              seen_off, seen_writers, seen_sealed, at, mygen, 
                              done >>

Release(self) == /\ pc[self] = "Release"
                 /\ seen_off' = [seen_off EXCEPT ![self] = hdr_off]
                 /\ seen_writers' = [seen_writers EXCEPT ![self] = hdr_writers]
                 /\ seen_sealed' = [seen_sealed EXCEPT ![self] = hdr_sealed]
                 /\ pc' = [pc EXCEPT ![self] = "Release2"]
                 /\ UNCHANGED << hdr_off, hdr_writers, hdr_sealed, gen, buf, 
                                 flushed, at, mygen, done >>

Release2(self) == /\ pc[self] = "Release2"
                  /\ IF (hdr_off = seen_off[self] /\ hdr_writers = seen_writers[self] /\ hdr_sealed = seen_sealed[self])
                        THEN /\ hdr_off' = seen_off[self]
                             /\ hdr_writers' = seen_writers[self] - 1
                             /\ hdr_sealed' = seen_sealed[self]
                             /\ done' = [done EXCEPT ![self] = TRUE]
                             /\ IF seen_sealed[self] /\ seen_writers[self] = 1 /\ Patient
                                   THEN /\ pc' = [pc EXCEPT ![self] = "Flush"]
                                   ELSE /\ pc' = [pc EXCEPT ![self] = "Look"]
                        ELSE /\ pc' = [pc EXCEPT ![self] = "Release"]
                             /\ UNCHANGED << hdr_off, hdr_writers, 
                                             hdr_sealed, done >>
                  /\ UNCHANGED << gen, buf, flushed, seen_off, seen_writers, 
                                  seen_sealed, at, mygen >>

Flush(self) == /\ pc[self] = "Flush"
               /\ flushed' = Append(flushed, hdr_off)
               /\ gen' = gen + 1
               /\ hdr_off' = 0
               /\ hdr_writers' = 0
               /\ hdr_sealed' = FALSE
               /\ pc' = [pc EXCEPT ![self] = "Look"]
               /\ UNCHANGED << buf, seen_off, seen_writers, seen_sealed, at, 
                               mygen, done >>

Finish(self) == /\ pc[self] = "Finish"
                /\ TRUE
                /\ pc' = [pc EXCEPT ![self] = "Done"]
                /\ UNCHANGED << hdr_off, hdr_writers, hdr_sealed, gen, buf, 
                                flushed, seen_off, seen_writers, seen_sealed, 
                                at, mygen, done >>

W(self) == Look(self) \/ Try(self) \/ Copy(self) \/ Release(self) \/ Release2(self) \/ Flush(self) \/ Finish(self)

Next == (\E self \in 1..N: W(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

HeaderInRange == hdr.off \in 0..Cap /\ hdr.writers \in 0..N
SealedIsFull == hdr.sealed => hdr.off = Cap
FlushedFull == \A k \in 1..Len(flushed) : flushed[k] = Cap
=============================================================================
