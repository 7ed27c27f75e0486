----------------------------- MODULE lost_update -----------------------------
(***************************************************************************)
(* The broken counterpart of cas_counter: read and write are two steps with *)
(* no compare-and-swap, so an increment can be lost and the final assert   *)
(* fails.  Exercises either/or, with, skip, strings, elsif.                *)
(***************************************************************************)
EXTENDS Naturals

(* --algorithm lost_update
variables cell = 0, mode = "idle", log = 0;

process Worker \in 1..2
  variables tmp = 0;
begin
  Pick:
    either
      mode := "add";
    or
      with d \in {2, 5} do
        log := log + d;
      end with;
      mode := "noise";
    end either;
  Read:  tmp := cell;
  Write:
    if mode = "add" then
      cell := tmp + 1;
    elsif mode = "noise" then
      cell := tmp + 1;
      log := log + 1;
    else
      skip;
    end if;
end process

process Checker = 3
begin
  Final:
    await pc[1] = "Done" /\ pc[2] = "Done";
    assert cell = 2;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES cell, mode, log, pc, tmp

vars == << cell, mode, log, pc, tmp >>

ProcSet == (1..2) \cup {3}

Init == (* Global variables *)
        /\ cell = 0
        /\ mode = "idle"
        /\ log = 0
        (* Process Worker *)
        /\ tmp = [self \in 1..2 |-> 0]
        /\ pc = [self \in ProcSet |-> CASE self \in 1..2 -> "Pick"
                                        [] self = 3 -> "Final"]

Pick(self) == /\ pc[self] = "Pick"
              /\ \/ /\ mode' = "add"
                    /\ UNCHANGED log
                 \/ /\ \E d \in {2, 5}:
                         /\ log' = log + d
                    /\ mode' = "noise"
              /\ pc' = [pc EXCEPT ![self] = "Read"]
              /\ UNCHANGED << cell, tmp >>

Read(self) == /\ pc[self] = "Read"
              /\ tmp' = [tmp EXCEPT ![self] = cell]
              /\ pc' = [pc EXCEPT ![self] = "Write"]
              /\ UNCHANGED << cell, mode, log >>

Write(self) == /\ pc[self] = "Write"
               /\ IF mode = "add"
                     THEN /\ cell' = tmp[self] + 1
                          /\ UNCHANGED log
                     ELSE /\ IF mode = "noise"
                                THEN /\ cell' = tmp[self] + 1
                                     /\ log' = log + 1
                                ELSE /\ TRUE
                                     /\ UNCHANGED << cell, log >>
               /\ pc' = [pc EXCEPT ![self] = "Done"]
               /\ UNCHANGED << mode, tmp >>

Worker(self) == Pick(self) \/ Read(self) \/ Write(self)

Final == /\ pc[3] = "Final"
         /\ pc[1] = "Done" /\ pc[2] = "Done"
         /\ Assert(cell = 2, 
                   "Failure of assertion at line 40, column 5.")
         /\ pc' = [pc EXCEPT ![3] = "Done"]
         /\ UNCHANGED << cell, mode, log, tmp >>

Checker == Final

Next == Checker
           \/ (\E self \in 1..2: Worker(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION
=============================================================================
