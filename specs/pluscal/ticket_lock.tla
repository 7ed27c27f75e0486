----------------------------- MODULE ticket_lock -----------------------------
(***************************************************************************)
(* A ticket lock (fetch-and-increment to take a ticket, spin until served). *)
(* Exercises a define block (operators with and without parameters),       *)
(* macros, CONSTANTS, a quantified invariant and a function variable.      *)
(***************************************************************************)
EXTENDS Naturals
CONSTANTS P, Rounds

(* --algorithm ticket_lock
variables next_ticket = 0, now_serving = 0,
          holding = [i \in 1..P |-> FALSE];

define
  Waiting == next_ticket - now_serving
  InCS(i) == holding[i]
  NobodyElse(i) == \A j \in 1..P : j = i \/ ~InCS(j)
end define;

macro fetch_and_inc(dst, cell)
begin
  dst := cell;
  cell := cell + 1;
end macro;

process Thread \in 1..P
  variables my = 0, laps = 0;
begin
  Start:
    while laps < Rounds do
      Take:  fetch_and_inc(my, next_ticket);
      Spin:  await now_serving = my;
             holding[self] := TRUE;
      Crit:  assert NobodyElse(self);
      Exit:  holding[self] := FALSE;
             now_serving := now_serving + 1;
             laps := laps + 1;
    end while;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES next_ticket, now_serving, holding, pc

(* define statement *)
Waiting == next_ticket - now_serving

InCS(i) == holding[i]

NobodyElse(i) == \A j \in 1..P : j = i \/ ~InCS(j)

VARIABLES my, laps

vars == << next_ticket, now_serving, holding, pc, my, laps >>

ProcSet == (1..P)

Init == (* Global variables *)
        /\ next_ticket = 0
        /\ now_serving = 0
        /\ holding = [i \in 1..P |-> FALSE]
        (* Process Thread *)
        /\ my = [self \in 1..P |-> 0]
        /\ laps = [self \in 1..P |-> 0]
        /\ pc = [self \in ProcSet |-> "Start"]

Start(self) == /\ pc[self] = "Start"
               /\ IF laps[self] < Rounds
                     THEN /\ pc' = [pc EXCEPT ![self] = "Take"]
                     ELSE /\ pc' = [pc EXCEPT ![self] = "Done"]
               /\ UNCHANGED << next_ticket, now_serving, holding, my, laps >>

Take(self) == /\ pc[self] = "Take"
              /\ my' = [my EXCEPT ![self] = next_ticket]
              /\ next_ticket' = next_ticket + 1
              /\ pc' = [pc EXCEPT ![self] = "Spin"]
              /\ UNCHANGED << now_serving, holding, laps >>

Spin(self) == /\ pc[self] = "Spin"
              /\ now_serving = my[self]
              /\ holding' = [holding EXCEPT ![self] = TRUE]
              /\ pc' = [pc EXCEPT ![self] = "Crit"]
              /\ UNCHANGED << next_ticket, now_serving, my, laps >>

Crit(self) == /\ pc[self] = "Crit"
              /\ Assert(NobodyElse(self), 
                        "Failure of assertion at line 34, column 14.")
              /\ pc' = [pc EXCEPT ![self] = "Exit"]
              /\ UNCHANGED << next_ticket, now_serving, holding, my, laps >>

Exit(self) == /\ pc[self] = "Exit"
              /\ holding' = [holding EXCEPT ![self] = FALSE]
              /\ now_serving' = now_serving + 1
              /\ laps' = [laps EXCEPT ![self] = laps[self] + 1]
              /\ pc' = [pc EXCEPT ![self] = "Start"]
              /\ UNCHANGED << next_ticket, my >>

Thread(self) == Start(self) \/ Take(self) \/ Spin(self) \/ Crit(self) \/ Exit(self)

Next == (\E self \in 1..P: Thread(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

Mutex == \A i \in 1..P : \A j \in 1..P : (i # j) => ~(holding[i] /\ holding[j])
Fifo == now_serving <= next_ticket /\ Waiting <= P
=============================================================================
