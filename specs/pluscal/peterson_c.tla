----------------------------- MODULE peterson_c -----------------------------
(***************************************************************************)
(* specs/pluscal/peterson.tla in PlusCal's c-syntax ("Pluscal has two      *)
(* forms, c and p.  They are functionally identical", README.md:212-215 of *)
(* the reference): braces instead of begin/end, tests in parentheses.      *)
(***************************************************************************)
EXTENDS Naturals

(* --algorithm peterson_c {
variables flag = [i \in 0..1 |-> FALSE], turn = 0, in_cs = 0;

process (Proc \in 0..1)
{
  Loop:
    while (TRUE) {
      Want:  flag[self] := TRUE;
      Yield: turn := 1 - self;
      Wait:  await flag[1 - self] = FALSE \/ turn = self;
      Enter: in_cs := in_cs + 1;
      CS:    assert in_cs = 1;
      Leave: in_cs := in_cs - 1;
      Reset: flag[self] := FALSE;
    }
}

} *)
\* BEGIN TRANSLATION
VARIABLES flag, turn, in_cs, pc

vars == << flag, turn, in_cs, pc >>

ProcSet == (0..1)

Init == (* Global variables *)
        /\ flag = [i \in 0..1 |-> FALSE]
        /\ turn = 0
        /\ in_cs = 0
        /\ pc = [self \in ProcSet |-> "Loop"]

Loop(self) == /\ pc[self] = "Loop"
              /\ pc' = [pc EXCEPT ![self] = "Want"]
              /\ UNCHANGED << flag, turn, in_cs >>

Want(self) == /\ pc[self] = "Want"
              /\ flag' = [flag EXCEPT ![self] = TRUE]
              /\ pc' = [pc EXCEPT ![self] = "Yield"]
              /\ UNCHANGED << turn, in_cs >>

Yield(self) == /\ pc[self] = "Yield"
               /\ turn' = 1 - self
               /\ pc' = [pc EXCEPT ![self] = "Wait"]
               /\ UNCHANGED << flag, in_cs >>

Wait(self) == /\ pc[self] = "Wait"
              /\ flag[1 - self] = FALSE \/ turn = self
              /\ pc' = [pc EXCEPT ![self] = "Enter"]
              /\ UNCHANGED << flag, turn, in_cs >>

Enter(self) == /\ pc[self] = "Enter"
               /\ in_cs' = in_cs + 1
               /\ pc' = [pc EXCEPT ![self] = "CS"]
               /\ UNCHANGED << flag, turn >>

CS(self) == /\ pc[self] = "CS"
            /\ Assert(in_cs = 1, 
                      "Failure of assertion at line 20, column 14.")
            /\ pc' = [pc EXCEPT ![self] = "Leave"]
            /\ UNCHANGED << flag, turn, in_cs >>

Leave(self) == /\ pc[self] = "Leave"
               /\ in_cs' = in_cs - 1
               /\ pc' = [pc EXCEPT ![self] = "Reset"]
               /\ UNCHANGED << flag, turn >>

Reset(self) == /\ pc[self] = "Reset"
               /\ flag' = [flag EXCEPT ![self] = FALSE]
               /\ pc' = [pc EXCEPT ![self] = "Loop"]
               /\ UNCHANGED << turn, in_cs >>

Proc(self) == Loop(self) \/ Want(self) \/ Yield(self) \/ Wait(self) \/ Enter(self) \/ CS(self) \/ Leave(self) \/ Reset(self)

Next == (\E self \in 0..1: Proc(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

MutualExclusion == in_cs <= 1
TurnInRange == turn \in 0..1
=============================================================================
