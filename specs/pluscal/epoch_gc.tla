------------------------------ MODULE epoch_gc ------------------------------
(***************************************************************************)
(* Epoch-based memory reclamation (the "lock-free epoch-based garbage      *)
(* collector" of the reference's roadmap, README.md:26-42), three epochs.  *)
(* A thread pins the current epoch before it touches shared memory         *)
(* (active, local), reads the shared pointer and dereferences it; as a     *)
(* writer it installs a fresh node with an atomic exchange and RETIRES the *)
(* old one into the current epoch.  The global epoch advances only when    *)
(* every pinned thread has seen it; a node retired in epoch e is freed     *)
(* when the epoch is e + Grace.  Grace = 2 is the algorithm; Grace = 1     *)
(* frees a node a reader pinned in epoch e may still hold: the dereference *)
(* in R3 hits freed memory (the assert fails).                             *)
(***************************************************************************)
EXTENDS Naturals
CONSTANTS N, Grace

(* --algorithm epoch_gc
variables epoch = 0,
          head = 1,
          state = [n \in 1..N + 1 |-> IF n = 1 THEN "live" ELSE "free"],
          retEpoch = [n \in 1..N + 1 |-> 0],
          local = [t \in 1..N |-> 0],
          active = [t \in 1..N |-> FALSE];

process T \in 1..N
  variables p = 0, fresh = 0, old = 0, k = 1;
begin
  R1:
    active[self] := TRUE;
    local[self] := epoch;
  R2:
    p := head;
  R3:
    assert state[p] # "free";
  R4:
    active[self] := FALSE;
    p := 0;
  W1:
    active[self] := TRUE;
    local[self] := epoch;
  W2:
    with n \in 1..N + 1 do
      await state[n] = "free";
      fresh := n;
      state[n] := "live";
    end with;
  W3:
    old := head || head := fresh;
  W4:
    state[old] := "retired";
    retEpoch[old] := epoch;
  W5:
    active[self] := FALSE;
  A1:
    if \A t \in 1..N : ~active[t] \/ local[t] = epoch then
      epoch := (epoch + 1) % 3;
    end if;
  A2:
    while k <= N + 1 do
      if state[k] = "retired" /\ retEpoch[k] = (epoch + 3 - Grace) % 3 then
        state[k] := "free";
      end if;
      k := k + 1;
    end while;
end process

end algorithm *)
\* BEGIN TRANSLATION
VARIABLES epoch, head, state, retEpoch, local, active, pc, p, fresh, old, k

vars == << epoch, head, state, retEpoch, local, active, pc, p, fresh, old, k >>

ProcSet == (1..N)

Init == (* Global variables *)
        /\ epoch = 0
        /\ head = 1
        /\ state = [n \in 1..N + 1 |-> IF n = 1 THEN "live" ELSE "free"]
        /\ retEpoch = [n \in 1..N + 1 |-> 0]
        /\ local = [t \in 1..N |-> 0]
        /\ active = [t \in 1..N |-> FALSE]
        (* Process T *)
        /\ p = [self \in 1..N |-> 0]
        /\ fresh = [self \in 1..N |-> 0]
        /\ old = [self \in 1..N |-> 0]
        /\ k = [self \in 1..N |-> 1]
        /\ pc = [self \in ProcSet |-> "R1"]

R1(self) == /\ pc[self] = "R1"
            /\ active' = [active EXCEPT ![self] = TRUE]
            /\ local' = [local EXCEPT ![self] = epoch]
            /\ pc' = [pc EXCEPT ![self] = "R2"]
            /\ UNCHANGED << epoch, head, state, retEpoch, p, fresh, old, k >>

R2(self) == /\ pc[self] = "R2"
            /\ p' = [p EXCEPT ![self] = head]
            /\ pc' = [pc EXCEPT ![self] = "R3"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, active, 
                            fresh, old, k >>

R3(self) == /\ pc[self] = "R3"
            /\ Assert(state[p[self]] # "free", 
                      "Failure of assertion at line 34, column 5.")
            /\ pc' = [pc EXCEPT ![self] = "R4"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, active, p, 
                            fresh, old, k >>

R4(self) == /\ pc[self] = "R4"
            /\ active' = [active EXCEPT ![self] = FALSE]
            /\ p' = [p EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "W1"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, fresh, old, 
                            k >>

W1(self) == /\ pc[self] = "W1"
            /\ active' = [active EXCEPT ![self] = TRUE]
            /\ local' = [local EXCEPT ![self] = epoch]
            /\ pc' = [pc EXCEPT ![self] = "W2"]
            /\ UNCHANGED << epoch, head, state, retEpoch, p, fresh, old, k >>

W2(self) == /\ pc[self] = "W2"
            /\ \E n \in 1..N + 1:
                 /\ state[n] = "free"
                 /\ fresh' = [fresh EXCEPT ![self] = n]
                 /\ state' = [state EXCEPT ![n] = "live"]
            /\ pc' = [pc EXCEPT ![self] = "W3"]
            /\ UNCHANGED << epoch, head, retEpoch, local, active, p, old, 
                            k >>

W3(self) == /\ pc[self] = "W3"
            /\ old' = [old EXCEPT ![self] = head]
            /\ head' = fresh[self]
            /\ pc' = [pc EXCEPT ![self] = "W4"]
            /\ UNCHANGED << epoch, state, retEpoch, local, active, p, fresh, 
                            k >>

W4(self) == /\ pc[self] = "W4"
            /\ state' = [state EXCEPT ![old[self]] = "retired"]
            /\ retEpoch' = [retEpoch EXCEPT ![old[self]] = epoch]
            /\ pc' = [pc EXCEPT ![self] = "W5"]
            /\ UNCHANGED << epoch, head, local, active, p, fresh, old, k >>

W5(self) == /\ pc[self] = "W5"
            /\ active' = [active EXCEPT ![self] = FALSE]
            /\ pc' = [pc EXCEPT ![self] = "A1"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, p, fresh, 
                            old, k >>

A1(self) == /\ pc[self] = "A1"
            /\ IF \A t \in 1..N : ~active[t] \/ local[t] = epoch
                  THEN /\ epoch' = (epoch + 1) % 3
                  ELSE /\ TRUE
                       /\ UNCHANGED epoch
            /\ pc' = [pc EXCEPT ![self] = "A2"]
            /\ UNCHANGED << head, state, retEpoch, local, active, p, fresh, 
                            old, k >>

A2(self) == /\ pc[self] = "A2"
            /\ IF k[self] <= N + 1
                  THEN /\ IF state[k[self]] = "retired" /\ retEpoch[k[self]] = (epoch + 3 - Grace) % 3
                             THEN /\ state' = [state EXCEPT ![k[self]] = "free"]
                             ELSE /\ TRUE
                                  /\ UNCHANGED state
                       /\ k' = [k EXCEPT ![self] = k[self] + 1]
                       /\ pc' = [pc EXCEPT ![self] = "A2"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "Done"]
                       /\ UNCHANGED << state, k >>
            /\ UNCHANGED << epoch, head, retEpoch, local, active, p, fresh, 
                            old >>

T(self) == R1(self) \/ R2(self) \/ R3(self) \/ R4(self) \/ W1(self) \/ W2(self) \/ W3(self) \/ W4(self) \/ W5(self) \/ A1(self) \/ A2(self)

Next == (\E self \in 1..N: T(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

HeadIsLive == state[head] = "live"
NoDanglingReader == \A t \in 1..N : (active[t] /\ p[t] # 0) => state[p[t]] # "free"
EpochInRange == epoch \in 0..2 /\ \A t \in 1..N : local[t] \in 0..2
=============================================================================
