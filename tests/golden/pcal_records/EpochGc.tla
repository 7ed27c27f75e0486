------------------------------- MODULE EpochGc -------------------------------
(***************************************************************************)
(* HAND-WRITTEN translation of specs/pluscal/epoch_gc.tla in the style of   *)
(* pcal2tla (p-manual App. B), written from the ALGORITHM text, not from    *)
(* the product's translation: oracle/tlaplus.py evaluates it, and the       *)
(* product's translation, the compiled program and the GPU must give the    *)
(* same state graph level by level (tests/golden/make_pcal_oracle_golden.py,*)
(* tests/test_pcal.py, tests/test_gpu_zz_channels.py; VERDICT round 5,      *)
(* next 4: goldens that do not come from the product).                      *)
(***************************************************************************)
EXTENDS Naturals, TLC
CONSTANTS N, Grace

VARIABLES epoch, head, state, retEpoch, local, active, pc, p, fresh, old, k

vars == << epoch, head, state, retEpoch, local, active, pc, p, fresh, old, k >>

ProcSet == (1..N)

Nodes == 1..N + 1

Init == /\ epoch = 0
        /\ head = 1
        /\ state = [n \in Nodes |-> IF n = 1 THEN "live" ELSE "free"]
        /\ retEpoch = [n \in Nodes |-> 0]
        /\ local = [t \in 1..N |-> 0]
        /\ active = [t \in 1..N |-> FALSE]
        /\ p = [self \in 1..N |-> 0]
        /\ fresh = [self \in 1..N |-> 0]
        /\ old = [self \in 1..N |-> 0]
        /\ k = [self \in 1..N |-> 1]
        /\ pc = [self \in ProcSet |-> "R1"]

Pin(self, next) == /\ active' = [active EXCEPT ![self] = TRUE]
                   /\ local' = [local EXCEPT ![self] = epoch]
                   /\ pc' = [pc EXCEPT ![self] = next]
                   /\ UNCHANGED << epoch, head, state, retEpoch, p, fresh, old, k >>

R1(self) == pc[self] = "R1" /\ Pin(self, "R2")

R2(self) == /\ pc[self] = "R2"
            /\ p' = [p EXCEPT ![self] = head]
            /\ pc' = [pc EXCEPT ![self] = "R3"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, active, fresh, old, k >>

R3(self) == /\ pc[self] = "R3"
            /\ Assert(state[p[self]] # "free", "Failure of assertion at R3: the reader dereferences freed memory")
            /\ pc' = [pc EXCEPT ![self] = "R4"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, active, p, fresh, old, k >>

R4(self) == /\ pc[self] = "R4"
            /\ active' = [active EXCEPT ![self] = FALSE]
            /\ p' = [p EXCEPT ![self] = 0]
            /\ pc' = [pc EXCEPT ![self] = "W1"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, fresh, old, k >>

W1(self) == pc[self] = "W1" /\ Pin(self, "W2")

W2(self) == /\ pc[self] = "W2"
            /\ \E n \in Nodes :
                 /\ state[n] = "free"
                 /\ fresh' = [fresh EXCEPT ![self] = n]
                 /\ state' = [state EXCEPT ![n] = "live"]
            /\ pc' = [pc EXCEPT ![self] = "W3"]
            /\ UNCHANGED << epoch, head, retEpoch, local, active, p, old, k >>

W3(self) == /\ pc[self] = "W3"
            /\ old' = [old EXCEPT ![self] = head]
            /\ head' = fresh[self]
            /\ pc' = [pc EXCEPT ![self] = "W4"]
            /\ UNCHANGED << epoch, state, retEpoch, local, active, p, fresh, k >>

W4(self) == /\ pc[self] = "W4"
            /\ state' = [state EXCEPT ![old[self]] = "retired"]
            /\ retEpoch' = [retEpoch EXCEPT ![old[self]] = epoch]
            /\ pc' = [pc EXCEPT ![self] = "W5"]
            /\ UNCHANGED << epoch, head, local, active, p, fresh, old, k >>

W5(self) == /\ pc[self] = "W5"
            /\ active' = [active EXCEPT ![self] = FALSE]
            /\ pc' = [pc EXCEPT ![self] = "A1"]
            /\ UNCHANGED << epoch, head, state, retEpoch, local, p, fresh, old, k >>

A1(self) == /\ pc[self] = "A1"
            /\ IF \A t \in 1..N : ~active[t] \/ local[t] = epoch
                  THEN epoch' = (epoch + 1) % 3
                  ELSE epoch' = epoch
            /\ pc' = [pc EXCEPT ![self] = "A2"]
            /\ UNCHANGED << head, state, retEpoch, local, active, p, fresh, old, k >>

A2(self) == /\ pc[self] = "A2"
            /\ IF k[self] <= N + 1
                  THEN /\ IF state[k[self]] = "retired" /\ retEpoch[k[self]] = (epoch + 3 - Grace) % 3
                             THEN state' = [state EXCEPT ![k[self]] = "free"]
                             ELSE state' = state
                       /\ k' = [k EXCEPT ![self] = k[self] + 1]
                       /\ pc' = [pc EXCEPT ![self] = "A2"]
                  ELSE /\ pc' = [pc EXCEPT ![self] = "Done"]
                       /\ UNCHANGED << state, k >>
            /\ UNCHANGED << epoch, head, retEpoch, local, active, p, fresh, old >>

T(self) == R1(self) \/ R2(self) \/ R3(self) \/ R4(self) \/ W1(self) \/ W2(self) \/ W3(self) \/ W4(self) \/ W5(self) \/ A1(self) \/ A2(self)

Next == (\E self \in 1..N: T(self))
           \/ ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

HeadIsLive == state[head] = "live"
NoDanglingReader == \A t \in 1..N : (active[t] /\ p[t] # 0) => state[p[t]] # "free"
EpochInRange == epoch \in 0..2 /\ \A t \in 1..N : local[t] \in 0..2
=============================================================================
