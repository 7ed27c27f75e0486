--------------------------- MODULE treiber_records ---------------------------
(* Treiber's lock-free stack with the two things the pointer version (treiber_stack.tla) leaves out: the nodes are RECORDS
   (value, next) in a memory array, and the head is a record (pointer, version) that both CAS loops compare and replace as
   a whole — the version tag is what defeats ABA.  Every worker pushes its own node and then pops one.
   The lock-free stack of the reference's roadmap (README.md:26-42); records: p-manual section 3.1 / 5.4. *)
EXTENDS Naturals, TLC
CONSTANT N

(* --algorithm treiber_records
variables
  mem = [a \in 1..N |-> [val |-> 0, next |-> 0]],
  top = [ptr |-> 0, ver |-> 0],
  popped = [p \in 1..N |-> 0];

process worker \in 1..N
variables old = [ptr |-> 0, ver |-> 0], nxt = 0;
begin
  Fill:     mem[self].val := 10 * self;
  PushRead: old := top;
  PushLink: mem[self].next := old.ptr;
  PushCas:  if top = old then
              top := [ptr |-> self, ver |-> old.ver + 1];
            else
              goto PushRead;
            end if;
  PopRead:  old := top;
            assert old.ptr # 0;
  PopNext:  nxt := mem[old.ptr].next;
  PopCas:   if top = old then
              top := [ptr |-> nxt, ver |-> old.ver + 1] || popped[self] := mem[old.ptr].val;
            else
              goto PopRead;
            end if;
end process
end algorithm *)
\* BEGIN TRANSLATION
VARIABLES mem_val, mem_next, top_ptr, top_ver, popped, pc, old_ptr, old_ver, nxt

vars == << mem_val, mem_next, top_ptr, top_ver, popped, pc, old_ptr, old_ver, nxt >>

(* record variables are kept field by field: r.f is r_f *)
mem == [a \in 1..N |-> [val |-> mem_val[a], next |-> mem_next[a]]]
top == [ptr |-> top_ptr, ver |-> top_ver]
old == [self \in 1..N |-> [ptr |-> old_ptr[self], ver |-> old_ver[self]]]

ProcSet == (1..N)

Init == (* Global variables *)
        /\ mem_val = [a \in 1..N |-> 0]
        /\ mem_next = [a \in 1..N |-> 0]
        /\ top_ptr = 0
        /\ top_ver = 0
        /\ popped = [p \in 1..N |-> 0]
        (* Process worker *)
        /\ old_ptr = [self \in 1..N |-> 0]
        /\ old_ver = [self \in 1..N |-> 0]
        /\ nxt = [self \in 1..N |-> 0]
        /\ pc = [self \in ProcSet |-> "Fill"]

Fill(self) == /\ pc[self] = "Fill"
              /\ mem_val' = [mem_val EXCEPT ![self] = 10 * self]
              /\ pc' = [pc EXCEPT ![self] = "PushRead"]
              /\ UNCHANGED << mem_next, top_ptr, top_ver, popped, old_ptr, 
                              old_ver, nxt >>

PushRead(self) == /\ pc[self] = "PushRead"
                  /\ old_ptr' = [old_ptr EXCEPT ![self] = top_ptr]
                  /\ old_ver' = [old_ver EXCEPT ![self] = top_ver]
                  /\ pc' = [pc EXCEPT ![self] = "PushLink"]
                  /\ UNCHANGED << mem_val, mem_next, top_ptr, top_ver, 
                                  popped, nxt >>

PushLink(self) == /\ pc[self] = "PushLink"
                  /\ mem_next' = [mem_next EXCEPT ![self] = old_ptr[self]]
                  /\ pc' = [pc EXCEPT ![self] = "PushCas"]
                  /\ UNCHANGED << mem_val, top_ptr, top_ver, popped, old_ptr, 
                                  old_ver, nxt >>

PushCas(self) == /\ pc[self] = "PushCas"
                 /\ IF (top_ptr = old_ptr[self] /\ top_ver = old_ver[self])
                       THEN /\ top_ptr' = self
                            /\ top_ver' = old_ver[self] + 1
                            /\ pc' = [pc EXCEPT ![self] = "PopRead"]
                       ELSE /\ pc' = [pc EXCEPT ![self] = "PushRead"]
                            /\ UNCHANGED << top_ptr, top_ver >>
                 /\ UNCHANGED << mem_val, mem_next, popped, old_ptr, old_ver, 
                                 nxt >>

PopRead(self) == /\ pc[self] = "PopRead"
                 /\ old_ptr' = [old_ptr EXCEPT ![self] = top_ptr]
                 /\ old_ver' = [old_ver EXCEPT ![self] = top_ver]
                 /\ Assert(old_ptr'[self] # 0, 
                           "Failure of assertion at line 27, column 13.")
                 /\ pc' = [pc EXCEPT ![self] = "PopNext"]
                 /\ UNCHANGED << mem_val, mem_next, top_ptr, top_ver, popped, 
                                 nxt >>

PopNext(self) == /\ pc[self] = "PopNext"
                 /\ nxt' = [nxt EXCEPT ![self] = mem_next[old_ptr[self]]]
                 /\ pc' = [pc EXCEPT ![self] = "PopCas"]
                 /\ UNCHANGED << mem_val, mem_next, top_ptr, top_ver, popped, 
                                 old_ptr, old_ver >>

PopCas(self) == /\ pc[self] = "PopCas"
                /\ IF (top_ptr = old_ptr[self] /\ top_ver = old_ver[self])
                      THEN /\ top_ptr' = nxt[self]
                           /\ top_ver' = old_ver[self] + 1
                           /\ popped' = [popped EXCEPT ![self] = mem_val[old_ptr[self]]]
                           /\ pc' = [pc EXCEPT ![self] = "Done"]
                      ELSE /\ pc' = [pc EXCEPT ![self] = "PopRead"]
                           /\ UNCHANGED << top_ptr, top_ver, popped >>
                /\ UNCHANGED << mem_val, mem_next, old_ptr, old_ver, nxt >>

worker(self) == Fill(self) \/ PushRead(self) \/ PushLink(self) \/ PushCas(self) \/ PopRead(self) \/ PopNext(self) \/ PopCas(self)

Next == (\E self \in 1..N: worker(self))
           \/ (* Disjunct to prevent deadlock on termination *)
              ((\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars)

Spec == Init /\ [][Next]_vars

Termination == <>(\A self \in ProcSet: pc[self] = "Done")

\* END TRANSLATION

PoppedOnce == \A p \in 1..N : \A q \in 1..N : (p # q /\ popped[p] # 0) => popped[p] # popped[q]
TopIsNode == top.ptr \in 0..N /\ top.ver <= 2 * N
NextIsNode == \A a \in 1..N : mem[a].next \in 0..N /\ mem[a].next # a
OldIsNode == \A p \in 1..N : old[p].ptr \in 0..N
=============================================================================
