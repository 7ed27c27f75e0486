---- MODULE treiber_procs ----
(* A Treiber stack (the lock-free stack of the reference's roadmap, README.md:26-42) written with PlusCal PROCEDURES: push and pop are
   procedures with their own variables, called by every worker; the compare-and-swap on `head` is one atomic step each.  Node i belongs
   to worker i, 0 is the null pointer.  mc expands the procedures into the workers (tla_rust_amd/csrc/pcal.h); the state graph is the
   one of pcal2tla's stack translation (tests/golden/pcal_procedures/TreiberStack.tla, written by hand). *)
EXTENDS Naturals, Sequences
CONSTANT N
(* --algorithm treiber_procs
variables head = 0, nxt = [i \in 1..N |-> 0], got = [i \in 1..N |-> 0];
procedure push(node)
  variables old = 0;
begin
  PU1: old := head;
  PU2: nxt[node] := old;
  PU3: if head = old then
           head := node;
           return;
       else
           goto PU1;
       end if;
end procedure;
procedure pop()
  variables top = 0, nx = 0;
begin
  PO1: top := head;
  PO2: if top = 0 then
           return;
       end if;
  PO3: nx := nxt[top];
  PO4: if head = top then
           head := nx;
           got[self] := top;
           return;
       else
           goto PO1;
       end if;
end procedure;
process w \in 1..N
begin
  W1: call push(self);
  W2: call pop();
  W3: skip;
end process;
end algorithm *)

PopsDistinct == \A i \in 1..N : \A j \in 1..N : (i # j /\ got[i] # 0) => got[i] # got[j]
====
