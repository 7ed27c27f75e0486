---- MODULE proc_demo ----
EXTENDS Naturals, Sequences
(* --algorithm ProcDemo
variables total = 0;
procedure add(n)
  variables t = 0;
begin
  A1: t := total;
  A2: total := t + n;
      return;
end procedure;
process p \in 1..2
begin
  P1: call add(self);
  P2: call add(10);
  P3: skip;
end process;
end algorithm *)
====
