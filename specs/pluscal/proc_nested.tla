---- MODULE proc_nested ----
(* PlusCal procedures in the c-syntax: a procedure that calls another (two call sites), a single process and a process set calling the
   same procedures (the expansion gives each process its own copy of the procedures' variables: by_a / by_b ...). *)
EXTENDS Naturals, Sequences
(* --algorithm proc_nested {
variables x = 0, log = 0;
procedure inc(by) {
  I1: x := x + by;
      return;
}
procedure twice(k)
  variables saved = 0;
{
  T1: saved := x;
      call inc(k);
  T2: call inc(k);
  T3: log := log + (x - saved);
      return;
}
process (a = 1) {
  A1: call twice(1);
  A2: call inc(5);
}
process (b \in {2, 3}) {
  B1: call twice(self);
}
} *)

XBound == x <= 17
Final == (\A p \in {1, 2, 3} : pc[p] = "Done") => x = 17
====
