"""Fingerprint-sharded BFS across the GPUs of one node, driven from Python: one process per GPU, one engine per process, the
seen-set partitioned by fingerprint high bits.

The level loop itself is NOT here any more: it is the C++ loop of the library (tla_rust_amd/csrc/shard_loop.h — replicated
prefix, "stay" rounds in three forms (exact sizes by default; `exchange=`), "move" rounds that rebalance, collective error status, counterexamples walked
back across ranks), the same one `mc X.tla -gpus P` and `bench.py --gpus N` run over RCCL directly (mc_comm_* / mc_shard_run).
This module is the "bring your own collectives" door of that loop (include/tlamc.h `mc_transport`): `TorchTransport` hands
torch.distributed's collectives to it as callbacks —

    backend "nccl"  = RCCL over xGMI, ordered on one torch stream (the loop makes that stream wait for its producers with events);
    backend "gloo"  = staged through the host; what the CPU tests use (with the host build of the lowerings, tests/_shim) and
                      what lets several ranks share ONE GPU on a one-GPU box;
    no process group = a single local shard (world 1): the step kernels of the sharded path against the fused run.

Per BFS level the loop exchanges, per generated in-model successor, 8 bytes of fingerprint to the owner of its seen-set slice
and 1 byte of answer back; new states stay on the rank that generated them (large balanced levels) or travel to their owner as
whole 64-state blocks (small or drifted levels): 9 * (G/D) [+ W] bytes per distinct state instead of (W + 8) * (G/D) for naive
owner-routed successors (SURVEY.md §7 hard part 5)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import binding as B
from .binding import Engine, Result

MC_ERCCL = -6


class TorchTransport:
    """mc_transport over torch.distributed (or over nothing: world 1).  `device`: where the exchange buffers live — the engine's
    GPU, or the CPU for the host build of the lowerings."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.on = dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.backend = dist.get_backend(group) if self.on else None
        self.cuda = self.device.type == "cuda"
        # gloo moves CPU tensors only: device buffers are staged through the host, synchronously
        self.staged = self.cuda and self.backend != "nccl"
        self.stream = torch.cuda.Stream(self.device) if self.cuda and self.backend == "nccl" else None
        self.bufs = {}
        self.error = None
        self._cb = (B.T_ALLOC(self._alloc), B.T_RELEASE(self._release), B.T_A2A(self._a2a), B.T_A2AV(self._a2av), B.T_GATHER(self._gather))
        self.c = B.Transport(None, self.rank, self.world, self.stream.cuda_stream if self.stream is not None else None, *self._cb)

    # ---- buffers: torch tensors, addressed by the loop through their base pointers
    def _alloc(self, _user, nbytes):
        try:
            t = torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=self.device)
            self.bufs[t.data_ptr()] = t
            return t.data_ptr()
        except Exception as e:  # noqa: BLE001 — an exception must not cross the C boundary
            self.error = e
            return None

    def _release(self, _user, p):
        if self.cuda:
            torch.cuda.synchronize(self.device)   # like hipFree: nothing enqueued still uses the block
        self.bufs.pop(p, None)

    def _view(self, p, nbytes):
        t = self.bufs.get(p)
        if t is not None:
            return t[:nbytes]
        for base, t in self.bufs.items():   # a pointer into a buffer
            if base <= p and p + nbytes <= base + t.numel():
                return t[p - base: p - base + nbytes]
        raise RuntimeError("transport: pointer outside the exchange buffers")

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return MC_ERCCL

    def _exchange(self, out, inp, osplit=None, isplit=None):
        if not self.on:
            out.copy_(inp)
            if self.cuda:   # no hip_stream was handed to the loop: it takes the exchange as COMPLETE when this returns, and its
                torch.cuda.synchronize(self.device)   # consumers run on other (non-blocking) streams — the copy must have landed
        elif self.staged:   # the loop made the host wait for the producers (no hip_stream): plain blocking copies
            src = inp.cpu()
            dst = torch.empty(out.numel(), dtype=torch.uint8)
            dist.all_to_all_single(dst, src, osplit, isplit, group=self.group)
            out.copy_(dst)
            torch.cuda.synchronize(self.device)
        elif self.stream is not None:
            with torch.cuda.stream(self.stream):
                dist.all_to_all_single(out, inp, osplit, isplit, group=self.group)
        else:
            dist.all_to_all_single(out, inp, osplit, isplit, group=self.group)

    def _a2a(self, _user, send, recv, nbytes):
        n = int(nbytes) * self.world
        return self._guard(lambda: n and self._exchange(self._view(recv, n), self._view(send, n)))

    def _a2av(self, _user, send, so, sb, recv, ro, rb):
        def go():
            P = self.world
            sb_, rb_ = [int(sb[p]) for p in range(P)], [int(rb[p]) for p in range(P)]
            assert [int(so[p]) for p in range(P)] == [sum(sb_[:p]) for p in range(P)] and [int(ro[p]) for p in range(P)] == [sum(rb_[:p]) for p in range(P)]
            ns, nr = sum(sb_), sum(rb_)
            inp = self._view(send, ns) if ns else torch.empty(0, dtype=torch.uint8, device=self.device)
            out = self._view(recv, nr) if nr else torch.empty(0, dtype=torch.uint8, device=self.device)
            self._exchange(out, inp, rb_, sb_)
        return self._guard(go)

    def _gather(self, _user, mine, all_out, nbytes):
        def go():
            n = int(nbytes)
            if not self.on:
                C.memmove(all_out, mine, n)
                return
            src = torch.frombuffer((C.c_uint8 * n).from_address(mine), dtype=torch.uint8).clone()
            if self.backend == "nccl":
                with torch.cuda.stream(self.stream):
                    d_src = src.to(self.device)
                    d_all = torch.empty(n * self.world, dtype=torch.uint8, device=self.device)
                    dist.all_gather_into_tensor(d_all, d_src, group=self.group)
                    out = d_all.cpu()
                self.stream.synchronize()
            else:
                parts = [torch.empty(n, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, src, group=self.group)
                out = torch.cat(parts)
            C.memmove(all_out, out.numpy().ctypes.data, n * self.world)
        return self._guard(go)

    def all_gather_object(self, obj):
        if not self.on:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out


class HipShard:
    """one rank's HIP engine behind the loop (mc_shard_run_transport / mc_shard_trace_transport of the C ABI)"""

    def __init__(self, spec, params, device, rank, world, chunk_states, table_capacity, arena_capacity, trace=False):
        self.spec, self.params = spec, list(params)
        self.device = torch.device("cuda", device)
        self.W = B.state_bytes(spec, params)
        self.eng = Engine(spec, params, device=device, table_capacity=table_capacity, arena_capacity=arena_capacity,
                          chunk_states=chunk_states, trace=trace, timing=False, shard_rank=rank, shard_count=world)

    def run_transport(self, t, opts, res):
        return B.lib().mc_shard_run_transport(self.eng._h, C.byref(t), C.byref(opts), C.byref(res))

    def trace_transport(self, t, states, slots, n, final):
        return B.lib().mc_shard_trace_transport(self.eng._h, C.byref(t), states, slots, n, final)

    def checkpoint(self, path):
        return B.lib().mc_shard_checkpoint(self.eng._h, str(path).encode())

    def restore(self, path):
        return B.lib().mc_shard_restore(self.eng._h, str(path).encode())

    def check(self, rc, what):
        B._check(rc, what)

    def format(self, st):
        return B.state_format(self.spec, self.params, st)

    def action_name(self, st, slot):
        return B.state_action_name(self.spec, self.params, st, slot)

    def apply(self, st, slot):
        return B.state_apply(self.spec, self.params, st, slot)

    def close(self):
        self.eng.close()


class ShardedChecker:
    """One rank of the sharded search.  run() -> Result (the global counters on every rank); counterexample() after a violation."""

    def __init__(self, spec, params, device=0, chunk_states=1 << 19, max_distinct=0, max_levels=0, table_capacity=1 << 27,
                 arena_capacity=1 << 25, engine=None, group=None, stay_threshold=1 << 16, rebalance_ratio=1.25, replicate_until=1 << 15,
                 packed_fanout=16, move_fanout=32, trace=False, exchange="exact", cap_safety_pct=0):
        self.spec, self.params = spec, list(params)
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        # States MOVE to the owner of their fingerprint while the frontier is small (that is what spreads a few states over
        # the ranks) and whenever the ranks' frontiers drift apart (rebalance_ratio); otherwise they STAY where they were
        # generated and only fingerprints and answers cross xGMI.  replicate_until: states per rank a level needs before the
        # search is sharded at all (below it every rank runs the same fused BFS); 0 = shard from Init on.
        self.opts = dict(chunk_states=chunk_states, max_distinct=max_distinct, max_levels=max_levels, replicate_until=replicate_until,
                         packed_fanout=packed_fanout, stay_threshold=stay_threshold, rebalance_ratio=rebalance_ratio, move_fanout=move_fanout,
                         exchange=exchange, cap_safety_pct=cap_safety_pct)
        self.eng = engine if engine is not None else HipShard(spec, params, device, rank, world, chunk_states, table_capacity,
                                                              arena_capacity, trace=trace)
        self.net = TorchTransport(self.eng.device, group)
        self.rank, self.world = self.net.rank, self.net.world
        self.W = self.eng.W
        self.stats = {}

    def run(self):
        st = B.ShardStats()
        o = B.shard_opts(stats=st, **self.opts)
        r = B.CResult()
        rc = self.eng.run_transport(self.net.c, o, r)
        if self.net.error is not None:
            raise self.net.error
        self.eng.check(rc, "mc_shard_run_transport")
        self.stats = {k: int(getattr(st, k)) for k, _ in B.ShardStats._fields_}
        return B._result(r)

    def checkpoint(self, path):
        """this rank's share of a run that ended without an error (normally on a budget): one file per rank (mc_shard_checkpoint)"""
        self.eng.check(self.eng.checkpoint(path), "mc_shard_checkpoint")

    def restore(self, path, max_distinct=None, max_levels=None):
        """load this rank's file of a checkpointed run; the next run() continues it (every rank must restore the same run)"""
        self.eng.check(self.eng.restore(path), "mc_shard_restore")
        if max_distinct is not None:
            self.opts["max_distinct"] = max_distinct
        if max_levels is not None:
            self.opts["max_levels"] = max_levels

    @property
    def local_distinct(self):
        return self.stats.get("distinct_local", 0)

    def counterexample(self):
        """After a run that ended in a violation (trace=True): the behaviour that leads to it, walked back parent by parent
        ACROSS ranks (mc_shard_trace_transport: the rank that holds a state looks it up, one small all-gather per step).
        Collective: every rank calls it and gets the same list of (action name, TLA+ text of the state), first the initial
        state.  None when no rank found a violation."""
        e = self.eng
        out = B._trace_from(lambda st, sl, n, fs: e.trace_transport(self.net.c, st, sl, n, fs), "mc_shard_trace_transport", self.W,
                            e.format, e.action_name, e.apply) if self.net.error is None else None
        if self.net.error is not None:
            raise self.net.error
        return out

    def close(self):
        self.eng.close()
