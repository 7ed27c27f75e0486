"""Fingerprint-sharded BFS across the GPUs of one node: one process per GPU, one engine per process,
the seen-set partitioned by fingerprint high bits, buckets moved with
torch.distributed all-to-all (backend "nccl" = RCCL over xGMI).

Per BFS level every rank walks its local frontier in rounds of `chunk_states`; a round is

    expand       frontier chunk -> candidate FINGERPRINTS bucketed by owner        (mc_shard_expand_launch/_finish)
    all-to-all   8 bytes per generated, in-model successor
    probe        owner inserts them in its seen-set slice, answers 1 byte each      (mc_shard_probe)
    all-to-all   answers, reversed
    materialise  sender builds the full state of every "new" answer, by owner       (mc_shard_materialise)
    all-to-all   W bytes per NEW state only  (two-phase: SURVEY.md §7 hard part 5)
    ingest       owner appends them to its next-level frontier                      (mc_shard_ingest)

so states live on the rank that owns their fingerprint (balanced), and xGMI carries
8*(G/D) + 1*(G/D) + W bytes per distinct state instead of (W+8)*(G/D).
Rounds are software-pipelined over two expand slots: while round r's buckets are exchanged, probed and
materialised (engine side stream + the collective stream), round r+1's expand already runs.
Ranks with a shorter frontier take part in every round with empty buckets: the number of rounds
per level is the all-reduced maximum.

This module only orchestrates; all compute is in libtlamc.so (HIP).  Tests drive the same loop
on CPU with gloo and the host build of the lowerings (tests/_shim)."""
import math
import os
import time

import torch
import torch.distributed as dist

from .binding import Engine, Result, VERDICTS, state_action_name, state_apply, state_bytes, state_format

SLOT_NONE, SLOT_INIT, SLOT_PARENT, SLOT_COPY = 0xFFFF, 0xFFFE, 0xFFFD, 0xFFFC  # slot codes of the engine (engine.hip)
NO_PARENT = 0xFFFFFFFF


class HipStepEngine:
    """mc_shard_* of include/tlamc.h on one GPU; buffers are torch CUDA tensors (plumbing only)."""

    def __init__(self, spec, params, device, rank, world, chunk_states, table_capacity, arena_capacity, trace=False):
        self.device = torch.device("cuda", device)
        self.W = state_bytes(spec, params)
        self.eng = Engine(spec, params, device=device, table_capacity=table_capacity, arena_capacity=arena_capacity,
                          chunk_states=chunk_states, trace=trace, timing=False, shard_rank=rank, shard_count=world)
        # the stream the collectives are issued on; the engine enqueues its side work on it without host syncs
        self.stream = torch.cuda.Stream(self.device)
        self.eng.shard_set_stream(self.stream.cuda_stream)

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def begin(self):
        self.eng.shard_begin()

    def begin_replicated(self, min_frontier, max_distinct=0, max_levels=0):
        return self.eng.shard_begin_replicated(min_frontier, max_distinct, max_levels)

    def level_size(self):
        return self.eng.shard_level_size()

    def expand_launch(self, slot, first, count, send_cap):
        self.eng.shard_expand_launch(slot, first, count, send_cap)

    def expand_finish(self, slot, send_fp):
        return self.eng.shard_expand_finish(slot, send_fp.data_ptr(), send_fp.numel())

    def probe(self, recv_fp, n, answers):
        self.eng.shard_probe(recv_fp.data_ptr(), n, answers.data_ptr())

    def materialise(self, slot, answers_back, send_states):
        return self.eng.shard_materialise(answers_back.data_ptr(), send_states.data_ptr(), send_states.numel() // self.W, slot)

    def ingest(self, recv_states, n):
        self.eng.shard_ingest(recv_states.data_ptr(), n)

    def keep(self, slot, answers_back):
        return self.eng.shard_keep(answers_back.data_ptr(), slot)

    # counterexamples across ranks (engine created with trace=True): parents travel with the states that move
    def materialise_parents(self, slot, send_parents):
        self.eng.shard_materialise_parents(slot, send_parents.data_ptr())

    def ingest_parents(self, recv_parents, n, src_rank):
        self.eng.shard_ingest_parents(recv_parents.data_ptr(), n, src_rank)

    def violation(self):
        return self.eng.shard_violation()

    def fetch(self, idx):
        return self.eng.shard_fetch(idx)

    # fixed-capacity rounds (include/tlamc.h mc_shard_*_pack): nothing waits for the host
    def expand_pack(self, slot, send_fp, cap):
        self.eng.shard_expand_pack(slot, send_fp.data_ptr(), cap)

    def probe_pack(self, recv_fp, cap, answers):
        self.eng.shard_probe_pack(recv_fp.data_ptr(), cap, answers.data_ptr())

    def keep_pack(self, slot, answers_back, cap):
        self.eng.shard_keep_pack(slot, answers_back.data_ptr(), cap)

    def end_level(self):
        return self.eng.shard_end_level()

    def counters(self):
        return self.eng.shard_counters()

    def check_frontier(self):
        self.eng.shard_check_frontier()

    def sync(self):
        self.stream.synchronize()

    def close(self):
        self.eng.close()


class ShardedChecker:
    def __init__(self, spec, params, device=0, chunk_states=1 << 19, max_distinct=0, max_levels=0, table_capacity=1 << 27,
                 arena_capacity=1 << 25, fanout_cap=32, new_cap=8, engine=None, group=None, stay_threshold=1 << 16,
                 rebalance_ratio=1.25, replicate_until=1 << 15, packed=True, packed_fanout=16, trace=False):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # with an initialised process group the collectives run even for world_size 1 (exercises the
        # RCCL path on a one-GPU box); without one the checker degenerates to a single local shard
        self.collective = dist.is_initialized()
        self.chunk = chunk_states
        self.max_distinct, self.max_levels = max_distinct, max_levels
        # States MOVE to the owner of their fingerprint while the frontier is small (that is what spreads the
        # single initial state over the ranks) and whenever the ranks' frontiers drift apart; otherwise they
        # STAY where they were generated and only fingerprints (8 B) and answers (1 B) cross xGMI.
        self.stay_threshold, self.rebalance_ratio = stay_threshold, rebalance_ratio
        # The first levels are tiny: every rank runs them itself (same fused BFS everywhere, no collective) until a
        # level has replicate_until states per rank, then keeps the states of it whose fingerprint it owns; 0 = shard
        # from Init on.
        self.replicate_until = replicate_until
        # "stay" rounds as fixed-capacity exchanges with in-band counts: the host only enqueues (no size all-to-all, no
        # device-to-host copy inside a round); packed=False keeps the host-paced rounds (variable-size all-to-alls) for A/B
        # packed_fanout: in-model successors per expanded state the fixed buckets allow for (the buckets are moved and scanned
        # whole, so the allowance is tighter than fanout_cap, which only sizes buffers; a level that exceeds it fails loudly)
        self.packed, self.packed_fanout = packed, min(packed_fanout, fanout_cap)
        # trace: every state keeps (rank, index, slot) of its parent — a state that moves to its owner takes them along — so that
        # a counterexample is walked back ACROSS ranks (counterexample()); costs 7 bytes per state and one more all-to-all per
        # moving round
        self.spec, self.params, self.trace = spec, list(params), trace
        self.eng = engine if engine is not None else HipStepEngine(spec, params, device, self.rank, self.world, chunk_states,
                                                                    table_capacity, arena_capacity, trace=trace)
        self.dev = self.eng.device
        self.W = self.eng.W
        backend = dist.get_backend(group) if dist.is_initialized() else None
        # gloo moves CPU tensors only: stage device buffers through the host for it
        self.comm_dev = torch.device("cpu") if backend == "gloo" else self.dev
        # two expand slots: the expand of round r+1 runs while round r is exchanged, probed and kept
        self.send_fp = [torch.empty(chunk_states * fanout_cap, dtype=torch.int64, device=self.dev) for _ in range(2)]
        self.send_states = torch.empty(chunk_states * new_cap * self.W, dtype=torch.uint8, device=self.dev)
        self.send_parents = torch.empty(chunk_states * new_cap if trace else 1, dtype=torch.int64, device=self.dev)

    # ---------------------------------------------------------------- collectives
    def _allreduce(self, value, op):
        if not self.collective:
            return int(value)
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.comm_dev)
        dist.all_reduce(t, op=op, group=self.group)
        return int(t.item())

    def _a2a(self, send, send_counts, elem):
        """all-to-all of `send` (first sum(counts)*elem elements, bucketed by destination).
        Returns (recv tensor on the engine's device, recv_counts)."""
        P = self.world
        if not self.collective:
            n = send_counts[0] * elem
            return send[:n], list(send_counts)
        sc = torch.tensor(send_counts, dtype=torch.int64, device=self.comm_dev)
        rc = torch.empty(P, dtype=torch.int64, device=self.comm_dev)
        dist.all_to_all_single(rc, sc, group=self.group)
        recv_counts = [int(x) for x in rc.tolist()]
        src = send[: sum(send_counts) * elem].to(self.comm_dev)
        recv = torch.empty(sum(recv_counts) * elem, dtype=send.dtype, device=self.comm_dev)
        dist.all_to_all_single(recv, src, [c * elem for c in recv_counts], [c * elem for c in send_counts], group=self.group)
        return recv.to(self.dev), recv_counts

    def _exchange_counts(self, send_counts):
        if not self.collective:
            return list(send_counts)
        sc = torch.tensor(send_counts, dtype=torch.int64, device=self.comm_dev)
        rc = torch.empty(self.world, dtype=torch.int64, device=self.comm_dev)
        dist.all_to_all_single(rc, sc, group=self.group)
        return [int(x) for x in rc.tolist()]

    def _a2a_back(self, send, send_counts, recv_counts):
        """reverse direction with known sizes (answers travel back along the fingerprints' path)"""
        if not self.collective:
            return send[: send_counts[0]]
        src = send[: sum(send_counts)].to(self.comm_dev)
        recv = torch.empty(sum(recv_counts), dtype=send.dtype, device=self.comm_dev)
        dist.all_to_all_single(recv, src, list(recv_counts), list(send_counts), group=self.group)
        return recv.to(self.dev)

    def _a2a_equal(self, send, n):
        """equal-split all-to-all of the first n elements of `send` (n a multiple of the world size): fixed-capacity buckets"""
        if not self.collective:
            return send[:n]
        src = send[:n].to(self.comm_dev)
        recv = torch.empty(n, dtype=send.dtype, device=self.comm_dev)
        dist.all_to_all_single(recv, src, group=self.group)
        return recv.to(self.dev)

    def _level_info(self, local_n, verdict):
        """ONE collective per level: every rank learns every rank's frontier size and the worst verdict."""
        if not self.collective:
            return [int(local_n)], int(verdict)
        t = torch.tensor([int(local_n), int(verdict)], dtype=torch.int64, device=self.comm_dev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        rows = [o.tolist() for o in out]
        return [int(r[0]) for r in rows], max(int(r[1]) for r in rows)

    # ---------------------------------------------------------------- BFS
    def run(self):
        with self.eng.stream_ctx():
            return self._run()

    def _run(self):
        e, SUM, MAX = self.eng, dist.ReduceOp.SUM, dist.ReduceOp.MAX
        prof = bool(os.environ.get("TLAMC_PHASES"))     # per-phase wall clock: adds host syncs, off by default
        self.phase_s = ph = {}

        def tick(key, t_prev):
            if not prof:
                return t_prev
            e.sync()
            now = time.perf_counter()
            ph[key] = ph.get(key, 0.0) + now - t_prev
            return now

        if self.replicate_until:
            levels = e.begin_replicated(self.replicate_until * self.world, self.max_distinct, self.max_levels)
            sizes, verdict = self._level_info(e.level_size(), e.counters()[2])
            frontier = sum(sizes) if verdict == 0 else levels[-1]
            cum, level, budget = sum(levels), len(levels), False
            ph["replicated_levels"] = len(levels)
        else:
            e.begin()
            sizes, verdict = self._level_info(e.level_size(), e.counters()[2])
            frontier = sum(sizes)
            levels, cum, level, budget = [frontier], frontier, 1, False
        while frontier > 0:
            if verdict != 0:
                break
            if (self.max_levels and level >= self.max_levels) or (self.max_distinct and cum >= self.max_distinct):
                budget = True
                break
            local_n = sizes[self.rank] if self.collective else sizes[0]
            rounds = max(math.ceil(n / self.chunk) for n in sizes)
            stay = frontier >= self.stay_threshold * self.world and max(sizes) * self.world <= self.rebalance_ratio * frontier
            ph["stay_levels" if stay else "move_levels"] = ph.get("stay_levels" if stay else "move_levels", 0) + 1
            ph["rounds"] = ph.get("rounds", 0) + rounds

            def launch(r):
                first = min(r * self.chunk, local_n)
                e.expand_launch(r & 1, first, min(self.chunk, local_n - first), self.send_fp[r & 1].numel())

            hold = []
            if rounds:
                launch(0)
            if stay and self.packed:
                # every rank derives the same capacity from the level's frontier sizes: the largest chunk of the round times the
                # fan-out allowance, split over the owners (+ slack and the count word)
                P = self.world
                for r in range(rounds):
                    slot = r & 1
                    n_round = max(min(self.chunk, max(n - r * self.chunk, 0)) for n in sizes)
                    # a rank routes (P - 1) / P of its candidates, spread over P owners (its own share is probed locally)
                    cap = min((n_round * self.packed_fanout * (P - 1)) // (P * P) + 1024, self.send_fp[slot].numel() // P)
                    t = time.perf_counter()
                    e.expand_pack(slot, self.send_fp[slot], cap)      # enqueued behind expand r: the host does not wait
                    if r + 1 < rounds:
                        launch(r + 1)
                    recv_fp = self._a2a_equal(self.send_fp[slot], P * cap)
                    t = tick("a2a_fp", t)
                    answers = torch.empty(P * cap, dtype=torch.uint8, device=self.dev)
                    e.probe_pack(recv_fp, cap, answers)
                    t = tick("probe", t)
                    back = self._a2a_equal(answers, P * cap)
                    t = tick("a2a_ans", t)
                    e.keep_pack(slot, back, cap)
                    hold += [recv_fp, answers, back]                  # inputs of kernels still queued: freed after end_level
                    t = tick("keep", t)
                rounds = 0
            for r in range(rounds):
                slot = r & 1
                t = time.perf_counter()
                counts = e.expand_finish(slot, self.send_fp[slot])     # waits for expand r only
                if r + 1 < rounds:
                    launch(r + 1)                                       # overlaps everything below
                t = tick("expand_wait", t)
                recv_fp, rcounts = self._a2a(self.send_fp[slot], counts, 1)
                n = sum(rcounts)
                answers = torch.empty(max(n, 1), dtype=torch.uint8, device=self.dev)
                t = tick("a2a_fp", t)
                e.probe(recv_fp, n, answers)
                t = tick("probe", t)
                back = self._a2a_back(answers, rcounts, counts)
                t = tick("a2a_ans", t)
                if stay:
                    e.keep(slot, back)      # runs on the engine's own stream, behind this point of ours
                    hold.append(back)       # ... so its input must outlive this round (freed after end_level)
                    t = tick("keep", t)
                    continue
                scounts = e.materialise(slot, back, self.send_states)
                t = tick("materialise", t)
                # full states travel as whole 64-state blocks per owner (coalesced at both ends)
                blocks = [(c + 63) // 64 for c in scounts]
                recv_states, rblocks = self._a2a(self.send_states, blocks, 64 * self.W)
                rsc = self._exchange_counts(scounts)
                recv_parents = None
                if self.trace:   # (index on the sending rank << 16 | slot) of every moved state, same owner order, no block padding
                    e.materialise_parents(slot, self.send_parents)
                    recv_parents, _ = self._a2a(self.send_parents, scounts, 1)
                t = tick("a2a_states", t)
                off = poff = 0
                for src_rank in range(len(rsc)):            # one bucket per source rank
                    if rsc[src_rank]:
                        e.ingest(recv_states[off * 64 * self.W:], rsc[src_rank])
                        if recv_parents is not None:
                            e.ingest_parents(recv_parents[poff:], rsc[src_rank], src_rank)
                    off += rblocks[src_rank]
                    poff += rsc[src_rank]
                t = tick("ingest", t)
            new_local = e.end_level()                       # waits for the engine's streams; arena fill level comes back
            hold.clear()
            sizes, verdict = self._level_info(new_local, e.counters()[2])
            frontier = sum(sizes)
            if frontier > 0:
                level += 1
                levels.append(frontier)
                cum += frontier
        if frontier > 0 and verdict == 0:
            e.check_frontier()  # a budget stop leaves a level unexpanded: its check-on-expand invariants (SI models) are due
        generated, _, verdict = e.counters()
        generated = self._allreduce(generated, SUM)
        verdict = self._allreduce(verdict, MAX)
        if verdict == 0 and budget:
            verdict = 5
        if prof and self.rank == 0:
            print("phases[s]:", {k: round(v, 4) for k, v in ph.items()}, flush=True)
        return Result(distinct=cum, generated=generated, queue_left=frontier, depth=level, verdict=VERDICTS[verdict],
                      violated_invariant=-1, trace_len=0, levels=levels, seconds=0.0)

    # ---------------------------------------------------------------- counterexample
    def counterexample(self):
        """After a run that ended in a violation (trace=True): the behaviour that leads to it, walked back parent by parent
        ACROSS ranks — the rank that holds a state looks it up and tells the others where its parent lives (one small
        broadcast per step; a counterexample has tens of states).  Collective: every rank calls it and gets the same list
        of (action name, TLA+ text of the state), first the initial state.  None when no rank found a violation."""
        e = self.eng
        mine = e.violation()   # (found, idx, slot, verdict, invariant)
        if self.collective:
            every = [None] * self.world
            dist.all_gather_object(every, mine, group=self.group)
        else:
            every = [mine]
        owners = [r for r, v in enumerate(every) if v[0]]
        if not owners:
            return None
        owner = owners[0]
        _, idx, vslot, _verdict, _inv = every[owner]
        steps = []           # (packed state, slot that produced it), last state first
        cur_rank, cur_idx = owner, idx
        for _ in range(1 << 16):
            obj = [e.fetch(cur_idx) if self.rank == cur_rank else None]
            if self.collective:
                src = dist.get_global_rank(self.group, cur_rank) if self.group is not None else cur_rank
                dist.broadcast_object_list(obj, src=src, group=self.group)
            state, prank, pidx, pslot = obj[0]
            if pslot == SLOT_COPY:      # the replicated prefix copied the state into this rank's slice: not a step
                cur_idx = pidx
                continue
            steps.append((state, pslot))
            if pidx == NO_PARENT:
                break
            cur_rank, cur_idx = prank, pidx
        steps.reverse()
        out = []
        for k, (state, pslot) in enumerate(steps):
            name = "Initial predicate" if k == 0 else state_action_name(self.spec, self.params, steps[k - 1][0], pslot)
            out.append((name, state_format(self.spec, self.params, state)))
        # an invariant violated by a SUCCESSOR: that state is not stored anywhere, it is rebuilt from its parent.  (A failed
        # Assert / an evaluation error has no successor: TLC's behaviour ends at the state the action was taken from.)
        if _verdict == "invariant" and vslot not in (SLOT_NONE, SLOT_PARENT, SLOT_INIT):
            last = steps[-1][0]
            out.append((state_action_name(self.spec, self.params, last, vslot),
                        state_format(self.spec, self.params, state_apply(self.spec, self.params, last, vslot))))
        return out

    def close(self):
        self.eng.close()
