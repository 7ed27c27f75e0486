"""Fingerprint-sharded BFS across the GPUs of one node: one process per GPU, one engine per process,
the seen-set partitioned by fingerprint high bits, buckets moved with
torch.distributed all-to-all (backend "nccl" = RCCL over xGMI).

Per BFS level every rank walks its local frontier in rounds of `chunk_states`; a round is

    expand       frontier chunk -> candidate FINGERPRINTS bucketed by owner        (mc_shard_expand)
    all-to-all   8 bytes per generated, in-model successor
    probe        owner inserts them in its seen-set slice, answers 1 byte each      (mc_shard_probe)
    all-to-all   answers, reversed
    materialise  sender builds the full state of every "new" answer, by owner       (mc_shard_materialise)
    all-to-all   W bytes per NEW state only  (two-phase: SURVEY.md §7 hard part 5)
    ingest       owner appends them to its next-level frontier                      (mc_shard_ingest)

so states live on the rank that owns their fingerprint (balanced), and xGMI carries
8*(G/D) + 1*(G/D) + W bytes per distinct state instead of (W+8)*(G/D).
Ranks with a shorter frontier take part in every round with empty buckets: the number of rounds
per level is the all-reduced maximum.

This module only orchestrates; all compute is in libtlamc.so (HIP).  Tests drive the same loop
on CPU with gloo and the host build of the lowerings (tests/_shim)."""
import math
import os
import time

import torch
import torch.distributed as dist

from .binding import Engine, Result, VERDICTS, state_bytes


class HipStepEngine:
    """mc_shard_* of include/tlamc.h on one GPU; buffers are torch CUDA tensors (plumbing only)."""

    def __init__(self, spec, params, device, rank, world, chunk_states, table_capacity, arena_capacity):
        self.device = torch.device("cuda", device)
        self.W = state_bytes(spec, params)
        self.eng = Engine(spec, params, device=device, table_capacity=table_capacity, arena_capacity=arena_capacity,
                          chunk_states=chunk_states, trace=False, timing=False, shard_rank=rank, shard_count=world)

    def begin(self):
        self.eng.shard_begin()

    def level_size(self):
        return self.eng.shard_level_size()

    def expand(self, first, count, send_fp):
        return self.eng.shard_expand(first, count, send_fp.data_ptr(), send_fp.numel())

    def probe(self, recv_fp, n, answers):
        self.eng.shard_probe(recv_fp.data_ptr(), n, answers.data_ptr())

    def materialise(self, answers_back, send_states):
        return self.eng.shard_materialise(answers_back.data_ptr(), send_states.data_ptr(), send_states.numel() // self.W)

    def ingest(self, recv_states, n):
        self.eng.shard_ingest(recv_states.data_ptr(), n)

    def keep(self, answers_back):
        return self.eng.shard_keep(answers_back.data_ptr())

    def end_level(self):
        return self.eng.shard_end_level()

    def counters(self):
        return self.eng.shard_counters()

    def sync(self):
        torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        self.eng.close()


class ShardedChecker:
    def __init__(self, spec, params, device=0, chunk_states=1 << 19, max_distinct=0, max_levels=0, table_capacity=1 << 27,
                 arena_capacity=1 << 25, fanout_cap=32, new_cap=8, engine=None, group=None, stay_threshold=1 << 16,
                 rebalance_ratio=1.25):
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
        self.eng = engine if engine is not None else HipStepEngine(spec, params, device, self.rank, self.world, chunk_states,
                                                                    table_capacity, arena_capacity)
        self.dev = self.eng.device
        self.W = self.eng.W
        backend = dist.get_backend(group) if dist.is_initialized() else None
        # gloo moves CPU tensors only: stage device buffers through the host for it
        self.comm_dev = torch.device("cpu") if backend == "gloo" else self.dev
        self.send_fp = torch.empty(chunk_states * fanout_cap, dtype=torch.int64, device=self.dev)
        self.send_states = torch.empty(chunk_states * new_cap * self.W, dtype=torch.uint8, device=self.dev)

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

    # ---------------------------------------------------------------- BFS
    def run(self):
        e, SUM, MAX = self.eng, dist.ReduceOp.SUM, dist.ReduceOp.MAX
        self.phase_s = {}   # wall seconds per phase of the last run (this rank)
        e.begin()
        frontier = self._allreduce(e.level_size(), SUM)
        levels, cum, level, budget = [frontier], frontier, 1, False
        while frontier > 0:
            _, _, verdict = e.counters()
            if self._allreduce(verdict, MAX) != 0:
                break
            if (self.max_levels and level >= self.max_levels) or (self.max_distinct and cum >= self.max_distinct):
                budget = True
                break
            local_n = e.level_size()
            rounds = self._allreduce(math.ceil(local_n / self.chunk), MAX)
            biggest = self._allreduce(local_n, MAX)
            stay = frontier >= self.stay_threshold * self.world and biggest * self.world <= self.rebalance_ratio * frontier
            self.phase_s["stay_levels" if stay else "move_levels"] = self.phase_s.get("stay_levels" if stay else "move_levels", 0) + 1
            for r in range(rounds):
                first = min(r * self.chunk, local_n)
                count = min(self.chunk, local_n - first)
                t0 = time.perf_counter()
                counts = e.expand(first, count, self.send_fp)
                e.sync()
                t1 = time.perf_counter()
                recv_fp, rcounts = self._a2a(self.send_fp, counts, 1)
                n = sum(rcounts)
                answers = torch.empty(max(n, 1), dtype=torch.uint8, device=self.dev)
                e.sync()
                t2 = time.perf_counter()
                e.probe(recv_fp, n, answers)
                t3 = time.perf_counter()
                back = self._a2a_back(answers, rcounts, counts)
                e.sync()
                t4 = time.perf_counter()
                if stay:
                    e.keep(back)
                    self.phase_s["keep"] = self.phase_s.get("keep", 0.0) + time.perf_counter() - t4
                    self.phase_s["rounds"] = self.phase_s.get("rounds", 0) + 1
                    for k, dt in zip(("expand", "a2a_fp", "probe", "a2a_ans"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                        self.phase_s[k] = self.phase_s.get(k, 0.0) + dt
                    continue
                scounts = e.materialise(back, self.send_states)
                t5 = time.perf_counter()
                # full states travel as whole 64-state blocks per owner (coalesced at both ends)
                blocks = [(c + 63) // 64 for c in scounts]
                recv_states, rblocks = self._a2a(self.send_states, blocks, 64 * self.W)
                rsc = self._exchange_counts(scounts)
                e.sync()
                t6 = time.perf_counter()
                off = 0
                for src_rank in range(len(rsc)):            # one bucket per source rank
                    if rsc[src_rank]:
                        e.ingest(recv_states[off * 64 * self.W:], rsc[src_rank])
                    off += rblocks[src_rank]
                t7 = time.perf_counter()
                for k, dt in zip(("expand", "a2a_fp", "probe", "a2a_ans", "materialise", "a2a_states", "ingest"),
                                 (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)):
                    self.phase_s[k] = self.phase_s.get(k, 0.0) + dt
                self.phase_s["rounds"] = self.phase_s.get("rounds", 0) + 1
            frontier = self._allreduce(e.end_level(), SUM)
            if frontier > 0:
                level += 1
                levels.append(frontier)
                cum += frontier
        generated, _, verdict = e.counters()
        generated = self._allreduce(generated, SUM)
        verdict = self._allreduce(verdict, MAX)
        if verdict == 0 and budget:
            verdict = 5
        if os.environ.get("TLAMC_PHASES") and self.rank == 0:
            print("phases[s]:", {k: round(v, 4) for k, v in self.phase_s.items()}, flush=True)
        return Result(distinct=cum, generated=generated, queue_left=frontier, depth=level, verdict=VERDICTS[verdict],
                      violated_invariant=-1, trace_len=0, levels=levels, seconds=0.0)

    def close(self):
        self.eng.close()
