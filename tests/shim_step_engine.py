"""CPU stand-in for tla_rust_amd.sharded.HipStepEngine, backed by tests/_shim (host build of the
device lowerings).  TEST ONLY: lets the multi-rank exchange loop run under gloo without a GPU."""
import ctypes as C

import torch

import helpers


class ShimStepEngine:
    def __init__(self, spec, params, rank, world):
        self.lib = helpers.shim_lib()
        L = self.lib
        L.shim_shard_create.restype = C.c_void_p
        L.shim_shard_create.argtypes = [C.POINTER(helpers.McSpecDesc), C.c_uint32, C.c_uint32]
        for name in ("begin", "level_size", "expand_launch", "expand_finish", "probe", "materialise", "ingest", "keep", "end_level", "counters", "destroy"):
            getattr(L, "shim_shard_" + name).restype = C.c_int if name != "destroy" else None
        L.shim_shard_begin.argtypes = [C.c_void_p]
        L.shim_shard_begin_replicated.restype = C.c_int
        L.shim_shard_begin_replicated.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.shim_shard_destroy.argtypes = [C.c_void_p]
        L.shim_shard_level_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.shim_shard_expand_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]
        L.shim_shard_expand_finish.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.shim_shard_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.shim_shard_materialise.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.shim_shard_ingest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.shim_shard_keep.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
        L.shim_shard_end_level.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.shim_shard_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        L.shim_shard_check_frontier.argtypes = [C.c_void_p]
        self.world = world
        d = helpers.spec_desc(spec, params)
        L.shim_state_bytes.restype = C.c_size_t
        L.shim_state_bytes.argtypes = [C.POINTER(helpers.McSpecDesc)]
        self.W = L.shim_state_bytes(C.byref(d))
        self.h = L.shim_shard_create(C.byref(d), rank, world)
        self.device = torch.device("cpu")

    def _ck(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} failed: {rc}")

    def begin(self):
        self._ck(self.lib.shim_shard_begin(self.h), "begin")

    def begin_replicated(self, min_frontier, max_distinct=0, max_levels=0):
        cap = C.c_uint32(4096)
        levels = (C.c_uint64 * 4096)()
        self._ck(self.lib.shim_shard_begin_replicated(self.h, min_frontier, max_distinct, max_levels, levels, C.byref(cap)), "begin_replicated")
        return [int(levels[i]) for i in range(cap.value)]

    def level_size(self):
        n = C.c_uint64()
        self.lib.shim_shard_level_size(self.h, C.byref(n))
        return n.value

    def expand_launch(self, slot, first, count, send_cap):
        self._ck(self.lib.shim_shard_expand_launch(self.h, slot, first, count), "expand_launch")

    def expand_finish(self, slot, send_fp):
        counts = (C.c_uint64 * self.world)()
        self._ck(self.lib.shim_shard_expand_finish(self.h, slot, send_fp.data_ptr(), send_fp.numel(), counts), "expand_finish")
        return list(counts)

    def probe(self, recv_fp, n, answers):
        self._ck(self.lib.shim_shard_probe(self.h, recv_fp.data_ptr(), n, answers.data_ptr()), "probe")

    def materialise(self, slot, answers_back, send_states):
        counts = (C.c_uint64 * self.world)()
        self._ck(self.lib.shim_shard_materialise(self.h, slot, answers_back.data_ptr(), send_states.data_ptr(),
                                                 send_states.numel() // self.W, counts), "materialise")
        return list(counts)

    def ingest(self, recv_states, n):
        self._ck(self.lib.shim_shard_ingest(self.h, recv_states.data_ptr(), n), "ingest")

    def keep(self, slot, answers_back):
        n = C.c_uint64()
        self._ck(self.lib.shim_shard_keep(self.h, slot, answers_back.data_ptr(), C.byref(n)), "keep")
        return n.value

    # fixed-capacity rounds, emulated on top of the variable-size step calls: packs / unpacks the in-band layout of
    # include/tlamc.h mc_shard_*_pack, so that the exchange loop of tla_rust_amd/sharded.py runs unchanged under gloo
    def expand_pack(self, slot, send_fp, cap):
        P = self.world
        tmp = torch.zeros(P * cap, dtype=torch.int64)
        counts = self.expand_finish(slot, tmp)
        assert max(counts) + 1 <= cap, "exchange bucket overflow"
        packed = torch.zeros(P * cap, dtype=torch.int64)
        off = 0
        for t in range(P):
            packed[t * cap] = counts[t]
            packed[t * cap + 1: t * cap + 1 + counts[t]] = tmp[off: off + counts[t]]
            off += counts[t]
        send_fp[: P * cap] = packed
        self._pack_counts = getattr(self, "_pack_counts", {})
        self._pack_counts[slot] = counts

    def probe_pack(self, recv_fp, cap, answers):
        P = self.world
        answers[: P * cap] = 0
        for s_ in range(P):
            n = int(recv_fp[s_ * cap])
            if n:
                fps = recv_fp[s_ * cap + 1: s_ * cap + 1 + n].contiguous()
                ans = torch.zeros(n, dtype=torch.uint8)
                self.probe(fps, n, ans)
                answers[s_ * cap + 1: s_ * cap + 1 + n] = ans

    def keep_pack(self, slot, answers_back, cap):
        counts = self._pack_counts[slot]
        parts = [answers_back[t * cap + 1: t * cap + 1 + counts[t]] for t in range(self.world)]
        flat = torch.cat(parts).contiguous() if sum(counts) else torch.zeros(1, dtype=torch.uint8)
        for t in range(self.world):  # answers outside the counts must be 0 (the HIP sender scans the whole packed range)
            assert int(answers_back[t * cap]) == 0 and int(answers_back[t * cap + 1 + counts[t]: (t + 1) * cap].sum()) == 0
        self.keep(slot, flat)

    def end_level(self):
        n = C.c_uint64()
        self.lib.shim_shard_end_level(self.h, C.byref(n))
        return n.value

    def check_frontier(self):
        self.lib.shim_shard_check_frontier(self.h)

    def counters(self):
        g, d, v = C.c_uint64(), C.c_uint64(), C.c_int32()
        self.lib.shim_shard_counters(self.h, C.byref(g), C.byref(d), C.byref(v))
        return g.value, d.value, v.value

    def sync(self):
        pass

    def stream_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    def close(self):
        self.lib.shim_shard_destroy(self.h)
