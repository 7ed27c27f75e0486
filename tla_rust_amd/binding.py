"""ctypes binding of include/tlamc.h.  Mirrors the C ABI one to one (same names, same argument
meaning, negative MC_E* codes raised as McError)."""
import ctypes as C
import os
import json
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["TLAMC_LIB"]) if os.environ.get("TLAMC_LIB") else PKG / "_build" / "libtlamc.so"   # (TLAMC_LIB: profiling builds, profiles/build_phase_prof.sh)

MC_MAX_LEVELS = 4096
SPEC_IDS = {"atomic_add": 1, "pcal_intro": 2, "raft": 3, "ssi": 4, "pcal": 5, "paxos": 6}
VERDICTS = ["ok", "invariant", "assert", "deadlock", "spec-error", "budget", "assume"]
MC_F_DEADLOCK, MC_F_TRACE, MC_F_TIMING, MC_F_MATRIX, MC_F_NOPROBE, MC_F_NOFAMILY = 1, 2, 4, 8, 16, 32
MC_F_UNVERIFIED = 512  # use the built-in lowering even when the module a wrapper EXTENDS cannot be found (include/tlamc.h)


class McError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what} (code {code})")
        self.code = code


class SpecDesc(C.Structure):
    _fields_ = [("spec_id", C.c_uint32), ("nparams", C.c_uint32), ("params", C.c_int64 * 16)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("table_capacity", C.c_uint64),
                ("arena_capacity", C.c_uint64), ("chunk_states", C.c_uint64), ("max_levels", C.c_uint64),
                ("max_distinct", C.c_uint64), ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32)]


class CResult(C.Structure):
    _fields_ = [("distinct", C.c_uint64), ("generated", C.c_uint64), ("queue_left", C.c_uint64),
                ("depth", C.c_uint32), ("verdict", C.c_int32), ("violated_invariant", C.c_int32),
                ("trace_len", C.c_uint32), ("levels", C.c_uint32), ("host_evaluated", C.c_uint32), ("unchecked_properties", C.c_uint32), ("seconds", C.c_double),
                ("level_distinct", C.c_uint64 * MC_MAX_LEVELS)]


class KernelStat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("ms_total", C.c_double), ("units", C.c_uint64)]


class KernelStats(C.Structure):
    _fields_ = [("expand", KernelStat), ("insert", KernelStat), ("materialise", KernelStat),
                ("state_bytes", C.c_uint64), ("cand_cells", C.c_uint64), ("inwave_states", C.c_uint64)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64)

MC_COMM_ID_BYTES = 128
MC_SHARD_NO_PREFIX = 1
MC_SHARD_FIXED_CAPS = 2
MC_SHARD_PACKED = 4
EXCHANGE_FLAGS = {"exact": 0, "measured": MC_SHARD_PACKED, "packed": MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS}


class ShardStats(C.Structure):
    """mc_shard_stats: what one rank's level loop did"""
    _fields_ = [("replicated_levels", C.c_uint64), ("stay_levels", C.c_uint64), ("move_levels", C.c_uint64), ("rounds", C.c_uint64),
                ("sent_bytes", C.c_uint64), ("distinct_local", C.c_uint64), ("max_frontier", C.c_uint64), ("mean_frontier", C.c_uint64),
                ("restarts", C.c_uint64), ("routed_candidates", C.c_uint64), ("fp_answer_bytes", C.c_uint64), ("measured_levels", C.c_uint64),
                ("engine_ns", C.c_uint64), ("collective_ns", C.c_uint64), ("collectives", C.c_uint64)]


class ShardOpts(C.Structure):
    """mc_shard_opts"""
    _fields_ = [("chunk_states", C.c_uint64), ("max_distinct", C.c_uint64), ("max_levels", C.c_uint64), ("replicate_until", C.c_uint64),
                ("packed_fanout", C.c_uint64), ("stay_threshold", C.c_uint64), ("rebalance_ratio", C.c_double), ("move_fanout", C.c_uint64),
                ("flags", C.c_uint32), ("cap_safety_pct", C.c_uint32), ("stats", C.POINTER(ShardStats))]


T_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
T_RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
T_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
T_A2AV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                     C.POINTER(C.c_uint64))
T_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class Transport(C.Structure):
    """mc_transport: the collectives the level loop runs on (RCCL from mc_comm_transport, or a host's own)"""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("hip_stream", C.c_void_p), ("alloc", T_ALLOC),
                ("release", T_RELEASE), ("all_to_all", T_A2A), ("all_to_all_v", T_A2AV), ("all_gather", T_GATHER),
                ("all_to_all_others", T_A2A)]   # optional: NULL (a positional constructor leaves it so) = use all_to_all


class Result(dict):
    __getattr__ = dict.__getitem__


_lib = None


def lib():
    """The loaded libtlamc.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m tla_rust_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both open the
    # GPU.  Loading torch first makes the dynamic linker resolve libtlamc.so's libamdhip64 (same SONAME)
    # to the copy torch already mapped, so device memory, streams and RCCL share one runtime.
    # (multi-process GPU work — RCCL over xGMI — needs dmabuf IPC on this pool's driver; the variable must be there before the HSA runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.path.exists("/dev/kfd"):  # (a box without a GPU never opens one: nothing to share, no reason to load torch's runtime)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(str(LIB_PATH))
    L.mc_engine_create.argtypes = [C.POINTER(SpecDesc), C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.mc_engine_run.argtypes = [C.c_void_p, C.POINTER(CResult)]
    L.mc_engine_step.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(CResult)]
    L.mc_engine_set_progress.argtypes = [C.c_void_p, PROGRESS_FN, C.c_void_p, C.c_double]
    L.mc_engine_request_stop.argtypes = [C.c_void_p]
    L.mc_engine_trace.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
    L.mc_engine_kernel_stats.argtypes = [C.c_void_p, C.POINTER(KernelStats)]
    L.mc_engine_read_states.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.mc_engine_debug_reexpand.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_double)]
    L.mc_engine_destroy.argtypes = [C.c_void_p]
    L.mc_engine_destroy.restype = None
    L.mc_state_bytes.argtypes = [C.POINTER(SpecDesc)]
    L.mc_state_bytes.restype = C.c_size_t
    L.mc_fp_owner.argtypes = [C.c_uint64, C.c_uint32]
    L.mc_fp_owner.restype = C.c_uint32
    L.mc_state_format.argtypes = [C.POINTER(SpecDesc), C.c_void_p, C.c_char_p, C.c_size_t]
    L.mc_action_name.argtypes = [C.POINTER(SpecDesc), C.c_int32]
    L.mc_action_name.restype = C.c_char_p
    L.mc_strerror.argtypes = [C.c_int]
    L.mc_strerror.restype = C.c_char_p
    L.mc_last_error.restype = C.c_char_p
    L.mc_device_count.restype = C.c_int
    U64P = C.POINTER(C.c_uint64)
    L.mc_shard_begin.argtypes = [C.c_void_p]
    L.mc_shard_begin_replicated.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, U64P, C.POINTER(C.c_uint32)]
    L.mc_shard_level_size.argtypes = [C.c_void_p, U64P]
    L.mc_shard_expand.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, U64P]
    L.mc_shard_set_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.mc_shard_expand_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64]
    L.mc_shard_expand_finish.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, U64P]
    L.mc_shard_materialise_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, U64P]
    L.mc_shard_keep_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, U64P]
    L.mc_engine_checkpoint.argtypes = [C.c_void_p, C.c_char_p]
    L.mc_engine_restore.argtypes = [C.c_void_p, C.c_char_p]
    L.mc_shard_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.mc_shard_materialise_parents.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.mc_shard_ingest_parents.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
    L.mc_shard_violation.argtypes = [C.c_void_p, C.POINTER(C.c_int32), U64P, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.mc_shard_fetch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32), U64P, C.POINTER(C.c_uint32)]
    L.mc_state_action.argtypes = [C.POINTER(SpecDesc), C.c_char_p, C.c_int32]
    L.mc_state_apply.argtypes = [C.POINTER(SpecDesc), C.c_char_p, C.c_int32, C.c_char_p]
    L.mc_shard_expand_pack.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.mc_shard_probe_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.mc_shard_keep_pack.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.mc_shard_materialise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, U64P]
    L.mc_shard_ingest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.mc_shard_keep.argtypes = [C.c_void_p, C.c_void_p, U64P]
    L.mc_shard_end_level.argtypes = [C.c_void_p, U64P]
    L.mc_shard_counters.argtypes = [C.c_void_p, U64P, U64P, C.POINTER(C.c_int32)]
    L.mc_shard_check_frontier.argtypes = [C.c_void_p]
    L.mc_comm_unique_id.argtypes = [C.c_char_p]
    L.mc_comm_create.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]
    L.mc_comm_destroy.argtypes = [C.c_void_p]
    L.mc_comm_destroy.restype = None
    L.mc_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.mc_comm_transport.argtypes = [C.c_void_p, C.POINTER(Transport)]
    L.mc_shard_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ShardOpts), C.POINTER(CResult)]
    L.mc_shard_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
    L.mc_shard_run_transport.argtypes = [C.c_void_p, C.POINTER(Transport), C.POINTER(ShardOpts), C.POINTER(CResult)]
    L.mc_shard_checkpoint.argtypes = [C.c_void_p, C.c_char_p]
    L.mc_shard_restore.argtypes = [C.c_void_p, C.c_char_p]
    L.mc_shard_note_levels.argtypes = [C.c_void_p, U64P, C.c_uint32, C.c_int32]
    L.mc_shard_resume.argtypes = [C.c_void_p, U64P, C.POINTER(C.c_uint32)]
    L.mc_shard_trace_transport.argtypes = [C.c_void_p, C.POINTER(Transport), C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_size_t),
                                           C.POINTER(C.c_int32)]
    L.mc_shard_info.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), U64P, C.POINTER(C.c_int32)]
    if hasattr(L, "mc_cfg_parse"):
        L.mc_cfg_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.mc_cfg_free.argtypes = [C.c_void_p]
        L.mc_cfg_free.restype = None
        L.mc_cfg_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.mc_spec_resolve.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(SpecDesc)]
        if hasattr(L, "mc_resolve_files"):
            L.mc_resolve_files.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(SpecDesc), C.POINTER(C.c_void_p)]
        L.mc_check_files.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Config), C.c_char_p, C.c_size_t, C.POINTER(CResult)]
    if hasattr(L, "mc_program_compile"):
        L.mc_pcal_translate.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.mc_program_compile.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.mc_program_spec.argtypes = [C.c_void_p, C.POINTER(SpecDesc)]
        L.mc_program_translated.argtypes = [C.c_void_p]
        L.mc_program_translated.restype = C.c_char_p
        L.mc_program_invariant.argtypes = [C.c_void_p, C.c_int]
        L.mc_program_invariant.restype = C.c_char_p
        L.mc_program_free.argtypes = [C.c_void_p]
        L.mc_program_free.restype = None
    _lib = L
    return L


def _check(rc, what):
    if rc < 0:
        L = lib()
        detail = L.mc_last_error().decode() or L.mc_strerror(rc).decode()
        raise McError(rc, f"{what}: {detail}")
    return rc


def spec_desc(spec, params):
    d = SpecDesc()
    d.spec_id = SPEC_IDS[spec] if isinstance(spec, str) else int(spec)
    d.nparams = len(params)
    for i, v in enumerate(params):
        d.params[i] = int(v)
    return d


def device_count():
    return lib().mc_device_count()


def state_bytes(spec, params):
    return lib().mc_state_bytes(C.byref(spec_desc(spec, params)))


def state_format(spec, params, state: bytes):
    buf = C.create_string_buffer(1 << 16)
    d = spec_desc(spec, params)
    n = _check(lib().mc_state_format(C.byref(d), state, buf, len(buf)), "mc_state_format")
    return buf.raw[:n].decode()


def state_action_name(spec, params, state: bytes, slot: int):
    """name of the action slot `slot` takes from the packed state (mc_state_action + mc_action_name)"""
    d = spec_desc(spec, params)
    a = _check(lib().mc_state_action(C.byref(d), state, slot), "mc_state_action")
    return lib().mc_action_name(C.byref(d), a).decode()


def state_apply(spec, params, state: bytes, slot: int):
    """the successor slot `slot` produces from the packed state (host evaluation, mc_state_apply)"""
    d = spec_desc(spec, params)
    out = C.create_string_buffer(lib().mc_state_bytes(C.byref(d)))
    _check(lib().mc_state_apply(C.byref(d), state, slot, out), "mc_state_apply")
    return out.raw


def _result(r: CResult):
    return Result(distinct=r.distinct, generated=r.generated, queue_left=r.queue_left, depth=r.depth,
                  verdict=VERDICTS[r.verdict], violated_invariant=r.violated_invariant, trace_len=r.trace_len,
                  levels=[r.level_distinct[i] for i in range(r.levels)], seconds=r.seconds, host_evaluated=bool(r.host_evaluated),
                  unchecked_properties=int(r.unchecked_properties))


class Engine:
    """One model-checking engine on one GPU (mc_engine_create / run / trace / destroy)."""

    def __init__(self, spec, params, device=0, table_capacity=0, arena_capacity=0, chunk_states=0, max_levels=0,
                 max_distinct=0, deadlock=True, trace=True, timing=False, matrix=False, shard_rank=0, shard_count=1, debug_flags=0, jit=False):
        self.spec, self.params = spec, list(params)
        self.desc = spec_desc(spec, params)
        # jit (MC_SPEC_PCAL): MC_F_JIT — the compiled program as generated code, built for the device when the engine is created
        flags = (MC_F_DEADLOCK if deadlock else 0) | (MC_F_TRACE if trace else 0) | (MC_F_TIMING if timing else 0) | \
            (MC_F_MATRIX if matrix else 0) | (262144 if jit else 0) | debug_flags
        self.cfg = Config(device, flags, table_capacity, arena_capacity, chunk_states, max_levels, max_distinct,
                          shard_rank, shard_count)
        self._h = C.c_void_p()
        _check(lib().mc_engine_create(C.byref(self.desc), C.byref(self.cfg), C.byref(self._h)), "mc_engine_create")

    def run(self):
        r = CResult()
        _check(lib().mc_engine_run(self._h, C.byref(r)), "mc_engine_run")
        return _result(r)

    def step(self, levels):
        """mc_engine_step: `levels` more BFS levels; continues in place after a budget stop, starts over after an end."""
        r = CResult()
        _check(lib().mc_engine_step(self._h, levels, C.byref(r)), "mc_engine_step")
        return _result(r)

    def trace(self):
        """[(action name, state text)] of the last counterexample."""
        W = lib().mc_state_bytes(C.byref(self.desc))
        cap = C.c_size_t(4096)
        states = C.create_string_buffer(W * cap.value)
        acts = (C.c_int32 * cap.value)()
        _check(lib().mc_engine_trace(self._h, states, acts, C.byref(cap)), "mc_engine_trace")
        out = []
        for k in range(cap.value):
            name = lib().mc_action_name(C.byref(self.desc), acts[k]).decode()
            out.append((name, state_format(self.spec, self.params, states.raw[k * W:(k + 1) * W])))
        return out

    def checkpoint(self, path):
        """TLC's checkpoint (testout1:10): states found so far + level boundaries + counters (+ parent pointers) -> file."""
        _check(lib().mc_engine_checkpoint(self._h, str(path).encode()), "mc_engine_checkpoint")

    def restore(self, path):
        """TLC's -recover: the next run() continues the checkpointed search."""
        _check(lib().mc_engine_restore(self._h, str(path).encode()), "mc_engine_restore")

    def shard_checkpoint(self, path):
        """this rank's file of a sharded run that ended without an error (one file per rank: mc_shard_checkpoint)"""
        _check(lib().mc_shard_checkpoint(self._h, str(path).encode()), "mc_shard_checkpoint")

    def shard_restore(self, path):
        """load this rank's file; the next Comm.shard_run / mc_shard_run_transport of the engines continues that run"""
        _check(lib().mc_shard_restore(self._h, str(path).encode()), "mc_shard_restore")

    def read_states(self, first, count):
        """Packed records of `count` states in discovery order starting at `first`."""
        W = lib().mc_state_bytes(C.byref(self.desc))
        buf = C.create_string_buffer(max(1, W * count))
        _check(lib().mc_engine_read_states(self._h, first, count, buf), "mc_engine_read_states")
        return [buf.raw[k * W:(k + 1) * W] for k in range(count)]

    def state_texts(self, first, count):
        return [state_format(self.spec, self.params, s) for s in self.read_states(first, count)]

    def debug_reexpand(self, extra_flags=0):
        ms = C.c_double()
        _check(lib().mc_engine_debug_reexpand(self._h, extra_flags, C.byref(ms)), "mc_engine_debug_reexpand")
        return ms.value

    def debug_flags(self, set=0, clear=0):
        """profiling only (mc_engine_debug_flags): switch A/B / ablation bits of the engine's flags between two calls"""
        L = lib()
        L.mc_engine_debug_flags.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.mc_engine_debug_flags.restype = C.c_int
        _check(L.mc_engine_debug_flags(self._h, set, clear), "mc_engine_debug_flags")

    def kernel_stats(self):
        ks = KernelStats()
        _check(lib().mc_engine_kernel_stats(self._h, C.byref(ks)), "mc_engine_kernel_stats")
        f = lambda s: dict(launches=s.launches, ms_total=s.ms_total, units=s.units)
        return dict(expand=f(ks.expand), insert=f(ks.insert), materialise=f(ks.materialise),
                    state_bytes=ks.state_bytes, cand_cells=ks.cand_cells, inwave_states=ks.inwave_states)

    # ---- sharded step API (mc_shard_*): raw device pointers in, counts out
    def shard_begin(self):
        _check(lib().mc_shard_begin(self._h), "mc_shard_begin")

    def shard_begin_replicated(self, min_frontier, max_distinct=0, max_levels=0):
        cap = C.c_uint32(4096)
        levels = (C.c_uint64 * 4096)()
        _check(lib().mc_shard_begin_replicated(self._h, min_frontier, max_distinct, max_levels, levels, C.byref(cap)), "mc_shard_begin_replicated")
        return [int(levels[i]) for i in range(cap.value)]

    def shard_level_size(self):
        n = C.c_uint64()
        _check(lib().mc_shard_level_size(self._h, C.byref(n)), "mc_shard_level_size")
        return n.value

    def shard_expand(self, first, count, send_fp_ptr, send_cap):
        counts = (C.c_uint64 * max(1, self.cfg.shard_count))()
        _check(lib().mc_shard_expand(self._h, first, count, send_fp_ptr, send_cap, counts), "mc_shard_expand")
        return list(counts)

    def shard_set_stream(self, hip_stream, enable=True):
        _check(lib().mc_shard_set_stream(self._h, hip_stream, int(enable)), "mc_shard_set_stream")

    def shard_expand_launch(self, slot, first, count, send_cap):
        _check(lib().mc_shard_expand_launch(self._h, slot, first, count, send_cap), "mc_shard_expand_launch")

    def shard_expand_finish(self, slot, send_fp_ptr, send_cap):
        counts = (C.c_uint64 * max(1, self.cfg.shard_count))()
        _check(lib().mc_shard_expand_finish(self._h, slot, send_fp_ptr, send_cap, counts), "mc_shard_expand_finish")
        return list(counts)

    def shard_probe(self, recv_fp_ptr, n, answers_ptr):
        _check(lib().mc_shard_probe(self._h, recv_fp_ptr, n, answers_ptr), "mc_shard_probe")

    def shard_materialise(self, answers_back_ptr, send_states_ptr, send_cap, slot=0):
        counts = (C.c_uint64 * max(1, self.cfg.shard_count))()
        _check(lib().mc_shard_materialise_slot(self._h, slot, answers_back_ptr, send_states_ptr, send_cap, counts), "mc_shard_materialise")
        return list(counts)

    def shard_ingest(self, recv_states_ptr, n):
        _check(lib().mc_shard_ingest(self._h, recv_states_ptr, n), "mc_shard_ingest")

    def shard_keep(self, answers_back_ptr, slot=0):
        n = C.c_uint64()
        _check(lib().mc_shard_keep_slot(self._h, slot, answers_back_ptr, C.byref(n)), "mc_shard_keep")
        return n.value

    # counterexamples of a sharded run (engine created with trace=True)
    def shard_materialise_parents(self, slot, send_parents_ptr):
        _check(lib().mc_shard_materialise_parents(self._h, slot, send_parents_ptr), "mc_shard_materialise_parents")

    def shard_ingest_parents(self, recv_parents_ptr, n, src_rank):
        _check(lib().mc_shard_ingest_parents(self._h, recv_parents_ptr, n, src_rank), "mc_shard_ingest_parents")

    def shard_violation(self):
        """(found, arena index, slot code, verdict name, invariant index) of this rank's first violation"""
        f, i, sl, v, inv = C.c_int32(), C.c_uint64(), C.c_uint32(), C.c_int32(), C.c_int32()
        _check(lib().mc_shard_violation(self._h, C.byref(f), C.byref(i), C.byref(sl), C.byref(v), C.byref(inv)), "mc_shard_violation")
        return bool(f.value), i.value, sl.value, VERDICTS[v.value], inv.value

    def shard_fetch(self, idx):
        """(packed state, parent rank, parent index, parent slot) of the state at arena index idx of this rank"""
        st = C.create_string_buffer(lib().mc_state_bytes(C.byref(self.desc)))
        pr, pi, ps = C.c_uint32(), C.c_uint64(), C.c_uint32()
        _check(lib().mc_shard_fetch(self._h, idx, st, C.byref(pr), C.byref(pi), C.byref(ps)), "mc_shard_fetch")
        return st.raw, pr.value, pi.value, ps.value

    def shard_expand_pack(self, slot, send_fp_ptr, cap):
        _check(lib().mc_shard_expand_pack(self._h, slot, send_fp_ptr, cap), "mc_shard_expand_pack")

    def shard_probe_pack(self, recv_fp_ptr, cap, answers_ptr):
        _check(lib().mc_shard_probe_pack(self._h, recv_fp_ptr, cap, answers_ptr), "mc_shard_probe_pack")

    def shard_keep_pack(self, slot, answers_back_ptr, cap):
        _check(lib().mc_shard_keep_pack(self._h, slot, answers_back_ptr, cap), "mc_shard_keep_pack")

    def shard_end_level(self):
        n = C.c_uint64()
        _check(lib().mc_shard_end_level(self._h, C.byref(n)), "mc_shard_end_level")
        return n.value

    def set_progress(self, fn, min_interval_seconds=1.0):
        """fn(levels, generated, distinct, queue) between BFS levels of run(), at most once per interval (mc_engine_set_progress)"""
        self._progress = PROGRESS_FN(lambda _u, lv, g, d, q: fn(lv, g, d, q)) if fn else PROGRESS_FN()
        _check(lib().mc_engine_set_progress(self._h, self._progress, None, min_interval_seconds), "mc_engine_set_progress")

    def request_stop(self):
        """from inside a progress callback: the run stops before its next BFS level, verdict "budget" (mc_engine_request_stop)"""
        _check(lib().mc_engine_request_stop(self._h), "mc_engine_request_stop")

    def shard_check_frontier(self):
        _check(lib().mc_shard_check_frontier(self._h), "mc_shard_check_frontier")

    def shard_counters(self):
        g, d, v = C.c_uint64(), C.c_uint64(), C.c_int32()
        _check(lib().mc_shard_counters(self._h, C.byref(g), C.byref(d), C.byref(v)), "mc_shard_counters")
        return g.value, d.value, v.value

    def close(self):
        if self._h:
            lib().mc_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """mc_comm_*: one rank's RCCL communicator (hip-rccl back-end of the C ABI).  Rank 0 makes the id (Comm.unique_id()) and the
    host ships its 128 bytes to the other ranks however it likes (a file, a TCP store, the environment)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(MC_COMM_ID_BYTES)
        _check(lib().mc_comm_unique_id(buf), "mc_comm_unique_id")
        return buf.raw

    def __init__(self, uid: bytes, rank: int, world: int, device: int):
        self._h = C.c_void_p()
        self.rank, self.world, self.device = rank, world, device
        _check(lib().mc_comm_create(uid, rank, world, device, C.byref(self._h)), "mc_comm_create")

    def all_gather_u64(self, value: int):
        """every rank's value (a barrier when the values are ignored)"""
        mine = C.c_uint64(int(value))
        out = (C.c_uint64 * self.world)()
        _check(lib().mc_comm_all_gather(self._h, C.byref(mine), out, 8), "mc_comm_all_gather")
        return [int(x) for x in out]

    def all_gather_f64(self, value: float):
        mine = C.c_double(float(value))
        out = (C.c_double * self.world)()
        _check(lib().mc_comm_all_gather(self._h, C.byref(mine), out, 8), "mc_comm_all_gather")
        return [float(x) for x in out]

    def shard_run(self, engine, **opts):
        """mc_shard_run: the whole sharded search over this communicator; returns (Result, stats dict)"""
        st = ShardStats()
        o = shard_opts(stats=st, **opts)
        r = CResult()
        _check(lib().mc_shard_run(engine._h, self._h, C.byref(o), C.byref(r)), "mc_shard_run")
        return _result(r), {k: int(getattr(st, k)) for k, _ in ShardStats._fields_}

    def shard_trace(self, engine):
        """mc_shard_trace (collective): [(action name, state text)] of the counterexample, or None"""
        sp, pr = engine.spec, engine.params
        return _trace_from(lambda st, sl, n, fs: lib().mc_shard_trace(engine._h, self._h, st, sl, n, fs), "mc_shard_trace", state_bytes(sp, pr),
                           lambda st: state_format(sp, pr, st), lambda st, slot: state_action_name(sp, pr, st, slot),
                           lambda st, slot: state_apply(sp, pr, st, slot))

    def close(self):
        if self._h:
            lib().mc_comm_destroy(self._h)
            self._h = C.c_void_p()


def shard_opts(stats=None, chunk_states=0, max_distinct=0, max_levels=0, replicate_until=None, packed_fanout=0, stay_threshold=0,
               rebalance_ratio=0.0, move_fanout=0, exchange="exact", cap_safety_pct=0):
    """mc_shard_opts; replicate_until = 0 shards from Init on (MC_SHARD_NO_PREFIX), None = the default prefix; exchange: how a stay
    level moves its candidates — "exact" (host-paced rounds, exact sizes: the default), "measured" (pipelined fixed-capacity rounds, buckets
    sized from the previous level's measured fill: MC_SHARD_PACKED) or "packed" (the same from packed_fanout alone: + MC_SHARD_FIXED_CAPS)"""
    o = ShardOpts()
    o.chunk_states, o.max_distinct, o.max_levels = chunk_states, max_distinct, max_levels
    o.replicate_until = replicate_until or 0
    o.flags = (MC_SHARD_NO_PREFIX if replicate_until == 0 else 0) | EXCHANGE_FLAGS[exchange]
    o.cap_safety_pct = cap_safety_pct
    o.packed_fanout, o.stay_threshold, o.rebalance_ratio, o.move_fanout = packed_fanout, stay_threshold, rebalance_ratio, move_fanout
    if stats is not None:
        o.stats = C.pointer(stats)
    return o


def _trace_from(call, what, W, fmt, action, apply):
    """shared by Comm.shard_trace and sharded.ShardedChecker.counterexample: run the collective walk (call fills packed states
    and the slot that led to each), name the actions and print the states: [(action name, TLA+ text)], or None"""
    cap = C.c_size_t(4096)
    states = C.create_string_buffer(W * cap.value)
    slots = (C.c_int32 * cap.value)()
    final = C.c_int32(-1)
    _check(call(states, slots, C.byref(cap), C.byref(final)), what)
    if cap.value == 0:
        return None
    sts = [states.raw[k * W:(k + 1) * W] for k in range(cap.value)]
    out = [("Initial predicate", fmt(sts[0]))]
    for k in range(1, len(sts)):
        out.append((action(sts[k - 1], slots[k]), fmt(sts[k])))
    if final.value >= 0:  # an invariant broken by a successor that is stored nowhere: rebuilt from its parent
        out.append((action(sts[-1], final.value), fmt(apply(sts[-1], final.value))))
    return out


def pcal_translate(tla_text: str) -> str:
    """`pcal2tla`: the module text with the TLA+ translation of its PlusCal algorithm inserted (host only)."""
    need = _check(lib().mc_pcal_translate(tla_text.encode(), None, 0), "mc_pcal_translate")
    buf = C.create_string_buffer(need + 1)
    _check(lib().mc_pcal_translate(tla_text.encode(), buf, len(buf)), "mc_pcal_translate")
    return buf.value.decode()


class Program:
    """A PlusCal module compiled for the GPU interpreter (mc_program_* of include/tlamc.h).
    Engine("pcal", program.params) checks it; the program must outlive its engines."""

    def __init__(self, tla_text: str, cfg_text: str = None):
        h = C.c_void_p()
        _check(lib().mc_program_compile(tla_text.encode(), cfg_text.encode() if cfg_text is not None else None, C.byref(h)),
               "mc_program_compile")
        self._h = h
        d = SpecDesc()
        _check(lib().mc_program_spec(h, C.byref(d)), "mc_program_spec")
        self.params = [int(d.params[0])]

    def translated(self):
        return lib().mc_program_translated(self._h).decode()

    def invariant(self, index):
        return lib().mc_program_invariant(self._h, index).decode()

    def close(self):
        if self._h:
            lib().mc_program_free(self._h)
            self._h = None


def cfg_parse(text: str):
    """Parse a TLC .cfg file (ConfigFileGrammar.tla:4-32) with the library's parser; returns a dict."""
    h = C.c_void_p()
    b = text.encode()
    _check(lib().mc_cfg_parse(b, len(b), C.byref(h)), "mc_cfg_parse")
    try:
        buf = C.create_string_buffer(1 << 20)
        _check(lib().mc_cfg_json(h, buf, len(buf)), "mc_cfg_json")
        return json.loads(buf.value.decode())
    finally:
        lib().mc_cfg_free(h)


def spec_resolve(module: str, cfg_text: str):
    """mc_spec_resolve: (module name, cfg text) -> (spec id, parameter list) of the lowering the registry picks."""
    h = C.c_void_p()
    b = cfg_text.encode()
    _check(lib().mc_cfg_parse(b, len(b), C.byref(h)), "mc_cfg_parse")
    try:
        d = SpecDesc()
        _check(lib().mc_spec_resolve(module.encode(), h, C.byref(d)), "mc_spec_resolve")
        return int(d.spec_id), [int(d.params[i]) for i in range(d.nparams)]
    finally:
        lib().mc_cfg_free(h)


class ResolvedSpec:
    """mc_resolve_files: X.tla + X.cfg -> (spec name, params) for Engine / sharded.ShardedChecker.  Keeps the compiled
    PlusCal program (if any) alive; close() after the engines."""

    def __init__(self, tla_path, cfg_path=None, generic=False, unverified=False):
        d, prog = SpecDesc(), C.c_void_p()
        _check(lib().mc_resolve_files(str(tla_path).encode(), str(cfg_path).encode() if cfg_path else None,
                                      (128 if generic else 0) | (MC_F_UNVERIFIED if unverified else 0), C.byref(d), C.byref(prog)), "mc_resolve_files")
        names = {v: k for k, v in SPEC_IDS.items()}
        self.spec = names[int(d.spec_id)]
        self.params = [int(d.params[i]) for i in range(d.nparams)]
        self._prog = prog

    def close(self):
        if self._prog:
            lib().mc_program_free(self._prog)
            self._prog = C.c_void_p()


def check_files(tla_path, cfg_path=None, device=0, **kw):
    """`tlc X.tla` end to end (reference Makefile:6-7): returns (Result, report text)."""
    cfg = Config(device, MC_F_DEADLOCK | MC_F_TRACE | (MC_F_UNVERIFIED if kw.get("unverified") else 0), kw.get("table_capacity", 0), kw.get("arena_capacity", 0),
                 kw.get("chunk_states", 0), kw.get("max_levels", 0), kw.get("max_distinct", 0), 0, 1)
    r = CResult()
    buf = C.create_string_buffer(1 << 20)
    rc = lib().mc_check_files(str(tla_path).encode(), str(cfg_path).encode() if cfg_path else None, C.byref(cfg), buf,
                              len(buf), C.byref(r))
    _check(rc, "mc_check_files")
    return _result(r), buf.value.decode()
