"""Test helpers: ctypes bindings of the CPU oracle (oracle/_build/liboracle.so) and of the
test-only host build of the device lowerings (tests/_shim/_build/libshim.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch the oracle.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
SHIM_DIR = ROOT / "tests" / "_shim"

OR_MAX_LEVELS = 4096
OR_MAX_TRACE = 4096
VERDICTS = ["ok", "invariant", "assert", "deadlock", "spec-error", "budget", "assume"]


class OrOptions(C.Structure):
    _fields_ = [("max_levels", C.c_uint64), ("max_distinct", C.c_uint64), ("check_deadlock", C.c_int),
                ("stop_on_violation", C.c_int), ("dump_path", C.c_char_p)]


class OrResult(C.Structure):
    _fields_ = [("distinct", C.c_uint64), ("generated", C.c_uint64), ("queue_left", C.c_uint64),
                ("depth", C.c_uint32), ("verdict", C.c_int), ("violated_invariant", C.c_int),
                ("level_distinct", C.c_uint64 * OR_MAX_LEVELS), ("level_generated", C.c_uint64 * OR_MAX_LEVELS),
                ("trace_len", C.c_uint32), ("trace_action", C.c_int * OR_MAX_TRACE), ("seconds", C.c_double),
                ("max_stat", C.c_uint64 * 8), ("arena_bytes", C.c_uint64)]


def build_oracle():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)
    return ORACLE_DIR / "_build" / "liboracle.so"


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        so = ORACLE_DIR / "_build" / "liboracle.so"
        if not so.exists():
            build_oracle()
        lib = C.CDLL(str(so))
        lib.oracle_run.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(OrOptions), C.POINTER(OrResult)]
        lib.oracle_run.restype = C.c_int
        lib.oracle_run_mt.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(OrOptions), C.c_int, C.c_double, C.POINTER(OrResult)]
        lib.oracle_run_mt.restype = C.c_int
        lib.oracle_trace_state.argtypes = [C.c_uint32]
        lib.oracle_trace_state.restype = C.c_char_p
        lib.oracle_action_name.argtypes = [C.c_char_p, C.c_int]
        lib.oracle_action_name.restype = C.c_char_p
        lib.oracle_last_error.restype = C.c_char_p
        _oracle = lib
    return _oracle


def oracle_run(spec, params, max_levels=0, max_distinct=0, check_deadlock=True, stop=1, dump=None):
    """Run the oracle; returns a dict with counts, verdict, per-level distinct and the trace."""
    lib = oracle_lib()
    p = (C.c_int64 * len(params))(*params)
    opt = OrOptions(max_levels, max_distinct, int(check_deadlock), stop, dump.encode() if dump else None)
    res = OrResult()
    rc = lib.oracle_run(spec.encode(), p, len(params), C.byref(opt), C.byref(res))
    if rc:
        raise RuntimeError(lib.oracle_last_error().decode())
    trace = [(res.trace_action[k], lib.oracle_trace_state(k).decode()) for k in range(res.trace_len)]
    return dict(distinct=res.distinct, generated=res.generated, queue_left=res.queue_left, depth=res.depth,
                verdict=VERDICTS[res.verdict], violated_invariant=res.violated_invariant,
                levels=[res.level_distinct[i] for i in range(res.depth)], trace=trace, seconds=res.seconds,
                max_stat=list(res.max_stat))


def oracle_run_mt(spec, params, threads, max_levels=0, max_distinct=0, check_deadlock=True, max_seconds=0.0):
    """The multi-threaded oracle (oracle/bfs_mt.c): counts, depth, per-level counts and verdict; no trace."""
    lib = oracle_lib()
    p = (C.c_int64 * len(params))(*params)
    opt = OrOptions(max_levels, max_distinct, int(check_deadlock), 1, None)
    res = OrResult()
    rc = lib.oracle_run_mt(spec.encode(), p, len(params), C.byref(opt), threads, max_seconds, C.byref(res))
    if rc:
        raise RuntimeError(lib.oracle_last_error().decode())
    return dict(distinct=res.distinct, generated=res.generated, queue_left=res.queue_left, depth=res.depth,
                verdict=VERDICTS[res.verdict], violated_invariant=res.violated_invariant,
                levels=[res.level_distinct[i] for i in range(res.depth)], seconds=res.seconds, max_stat=list(res.max_stat))


def raft_oracle_params(dev):
    """lowering's raft parameter vector {n, MCR, MaxTerm, MaxLogLen, MaxMsgs, invMask, cm, ce, ca, MaxMsgKeys} -> the
    oracle's {n, MCR, MaxTerm, MaxLogLen, MaxMsgs, invMask, naive, MaxMsgKeys} (the oracle has no slot capacities)"""
    dev = list(dev)
    return dev[:6] + ([0, dev[9]] if len(dev) > 9 and dev[9] else [])


def raft_device_params(orc, cm=0, ce=0, ca=0):
    """inverse of raft_oracle_params: the oracle's vector -> the lowering's, with slot-array capacities (0 = defaults; with a
    MaxMsgKeys bound the message array needs exactly that many slots)"""
    orc = list(orc)
    keys = orc[7] if len(orc) > 7 else 0
    if not (cm or ce or ca or keys):
        return orc[:6]
    return orc[:6] + [cm or keys, ce, ca] + ([keys] if keys else [])


# ---------------------------------------------------------------------------------- shim
MC_MAX_LEVELS = 4096
SPEC_IDS = {"atomic_add": 1, "pcal_intro": 2, "raft": 3, "ssi": 4, "pcal": 5, "paxos": 6}


class McSpecDesc(C.Structure):
    _fields_ = [("spec_id", C.c_uint32), ("nparams", C.c_uint32), ("params", C.c_int64 * 16)]


def spec_desc(spec, params):
    d = McSpecDesc()
    d.spec_id = SPEC_IDS[spec]
    d.nparams = len(params)
    for i, v in enumerate(params):
        d.params[i] = v
    return d


class ShimResult(C.Structure):
    _fields_ = [("distinct", C.c_uint64), ("generated", C.c_uint64), ("queue_left", C.c_uint64),
                ("depth", C.c_uint32), ("verdict", C.c_int32), ("violated_invariant", C.c_int32),
                ("trace_len", C.c_uint32), ("levels", C.c_uint32), ("fp_mismatch", C.c_uint64),
                ("level_distinct", C.c_uint64 * MC_MAX_LEVELS)]


def build_shim():
    out = SHIM_DIR / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libshim.so"
    csrc = ROOT / "tla_rust_amd" / "csrc"
    pcal = [csrc / "pcal.cpp", csrc / "pcal_compile.cpp", csrc / "pcal_codegen.cpp"]  # the PlusCal front-end is host code: linked as is
    srcs = [SHIM_DIR / "shim.cpp"] + pcal + list(csrc.glob("*.h")) + [ROOT / "include" / "tlamc.h"]
    def fresh():
        return so.exists() and all(so.stat().st_mtime >= s.stat().st_mtime for s in srcs)
    if fresh():
        return so
    # several ranks of a multi-process test may arrive here together: one builds (into a temporary name, renamed when
    # complete), the others wait for the lock and find the library fresh
    import fcntl
    with open(out / ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not fresh():
            tmp = out / f"libshim.{os.getpid()}.so"
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(tmp), str(SHIM_DIR / "shim.cpp")] + [str(x) for x in pcal],
                           check=True)
            os.replace(tmp, so)
    return so


def build_tlaeval_door():
    """tests/_tlaeval: the host evaluator of the product (tla_rust_amd/csrc/tlaeval.cpp) behind a test-only door, so that it can be
    run on module texts the product itself never evaluates on the host (the ones with a GPU lowering)"""
    d = ROOT / "tests" / "_tlaeval"
    out = d / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libtlaeval_door.so"
    csrc = ROOT / "tla_rust_amd" / "csrc"
    srcs = [d / "door.cpp", csrc / "tlaeval.cpp", csrc / "tlaeval.h"]
    if not so.exists() or any(so.stat().st_mtime < s.stat().st_mtime for s in srcs):
        tmp = out / f"libtlaeval_door.{os.getpid()}.so"
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(tmp), str(srcs[0]), str(srcs[1]), "-lpthread"], check=True)
        os.replace(tmp, so)
    return so


def tlaeval_run(tla, cfg, search=(), max_levels=0, deadlock=True, dump=None, order=(), symmetry=True):
    """the host evaluator on module + cfg files -> dict(rc, distinct, generated, depth, verdict (MC_V_*), levels, ...)"""
    import ctypes as C
    import json
    lib = C.CDLL(str(build_tlaeval_door()))
    lib.tlaeval_door.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 16)
    lib.tlaeval_door(str(tla).encode(), str(cfg).encode(), ":".join(str(s) for s in search).encode(), max_levels, 1 if deadlock else 0, 1 if symmetry else 0,
                     str(dump).encode() if dump else None, ",".join(order).encode(), buf, len(buf))
    return json.loads(buf.value.decode())


def build_fakerccl():
    """tests/_fakerccl: the librccl stand-in that lets P ranks of the hip-rccl back-end share ONE GPU ($TLAMC_RCCL)"""
    d = ROOT / "tests" / "_fakerccl"
    out = d / "_build"
    out.mkdir(exist_ok=True)
    so = out / "libfakerccl.so"
    src = d / "fakerccl.cpp"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        tmp = out / f"libfakerccl.{os.getpid()}.so"
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(tmp), str(src),
                        "-lpthread", "-lrt"], check=True)
        os.replace(tmp, so)
    return so


_shim = None


def shim_lib():
    global _shim
    if _shim is None:
        lib = C.CDLL(str(build_shim()))
        lib.shim_run.argtypes = [C.POINTER(McSpecDesc), C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.POINTER(ShimResult)]
        lib.shim_run.restype = C.c_int
        _shim = lib
    return _shim


def shim_run(spec, params, max_levels=0, max_distinct=0, check_deadlock=True, dump=None):
    lib = shim_lib()
    d = spec_desc(spec, params)
    res = ShimResult()
    rc = lib.shim_run(C.byref(d), max_levels, max_distinct, int(check_deadlock), dump.encode() if dump else None, C.byref(res))
    if rc:
        raise RuntimeError(f"shim_run failed: {rc}")
    return dict(distinct=res.distinct, generated=res.generated, queue_left=res.queue_left, depth=res.depth,
                verdict=VERDICTS[res.verdict], violated_invariant=res.violated_invariant, trace_len=res.trace_len,
                levels=[res.level_distinct[i] for i in range(res.levels)], fp_mismatch=res.fp_mismatch)


class ShimProgram:
    """A PlusCal module compiled by the host build of the front-end (tests only)."""

    def __init__(self, tla_text, invariants=(), constants=None, constraints=()):
        lib = shim_lib()
        lib.shim_program_compile2.restype = C.c_void_p
        lib.shim_program_compile2.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.shim_pcal_error.restype = C.c_char_p
        lib.shim_program_free.argtypes = [C.c_void_p]
        lib.shim_program_translated.restype = C.c_char_p
        lib.shim_program_translated.argtypes = [C.c_void_p]
        consts = ",".join(f"{k}={int(v)}" for k, v in (constants or {}).items())   # (TRUE / FALSE: 1 / 0, like frontend.cpp to_const)
        self.h = lib.shim_program_compile2(tla_text.encode(), ",".join(invariants).encode(), consts.encode(), ",".join(constraints).encode())
        if not self.h:
            raise RuntimeError(lib.shim_pcal_error().decode())
        self.lib = lib

    @property
    def params(self):
        return [self.h]

    def translated(self):
        return self.lib.shim_program_translated(self.h).decode()

    def close(self):
        if self.h:
            self.lib.shim_program_free(self.h)
            self.h = None


def program_codegen(prog):
    """the generated C++ of a ShimProgram (pcal_codegen.cpp through the host build's door); RuntimeError when the translator refuses it"""
    lib = shim_lib()
    lib.shim_program_codegen.restype = C.c_long
    lib.shim_program_codegen.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.shim_last_error.restype = C.c_char_p
    n = lib.shim_program_codegen(prog.h, None, 0)
    if n < 0:
        raise RuntimeError(lib.shim_last_error().decode())
    buf = C.create_string_buffer(n + 1)
    lib.shim_program_codegen(prog.h, buf, n + 1)
    return buf.value.decode()


class GenCheck(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("distinct", "generated", "mismatches", "states_checked", "pairs_checked", "first_bad_state", "first_bad_slot")] + \
               [("depth", C.c_uint32), ("first_bad_kind", C.c_int32), ("stored_words", C.c_uint32), ("vm_words", C.c_uint32)]


def gen_check(prog, max_states=0):
    """tests/_gen/harness.cpp built around the generated code of `prog` (g++, cached by the text's hash): generated code against the
    interpreter on every reachable state and slot"""
    import hashlib
    text = program_codegen(prog)
    out = ROOT / "tests" / "_gen" / "_build"
    out.mkdir(exist_ok=True)
    defs = os.environ.get("GEN_CHECK_DEFS", "").split()   # (e.g. -DMC_GEN_FP_SUM=1: the A/B forms of spec_gen.h)
    tag = hashlib.sha256((text + (ROOT / "tests" / "_gen" / "harness.cpp").read_text() + (ROOT / "tla_rust_amd" / "csrc" / "spec_gen.h").read_text() + " ".join(defs)).encode()).hexdigest()[:16]
    so, hdr = out / f"libgen_{tag}.so", out / f"gen_{tag}.h"
    if not so.exists():
        hdr.write_text(text)
        tmp = out / f"libgen_{tag}.{os.getpid()}.tmp"
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-I", str(ROOT / "tla_rust_amd" / "csrc"), "-I", str(ROOT / "include"),
                        f'-DGEN_HEADER="{hdr}"', *defs, "-o", str(tmp), str(ROOT / "tests" / "_gen" / "harness.cpp")], check=True)
        os.replace(tmp, so)
    C.CDLL(str(build_shim()), mode=C.RTLD_GLOBAL)   # the interpreter's host helpers (vm_make_params, ...) live in the front-end
    lib = C.CDLL(str(so))
    lib.gen_check.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(GenCheck)]
    r = GenCheck()
    rc = lib.gen_check(prog.h, max_states, C.byref(r))
    if rc:
        raise RuntimeError(f"gen_check: {rc}")
    return {k: getattr(r, k) for k, _ in GenCheck._fields_}


def pcal_translate(tla_text):
    lib = shim_lib()
    lib.shim_pcal_translate.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    lib.shim_pcal_error.restype = C.c_char_p
    buf = C.create_string_buffer(1 << 20)
    n = lib.shim_pcal_translate(tla_text.encode(), buf, len(buf))
    if n < 0:
        raise RuntimeError(lib.shim_pcal_error().decode())
    return buf.value.decode()


def read_dump(path):
    """dump file -> {level: sorted list of state texts}"""
    by_level = {}
    with open(path) as f:
        for line in f:
            lvl, txt = line.rstrip("\n").split(" ", 1)
            by_level.setdefault(int(lvl[1:]), []).append(txt)
    return {k: sorted(v) for k, v in by_level.items()}


def run_with_master_port(make_cmd, tries=4, **kw):
    """subprocess.run(make_cmd(port), ...) for a torch.distributed.run launch that needs a free rendezvous port.  A port found by bind(0) +
    close can be taken again before the launcher's agent binds it (round 6: EADDRINUSE on the GPU box ended a `pytest -x` run): another
    port is tried when the launcher says so."""
    import socket
    p = None
    for _ in range(tries):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        p = subprocess.run(make_cmd(port), **kw)
        if p.returncode == 0 or "EADDRINUSE" not in (p.stderr or "") + (p.stdout or ""):
            return p
    return p
