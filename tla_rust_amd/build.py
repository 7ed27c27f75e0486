"""Builds libtlamc.so (HIP kernels + C ABI, gfx950 only) in-tree under tla_rust_amd/_build/.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only dev container."""
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "_build"
LIB = OUT / "libtlamc.so"
CLI = OUT / "mc"

SOURCES = ["engine.hip", "frontend.cpp"]


def _stale(target, deps):
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force=False, verbose=False):
    OUT.mkdir(exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "tlamc.h"]
    srcs = [str(CSRC / s) for s in SOURCES if (CSRC / s).exists()]
    # engine.hip is compiled once per group of specs (MC_TU = 1..7) plus once for the C ABI (MC_TU = 0), in
    # parallel: the unrolled per-spec kernels dominate compile time.  Each object is rebuilt only when one of the
    # files it really depends on changed (a TU instantiates the kernels of its own spec header only).
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
              "-I", str(PKG.parent / "include")]
    if os.environ.get("TLAMC_LINE_TABLES"):  # source lines for rocprofv3's PC sampling (profiles/pcsample.sh); code generation unchanged
        common.append("-gline-tables-only")
    if os.environ.get("TLAMC_EXTRA_DEFS"):   # A/B builds: extra -D flags (e.g. -DMC_NT_PARENT), space-separated
        common += os.environ["TLAMC_EXTRA_DEFS"].split()
    if os.environ.get("TLAMC_PHASE_PROF"):   # per-phase cycle counters inside k_expand_family (profiles/phase_prof.py): a profiling build
        common.append("-DMC_PHASE_PROF")
    base = [CSRC / "engine.hip", CSRC / "engine_kernels.h", CSRC / "mc_common.h", CSRC / "spec_registry.h", PKG.parent / "include" / "tlamc.h"]
    own = {0: ["spec_pluscal.h", "spec_raft.h", "spec_ssi.h", "spec_vm.h", "spec_paxos.h", "engine_pairs.h"], 7: ["spec_paxos.h"], 1: ["spec_pluscal.h"], 2: ["spec_raft.h"],
           3: ["spec_raft.h"], 4: ["spec_raft.h"], 5: ["spec_ssi.h", "engine_pairs.h"], 6: ["spec_vm.h"]}
    jobs, objs = [], []
    for tu in range(8):
        obj = OUT / f"engine_tu{tu}.o"
        objs.append(obj)
        if force or _stale(obj, base + [CSRC / h for h in own[tu]]):
            jobs.append((obj, subprocess.Popen(common + ["-x", "hip", f"-DMC_TU={tu}", "-c", str(CSRC / "engine.hip"), "-o", str(obj)])))
    host_deps = {"frontend": ["frontend.cpp", "pcal.h", "spec_vm.h", "mc_common.h", "tlaeval.h"], "pcal": ["pcal.cpp", "pcal.h"],
                 "tlaeval": ["tlaeval.cpp", "tlaeval.h"],
                 "pcal_compile": ["pcal_compile.cpp", "pcal.h", "spec_vm.h", "mc_common.h"],
                 "pcal_codegen": ["pcal_codegen.cpp", "pcal.h", "spec_vm.h", "mc_common.h"]}
    for name, dd in host_deps.items():
        obj = OUT / f"{name}.o"
        objs.append(obj)
        if force or _stale(obj, [CSRC / d for d in dd] + [PKG.parent / "include" / "tlamc.h"]):
            jobs.append((obj, subprocess.Popen(common + ["-x", "hip", "-c", str(CSRC / f"{name}.cpp"), "-o", str(obj)])))
    # the hip-rccl back-end: host code over the public C ABI, HIP runtime and RCCL
    obj = OUT / "shard_rccl.o"
    objs.append(obj)
    if force or _stale(obj, [CSRC / "shard_rccl.cpp", CSRC / "shard_loop.h", PKG.parent / "include" / "tlamc.h"]):
        jobs.append((obj, subprocess.Popen(common + ["-x", "hip", "-c", str(CSRC / "shard_rccl.cpp"), "-o", str(obj)])))
    for obj, pr in jobs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed for {obj.name}")
    if force or jobs or not LIB.exists():
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o in objs] + ["-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    main = CSRC / "mc_main.cpp"
    if main.exists() and (force or _stale(CLI, deps + [LIB])):
        cmd = [hipcc, "-O2", "-std=c++17", "-I", str(PKG.parent / "include"), "-o", str(CLI), str(main),
               "-L", str(OUT), "-ltlamc", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    # profiling aid, not part of the product: known-byte access patterns to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE (profiles/calib)
    calib_src = PKG.parent / "profiles" / "calib" / "calib_fetch.hip"
    calib = OUT / "calib_fetch"
    if calib_src.exists() and (force or _stale(calib, [calib_src])):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-Wno-unused-result", "-o", str(calib), str(calib_src)], check=True)
    # ... and the random-probe rate against the working set / the cost of partitioning candidates by table region (profiles/calib/probe_locality.hip)
    loc_src = PKG.parent / "profiles" / "calib" / "probe_locality.hip"
    loc = OUT / "probe_locality"
    if loc_src.exists() and (force or _stale(loc, [loc_src])):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-Wno-unused-result", "-o", str(loc), str(loc_src)], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
