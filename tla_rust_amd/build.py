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
    if force or _stale(LIB, [d for d in deps if d.name != "mc_main.cpp"]):
        # engine.hip is compiled once per group of specs (MC_TU = 1..5) plus once for the C ABI (MC_TU = 0), in
        # parallel: the unrolled per-spec kernels dominate compile time
        common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
                  "-I", str(PKG.parent / "include")]
        jobs = []
        for tu in range(6):
            obj = OUT / f"engine_tu{tu}.o"
            jobs.append((obj, subprocess.Popen(common + ["-x", "hip", f"-DMC_TU={tu}", "-c", str(CSRC / "engine.hip"), "-o", str(obj)])))
        fobj = OUT / "frontend.o"
        jobs.append((fobj, subprocess.Popen(common + ["-x", "hip", "-c", str(CSRC / "frontend.cpp"), "-o", str(fobj)])))
        for obj, pr in jobs:
            if pr.wait() != 0:
                raise RuntimeError(f"hipcc failed for {obj.name}")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(o) for o, _ in jobs]
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
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
