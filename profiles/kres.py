#!/usr/bin/env python
"""profiles/kres.py OBJ [FILTER] — registers, spills, scratch, LDS and code bytes of the kernels in a hipcc object (gfx950 code object
inside its fat binary).  A build-time check for the hot kernels: VGPRs <= 128 (4 wavefronts per SIMD), LDS <= 40960 (4 workgroups per CU)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "expand_family")
    with tempfile.TemporaryDirectory() as d:
        fat, co = Path(d) / "f.bin", Path(d) / "k.co"
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True)
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", str(co)], capture_output=True, text=True).stdout
        syms = subprocess.run([LLVM + "llvm-readelf", "-sW", str(co)], capture_output=True, text=True).stdout
    size = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    cur = {}
    out = []
    for ln in notes.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(.*)", ln)
        if not m:
            continue
        k, v = m.groups()
        if k == "name":
            cur["name"] = v
        elif k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
            cur[k] = int(v)
        if k == "wavefront_size":   # last key of a kernel's record
            if "name" in cur:
                out.append(cur)
            cur = {}
    dem = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    for k in out:
        if flt in k["name"]:
            nm = re.sub(r"\(.*", "", dem(k["name"]))
            print(f"{nm:70s} vgpr {k.get('vgpr_count')} (spill {k.get('vgpr_spill_count')})  sgpr spill {k.get('sgpr_spill_count')}  scratch {k.get('private_segment_fixed_size')}  "
                  f"LDS {k.get('group_segment_fixed_size')}  code {size.get(k['name'], '?')} B")
    for n, sz in size.items():
        if flt in n and not any(n == k["name"] for k in out):
            print(f"  (function) {re.sub(r'[(].*', '', dem(n))[:90]}  code {sz} B")


if __name__ == "__main__":
    main()
