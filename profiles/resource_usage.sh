#!/bin/bash
# registers / scratch / LDS / occupancy of every engine kernel of one translation unit (MC_TU: 3 = raft 3 servers, 4 = raft 5, 5 = SSI ...)
# usage: profiles/resource_usage.sh 3 [extra -D flags]
TU=${1:-3}; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -I include -x hip -DMC_TU=$TU "$@" \
    -c tla_rust_amd/csrc/engine.hip -o /tmp/tu$TU.o -Rpass-analysis=kernel-resource-usage 2>/tmp/tu$TU.remarks || { grep -m20 error /tmp/tu$TU.remarks; exit 1; }
python3 - "$TU" <<'PY'
import re, subprocess, sys
t = open(f'/tmp/tu{sys.argv[1]}.remarks').read()
pat = (r'Function Name: (\S+).*?\n(?:.*?\n)*?.*?TotalSGPRs: (\d+).*?\n.*?VGPRs: (\d+).*?\n(?:.*?\n)*?.*?ScratchSize \[bytes/lane\]: (\d+).*?\n(?:.*?\n)*?'
       r'.*?Occupancy \[waves/SIMD\]: (\d+).*?\n.*?SGPRs Spill: (\d+).*?\n.*?VGPRs Spill: (\d+).*?\n.*?LDS Size \[bytes/block\]: (\d+)')
for m in re.finditer(pat, t):
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    if 'rocprim' in name:
        continue
    print(f"{name[:100]:100s} SGPR {m.group(2):>3} VGPR {m.group(3):>3} scratch {m.group(4):>4} occ {m.group(5)} sspill {m.group(6):>3} vspill {m.group(7):>3} lds {m.group(8)}")
PY
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading /tmp/tu$TU.o >/dev/null 2>&1
