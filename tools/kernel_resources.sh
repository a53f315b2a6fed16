#!/bin/bash
# tools/kernel_resources.sh <file.hip> [name filter] [extra -D flags ...] -- registers, spills, LDS and occupancy of the kernels of one translation unit
# (device-only compile with the product flags, -Rpass-analysis=kernel-resource-usage); runs without a GPU.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$1"; FILTER="${2:-.}"; shift; shift || true
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize \
    -I"$ROOT/include" "$@" -x hip --cuda-device-only -c "$SRC" -o "$OUT/k.co" -Rpass-analysis=kernel-resource-usage 2> "$OUT/remarks" || { grep -v remark: "$OUT/remarks" | head -40; exit 1; }
python3 - "$OUT/remarks" "$FILTER" <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
t = "\n".join(l for l in t.split("\n") if "remark:" in l)      # drop the echoed source lines
pat = (r"Function Name: (\S+).*?\n.*?TotalSGPRs: (\d+).*?\n.*?VGPRs: (\d+).*?\n.*?AGPRs: (\d+).*?\n.*?ScratchSize \[bytes/lane\]: (\d+).*?\n.*?\n"
       r".*?Occupancy \[waves/SIMD\]: (\d+).*?\n.*?SGPRs Spill: (\d+).*?\n.*?VGPRs Spill: (\d+).*?\n.*?LDS Size \[bytes/block\]: (\d+)")
for m in re.finditer(pat, t):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    if re.search(sys.argv[2], name):
        print("%-64s sgpr %3s vgpr %3s agpr %3s scratch %4s occ %s sgpr-spill %3s vgpr-spill %3s lds %6s" % ((name[-64:],) + m.groups()[1:]))
PY
rm -rf "$OUT"
