#!/bin/bash
# VGPR / SGPR / scratch / LDS use of every kernel of one source file (device-only compile, no GPU needed):
#   tools/kernel_resources.sh ov2slam_amd/csrc/lk3.hip [extra -D flags]
SRC=$1; shift
OUT=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S "$@" "$SRC" -o $OUT/k.s 2>/dev/null || exit 1
python3 - $OUT/k.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", body) or [None, "?"])[1]
    print("%-60s next_free_vgpr %-4s sgpr %-4s scratch %-5s lds %-6s" % (name[:60], g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
for m in re.finditer(r"; Function info:.*?\n(.*?)\n\n", txt, re.S):
    pass
# per-kernel occupancy / spill comments emitted by the backend
for m in re.finditer(r"; Kernel info:.*?; Occupancy: (\d+).*?\n", txt, re.S):
    pass
for blk in re.findall(r"(; NumVgprs: \d+.*?; Occupancy: \d+)", txt, re.S):
    d = dict(re.findall(r"; (\w+): (\d+)", blk))
    print("   NumVgprs %s  ScratchSize %s  Occupancy %s  spills(v) %s" % (d.get("NumVgprs"), d.get("ScratchSize"), d.get("Occupancy"), d.get("VGPRSpill", "?")))
PY
rm -rf $OUT
