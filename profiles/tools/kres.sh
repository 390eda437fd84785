#!/bin/bash
# register / LDS / spill figures of every kernel of one source:  kres.sh csrc/colchain.hip   (device-only assembly, gfx950)
src=$1; shift
flags=$(head -1 "$src" | sed -n 's,^// gfv-build-flags:,,p')
out=/tmp/kres_$(basename "$src" .hip).s
[ -n "$KRES_REUSE" -a -f "$out" ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $flags "$@" --cuda-device-only -S "$src" -o "$out" 2>/dev/null || exit 1
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk).group(1)
    name = g("name")
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").split("(")[0]
    print("%-70s vgpr %3s agpr %3s spill %3s sgpr %3s sspill %3s lds %6s scratch %4s" % (
        dem[:70], g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size")))
PY
