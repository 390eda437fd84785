#!/bin/bash
# per-kernel rocprofv3 stats of bench.py for several library builds on ONE box:  kstats.sh <name1> <name2> ...  (base = libgfv.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/gen-fvgn-steady_amd/gfv/libgfv.so
  [ $v != base ] && lib=$R/gen-fvgn-steady_amd/gfv/libgfv_$v.so
  export GFV_LIB=$lib
  O=$R/gpurun_out/kstats_$v
  rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --cpu-budget 0 --min-time 0.5 --graph list --skip-fp32-form --skip-drop-in --profile-steps 1 > $O/bench.json 2> $O/err.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:34]:
    print("%-60s %6s %9.1f" % (r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1000))
PY
  cp "$f" $O/kernel_stats.csv; rm -rf $O/prof   # (the trace itself is tens of MB)
done
