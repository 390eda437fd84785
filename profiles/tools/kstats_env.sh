#!/bin/bash
# per-kernel rocprofv3 stats of bench.py under several environment settings on ONE box:
#   gpurun -- 'bash profiles/tools/kstats_env.sh "GFV_RECOMPUTE=1" "GFV_RECOMPUTE=0"'      (extra bench flags: KSTATS_FLAGS)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  O=$R/gpurun_out/kstats_env_$i
  rm -rf $O; mkdir -p $O
  export $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --cpu-budget 0 --min-time 0.5 --graph list --skip-fp32-form --skip-drop-in --profile-steps 0 $KSTATS_FLAGS > $O/bench.json 2> $O/err.txt
  for kv in $cfg; do unset ${kv%%=*}; done
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== $cfg   $(python3 -c "import json,sys; print(json.loads(open('$O/bench.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null) ms/step"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print("%-78s %6s %9.1f %6s" % (r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:78], r["Calls"], float(r["AverageNs"])/1000, r["Percentage"][:5]))
PY
  cp "$f" $O/kernel_stats.csv; rm -rf $O/prof
done
