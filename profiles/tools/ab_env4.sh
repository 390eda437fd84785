#!/bin/bash
# A/B of environment settings on ONE box, interleaved, extra bench flags in ABFLAGS:
#   gpurun -- 'ABFLAGS="--meshes-per-gpu 8" bash profiles/tools/ab_env4.sh "A=1" "A=0"'
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "$@"; do
    ms=$(env ${cfg//,/ } python3 $R/bench.py --cpu-budget 0 --skip-fp32-form --profile-steps 0 --skip-drop-in --skip-copy-rate --min-time 1.2 --graph list $ABFLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$cfg : $ms"
  done
done
