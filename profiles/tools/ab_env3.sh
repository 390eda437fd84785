#!/bin/bash
# A/B of environment settings on ONE box, interleaved:  gpurun -- 'bash profiles/tools/ab_env3.sh "A=1 B=2" "A=2 B=2" ...'
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python3 $R/bench.py --cpu-budget 0 --skip-fp32-form --skip-drop-in --profile-steps 0 --min-time 1.2 --graph list 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$cfg : $ms"
  done
done
