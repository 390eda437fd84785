#!/bin/bash
# ms / step of the bench step under several environment settings, alternating, N repetitions on ONE box:
#   gpurun -- 'bash profiles/tools/ab_env2.sh 3 "GFV_COLCHAIN=0" "GFV_COLCHAIN=1 GFV_COLCHAIN_LITE=0" ...'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
n=$1; shift
for rep in $(seq $n); do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --graph list --min-time 1.0 --cpu-budget 0 --profile-steps 0 --skip-fp32-form --skip-drop-in 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep  [$cfg]  $ms ms/step"
  done
done
