#!/bin/bash
# env-variable experiments on ONE box: each argument is a "NAME=VALUE[,NAME=VALUE...]" set, compared with the plain run
#   gpurun -- 'bash profiles/tools/ab_env.sh GFV_DW_WGS_SMALL=256 GFV_SLICE_CHUNK=128'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab_env
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in base "$@"; do
    envs=""
    [ "$v" != base ] && envs=$(echo $v | tr ',' ' ')
    tag=$(echo $v | tr -c 'A-Za-z0-9=_' '_')
    env $envs timeout 300 python3 $R/bench.py --cpu-budget 0 --min-time 1.5 --graph list --skip-fp32-form --skip-drop-in > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
    python3 -c "
import json
d=json.load(open('$O/${tag}_$rep.json'))
print('$v', $rep, d['ms_per_step'])
"
  done
done
