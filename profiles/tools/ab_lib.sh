#!/bin/bash
# A/B of library builds (and environment settings) on ONE box, interleaved.  Each argument: "<variant|base> [ENV=VAL ...]"
#   gpurun -- 'bash profiles/tools/ab_lib.sh "base" "nopipe" "base GFV_RECOMPUTE=0"'      (extra bench flags: ABFLAGS)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "$@"; do
    set -- $cfg; v=$1; shift; envs="$@"
    lib=$R/gen-fvgn-steady_amd/gfv/libgfv.so
    [ $v != base ] && lib=$R/profiles/tools/variants/libgfv_$v.so
    ms=$(env GFV_LIB=$lib $envs python3 $R/bench.py --cpu-budget 0 --skip-fp32-form --skip-drop-in --profile-steps 0 --min-time 1.2 --graph list $ABFLAGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$cfg : $ms"
  done
done
