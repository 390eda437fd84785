#!/bin/bash
# Only the HBM-traffic PMC passes of profiles/collect.sh (FETCH_SIZE | WRITE_SIZE, separate passes, eager launches), with extra
# environment for the profiled command:   gpurun -- 'bash profiles/tools/pmc_only.sh r04rc GFV_RECOMPUTE=1'
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmc}; shift
O=$R/gpurun_out/$tag
mkdir -p $O
for kv in "$@"; do export $kv; done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --graph off --min-time 0 --cpu-budget 0 --profile-steps 1 --skip-fp32-form --skip-drop-in --skip-copy-rate $PMCFLAGS > $O/pmc_$c.log 2>&1
done
NSTEPS=$(python3 -c "import json,sys; print([json.loads(l) for l in open('$O/pmc_FETCH_SIZE.log') if l.startswith('{')][-1]['steps_executed'])")
python3 $R/profiles/pmc_aggregate.py $O $NSTEPS > $O/hbm_pmc.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
echo "== $tag $@"; head -12 $O/hbm_pmc.txt | tail -9; tail -1 $O/hbm_pmc.txt
