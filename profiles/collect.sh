#!/bin/bash
# Collect the round's judged measurements on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'profiles/collect.sh r01'
# writes under gpurun_out/<tag>/ : bench.json (default bench.py line), kernel_stats.csv (rocprofv3 --kernel-trace
# --stats of the same command), hbm_pmc.txt + pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes).
# Copy the results into profiles/ afterwards (gpurun_out/ is scratch).
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r01}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --cpu-budget 0 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --cpu-budget 0 --profile-steps 1 > $O/pmc_$c.log 2>&1
done
python3 $R/profiles/pmc_aggregate.py $O > $O/hbm_pmc.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/hbm_pmc.txt
