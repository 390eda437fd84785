#!/bin/bash
# Floor + slope of every kernel of the step (round 5): the SAME step on cavity meshes of 1 k / 5 k / 25 k / 75 k
# cells (N ~ cells node rows, E ~ 2 x cells edge rows), single stream (GFV_OVERLAP=0: every kernel's duration is its own), command-list
# replay, rocprofv3 --kernel-trace; per (kernel, grid size) the average duration at each mesh size and a straight-line fit.
#   gpurun -- 'bash profiles/tools/latency_floor.sh [tag]'   ->  gpurun_out/<tag>/latency_floor.txt   (extra env: LF_ENV="A=1 B=2")
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-latency_floor}
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GFV_OVERLAP=0
for kv in $LF_ENV; do export $kv; done
for cells in ${LF_CELLS:-1024 5041 25281 75076}; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$cells -- python3 $R/bench.py --workload cavity --cells $cells --cpu-budget 0 --min-time 0.25 --steps 20 --graph list --skip-fp32-form --skip-drop-in --profile-steps 0 --skip-copy-rate > $O/bench_$cells.json 2> $O/err_$cells.txt
  f=$(find $O/prof_$cells -name "*kernel_trace.csv" | head -1)
  python3 $R/profiles/tools/latency_floor.py --reduce "$f" $O/trace_$cells.json
  rm -rf $O/prof_$cells
done
python3 $R/profiles/tools/latency_floor.py --table $O > $O/latency_floor.txt
cat $O/latency_floor.txt
