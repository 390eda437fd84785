#!/bin/bash
# Collect the round's judged measurements on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh r02'
# (r03: + kernel_stats_in_step.json - the in-step kernel averages bench.py prices as frac_in_step -, hbm_pmc_b8.txt)
# writes under gpurun_out/<tag>/ : bench.json (the default bench.py line, with cpu_baseline), bench_b8.json (8 meshes per
# GPU: BASELINE config 4's per-GPU load), bench_cavity.json (config 2: 71 x 71 lid-driven cavity), bench_poly.json (config 5's shape: the reference's polygon example mesh,
# unsteady solve loop with a time advance every 20 iterations), kernel_stats.csv
# (rocprofv3 --kernel-trace --stats of the default bench command), hbm_pmc.txt + pmc_traffic.json (separate --pmc FETCH_SIZE /
# WRITE_SIZE passes of a short eager bench run), parity_fp64.txt (HIP path and fp32 oracle against the float64 oracle, from
# the -m gpu tests), pytest.log.  Copy the results into profiles/ afterwards (gpurun_out/ is scratch).
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r02}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
GFV_RCCL_REPORT=$O/rccl_two_ranks.txt GFV_PARITY_REPORT=$O/parity_fp64.txt timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python3 $R/bench.py --meshes-per-gpu 8 --cpu-budget 0 --steps 10 --warmup 4 > $O/bench_b8.json 2> $O/bench_b8.err
timeout 900 python3 $R/bench.py --workload cavity --cells 5041 --cpu-budget 8 > $O/bench_cavity.json 2> $O/bench_cavity.err
timeout 900 python3 $R/bench.py --workload poly --cpu-budget 8 > $O/bench_poly.json 2> $O/bench_poly.err
# one form of the step only (command-list replay, split-fp16 products): what the in-step averages are taken from
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --cpu-budget 0 --graph list --skip-fp32-form --profile-steps 0 --skip-copy-rate --skip-drop-in > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
python3 $R/profiles/instep_aggregate.py $O/kernel_stats.csv $O/kernel_stats_in_step.json $R/gen-fvgn-steady_amd/gfv/libgfv.so.srchash > $O/kernel_stats_in_step.txt
# round 6: the same at BASELINE config 4's per-GPU load (8 meshes per GPU): kernel statistics of the two-queue step
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -- python3 $R/bench.py --meshes-per-gpu 8 --steps 10 --warmup 4 --cpu-budget 0 --graph list --skip-fp32-form --profile-steps 0 --skip-copy-rate --skip-drop-in > $O/prof_b8.log 2>&1
find $O/prof8 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_b8.csv \;
rm -rf $O/prof8
# PMC passes: eager launches only, split-fp16 form only; the JSON line says how many steps ran (steps_executed)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --graph off --min-time 0 --cpu-budget 0 --profile-steps 1 --skip-fp32-form --skip-copy-rate --skip-drop-in > $O/pmc_$c.log 2>&1
done
NSTEPS=$(python3 -c "import json,sys; print([json.loads(l) for l in open('$O/pmc_FETCH_SIZE.log') if l.startswith('{')][-1]['steps_executed'])")
python3 $R/profiles/pmc_aggregate.py $O $NSTEPS > $O/hbm_pmc.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# the same two passes with 8 meshes per GPU (BASELINE config 4's per-GPU load; the working set no longer fits the Infinity Cache)
mkdir -p $O/b8
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/b8/pmc_$c -- python3 $R/bench.py --meshes-per-gpu 8 --steps 3 --warmup 1 --graph off --min-time 0 --cpu-budget 0 --profile-steps 1 --skip-fp32-form --skip-copy-rate --skip-drop-in > $O/b8/pmc_$c.log 2>&1
done
NSTEPS8=$(python3 -c "import json,sys; print([json.loads(l) for l in open('$O/b8/pmc_FETCH_SIZE.log') if l.startswith('{')][-1]['steps_executed'])")
python3 $R/profiles/pmc_aggregate.py $O/b8 $NSTEPS8 > $O/hbm_pmc_b8.txt
rm -rf $O/b8/pmc_FETCH_SIZE $O/b8/pmc_WRITE_SIZE
# round 5: floor + slope of every kernel of the step over four mesh sizes (single stream), and one step's two-queue timeline on
# the headline mesh and on the 5 k-cell cavity
bash $R/profiles/tools/latency_floor.sh ${tag}_lf > /dev/null 2>&1
cp $R/gpurun_out/${tag}_lf/latency_floor.txt $O/latency_floor.txt
TL_TAG=${tag}_tl bash $R/profiles/tools/timeline.sh > $O/timeline.txt 2>&1
ABFLAGS="--workload cavity --cells 5041" TL_TAG=${tag}_tlc bash $R/profiles/tools/timeline.sh > $O/timeline_cavity.txt 2>&1
ABFLAGS="--meshes-per-gpu 8 --steps 10 --warmup 4 --min-time 1.5" TL_TAG=${tag}_tl8 bash $R/profiles/tools/timeline.sh > $O/timeline_b8.txt 2>&1
# round 6: where the host time of the drop-in iteration goes (cProfile of the reference driver's call sequence, 5 k-cell cavity)
timeout 600 python3 $R/profiles/tools/dropin_profile.py 5041 gfv > $O/dropin_host_profile.txt 2>&1
tail -3 $O/pytest.log
head -c 400 $O/bench.json; echo
tail -4 $O/hbm_pmc.txt
