#!/bin/bash
# Host-thread placement A/B (round 6): bench.py with the loop's threads confined to one L3 (default) against --no-pin-host, interleaved
# on ONE box;  gpurun -- 'bash profiles/tools/ab_pin.sh [bench flags]'
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for pin in "" "--no-pin-host"; do
    python3 $R/bench.py --cpu-budget 0 --skip-fp32-form --profile-steps 0 --skip-copy-rate --min-time 1.2 --graph list $pin "$@" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); di=d['drop_in']
print('${pin:-pinned to one L3}', ':', 'TrainStep', d['ms_per_step'], 'ms | drop-in torch.optim.Adam', di['torch_adam']['ms_per_step'], '| gfv.optim.Adam', di['gfv_adam']['ms_per_step'], '| eager', di['torch_adam_eager']['ms_per_step'], '| cpus', d['host_threads']['cpus'])"
  done
done
