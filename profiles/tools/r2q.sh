R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { echo "--- $*"; env "$@" timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2976$((RANDOM%9)) $R/profiles/tools/launch_cost.py 2>&1 | grep "ms/step" | tail -1; }
run EARLY=1
run EARLY=2
run EARLY=1 GPU_MAX_HW_QUEUES=8
echo "--- plain"; timeout 300 python3 $R/profiles/tools/launch_cost.py 2>&1 | grep "ms/step" | tail -1
echo "--- bench under torchrun nccl"; GFV_DIST_FORCE=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29769 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --min-time 0.5 --skip-fp32-form --skip-drop-in 2>/dev/null | python3 -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print(d['value'], d['ms_per_step'], d['step_modes'], d['rccl_ranks'])"
