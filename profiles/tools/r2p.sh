R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { echo "--- $*"; env "$@" timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2975$((RANDOM%9)) $R/profiles/tools/launch_cost.py 2>&1 | grep "ms/step" | tail -1; }
run EARLY=1
run EARLY=1 GPU_MAX_HW_QUEUES=8
run EARLY=1 GPU_MAX_HW_QUEUES=16
run EARLY=1 GFV_SIDE_PRIO=-1
run EARLY=1 GFV_SIDE_PRIO=10
run EARLY=1 GPU_MAX_HW_QUEUES=2
echo "--- plain, GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 300 python3 $R/profiles/tools/launch_cost.py 2>&1 | grep "ms/step" | tail -1
echo "--- plain, GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 300 python3 $R/profiles/tools/launch_cost.py 2>&1 | grep "ms/step" | tail -1
