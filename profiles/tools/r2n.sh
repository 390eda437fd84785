R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
echo "--- plain python"; timeout 300 python3 $R/profiles/tools/launch_cost.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -4
echo "--- torch.distributed.run"; timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29731 $R/profiles/tools/launch_cost.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -10
echo "--- torch.distributed.run, OMP_NUM_THREADS=16"; OMP_NUM_THREADS=16 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29732 $R/profiles/tools/launch_cost.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
