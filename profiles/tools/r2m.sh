R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29721 $R/profiles/tools/dist_overhead.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -12
