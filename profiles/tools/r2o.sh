R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in "" 1 2; do
echo "--- EARLY=$e"; EARLY=$e timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2974$((RANDOM%9)) $R/profiles/tools/launch_cost.py 2>&1 | grep "ms/step" | tail -3
done
