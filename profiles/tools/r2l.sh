R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) bench.py under torch.distributed.run, one rank, backend nccl (= RCCL): the driver's N > 1 code path on the one GPU here
GFV_DIST_FORCE=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-budget 0 --min-time 0.5 > $O/bench_nccl1.json 2> $O/bench_nccl1.err
echo "nccl1 rc=$?"; tail -c 600 $O/bench_nccl1.json | head -c 600; echo
python3 -c "
import json
d=[json.loads(l) for l in open('$O/bench_nccl1.json') if l.startswith('{')][-1]
print('nccl1', d['value'], d['ms_per_step'], d['step_modes'], d['rccl_ranks'], d['dist_backend'], d['distinct_gpus'])
"
# (2) two ranks sharing the GPU (gloo; RCCL refuses two ranks on one device): multi-rank control flow of bench.py
GFV_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 $R/bench.py --gpus 2 --steps 10 --warmup 3 --cpu-budget 0 --min-time 0.3 --allow-shared-gpu --cells 12000 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
echo "gloo2 rc=$?"
python3 -c "
import json
d=[json.loads(l) for l in open('$O/bench_gloo2.json') if l.startswith('{')][-1]
print('gloo2', d['value'], d['ms_per_step'], d['step_modes'], d['rccl_ranks'], d['dist_backend'], d['distinct_gpus'], d['n_gpus'])
"
tail -5 $O/bench_gloo2.err
# (3) refusing more ranks than GPUs
GFV_DIST_BACKEND=gloo timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 $R/bench.py --gpus 2 --steps 2 --warmup 1 --cpu-budget 0 > $O/refuse.json 2> $O/refuse.err; echo "refuse rc=$? (non-zero expected)"; grep -m1 "WORLD_SIZE" $O/refuse.err
