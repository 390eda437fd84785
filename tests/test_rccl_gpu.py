"""RCCL (backend "nccl" on ROCm) on the one GPU of the test box: a one-rank process group drives the data-parallel
TrainStep through its real collectives.  Child process via torch.distributed.run - the
parent never touches the GPU before spawning and nothing is exec'ed from a GPU process."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_trainstep_through_rccl_one_rank_bit_identical():
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_gpu_worker.py")
    port = 29900 + (os.getpid() % 90)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = re.findall(r"RCCLRESULT graph=(\d) same=(\d) bucket=(\d+) allreduce=(\d+) early_pending=(\d+) backend=(\S+) world=(\d+)",
                       r.stdout)
    assert r.returncode == 0 and len(lines) == 3 and "RCCLOK 1" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    for graph, same, bucket, allreduce, early, backend, world in lines:
        assert backend == "nccl" and world == "1"
        assert same == "1", "a one-rank RCCL step must be bit-identical to the non-distributed step"
        assert allreduce == "5"
