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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for attempt in range(2):   # (one more attempt on another port when the launcher produced no result line at all)
        port = 29900 + (os.getpid() % 90) + 131 * attempt
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                           capture_output=True, text=True, timeout=900, env=env)
        lines = re.findall(r"RCCLRESULT graph=(\d) same=(\d) bucket=(\d+) allreduce=(\d+) early_pending=(\d+) backend=(\S+) world=(\d+)",
                           r.stdout)
        if lines:
            break
    assert r.returncode == 0 and len(lines) == 3 and "RCCLOK 1" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    for graph, same, bucket, allreduce, early, backend, world in lines:
        assert backend == "nccl" and world == "1"
        assert same == "1", "a one-rank RCCL step must be bit-identical to the non-distributed step"
        assert allreduce == "7"


@pytest.mark.parametrize("mode", ["eager", "list"])
def test_trainstep_n_ranks_through_rccl(mode):
    """N ranks, one mesh each, RCCL all-reduce of the gradient buckets and of the Normalizer statistics (SURVEY.md 8e;
    VERDICT r2 item 7, r3 item 3): all ranks end bit-identical and equal (to rounding) to one process stepping on the N-mesh
    batch.  N = the GPUs of the box, at most 8 (one GPU per rank), so an 8-GPU node exercises the whole node without a code
    change; `list`: command-list replay with the early bucket recorded into the list.  On a ONE-GPU box two ranks name GPU 0,
    which RCCL refuses ("Duplicate GPU detected" - a communicator takes every device once): the worker reports that, the refusal
    is recorded here and the test is skipped - the two-rank data path is then covered by the gloo form of the same worker
    (tests/test_operators_gpu.py::test_data_parallel_trainstep_two_ranks) and the RCCL calls by the one-rank test above."""
    import torch
    ngpu = torch.cuda.device_count()        # (counting devices does not initialise the GPU in this process)
    nproc = max(2, min(ngpu, 8))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_gpu_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", GFV_TEST_BACKEND="nccl", GFV_TEST_MODE=mode)
    for attempt in range(2):   # (one more attempt on another port when the launcher produced no result line of either kind)
        port = 29700 + (os.getpid() % 90) + (97 if mode == "list" else 0) + 211 * attempt
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                            "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                           capture_output=True, text=True, timeout=900, env=env)
        if "DPUNSUPPORTED" in r.stdout or "DPRESULT" in r.stdout:
            break
    refused = re.search(r"DPUNSUPPORTED rank=\d devices=(\d+) (.*)", r.stdout)
    if refused:
        assert refused.group(1) == "1", "RCCL refused the group although the box has one GPU per rank: " + refused.group(0)
        report = os.environ.get("GFV_RCCL_REPORT")
        if report:
            with open(report, "a") as f:
                f.write("two ranks on ONE GPU through RCCL: " + refused.group(0)[:700] + "\n")
        pytest.skip("RCCL takes every device once; this box has one GPU: " + refused.group(2)[:200])
    m = re.search(r"DPRESULT same=(\d) param_err=(\S+) norm_err=(\S+) backend=(\S+) gpus=(\d+) world=(\d+) mode=(\S+)", r.stdout)
    assert r.returncode == 0 and m, r.stdout[-1500:] + r.stderr[-3000:]
    assert m.group(4) == "nccl" and m.group(1) == "1" and int(m.group(6)) == nproc and m.group(7) == mode, m.group(0)
    assert float(m.group(2)) < 2e-5 and float(m.group(3)) < 1e-6, m.group(0)
