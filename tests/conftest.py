import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gen-fvgn-steady_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle inside the tests is launch-bound eager PyTorch: on the GPU box's 256-thread host it is fastest at 16
    # intra-op threads (bench.py cpu_baseline: 4.3 / 3.5 / 3.6 / 5.2 / 188 s per 50 k-cell step at 8 / 16 / 32 / 64 / 256) and
    # 2 - 3 x slower at torch's default there; GFV_TEST_THREADS overrides
    try:
        import torch
        n = int(os.environ.get("GFV_TEST_THREADS", "0")) or min(16, os.cpu_count() or 1)
        torch.set_num_threads(n)
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _clean_status_word(request):
    """GPU tests start from a cleared device status word (and mirror): a kernel test that overflows on purpose must not make an
    unrelated TrainStep / NNmodel test - which now READ the word (gfv.lib.raise_on_status) - raise."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        from gfv import lib as L
        flags = C.c_int32(0)
        L.load(raw=True).gfv_status_flags(C.byref(flags))
        if L._status_word is not None:
            L._status_word[0] = 0
    yield


@pytest.fixture
def gfv_limits():
    """`gfv_limits(GFV_CBWD_MAX_M=100000)`: move dispatch limits of the library (csrc/gfv_limits.h, gfv_set_limit) for one test."""
    from gfv import lib as L
    stack = []

    def move(**kw):
        ctx = L.limits(**kw)
        ctx.__enter__()
        stack.append(ctx)
    yield move
    for ctx in reversed(stack):
        ctx.__exit__(None, None, None)
