"""Command-list replay of the training step: the launch sequence of ONE eager step, recorded at the C-ABI boundary and
replayed from a flat list.

Why: one step is ~245 launches on two HIP streams.  Launched eagerly from Python (struct filling, allocation, plan
lookups: ~18 us of host time per launch) the host keeps up with the device only just, and any burst of side-stream
launches starves the main queue (rocprofv3 --kernel-trace: 100-260 us holes).  A hipGraph of the same step replays
with no host work but overlaps its two branches much less (5.3 ms against 4.9 ms eager).  The command list keeps the
eager execution model - the same two streams, the same event edges - and cuts the host cost per launch to one ctypes
call (~1.5 us): every `lib.gfv_*` launch of the recorded step is stored as (function, converted arguments), every
stream fork / join and the few tensor copies as (callable, arguments, stream), and `replay()` walks the list.

What makes the recorded pointers valid on replay: the recorded step allocates from a private `torch.cuda.MemPool`
that stays alive with the list, so nothing else is ever placed in that memory; inside the pool the step's own
free / re-use pattern repeats exactly, because replay issues the same commands in the same order on the same
streams.  Tensors the caller reads after a step (losses, fields) are the recorded step's tensors.
Host-side decisions (shapes, plan tables, which kernel form) are made once, at record time: a list belongs to one
batch, one parameter set and one (accumulate, distributed) mode - like a captured hipGraph - and TrainStep drops it
when any of those change (Engine.capture_signature).
"""
from __future__ import annotations

import torch

# entry points that return a value to the host and launch nothing: never recorded
import os

# GFV_CMDLIST_NATIVE=0: the round-2 form - every library call of the recorded step kept as (ctypes function, arguments) and
# re-issued from a Python loop.  Default (round 5): the library notes its own kernel launches while the step is recorded
# (include/gfv.h gfv_record_*, csrc/gfv_launch.h) and replays them from C - one hipLaunchKernelGGL per launch, no ctypes call,
# no argument checks, no kernel-family choice; only the few host-side commands (tensor copies, collectives) stay in the Python
# list, each with the number of native launches that precede it.  Host time to issue the 5 k-cell cavity step: 1.22 -> see
# profiles/r05_launch_cost.txt.
NATIVE = os.environ.get("GFV_CMDLIST_NATIVE", "1") != "0"

_QUERIES = frozenset((
    "gfv_abi_version", "gfv_struct_size", "gfv_rowtile_tiles", "gfv_rowtile_last_path", "gfv_dw_chunks", "gfv_dw_slabs",
    "gfv_linear_dw_workspace_floats", "gfv_dw_multi_workspace_floats", "gfv_f16split_enabled", "gfv_set_f16split", "gfv_set_f16split_thread", "gfv_hidden_size",
    "gfv_weight_image_bytes", "gfv_normalizer_blocks", "gfv_slice_softmax_bwd_blocks", "gfv_profile_enable",
    "gfv_profile_collect", "gfv_profile_reset", "gfv_profile_set_sizes", "gfv_status_flags", "gfv_status_mirror", "gfv_prep_workspace_bytes", "gfv_get_limit", "gfv_set_limit", "gfv_limit_name", "gfv_rowtile_dw_partials", "gfv_rowtile_dw_partials_m",
    "gfv_rowtile_fuses_dw", "gfv_graph_norm_workspace_bytes", "gfv_plan_create", "gfv_plan_destroy", "gfv_plan_table", "gfv_plan_sizes",
    "gfv_record_begin", "gfv_record_count", "gfv_record_end", "gfv_record_length", "gfv_record_replay", "gfv_record_free",
    "gfv_record_delay_side"))


class CommandList:
    def __init__(self):
        self.cmds = []          # (callable, args, stream or None[, native launches issued before it])
        self.pool = None
        self.main = None        # the stream the step was recorded on
        self.keep = []          # results of recorded calls that returned tensors (kept alive with the list)
        self.native = 0         # handle of the library's own list of this step's launches (0: Python-level list)
        self.n_native = 0

    def __len__(self):
        return len(self.cmds) + self.n_native

    def __del__(self):
        if self.native:
            try:
                from . import lib as L
                L.load(raw=True).gfv_record_free(self.native)
            except Exception:
                pass

    def replay(self):
        cur = torch.cuda.current_stream()
        if cur != self.main:
            raise RuntimeError("a command list replays on the stream it was recorded on")
        if self.native:
            from . import lib as L
            lib, h, pos = L.load(raw=True), self.native, 0
            for fn, args, st, idx in self.cmds:
                if idx > pos:
                    L.check(lib.gfv_record_replay(h, pos, idx), "gfv_record_replay")
                    pos = idx
                if st is None or st == cur:
                    fn(*args)
                else:
                    with torch.cuda.stream(st):
                        fn(*args)
            if self.n_native > pos:
                L.check(lib.gfv_record_replay(h, pos, self.n_native), "gfv_record_replay")
            return
        for fn, args, st in self.cmds:
            if st is None or st == cur:
                fn(*args)
            else:
                with torch.cuda.stream(st):
                    fn(*args)


_ACTIVE = None   # the CommandList being recorded, or None


def active():
    return _ACTIVE


_IN_CALL = 0     # depth of cmdlist.call frames: torch ops issued inside them ARE recorded (as the call)


def call(fn, *args):
    """Run a host-side callable that enqueues device work (tensor copy, stream wait) and, while recording, note it with
    the stream it ran under."""
    global _IN_CALL
    _IN_CALL += 1
    try:
        r = fn(*args)
    finally:
        _IN_CALL -= 1
    if _ACTIVE is not None:
        if _ACTIVE.native:
            from . import lib as L
            _ACTIVE.cmds.append((fn, args, torch.cuda.current_stream(), L.load(raw=True).gfv_record_count()))
        else:
            _ACTIVE.cmds.append((fn, args, torch.cuda.current_stream()))
    return r


# torch ops that enqueue nothing on the device: allocation from the recording pool, views
_NO_KERNEL = frozenset((
    "aten::empty.memory_format", "aten::empty_strided", "aten::empty_like", "aten::new_empty", "aten::new_empty_strided",
    "aten::detach", "aten::alias", "aten::lift_fresh", "aten::_unsafe_view", "aten::is_pinned", "aten::resize_",
    "aten::set_.source_Storage", "aten::set_.source_Storage_storage_offset", "aten::set_.source_Tensor", "aten::_local_scalar_dense",
    "aten::is_same_size", "aten::stride", "aten::sym_size", "aten::sym_stride", "aten::sym_numel", "aten::sym_storage_offset"))


def _guard_mode():
    """A torch op that launches device work while a step is being recorded - and does not go through `call` or the library
    proxy - runs now and is silently MISSING from every replay.  The guard turns that into an error at record time
    (GFV_CMDLIST_GUARD=0 disables it)."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils._pytree import tree_leaves

    class Guard(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            if _IN_CALL == 0:
                name = func._schema.name + ("." + func._overloadname if func._overloadname and func._overloadname != "default" else "")
                if name not in _NO_KERNEL and not getattr(func, "is_view", False):
                    leaves = tree_leaves((args, kwargs, out))
                    if any(isinstance(t, torch.Tensor) and t.is_cuda for t in leaves):
                        raise RuntimeError(
                            f"cmdlist: torch op {name} enqueued device work outside cmdlist.call while a step was being "
                            "recorded; it would be dropped on replay (wrap it in cmdlist.call, move it to set-up, or run the "
                            "step with use_graph=False)")
            return out
    return Guard()


class _RecordingLib:
    """Stands in for the ctypes library while a step is recorded: launches go through and are noted."""

    def __init__(self, cdll, target):
        self._cdll, self._target, self._cache = cdll, target, {}

    def __getattr__(self, name):
        hit = self._cache.get(name)
        if hit is not None:
            return hit
        f = getattr(self._cdll, name)
        if name in _QUERIES:
            self._cache[name] = f
            return f
        cmds = self._target.cmds
        if self._target.native:
            # the library notes its own launches (gfv_record_begin is open on this thread): nothing to keep here
            self._cache[name] = f
            return f

        def launch(*args):
            rc = f(*args)
            cmds.append((f, args, None))
            return rc
        self._cache[name] = launch
        return launch


class record:
    """Context manager: `with record() as cl:` runs one step eagerly, allocating from a private pool, and leaves the
    launch sequence in `cl`."""

    def __init__(self):
        self.cl = CommandList()

    def __enter__(self):
        global _ACTIVE
        from . import lib as L
        if _ACTIVE is not None:
            raise RuntimeError("nested recording")
        self.cl.main = torch.cuda.current_stream()
        self.cl.pool = torch.cuda.MemPool()
        self._ctx = torch.cuda.use_mem_pool(self.cl.pool)
        self._ctx.__enter__()
        _ACTIVE = self.cl
        if NATIVE:
            rc = L.load(raw=True).gfv_record_begin()
            if rc != 0:
                # (a recording is already open on this thread at the library level): leave no half-entered state behind
                _ACTIVE = None
                self._ctx.__exit__(None, None, None)
                L.check(rc, "gfv_record_begin")
            self.cl.native = -1   # (open; the handle arrives at the end)
        L._recording = _RecordingLib(L.load(raw=True), self.cl)
        self._guard = None
        if __import__("os").environ.get("GFV_CMDLIST_GUARD", "1") != "0":
            self._guard = _guard_mode()
            self._guard.__enter__()
        return self.cl

    def __exit__(self, *exc):
        global _ACTIVE
        from . import lib as L
        if self._guard is not None:
            self._guard.__exit__(*exc)
        L._recording = None
        _ACTIVE = None
        if self.cl.native:
            lib = L.load(raw=True)
            self.cl.n_native = max(lib.gfv_record_count(), 0)
            self.cl.native = int(lib.gfv_record_end())
            # issue order: side-stream bursts behind a few of the main stream's launches (include/gfv.h gfv_record_delay_side) -
            # only for lists without host-side commands in between (their positions count native launches)
            delay = int(os.environ.get("GFV_SIDE_DELAY", "0"))
            if delay > 0 and self.cl.native and not self.cl.cmds and not (exc and exc[0] is not None):
                self.cl.delayed = lib.gfv_record_delay_side(self.cl.native, self.cl.main.cuda_stream, delay)
            if exc and exc[0] is not None and self.cl.native:
                lib.gfv_record_free(self.cl.native)
                self.cl.native = 0
        self._ctx.__exit__(*exc)
        return False
