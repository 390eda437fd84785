"""Hand-orchestrated forward / backward of the Gen-FVGN hot path on the HIP kernels of libgfv.

This is the host side of SURVEY.md 8 rows a-1 ... a-14: a fixed sequence of C-ABI launches per training step with
explicitly managed saved tensors (no autograd tape inside).  `FVMmodel.importer.NNmodel` wraps it in ONE
torch.autograd.Function so the reference's drivers (`loss.backward(); optimizer.step()`) work unchanged, and
`gfv.trainer.TrainStep` drives it directly (flat gradient buffer, fused Adam, hipGraph capture).

Parameters are addressed by the reference's state_dict names.
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import cmdlist
from . import lib as L
from . import ops
from .ops import LayerSpec, Seg

_MODE = {"explicit": 0, "implicit": 1, "imex": 2}


def _empty(dev, *shape):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def pick_concurrent_stream(priority=0, candidates=8):
    """A stream that really runs beside the current one.  The HIP runtime multiplexes streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, default 4) round-robin in creation order, and two streams on one hardware queue execute in order:
    with an RCCL process group created before the first step (what every multi-GPU run does) the weight-gradient stream
    landed on the main stream's queue and the step ran at its no-overlap time, 5.65 instead of 4.90 ms
    (profiles/tools/launch_cost.py).  So the choice is checked, not assumed: two spin kernels, one per stream - a candidate
    on its own queue finishes them in the time of one."""
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(priority=priority)
    main = torch.cuda.current_stream()
    cycles = 400_000

    def timed(side):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                torch.cuda._sleep(cycles)
        torch.cuda._sleep(cycles)
        if side is not None:
            main.wait_stream(side)
        e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    timed(None)
    alone = min(timed(None) for _ in range(2))
    first = None
    for _ in range(candidates):
        cand = torch.cuda.Stream(priority=priority)
        first = first or cand
        timed(cand)
        if min(timed(cand) for _ in range(2)) < 1.5 * alone:
            return cand
    return first   # none overlapped (one hardware queue for everything): the step still runs, at its no-overlap time


class GradStore:
    """Flat fp32 gradient storage with named views; every tensor starts on a 16-byte boundary, tensors keep the
    order they were given in (state_dict order), so the gradients of one MLP / Linear form one contiguous block."""

    def __init__(self, names, shapes, device, flat=None, skip=()):
        self.off, self.shape = {}, {}
        off = 0
        for n, sh in zip(names, shapes):
            k = 1
            for d in sh:
                k *= int(d)
            self.off[n], self.shape[n] = off, tuple(sh)
            off += (k + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device) if flat is None else flat
        self.skip = set(skip)

    def numel(self, n):
        k = 1
        for d in self.shape[n]:
            k *= d
        return k

    def view(self, n):
        if n in self.skip:
            return None
        return self.flat[self.off[n]:self.off[n] + self.numel(n)].view(self.shape[n])

    def block(self, first, last):
        """(offset, length) of the contiguous block first..last (inclusive, with padding)."""
        end = self.off[last] + (self.numel(last) + 3) // 4 * 4
        return self.off[first], end - self.off[first]


class Engine:
    def __init__(self, message_passing_num=3, integrator="imex", ncn_smooth=True, net="TransFVGN_v2", conserved_form=True,
                 order="2nd", hidden=128):
        # hidden_size of the model (utils/get_param.py:69): below 128 the parameters arrive zero-padded to the kernels' 128
        # columns (FVMmodel/padding.py) and the launches need the true width (LayerNorm statistics, attention scale)
        self.hidden = int(hidden)
        self._dw_ws, self._dw_ws_main = None, None
        self._pending = []
        self._defer_mode = os.environ.get("GFV_DEFER", "1") == "1"
        self._wt, self._wt_key, self._wt_live = {}, None, False
        self.mp = message_passing_num
        self.mode = _MODE[integrator]
        self.smooth = 1 if ncn_smooth else 0
        self.nc = 0 if conserved_form else 1   # residuals of the non-conserved form (FVscheme.py:276-511), row f4
        if order not in ("1st", "2nd", "3rd", "4th"):   # FVgrad.py:261-262
            raise ValueError("order must be specified in [\"1st\", \"2nd\", \"3rd\", \"4th\"]")
        self.order_terms = {"1st": 2, "2nd": 5, "3rd": 9, "4th": 14}[order]   # WLSQ Taylor terms (FVorder.py:23-72), row f4
        self.n_proc = 2 if net in ("TransFVGN_v2", "TransFVGN") else 1
        self.net = net
        # weight-gradient kernels run on a side stream, concurrently with the dX chain of the next layer block: both
        # kernel types leave the MFMA pipe half idle on their own and fit on a CU together (registers + LDS)
        self.overlap = os.environ.get("GFV_OVERLAP", "1") != "0"
        # EdgeBlock first layer factored through the nodes: W1 [x_s | x_r | e] = (W1a x)[s] + (W1b x)[r] + W1c e, the two
        # node-level products (and their adjoints / weight gradients) run over N rows instead of E = 3N
        self.factor = os.environ.get("GFV_EDGE_FACTOR", "1") != "0"
        # fp32 products of the chain kernels as split-fp16 on the f16 MFMA pipe (include/gfv.h, gfv_weight_images)
        self.f16split = os.environ.get("GFV_F16SPLIT", "1") != "0"
        # neighbour sums as the prologue of the chain launch that consumes them (gfv_seg_t.csr_rowptr) instead of launches
        # of their own: 24 fewer launches per step on the main stream
        self.csr_fuse = ops.csr_prologue_enabled()
        # ... where it pays (GFV_CSR_FUSE_MASK: 1 = the EdgeBlock's node-level projection, 2 = the NodeBlock MLP, 4 = the
        # per-side scatter of the factored EdgeBlock's adjoint).  Since the plain forward instantiation runs at 3 waves per
        # SIMD (tchain_fwd.hip) the two forward uses lose to a seg_gather_sum launch + a plain chain; step time, one-box
        # A/B, masks 7 / 3 / 1 / 2 / 0 / 6 / 5 / 4: 4.313 / 4.345 / 4.310 / 4.308 / 4.273 / 4.282 / 4.275 / 4.252 ms
        self._fuse_mask = int(os.environ.get("GFV_CSR_FUSE_MASK", "4"))
        self._csr1_max = int(os.environ.get("GFV_CSR1_MAX_M", "16384"))   # bit 1 regardless of the mask up to this many node rows
        # weight gradients fused into the dX chain of the big MLP launches (column-owner backward family, include/gfv.h
        # gfv_rowtile_args_t.dw_partial): the chain launch accumulates dW3, dW2 (and dW1 of a 128-deep first layer), the bias
        # gradients and the LayerNorm's per workgroup; one reduction launch per MLP sums the blocks
        self.fuse_dw = os.environ.get("GFV_FUSE_DW", "1") != "0"
        # ... and recompute z2 and the LayerNorm input from z1 in that launch instead of saving them in the forward and reading
        # them back (gfv_rowtile_args_t.rc_Wh): the forward of such an MLP writes z1, the row statistics and its outputs only.
        # Parity-green, 1.17 GB per step less HBM traffic (12.49 -> 11.32 GB on the counters) - and SLOWER: 3.92 against 3.82 ms at B = 1, 23.44 against 23.09 ms at 8
        # meshes per GPU (profiles/r04_ab_recompute.txt).  The forward chains gain 8 us per launch, the backward loses 25 (edge
        # level): that kernel is bound by instruction issue inside the CU (vector + LDS + matrix time add up,
        # profiles/r04_colchain_phases.txt: a fifth of the read traffic changes its time by 7 %), not by bytes.  Opt-in.
        self.recompute = os.environ.get("GFV_RECOMPUTE", "0") != "0"
        self._slice_fuse = os.environ.get("GFV_SLICE_FUSE", "1") != "0"   # Transolver adjoint: one pass behind the attention
        # the row-local Linear chains of a Transolver block (to_out .. linear_post and their adjoints) as one launch each (round 5)
        self._trans_fuse = os.environ.get("GFV_TRANS_FUSE", "1") != "0"
        # (above the small-tile single-layer family's range: 3.60 - 3.62 against 3.64 ms on the 50 k-cell mesh, 22.06 against 22.17 at
        # 8 meshes per GPU; on the 5 k-cell cavity the three small-tile launches are as fast: profiles/r05_ab_transmlp.txt)
        self._trans_fuse_min = int(os.environ.get("GFV_TRANS_FUSE_MIN_M", "16385"))
        # the FORWARD chain has a small-tile form as well (csrc/ctrans.hip, up to GFV_CTRANS_MAX_M rows): fused at every size
        self._trans_fuse_fwd_small = bool(L.get_limit("GFV_CTRANS")) if torch.cuda.is_available() else True
        self._trans_reduce_merge = os.environ.get("GFV_TRANS_REDUCE_MERGE", "1") != "0"
        # the end of the backward: how many of the trailing weight-gradient flushes run on the MAIN stream (GFV_TAIL_MAIN: 1 = the last
        # encoder's, 2 = both encoders', 3 = the first GnBlock's too) and how the first GnBlock's flush is split between the streams
        # (GFV_TAIL_SPLIT: 0 = not at all, 1 / 2 = its first / its other pieces on main).  Unset: by launch size (`_tail_cfg`: (2, 0)
        # below 32 k edge rows, (2, 2) up to 150 k, (2, 1) above; through round 5 (3, 0) between 30 k and 300 k, (2, 1) outside).  Measured with the round-4 kernels (the encoders' narrow weight gradients now
        # take 12 - 36 us instead of 27 - 66, which left the side queue as the tail): one 50 k-cell mesh 3.704 against 3.731 ms,
        # the reference's 15 k-cell polygon mesh 3.239 against 3.280; 8 meshes 21.92 against 21.84 and the 5 k-cell cavity 2.022
        # against 2.005 the other way round (latency-bound and bandwidth-bound ends) - interleaved on one box
        self._tail_env = ("GFV_TAIL_MAIN" in os.environ) or ("GFV_TAIL_SPLIT" in os.environ)
        self._tail_main = int(os.environ.get("GFV_TAIL_MAIN", "2"))
        self._tail_split = int(os.environ.get("GFV_TAIL_SPLIT", "1"))   # last GnBlock's flush: 1 / 2 = its first / its other pieces on main
        self._fuse_noout = os.environ.get("GFV_FUSE_NOOUT", "1") != "0"   # ... also where the input needs no gradient (encoders)
        # (round 5: 2 048 instead of 16 384 rows - on a 5 k-cell mesh the persistent backward with one tile per workgroup takes
        # 26 - 31 us per launch where the row-owner dX chain took 33 - 41 and left three weight-gradient tiles to the side queue:
        # 1.836 -> 1.77 ms per step, profiles/r05_cfwd.txt)
        self._fuse_dw_min = int(os.environ.get("GFV_COLCHAIN_BWD_MIN_M", "2048"))
        # up to this many rows the dX chain of an MLP runs on the column-owner small-tile backward (csrc/cbwd.hip) and its weight
        # gradients as one launch of the side queue; above it the persistent backward with fused weight gradients
        # (the library's own dispatch limits, csrc/gfv_limits.h: read when a forward starts, `_cbwd_max` below)
        self._cbwd_max_fixed = None
        self._wi, self._wi_key, self._wmax, self._wi_abs = None, None, None, None
        self._pkey_cache = None
        self._zero_e = None
        self._etmp = None
        self._prep_ws = None
        self._fvm_cnt = None
        self._fvm_fuse = os.environ.get("GFV_FVM_FUSE", "1") != "0"   # round 6: forward tail as one launch, backward as three
        # round 6: the EdgeBlock forward does not write its LayerNorm output a second time without the residual (512 B per edge row
        # on a launch that runs at the rate of its bytes) - the node aggregation applies the LayerNorm to the saved pre-LayerNorm rows
        # on the way in (gfv_seg_gather_sum_ln: the launch's own expression and statistics, bit-identical sums).  Interleaved on one
        # box: 3.59 -> 3.52 ms per step at 50 k cells, 22.6 -> 22.2 ms per step of 8 meshes
        self._agg_ln = os.environ.get("GFV_AGG_LN", "1") != "0"
        self._side, self._sides = None, []
        self._keep = []
        # data parallel: (world, process group) set by TrainStep; the Normalizer statistics are exchanged inside the
        # forward, between accumulation and use
        self.dist_world, self.dist_group, self.dist_force = 1, None, False
        self._retired = []   # superseded image sets / descriptor tables: kept alive while captured graphs may point at them
        # called once in the backward, when every gradient of the LAST processor and of the decoder has been launched
        # (TrainStep starts the all-reduce of that half of the flat gradient there, overlapped with the rest)
        self.bucket_hook = None

    # ------------------------------------------------------------------------------------------------------------
    # side stream for work nothing downstream of the backward chain waits for (dW, LayerNorm dgamma/dbeta)
    # ------------------------------------------------------------------------------------------------------------
    class _Fork:
        def __init__(self, eng, keep):
            self.eng, self.keep = eng, keep

        def __enter__(self):
            e = self.eng
            if not e.overlap:
                return self
            if e._side is None:
                e._side = pick_concurrent_stream()
                # ONE side queue: the weight-gradient launches share one slab workspace and the stand-in gradient blocks, so
                # two side queues would race on them (the several-queues experiment of round 2 is gone with its knob)
                e._sides = [e._side]
            L.stream_wait(e._side, torch.cuda.current_stream())
            # tensors the side stream reads must outlive this call: the caching allocator would hand their blocks to
            # the next allocation on the main stream while the side stream is still reading them
            e._keep.extend(t for t in self.keep if t is not None)
            self._ctx = torch.cuda.stream(e._side)
            self._ctx.__enter__()
            return self

        def __exit__(self, *exc):
            if self.eng.overlap:
                self._ctx.__exit__(*exc)
            return False

    def fork(self, *keep):
        return Engine._Fork(self, keep)

    @contextlib.contextmanager
    def model_width(self):
        """The library's process-wide hidden size (include/gfv.h gfv_set_hidden_size) is this model's while its launches are
        issued, and back at 128 - what every stand-alone operator assumes - afterwards."""
        lib = L.load()
        if self.hidden == 128 and lib.gfv_hidden_size() == 128:
            yield
            return
        L.check(lib.gfv_set_hidden_size(self.hidden), "gfv_set_hidden_size")
        try:
            yield
        finally:
            L.check(lib.gfv_set_hidden_size(128), "gfv_set_hidden_size")

    @property
    def _cbwd_max(self):
        if self._cbwd_max_fixed is not None:
            return self._cbwd_max_fixed
        return L.get_limit("GFV_CBWD_MAX_M") if L.get_limit("GFV_CBWD") else 0

    def _tail_cfg(self, pl):
        """(GFV_TAIL_MAIN, GFV_TAIL_SPLIT) of this batch: the environment's, or by the number of edge rows (see __init__)."""
        if self._tail_env:
            return self._tail_main, self._tail_split
        # round 6, over thirteen meshes from 2 k to 180 k edge rows with the final kernels (profiles/r06_tail_sweep.txt): the rule of
        # round 5 - (3, 0) between 30 k and 300 k edge rows - was the slowest or second slowest pair at every size
        # ((3, k) with k > 0 issues the same launches as (2, k): a split flush ignores `on_main`)
        if pl.E < 32000:
            return 2, 0
        return (2, 2) if pl.E < 150000 else (2, 1)

    def defer(self, fn, *keep):
        """Parameter-gradient work (nothing downstream of the backward chain reads it): queued and launched on the side
        stream with one fork per block (`flush`), the last block's on the main stream - 6.5 us less bubble per fork.
        (GFV_DEFER=0 forks at once for every piece instead.  Which of the two wins moved with the kernels: with the
        round-1 chain kernels the immediate fork was ahead, 4.90 against 5.00 ms / step - a whole block's weight gradients
        on top of the next block's 75 k-row dX chain; with the round-2 ones the queued form is, 4.29 against 4.33.)
        Without the side stream the work runs inline - results do not depend on where it runs (disjoint gradient blocks,
        fixed summation orders)."""
        if not self.overlap:
            fn()
        elif self._defer_mode:
            self._pending.append((fn, keep))
        else:
            with self.fork(*keep):
                fn()

    def flush(self, on_main=False, split=0):
        """Launch the queued work on the side stream (one fork), or - `on_main` - on the main stream: the last pieces of
        the backward have nothing left to hide behind, both queues then drain together."""
        if not self._pending:
            return
        pend, self._pending = self._pending, []
        if split and len(pend) > 1:
            # part of the queue on each stream (the end of the backward: both queues should drain together)
            k = 1 if split == 1 else len(pend) - 1
            main_part, side_part = (pend[:k], pend[k:]) if split == 1 else (pend[k:], pend[:k])
            self._pending = side_part
            self.flush()
            self._pending = main_part
            self.flush(on_main=True)
            return
        if on_main:
            ws, self._dw_ws = self._dw_ws, self._dw_ws_main   # the side stream may still be using its workspace
            try:
                for fn, _ in pend:
                    fn()
            finally:
                self._dw_ws_main, self._dw_ws = self._dw_ws, ws
            return
        with self.fork(*[t for _, keep in pend for t in keep]):
            for fn, _ in pend:
                fn()

    def join(self):
        """Main stream waits for the side stream (call before anything reads the gradients)."""
        self.flush()
        if self.overlap and self._side is not None:
            for sd in self._sides:
                L.stream_wait(torch.cuda.current_stream(), sd)
        self._keep.clear()

    # ------------------------------------------------------------------------------------------------------------
    # split-fp16 weight images: one set for the forward launches (built from the parameters when the forward starts),
    # one for the backward (built from the transposed copies right after prepare_transposes)
    # ------------------------------------------------------------------------------------------------------------
    def _pkey(self, P, fresh=False):
        """Identity of a parameter set (the data pointers of its tensors); computed when a forward starts and reused
        by the backward of the same step (same dict object)."""
        hit = self._pkey_cache
        if fresh or hit is None or hit[0] is not P:
            self._pkey_cache = hit = (P, tuple(t.data_ptr() for t in P.values()))
        return hit[1]

    def _wi_enter(self, phase, P):
        if not self.f16split:
            return None
        key = self._pkey(P, fresh=(phase == "fwd"))
        if self._wi_key != key:
            dev = next(iter(P.values())).device
            self._retired.append((self._wi, self._wmax, self._wi_abs))
            self._wmax = torch.zeros((1,), dtype=torch.float32, device=dev)
            self._wi = {ph: ops.WeightImages(dev, self._wmax) for ph in ("fwd", "bwd")}
            self._wi["fwd"].add_static(P.values())
            mats = [t for t in P.values() if t.dim() == 2 and t.stride(1) == 1]
            self._wi_abs = ops.WeightImages._upload([((t.data_ptr(), t.stride(0), t.shape[0], t.shape[1]), t) for t in mats], dev)
            self._wi_key = key
        wi = self._wi[phase]
        if phase == "fwd":
            # one power-of-two scale for all weight images of this step, from max|W| over every weight matrix: a 4-byte fill + integer
            # atomicMax on the bit pattern (order independent).  (A one-launch form - the maximum written by the workgroup that
            # arrives last - was measured twice, neutral in round 5, 13.7 us against 4.7 + 5.7 in round 6's timeline: removed.)
            L.check(L.load().gfv_weight_absmax(self._wi_abs[0].data_ptr(), self._wi_abs[1], self._wmax.data_ptr(),
                                               L.stream_ptr()), "gfv_weight_absmax")
        elif not wi.static and self._wt:
            wi.add_static(self._wt.values())
            if self.recompute:
                wi.add_static(P.values())   # the recompute form also takes FORWARD images of the second / third Linear here
        wi.build()
        return ops.set_weight_images(wi)

    def capture_signature(self):
        """Identity of everything a captured step points at inside the engine (weight images, descriptor tables,
        transposed copies); TrainStep drops its hipGraphs when it changes."""
        wi = self._wi or {}
        return (self._wi_key, self._wt_key, L.load().gfv_f16split_enabled(),   # (the product form: launches and images depend on it)
                tuple((ph, len(w.images), None if w._desc is None else w._desc.data_ptr()) for ph, w in sorted(wi.items())))

    def _wi_exit(self, phase, prev):
        if not self.f16split:
            return
        self._wi[phase].invalidate()    # the images are only valid for this step's parameter values
        ops.set_weight_images(prev)

    # ------------------------------------------------------------------------------------------------------------
    # transposed weights for the dX chains: one batched launch per step (prepare_transposes) or on demand
    # ------------------------------------------------------------------------------------------------------------
    def _T(self, W, perm=False):
        """W^T ([in, out]); perm=True: NodeBlock first layer [128, 64+128] -> rows for x (128) first, then nbm (64)."""
        hit = self._wt.get((W.data_ptr(), perm)) if self._wt_live else None
        if hit is not None:
            return hit
        if not perm:
            return ops.transpose(W)
        if perm == "c":      # (W1[:, 256:384])^T
            return ops.transpose(W, col0=256, ncols=128)
        if perm == "ab":     # [ (W1[:, 0:128])^T | (W1[:, 128:256])^T ]  ([128, 256])
            return torch.cat((ops.transpose(W, col0=0, ncols=128), ops.transpose(W, col0=128, ncols=128)), 1).contiguous()
        out = _empty(W.device, 192, 128)
        ops.transpose(W, out=out[0:128], col0=64, ncols=128)
        ops.transpose(W, out=out[128:192], col0=0, ncols=64)
        return out

    def prepare_transposes(self, P):
        """Transpose every weight the backward needs in ONE launch (descriptor table cached per parameter set)."""
        lib = L.load()
        key = self._pkey(P)
        if self._wt_key != key:
            import ctypes as C
            rows = []
            self._retired.append((self._wt, getattr(self, "_wt_desc", None)))
            self._wt = {}
            dev = next(iter(P.values())).device
            for n, W in P.items():
                if not n.endswith(".weight") or W.dim() != 2:
                    continue
                if any(t in n for t in (".to_q.", ".to_k.", ".to_v.", ".in_project_slice.")):
                    continue
                if ".encoder." in n and n.endswith(".0.0.weight"):
                    continue  # encoder inputs need no gradient
                r, c = W.shape
                if ".nb_module.net.0.0.weight" in n:
                    out = _empty(dev, 192, 128)
                    rows.append((W.data_ptr() + 4 * 64, out.data_ptr(), r, 128, c, 0))
                    rows.append((W.data_ptr(), out.data_ptr() + 4 * 128 * 128, r, 64, c, 0))
                    self._wt[(W.data_ptr(), True)] = out
                elif self.factor and ".eb_module.net.0.0.weight" in n:
                    oc, oab = _empty(dev, 128, 128), _empty(dev, 128, 256)
                    rows.append((W.data_ptr() + 4 * 256, oc.data_ptr(), r, 128, c, 0))
                    rows.append((W.data_ptr(), oab.data_ptr(), r, 128, c, 256))
                    rows.append((W.data_ptr() + 4 * 128, oab.data_ptr() + 4 * 128, r, 128, c, 256))
                    self._wt[(W.data_ptr(), "c")] = oc
                    self._wt[(W.data_ptr(), "ab")] = oab
                else:
                    out = _empty(dev, c, r)
                    rows.append((W.data_ptr(), out.data_ptr(), r, c, c, 0))
                    self._wt[(W.data_ptr(), False)] = out
            import struct
            blob = b"".join(struct.pack("<QQiiii", a, b, r, c, ld, ldo) for a, b, r, c, ld, ldo in rows)
            self._wt_desc = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            self._wt_n = len(rows)
            self._wt_max = (max(r[2] for r in rows), max(r[3] for r in rows))
            self._wt_key = key
        L.check(lib.gfv_transpose_batch(self._wt_desc.data_ptr(), self._wt_n, self._wt_max[0], self._wt_max[1],
                                        L.stream_ptr()), "gfv_transpose_batch")
        self._wt_live = True

    # ------------------------------------------------------------------------------------------------------------
    # fused 3-layer MLP (EPD.py:10-63)
    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _mlp_names(prefix, ln):
        lin = prefix + ".0" if ln else prefix
        return [f"{lin}.0.weight", f"{lin}.0.bias", f"{lin}.2.weight", f"{lin}.2.bias", f"{lin}.4.weight",
                f"{lin}.4.bias"] + ([f"{prefix}.1.weight", f"{prefix}.1.bias"] if ln else [])

    def mlp3_fwd(self, P, prefix, M, segs, *, ln=True, res=None, in_add=None, want_nores=False, keep=True, w1=None,
                 padd=None, saved_segs=None, nores_from_saved=False):
        """w1: column block of the first weight that multiplies `segs` (default: all of it); padd = (t [*,256], s, r):
        gathered addend t[s[m], :128] + t[r[m], 128:] to the first pre-activation (factored EdgeBlock).
        nores_from_saved: the caller can form the output without the residual from what the launch saves anyway (pre-LayerNorm
        rows + row statistics): where both exist the second output is not written and None is returned for it."""
        names = self._mlp_names(prefix, ln)
        W1, b1, W2, b2, W3, b3 = (P[n] for n in names[:6])
        if w1 is not None:
            W1 = w1
        dev = W1.device
        nout = W3.shape[0]
        # (mean, 1 / std) of the LayerNorm rows for a backward launch that fuses the weight gradients (it does not recompute them)
        fused_bwd = (keep and ln and self.fuse_dw and self.f16split and self.hidden == 128 and nout == 128
                     and M >= self._fuse_dw_min and M > self._cbwd_max
                     and (self._fuse_noout or all(sg.width % 32 == 0 for sg in segs)))
        # (the small-tile backward reads them as well)
        stats = _empty(dev, M, 2) if (fused_bwd or (keep and ln and self.f16split and nout == 128 and M <= self._cbwd_max)) else None
        # that launch rebuilds z2 and the LayerNorm input from z1 (recompute form): they are not written here at all
        lean = fused_bwd and self.recompute
        z1 = _empty(dev, M, 128) if keep else None
        z2 = _empty(dev, M, 128) if (keep and not lean) else None
        y3 = _empty(dev, M, 128) if (keep and ln and not lean) else None
        out = _empty(dev, M, nout)
        if want_nores and nores_from_saved and y3 is not None and stats is not None and nout == 128:
            want_nores = False
        nores = _empty(dev, M, 128) if want_nores else None
        ops.rowtile_chain(
            M, segs,
            [LayerSpec(W1, b1, L.OP_BIAS_GELU, save=z1), LayerSpec(W2, b2, L.OP_BIAS_GELU, save=z2), LayerSpec(W3, b3)],
            [out], in_add=in_add, fin_op=L.FIN_LN if ln else L.FIN_PLAIN,
            fin_gamma=P[names[6]] if ln else None, fin_beta=P[names[7]] if ln else None, fin_presave=y3,
            res=[res] if res is not None else None, out_nores=nores, fin_stats=stats,
            **(dict(padd=padd[0], padd_s=padd[1], padd_r=padd[2]) if padd is not None else {}))
        # (a segmented-sum input segment is kept for the backward in its assembled form, written by the launch itself)
        saved = dict(z1=z1, z2=z2, y3=y3, segs=segs if saved_segs is None else saved_segs, in_add=in_add, M=M, ln=ln,
                     prefix=prefix, stats=stats, lean=lean, fused=fused_bwd)
        return out, nores, saved

    def _rematerialize(self, P, sv):
        """z2 and the LayerNorm input of an MLP whose forward kept z1 only (recompute form), for a backward that turns out not
        to run fused after all (a gradient with a foreign row stride, a stand-alone operator called with other arguments): one
        two-layer launch from z1."""
        if not sv.get("lean") or sv["z2"] is not None:
            return
        names = self._mlp_names(sv["prefix"], sv["ln"])
        M, dev = sv["M"], sv["z1"].device
        sv["z2"], sv["y3"] = _empty(dev, M, 128), _empty(dev, M, 128)
        ops.rowtile_chain(M, [Seg(sv["z1"])], [LayerSpec(P[names[2]], P[names[3]], L.OP_BIAS_GELU, save=sv["z2"]),
                                              LayerSpec(P[names[4]], P[names[5]])], [sv["y3"]], in_op=L.IN_GELU)

    def mlp3_bwd(self, P, sv, G, grads, *, outs=None, res=None, gadd=None, W1t=None, g_ld=None, g_add=None):
        """G: grad wrt the MLP output (after LayerNorm, before the residual) [M, nout].  outs: per 128-chunk of the
        (possibly permuted) input, or None when the input needs no gradient (encoders)."""
        prefix, M, ln = sv["prefix"], sv["M"], sv["ln"]
        names = self._mlp_names(prefix, ln)
        W1, W2, W3 = P[names[0]], P[names[2]], P[names[4]]
        dev = W1.device
        nout = W3.shape[0]
        W3t, W2t = self._T(W3), self._T(W2)
        if (self.fuse_dw and ln and sv.get("fused", sv.get("stats") is not None) and G.stride(0) == 128 and g_ld is None and gadd is None
                and (outs is not None or (res is None and self._fuse_noout))
                and self._mlp3_bwd_fused(P, sv, G, grads, names, W3t, W2t,
                                         None if outs is None else (W1t if W1t is not None else self._T(W1)), outs, res, g_add)):
            return
        self._rematerialize(P, sv)
        gz2, gz1 = _empty(dev, M, 128), _empty(dev, M, 128)
        part = _empty(dev, ops.ln_rows(M), 2, 128) if ln else None
        g3 = _empty(dev, M, 128) if ln else G
        gseg = Seg(G, width=nout, ld=G.stride(0) if g_ld is None else g_ld)
        kw = dict(in_op=L.IN_LNBWD, in_gamma=P[names[6]], in_aux=sv["y3"], in_save=g3, ln_partial=part,
                  in_stats=sv.get("stats")) if ln else {}
        if gadd is not None:
            kw.update(gadd=gadd[0], gadd_s=gadd[1], gadd_r=gadd[2])
        if g_add is not None:   # G + g_add is the gradient (added in the prologue, before the LayerNorm backward)
            assert ln, "the addend is folded into the LayerNorm-backward prologue"
            kw.update(in_add=g_add)
        gs = _empty(dev, 3, ops.gscale_ld(M))   # per-16-row scales of (g3 | G, gz2, gz1) for the weight-gradient launch
        if outs is not None:
            if W1t is None:
                W1t = self._T(W1)
            have = ops.rowtile_chain(M, [gseg],
                                     [LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=sv["z2"]),
                                      LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=sv["z1"]), LayerSpec(W1t)],
                                     outs, res=res, gscale=gs, **kw)
        else:
            have = ops.rowtile_chain(M, [gseg],
                                     [LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=sv["z2"]),
                                      LayerSpec(W2t, None, L.OP_MUL_DGELU, aux=sv["z1"])], [gz1], gscale=gs, **kw)
        tiles_n = ops.last_ln_rows()   # (one row per 64 rows, per 32 when the small-tile backward took the launch)
        s0, s1, s2 = (gs[0], gs[1], gs[2]) if have else (None, None, None)
        if not ln and (gadd is not None or g_add is not None):
            s0 = None   # slot 0 describes the prologue RESULT (= g3 with LayerNorm); without it the weight gradient reads G
        # (narrow raw inputs - the encoders' x [N,12] / edge_attr [E,15] with geometric columns at mesh-spacing scale - get
        # per-column scales in the split-fp16 weight gradient; 128-wide latent segments are O(1) and go unscaled)
        tiles = [self._tile(gz1, 128, sg, in_add=sv["in_add"] if i == 0 else None, gscale=s2,
                            a_op=L.DW_COLSCALE if sg.width <= 16 else 0) for i, sg in enumerate(sv["segs"])]
        tiles.append(self._tile(gz2, 128, Seg(sv["z1"]), a_op=1, gscale=s1))
        tiles.append(self._tile(g3, nout, Seg(sv["z2"]), a_op=1, ldg=(G.stride(0) if g_ld is None else g_ld) if not ln else None,
                                gscale=s0))
        def side():
            # all weight gradients of the MLP in one launch, all their reductions (slab partials -> the gradient block,
            # per-tile LayerNorm partials -> dgamma | dbeta) in one more
            wptr, slabs, blen, _ = self._dw_block(
                grads, [(names[0], names[1], len(sv["segs"])), (names[2], names[3], 1), (names[4], names[5], 1)], tiles, M,
                reduce=False)
            off0, _ = grads.block(names[0], names[5])
            pieces = [dict(partial=wptr, out=grads.flat.data_ptr() + 4 * off0, n_chunks=slabs, chunk_stride=blen, rows=1, cols=blen)]
            if ln:
                pieces.append(dict(partial=part, out=self._gview2(grads, names[6], names[7]), n_chunks=tiles_n, chunk_stride=256,
                                   rows=1, cols=256))
            ops.reduce_multi(pieces)
        self.defer(side, gz1, gz2, g3, G, part, gs, sv["z1"], sv["z2"], sv["in_add"], *[sg.t for sg in sv["segs"]],
                   *[sg.idx for sg in sv["segs"]])

    def _mlp3_bwd_fused(self, P, sv, G, grads, names, W3t, W2t, W1t, outs, res, g_add):
        """mlp3_bwd with the weight gradients of the third and second Linear (and their bias / LayerNorm gradients) fused into
        the chain launch (column-owner backward family); the first layer's by the weight-gradient kernel over its input segments,
        with the row scales the chain leaves behind.  False: the library would not run this launch fused."""
        M = sv["M"]
        dev = G.device
        lib = L.load()
        nwg = lib.gfv_rowtile_dw_partials_m(M)
        FL = L.DW_FUSED_FLOATS
        dwp = _empty(dev, nwg, FL)
        gz1 = _empty(dev, M, 128)
        gs = _empty(dev, 3, ops.gscale_ld(M))
        if outs is None:
            # the input needs no gradient (the encoders): two chain layers, the launch's output is gz1
            layers = [LayerSpec(W3t, None, L.OP_MUL_DGELU, aux=sv["z2"]), LayerSpec(W2t, None, L.OP_MUL_DGELU, aux=sv["z1"])]
            outs = [gz1]
        else:
            layers = [LayerSpec(W3t, None, L.OP_MUL_DGELU, aux=sv["z2"]),
                      LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=sv["z1"]), LayerSpec(W1t)]
        kw = dict(res=res, in_op=L.IN_LNBWD, in_gamma=P[names[6]], in_aux=sv["y3"], in_stats=sv["stats"], dw_partial=dwp, gscale=gs,
                  in_add=g_add, rc=(P[names[2]], P[names[3]], P[names[4]], P[names[5]]) if sv["z2"] is None else None)
        if not ops.rowtile_chain(M, [Seg(G)], layers, outs, query_fused=True, **kw):
            return False
        ops.rowtile_chain(M, [Seg(G)], layers, outs, **kw)
        segs = sv["segs"]

        base = dwp.data_ptr()
        piece = lambda off, out, cols: dict(partial=base + 4 * off, out=out, n_chunks=nwg, chunk_stride=FL, rows=1, cols=cols)
        fused_pieces = [piece(16384 + 128, self._gview2(grads, names[2], names[3]), 16384 + 128),   # dW2 | db2
                        piece(0, self._gview2(grads, names[4], names[5]), 16384 + 128),             # dW3 | db3
                        piece(2 * 16384 + 256, self._gview2(grads, names[6], names[7]), 256)]       # dgamma | dbeta
        keep = (dwp, gz1, gs, sv["in_add"], *[sg.t for sg in segs], *[sg.idx for sg in segs])

        def side():
            lay = [(names[0], names[1], len(segs))]
            tiles = [self._tile(gz1, 128, sg, in_add=sv["in_add"] if i == 0 else None, gscale=gs[2],
                                a_op=L.DW_COLSCALE if sg.width <= 16 else 0) for i, sg in enumerate(segs)]
            w1, slabs1, blen1, _ = self._dw_block(grads, lay, tiles, M, reduce=False)
            off0, _ = grads.block(names[0], names[1])
            ops.reduce_multi([dict(partial=w1, out=grads.flat.data_ptr() + 4 * off0, n_chunks=slabs1, chunk_stride=blen1, rows=1,
                                   cols=blen1)] + fused_pieces)
        self.defer(side, *keep)
        return True

    @staticmethod
    def _gview2(grads, n0, n1):
        off, length = grads.block(n0, n1)
        return grads.flat[off:off + length]

    @staticmethod
    def _tile(G, n_out, seg, *, a_op=0, a_gamma=None, a_beta=None, in_add=None, ldg=None, g_offset=0, gscale=None):
        """gscale: a slot (row) of the buffer the chain launch that produced G filled, or None."""
        return dict(G=G, n_out=n_out, seg=seg, a_op=a_op, a_gamma=a_gamma, a_beta=a_beta, in_add=in_add,
                    ldg=G.stride(0) if ldg is None else ldg, g_offset=g_offset, gscale=gscale)

    def _workspace(self, n_floats, dev):
        if self._dw_ws is None or self._dw_ws.numel() < n_floats or self._dw_ws.device != dev:
            self._dw_ws = torch.zeros(int(n_floats * 1.25) + 1024, dtype=torch.float32, device=dev)
        return self._dw_ws

    def _dw_block(self, grads, layers, tiles, M, row0s=None, reduce=True, ws_offset=0):
        """layers: [(weight name, bias name or None, n tiles of that weight)], in parameter order; tiles in the same
        order.  One launch + (reduce=True) one reduction into the contiguous gradient block of those parameters;
        reduce=False leaves the slab partials in the workspace at float offset `ws_offset` and returns
        (pointer, slabs, floats per slab, workspace floats used) for a caller that folds several reductions into one launch."""
        lib = L.load()
        first = layers[0][0]
        last = layers[-1][1] if layers[-1][1] is not None else layers[-1][0]
        off0, blen = grads.block(first, last)
        ct = (L.DwTile * 6)()
        ti = 0
        for li, (wname, bname, nt) in enumerate(layers):
            K = grads.shape[wname][1]
            koff = 0
            for j in range(nt):
                t, c = tiles[ti], ct[ti]
                sg = t["seg"]
                row0 = 0 if row0s is None else row0s[ti]
                c.G = t["G"].data_ptr() + 4 * t["g_offset"]
                c.A = sg.t.data_ptr() + 4 * sg.offset
                c.idx = None if sg.idx is None else sg.idx.data_ptr()
                c.in_add = None if t["in_add"] is None else t["in_add"].data_ptr()
                c.a_gamma = None if t["a_gamma"] is None else t["a_gamma"].data_ptr()
                c.a_beta = None if t["a_beta"] is None else t["a_beta"].data_ptr()
                c.gscale = None if t.get("gscale") is None else t["gscale"].data_ptr()
                c.ldg, c.n_out, c.width, c.ld = t["ldg"], t["n_out"], sg.width, sg.ld
                c.a_op, c.ld_out = t["a_op"], K
                if row0s is None:
                    c.out_off = grads.off[wname] - off0 + koff
                    c.db_off = (grads.off[bname] - off0) if (bname is not None and j == 0) else -1
                    koff += sg.width
                else:
                    c.out_off = grads.off[wname] - off0 + row0 * K
                    c.db_off = (grads.off[bname] - off0 + row0) if bname is not None else -1
                ti += 1
        need = lib.gfv_dw_multi_workspace_floats(M, ti, blen)
        dev = grads.flat.device
        ws = self._workspace(ws_offset + need, dev)
        wptr = ws.data_ptr() + 4 * ws_offset
        L.check(lib.gfv_dw_multi(ct, ti, M, blen, wptr, (grads.flat.data_ptr() + 4 * off0) if reduce else None, 0,
                                 L.stream_ptr()), "gfv_dw_multi")
        if not reduce:
            return wptr, lib.gfv_dw_slabs(M, ti, None), blen, (need + 3) // 4 * 4

    def _dw_jobs(self, grads, jobs, pieces=()):
        """Several weight-gradient launches whose reductions are ONE launch (round 6: a Transolver block's four weight-gradient
        launches used to be followed by six reduction launches on the side queue).  jobs: dicts (layers, tiles, M, row0s) as
        `_dw_block` takes them; every job gets its own region of the slab workspace (sized for all of them up front: the
        workspace must not move under launches already issued); `pieces`: further reduction pieces (LayerNorm / attention
        partials) folded into the same launch."""
        lib = L.load()
        dev = grads.flat.device
        needs = []
        for j in jobs:
            first = j["layers"][0][0]
            last = j["layers"][-1][1] if j["layers"][-1][1] is not None else j["layers"][-1][0]
            blen = grads.block(first, last)[1]
            needs.append((lib.gfv_dw_multi_workspace_floats(j["M"], len(j["tiles"]), blen) + 3) // 4 * 4)
        self._workspace(sum(needs) + 8, dev)
        out, off = [], 0
        for j, need in zip(jobs, needs):
            first = j["layers"][0][0]
            last = j["layers"][-1][1] if j["layers"][-1][1] is not None else j["layers"][-1][0]
            wptr, slabs, blen, used = self._dw_block(grads, j["layers"], j["tiles"], j["M"], row0s=j.get("row0s"), reduce=False,
                                                     ws_offset=off)
            assert used <= need
            off0, _ = grads.block(first, last)
            out.append(dict(partial=wptr, out=grads.flat.data_ptr() + 4 * off0, n_chunks=slabs, chunk_stride=blen, rows=1, cols=blen))
            off += need
        out += list(pieces)
        for k in range(0, len(out), 12):   # (gfv_reduce_multi takes 12 pieces)
            ops.reduce_multi(out[k:k + 12])

    @staticmethod
    def _put(grads, name, value):
        cmdlist.call(grads.view(name).copy_, value.reshape(grads.shape[name]))

    # ------------------------------------------------------------------------------------------------------------
    # GnBlock (EPD.py:177-195, blocks.py)
    # ------------------------------------------------------------------------------------------------------------
    def gn_fwd(self, P, prefix, x, e, pl):
        N, E = pl.N, pl.E
        fuse = self.csr_fuse
        if self.factor:
            W1 = P[f"{prefix}.eb_module.net.0.0.weight"]                   # [128, 384] = [W1a | W1b | W1c]
            pab = _empty(x.device, N, 256)                                  # [W1a nb | W1b nb] per node
            # (on short launches - the small-tile single-layer family's range, csrc/lin1s.hip - the prologue form always: one launch
            # of ~6 us instead of two of 5 + 8)
            if fuse and ((self._fuse_mask & 1) or N <= self._csr1_max) and ops.stack_ready(W1[:, 0:128], W1[:, 128:256], rows=True):
                # nb = sum over the neighbours (blocks.py:84-99) formed in the prologue of the launch that multiplies it
                nb = _empty(x.device, N, 128)
                ops.rowtile_chain(N, [Seg(x, csr=(pl.n_rowptr, pl.n_col_node), save=nb)],
                                  [LayerSpec(W1[:, 0:128], stack=W1[:, 128:256])], [(pab, 256), (pab.data_ptr() + 512, 256)])
            else:
                nb = ops.seg_gather_sum(x, pl.n_rowptr, pl.n_col_node, N)
                ops.rowtile_chain(N, [Seg(nb)], [LayerSpec(W1[:, 0:128], stack=W1[:, 128:256])],
                                  [(pab, 256), (pab.data_ptr() + 512, 256)])
            e_out, e_new, sv_e = self.mlp3_fwd(P, f"{prefix}.eb_module.net", E, [Seg(e)], res=e, want_nores=True,
                                               w1=W1[:, 256:384], padd=(pab, pl.es, pl.er), nores_from_saved=self._agg_ln)
            sv_e["nb"] = nb
        else:
            nb = ops.seg_gather_sum(x, pl.n_rowptr, pl.n_col_node, N)
            e_out, e_new, sv_e = self.mlp3_fwd(P, f"{prefix}.eb_module.net", E, [Seg(nb, pl.es), Seg(nb, pl.er), Seg(e)],
                                               res=e, want_nores=True, nores_from_saved=self._agg_ln)
        if e_new is None:
            # (the aggregation of blocks.py:35-42 over LayerNorm(y3), formed on the way in from the rows and statistics the launch saved)
            ln_names = self._mlp_names(f"{prefix}.eb_module.net", True)
            agg = ops.seg_gather_sum_ln(sv_e["y3"], sv_e["stats"], P[ln_names[6]], P[ln_names[7]], pl.n_rowptr, pl.n_col_edge2, N)
        else:
            agg = ops.seg_gather_sum(e_new.view(2 * E, 64), pl.n_rowptr, pl.n_col_edge2, N)
        if fuse and (self._fuse_mask & 2):
            # nbm = mean over the neighbours of the aggregates (blocks.py:44-51), in the node MLP's prologue; the launch
            # leaves the assembled rows ([N,128] buffer, columns 0:64) for the weight gradient of the first layer
            nbm = _empty(x.device, N, 128)
            x_out, _, sv_n = self.mlp3_fwd(P, f"{prefix}.nb_module.net", N,
                                           [Seg(agg, csr=(pl.n_rowptr, pl.n_col_node), scale=pl.inv_deg, save=nbm), Seg(x)],
                                           res=x, saved_segs=[Seg(nbm, width=64, ld=128), Seg(x)])
        else:
            nbm = ops.seg_gather_sum(agg, pl.n_rowptr, pl.n_col_node, N, scale=pl.inv_deg)
            x_out, _, sv_n = self.mlp3_fwd(P, f"{prefix}.nb_module.net", N, [Seg(nbm), Seg(x)], res=x)
        return x_out, e_out, dict(sv_e=sv_e, sv_n=sv_n, prefix=prefix)

    def _edge_tmp(self, dev):
        if self._etmp is None or self._etmp[0].flat.device != dev:
            e = GradStore(["W1c", "b1", "W2", "b2", "W3", "b3"], [(128, 128), (128,), (128, 128), (128,), (128, 128), (128,)], dev)
            n = GradStore(["W1ab"], [(128, 256)], dev)
            self._etmp = (e, n)
        return self._etmp

    def edge_bwd_factored(self, P, sv, G, grads, pl, gadd):
        """Adjoint of the factored EdgeBlock MLP.  Returns (grad wrt nb [N,128], grad wrt e [E,128] incl. residual)."""
        prefix, M, N = sv["prefix"], sv["M"], pl.N
        names = self._mlp_names(prefix, True)
        W1, W2, W3 = P[names[0]], P[names[2]], P[names[4]]
        dev = W1.device
        W3t, W2t, W1ct, Wabt = self._T(W3), self._T(W2), self._T(W1, perm="c"), self._T(W1, perm="ab")
        e = sv["segs"][0].t
        if self.fuse_dw and sv.get("fused", sv.get("stats") is not None) and e.stride(0) == 128 and G.stride(0) == 128:
            fused = self._edge_bwd_fused(P, sv, G, grads, pl, gadd, names, (W3t, W2t, W1ct, Wabt))
            if fused is not None:
                return fused
        self._rematerialize(P, sv)
        gz2, gz1, g3, g_e_in = (_empty(dev, M, 128) for _ in range(4))
        part = _empty(dev, ops.ln_rows(M), 2, 128)
        gs = _empty(dev, 3, ops.gscale_ld(M))
        have = ops.rowtile_chain(M, [Seg(G)],
                                 [LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=sv["z2"]),
                                  LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=sv["z1"]), LayerSpec(W1ct)],
                                 [g_e_in], res=[G], in_op=L.IN_LNBWD, in_gamma=P[names[6]], in_aux=sv["y3"], in_save=g3,
                                 ln_partial=part, gadd=gadd[0], gadd_s=gadd[1], gadd_r=gadd[2], gscale=gs, in_stats=sv.get("stats"))
        tiles_n = ops.last_ln_rows()
        s0, s1, s2 = (gs[0], gs[1], gs[2]) if have else (None, None, None)
        # adjoint of the gathers (W1a nb)[s], (W1b nb)[r]: per-side scatter of dz1 to the nodes, then ONE node-level GEMM
        g_nb = _empty(dev, N, 128)
        if self.csr_fuse and (self._fuse_mask & 4):
            G_s, G_r = _empty(dev, N, 128), _empty(dev, N, 128)
            ops.rowtile_chain(N, [Seg(gz1, csr=(pl.s_rowptr, pl.s_col), save=G_s), Seg(gz1, csr=(pl.r_rowptr, pl.r_col), save=G_r)],
                              [LayerSpec(Wabt)], [g_nb])
        else:
            G_s = ops.seg_gather_sum(gz1, pl.s_rowptr, pl.s_col, N)
            G_r = ops.seg_gather_sum(gz1, pl.r_rowptr, pl.r_col, N)
            ops.rowtile_chain(N, [Seg(G_s), Seg(G_r)], [LayerSpec(Wabt)], [g_nb])
        nb = sv["nb"]
        def side():
            # slab partials are laid out like small stand-in blocks ([W1c | b1 | W2 | b2 | W3 | b3] and [W1a | W1b]), in two
            # regions of the workspace; ONE launch then reduces every piece straight into its place of the real gradient
            # block (W1c / W1ab are column blocks of W1 [128, 384]) together with the per-tile LayerNorm partials
            tmpE, tmpN = self._edge_tmp(dev)
            gW1 = grads.view(names[0])
            lib = L.load()
            self._workspace(lib.gfv_dw_multi_workspace_floats(M, 3, tmpE.block("W1c", "b3")[1])
                            + lib.gfv_dw_multi_workspace_floats(N, 2, tmpN.block("W1ab", "W1ab")[1]) + 8, dev)   # both regions, once
            w1, slabs1, blen1, used = self._dw_block(tmpE, [("W1c", "b1", 1), ("W2", "b2", 1), ("W3", "b3", 1)],
                                                     [self._tile(gz1, 128, Seg(e), gscale=s2),
                                                      self._tile(gz2, 128, Seg(sv["z1"]), a_op=1, gscale=s1),
                                                      self._tile(g3, 128, Seg(sv["z2"]), a_op=1, gscale=s0)], M, reduce=False)
            w2, slabs2, blen2, _ = self._dw_block(tmpN, [("W1ab", None, 2)],
                                                  [self._tile(G_s, 128, Seg(nb)), self._tile(G_r, 128, Seg(nb))], N,
                                                  reduce=False, ws_offset=used)
            assert w2 == w1 + 4 * used
            off, length = grads.block(names[1], names[5])
            o2, l2 = tmpE.block("b1", "b3")
            assert l2 == length
            ops.reduce_multi([
                dict(partial=w1, out=gW1.data_ptr() + 4 * 256, n_chunks=slabs1, chunk_stride=blen1, rows=128, cols=128, ld_out=384),
                dict(partial=w1 + 4 * o2, out=grads.flat.data_ptr() + 4 * off, n_chunks=slabs1, chunk_stride=blen1, rows=1, cols=l2),
                dict(partial=w2, out=gW1.data_ptr(), n_chunks=slabs2, chunk_stride=blen2, rows=128, cols=256, ld_out=384),
                dict(partial=part, out=self._gview2(grads, names[6], names[7]), n_chunks=tiles_n, chunk_stride=256, rows=1,
                     cols=256)])
        self.defer(side, gz1, gz2, g3, G_s, G_r, nb, e, part, gs, sv["z1"], sv["z2"])
        return g_nb, g_e_in

    def _edge_bwd_fused(self, P, sv, G, grads, pl, gadd, names, Wt):
        """edge_bwd_factored with every edge-level weight gradient fused into the chain launch (column-owner backward family):
        one chain launch over the E rows leaves per-workgroup blocks [dW3 | db3 | dW2 | db2 | dgamma | dbeta | dW1c | db1]; the
        side stream then needs only the node-level pair of W1a / W1b tiles and ONE reduction launch.  None: the library would not
        run this launch fused (shape / size / product form) - the caller takes the separate path."""
        M, N = sv["M"], pl.N
        W3t, W2t, W1ct, Wabt = Wt
        dev = G.device
        e = sv["segs"][0].t
        lib = L.load()
        nwg = lib.gfv_rowtile_dw_partials_m(M)
        lean = sv["z2"] is None   # recompute form: the forward kept z1 only
        FL = L.DW_FUSED_FLOATS
        dwp = _empty(dev, nwg, FL)
        gz1, g_e_in = _empty(dev, M, 128), _empty(dev, M, 128)
        layers = [LayerSpec(W3t, None, L.OP_MUL_DGELU, aux=sv["z2"]), LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=sv["z1"]),
                  LayerSpec(W1ct)]
        gs = _empty(dev, 3, ops.gscale_ld(M))   # slot 2: the scales of gz1's rows for the first layer's weight-gradient launch
        kw = dict(res=[G], in_op=L.IN_LNBWD, in_gamma=P[names[6]], in_aux=sv["y3"], in_stats=sv["stats"], gadd=gadd[0], gadd_s=gadd[1],
                  gadd_r=gadd[2], dw_partial=dwp, gscale=gs,
                  rc=(P[names[2]], P[names[3]], P[names[4]], P[names[5]]) if lean else None)
        if not ops.rowtile_chain(M, [Seg(G)], layers, [g_e_in], query_fused=True, **kw):
            return None
        ops.rowtile_chain(M, [Seg(G)], layers, [g_e_in], **kw)
        # adjoint of the gathers (W1a nb)[s], (W1b nb)[r]: per-side scatter of dz1 to the nodes, then ONE node-level GEMM
        g_nb = _empty(dev, N, 128)
        if self.csr_fuse and (self._fuse_mask & 4):
            G_s, G_r = _empty(dev, N, 128), _empty(dev, N, 128)
            ops.rowtile_chain(N, [Seg(gz1, csr=(pl.s_rowptr, pl.s_col), save=G_s), Seg(gz1, csr=(pl.r_rowptr, pl.r_col), save=G_r)],
                              [LayerSpec(Wabt)], [g_nb])
        else:
            G_s = ops.seg_gather_sum(gz1, pl.s_rowptr, pl.s_col, N)
            G_r = ops.seg_gather_sum(gz1, pl.r_rowptr, pl.r_col, N)
            ops.rowtile_chain(N, [Seg(G_s), Seg(G_r)], [LayerSpec(Wabt)], [g_nb])
        nb = sv["nb"]

        def side():
            tmpE, tmpN = self._edge_tmp(dev)
            gW1 = grads.view(names[0])
            base = dwp.data_ptr()
            piece = lambda off, out, rows, cols, ld_out=None: dict(partial=base + 4 * off, out=out, n_chunks=nwg, chunk_stride=FL,
                                                                   rows=rows, cols=cols, ld_in=cols, ld_out=cols if ld_out is None else ld_out)
            pieces = [piece(16384 + 128, self._gview2(grads, names[2], names[3]), 1, 16384 + 128),   # dW2 | db2
                      piece(0, self._gview2(grads, names[4], names[5]), 1, 16384 + 128),             # dW3 | db3
                      piece(2 * 16384 + 256, self._gview2(grads, names[6], names[7]), 1, 256)]       # dgamma | dbeta
            # the first layer's c block by the weight-gradient kernel (one tile over the E rows): slab partials laid out
            # like a stand-in [W1c | b1] block
            lay = [("W1c", "b1", 1)]
            self._workspace(lib.gfv_dw_multi_workspace_floats(M, 1, tmpE.block("W1c", "b1")[1])
                            + lib.gfv_dw_multi_workspace_floats(N, 2, tmpN.block("W1ab", "W1ab")[1]) + 8, dev)   # both regions, once
            w1, slabs1, blen1, used = self._dw_block(tmpE, lay, [self._tile(gz1, 128, Seg(e), gscale=gs[2])], M, reduce=False)
            ob1, lb1 = tmpE.block("b1", "b1")
            pieces += [dict(partial=w1, out=gW1.data_ptr() + 4 * 256, n_chunks=slabs1, chunk_stride=blen1, rows=128, cols=128, ld_out=384),
                       dict(partial=w1 + 4 * ob1, out=grads.view(names[1]), n_chunks=slabs1, chunk_stride=blen1, rows=1, cols=128)]
            w2, slabs2, blen2, _ = self._dw_block(tmpN, [("W1ab", None, 2)],
                                                  [self._tile(G_s, 128, Seg(nb)), self._tile(G_r, 128, Seg(nb))], N, reduce=False,
                                                  ws_offset=used)
            pieces.append(dict(partial=w2, out=gW1.data_ptr(), n_chunks=slabs2, chunk_stride=blen2, rows=128, cols=256, ld_out=384))
            ops.reduce_multi(pieces)
        self.defer(side, dwp, gz1, e, G_s, G_r, nb, gs)
        return g_nb, g_e_in

    def gn_bwd(self, P, sv, g_x_out, g_e_out, grads, pl):
        N, E = pl.N, pl.E
        dev = g_x_out.device
        prefix = sv["prefix"]
        W1n = P[f"{prefix}.nb_module.net.0.0.weight"]                      # [128, 64 + 128]
        W1t = self._T(W1n, perm=True)                                       # rows for x first, then nbm
        g_x_in, g_nbm = _empty(dev, N, 128), _empty(dev, N, 64)
        self.mlp3_bwd(P, sv["sv_n"], g_x_out, grads, outs=[g_x_in, (g_nbm, 64)], res=[g_x_out, None], W1t=W1t)
        g_agg = ops.seg_gather_sum(g_nbm, pl.n_rowptr, pl.n_col_node, N, src_scale=pl.inv_deg)
        if "nb" in sv["sv_e"]:
            g_nb, g_e_in = self.edge_bwd_factored(P, sv["sv_e"], g_e_out, grads, pl, (g_agg, pl.es, pl.er))
        else:
            gnb2, g_e_in = _empty(dev, E, 256), _empty(dev, E, 128)
            self.mlp3_bwd(P, sv["sv_e"], g_e_out, grads, outs=[(gnb2, 256), (gnb2.data_ptr() + 512, 256), g_e_in],
                          res=[None, None, g_e_out], gadd=(g_agg, pl.es, pl.er))
            g_nb = ops.seg_gather_sum(gnb2.view(2 * E, 128), pl.n_rowptr, pl.n_col_edge2, N)
        ops.seg_gather_sum(g_nb, pl.n_rowptr, pl.n_col_node, N, out=g_x_in, accumulate=True)
        # the block's weight gradients: one fork (the last block's may go to the main stream: GFV_TAIL_MAIN >= 3)
        last = self._defer_mode and getattr(self, "_last_gn", False)
        tail_main, tail_split = self._tail_cfg(pl)
        self.flush(on_main=last and tail_main >= 3, split=tail_split if last else 0)
        return g_x_in, g_e_in

    # ------------------------------------------------------------------------------------------------------------
    # Transolver block (GraphTransolver.py:48-95,163-169)
    # ------------------------------------------------------------------------------------------------------------
    def trans_fwd(self, P, prefix, xg, emb, pl):
        lib = L.load()
        st = L.stream_ptr()
        N, B = pl.N, pl.B
        dev = xg.device
        a = f"{prefix}.Attn"
        fx_in, fx_mid, x_mid = _empty(dev, N, 128), _empty(dev, N, 128), _empty(dev, N, 128)
        # fx_mid = in_project_fx(xg + emb), x_mid = in_project_x(xg + emb): one launch over the row-stacked pair
        ops.rowtile_chain(N, [Seg(xg)],
                          [LayerSpec(P[f"{a}.in_project_fx.weight"], P[f"{a}.in_project_fx.bias"],
                                     stack=P[f"{a}.in_project_x.weight"], bias2=P[f"{a}.in_project_x.bias"])],
                          [fx_mid, x_mid], in_add=emb, in_save=fx_in)
        w = _empty(dev, N, 256)
        temp = P[f"{a}.graph_temperature"]
        partial = _empty(dev, pl.n_chunks, 256, 17)
        if self._slice_fuse:
            # slice softmax + per-chunk token sums in one pass (w is written on the way, not re-read)
            L.check(lib.gfv_slice_softmax_token(x_mid.data_ptr(), P[f"{a}.in_project_slice.weight"].data_ptr(),
                                                P[f"{a}.in_project_slice.bias"].data_ptr(), temp.data_ptr(), fx_mid.data_ptr(),
                                                pl.chunk_beg.data_ptr(), pl.chunk_end.data_ptr(), pl.n_chunks, w.data_ptr(),
                                                partial.data_ptr(), st), "slice_softmax_token")
        else:
            L.check(lib.gfv_slice_softmax_fwd(x_mid.data_ptr(), P[f"{a}.in_project_slice.weight"].data_ptr(),
                                              P[f"{a}.in_project_slice.bias"].data_ptr(), temp.data_ptr(), w.data_ptr(), N,
                                              st), "slice_softmax_fwd")
            L.check(lib.gfv_slice_token_partial(w.data_ptr(), fx_mid.data_ptr(), pl.chunk_beg.data_ptr(),
                                                pl.chunk_end.data_ptr(), pl.n_chunks, partial.data_ptr(), st), "token_partial")
        partial, gptr = self._graph_partials(partial, pl)
        token, norm = _empty(dev, B, 8, 32, 16), _empty(dev, B, 8, 32)
        attn, out_token = _empty(dev, B, 8, 32, 32), _empty(dev, B, 8, 32, 16)
        L.check(lib.gfv_slice_attention_fwd(partial.data_ptr(), gptr.data_ptr(), B, P[f"{a}.to_q.weight"].data_ptr(),
                                            P[f"{a}.to_k.weight"].data_ptr(), P[f"{a}.to_v.weight"].data_ptr(),
                                            token.data_ptr(), norm.data_ptr(), attn.data_ptr(), out_token.data_ptr(), st),
                "slice_attention_fwd")
        out_x = _empty(dev, N, 128)
        L.check(lib.gfv_deslice(w.data_ptr(), out_token.data_ptr(), pl.batch.data_ptr(), out_x.data_ptr(), N,
                                4 if B == 1 else 0, st), "deslice")
        fx1, z, out = _empty(dev, N, 128), _empty(dev, N, 256), _empty(dev, N, 128)
        # to_out + residual, ln_2, linear_pre, GELU, linear_post + residual: nothing crosses rows - ONE launch (csrc/transmlp.hip)
        if not (self._trans_fuse and (N >= self._trans_fuse_min or self._trans_fuse_fwd_small) and ops.trans_mlp_fwd(
                out_x, fx_in, P[f"{a}.to_out.0.weight"], P[f"{a}.to_out.0.bias"], P[f"{prefix}.ln_2.weight"], P[f"{prefix}.ln_2.bias"],
                P[f"{prefix}.mlp.linear_pre.0.weight"], P[f"{prefix}.mlp.linear_pre.0.bias"], P[f"{prefix}.mlp.linear_post.weight"],
                P[f"{prefix}.mlp.linear_post.bias"], fx1, z, out)):
            ops.rowtile_chain(N, [Seg(out_x)], [LayerSpec(P[f"{a}.to_out.0.weight"], P[f"{a}.to_out.0.bias"])], [fx1],
                              res=[fx_in])
            ops.rowtile_chain(N, [Seg(fx1)],
                              [LayerSpec(P[f"{prefix}.mlp.linear_pre.0.weight"], P[f"{prefix}.mlp.linear_pre.0.bias"])],
                              [(z, 256), (z.data_ptr() + 512, 256)], in_op=L.IN_LN, in_gamma=P[f"{prefix}.ln_2.weight"],
                              in_beta=P[f"{prefix}.ln_2.bias"])
            ops.rowtile_chain(N, [Seg(z, width=128, ld=256), Seg(z, width=128, ld=256, offset=128)],
                              [LayerSpec(P[f"{prefix}.mlp.linear_post.weight"], P[f"{prefix}.mlp.linear_post.bias"])], [out],
                              in_op=L.IN_GELU, res=[fx1])
        sv = dict(prefix=prefix, fx_in=fx_in, fx_mid=fx_mid, x_mid=x_mid, w=w, token=token, norm=norm, attn=attn,
                  out_token=out_token, out_x=out_x, fx1=fx1, z=z)
        return out, sv

    def _graph_partials(self, partial, pl):
        """[n_chunks, 256, 17] per-chunk slice tokens -> [B, 256, 17] per-graph sums (one wide launch; the attention
        blocks - 8 per graph - then read one row instead of walking every chunk of their graph) and the chunk ranges to hand
        the attention kernel.  (Letting the attention blocks walk the chunks themselves on small meshes saves the launch and costs
        more than it: 6.7 -> 17.8 us forward, 11 -> 21 us backward at 81 chunks, profiles/r06_timeline_cavity_merged_first.txt.)"""
        out = _empty(partial.device, pl.B, 256, 17)
        L.check(L.load().gfv_reduce_partials_seg(partial.data_ptr(), pl.gchunk_ptr.data_ptr(), pl.B, 256 * 17,
                                                 out.data_ptr(), L.stream_ptr()), "reduce_partials_seg")
        return out, pl.gunit_ptr

    def trans_bwd(self, P, sv, g_out, grads, pl, g_add=None):
        """g_add: optional addend of the incoming gradient (g_out + g_add is the gradient); the sum is formed in the
        prologue of the first launch and kept (no separate elementwise pass)."""
        lib = L.load()
        st = L.stream_ptr()
        N, B = pl.N, pl.B
        dev = g_out.device
        prefix = sv["prefix"]
        a = f"{prefix}.Attn"
        z, fx1, fx_in = sv["z"], sv["fx1"], sv["fx_in"]
        zsegs = [Seg(z, width=128, ld=256), Seg(z, width=128, ld=256, offset=128)]
        # linear_post
        Wpost, Wpre = P[f"{prefix}.mlp.linear_post.weight"], P[f"{prefix}.mlp.linear_pre.0.weight"]
        g_z = _empty(dev, N, 256)
        g_sum = _empty(dev, N, 128) if g_add is not None else None
        gs = _empty(dev, 3, ops.gscale_ld(N))
        tiles = ops.rowtile_tiles(N)
        part = _empty(dev, ops.ln_rows(N), 2, 128)
        g_fx1, g_out_x = _empty(dev, N, 128), _empty(dev, N, 128)
        gam2, bet2 = P[f"{prefix}.ln_2.weight"], P[f"{prefix}.ln_2.bias"]
        # the three adjoint launches (GELU' epilogue, LayerNorm-backward epilogue, plain) as ONE (csrc/transmlp.hip; its small-tile
        # form, csrc/ctrans.hip, up to GFV_CTRANS_MAX_M rows)
        fused = (self._trans_fuse and (N >= self._trans_fuse_min or self._trans_fuse_fwd_small) and
                 ops.trans_mlp_bwd(g_out, g_add, g_sum, z, fx1, self._T(Wpost), self._T(Wpre), self._T(P[f"{a}.to_out.0.weight"]),
                                   gam2, g_z, g_fx1, g_out_x, part, gs[0]))
        if fused:
            have = True
            tiles = lib.gfv_trans_mlp_ln_rows(N)   # (one ln_partial row per 64 rows, per 32 from the small-tile form)
        else:
            have = ops.rowtile_chain(N, [Seg(g_out)], [LayerSpec(self._T(Wpost), None, L.OP_MUL_DGELU, aux=z)],
                                     [(g_z, 256), (g_z.data_ptr() + 512, 256)], in_add=g_add, in_save=g_sum, gscale=gs)
        s_post = gs[0] if have else None   # scale of the prologue result g_out (+ g_add) = the rows linear_post's dW reads
        if g_add is not None:
            g_out = g_sum
        g_post = g_out
        merge = self._trans_reduce_merge
        jobs, extra, keep = [], [], [g_post, z, gs]
        if merge:
            jobs.append(dict(layers=[(f"{prefix}.mlp.linear_post.weight", f"{prefix}.mlp.linear_post.bias", 2)],
                             tiles=[self._tile(g_post, 128, zs, a_op=1, gscale=s_post) for zs in zsegs], M=N))
        else:
            self.defer(lambda: self._dw_block(grads, [(f"{prefix}.mlp.linear_post.weight", f"{prefix}.mlp.linear_post.bias", 2)],
                                              [self._tile(g_post, 128, zs, a_op=1, gscale=s_post) for zs in zsegs], N), g_post, z, gs)
        # linear_pre behind LayerNorm ln_2
        if not fused:
            ops.rowtile_chain(N, [Seg(g_z, width=128, ld=256), Seg(g_z, width=128, ld=256, offset=128)],
                              [LayerSpec(self._T(Wpre))], [g_fx1], fin_op=L.FIN_LNBWD, fin_gamma=gam2, fin_aux=fx1,
                              ln_partial=part, res=[g_out])
        def side_pre():
            wn, bn = f"{prefix}.mlp.linear_pre.0.weight", f"{prefix}.mlp.linear_pre.0.bias"
            wptr, slabs, blen, _ = self._dw_block(
                grads, [(wn, bn, 2)],
                [self._tile(g_z, 128, Seg(fx1), a_op=2, a_gamma=gam2, a_beta=bet2, ldg=256, g_offset=128 * h) for h in range(2)],
                N, row0s=[0, 128], reduce=False)
            off0, _ = grads.block(wn, bn)
            ops.reduce_multi([
                dict(partial=wptr, out=grads.flat.data_ptr() + 4 * off0, n_chunks=slabs, chunk_stride=blen, rows=1, cols=blen),
                dict(partial=part, out=self._gview2(grads, f"{prefix}.ln_2.weight", f"{prefix}.ln_2.bias"), n_chunks=tiles,
                     chunk_stride=256, rows=1, cols=256)])
        if merge:
            jobs.append(dict(layers=[(f"{prefix}.mlp.linear_pre.0.weight", f"{prefix}.mlp.linear_pre.0.bias", 2)],
                             tiles=[self._tile(g_z, 128, Seg(fx1), a_op=2, a_gamma=gam2, a_beta=bet2, ldg=256, g_offset=128 * h)
                                    for h in range(2)], M=N, row0s=[0, 128]))
            extra.append(dict(partial=part, out=self._gview2(grads, f"{prefix}.ln_2.weight", f"{prefix}.ln_2.bias"), n_chunks=tiles,
                              chunk_stride=256, rows=1, cols=256))
            keep += [g_z, fx1, part]
        else:
            self.defer(side_pre, g_z, fx1, part)
        # to_out
        if not fused:
            ops.rowtile_chain(N, [Seg(g_fx1)], [LayerSpec(self._T(P[f"{a}.to_out.0.weight"]))], [g_out_x])
        if merge:
            jobs.append(dict(layers=[(f"{a}.to_out.0.weight", f"{a}.to_out.0.bias", 1)],
                             tiles=[self._tile(g_fx1, 128, Seg(sv["out_x"]))], M=N))
            keep += [g_fx1, sv["out_x"]]
        else:
            self.defer(lambda: self._dw_block(grads, [(f"{a}.to_out.0.weight", f"{a}.to_out.0.bias", 1)],
                                              [self._tile(g_fx1, 128, Seg(sv["out_x"]))], N), g_fx1, sv["out_x"])
        # de-slice / attention / slice
        w, batch = sv["w"], pl.batch
        fused_post = self._slice_fuse
        if not fused_post:
            gw = _empty(dev, N, 256)
            L.check(lib.gfv_slice_gw(g_out_x.data_ptr(), sv["out_token"].data_ptr(), None, batch.data_ptr(), gw.data_ptr(), N,
                                     0, st), "slice_gw")
        gpartial = _empty(dev, pl.n_chunks, 256, 17)
        L.check(lib.gfv_slice_token_partial(w.data_ptr(), g_out_x.data_ptr(), pl.chunk_beg.data_ptr(),
                                            pl.chunk_end.data_ptr(), pl.n_chunks, gpartial.data_ptr(), st), "token_partial")
        g_raw, g_norm = _empty(dev, B, 8, 32, 16), _empty(dev, B, 8, 32)
        dwp = _empty(dev, B * 8, 3, 16, 16)
        gpartial, gptr = self._graph_partials(gpartial, pl)
        L.check(lib.gfv_slice_attention_bwd(gpartial.data_ptr(), gptr.data_ptr(), B,
                                            P[f"{a}.to_q.weight"].data_ptr(), P[f"{a}.to_k.weight"].data_ptr(),
                                            P[f"{a}.to_v.weight"].data_ptr(), sv["token"].data_ptr(), sv["norm"].data_ptr(),
                                            sv["attn"].data_ptr(), g_raw.data_ptr(), g_norm.data_ptr(), dwp.data_ptr(), st),
                "slice_attention_bwd")
        def side_qkv():   # parameter gradients only: off the critical path; one launch, straight into the three tensors
            ops.reduce_multi([dict(partial=dwp.data_ptr() + 4 * 256 * i, out=grads.view(f"{a}.{nm}.weight"), n_chunks=B * 8,
                                   chunk_stride=768, rows=1, cols=256) for i, nm in enumerate(("to_q", "to_k", "to_v"))])
        if merge:
            extra += [dict(partial=dwp.data_ptr() + 4 * 256 * i, out=grads.view(f"{a}.{nm}.weight"), n_chunks=B * 8,
                           chunk_stride=768, rows=1, cols=256) for i, nm in enumerate(("to_q", "to_k", "to_v"))]
            keep.append(dwp)
        else:
            self.defer(side_qkv, dwp)
        g_fx_mid = _empty(dev, N, 128)
        nblk = lib.gfv_slice_softmax_bwd_blocks(N)
        sp = _empty(dev, nblk, 552)
        g_x_mid = _empty(dev, N, 128)
        temp = P[f"{a}.graph_temperature"]
        if fused_post:
            # de-slice of g_raw, both slice_gw products and the slice-softmax adjoint in one pass (gw stays in registers)
            L.check(lib.gfv_slice_post_bwd(sv["x_mid"].data_ptr(), P[f"{a}.in_project_slice.weight"].data_ptr(),
                                           P[f"{a}.in_project_slice.bias"].data_ptr(), temp.data_ptr(), w.data_ptr(),
                                           g_out_x.data_ptr(), sv["out_token"].data_ptr(), sv["fx_mid"].data_ptr(),
                                           g_raw.data_ptr(), g_norm.data_ptr(), batch.data_ptr(), g_x_mid.data_ptr(),
                                           g_fx_mid.data_ptr(), sp.data_ptr(), N, B, st), "slice_post_bwd")
        else:
            L.check(lib.gfv_deslice(w.data_ptr(), g_raw.data_ptr(), batch.data_ptr(), g_fx_mid.data_ptr(), N, 0, st), "deslice")
            L.check(lib.gfv_slice_gw(sv["fx_mid"].data_ptr(), g_raw.data_ptr(), g_norm.data_ptr(), batch.data_ptr(),
                                     gw.data_ptr(), N, 1, st), "slice_gw")
            L.check(lib.gfv_slice_softmax_bwd(sv["x_mid"].data_ptr(), P[f"{a}.in_project_slice.weight"].data_ptr(),
                                              P[f"{a}.in_project_slice.bias"].data_ptr(), temp.data_ptr(), w.data_ptr(),
                                              gw.data_ptr(), g_x_mid.data_ptr(), sp.data_ptr(), N, st), "slice_softmax_bwd")
        def side_slice():
            ops.reduce_multi([dict(partial=sp.data_ptr() + 4 * o, out=grads.view(nm), n_chunks=nblk, chunk_stride=552, rows=1,
                                   cols=c)
                              for o, c, nm in ((0, 512, f"{a}.in_project_slice.weight"), (512, 32, f"{a}.in_project_slice.bias"),
                                               (544, 8, f"{a}.graph_temperature"))])
        if merge:
            extra += [dict(partial=sp.data_ptr() + 4 * o, out=grads.view(nm), n_chunks=nblk, chunk_stride=552, rows=1, cols=c)
                      for o, c, nm in ((0, 512, f"{a}.in_project_slice.weight"), (512, 32, f"{a}.in_project_slice.bias"),
                                       (544, 8, f"{a}.graph_temperature"))]
            keep.append(sp)
        else:
            self.defer(side_slice, sp)
        # projections; fx_in also feeds the to_out residual
        t1, g_fx_in = _empty(dev, N, 128), _empty(dev, N, 128)
        Wfxt, Wxt = self._T(P[f"{a}.in_project_fx.weight"]), self._T(P[f"{a}.in_project_x.weight"])
        if ops.stack_ready(Wfxt, Wxt):   # g_fx_in = g_fx_mid Wfx + g_x_mid Wx + g_fx1: one launch, column-stacked pair
            ops.rowtile_chain(N, [Seg(g_fx_mid), Seg(g_x_mid)], [LayerSpec(Wfxt, stack_cols=Wxt)], [g_fx_in], res=[g_fx1])
        else:
            ops.rowtile_chain(N, [Seg(g_fx_mid)], [LayerSpec(Wfxt)], [t1], res=[g_fx1])
            ops.rowtile_chain(N, [Seg(g_x_mid)], [LayerSpec(Wxt)], [g_fx_in], res=[t1])
        if merge:
            jobs.append(dict(layers=[(f"{a}.in_project_x.weight", f"{a}.in_project_x.bias", 1),
                                     (f"{a}.in_project_fx.weight", f"{a}.in_project_fx.bias", 1)],
                             tiles=[self._tile(g_x_mid, 128, Seg(fx_in)), self._tile(g_fx_mid, 128, Seg(fx_in))], M=N))
            keep += [g_x_mid, g_fx_mid, fx_in]
            # the block's four weight-gradient launches, then ONE reduction launch for all of their slab partials, the LayerNorm
            # partials and the attention / slice-projection partials (11 pieces; was six reduction launches)
            self.defer(lambda: self._dw_jobs(grads, jobs, extra), *keep)
        else:
            self.defer(lambda: self._dw_block(grads, [(f"{a}.in_project_x.weight", f"{a}.in_project_x.bias", 1),
                                                      (f"{a}.in_project_fx.weight", f"{a}.in_project_fx.bias", 1)],
                                              [self._tile(g_x_mid, 128, Seg(fx_in)), self._tile(g_fx_mid, 128, Seg(fx_in))], N),
                       g_x_mid, g_fx_mid, fx_in)
        self.flush()   # the block's parameter gradients: one fork
        return g_fx_in

    # ------------------------------------------------------------------------------------------------------------
    # finite-volume integrator (FVscheme.py:618-724 -> conserved_form :50-274)
    # ------------------------------------------------------------------------------------------------------------
    def fvm_fwd(self, dec, uv_old, pl, want_outputs=True, train_loss=None):
        lib = L.load()
        st = L.stream_ptr()
        N, E, C, B = pl.N, pl.E, pl.C, pl.B
        dev = dec.device
        phi = _empty(dev, N, 8)
        L.check(lib.gfv_phi_fwd(dec.data_ptr(), pl.y.data_ptr(), pl.node_type.data_ptr(), uv_old.data_ptr(), phi.data_ptr(),
                                N, self.mode, st), "phi_fwd")
        losses, uvp_node, uvp_cell, sv = self.fvm_core_fwd(phi, pl, want_outputs, train_loss=train_loss)
        sv["dec"] = dec
        return losses, uvp_node, uvp_cell, sv

    def _fvm_mesh(self, pl, raw=False):
        """gfv_fvm_mesh_t of a plan (include/gfv.h), built once per plan and output mode and kept on it."""
        cache = pl.__dict__.setdefault("_fvm_mesh_cache", {})
        key = (self.mode, self.smooth, self.order_terms, bool(raw))
        m = cache.get(key)
        if m is None:
            m = L.FvmMesh(N=pl.N, E=pl.E, C=pl.C, B=pl.B, terms=pl.M, mode=self.mode, smooth=self.smooth | (2 if raw else 0), reserved=0)
            for name, t in (("node_type", pl.node_type), ("batch", pl.batch), ("ftype", pl.ftype), ("cbatch", pl.cbatch),
                            ("gcell_ptr", pl.gcell_ptr), ("pos", pl.pos), ("fpos", pl.fpos), ("y", pl.y), ("centroid", pl.centroid),
                            ("area", pl.area), ("theta", pl.theta), ("sigma", pl.sigma), ("uvp_dim", pl.uvp_dim), ("dt", pl.dt),
                            ("crow", pl.crow), ("kcell", pl.kcell), ("kS", pl.kS), ("frow", pl.frow), ("fk", pl.fk),
                            ("nfrow", pl.n_rowptr), ("nfcol2", pl.n_col_edge2), ("nrow", pl.nrow), ("ncell", pl.ncell),
                            ("An", pl.An), ("rn", pl.rn), ("xo_rowptr", pl.xo_rowptr), ("xo_in", pl.xo_in), ("xo_B", pl.xo_B),
                            ("sumB", pl.sumB)):
                setattr(m, name, t.data_ptr())
            cache[key] = m
        return m

    def _fvm_counter(self, dev):
        if self._fvm_cnt is None or self._fvm_cnt.device != dev:
            self._fvm_cnt = torch.zeros(4, dtype=torch.int32, device=dev)   # (arrival counter of the tail launch: zero between launches)
        return self._fvm_cnt

    def fvm_core_fwd(self, phi, pl, want_outputs=True, raw_outputs=False, train_loss=None):
        """phi [N,8] = (uvp_new, uv_hat, uv_old, 0) -> residual losses [B,4], smoothed node field, cell field.
        train_loss = (hyper [8], loss [1], gloss [B,4]) device tensors: the training loss of pre_train_Adam.py:177-184 and its
        gradient with respect to the four residuals are formed behind the residual norms (TrainStep).
        raw_outputs: the two fields as the reference's stand-alone Intergrator returns them (FVscheme.py:253-262,718-724) - the
        smoothed node field before the Dirichlet overwrite and neither field re-dimensionalised (importer.py:223-231 does both
        afterwards)."""
        lib = L.load()
        st = L.stream_ptr()
        N, E, C, B = pl.N, pl.E, pl.C, pl.B
        dev = phi.device
        grad = _empty(dev, N, 16)
        # pl.M: Taylor terms of the reconstruction order the mesh's moment matrices were built for (2 / 5 / 9 / 14)
        if pl.M != self.order_terms:
            raise ValueError(f"params.order needs {self.order_terms}-term WLSQ moments, the batch carries {pl.M}-term ones "
                             "(Load_mesh.py:543-546 builds them for params.order)")
        L.check(lib.gfv_wlsq_fwd_ex(phi.data_ptr(), pl.x_rowptr.data_ptr(), pl.x_out.data_ptr(), pl.x_B.data_ptr(),
                                    pl.An.data_ptr(), pl.rn.data_ptr(), grad.data_ptr(), None, N, pl.M, st), "wlsq_fwd")
        Ff = _empty(dev, E, 16)
        L.check(lib.gfv_face_fwd(phi.data_ptr(), grad.data_ptr(), pl.es.data_ptr(), pl.er.data_ptr(), pl.pos.data_ptr(),
                                 pl.fpos.data_ptr(), pl.ftype.data_ptr(), pl.y.data_ptr(), Ff.data_ptr(), E, st), "face_fwd")
        phic, cres = _empty(dev, C, 8), _empty(dev, C, 4)
        uvp_cell = _empty(dev, C, 3) if want_outputs else None
        gradc = _empty(dev, C, 16) if self.nc else None
        L.check(lib.gfv_cell_fwd_ex(phi.data_ptr(), grad.data_ptr(), Ff.data_ptr(), pl.pos.data_ptr(), pl.crow.data_ptr(),
                                    pl.kface.data_ptr(), pl.knode.data_ptr(), pl.kS.data_ptr(), pl.ftype.data_ptr(),
                                    pl.centroid.data_ptr(), pl.area.data_ptr(), pl.cbatch.data_ptr(), pl.theta.data_ptr(),
                                    pl.dt.data_ptr(), pl.uvp_dim.data_ptr(), pl.sigma.data_ptr(), phic.data_ptr(),
                                    cres.data_ptr(), None if uvp_cell is None else uvp_cell.data_ptr(), C, self.nc,
                                    None if gradc is None else gradc.data_ptr(), st), "cell_fwd")
        sums, losses = _empty(dev, B, 4), _empty(dev, B, 4)
        uvp_node = _empty(dev, N, 3) if want_outputs else None
        if self._fvm_fuse:
            # residual norms + (train: the training loss and its gradient, by the workgroup that sees the last graph's norms) +
            # node smoothing as ONE launch (round 6, csrc/fvm.hip fvm_tail_kernel)
            tl = train_loss
            L.check(lib.gfv_fvm_fwd_tail(self._fvm_mesh(pl, raw_outputs), cres.data_ptr(), phic.data_ptr(), phi.data_ptr(),
                                         sums.data_ptr(), losses.data_ptr(), None if uvp_node is None else uvp_node.data_ptr(),
                                         None if tl is None else tl[0].data_ptr(), None if tl is None else tl[1].data_ptr(),
                                         None if tl is None else tl[2].data_ptr(), self._fvm_counter(dev).data_ptr(), st), "fvm_fwd_tail")
        else:
            L.check(lib.gfv_graph_loss(cres.data_ptr(), pl.gcell_ptr.data_ptr(), pl.theta.data_ptr(), pl.sigma.data_ptr(),
                                       sums.data_ptr(), losses.data_ptr(), B, st), "graph_loss")
            if want_outputs:
                L.check(lib.gfv_cell_to_node(phic.data_ptr(), pl.nrow.data_ptr(), pl.ncell.data_ptr(), pl.pos.data_ptr(),
                                             pl.centroid.data_ptr(), pl.node_type.data_ptr(), pl.y.data_ptr(),
                                             pl.batch.data_ptr(), pl.uvp_dim.data_ptr(), pl.sigma.data_ptr(), phi.data_ptr(),
                                             self.smooth | (2 if raw_outputs else 0), uvp_node.data_ptr(), N, st), "cell_to_node")
            if train_loss is not None:
                L.check(lib.gfv_train_loss_dev(losses.data_ptr(), B, train_loss[0].data_ptr(), train_loss[1].data_ptr(),
                                               train_loss[2].data_ptr(), st), "train_loss")
        if want_outputs and raw_outputs:
            uvp_cell = phic[:, 0:3].clone()
        sv = dict(Ff=Ff, cres=cres, sums=sums, phi=phi, grad=grad, phic=phic, gradc=gradc)
        return losses, uvp_node, uvp_cell, sv

    def fvm_bwd(self, sv, gloss, pl):
        lib = L.load()
        if self._fvm_fuse and not self.nc:
            # conserved form: three launches from the loss gradients to the decoder-output gradient (round 6, csrc/fvm.hip)
            dev = gloss.device
            gFf, gphi, grhs = _empty(dev, pl.E, 16), _empty(dev, pl.N, 8), _empty(dev, pl.N, 8, pl.M)
            gdec = _empty(dev, pl.N, 3)
            L.check(lib.gfv_fvm_bwd_fused(self._fvm_mesh(pl), sv["cres"].data_ptr(), sv["sums"].data_ptr(), gloss.data_ptr(),
                                          sv["Ff"].data_ptr(), sv["dec"].data_ptr(), gFf.data_ptr(), gphi.data_ptr(), grhs.data_ptr(),
                                          gdec.data_ptr(), L.stream_ptr()), "fvm_bwd_fused")
            return gdec
        gphi = self.fvm_core_bwd(sv, gloss, pl)
        gdec = _empty(gloss.device, pl.N, 3)
        L.check(lib.gfv_phi_bwd(gphi.data_ptr(), sv["dec"].data_ptr(), pl.node_type.data_ptr(), gdec.data_ptr(), pl.N,
                                self.mode, L.stream_ptr()), "phi_bwd")
        return gdec

    def fvm_core_bwd(self, sv, gloss, pl):
        """-> gphi [N,8] (channels 0..4 carry gradient; uv_old has none)."""
        lib = L.load()
        st = L.stream_ptr()
        N, E, C = pl.N, pl.E, pl.C
        dev = gloss.device
        gc, gFf = _empty(dev, C, 4), _empty(dev, E, 16)
        gphi, ggrad = _empty(dev, N, 8), _empty(dev, N, 16)
        L.check(lib.gfv_fvm_bwd_ex(sv["cres"].data_ptr(), sv["sums"].data_ptr(), gloss.data_ptr(), sv["Ff"].data_ptr(),
                                pl.cbatch.data_ptr(), pl.theta.data_ptr(), pl.sigma.data_ptr(), pl.dt.data_ptr(),
                                pl.frow.data_ptr(), pl.fk.data_ptr(), pl.kcell.data_ptr(), pl.kS.data_ptr(),
                                pl.ftype.data_ptr(), pl.n_rowptr.data_ptr(), pl.n_col_edge2.data_ptr(), pl.nrow.data_ptr(),
                                pl.ncell.data_ptr(), pl.crow.data_ptr(), pl.pos.data_ptr(), pl.fpos.data_ptr(),
                                pl.centroid.data_ptr(), pl.area.data_ptr(), gc.data_ptr(), gFf.data_ptr(), gphi.data_ptr(),
                                ggrad.data_ptr(), N, E, C, self.nc, None if not self.nc else sv["gradc"].data_ptr(),
                                None if not self.nc else sv["phic"].data_ptr(), st), "fvm_bwd")
        grhs = _empty(dev, N, 8, pl.M)
        L.check(lib.gfv_wlsq_bwd_ex(ggrad.data_ptr(), None, pl.An.data_ptr(), pl.rn.data_ptr(), pl.xo_rowptr.data_ptr(),
                                    pl.xo_in.data_ptr(), pl.xo_B.data_ptr(), pl.sumB.data_ptr(), grhs.data_ptr(),
                                    gphi.data_ptr(), N, pl.M, st), "wlsq_bwd")
        return gphi

    # ------------------------------------------------------------------------------------------------------------
    # input preparation (importer.py:166-178)
    # ------------------------------------------------------------------------------------------------------------
    def prep_fwd(self, x, buffers, pl, norm_global, accumulate, want_edge_attr15=True, x_raw=None):
        """In place on x [N,12]; returns (uv_old [N,2], edge_attr16 [E,16], edge_attr15 [E,15] or None).
        Round 6: two launches - the per-graph statistics (+ the Normalizer's mean / std when it does not accumulate; + a copy of
        the raw rows when the caller has none), then node normalisation and edge features together (csrc/misc.hip prep_stats /
        prep_apply).  x_raw: the un-normalised rows in a tensor of the caller's that survives the step (TrainStep's backup: its
        per-step restore copy is then not needed at all); None: `x` holds them and is normalised in place (importer.py:123-130)."""
        lib = L.load()
        st = L.stream_ptr()
        N, E, B = pl.N, pl.E, pl.B
        dev = x.device
        assert x.is_contiguous() and x.shape[1] == 12
        if os.environ.get("GFV_PREP_FUSE", "1") == "0" or B > 1024:   # (the fused launch keeps 1024 arrival counters)
            return self._prep_fwd_unfused(x, buffers, pl, norm_global, accumulate, want_edge_attr15, x_raw)
        stats = _empty(dev, B, 6)
        ws = self._prep_ws
        need = lib.gfv_prep_workspace_bytes(B) // 4
        if ws is None or ws.numel() < need or ws.device != dev:
            # (allocated by the warm-up steps that precede any recording / capture; the arrival counters start at zero and every
            # launch leaves them at zero)
            ws = self._prep_ws = torch.zeros(need, dtype=torch.float32, device=dev)
        mean_std = _empty(dev, 18)
        fused_norm = False
        if norm_global:
            sync = accumulate and (self.dist_world > 1 or self.dist_force)
            if accumulate:
                nb = lib.gfv_normalizer_blocks(N)
                pws = _empty(dev, nb, 18)
                src = x if x_raw is None else x_raw
                if sync:
                    from . import parallel
                    before = parallel.snapshot_normalizer(buffers)

                def update(acc):
                    L.check(lib.gfv_normalizer_update(src.data_ptr(), 12, N, 1 if acc else 0, buffers["acc_count"].data_ptr(),
                                                      buffers["num_accumulations"].data_ptr(), buffers["acc_sum"].data_ptr(),
                                                      buffers["acc_sum_squared"].data_ptr(), pws.data_ptr(),
                                                      mean_std.data_ptr(), L.stream_ptr()), "normalizer_update")
                update(True)
                if sync:
                    # statistics of the GLOBAL batch on every rank, then mean / std recomputed from them (finalize only)
                    parallel.allreduce_normalizer(buffers, before, self.dist_world, self.dist_group, force=self.dist_force)
                    update(False)
            else:
                fused_norm = True   # mean / std from the running buffers inside the statistics launch
        own_raw = x_raw is None
        if own_raw:
            x_raw = _empty(dev, N, 12)
        L.check(lib.gfv_prep_stats((x if own_raw else x_raw).data_ptr(), 12, pl.gnode_ptr.data_ptr(), B, stats.data_ptr(), ws.data_ptr(),
                                   x_raw.data_ptr() if own_raw else None,
                                   buffers["acc_count"].data_ptr() if fused_norm else None,
                                   buffers["acc_sum"].data_ptr() if fused_norm else None,
                                   buffers["acc_sum_squared"].data_ptr() if fused_norm else None,
                                   mean_std.data_ptr() if fused_norm else None, st), "prep_stats")
        uv_old = _empty(dev, N, 2)
        ea16 = _empty(dev, E, 16)
        ea15 = _empty(dev, E, 15) if want_edge_attr15 else None
        L.check(lib.gfv_prep_apply(x_raw.data_ptr(), x.data_ptr(), pl.batch.data_ptr(), stats.data_ptr(), pl.uvp_dim.data_ptr(),
                                   mean_std.data_ptr(), 1 if norm_global else 0, uv_old.data_ptr(), N, pl.pos.data_ptr(),
                                   pl.es.data_ptr(), pl.er.data_ptr(), ea16.data_ptr(),
                                   None if ea15 is None else ea15.data_ptr(), E, st), "prep_apply")
        return uv_old, ea16, ea15

    def _prep_fwd_unfused(self, x, buffers, pl, norm_global, accumulate, want_edge_attr15=True, x_raw=None):
        """The six-launch form of rounds 1 - 5 (GFV_PREP_FUSE=0; what the fused launches are tested against)."""
        lib = L.load()
        st = L.stream_ptr()
        N, E, B = pl.N, pl.E, pl.B
        dev = x.device
        if x_raw is not None:
            cmdlist.call(x.copy_, x_raw)
        stats = _empty(dev, B, 6)
        # per-graph mean / std over 64 workgroups per graph (double partial sums, folded in a fixed order)
        nws = _empty(dev, B, 64 * 12)   # = gfv_graph_norm_workspace_bytes(B) / 4 floats (64 x 6 doubles per graph)
        L.check(lib.gfv_graph_norm_stats_ws(x.data_ptr(), 12, pl.gnode_ptr.data_ptr(), B, stats.data_ptr(), nws.data_ptr(), st),
                "norm_stats")
        mean_std = _empty(dev, 18)
        if norm_global:
            nb = lib.gfv_normalizer_blocks(N)
            pws = _empty(dev, nb, 18)
            sync = accumulate and (self.dist_world > 1 or self.dist_force)
            if sync:
                from . import parallel
                before = parallel.snapshot_normalizer(buffers)

            def update(acc):
                L.check(lib.gfv_normalizer_update(x.data_ptr(), 12, N, 1 if acc else 0, buffers["acc_count"].data_ptr(),
                                                  buffers["num_accumulations"].data_ptr(), buffers["acc_sum"].data_ptr(),
                                                  buffers["acc_sum_squared"].data_ptr(), pws.data_ptr(),
                                                  mean_std.data_ptr(), L.stream_ptr()), "normalizer_update")
            update(accumulate)
            if sync:
                # statistics of the GLOBAL batch on every rank, then mean / std recomputed from them (finalize only)
                parallel.allreduce_normalizer(buffers, before, self.dist_world, self.dist_group, force=self.dist_force)
                update(False)
        uv_old = _empty(dev, N, 2)
        L.check(lib.gfv_node_prep(x.data_ptr(), 12, pl.batch.data_ptr(), stats.data_ptr(), pl.uvp_dim.data_ptr(),
                                  mean_std.data_ptr(), 1 if norm_global else 0, uv_old.data_ptr(), N, st), "node_prep")
        ea16 = _empty(dev, E, 16)
        ea15 = _empty(dev, E, 15) if want_edge_attr15 else None
        L.check(lib.gfv_edge_attr(x.data_ptr(), 12, pl.pos.data_ptr(), pl.es.data_ptr(), pl.er.data_ptr(), ea16.data_ptr(),
                                  None if ea15 is None else ea15.data_ptr(), E, st), "edge_attr")
        return uv_old, ea16, ea15

    # ------------------------------------------------------------------------------------------------------------
    # simulator (TransFVGN_v2.py:54-105 / EPD.py:222-270)
    # ------------------------------------------------------------------------------------------------------------
    def simulator_fwd(self, P, x, ea16, pl, prefix="simulator"):
        N, E = pl.N, pl.E
        xn, _, sv_nenc = self.mlp3_fwd(P, f"{prefix}.encoder.nb_encoder", N, [Seg(x, width=12, ld=12)])
        en, _, sv_eenc = self.mlp3_fwd(P, f"{prefix}.encoder.eb_encoder", E, [Seg(ea16, width=15, ld=16)])
        procs = []
        if self.net == "EPD":
            blocks = []
            for ig in range(self.mp):
                xn, en, sv = self.gn_fwd(P, f"{prefix}.GN_block_list.{ig}", xn, en, pl)
                blocks.append(sv)
            procs.append(dict(blocks=blocks, trans=None))
        else:
            # TransFVGN_v2: two AttnProcessors (TransFVGN_v2.py:69-76); TransFVGN_v1: one processor whose modules hang
            # directly off the simulator (TransFVGN_v1.py:30-46,53-74)
            pps = [prefix] if self.net == "TransFVGN_v1" else [f"{prefix}.processpr_list.{ip}" for ip in range(self.n_proc)]
            for pp in pps:
                emb = xn
                blocks = []
                for ig in range(self.mp):
                    xn, en, sv = self.gn_fwd(P, f"{pp}.GN_block_list.{ig}", xn, en, pl)
                    blocks.append(sv)
                xn, svt = self.trans_fwd(P, f"{pp}.TransBlock", xn, emb, pl)
                procs.append(dict(blocks=blocks, trans=svt))
        dec, _, sv_dec = self.mlp3_fwd(P, f"{prefix}.decoder.node_decode_module", N, [Seg(xn)], ln=False)
        return dec, dict(sv_nenc=sv_nenc, sv_eenc=sv_eenc, procs=procs, sv_dec=sv_dec)

    def simulator_bwd(self, P, sv, g_dec, grads, pl):
        N, E = pl.N, pl.E
        dev = g_dec.device
        g_x = _empty(dev, N, 128)
        self.mlp3_bwd(P, sv["sv_dec"], g_dec, grads, outs=[g_x])
        self.flush()
        g_e = None
        pending = None   # an addend of g_x that the next consumer folds into its first launch
        for proc in reversed(sv["procs"]):
            g_emb = None
            if proc["trans"] is not None:
                g_x = self.trans_bwd(P, proc["trans"], g_x, grads, pl, g_add=pending)   # grad wrt (x_GN + emb)
                pending = None
                g_emb = g_x
            elif pending is not None:
                g_x, pending = g_x + pending, None
            for blk in reversed(proc["blocks"]):
                if g_e is None:
                    # last block's edges feed nothing: a zero gradient, kept across steps (read-only: gn_bwd returns a
                    # fresh tensor for the block before)
                    if self._zero_e is None or self._zero_e.shape[0] != E or self._zero_e.device != dev:
                        self._zero_e = torch.zeros((E, 128), dtype=torch.float32, device=dev)
                    g_e = self._zero_e
                self._last_gn = proc is sv["procs"][0] and blk is proc["blocks"][0]
                g_x, g_e = self.gn_bwd(P, blk, g_x, g_e, grads, pl)
            if g_emb is not None:
                pending = g_emb  # the processor input also entered the Transolver residual (TransFVGN_v2.py:46-49)
            if self.bucket_hook is not None and proc is sv["procs"][-1] and len(sv["procs"]) > 1:
                self.flush()
                self.bucket_hook()
        # (the two encoders end the backward; running the big one - edge encoder, 75 k rows - first so that its weight
        # gradient overlaps the node encoder's chain was measured: 4.892 against 4.877 ms / step in this order)
        tail = self._tail_cfg(pl)[0] if self._defer_mode else 0   # how many of the trailing flushes run on the main stream
        self.mlp3_bwd(P, sv["sv_eenc"], g_e, grads)
        self.flush(on_main=tail >= 2)
        self.mlp3_bwd(P, sv["sv_nenc"], g_x, grads, g_add=pending)
        self.flush(on_main=tail >= 1)

    # ------------------------------------------------------------------------------------------------------------
    # whole model (importer.py:156-240)
    # ------------------------------------------------------------------------------------------------------------
    def forward(self, P, buffers, x, pl, *, norm_global=True, accumulate=True, want_outputs=True, want_edge_attr15=True,
                before_prep=None, x_raw=None, train_loss=None):
        """before_prep: main-stream work of the caller that only the input preparation waits for (TrainStep's restore of the
        un-normalised node state): issued BEHIND the fork, so that the side stream's image build - which the first encoder launch
        waits for, ~16 us in round 5's timelines - starts a copy and a cross-queue signal earlier."""
        # the per-step weight images are built on the side stream while the input preparation runs on the main one
        with self.fork():
            prev = self._wi_enter("fwd", P)
        try:
            if before_prep is not None:
                before_prep()
            uv_old, ea16, ea15 = self.prep_fwd(x, buffers, pl, norm_global, accumulate, want_edge_attr15, x_raw=x_raw)
            self.join()
            dec, sv_sim = self.simulator_fwd(P, x, ea16, pl)
        finally:
            self._wi_exit("fwd", prev)
        losses, uvp_node, uvp_cell, sv_fvm = self.fvm_fwd(dec, uv_old, pl, want_outputs, train_loss=train_loss)
        return losses, uvp_node, uvp_cell, ea15, dict(sim=sv_sim, fvm=sv_fvm)

    def backward(self, P, ctx, gloss, grads, pl):
        """gloss [B,4] = dL/d(cont, mom_x, mom_y, press); fills `grads` (name -> preallocated tensor)."""
        with self.fork():   # transposed copies + their images: beside the finite-volume adjoint
            self.prepare_transposes(P)
            prev = self._wi_enter("bwd", P)
        try:
            g_dec = self.fvm_bwd(ctx["fvm"], gloss, pl)
            self.join()
            self.simulator_bwd(P, ctx["sim"], g_dec, grads, pl)
            self.join()
        finally:
            self._wt_live = False  # the cached transposes are only valid for this step's parameter values
            self._wi_exit("bwd", prev)
        return grads
