"""`gfv.optim.Adam` - torch.optim.Adam for users who change ONE import: the same constructor, `step()`, `zero_grad()`,
`param_groups` (so `torch.optim.lr_scheduler.*` drive it), `state_dict()` / `load_state_dict()` nesting - over flat fp32
buffers and the library's fused Adam launch (include/gfv.h gfv_adam_step_dev) instead of ~12 `_foreach_*` passes over 159
tensors (pre_train_Adam.py:79,191; solve_with_grad_GPU.py:181).

What it does to the model: the parameters become views of one flat buffer (16-byte aligned, in the order they were given - as
`gfv.trainer.TrainStep` lays them out); `NNmodel`'s backward hands out gradients as views of one flat tensor in the same
layout, which `step()` recognises and feeds to the kernel as it is (otherwise the gradients are gathered into a flat buffer
first: still one optimiser launch).  Supported: one parameter group, `weight_decay=0`, `amsgrad=False`, `maximize=False`
(torch.optim.Adam's defaults, what both reference drivers use); anything else raises at construction.
"""
from __future__ import annotations

import torch

from . import lib as L
from .engine import GradStore


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None, grad_scale=1.0):
        if weight_decay != 0 or amsgrad or maximize or differentiable:
            raise NotImplementedError("gfv.optim.Adam: weight_decay=0, amsgrad=False, maximize=False only (torch.optim.Adam's "
                                      "defaults; the reference drivers use nothing else)")
        defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=0, amsgrad=False,
                        maximize=False)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("gfv.optim.Adam: one parameter group (one flat buffer, one launch)")
        ps = self.param_groups[0]["params"]
        if not ps or any((not p.is_cuda) or p.dtype != torch.float32 for p in ps):
            raise RuntimeError("gfv.optim.Adam: fp32 parameters on the GPU (HIP kernels only, no CPU fallback)")
        dev = ps[0].device
        self._params = list(ps)
        self.G = GradStore([str(i) for i in range(len(ps))], [p.shape for p in ps], dev)
        total = self.G.total
        self.flat_g = self.G.flat
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._offs = [self.G.off[str(i)] for i in range(len(ps))]
        self._adopt()
        self.adam_state = torch.zeros(16, dtype=torch.float32, device=dev)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._grad_scale = float(grad_scale)
        self._hyper_host = None
        self._sync_hyper()
        L.status_mirror()   # the launch publishes the device status word (include/gfv.h gfv_status_mirror)

    # parameters as views of the flat buffer -------------------------------------------------------------------------
    def _adopt(self):
        for p, off in zip(self._params, self._offs):
            if p.data_ptr() != self.flat_p.data_ptr() + 4 * off:
                view = self.flat_p[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view

    def _sync_hyper(self, steps_done=None):
        g = self.param_groups[0]
        vals = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self._grad_scale, 0.0, 0.0, 0.0)
        moved = self._hyper_host is None or vals[1:3] != self._hyper_host[1:3]
        if vals != self._hyper_host:
            self.hyper.copy_(torch.tensor(vals, dtype=torch.float32))
            self._hyper_host = vals
        if moved or steps_done is not None:
            t = float(self.adam_state[0]) if steps_done is None else float(steps_done)
            L.check(L.load().gfv_adam_state_init(self.adam_state.data_ptr(), float(g["betas"][0]), float(g["betas"][1]), t,
                                                 L.stream_ptr()), "adam_state_init")

    def _flat_grad(self):
        """The gradients as ONE flat tensor in this object's layout.  NNmodel's backward returns exactly that - views of one
        flat tensor (gfv/functions.py ModelFn.backward), which autograd keeps as `.grad` without copying: recognised by every
        gradient sitting at its offset of one storage of the right size - else gathered into a flat buffer first."""
        ps, offs = self._params, self._offs
        grads = [p.grad for p in ps]
        first = next((i for i, g in enumerate(grads) if g is not None), None)
        if first is None:
            return None
        g0 = grads[first]
        st = g0.untyped_storage()
        b0 = g0.data_ptr() - 4 * offs[first]
        if (g0.dtype == torch.float32 and st.data_ptr() == b0 and st.nbytes() >= 4 * self.G.total
                and all(g is not None and g.data_ptr() == b0 + 4 * o for g, o in zip(grads, offs) if g is not None)
                and self._none_grad_ok(grads)):
            return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, 0, (self.G.total,), (1,))
        flat = self.flat_g
        views, srcs = [], []
        for p, g, off in zip(ps, grads, offs):
            v = flat[off:off + p.numel()]
            if g is None:
                v.zero_()
            else:
                views.append(v.view(p.shape))
                srcs.append(g if g.dtype == torch.float32 else g.float())
        torch._foreach_copy_(views, srcs)
        return flat

    def _none_grad_ok(self, grads):
        """A parameter WITHOUT a gradient must keep its value (torch.optim.Adam skips it).  In NNmodel's flat gradient the slots of
        the parameters that never receive one (the unused ln_1 / Attn.temperature: gfv.functions.unused_param_names) hold zeros,
        which leaves m, v and the parameter as they are; any OTHER missing gradient (a layer frozen by `p.grad = None`) has a
        live value in its slot, so the flat tensor cannot be used as it is.  The backward says which slots are of the first kind
        (gfv.functions.LAST_FLAT: storage address + positions without a gradient)."""
        from . import functions as GF
        info = GF.LAST_FLAT
        none = frozenset(i for i, g in enumerate(grads) if g is None)
        first = next(g for g in grads if g is not None)
        return info is not None and info[0] == first.untyped_storage().data_ptr() and info[1] == none

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L.raise_on_status("gfv.optim.Adam.step")
        self._adopt()        # (a .to() / load_state_dict(assign=True) since the last step re-pointed the parameters)
        self._sync_hyper()   # lr_scheduler.step() edits param_groups[0]["lr"]
        g = self._flat_grad()
        if g is None:
            return loss
        L.check(L.load().gfv_adam_step_dev(self.flat_p.data_ptr(), g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                           self.G.total, self.adam_state.data_ptr(), self.hyper.data_ptr(), L.stream_ptr()),
                "adam_step")
        return loss

    # torch.optim.Adam's checkpoint nesting (importer.py:292-313 stores it under `optimizer0`) ---------------------------
    def state_dict(self):
        t = self.adam_state[0:1].detach().cpu().clone().reshape(())
        state = {}
        for i, (p, off) in enumerate(zip(self._params, self._offs)):
            k = p.numel()
            state[i] = {"step": t.clone(), "exp_avg": self.flat_m[off:off + k].view(p.shape).detach().cpu().clone(),
                        "exp_avg_sq": self.flat_v[off:off + k].view(p.shape).detach().cpu().clone()}
        g = self.param_groups[0]
        group = {k: v for k, v in g.items() if k != "params"}
        group["params"] = list(range(len(self._params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        if len(g["params"]) != len(self._params):
            raise ValueError("optimizer state belongs to a different parameter set")
        step = None
        self.flat_m.zero_()
        self.flat_v.zero_()
        for i, st in sd["state"].items():
            i = int(i)
            off, k = self._offs[i], self._params[i].numel()
            self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            s = float(st["step"])
            step = s if step is None else step
            if s != step:
                raise ValueError("per-parameter step counts differ: not a state this fused Adam can resume")
        for k, v in g.items():
            if k != "params":
                self.param_groups[0][k] = v
        self._sync_hyper(steps_done=0.0 if step is None else step)
