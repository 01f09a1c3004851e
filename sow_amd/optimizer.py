"""TT optimizers -- host-side mirror of tn_gradient/optimizer/ttadam.py and ttsgd.py.

Optimizer state is held as TensorTrain cores between steps (decompress -> dense update -> recompress);
the dense update is one fused kernel (sow_ttadam_dense), (de)compression uses the QR / GEMM kernels.
FactorAdamW is the fused flat-bucket AdamW for the SoW factor group.
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, Tuple

import torch
import torch.nn as nn

from . import ops
from .tt import TensorTrain


class TTAdam(torch.optim.Optimizer):
    """ttadam.py:10-117."""

    def __init__(self, params: Iterable[nn.parameter.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0, amsgrad: bool = False, correct_bias: bool = True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, correct_bias=correct_bias)
        super().__init__(params, defaults)

    batched = True    # TT groups step through sow_ttadam_batch (one launch sequence for all parameters); False: per parameter

    def _step_tt_group_batched(self, group) -> set:
        """ttadam.py:68-115 for every parameter of a TT group in ONE C call (sow_ttadam_batch): reconstruct m and v, clamp,
        Adam update, re-decompose -- 1 + 4 (order - 1) launches per 8 parameters instead of ~40 per parameter.  Returns the
        ids of the parameters it stepped; the others (non-fp32, CPU, ranks the kernels do not cover) keep the loop below."""
        import ctypes

        from . import _lib
        from .tt import _empty_train, _tt_desc
        lib = _lib.load()
        ranks = list(group["ranks"])
        beta1, beta2 = group["betas"]
        items, keep, done = [], [], set()
        for p in group["params"]:
            if p.grad is None or p.grad.is_sparse or p.dim() != 2:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                continue
            grad = p.grad if (p.grad.dtype == torch.float32 and p.grad.stride(1) == 1) else p.grad.float().contiguous()
            state = self.state[p]
            has = "exp_avg" in state and "exp_avg_sq" in state
            if has:
                tm, tv = state["exp_avg"], state["exp_avg_sq"]
                if not (isinstance(tm, TensorTrain) and isinstance(tv, TensorTrain) and list(tm.ranks) == ranks
                        and list(tv.ranks) == ranks):
                    continue
            else:
                tm, tv = _empty_train(ranks, p.shape, p.device), _empty_train(ranks, p.shape, p.device)
            dm = _tt_desc(tm.cores, tm.ranks, tm.input_shape, tm.output_shape, p.shape[0], p.shape[1])
            dv = _tt_desc(tv.cores, tv.ranks, tv.input_shape, tv.output_shape, p.shape[0], p.shape[1])
            if dm is None or dv is None:
                continue
            step = state.get("step", 0) + 1
            step_size = group["lr"]
            if group["correct_bias"]:
                step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
            nws = int(lib.sow_ttadam_workspace_bytes(ctypes.byref(dm)))
            ws = torch.empty(nws, dtype=torch.uint8, device=p.device)
            it = _lib.TtAdamItem()
            it.m, it.v = dm, dv
            it.param, it.grad, it.ld_param, it.ld_grad = p.data_ptr(), grad.data_ptr(), p.stride(0), grad.stride(0)
            it.step_size = step_size
            it.lr_times_wd = group["lr"] * group["weight_decay"] if group["weight_decay"] > 0.0 else 0.0
            it.has_state, it.workspace, it.workspace_bytes = int(has), ws.data_ptr(), nws
            items.append(it)
            keep.append((p, grad, ws, tm, tv, step))
        if not items:
            return done
        by_dev = {}
        for it, k in zip(items, keep):
            by_dev.setdefault(k[0].device, []).append((it, k))
        for dev, lst in by_dev.items():
            arr = (_lib.TtAdamItem * len(lst))(*[it for it, _ in lst])
            ops._launch(dev, "sow_ttadam_batch", lib.sow_ttadam_batch, arr, len(lst), float(beta1), float(beta2),
                        float(group["eps"]))
            for _, (p, _, _, tm, tv, step) in lst:
                st = self.state[p]
                st["step"], st["exp_avg"], st["exp_avg_sq"] = step, tm, tv
                done.add(id(p))
        return done

    @torch.no_grad()
    def step(self, closure: Callable = None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            done = self._step_tt_group_batched(group) if ("ranks" in group and self.batched) else set()
            for p in group["params"]:
                if p.grad is None or id(p) in done:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError("TTAdam does not support sparse gradients, please consider SparseAdam instead")
                state = self.state[p]
                if "step" not in state:
                    state["step"] = 0
                tt_mode = "ranks" in group
                clamp = False
                if "exp_avg" not in state:
                    m = torch.zeros_like(grad, dtype=torch.float32)
                elif tt_mode:
                    m = state["exp_avg"].to_matrix(grad.shape).contiguous().float()   # ttadam.py:71-74
                else:
                    m = state["exp_avg"]
                if "exp_avg_sq" not in state:
                    v = torch.zeros_like(grad, dtype=torch.float32)
                elif tt_mode:
                    v = state["exp_avg_sq"].to_matrix(grad.shape).contiguous().float()  # ttadam.py:79-84
                    clamp = True                                                         # v[v < 0] = 0
                else:
                    v = state["exp_avg_sq"]
                state["step"] += 1
                beta1, beta2 = group["betas"]
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - beta2 ** state["step"]) / (1.0 - beta1 ** state["step"])
                lr_wd = group["lr"] * group["weight_decay"] if group["weight_decay"] > 0.0 else 0.0
                if p.dtype == torch.float32 and p.is_contiguous():
                    ops.ttadam_dense_(p.data, grad.contiguous().float(), m, v, beta1=beta1, beta2=beta2, eps=group["eps"],
                                      step_size=step_size, lr_times_wd=lr_wd, clamp_v=clamp)
                else:
                    p32 = p.data.float().contiguous()
                    ops.ttadam_dense_(p32, grad.contiguous().float(), m, v, beta1=beta1, beta2=beta2, eps=group["eps"],
                                      step_size=step_size, lr_times_wd=lr_wd, clamp_v=clamp)
                    p.data.copy_(p32)
                if tt_mode:
                    state["exp_avg"] = TensorTrain.from_matrix(m, ranks=group["ranks"], padding=True)     # ttadam.py:113-115
                    state["exp_avg_sq"] = TensorTrain.from_matrix(v, ranks=group["ranks"], padding=True)
                else:
                    state["exp_avg"], state["exp_avg_sq"] = m, v
        return loss


class TTRAdam(torch.optim.Optimizer):
    """Empty stub in the reference as well (ttadam.py:120-121)."""
    pass


class TTSGD(torch.optim.Optimizer):
    """ttsgd.py:8-86, including its quirk: the momentum buffer stored at the first step is never
    updated afterwards (the new value is bound to a local name only, ttsgd.py:68-69)."""

    def __init__(self, params: Iterable[nn.parameter.Parameter], lr: float = 1e-3, momentum: float = 0.9, dampening: float = 0,
                 weight_decay: float = 0, nesterov: bool = False):
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure: Callable = None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                grad_shape = grad.shape
                if grad.is_sparse:
                    raise RuntimeError("TTSGD does not support sparse gradients, please consider SparseAdam instead")
                state = self.state[p]
                if "step" not in state:
                    state["step"] = 0
                tt_mode = "ranks" in group
                d_p = TensorTrain.from_matrix(grad, ranks=group["ranks"], padding=True) if tt_mode else grad
                if group["weight_decay"] != 0:
                    d_p = d_p.add(p, alpha=group["weight_decay"])  # TensorTrain has no .add: raises like ttsgd.py:62
                if group["momentum"] != 0:
                    if "momentum_buffer" not in state:
                        buf = state["momentum_buffer"] = d_p.clone().detach()
                    else:
                        buf = state["momentum_buffer"]
                        buf = group["momentum"] * buf + (1 - group["dampening"]) * d_p
                    d_p = d_p + group["momentum"] * buf if group["nesterov"] else buf
                if tt_mode:
                    d_p = d_p.to_matrix(grad_shape)
                d_p = d_p.to(p.dtype).contiguous()
                if p.is_contiguous():
                    ops.axpby_(d_p, p.data, -group["lr"], 1.0)   # p += -lr * d_p   (ttsgd.py:78)
                else:
                    tmp = p.data.contiguous()
                    ops.axpby_(d_p, tmp, -group["lr"], 1.0)
                    p.data.copy_(tmp)
                if group["weight_decay"] > 0.0:
                    ops.axpby_(p.data.clone(), p.data, -group["lr"] * group["weight_decay"], 1.0)
        return loss


class FactorAdamW:
    """Fused AdamW for the SoW factor parameter group (simple_train.py:502-506): all factors and
    their gradients live in ONE flat buffer each, so the step is one kernel launch and the DP
    all-reduce one collective (see sow_amd/dp.py).  torch.optim.AdamW semantics."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, state_dtype=None):
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        sd = state_dtype or bucket.flat_param.dtype
        self.exp_avg = torch.zeros_like(bucket.flat_param, dtype=sd)
        self.exp_avg_sq = torch.zeros_like(bucket.flat_param, dtype=sd)
        self.step_count = 0

    def _sync_from_group(self):
        """Hyper-parameters written through param_groups[0] (what LR schedulers and drivers do: `group["lr"] = v`,
        simple_train.py:537-563) are the ones the next step() uses and state_dict() saves."""
        g = self.__dict__.get("_group")
        if g is not None:
            self.lr, self.betas, self.eps, self.weight_decay = g["lr"], g["betas"], g["eps"], g["weight_decay"]

    def step(self, grad_scale: float = 1.0):
        self._sync_from_group()
        self.bucket.finalize()   # pending deferred weight-gradient reductions (FactorBucket.attach)
        for p, o in zip(self.bucket.params, self.bucket.offsets):
            if p.data_ptr() != self.bucket.flat_param.data_ptr() + o * self.bucket.flat_param.element_size():
                raise RuntimeError("FactorAdamW.step: a factor no longer lives in the flat buffer (SoWLinear.accumulate() "
                                   "rebinds .data) -- call bucket.rebind() after accumulate(); sow_amd.accumulate(model) does")
        self.step_count += 1
        ops.adamw_flat_(self.bucket.flat_param, self.bucket.flat_grad, self.exp_avg, self.exp_avg_sq, lr=self.lr,
                        betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, step=self.step_count,
                        grad_scale=grad_scale)

    def reset_state(self):
        """reset_optimizer for the factor group (training_utils.py:257-277) as one launch."""
        ops.zero_([self.exp_avg, self.exp_avg_sq])
        self.step_count = 0

    # ---- checkpoint / scheduler surface (simple_train.py:182, 537-563 save and restore optimizer + scheduler state)
    @property
    def param_groups(self):
        """One group, torch.optim style.  A driver's own scheduler code (`for g in opt.param_groups: g["lr"] = lr`) works on
        it: step() and state_dict() read the group's values.  torch.optim.lr_scheduler classes insist on a real
        torch.optim.Optimizer instance and do not accept this object -- drive the factor lr from the loop instead (the
        reference computes its schedule with a LambdaLR over the torch optimizer; read `scheduler.get_last_lr()` and write
        it here)."""
        if not hasattr(self, "_group"):
            self._group = {"params": self.bucket.params, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                           "weight_decay": self.weight_decay}
        else:
            self._sync_from_group()
        return [self._group]

    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero_grad()

    def state_dict(self):
        self._sync_from_group()
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                "numel": self.bucket.padded_numel}

    def load_state_dict(self, sd):
        if int(sd["numel"]) != self.bucket.padded_numel:
            raise ValueError("FactorAdamW.load_state_dict: the checkpoint belongs to a different factor layout")
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr, self.betas, self.eps, self.weight_decay = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])
        if hasattr(self, "_group"):
            self._group.update(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)
