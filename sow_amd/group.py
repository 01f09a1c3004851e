"""Grouped execution of sibling SoWLinear layers -- the module-level face of sow_forward_group / sow_backward_group.

An HF decoder block calls `q_proj(h)`, `k_proj(h)`, `v_proj(h)` (and `gate_proj(h)`, `up_proj(h)`) one after the other
on the SAME hidden state (the reference swaps each of them for a SoWLinear, prepare.py:98-168, and the model code calls
sow.py:107-126 three times).  The layers are independent, so on MI355X they share one grid per kernel (DESIGN.md
section 4: a launch costs ~8 us of ramp and write drain whatever its size).  `group_siblings(model)` arranges that without
touching the model code: the first sibling that sees a new input computes the whole group through ONE autograd node and
parks the other outputs; the other siblings, called with the same tensor, pick theirs up.  Every layer runs its own
workgroups unchanged, so outputs and weight gradients for given inputs are bit-identical to the ungrouped calls; only the
SUM of the siblings' input gradients may round differently from autograd's own accumulation.  A sibling called with a
different input, or a group that the batched path does not cover, simply runs on its own.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib, ops
from .layer import SoWLinear

DEFAULT_GROUPS = (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"),        # Llama (simple_train.py / finetune.py targets)
                  ("query", "key", "value"))                                       # RoBERTa self-attention (run_glue.py:572)


class _SoWGroupFunction(torch.autograd.Function):
    """y_i = SoWLinear_i(x) for n layers on one input.  Tensor arguments per layer: A, B, acc_down, acc_up, bias."""

    @staticmethod
    def forward(ctx, x, scales, sinks, *tensors):
        n = len(scales)
        ctx.sinks = sinks
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        per = [tensors[5 * i:5 * i + 5] for i in range(n)]
        need_bwd = any(ctx.needs_input_grad)
        calls = []
        for (A, B, acc_down, acc_up, bias), s in zip(per, scales):
            kind = ops.acc_kind(acc_down, acc_up)
            calls.append(ops.LayerCall(x2, A.contiguous(), B.contiguous(),
                                       acc_down=acc_down.contiguous() if kind != _lib.ACC_NONE else None,
                                       acc_up=acc_up.contiguous() if kind == _lib.ACC_LOWRANK else None,
                                       bias=bias, scale=s, forward_only=True, save_h=need_bwd))
        ops.LayerGroup(calls).forward()
        if need_bwd:
            ctx.save_for_backward(x2, *[c.h for c in calls], *tensors)
        ctx.scales, ctx.n, ctx.x_shape = scales, n, x.shape
        return tuple(c.y.reshape(*lead, c.y.shape[1]) for c in calls)

    @staticmethod
    def backward(ctx, *dys):
        n = ctx.n
        saved = ctx.saved_tensors
        x2, hs, tensors = saved[0], saved[1:1 + n], saved[1 + n:]
        sinks = ctx.sinks
        if sinks is not None and x2.shape[0] > 0 and all(
                s.usable(tensors[5 * i], tensors[5 * i + 1]) and tensors[5 * i + 4] is None for i, s in enumerate(sinks)):
            # FactorBucket.attach(): ONE data-gradient launch for the siblings, weight gradients queued with their decoder
            # block (dp._GradSink); autograd gets None for the factors
            calls, recs = [], []
            for i, sink in enumerate(sinks):
                A, B, acc_down, acc_up, _ = tensors[5 * i:5 * i + 5]
                kind, r_acc = sink.prepare(x2, B, acc_down, acc_up)
                T, d_out = x2.shape[0], B.shape[1]
                dy = dys[i]
                dy2 = (torch.zeros(T, d_out, dtype=x2.dtype, device=x2.device) if dy is None else dy.reshape(-1, d_out).contiguous())
                calls.append(ops.LayerCall(x2, A, B, acc_down=acc_down if kind != _lib.ACC_NONE else None,
                                           acc_up=acc_up if kind == _lib.ACC_LOWRANK else None, scale=ctx.scales[i], h=hs[i],
                                           dy2=dy2, dx=torch.empty_like(x2), out=(sink.pA.grad, sink.pB.grad, None),
                                           grad_beta=1.0, y=dy2, workspace=sink.ws))
                recs.append((sink, dy2, A, B, acc_down, acc_up, kind, r_acc))
            ops.LayerGroup(calls).backward(_lib.BWD_DATA)
            for i, (sink, dy2, A, B, acc_down, acc_up, kind, r_acc) in enumerate(recs):
                sink.queue(dy2, x2, hs[i], A, B, acc_down, acc_up, ctx.scales[i], kind, r_acc)
            dx = calls[-1].dx
            for c in reversed(calls[:-1]):
                dx = dx + c.dx
            return (dx.reshape(ctx.x_shape), None, None, *([None] * (5 * n)))
        calls, outs = [], []
        for i in range(n):
            A, B, acc_down, acc_up, bias = tensors[5 * i:5 * i + 5]
            kind = ops.acc_kind(acc_down, acc_up)
            T, d_out = x2.shape[0], B.shape[1]
            dy = dys[i]
            dy2 = (torch.zeros(T, d_out, dtype=x2.dtype, device=x2.device) if dy is None
                   else dy.reshape(-1, d_out).contiguous())
            out = (torch.empty_like(A), torch.empty_like(B), torch.empty_like(bias) if bias is not None else None)
            outs.append(out)
            calls.append(ops.LayerCall(x2, A.contiguous(), B.contiguous(),
                                       acc_down=acc_down.contiguous() if kind != _lib.ACC_NONE else None,
                                       acc_up=acc_up.contiguous() if kind == _lib.ACC_LOWRANK else None,
                                       bias=bias, scale=ctx.scales[i], h=hs[i], dy2=dy2, dx=torch.empty_like(x2), out=out,
                                       grad_beta=0.0, y=dy2))   # y is not written by backward: any valid buffer
        ops.LayerGroup(calls).backward(_lib.BWD_DATA | _lib.BWD_WEIGHTS)
        # the siblings share x: its gradient is the sum of theirs.  Summed last sibling first -- the order in which autograd
        # accumulates the contributions of separately called layers (nodes run in reverse creation order), so the rounding
        # matches the ungrouped model as closely as it can
        dx = calls[-1].dx
        for c in reversed(calls[:-1]):
            dx = dx + c.dx
        grads: List[Optional[torch.Tensor]] = []
        for (dA, dB, dbias) in outs:
            grads += [dA, dB, None, None, dbias]
        return (dx.reshape(ctx.x_shape), None, None, *grads)


class SiblingGroup:
    """SoWLinear layers of one parent module that the model calls with the same input tensor."""

    def __init__(self, layers: Sequence[SoWLinear]):
        self.layers = list(layers)
        self._key = None
        self._x = None                 # keeps the input alive while outputs are parked (no id() reuse)
        self._parked: dict = {}

    def usable(self, x: torch.Tensor) -> bool:
        if not x.is_cuda or x.dtype not in ops._DT:
            return False
        sinks = [getattr(m, "_grad_sink", None) for m in self.layers]
        if any(s is not None for s in sinks) and not all(s is not None for s in sinks):
            return False           # some siblings attached to a FactorBucket, some not: every layer runs on its own
        for m in self.layers:
            if m.n_iter != 1 or m.downscale_weights[0].dtype != x.dtype:
                return False
            # an accumulator the grouped launch cannot take as it stands (dtype / shape / device of a checkpoint that was
            # loaded in another precision, load_sow): the layer runs on its own and raises exactly what the ungrouped
            # call raises
            try:
                ops.check_accumulator(x, m.in_features, m.out_features, m.acc_downweight, m.acc_upweight)
            except (TypeError, ValueError, RuntimeError):
                return False
        return True

    def forward(self, layer: SoWLinear, x: torch.Tensor) -> Optional[torch.Tensor]:
        key = (id(x), x._version, x.data_ptr(), tuple(x.shape), torch.is_grad_enabled())
        if self._key == key and id(layer) in self._parked:
            y = self._parked.pop(id(layer))
            if not self._parked:
                self._key = self._x = None
            return y
        if not self.usable(x):
            return None
        tensors = []
        for m in self.layers:
            tensors += [m.downscale_weights._parameters["0"], m.upscale_weights._parameters["0"], m.acc_downweight,
                        m.acc_upweight, m.bias]
        sinks = tuple(getattr(m, "_grad_sink", None) for m in self.layers)
        ys = _SoWGroupFunction.apply(x, tuple(float(m.scale) for m in self.layers), sinks if sinks[0] is not None else None,
                                     *tensors)
        self._key, self._x = key, x
        self._parked = {id(m): y for m, y in zip(self.layers, ys) if m is not layer}
        return ys[self.layers.index(layer)]


def group_siblings(model: nn.Module, groups: Iterable[Sequence[str]] = DEFAULT_GROUPS) -> int:
    """Group sibling SoWLinear layers (same parent, same in_features, one factor pair) that the model calls on the same
    input.  Returns the number of groups installed.  `ungroup_siblings(model)` removes them."""
    n = 0
    for parent in model.modules():
        for names in groups:
            mods = [getattr(parent, nm, None) for nm in names]
            if (all(isinstance(m, SoWLinear) for m in mods) and len({m.in_features for m in mods}) == 1
                    and all(m.n_iter == 1 for m in mods)):
                g = SiblingGroup(mods)
                for m in mods:
                    m._sibling_group = g
                n += 1
    return n


def ungroup_siblings(model: nn.Module) -> None:
    for m in model.modules():
        if isinstance(m, SoWLinear) and hasattr(m, "_sibling_group"):
            del m._sibling_group
