"""TensorTrainLinear -- host-side mirror of tn_gradient/layer/tensor_linear.py:9-84."""
from __future__ import annotations

from math import ceil, sqrt

import torch
import torch.nn as nn

from . import ops
from .tt import TensorTrain


class TensorTrainLinear(nn.Module):
    def __init__(self, in_features, out_features, ranks, bias=True, device=None, type=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.ranks = ranks
        self.order = len(ranks) - 1
        self.contract_expr = None
        self.in_core_features = ceil(in_features ** (1 / self.order))    # tensor_linear.py:20-21 (double pow + ceil)
        self.out_core_features = ceil(out_features ** (1 / self.order))
        self.tt = TensorTrain.zeros(input_shape=[self.in_core_features] * self.order,
                                    output_shape=[self.out_core_features] * self.order, ranks=ranks, device=device)
        self.tt.to_params()
        if type is not None:  # the reference calls core.type(None), which turns the cores into strings
            self.tt.type(type)
        if bias:
            # the reference's reset_parameters dereferences a non-existent self.weight when bias=True
            # (tensor_linear.py:48): only bias=False is usable there; fail the same way, early.
            raise AttributeError("'TensorTrainLinear' object has no attribute 'weight' (bias=True is unusable in the "
                                 "reference, tensor_linear.py:48)")
        self.register_parameter("bias", None)
        self.reset_parameters()

    def to(self, device):
        self.tt.to(device)
        return super().to(device)

    def reset_parameters(self):
        for core in self.tt.cores:
            nn.init.kaiming_uniform_(core, a=sqrt(5))

    def forward(self, input):
        """tensor_linear.py:54-84: pad the last dimension to i^order, contract the input with the cores ONE BOND AT A TIME,
        keep out_features columns.  The dense in x out weight is never formed (that is what the layer exists to avoid):
        before step k the state is [b, (o_1..o_k), r_k, i_{k+1}, (i_{k+2}..i_d)]; the step contracts (r_k, i_{k+1}) against
        core k viewed as [(r_k i_{k+1}), (o_{k+1} r_{k+1})] -- one GEMM on the MFMA kernel (ops.matmul, with autograd to the
        input and to the core) -- and moves o_{k+1} to the outputs.  Work and memory are O(b * max intermediate), e.g.
        100 -> 60 with ranks [1, 4, 4, 1]: 3 GEMMs with K = 5, 20, 20 instead of a 125 x 64 weight.  Summation order differs
        from opt_einsum's path: fp32 agreement ~1e-6."""
        shape = input.shape
        i, o, d = self.in_core_features, self.out_core_features, self.order
        pad = i ** d - self.in_features
        x = torch.nn.functional.pad(input, (0, pad), "constant", 0).reshape(-1, i ** d)
        b = x.shape[0]
        z = x.reshape(b, 1, 1, i, i ** (d - 1))                      # [b, O = 1, r_0 = 1, i_1, rest]
        O = 1
        for k, core in enumerate(self.tt.cores):
            rk, ik, ok, rn = core.shape
            rest = z.shape[4]
            zz = z.permute(0, 1, 4, 2, 3).reshape(b * O * rest, rk * ik).contiguous()
            y = ops.matmul(zz, core.reshape(rk * ik, ok * rn).contiguous())          # [(b O rest), (o_{k+1} r_{k+1})]
            y = y.reshape(b, O, rest, ok, rn).permute(0, 1, 3, 4, 2)                 # [b, O, o_{k+1}, r_{k+1}, rest]
            O *= ok
            nxt = i if k + 1 < d else 1
            z = y.reshape(b, O, rn, nxt, max(rest // nxt, 1))
        y = z.reshape(b, O)[:, : self.out_features].reshape(*shape[:-1], self.out_features)
        if self.bias is not None:
            y = y + self.bias
        return y


class ComposedLinear(nn.Module):
    """Empty stub in the reference as well (tensor_linear.py:86-103)."""

    def __init__(self, in_features, out_features, rank, bias=True, composition=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.rank = rank
        self.bias = bias
