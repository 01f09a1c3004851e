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
        """tensor_linear.py:54-84: pad to i^order, contract with the cores, keep out_features columns."""
        shape = input.shape
        pad = self.in_core_features ** self.order - self.in_features
        x = torch.nn.functional.pad(input, (0, pad), "constant", 0).reshape(-1, self.in_core_features ** self.order)
        w = self.tt.reconstruct().reshape(self.in_core_features ** self.order, -1)   # (i1..in) x (o1..on)
        y = ops.matmul(x.contiguous(), w.contiguous())
        y = y[:, : self.out_features].reshape(*shape[:-1], self.out_features)
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
