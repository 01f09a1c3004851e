"""SoWLinear / SoWParameter -- host-side mirror of tn_gradient/layer/sow.py (reference lines cited per
method) whose arithmetic runs in libsow_amd.so.

Same constructor signature, attribute names and state-dict keys as the reference
(`acc_upweight`, `acc_downweight`, `downscale_weights.{i}`, `upscale_weights.{i}`, `bias`), so the
training drivers (`simple_train.py`, `finetune.py`) need no change.  Parameters stay ordinary
nn.Parameters (AdamW, DDP, save_pretrained, gradient checkpointing keep working).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import ops


class SoWParameter(nn.ParameterList):
    """n_iter factor matrices [in_features, out_features] (reference sow.py:15-42)."""

    def __init__(self, in_features: int, out_features: int, n_iter: int = 1, device=None, dtype=None) -> None:
        kw = {"device": device, "dtype": dtype}
        super().__init__([nn.Parameter(torch.empty(in_features, out_features, **kw)) for _ in range(n_iter)])
        self.in_features = in_features
        self.out_features = out_features
        self.n_iter = n_iter

    def from_weights(self, weights: List[torch.Tensor]) -> None:
        # rebinding .data keeps the Parameter objects (optimizer / DDP references survive), sow.py:37-39
        for i, w in enumerate(weights):
            self[i].data = w.data

    def extra_repr(self) -> str:
        return f"{self.n_iter} x ({self.in_features}, {self.out_features})"


class _SoWFunction(torch.autograd.Function):
    """y = acc_term + scale * (x @ A) @ B + bias through sow_forward / sow_backward (include/sow_amd.h)."""

    @staticmethod
    def forward(ctx, x, A, B, acc_down, acc_up, bias, scale, sink=None):
        lead = x.shape[:-1]
        # the kernels take dense row-major buffers: a strided view (x = big[..., :d], seq[:, 0, :], W.t()) is packed ONCE
        # here so that forward and backward read the same rows (backward passes raw pointers of the saved tensors)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        A, B = A.contiguous(), B.contiguous()
        if acc_down is not None and acc_down.numel():
            acc_down = acc_down.contiguous()
        if acc_up is not None and acc_up.numel():
            acc_up = acc_up.contiguous()
        y, h = ops.sow_forward(x2, A, B, acc_down, acc_up, bias, scale)
        ctx.save_for_backward(x2, h, A, B, acc_down, acc_up)
        ctx.scale = scale
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        ctx.sink = sink
        return y.reshape(*lead, B.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, h, A, B, acc_down, acc_up = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        sink = ctx.sink
        if sink is not None and not ctx.has_bias and x2.shape[0] > 0 and sink.usable(A, B):
            # FactorBucket.attach(): the weight gradients accumulate straight into the flat gradient buffer (p.grad is a
            # view of it), so autograd gets None for A and B; their slab-partial sums are reduced for all layers in one
            # launch by FactorBucket.finalize() (sow_reduce_batch).
            dx = sink.backward(dy2, x2, h, A, B, acc_down, acc_up, ctx.scale)
            return dx.reshape(ctx.x_shape), None, None, None, None, None, None, None
        dx, dA, dB, dbias = ops.sow_backward(dy2, x2, h, A, B, acc_down, acc_up, ctx.scale, ctx.has_bias)
        return dx.reshape(ctx.x_shape), dA, dB, None, None, dbias, None, None


class SoWLinear(nn.Module):
    """Sum-of-Weights low-rank linear layer (reference sow.py:45-181)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, rank: int = 16, n_iter: int = 1,
                 scale: float = 1, init_method: str = "normal_QR", device=None, dtype=None, init_params=True) -> None:
        kw = {"device": device, "dtype": dtype}
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.n_iter = n_iter
        self.rank = rank
        self.scale = scale
        self.virtual_rank = min(rank * n_iter, in_features, out_features)  # sow.py:67
        # frozen accumulator, zero-numel until the first accumulate() / prepare_sow (sow.py:69-70)
        self.acc_upweight = nn.Parameter(torch.empty(0), requires_grad=False)
        self.acc_downweight = nn.Parameter(torch.empty(0), requires_grad=False)
        self.init_method = init_method
        self.downscale_weights = SoWParameter(in_features, rank, n_iter=n_iter, device=device, dtype=dtype)
        self.upscale_weights = SoWParameter(rank, out_features, n_iter=n_iter, device=device, dtype=dtype)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, **kw))
        else:
            self.register_parameter("bias", None)
        if init_params:
            self.reset_parameters()

    # ------------------------------------------------------------------------------------------
    def _fresh_gaussian(self, shape, device, dtype):
        w = torch.zeros(shape, device=device, dtype=dtype)
        return nn.init.normal_(w, mean=0.0, std=0.02)  # std hard-coded as in sow.py:96 / :169

    _fresh_gaussian_default = _fresh_gaussian   # lets the batched accumulate see whether a test has replaced the draw

    def reset_parameters(self, reset_scale=1.0) -> None:
        """sow.py:89-105.  normal_QR: A = Q[:, :r], B = R[:r, :] of the QR of an fp32 N(0, 0.02^2)
        [in, out] draw (the reference draws it on "cuda", :91 -- here on the factors' GPU)."""
        for i in range(self.n_iter):
            if i / self.n_iter >= 1 - reset_scale:
                a, b = self.downscale_weights[i], self.upscale_weights[i]
                if self.init_method == "normal_QR":
                    if not a.is_cuda:
                        raise RuntimeError("SoWLinear(init_method='normal_QR') initialises on the GPU "
                                           "(reference sow.py:91 hard-codes 'cuda'); construct with device='cuda'")
                    w = self._fresh_gaussian((self.in_features, self.out_features), a.device, torch.float32)
                    q, r = ops.qr_thin(w, self.rank, need_r=True, out_dtype=torch.float32)
                    self.downscale_weights[i] = q.to(a.dtype).contiguous()
                    self.upscale_weights[i] = r.to(b.dtype).contiguous()
                else:
                    nn.init.normal_(a, std=0.02)
                    nn.init.normal_(b, std=0.02)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    # ------------------------------------------------------------------------------------------
    def _cat_factors(self):
        if self.n_iter == 1:
            # direct dict access: ParameterList.__getitem__ costs ~5 us per lookup on the per-call path
            return self.downscale_weights._parameters["0"], self.upscale_weights._parameters["0"]
        # sum_i A_i B_i = [A_1 .. A_n] [B_1; ..; B_n]; autograd splits the gradients back
        return torch.cat(list(self.downscale_weights), dim=1), torch.cat(list(self.upscale_weights), dim=0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """sow.py:107-126 in one fused call: accumulator term (not scaled) + scale * (x A) B + bias."""
        group = self.__dict__.get("_sibling_group")     # sow_amd.group.group_siblings: q/k/v or gate/up in one launch
        if group is not None:
            y = group.forward(self, x)
            if y is not None:
                return y
        A, B = self._cat_factors()
        if not (torch.is_grad_enabled() and (x.requires_grad or A.requires_grad or B.requires_grad
                                             or (self.bias is not None and self.bias.requires_grad))):
            # no backward will follow (eval / generate, commonsense_evaluate.py:268-287; the first pass of activation
            # checkpointing): the projection h = scale * x A is not written to HBM
            x2 = x.reshape(-1, x.shape[-1])
            y, _ = ops.sow_forward(x2, A, B, self.acc_downweight, self.acc_upweight, self.bias, float(self.scale), save_h=False)
            return y.reshape(*x.shape[:-1], B.shape[1])
        return _SoWFunction.apply(x, A, B, self.acc_downweight, self.acc_upweight, self.bias, float(self.scale),
                                  getattr(self, "_grad_sink", None))

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def accumulate(self):
        """sow.py:128-178: fold scale * sum_i A_i B_i into the accumulator, optionally re-factor it by
        truncated QR, re-initialise A, zero B."""
        A = torch.cat([w.data for w in self.downscale_weights], dim=1) if self.n_iter > 1 else self.downscale_weights[0].data
        B = torch.cat([w.data for w in self.upscale_weights], dim=0) if self.n_iter > 1 else self.upscale_weights[0].data
        if not A.is_cuda:
            raise RuntimeError("SoWLinear.accumulate runs on the GPU; there is no CPU fallback")
        has_down = self.acc_downweight.numel() != 0
        has_up = self.acc_upweight.numel() != 0
        scale = float(self.scale)
        if has_down and has_up:      # W_acc = Q R, then += scale * A B   (sow.py:137-138)
            acc = ops.gemm(self.acc_downweight.data.to(A.dtype), self.acc_upweight.data.to(A.dtype))
            ops.gemm(A, B, out=acc, alpha=scale, beta=1.0)
        elif has_down:               # dense accumulator updated in place (sow.py:139-140)
            acc = self.acc_downweight.data
            if acc.dtype != A.dtype:
                acc = acc.to(A.dtype)
            ops.gemm(A, B, out=acc, alpha=scale, beta=1.0)
        else:
            acc = ops.gemm(A, B, alpha=scale)

        if self.virtual_rank < min(self.in_features, self.out_features):   # sow.py:144-150
            q, r = ops.qr_thin(acc, self.virtual_rank, need_r=True)
            self.acc_downweight = nn.Parameter(q, requires_grad=False)
            self.acc_upweight = nn.Parameter(r, requires_grad=False)
            self.virtual_rank = min(self.virtual_rank + self.rank * self.n_iter, self.in_features, self.out_features)
        else:                                                              # sow.py:151-153
            self.acc_downweight = nn.Parameter(acc, requires_grad=False)
            # zero-numel placeholder as in the reference (default dtype), but on the layer's device so that a state dict
            # taken after accumulate() does not mix devices
            self.acc_upweight = nn.Parameter(torch.empty(0, device=acc.device), requires_grad=False)

        # new factors: B <- 0 (sow.py:159), A <- Q[:, :r] of a fresh Gaussian (sow.py:161-172) or a Gaussian (:174)
        new_down = [torch.zeros_like(w) for w in self.downscale_weights]
        new_up = [torch.empty_like(w) for w in self.upscale_weights]
        ops.zero_(new_up)
        for i in range(self.n_iter):
            if self.init_method == "normal_QR":
                w = self._fresh_gaussian((self.in_features, self.out_features), self.acc_downweight.device,
                                         self.acc_downweight.dtype)
                q, _ = ops.qr_thin(w, self.rank, need_r=False)
                new_down[i] = q.contiguous()
            else:   # plain Gaussian re-init (sow.py:174), through the same draw hook as the QR branch
                new_down[i] = self._fresh_gaussian(tuple(new_down[i].shape), new_down[i].device, new_down[i].dtype)
        self.downscale_weights.from_weights(new_down)
        self.upscale_weights.from_weights(new_up)

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"rank={self.rank}, n_iter={self.n_iter}")
