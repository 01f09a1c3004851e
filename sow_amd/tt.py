"""TensorTrain -- host-side mirror of tn_gradient/tt.py (TT-matrix container + algebra).

Cores are [r_k, i_k, o_k, r_{k+1}] fp32 device tensors.  Arithmetic (QR sweeps, bond contractions,
Hadamard core products, scalings) runs in libsow_amd.so through sow_amd.ops; torch is used for
layout only (reshape / permute / cat / pad / slicing).  Quirks of the reference that change results
are kept and cited.
"""
from __future__ import annotations

import math
from math import ceil, floor, log

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .utils import closest_factorization, pad_matrix, unpad_matrix  # noqa: F401  (re-exported like the reference)


def _prod(xs):
    return int(math.prod(int(x) for x in xs))


class TensorTrain:
    def __init__(self, ranks, input_shape, output_shape, device=None) -> None:
        self.order = len(ranks) - 1
        self.ranks = ranks
        self.input_shape = input_shape
        self.output_shape = output_shape
        self.cores = [None for _ in range(self.order)]
        self.device = device
        self.contract_expr = None  # kept for API compatibility (the reference caches an opt_einsum expression)

    # ------------------------------------------------------------------ constructors (tt.py:26-84)
    @staticmethod
    def from_tensor(tensor: torch.Tensor, ranks: list):
        """Axes (*input_shape, *output_shape); permuted to interleaved (i1,o1,i2,o2,...) (tt.py:33)."""
        half = len(tensor.shape) // 2
        tt = TensorTrain(ranks, tensor.shape[:half], tensor.shape[half:])
        perm = [ax for pair in zip(range(tt.order), range(tt.order, 2 * tt.order)) for ax in pair]
        tt.decompose(tensor.permute(*perm))
        return tt

    @staticmethod
    def from_cores(cores):
        tt = TensorTrain([c.shape[0] for c in cores] + [1], [c.shape[1] for c in cores], [c.shape[2] for c in cores],
                         device=cores[0].device)
        tt.cores = cores
        return tt

    @staticmethod
    def from_matrix(matrix: torch.Tensor, ranks: list, padding=True):
        """tt.py:48-67.  mm = ceil(M ** (1/order)) in double precision (bit-exact with the reference,
        including 3125 ** (1/5) -> 6)."""
        order = len(ranks) - 1
        M, N = matrix.shape
        mm = ceil(M ** (1 / order))
        nn_ = ceil(N ** (1 / order))
        if padding:
            matrix = pad_matrix(matrix, (mm ** order, nn_ ** order))
        tensor = matrix.reshape((mm,) * order + (nn_,) * order)
        return TensorTrain.from_tensor(tensor, ranks).to(matrix.device)

    # ------------------------------------------------------------------ constant trains and container plumbing
    # (behaviour of reference tt.py:68-109: same names and results -- `clone` shares the core tensors, `to` moves them in
    # place through `.data`, `type` rebinds them -- expressed through two private helpers)
    def _core_shapes(self):
        return [(self.ranks[k], self.input_shape[k], self.output_shape[k], self.ranks[k + 1]) for k in range(self.order)]

    def _sibling(self, cores):
        """A new train with this one's ranks / shapes around `cores`."""
        other = TensorTrain(list(self.ranks), self.input_shape, self.output_shape)
        other.cores = cores
        return other

    @classmethod
    def _constant(cls, fill, ranks, input_shape, output_shape, device):
        tt = cls(ranks, input_shape, output_shape)
        tt.cores = [fill(shape) for shape in tt._core_shapes()]
        return tt.to(device)

    @staticmethod
    def zeros(ranks, input_shape, output_shape, device="cpu"):
        return TensorTrain._constant(torch.zeros, ranks, input_shape, output_shape, device)

    @staticmethod
    def ones(ranks, input_shape, output_shape, device="cpu"):
        return TensorTrain._constant(torch.ones, ranks, input_shape, output_shape, device)

    def numel(self):
        return sum(int(c.numel()) for c in self.cores)

    def size(self):
        return [c.size() for c in self.cores]

    def to(self, device):
        self.device = device
        if device is not None and self.cores[0].device != torch.device(device):
            for c in self.cores:
                c.data = c.data.to(device)           # in place: views and optimizer references stay valid
        return self

    def clone(self):
        return self._sibling(list(self.cores))       # shallow on purpose (reference tt.py:96-99)

    def detach(self):
        return self._sibling([c.detach() for c in self.cores])

    def type(self, dtype):
        self.cores = [c.type(dtype) for c in self.cores]
        return self

    def requires_grad_(self, flag):
        for c in self.cores:
            c.requires_grad_(flag)
        return self

    # ------------------------------------------------------------------ decomposition (tt.py:111-140)
    def decompose(self, tensor: torch.Tensor):
        """Sequential truncated QR: L = reshape(r_k*i_k*o_k, -1); Q, R = qr(L, 'complete');
        core_k = Q[:, :r_{k+1}], remainder = R[:r_{k+1}, :]."""
        rest = tensor
        for k in range(self.order - 1):
            rows = self.ranks[k] * self.input_shape[k] * self.output_shape[k]
            left = rest.reshape(rows, -1).float().contiguous()
            q, r = ops.qr_thin(left, self.ranks[k + 1], need_r=True)
            self.cores[k] = q.reshape(self.ranks[k], self.input_shape[k], self.output_shape[k], self.ranks[k + 1])
            rest = r
        self.cores[-1] = rest.reshape(self.ranks[-2], self.input_shape[-1], self.output_shape[-1], self.ranks[-1])
        return self

    def left_matrix(self, index):
        return self.cores[index].reshape(self.ranks[index] * self.input_shape[index] * self.output_shape[index], -1)

    def right_matrix(self, index):
        return self.cores[index].reshape(-1, self.input_shape[index] * self.output_shape[index] * self.ranks[index + 1])

    def to_core(self, matrix, index):
        return matrix.reshape(self.ranks[index], self.input_shape[index], self.output_shape[index], self.ranks[index + 1])

    def orthogonalize(self, mode="left", new_ranks=None, inplace=False):
        """tt.py:142-180."""
        if not inplace:
            tt = self.clone()
            tt.orthogonalize(mode, new_ranks, inplace=True)
            return tt
        if mode == "left":
            for k in range(self.order - 1):
                L = self.left_matrix(k).contiguous()
                R = self.right_matrix(k + 1).contiguous()
                q, s = ops.qr_thin(L, min(L.shape), need_r=True)
                w = ops.gemm(s, R)
                if new_ranks:
                    q, w = q[:, : new_ranks[k]], w[: new_ranks[k], :]
                self.ranks[k + 1] = q.shape[1]
                self.cores[k] = self.to_core(q, k)
                self.cores[k + 1] = self.to_core(w, k + 1)
        elif mode == "right":
            for k in range(self.order - 1, 0, -1):
                L = self.left_matrix(k - 1).contiguous()
                R = self.right_matrix(k).contiguous()
                rt = R.t().contiguous()
                q, s = ops.qr_thin(rt, min(rt.shape), need_r=True)
                w = ops.gemm(L, s, trans_b=True)  # L @ S^T
                if new_ranks:
                    q, w = q[:, : new_ranks[k]], w[: new_ranks[k], :]
                    self.ranks[k] = new_ranks[k]
                self.ranks[k] = w.shape[1]
                self.cores[k - 1] = self.to_core(w, k - 1)
                self.cores[k] = self.to_core(q.t(), k)
        return self

    def round(self, new_ranks=None, inplace=False, like=None):
        """tt.py:182-211."""
        if type(new_ranks) == int:
            new_ranks = [1] + [new_ranks] * (self.order - 1) + [1]
        elif not new_ranks and not like:
            new_ranks = [1] + [i * o for i, o in zip(self.input_shape, self.output_shape)] + [1]
        elif like:
            new_ranks = like.ranks
        if not inplace:
            tt = self.clone()
            tt.round(new_ranks, inplace=True)
            return tt
        self.orthogonalize(mode="right", inplace=True)
        for k in range(self.order - 1):
            L = self.left_matrix(k).contiguous()
            R = self.right_matrix(k + 1).contiguous()
            q, s = ops.qr_thin(L, new_ranks[k + 1], need_r=True)  # complete-mode QR truncated to new_ranks[k+1]
            w = ops.gemm(s, R)
            self.ranks[k] = new_ranks[k]
            self.ranks[k + 1] = new_ranks[k + 1]
            self.cores[k] = self.to_core(q, k)
            self.cores[k + 1] = self.to_core(w, k + 1)
        return self

    # ------------------------------------------------------------------ reconstruction (tt.py:213-247)
    def reconstruct(self) -> torch.Tensor:
        """Chain contraction over the bonds, left to right; output axes (i1..in, o1..on).
        The reference lets opt_einsum pick the order, so fp32 results agree to ~1e-6, not bit-exactly."""
        acc = self.cores[0]
        r0 = acc.shape[0]
        acc = acc.reshape(-1, acc.shape[-1])
        dims = [(self.cores[0].shape[1], self.cores[0].shape[2])]
        for c in self.cores[1:]:
            rk, ik, ok, rn = c.shape
            acc = ops.matmul(acc.contiguous(), c.reshape(rk, -1).contiguous()).reshape(-1, rn)
            dims.append((ik, ok))
        rn = acc.shape[-1]
        full = acc.reshape(r0, -1, rn)
        full = full.sum(dim=0).sum(dim=-1) if (r0 != 1 or rn != 1) else full.reshape(-1)
        full = full.reshape([d for pair in dims for d in pair])
        perm = list(range(0, 2 * self.order, 2)) + list(range(1, 2 * self.order, 2))
        return full.permute(*perm)

    def to_tensor(self) -> torch.Tensor:
        return self.reconstruct()

    def to_matrix(self, shape) -> torch.Tensor:
        matrix = self.to_tensor().reshape(_prod(self.input_shape), _prod(self.output_shape))
        return unpad_matrix(matrix, shape)

    # ------------------------------------------------------------------ inner products (tt.py:253-277)
    def norm(self, mode="full"):
        return self.inner(self, mode=mode)

    def inner(self, other, mode="right"):
        if mode == "full":
            env = None
            for ca, cb in zip(self.cores, other.cores):
                ra, i, j, rb = ca.shape
                rc, _, _, rd = cb.shape
                if env is None:
                    env = torch.ones((ra, rc), dtype=torch.float32, device=ca.device)
                x = ops.gemm(env, ca.reshape(ra, -1).contiguous().float(), trans_a=True)      # [rc, (i j rb)]
                x = x.reshape(rc * i * j, rb)
                env = ops.gemm(x, cb.reshape(rc * i * j, rd).contiguous().float(), trans_a=True)  # [rb, rd]
            return float(env.sum()) if env.numel() != 1 else float(env.squeeze())
        elif mode == "right":
            la, lb = self.cores[-1], other.cores[-1]
            out = ops.gemm(la.reshape(-1, la.shape[-1]).contiguous().float(),
                           lb.reshape(-1, lb.shape[-1]).contiguous().float(), trans_a=True)
            return float(out.squeeze())

    # ------------------------------------------------------------------ Newton iterations (tt.py:279-341)
    def _absmax(self, cores):
        return float(max(ops.absmax(c.detach()) for c in cores))

    def sqrtinv(self, threshold=1e-8, max_iter=4):
        max_value = self._absmax(self.cores)
        max_value = _prod(self.ranks) * (max_value ** (self.order // 2))
        k = floor(log(max_value) / log(4))
        c, revc = (1 / (4 ** k)), 2 ** k
        A = c * self.clone()
        max_ranks = [1] + [i * o for i, o in zip(self.input_shape, self.output_shape)] + [1]
        while max_iter > 0:
            B = -1 / 2 * (self * (A * A).round(max_ranks)).add_(-3)
            B = B.round(max_ranks)
            C = A * B
            C = C.round(max_ranks)
            if threshold:
                norm = abs((C - A).norm())
                if norm < threshold:
                    return revc * C
            A = C
            max_iter -= 1
        return revc * A

    def sqrt(self, threshold=1e-3, max_iter=4):
        max_value = ops.absmax(self.cores[-1].detach())
        max_value = _prod(self.ranks) * (max_value ** 1)
        k = floor(log(max_value) / log(4))
        A = (1 / (4 ** k)) * self.clone()
        C = A.clone().add_(-1)
        ranks = list(A.ranks)
        while max_iter > 0 and (A - C).norm() > threshold:
            B = A - 1 / 2 * (A * C)
            B = B.round(ranks)
            D = 1 / 4 * (C * C).round(ranks) * (C.add_(-3))
            D = D.round(ranks)
            max_iter -= 1
            A, C = B, D
        return 2 ** k * A

    # ------------------------------------------------------------------ algebra (tt.py:343-494)
    @staticmethod
    def _block_concat(cores_a, cores_b, ranks_a, ranks_b):
        """Core-wise block concatenation exactly as tt.py:400-418: the right-hand middle core is padded
        on its RIGHT-bond axis by self.ranks[i] (the LEFT bond rank of self), so -- like the reference --
        middle cores need r_i == r_{i+1} on the left operand."""
        order = len(cores_a)
        out = []
        for i in range(order):
            ca, cb = cores_a[i], cores_b[i]
            if i == 0:
                out.append(torch.cat((ca, cb), dim=-1))
            elif i == order - 1:
                out.append(torch.cat((ca, cb), dim=0))
            else:
                ca_p = F.pad(ca, (0, ranks_b[i + 1], 0, 0))
                cb_p = F.pad(cb, (ranks_a[i], 0, 0, 0))
                out.append(torch.cat([ca_p, cb_p], dim=0))
        return out

    def add_(self, constant):
        """tt.py:343-379: block-concatenate a constant train whose entries are
        sign * (|constant| / prod(ranks)) ** (1/order).  Returns a NEW train (like the reference)."""
        sub = constant / _prod(self.ranks)
        neg = sub < 0
        sub = abs(sub) ** (1 / self.order)
        fill = (-1 if neg else 1) * sub
        const_cores = [torch.full_like(c, fill) for c in self.cores]
        return TensorTrain.from_cores(self._block_concat(self.cores, const_cores, self.ranks, self.ranks))

    def __add__(self, other):
        return TensorTrain.from_cores(self._block_concat(self.cores, other.cores, self.ranks, other.ranks))

    def __sub__(self, other):
        return self + (-1) * other

    def __rmul__(self, constant):
        """tt.py:428-447: every core times sign(c) * |c| ** (1/order) (sign on ALL cores -- kept)."""
        neg = constant < 0
        sub = abs(constant) ** (1 / self.order)
        f = (-1 if neg else 1) * sub
        cores = []
        for core in self.cores:
            c = core.detach().contiguous().float()
            cores.append(ops.axpby_(c, torch.empty_like(c), f, 0.0))
        return TensorTrain.from_cores(cores)

    def __mul__(self, other):
        """Hadamard product, tt.py:449-478."""
        return TensorTrain.from_cores([ops.tt_kron_core(a, b) for a, b in zip(self.cores, other.cores)])

    def reciprocal(self):
        """tt.py:480-494: first and last cores copied, every middle core's [:, i, j, :] slice inverted
        (which requires r_k == r_{k+1} there, as in the reference)."""
        cores = []
        for i, core in enumerate(self.cores):
            if i == 0 or i == self.order - 1:
                cores.append(core.detach().clone())
            else:
                r0, a, b, r1 = core.shape
                mats = core.detach().permute(1, 2, 0, 3).reshape(a * b, r0, r1)
                inv = ops.small_inverse(mats)
                cores.append(inv.reshape(a, b, r0, r1).permute(2, 0, 1, 3).contiguous())
        return TensorTrain.from_cores(cores)

    def to_params(self):
        cores = nn.ParameterList()
        for core in self.cores:
            cores.append(nn.Parameter(core))
        self.cores = cores
        return self


# ---------------------------------------------------------------------------------------------------------------------
# batched entry points: many trains per launch through sow_tt_reconstruct_batch / sow_tt_decompose_batch
# ---------------------------------------------------------------------------------------------------------------------
def _tt_desc(cores, ranks, input_shape, output_shape, rows, cols):
    """sow_tt_desc (include/sow_amd.h) of a train whose cores are fp32 contiguous device tensors; None when the batched
    kernels do not cover it (rank > 32, order > 6, a truncation rank larger than its unfolding, CPU cores)."""
    from . import _lib
    order = len(cores)
    if order < 1 or order > _lib.TT_MAX_ORDER or max(ranks) > 32 or ranks[0] != 1 or ranks[-1] != 1:
        return None
    for k, c in enumerate(cores):
        if not (c.is_cuda and c.dtype == torch.float32 and c.is_contiguous()):
            return None
        if k + 1 < order and ranks[k + 1] > ranks[k] * input_shape[k] * output_shape[k]:
            return None
    d = _lib.TtDesc()
    d.order = order
    for k in range(order):
        d.cores[k] = cores[k].data_ptr()
        d.in_dims[k], d.out_dims[k] = int(input_shape[k]), int(output_shape[k])
    for k in range(order + 1):
        d.ranks[k] = int(ranks[k])
    d.rows, d.cols = int(rows), int(cols)
    return d


def _empty_train(ranks, shape, device):
    """Cores (uninitialised) of the train TensorTrain.from_matrix(matrix of `shape`, ranks, padding=True) would return."""
    order = len(ranks) - 1
    mm, nn_ = ceil(shape[0] ** (1 / order)), ceil(shape[1] ** (1 / order))     # tt.py:53-54, double pow + ceil
    tt = TensorTrain(list(ranks), (mm,) * order, (nn_,) * order)
    tt.cores = [torch.empty(s, dtype=torch.float32, device=device) for s in tt._core_shapes()]
    tt.device = device
    return tt


def to_matrix_batch(trains, shapes):
    """[tt.to_matrix(shape) for tt, shape in zip(trains, shapes)] in one launch per 16 trains (reference tt.py:213-247)."""
    import ctypes

    from . import _lib
    descs = [_tt_desc(t.cores, t.ranks, t.input_shape, t.output_shape, s[0], s[1]) for t, s in zip(trains, shapes)]
    if not trains or any(d is None for d in descs):
        return [t.to_matrix(s) for t, s in zip(trains, shapes)]
    dev = trains[0].cores[0].device
    outs = [torch.empty(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    n = len(trains)
    arr = (_lib.TtDesc * n)(*descs)
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    lds = (ctypes.c_int64 * n)(*[o.stride(0) for o in outs])
    ops._launch(dev, "sow_tt_reconstruct_batch", _lib.load().sow_tt_reconstruct_batch, arr, ptrs, lds, n)
    return outs


def from_matrix_batch(matrices, ranks):
    """[TensorTrain.from_matrix(m, ranks, padding=True) for m in matrices] with ONE Householder-panel launch per bond for
    all of them (reference tt.py:48-67, 111-140)."""
    import ctypes

    from . import _lib
    if not matrices:
        return []
    lib = _lib.load()
    dev = matrices[0].device
    mats = [m if (m.dtype == torch.float32 and m.stride(1) == 1) else m.float().contiguous() for m in matrices]
    trains = [_empty_train(ranks, m.shape, dev) if m.is_cuda else None for m in mats]
    descs = [None if t is None else _tt_desc(t.cores, t.ranks, t.input_shape, t.output_shape, m.shape[0], m.shape[1])
             for t, m in zip(trains, mats)]
    if any(d is None for d in descs):
        return [TensorTrain.from_matrix(m, ranks=ranks, padding=True) for m in matrices]
    n = len(mats)
    arr = (_lib.TtDesc * n)(*descs)
    sizes = [int(lib.sow_tt_decompose_workspace_bytes(ctypes.byref(d))) for d in descs]
    ws = [torch.empty(sz, dtype=torch.uint8, device=dev) for sz in sizes]
    src = (ctypes.c_void_p * n)(*[m.data_ptr() for m in mats])
    lds = (ctypes.c_int64 * n)(*[m.stride(0) for m in mats])
    wsp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
    wsb = (ctypes.c_size_t * n)(*sizes)
    ops._launch(dev, "sow_tt_decompose_batch", lib.sow_tt_decompose_batch, arr, src, lds, n, wsp, wsb)
    return trains
