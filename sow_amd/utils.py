"""Numerics helpers -- host-side mirror of tn_gradient/utils.py (hot-path subset)."""
from __future__ import annotations

from math import ceil

import torch

from . import ops
from .summary import __colorized_str__, module_summary  # noqa: F401  (simple_train.py:45 imports it from utils)


def qr_weight(weight: torch.Tensor, rank: int = None):
    """utils.py:8-30: reduced Householder QR with fp32 internals, truncated to `rank`, cast back."""
    m, n = weight.shape
    k = min(m, n)
    if rank:
        k_keep = min(rank, k)
    else:
        k_keep = k
    q, r = ops.qr_thin(weight, k_keep, need_r=True, out_dtype=weight.dtype)
    return q, r


def pad_matrix(matrix, new_shape):
    """utils.py:78-84 (layout only): zero-pad, top-left aligned; result is float32 like the reference's
    torch.zeros default."""
    out = torch.zeros(tuple(new_shape), device=matrix.device)
    out[: matrix.shape[0], : matrix.shape[1]] = matrix
    return out


def unpad_matrix(matrix, shape):
    """utils.py:86-87."""
    return matrix[: shape[0], : shape[1]]


def closest_factorization(n, d):
    """utils.py:89-99, stale product included (e.g. (1376, 3) -> ([12, 11, 11], 1320))."""
    factors = []
    p, o = 1, n
    while n > 1:
        k = ceil(n ** (1 / d))
        factors.append(k)
        n, p, d = n // k, p * k, d - 1
        if n == 1:
            if p < o:
                factors[-1] += n
            return factors, p


def svd_weight(weight: torch.Tensor, rank: int = None):
    """utils.py:32-57.  Analysis helper only (the reference uses it in export_alignment, outside the hot
    path): forwards to torch.linalg.svd with the same fp32 up-cast / truncation / cast-back."""
    src = weight.dtype
    w = weight if src == torch.float32 else weight.to(torch.float32)
    u, s, v = torch.linalg.svd(w)
    if rank:
        u, s, v = u[:, :rank], s[:rank], v[:rank, :]
    if src != torch.float32:
        u, s, v = u.type(src), s.type(src), v.type(src)
    return u, s, v


# ---- analysis helpers of utils.py:59-141 (off the hot path; kept so `from tn_gradient.utils import ...` resolves) ----
def randhaar(n):
    """utils.py:59-62: Haar-distributed orthogonal [n, n] matrix (scipy's ortho_group), float32."""
    from scipy.stats import ortho_group
    return torch.from_numpy(ortho_group.rvs(dim=n)).to(torch.float32)


def randuptri(n, scale=1.0):
    """utils.py:64-70: upper-triangular Gaussian with chi-distributed diagonal (the R of a Gaussian matrix's QR)."""
    r = torch.randn(n, n).triu_()
    dof = torch.arange(n, 0, -1, dtype=torch.float32)
    r.diagonal().copy_(torch.distributions.Chi2(df=dof).sample().sqrt() * scale)
    return r


def perturbe_random(matrix: torch.Tensor, scale=0.02):
    """utils.py:72-76."""
    return matrix + scale * torch.randn(matrix.size(), device=matrix.device)


def generate_rank_k(shape, rank, mix=1, pos=False):
    """utils.py:101-112: sum of `mix` random CP tensors of the given rank (uniform factors, centred unless pos)."""
    letters = "abcdefghij"[: len(shape)]
    eq = ",".join(c + "z" for c in letters) + "->" + letters
    total = torch.zeros(shape)
    for _ in range(mix):
        factors = [torch.rand(dim, rank) for dim in shape]
        if not pos:
            factors = [2 * f - 1 for f in factors]
        total += torch.einsum(eq, *factors)
    return total


def unfolding(tensor, mode):
    """utils.py:114-133: mode-`mode` unfolding [a_mode, rest]."""
    d = tensor.dim()
    if not -d <= mode < d:
        raise ValueError("Mode must be between 1 - d and d + 1, d being the number of dimensions of the tensor")
    return torch.movedim(tensor, mode % d, 0).reshape(tensor.shape[mode % d], -1)


def left_unfolding(tensor):
    """utils.py:135-137: [a_1 * ... * a_{d-1}, a_d]."""
    return unfolding(tensor, -1).t()


def right_unfolding(tensor):
    """utils.py:139-141: [a_1, a_2 * ... * a_d]."""
    return unfolding(tensor, 0)
