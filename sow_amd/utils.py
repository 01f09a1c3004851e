"""Numerics helpers -- host-side mirror of tn_gradient/utils.py (hot-path subset)."""
from __future__ import annotations

from math import ceil

import torch

from . import ops


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
