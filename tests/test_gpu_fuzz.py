"""Seeded random sweep of the C-ABI forward / backward against the oracle: ragged token counts, widths that are
and are not multiples of 2 / 4 / 8 / 64, ranks on both sides of 64, every accumulator kind, both dtypes.  The
shapes are drawn so that every dispatch branch of sow_forward / sow_backward (streaming and generic chain kernels,
DMA and generic skinny-TN, GEMM composition for r > 64, streaming GEMM) is hit by some case."""
import random

import pytest
import torch

from conftest import rel_err
from oracle import sow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        gran = rng.choice([1, 2, 4, 8, 8, 64])
        d_in = gran * rng.randint(1, max(1, 640 // gran))
        d_out = gran * rng.randint(1, max(1, 640 // gran))
        T = rng.choice([1, 2, 63, 64, 65, 127, 300, 1000, 2049, 4097, rng.randint(1, 5000)])
        r = rng.choice([1, 2, 7, 8, 16, 31, 50, 63, 64, 65, 80])
        r = min(r, max(1, min(d_in, d_out)))
        acc = rng.choice([None, None, "dense", "lowrank", "lowrank_big"])
        bias = rng.random() < 0.5
        scale = rng.choice([1.0, 0.5, 1.0 / max(r, 1), 2.0])
        dtype = rng.choice([torch.float32, torch.bfloat16])
        out.append((i, T, d_in, d_out, r, acc, bias, scale, dtype))
    return out


def _cases_streaming(n, seed):
    """Aligned shapes that take the streaming kernels: widths in multiples of 8 up to 1400, T from 64 to 20000 (the
    long ones reach the streaming GEMM's 160-tile threshold with a dense accumulator), ranks 2..64."""
    rng = random.Random(seed)
    out = []
    for i in range(n):
        d_in = 8 * rng.randint(4, 175)
        d_out = 8 * rng.randint(4, 175)
        T = rng.choice([64, 200, 1000, 4096, 4100, 8200, 16500, 20000])
        r = rng.choice([2, 4, 6, 8, 10, 16, 24, 32, 48, 50, 56, 62, 64])
        acc = rng.choice([None, "dense", "dense", "lowrank"])
        out.append((1000 + i, T, d_in, d_out, r, acc, rng.random() < 0.5, rng.choice([1.0, 0.25]),
                    rng.choice([torch.float32, torch.bfloat16])))
    return out


CASES = _cases(48, 20240611) + _cases_streaming(28, 777)


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-T{c[1]}-{c[2]}x{c[3]}-r{c[4]}-{c[5]}-{'b' if c[6] else 'nb'}-{str(c[8])[6:]}" for c in CASES])
def test_random_shape_vs_oracle(case):
    from sow_amd import ops
    i, T, d_in, d_out, r, acc, bias, scale, dtype = case
    gen = torch.Generator().manual_seed(9000 + i)
    x = torch.randn(T, d_in, generator=gen)
    dy = torch.randn(T, d_out, generator=gen)
    A = torch.randn(d_in, r, generator=gen) * (1.0 / max(d_in, 1) ** 0.5)
    B = torch.randn(r, d_out, generator=gen) * 0.05
    b = torch.randn(d_out, generator=gen) * 0.1 if bias else None
    ad = au = None
    if acc == "dense":
        ad = torch.randn(d_in, d_out, generator=gen) * 0.02
    elif acc == "lowrank":
        vr = min(24, d_in, d_out)
        ad, au = torch.randn(d_in, vr, generator=gen) * 0.1, torch.randn(vr, d_out, generator=gen) * 0.1
    elif acc == "lowrank_big":
        vr = min(100, d_in, d_out)
        ad, au = torch.randn(d_in, vr, generator=gen) * 0.1, torch.randn(vr, d_out, generator=gen) * 0.1
    cast = lambda t: None if t is None else t.to(dtype)
    xq, dyq, Aq, Bq, bq, adq, auq = map(cast, (x, dy, A, B, b, ad, au))
    f = lambda t: None if t is None else t.float()
    y_ref = O.sow_forward(f(xq), [f(Aq)], [f(Bq)], f(adq), f(auq), scale, f(bq))
    dx_ref, dA_ref, dB_ref, db_ref = O.sow_backward(f(dyq), f(xq), [f(Aq)], [f(Bq)], f(adq), f(auq), scale, bias)
    g = lambda t: None if t is None else t.to(DEV)
    y, h = ops.sow_forward(g(xq), g(Aq), g(Bq), g(adq), g(auq), g(bq), scale)
    dx, dA, dB, db = ops.sow_backward(g(dyq), g(xq), h, g(Aq), g(Bq), g(adq), g(auq), scale, bias)
    # fp32: north_star's 1e-5 (2e-5 for the K = T reductions of dA / dB, summation-order noise); bf16: 2e-2 of the largest magnitude
    tol, tol_w = (1e-5, 2e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    assert rel_err(y.float().cpu(), y_ref) < tol
    assert rel_err(dx.float().cpu(), dx_ref) < tol
    assert rel_err(dA.float().cpu(), dA_ref[0]) < tol_w
    assert rel_err(dB.float().cpu(), dB_ref[0]) < tol_w
    if bias:
        assert rel_err(db.float().cpu(), db_ref) < tol_w


# ------------------------------------------------------------------------------------------------ sow_gemm
def _gemm_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        gran = rng.choice([1, 2, 4, 8, 8])
        M = rng.choice([1, 7, 64, 129, 300, 1000, 2300]) if i % 4 else 8 * rng.randint(5200, 6000)   # every 4th: >= 160 tiles of 256 x 256
        N = gran * rng.randint(1, max(1, 900 // gran)) if i % 4 else 8 * rng.randint(100, 170)
        K = gran * rng.randint(1, max(1, 700 // gran)) if i % 4 else 8 * rng.randint(5, 90)
        out.append((i, M, N, K, rng.random() < 0.5 and i % 4 != 0, rng.random() < 0.5, rng.choice([torch.float32, torch.bfloat16]),
                    rng.choice([1.0, 0.5, -2.0]), rng.choice([0.0, 0.0, 1.0, 0.5]), rng.random() < 0.4))
    return out


GEMM_CASES = _gemm_cases(32, 4242)


@pytest.mark.parametrize("case", GEMM_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}x{c[3]}-{'T' if c[4] else 'N'}{'T' if c[5] else 'N'}-{str(c[6])[6:]}" for c in GEMM_CASES])
def test_random_gemm(case):
    """sow_gemm over every transpose combination, ragged and aligned extents, alpha / beta / bias, including shapes with
    >= 160 output tiles that take the streaming kernel."""
    from sow_amd import ops
    i, M, N, K, ta, tb, dtype, alpha, beta, use_bias = case
    gen = torch.Generator().manual_seed(500 + i)
    a = torch.randn((K, M) if ta else (M, K), generator=gen).to(dtype)
    b = (torch.randn((N, K) if tb else (K, N), generator=gen) * (1.0 / max(K, 1) ** 0.5)).to(dtype)
    c0 = torch.randn(M, N, generator=gen).to(dtype)
    bias = torch.randn(N, generator=gen).to(dtype) if use_bias else None
    ref = alpha * ((a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())) + beta * c0.float()
    if use_bias:
        ref = ref + bias.float()
    out = ops.gemm(a.to(DEV), b.to(DEV), trans_a=ta, trans_b=tb, out=c0.to(DEV).clone(), alpha=alpha, beta=beta,
                   bias=None if bias is None else bias.to(DEV))
    assert rel_err(out.float().cpu(), ref) < (1e-5 if dtype == torch.float32 else 2e-2)


# ------------------------------------------------------------------------------------------------ sow_qr_thin
def _qr_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        m = rng.choice([8, 50, 64, 100, 256, 512, 700, 1376])
        ncols = rng.choice([8, 40, 64, 200, 512, 1376])
        k = rng.randint(1, min(m, ncols, 160))
        out.append((i, m, ncols, k, rng.choice([torch.float32, torch.bfloat16])))
    return out


QR_CASES = _qr_cases(16, 99)


@pytest.mark.parametrize("case", QR_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}-k{c[3]}-{str(c[4])[6:]}" for c in QR_CASES])
def test_random_qr_thin(case):
    """Truncated Householder QR against torch.linalg.qr (same LAPACK sign convention) on well-conditioned Gaussians:
    Q[:, :k], R[:k, :], orthonormality, and Q R = the rank-k part the reference's qr_weight returns (utils.py:8-30)."""
    from sow_amd import ops
    i, m, ncols, k, dtype = case
    gen = torch.Generator().manual_seed(700 + i)
    w = (torch.randn(m, ncols, generator=gen) * 0.02).to(dtype)
    q, r = ops.qr_thin(w.to(DEV), k, need_r=True)
    q_ref, r_ref = O.qr_weight(w, k)
    tol = 5e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(q.float().cpu(), q_ref.float()) < tol
    assert rel_err(r.float().cpu(), r_ref.float()) < tol
    if dtype == torch.float32:
        assert rel_err((q.t() @ q).cpu(), torch.eye(k)) < 1e-5


# ------------------------------------------------------------------------------------------------ module level
def _layer_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        out.append((i, rng.choice([(3, 17), (2, 5, 11), (64,), (4, 300)]), 8 * rng.randint(2, 40), 8 * rng.randint(2, 40),
                    rng.choice([4, 8, 16]), rng.choice([1, 2, 3]), rng.random() < 0.5, rng.choice([torch.float32, torch.bfloat16])))
    return out


LAYER_CASES = _layer_cases(12, 31337)


@pytest.mark.parametrize("case", LAYER_CASES, ids=[f"{c[0]}-{'x'.join(map(str, c[1]))}-{c[2]}x{c[3]}-r{c[4]}-n{c[5]}-{str(c[7])[6:]}" for c in LAYER_CASES])
def test_random_layer_autograd(case):
    """SoWLinear through torch.autograd with n_iter factor pairs (concatenated for the C ABI), N-D inputs and bias:
    outputs and every parameter gradient against the oracle."""
    from sow_amd import SoWLinear
    i, lead, d_in, d_out, r, n_iter, bias, dtype = case
    torch.manual_seed(100 + i)
    layer = SoWLinear(d_in, d_out, bias=bias, rank=r, n_iter=n_iter, scale=0.5, init_method="normal", device=DEV, dtype=dtype)
    with torch.no_grad():
        for p_ in list(layer.downscale_weights) + list(layer.upscale_weights):
            p_.copy_((torch.randn(p_.shape) * 0.1).to(dtype))
        if bias:
            layer.bias.copy_((torch.randn(d_out) * 0.1).to(dtype))
    x = torch.randn(*lead, d_in).to(dtype).to(DEV).requires_grad_(True)
    dy = torch.randn(*lead, d_out).to(dtype).to(DEV)
    y = layer(x)
    y.backward(dy)
    f = lambda t: t.detach().float().cpu()
    As, Bs = [f(p_) for p_ in layer.downscale_weights], [f(p_) for p_ in layer.upscale_weights]
    y_ref = O.sow_forward(f(x), As, Bs, None, None, 0.5, f(layer.bias) if bias else None)
    dx_ref, dA_ref, dB_ref, db_ref = O.sow_backward(f(dy), f(x), As, Bs, None, None, 0.5, bias)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert y.shape == (*lead, d_out)
    assert rel_err(f(y), y_ref) < tol and rel_err(f(x.grad), dx_ref) < tol
    for j in range(n_iter):
        assert rel_err(f(layer.downscale_weights[j].grad), dA_ref[j]) < 2 * tol
        assert rel_err(f(layer.upscale_weights[j].grad), dB_ref[j]) < 2 * tol
    if bias:
        assert rel_err(f(layer.bias.grad), db_ref) < 2 * tol


# ------------------------------------------------------------------------------------------------ TensorTrain
def _tt_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        order = rng.choice([2, 3, 4])
        M, N = rng.randint(9, 300), rng.randint(9, 300)
        rk = rng.choice([2, 3, 4, 6])
        out.append((i, M, N, order, [1] + [rk] * (order - 1) + [1]))
    return out


TT_CASES = _tt_cases(12, 5150)


@pytest.mark.parametrize("case", TT_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}-o{c[3]}-r{c[4][1]}" for c in TT_CASES])
def test_random_tt_from_matrix(case):
    """TensorTrain.from_matrix on random Gaussian matrices (tt.py:48-67, 111-140): padded core dimensions
    (ceil(n**(1/order)), bit-exact), core shapes, and the reconstruction / to_matrix / norm against the oracle's cores.
    Individual cores are compared through the reconstruction only (Q columns are defined up to rounding when a
    truncated unfolding is near-degenerate)."""
    from sow_amd import TensorTrain
    i, M, N, order, ranks = case
    gen = torch.Generator().manual_seed(300 + i)
    mat = torch.randn(M, N, generator=gen)
    cores_ref = O.tt_from_matrix(mat, ranks, padding=True)
    tt = TensorTrain.from_matrix(mat.to(DEV), list(ranks), padding=True)
    assert [tuple(c.shape) for c in tt.cores] == [tuple(c.shape) for c in cores_ref]
    assert int(tt.cores[0].shape[1]) == O.tt_core_dim(M, order) and int(tt.cores[0].shape[2]) == O.tt_core_dim(N, order)
    rec_ref = O.tt_to_matrix(cores_ref, (M, N))
    assert rel_err(tt.to_matrix((M, N)).cpu(), rec_ref) < 1e-4
    nrm_ref = float(O.tt_inner(cores_ref, cores_ref, mode="full"))   # the reference's norm() is the SQUARED norm (tt.py:257-260)
    assert abs(float(tt.norm()) - nrm_ref) / nrm_ref < 1e-4
    both = tt + tt
    assert rel_err(both.to_matrix((M, N)).cpu(), 2 * rec_ref) < 1e-4
