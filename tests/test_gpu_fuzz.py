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
