"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(ctypes -> libsow_amd.so), against the golden vectors generated from the reference and against the
CPU oracle on seeded inputs.  Tolerances: fp32 1e-5 relative to the largest reference magnitude
(BASELINE.json north_star), bf16 2e-2; integer / index work bit-exact."""
import math

import pytest
import torch
import torch.nn as nn

from conftest import load_golden, rel_err
from oracle import sow_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda"


def _mk_layer(g, dtype=torch.float32):
    from sow_amd import SoWLinear
    n_iter, rank = int(g["n_iter"]), int(g["rank"])
    d_in, d_out = g["A0"].shape[0], g["B0"].shape[1]
    layer = SoWLinear(d_in, d_out, bias="bias" in g, rank=rank, n_iter=n_iter, scale=float(g["scale"]),
                      init_method="normal", device=DEV, dtype=dtype, init_params=False)
    for i in range(n_iter):
        layer.downscale_weights[i].data = g[f"A{i}"].to(DEV, dtype)
        layer.upscale_weights[i].data = g[f"B{i}"].to(DEV, dtype)
    if "bias" in g:
        layer.bias.data = g["bias"].to(DEV, dtype)
    if "acc_down" in g:
        layer.acc_downweight = nn.Parameter(g["acc_down"].to(DEV, dtype), requires_grad=False)
    if "acc_up" in g:
        layer.acc_upweight = nn.Parameter(g["acc_up"].to(DEV, dtype), requires_grad=False)
    return layer


FWD = ["cfg1_noacc", "cfg1_bias_dense", "cfg1_lowrank", "r50_3d", "r50_3d_dense", "niter2", "niter3_odd", "tiny_T1"]


def test_library_loaded():
    from sow_amd import _lib
    assert _lib.load().sow_version() >= 100
    assert torch.cuda.is_available()


@pytest.mark.parametrize("name", FWD)
def test_forward_backward_golden_fp32(name):
    g = load_golden("fwdbwd_" + name)
    layer = _mk_layer(g)
    x = g["x"].to(DEV).requires_grad_(True)
    y = layer(x)
    assert y.shape == g["y"].shape
    assert rel_err(y.detach().cpu(), g["y"]) < TOL
    y.backward(g["dy"].to(DEV))
    assert rel_err(x.grad.cpu(), g["dx"]) < TOL
    for i in range(int(g["n_iter"])):
        assert rel_err(layer.downscale_weights[i].grad.cpu(), g[f"dA{i}"]) < TOL
        assert rel_err(layer.upscale_weights[i].grad.cpu(), g[f"dB{i}"]) < TOL
    if "bias" in g:
        assert rel_err(layer.bias.grad.cpu(), g["dbias"]) < TOL


SHAPES = [
    # T, d_in, d_out, r, acc, bias
    (64, 256, 256, 8, None, False),           # BASELINE config 1
    (1000, 512, 512, 50, None, False),        # llama_60m attention proj
    (777, 512, 1376, 50, "dense", False),     # llama_60m gate/up after the first accumulate
    (515, 1376, 512, 50, None, True),
    (300, 768, 768, 50, "lowrank", True),     # north-star width
    (130, 96, 200, 64, None, True),           # r = 64: no free column for the bias trick
    (257, 100, 36, 70, "dense", True),        # r > 64: GEMM composition path
    (90, 50, 70, 7, "lowrank", False),        # odd everything: scalar paths
    (200, 128, 64, 16, "lowrank_big", False), # low-rank accumulator wider than 64
    # T >= 4096 takes the bf16 streaming kernels (chain2: LDS-DMA rings + loader wave + transposed reads)
    (4100, 512, 1376, 50, None, True),        # ragged token tail, partial last factor chunk (1376 = 5.375 * 256)
    (4224, 1376, 512, 50, "dense", False),    # beta = 1 epilogue on top of the dense-accumulator GEMM
    (4096, 768, 768, 8, None, False),         # rank 8 (one MFMA k-step), north-star width
    (4100, 512, 1376, 50, "dense", True),     # dense accumulator + bias below the streaming GEMM's tile threshold: GEMM + chain (beta = 1)
    (16500, 776, 1000, 50, "dense", True),    # dense accumulator + bias: streaming GEMM with K-extension (>= 160 tiles both ways), ragged M / N tiles, K tails of 8
    (5000, 264, 72, 50, "lowrank", True),     # widths that are not multiples of 64
    (1100, 2304, 8200, 8, "dense", True),     # finetune-like: short T, long K -> one-wave-per-SIMD streaming GEMM (gemm3) with the K-extension
]


def _rand_case(T, d_in, d_out, r, acc, bias, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(T, d_in, generator=gen)
    dy = torch.randn(T, d_out, generator=gen)
    A = torch.linalg.qr(torch.randn(d_in, max(r, 1), generator=gen) * 0.02)[0][:, :r].contiguous() if r <= d_in else torch.randn(d_in, r, generator=gen) * 0.05
    B = torch.randn(r, d_out, generator=gen) * 0.02
    b = torch.randn(d_out, generator=gen) * 0.1 if bias else None
    ad = au = None
    if acc == "dense":
        ad = torch.randn(d_in, d_out, generator=gen) * 0.02
    elif acc == "lowrank":
        ad, au = torch.randn(d_in, 24, generator=gen) * 0.1, torch.randn(24, d_out, generator=gen) * 0.1
    elif acc == "lowrank_big":
        ad, au = torch.randn(d_in, 100, generator=gen) * 0.1, torch.randn(100, d_out, generator=gen) * 0.1
    return x, dy, A, B, b, ad, au


@pytest.mark.parametrize("idx", range(len(SHAPES)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_backward_vs_oracle(idx, dtype):
    from sow_amd import ops
    T, d_in, d_out, r, acc, bias = SHAPES[idx]
    scale = 0.75
    x, dy, A, B, b, ad, au = _rand_case(T, d_in, d_out, r, acc, bias, 1234 + idx)
    cast = lambda t: None if t is None else t.to(dtype)
    xq, dyq, Aq, Bq, bq, adq, auq = map(cast, (x, dy, A, B, b, ad, au))
    f = lambda t: None if t is None else t.float()
    y_ref = O.sow_forward(f(xq), [f(Aq)], [f(Bq)], f(adq), f(auq), scale, f(bq))
    dx_ref, dA_ref, dB_ref, db_ref = O.sow_backward(f(dyq), f(xq), [f(Aq)], [f(Bq)], f(adq), f(auq), scale, bias)
    g = lambda t: None if t is None else t.to(DEV)
    y, h = ops.sow_forward(g(xq), g(Aq), g(Bq), g(adq), g(auq), g(bq), scale)
    dx, dA, dB, db = ops.sow_backward(g(dyq), g(xq), h, g(Aq), g(Bq), g(adq), g(auq), scale, bias)
    tol = TOL if dtype == torch.float32 else 2e-2
    assert rel_err(y.float().cpu(), y_ref) < tol
    assert rel_err(dx.float().cpu(), dx_ref) < tol
    assert rel_err(dA.float().cpu(), dA_ref[0]) < tol
    assert rel_err(dB.float().cpu(), dB_ref[0]) < tol
    if bias:
        assert rel_err(db.float().cpu(), db_ref) < tol


def test_empty_batch():
    from sow_amd import SoWLinear
    layer = SoWLinear(32, 16, bias=True, rank=4, init_method="normal", device=DEV)
    x = torch.zeros(0, 32, device=DEV, requires_grad=True)
    y = layer(x)
    assert y.shape == (0, 16)
    y.sum().backward()
    assert float(layer.downscale_weights[0].grad.abs().max()) == 0.0


def test_grad_accumulation_beta():
    """grad_beta = 1 accumulates into existing gradient buffers (flat-bucket path)."""
    from sow_amd import ops
    x, dy, A, B, b, _, _ = _rand_case(300, 128, 96, 8, None, True, 77)
    g = lambda t: t.to(DEV)
    y, h = ops.sow_forward(g(x), g(A), g(B), None, None, g(b), 1.0)
    _, dA1, dB1, db1 = ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 1.0, True)
    bufs = (dA1.clone(), dB1.clone(), db1.clone())
    ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 1.0, True, out=bufs, grad_beta=1.0)
    assert rel_err(bufs[0].cpu(), 2 * dA1.cpu()) < TOL and rel_err(bufs[1].cpu(), 2 * dB1.cpu()) < TOL
    assert rel_err(bufs[2].cpu(), 2 * db1.cpu()) < TOL


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_streaming_kernels_agree_with_generic_kernels(dtype):
    """A/B: the streaming chain kernels (chain2 bf16 / chain2f fp32) against the generic chain kernel on the same inputs."""
    from sow_amd import _lib, ops
    T, di, do, r = 4608, 512, 1376, 50
    x, dy, A, B, b, _, _ = _rand_case(T, di, do, r, None, True, 55)
    g = lambda t: t.to(DEV, dtype)
    y2, h2 = ops.sow_forward(g(x), g(A), g(B), None, None, g(b), 0.5)
    dx2 = ops.sow_backward(g(dy), g(x), h2, g(A), g(B), None, None, 0.5, True)[0]
    with _lib.switch(FORCE_CHAIN_V1=1):
        y1, h1 = ops.sow_forward(g(x), g(A), g(B), None, None, g(b), 0.5)
        dx1 = ops.sow_backward(g(dy), g(x), h1, g(A), g(B), None, None, 0.5, True)[0]
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-6
    assert rel_err(y2.float().cpu(), y1.float().cpu()) < tol and rel_err(dx2.float().cpu(), dx1.float().cpu()) < tol
    assert rel_err(h2.float().cpu().view(T, 64)[:, :50], h1.float().cpu().view(T, 64)[:, :50]) < tol
    assert torch.equal(h2.view(T, 64)[:, 63].float().cpu(), torch.ones(T)) and float(h2.view(T, 64)[:, 50:63].abs().max()) == 0.0


FUSED_H = [
    # T, d_in, d_out, r, bias      (bf16, dense accumulator; >= 160 output tiles and N <= 512 select gemm2h)
    (20500, 512, 512, 50, True),    # llama_60m attention shape: forward and backward fused, ragged token tail, 1 fix-up dword
    (20500, 776, 264, 34, False),   # forward fused (ragged second column tile, K tail of 8); backward N = 776 stays unfused
    (41000, 256, 512, 44, True),    # backward with a single column tile; 2 fix-up dwords
    (20500, 512, 504, 46, False),   # 3 fix-up dwords, N % 64 != 0
    (20500, 512, 512, 48, True),    # r % 8 == 0: no straddling piece
    (20500, 384, 512, 64, False),   # r = 64: no ones column
    (20500, 512, 384, 4, False),    # r = 4: smallest fused rank (a 16-byte piece spans two rows of A)
    (20500, 512, 384, 2, False),    # r = 2: not fused (falls back to GEMM + generic chain)
]


@pytest.mark.parametrize("kernel", ["gemm4h", "gemm2h"])
@pytest.mark.parametrize("idx", range(len(FUSED_H)))
def test_dense_layer_with_in_kernel_projection(idx, kernel):
    """gemm4h / gemm2h (projection h = s x A computed inside the dense GEMM, one launch per pass: gemm4h by a streaming pass
    of its own ahead of the anti-phase main loop, gemm2h -- NO_GEMM4H -- alongside its main loop) against the oracle and
    against the two-launch path (H-only chain + K-extended GEMM).  The last row of A is spiked so that a missing fix-up of its
    straddling 16-byte piece (the only piece whose tail crosses the end of the buffer) cannot hide in the tolerance."""
    from sow_amd import ops
    T, d_in, d_out, r, bias = FUSED_H[idx]
    scale = 0.75
    x, dy, A, B, b, ad, _ = _rand_case(T, d_in, d_out, r, "dense", bias, 4321 + idx)
    A = A.clone()
    A[-1] *= 30.0
    cast = lambda t: None if t is None else t.to(torch.bfloat16)
    xq, dyq, Aq, Bq, bq, adq = map(cast, (x, dy, A, B, b, ad))
    f = lambda t: None if t is None else t.float()
    y_ref = O.sow_forward(f(xq), [f(Aq)], [f(Bq)], f(adq), None, scale, f(bq))
    dx_ref, dA_ref, dB_ref, db_ref = O.sow_backward(f(dyq), f(xq), [f(Aq)], [f(Bq)], f(adq), None, scale, bias)
    g = lambda t: None if t is None else t.to(DEV)
    from sow_amd import _lib
    with _lib.switch(NO_GEMM4H=1 if kernel == "gemm2h" else 0):
        y, h = ops.sow_forward(g(xq), g(Aq), g(Bq), g(adq), None, g(bq), scale)
        dx, dA, dB, db = ops.sow_backward(g(dyq), g(xq), h, g(Aq), g(Bq), g(adq), None, scale, bias)
    with _lib.switch(NO_FUSED_H=1):
        y0, h0 = ops.sow_forward(g(xq), g(Aq), g(Bq), g(adq), None, g(bq), scale)
        dx0, dA0, dB0, db0 = ops.sow_backward(g(dyq), g(xq), h0, g(Aq), g(Bq), g(adq), None, scale, bias)
    tol = 2e-2
    assert rel_err(y.float().cpu(), y_ref) < tol and rel_err(dx.float().cpu(), dx_ref) < tol
    assert rel_err(dA.float().cpu(), dA_ref[0]) < tol and rel_err(dB.float().cpu(), dB_ref[0]) < tol
    if bias:
        assert rel_err(db.float().cpu(), db_ref) < tol
    # the column of dX that the last row of A feeds, and the ranks of h its straddling piece holds
    assert rel_err(dx.float().cpu()[:, -1], dx_ref[:, -1]) < tol
    hv, h0v = h.view(T, 64).float().cpu(), h0.view(T, 64).float().cpu()
    h_ref = scale * (f(xq) @ f(Aq))
    lo = (r // 8) * 8 if r % 8 else r - 8
    assert rel_err(hv[:, lo:r], h_ref[:, lo:r]) < 1e-2
    assert rel_err(hv[:, :r], h0v[:, :r]) < 1e-2 and rel_err(y.float().cpu(), y0.float().cpu()) < 1e-2
    assert rel_err(dx.float().cpu(), dx0.float().cpu()) < 1e-2
    if r < 64:
        assert torch.equal(hv[:, 63], torch.ones(T))
        if r < 63:
            assert float(hv[:, r:63].abs().max()) == 0.0


def test_backward_phases_split_equals_fused():
    """sow_backward_ex: DATA then WEIGHTS on the same workspace == the fused call (bit-exact)."""
    from sow_amd import _lib, ops
    for dtype, (T, di, do, r) in ((torch.bfloat16, (4200, 512, 264, 50)), (torch.float32, (300, 96, 160, 8))):
        x, dy, A, B, b, _, _ = _rand_case(T, di, do, r, None, True, 99)
        g = lambda t: t.to(DEV, dtype)
        y, h = ops.sow_forward(g(x), g(A), g(B), None, None, g(b), 0.5)
        ref = ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 0.5, True)
        ws = torch.empty(ops.workspace_bytes(T, di, do, r, 0, 0, dtype) + 256, dtype=torch.uint8, device=DEV)
        dx = torch.empty(T, di, dtype=dtype, device=DEV)
        outs = (torch.empty_like(ref[1]), torch.empty_like(ref[2]), torch.empty_like(ref[3]))
        ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 0.5, True, out=outs, phases=_lib.BWD_DATA, dx=dx, workspace=ws)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 0.5, True, out=outs, phases=_lib.BWD_WEIGHTS, dx=dx, workspace=ws)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(dx, ref[0]) and torch.equal(outs[0], ref[1]) and torch.equal(outs[1], ref[2]) and torch.equal(outs[2], ref[3])
        # three calls: DATA, the slab partials, their reduction (the last one on another stream)
        outs3 = (torch.zeros_like(ref[1]), torch.zeros_like(ref[2]), torch.zeros_like(ref[3]))
        dx3 = torch.empty(T, di, dtype=dtype, device=DEV)
        kw = dict(out=outs3, dx=dx3, workspace=ws)
        ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 0.5, True, phases=_lib.BWD_DATA | _lib.BWD_WEIGHTS_PARTIAL, **kw)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 0.5, True, phases=_lib.BWD_WEIGHTS_REDUCE, **kw)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(dx3, ref[0]) and torch.equal(outs3[0], ref[1]) and torch.equal(outs3[1], ref[2]) and torch.equal(outs3[2], ref[3])


def test_deferred_batched_reduce_is_bit_identical():
    """sow_reduce_batch: DATA | WEIGHTS_PARTIAL per layer (own workspaces), one batched reduction at the end == the
    per-layer WEIGHTS phase, bit for bit; second step reuses the descriptors; grad_beta = 1 accumulates."""
    from sow_amd import _lib, ops
    cases = [(4200, 512, 264, 50, True), (1000, 96, 160, 8, False), (5000, 264, 520, 34, True), (300, 128, 64, 62, True)]
    for dtype in (torch.bfloat16, torch.float32):
        g = lambda t: None if t is None else t.to(DEV, dtype)
        layers = []
        for i, (T, di, do, r, bias) in enumerate(cases):
            x, dy, A, B, b, _, _ = _rand_case(T, di, do, r, None, bias, 500 + i)
            x, dy, A, B, b = map(g, (x, dy, A, B, b))
            y, h = ops.sow_forward(x, A, B, None, None, b, 0.5)
            ref = ops.sow_backward(dy, x, h, A, B, None, None, 0.5, bias)
            ws = torch.empty(ops.workspace_bytes(T, di, do, r, 0, 0, dtype) + 256, dtype=torch.uint8, device=DEV)
            out = (torch.zeros_like(ref[1]), torch.zeros_like(ref[2]), torch.zeros_like(ref[3]) if bias else None)
            layers.append((x, dy, h, A, B, bias, ref, ws, out, torch.empty_like(ref[0])))
        dr = ops.DeferredReduce()
        for step, beta in enumerate((0.0, 0.0, 1.0)):
            for (x, dy, h, A, B, bias, ref, ws, out, dx) in layers:
                ops.sow_backward(dy, x, h, A, B, None, None, 0.5, bias, out=out, grad_beta=beta, dx=dx, workspace=ws,
                                 phases=_lib.BWD_DATA | _lib.BWD_WEIGHTS_PARTIAL)
                dr.add(x, B, out, beta, ws)
            dr.run()
            torch.cuda.synchronize()
            for (x, dy, h, A, B, bias, ref, ws, out, dx) in layers:
                if beta == 0.0:
                    assert torch.equal(dx, ref[0]) and torch.equal(out[0], ref[1]) and torch.equal(out[1], ref[2])
                    assert (not bias) or torch.equal(out[2], ref[3])
                else:   # accumulated onto the previous step's gradients
                    assert rel_err(out[0].float().cpu(), 2 * ref[1].float().cpu()) < 1e-2
                    assert rel_err(out[1].float().cpu(), 2 * ref[2].float().cpu()) < 1e-2


def test_bucket_attach_gradients_equal_autograd_path():
    """FactorBucket.attach(): SoWLinear's backward accumulates straight into the flat gradient buffer and the reductions
    of all layers run in one launch at finalize() -- same gradients, bit for bit, as the ordinary autograd path; a second
    backward before finalize() (gradient accumulation) adds up."""
    import copy
    from sow_amd import SoWLinear
    from sow_amd.dp import FactorBucket, factor_parameters
    from sow_amd.optimizer import FactorAdamW
    torch.manual_seed(3)
    dims = [(96, 160, 8, False), (160, 64, 16, False), (64, 200, 8, True), (200, 72, 34, False)]
    for dtype in (torch.bfloat16, torch.float32):
        ref = torch.nn.ModuleList([SoWLinear(i, o, bias=b, rank=r, init_method="normal", device=DEV, dtype=dtype) for i, o, r, b in dims])
        for m in ref:
            torch.nn.init.normal_(m.upscale_weights[0], std=0.05)
        net = copy.deepcopy(ref)
        bucket = FactorBucket(factor_parameters(net))
        assert bucket.attach(net) == 3          # the layer with a bias keeps the autograd path
        x = torch.randn(700, 96, device=DEV, dtype=dtype)

        def run(mods, xin):
            h = xin
            for m in mods:
                h = torch.tanh(m(h))
            return h.float().square().mean()

        run(ref, x).backward()
        bucket.zero_grad()
        run(net, x).backward()
        bucket.finalize()
        torch.cuda.synchronize()
        for a, b in zip(ref, net):
            assert torch.equal(a.downscale_weights[0].grad, b.downscale_weights[0].grad)
            assert torch.equal(a.upscale_weights[0].grad, b.upscale_weights[0].grad)
            if a.bias is not None:
                assert torch.equal(a.bias.grad, b.bias.grad)
        # accumulation: a second backward without zero_grad; FactorAdamW.step finalizes by itself
        run(ref, x).backward()
        run(net, x).backward()
        run(ref, x).backward()
        run(net, x).backward()      # partials of the previous backward still pending: the sink finalizes them first
        opt = FactorAdamW(bucket, lr=1e-3)
        opt.step()
        torch.cuda.synchronize()
        for a, b in zip(ref, net):
            assert rel_err(b.downscale_weights[0].grad.float().cpu(), a.downscale_weights[0].grad.float().cpu()) < (1e-2 if dtype == torch.bfloat16 else 1e-5)


def test_full_size_properties_bf16():
    """North-star size (T=32768, d=768, r=50, bf16): size-independent properties.
    y is linear in x; <dY, Y> = <dA, A> = <dB, B> (y is homogeneous of degree 1 in A and in B);
    <dY, Y> = <dX, X> (adjoint identity)."""
    from sow_amd import ops
    T, d, r = 32768, 768, 50
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(T, d, generator=gen, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn(T, d, generator=gen, device=DEV, dtype=torch.bfloat16)
    A = (torch.randn(d, r, generator=gen, device=DEV) * 0.04).bfloat16()
    B = (torch.randn(r, d, generator=gen, device=DEV) * 0.04).bfloat16()
    y, h = ops.sow_forward(x, A, B, None, None, None, 1.0)
    y2, _ = ops.sow_forward((2 * x.float()).bfloat16(), A, B, None, None, None, 1.0)
    assert rel_err(y2.float().cpu(), 2 * y.float().cpu()) < 2e-2
    dx, dA, dB, _ = ops.sow_backward(dy, x, h, A, B, None, None, 1.0, False)
    ip = lambda a, b: float((a.double() * b.double()).sum())
    base = ip(dy, y)
    assert abs(ip(dA, A) - base) < 2e-2 * abs(base)
    assert abs(ip(dB, B) - base) < 2e-2 * abs(base)
    assert abs(ip(dx, x) - base) < 2e-2 * abs(base)
    # spot-check 64 rows against the oracle
    rows = torch.arange(0, T, T // 64)
    y_ref = O.sow_forward(x[rows].float().cpu(), [A.float().cpu()], [B.float().cpu()], None, None, 1.0, None)
    assert rel_err(y[rows].float().cpu(), y_ref) < 2e-2


def test_full_size_fp32_rows_vs_oracle():
    from sow_amd import ops
    T, d_in, d_out, r = 32768, 512, 1376, 50
    gen = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(T, d_in, generator=gen, device=DEV)
    dy = torch.randn(T, d_out, generator=gen, device=DEV)
    A = torch.randn(d_in, r, generator=gen, device=DEV) * 0.04
    B = torch.randn(r, d_out, generator=gen, device=DEV) * 0.04
    y, h = ops.sow_forward(x, A, B, None, None, None, 0.5)
    dx, dA, dB, _ = ops.sow_backward(dy, x, h, A, B, None, None, 0.5, False)
    rows = torch.arange(7, T, T // 50)
    y_ref = O.sow_forward(x[rows].cpu(), [A.cpu()], [B.cpu()], None, None, 0.5, None)
    assert rel_err(y[rows].cpu(), y_ref) < TOL
    dx_ref, dA_ref, dB_ref, _ = O.sow_backward(dy.cpu(), x.cpu(), [A.cpu()], [B.cpu()], None, None, 0.5, False)
    assert rel_err(dx.cpu(), dx_ref) < TOL
    assert rel_err(dA.cpu(), dA_ref[0]) < 2e-5   # K = 32768 reduction: summation-order noise
    assert rel_err(dB.cpu(), dB_ref[0]) < 2e-5


# ---------------------------------------------------------------------------------------------
ACC = {"lowrank_grow": (48, 40, "normal_QR"), "niter2_normal": (36, 44, "normal"),
       "dense_prepare_style": (64, 96, "normal_QR"), "cfg1": (256, 256, "normal_QR")}
QR_TOL = 5e-5  # Householder on the GPU vs LAPACK: same reflectors, different summation order


@pytest.mark.parametrize("name", list(ACC))
def test_accumulate_trace_golden(name, monkeypatch):
    from sow_amd import SoWLinear
    g = load_golden("accumulate_" + name)
    d_in, d_out, init = ACC[name]
    rank, n_iter, n_calls = int(g["rank"]), int(g["n_iter"]), int(g["n_calls"])
    layer = SoWLinear(d_in, d_out, bias=False, rank=rank, n_iter=n_iter, scale=float(g["scale"]), init_method=init,
                      device=DEV, init_params=False)
    layer.virtual_rank = int(g["vr0"])
    for c in range(n_calls):
        for i in range(n_iter):
            layer.downscale_weights[i].data = g[f"c{c}_A{i}_in"].to(DEV)
            layer.upscale_weights[i].data = g[f"c{c}_B{i}_in"].to(DEV)
        draws = [g[f"c{c}_draw{i}"].to(DEV) for i in range(n_iter)]
        it = iter(draws)
        if init == "normal_QR":
            monkeypatch.setattr(layer, "_fresh_gaussian", lambda shape, device, dtype: next(it).to(dtype))
        else:
            orig = nn.init.normal_
            monkeypatch.setattr(nn.init, "normal_", lambda t, *a, **k: t.copy_(next(it)))
        layer.accumulate()
        if init != "normal_QR":
            monkeypatch.setattr(nn.init, "normal_", orig)
        assert layer.virtual_rank == int(g[f"c{c}_vr"])          # integer schedule bit-exact
        assert tuple(layer.acc_downweight.shape) == tuple(g[f"c{c}_acc_down"].shape)
        assert tuple(layer.acc_upweight.shape) == tuple(g[f"c{c}_acc_up"].shape)
        if layer.acc_upweight.numel():
            # Q, R individually (same sign convention) and their product
            assert rel_err(layer.acc_downweight.cpu(), g[f"c{c}_acc_down"]) < QR_TOL
            assert rel_err(layer.acc_upweight.cpu(), g[f"c{c}_acc_up"]) < QR_TOL
        else:
            # a dense accumulator that absorbed earlier truncated-QR stages inherits their tolerance
            assert rel_err(layer.acc_downweight.cpu(), g[f"c{c}_acc_down"]) < (TOL if c == 0 or name == "dense_prepare_style" else QR_TOL)
        for i in range(n_iter):
            assert rel_err(layer.downscale_weights[i].data.cpu(), g[f"c{c}_A{i}_out"]) < QR_TOL
            assert float(layer.upscale_weights[i].data.abs().max()) == 0.0


def test_reset_parameters_normal_qr_matches_oracle(monkeypatch):
    """a3: SoWLinear(init_method='normal_QR') -> A = Q[:, :r], B = R[:r, :] of the QR of the Gaussian draw."""
    from sow_amd import SoWLinear
    gen = torch.Generator().manual_seed(12)
    draw = torch.randn(384, 200, generator=gen) * 0.02
    monkeypatch.setattr(SoWLinear, "_fresh_gaussian", lambda self, shape, device, dtype: draw.to(device, dtype))
    for dtype in (torch.float32, torch.bfloat16):
        layer = SoWLinear(384, 200, bias=True, rank=16, init_method="normal_QR", device=DEV, dtype=dtype)
        a_ref, b_ref = O.sow_init_normal_qr(draw, 16, dtype)
        tol = QR_TOL if dtype == torch.float32 else 1e-2
        assert layer.downscale_weights[0].dtype == dtype and layer.upscale_weights[0].shape == (16, 200)
        assert rel_err(layer.downscale_weights[0].data.float().cpu(), a_ref.float()) < tol
        assert rel_err(layer.upscale_weights[0].data.float().cpu(), b_ref.float()) < tol
        assert float(layer.bias.data.abs().max()) == 0.0


def test_qr_golden():
    from sow_amd import ops
    g = load_golden("qr_svd")
    for k in ("tall", "wide", "square", "gauss002"):
        q, r = ops.qr_thin(g[f"{k}_in"].to(DEV), int(g[f"{k}_rank"]))
        assert rel_err(q.cpu(), g[f"{k}_q"]) < QR_TOL, k
        assert rel_err(r.cpu(), g[f"{k}_r"]) < QR_TOL, k
        m, n = g[f"{k}_in"].shape
        qf, rf = ops.qr_thin(g[f"{k}_in"].to(DEV), min(m, n))
        assert rel_err(qf.cpu(), g[f"{k}_qfull"]) < 2e-4, k   # later columns accumulate more rounding
        assert rel_err((qf @ rf).cpu(), g[f"{k}_in"]) < TOL, k
    # adversarial input: the 8 leading columns are parallel, so reflectors 2..8 are built from rounding
    # noise and differ between ANY two implementations (LAPACK builds included).  Well-defined parts:
    # the first reflector, orthonormality, exact reproduction of the factored panel, and the (kept, not
    # "fixed") lossiness of unpivoted truncation.
    w = g["rankdef_in"].to(DEV)
    q, r = ops.qr_thin(w, 8)
    assert rel_err(q[:, 0].cpu(), g["rankdef_q"][:, 0]) < QR_TOL and rel_err(r[0].cpu(), g["rankdef_r"][0]) < QR_TOL
    assert rel_err((q.t() @ q).cpu(), torch.eye(8)) < 1e-5
    assert rel_err((q @ r)[:, :8].cpu(), g["rankdef_in"][:, :8]) < 1e-5
    assert rel_err((q @ r).cpu(), g["rankdef_in"]) > 0.1
    qb, rb = ops.qr_thin(g["bf16_in"].to(DEV, torch.bfloat16), 6)
    assert qb.dtype == torch.bfloat16
    assert rel_err(qb.float().cpu(), g["bf16_q"]) < 1e-2 and rel_err(rb.float().cpu(), g["bf16_r"]) < 1e-2


def test_qr_llama_shapes_orthonormal():
    from sow_amd import ops
    for m, n, k in ((512, 512, 50), (1376, 512, 50), (512, 1376, 50), (4096, 64, 8)):
        w = torch.randn(m, n, device=DEV) * 0.02
        q, _ = ops.qr_thin(w, k, need_r=False)
        eye = (q.t() @ q).cpu()
        assert rel_err(eye, torch.eye(k)) < 1e-5
        q_ref, _ = O.qr_weight(w.cpu(), k)
        assert rel_err(q.cpu(), q_ref) < QR_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm(dtype, ta, tb):
    from sow_amd import ops
    gen = torch.Generator().manual_seed(3)
    for (M, N, K) in ((256, 256, 256), (130, 70, 50), (512, 1376, 50), (33, 257, 129)):
        a = torch.randn((K, M) if ta else (M, K), generator=gen).to(dtype)
        b = torch.randn((N, K) if tb else (K, N), generator=gen).to(dtype)
        c0 = torch.randn(M, N, generator=gen).to(dtype)
        ref = 0.5 * ((a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())) + 2.0 * c0.float()
        out = ops.gemm(a.to(DEV), b.to(DEV), trans_a=ta, trans_b=tb, out=c0.to(DEV).clone(), alpha=0.5, beta=2.0)
        assert rel_err(out.float().cpu(), ref) < (TOL if dtype == torch.float32 else 2e-2), (M, N, K)


@pytest.mark.parametrize("kernel", ["auto", "8wave", "1wave", "small", "gemm4"])
@pytest.mark.parametrize("tb", [False, True])
def test_gemm_streaming_bf16(tb, kernel):
    """bf16 products with >= 160 tiles of 256x256 take the LDS-DMA streaming kernels (gemm2: 8 waves, K < 2048; gemm3: one
    wave per SIMD, K >= 2048; the GEMM3 / GEMM3S switches force either on every shape): ragged M / N tiles, K tails of 8
    and 32, one- and two-stage K, alpha / beta / bias epilogue."""
    from sow_amd import _lib, ops
    if kernel == "small":
        sel = dict(GEMM4=0, GEMM3S=1)                         # 128x128-tile kernel (gemm3s) on every shape
    elif kernel == "gemm4":
        sel = dict(GEMM4=1)                                   # anti-phase wave groups, 64-wide K-tiles (gemm4) on every shape
    elif kernel != "auto":
        sel = dict(GEMM4=0, GEMM3S=0, GEMM3=1 if kernel == "1wave" else 0)
    else:
        sel = {}
    gen = torch.Generator().manual_seed(5)
    with _lib.switch(**sel):
        for (M, N, K) in ((40960, 256, 64), (16484, 1376, 1376), (20000, 520, 520), (45000, 72, 40), (2100, 4500, 2056), (1000, 2056, 2304)):
            a = torch.randn(M, K, generator=gen).to(torch.bfloat16)
            b = (torch.randn((N, K) if tb else (K, N), generator=gen) * 0.1).to(torch.bfloat16)
            c0 = torch.randn(M, N, generator=gen).to(torch.bfloat16)
            bias = torch.randn(N, generator=gen).to(torch.bfloat16)
            ref = 0.5 * (a.float() @ (b.float().t() if tb else b.float())) + 2.0 * c0.float() + bias.float()
            out = ops.gemm(a.to(DEV), b.to(DEV), trans_b=tb, out=c0.to(DEV).clone(), alpha=0.5, beta=2.0, bias=bias.to(DEV))
            assert rel_err(out.float().cpu(), ref) < 2e-2, (M, N, K)
            ref0 = a.float() @ (b.float().t() if tb else b.float())
            out0 = ops.gemm(a.to(DEV), b.to(DEV), trans_b=tb)
            assert rel_err(out0.float().cpu(), ref0) < 1e-2, (M, N, K)
    assert all(_lib.load().sow_get_switch(k) == -1 for k in (b"GEMM3S", b"GEMM3", b"GEMM4"))


def test_zero_state_and_reset_optimizer():
    from sow_amd import reset_optimizer
    torch.manual_seed(0)
    ps = [nn.Parameter(torch.randn(s, device=DEV)) for s in ((6, 4), (4, 3), (129,), (3, 5))]
    opt = torch.optim.AdamW([{"params": ps[:1]}, {"params": ps[1:]}], lr=1e-2)
    for _ in range(2):
        for p in ps:
            p.grad = torch.randn_like(p)
        opt.step()
    keep = opt.state[ps[0]]["exp_avg"].clone()
    reset_optimizer(opt, group_id=1)
    assert torch.equal(opt.state[ps[0]]["exp_avg"], keep)
    for p in ps[1:]:
        st = opt.state[p]
        assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
        assert float(st["step"]) == 0.0


def test_adamw_flat_matches_torch():
    from sow_amd import ops
    torch.manual_seed(1)
    p = torch.randn(1000, device=DEV)
    ref = nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.1)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.randn(1000, device=DEV)
        ref.grad = g.clone()
        opt.step()
        ops.adamw_flat_(p, g, m, v, lr=1e-2, weight_decay=0.1, step=step)
        assert rel_err(p.cpu(), ref.data.cpu()) < 1e-5
