#!/usr/bin/env python3
"""gemm4 (anti-phase wave groups, 64-wide K-tiles) against the round-2 kernels (gemm2 / gemm3 / gemm3s by shape) and
against torch.matmul (hipBLASLt; a yardstick only, never on the product path): correctness vs an fp32 product of the same
bf16 operands, then time per call (HIP events, interleaved rounds in one process).
usage: gemm4_probe.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import _lib, ops

dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def timeit(fn, iters=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


SHAPES = [(32768, 512, 512), (32768, 512, 1376), (32768, 1376, 512), (4096, 4096, 4096), (1024, 4096, 4096),
          (1024, 4096, 11008), (1024, 11008, 4096), (8192, 768, 3072), (20000, 520, 520), (2100, 2056, 4500), (300, 64, 72)]
if quick:
    SHAPES = SHAPES[:4] + SHAPES[-3:]
for (M, K, N) in SHAPES:
    a = torch.randn(M, K, device=dev).bfloat16()
    for tb in (False, True):
        b = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).bfloat16()
        bb = b.t() if tb else b
        ref = a.float() @ bb.float()
        out_old = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out_new = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out_lib = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def f_old():
            with _lib.switch(GEMM4=0):
                ops.gemm(a, b, trans_b=tb, out=out_old)

        def f_new():
            with _lib.switch(GEMM4=1):
                ops.gemm(a, b, trans_b=tb, out=out_new)

        def f_lib():
            torch.matmul(a, bb, out=out_lib)

        f_old(), f_new(), f_lib()
        torch.cuda.synchronize()
        e_old, e_new = rel(out_old.float(), ref), rel(out_new.float(), ref)
        # time the raw calls with the switch set once (the context manager costs host time)
        ts = {}
        for name, val, fn in (("old", 0, lambda: ops.gemm(a, b, trans_b=tb, out=out_old)),
                              ("new", 1, lambda: ops.gemm(a, b, trans_b=tb, out=out_new))):
            with _lib.switch(GEMM4=val):
                for _ in range(3):
                    fn()
                ts[name] = min(timeit(fn) for _ in range(3))
        for _ in range(3):
            f_lib()
        ts["lib"] = min(timeit(f_lib) for _ in range(3))
        fl = 2.0 * M * N * K
        print(f"M={M:6d} K={K:6d} N={N:6d} {'NT' if tb else 'NN'}  err old {e_old:.1e} new {e_new:.1e} | "
              f"old {ts['old']:7.1f} us ({fl / ts['old'] / 1e6:5.0f} TF)  gemm4 {ts['new']:7.1f} us ({fl / ts['new'] / 1e6:5.0f} TF)  "
              f"torch.matmul {ts['lib']:7.1f} us ({fl / ts['lib'] / 1e6:5.0f} TF)", flush=True)
        assert e_new < 1e-2, "gemm4 result is wrong"
