#!/usr/bin/env python3
"""A/B of the dense-accumulator layer: in-kernel projection (gemm2h, one launch per pass) vs H-only chain + K-extended
GEMM (switch NO_FUSED_H).  Graph-replayed fwd+bwd(data) over 4 rotating buffer sets; bf16, T = 32768, r = 50."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import _lib, ops
T, r = 32768, 50
dev = torch.device("cuda:0")
for (di, do) in ((512, 512), (512, 1376), (1376, 512), (768, 768)):
    xs = [torch.randn(T, di, device=dev).bfloat16() for _ in range(4)]
    dys = [torch.randn(T, do, device=dev).bfloat16() for _ in range(4)]
    A = torch.linalg.qr(torch.randn(di, r, device=dev) * 0.02)[0].bfloat16().contiguous()
    B = (torch.randn(r, do, device=dev) * 0.02).bfloat16()
    W = (torch.randn(di, do, device=dev) * 0.02).bfloat16()
    res = {}
    for mode in ("fused", "two-launch"):
        _lib.load().sow_set_switch(b"NO_FUSED_H", 1 if mode == "two-launch" else -1)
        def step():
            for i in range(4):
                _, h = ops.sow_forward(xs[i], A, B, W, None, None, 1.0)
                ops.sow_backward(dys[i], xs[i], h, A, B, W, None, 1.0, False)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(20): g.replay()
            e1.record(s); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 20 / 4 * 1e3
    print(f"{di:5d} -> {do:5d}  fused {res['fused']:7.1f} us   two-launch {res['two-launch']:7.1f} us", flush=True)
_lib.load().sow_set_switch(b"NO_FUSED_H", -1)
