#!/usr/bin/env python3
"""sow_gemm (bf16) at short-T finetune shapes (configs 4-5): streaming kernel vs 128x128 kernel (SOW_AMD_FORCE_GEMM_V1=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, tb) in ((1024, 11008, 4096, False), (1024, 4096, 11008, True), (1024, 4096, 4096, False), (1024, 4096, 4096, True),
                      (8192, 3072, 768, False), (8192, 768, 3072, True), (2048, 11008, 4096, False), (4096, 4096, 4096, False)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): ops.gemm(a, b, trans_b=tb, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): ops.gemm(a, b, trans_b=tb, out=out)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); g.replay(); e1.record(s); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"M={M:5d} N={N:5d} K={K:5d} {'NT' if tb else 'NN'}  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
