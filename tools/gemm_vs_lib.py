#!/usr/bin/env python3
"""bf16 dense products at the finetune shapes of config 5 (M = 1024) and the llama_60m shapes: sow_gemm vs torch.matmul
(hipBLASLt / rocBLAS).  A yardstick only -- the product path never calls the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
dev = torch.device("cuda:0")

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (M, K, N) in ((1024, 4096, 4096), (1024, 4096, 11008), (1024, 11008, 4096), (4096, 4096, 4096), (32768, 512, 512), (32768, 512, 1376), (32768, 1376, 512)):
    a = torch.randn(M, K, device=dev).bfloat16()
    for tb in (False, True):
        b = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us_own = timeit(lambda: ops.gemm(a, b, trans_b=tb, out=out))
        bb = b.t() if tb else b
        us_lib = timeit(lambda: torch.matmul(a, bb, out=out))
        fl = 2.0 * M * N * K
        print(f"M={M:6d} K={K:6d} N={N:6d} {'NT' if tb else 'NN'}  sow_gemm {us_own:7.1f} us ({fl/us_own/1e6:6.0f} TF)   torch.matmul {us_lib:7.1f} us ({fl/us_lib/1e6:6.0f} TF)", flush=True)
