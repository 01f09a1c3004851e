#!/usr/bin/env python3
"""Time sow_gemm (bf16) on llama_60m dense-accumulator shapes.  usage: gemm_probe.py [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
T = 32768
for (N, K, tb) in ((512, 512, False), (512, 512, True), (1376, 512, False), (512, 1376, True), (512, 1376, False), (1376, 512, True)):
    a = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, b, trans_b=tb, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(a, b, trans_b=tb, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * T * N * K
    by = 2.0 * (T * K + T * N + N * K)
    print(f"N={N:5d} K={K:5d} {'NT' if tb else 'NN'}  {us:7.1f} us  {fl/us/1e6:7.1f} TF  {by/us/1e6:6.2f} TB/s", flush=True)
