"""Workload for the SQ-counter pass over the streaming GEMM kernels (gemm3, gemm3s, gemm2h): see profiles/r01_pmc_sq_gemm3_gemm2h.txt."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
dev = torch.device("cuda:0")
# gemm3 (long K), gemm3s (short M), gemm2h via the dense layer (512 -> 512)
for (M, K, N, tb) in ((4096, 4096, 4096, True), (4096, 4096, 4096, False), (1024, 4096, 4096, True), (1024, 11008, 4096, True)):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5): ops.gemm(a, b, trans_b=tb, out=out)
T, di, do, r = 32768, 512, 512, 50
x = torch.randn(T, di, device=dev).bfloat16(); dy = torch.randn(T, do, device=dev).bfloat16()
A = torch.linalg.qr(torch.randn(di, r, device=dev) * 0.02)[0].bfloat16().contiguous()
B = (torch.randn(r, do, device=dev) * 0.02).bfloat16()
W = (torch.randn(di, do, device=dev) * 0.02).bfloat16()
for _ in range(5):
    y, h = ops.sow_forward(x, A, B, W, None, None, 1.0)
    ops.sow_backward(dy, x, h, A, B, W, None, 1.0, False)
torch.cuda.synchronize()
