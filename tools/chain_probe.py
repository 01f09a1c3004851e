"""Kernel-level timing experiments on chain2 (run under rocprofv3 --kernel-trace; host timing is launch-bound).
usage: chain_probe.py FLAGS   (SOW_AMD_CHAIN2_DEBUG bits: 2 = no h_save store, 4 = no y stores)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SOW_AMD_CHAIN2_DEBUG"] = sys.argv[1] if len(sys.argv) > 1 else "0"
from sow_amd import ops
T = 32768
for (di, do) in ((512, 512), (512, 1376), (1376, 512)):
    xs = [torch.randn(T, di, device="cuda", dtype=torch.bfloat16) for _ in range(6)]
    dys = [torch.randn(T, do, device="cuda", dtype=torch.bfloat16) for _ in range(6)]
    A = (torch.randn(di, 50, device="cuda") * 0.04).bfloat16(); B = (torch.randn(50, do, device="cuda") * 0.04).bfloat16()
    for it in range(12):
        y, h = ops.sow_forward(xs[it % 6], A, B, None, None, None, 1.0)
        ops.sow_backward(dys[it % 6], xs[it % 6], h, A, B, None, None, 1.0, False)
    torch.cuda.synchronize()
