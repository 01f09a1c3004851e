"""Race screen: the kernels are deterministic by construction (fixed reduction orders, no float atomics), so repeated
launches on the same inputs must be BIT-identical; a difference means a missing wait / barrier somewhere in the
LDS-DMA pipelines.  Runs every streaming path (chain2, chain2f, skinny-TN, streaming GEMM with and without the in-kernel projection, short-T split) many times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for dtype in (torch.bfloat16, torch.float32):
    for (T, di, do, r, acc) in ((32768, 512, 512, 50, None), (32768, 512, 1376, 50, None), (32768, 1376, 512, 50, None),
                                (32768, 512, 1376, 50, "dense"), (16500, 776, 1000, 50, "dense"),
                                (32768, 512, 512, 50, "dense"), (20500, 776, 264, 34, "dense"), (32768, 1376, 512, 46, "dense"), (4100, 776, 1000, 62, None),
                                (1000, 4096, 1024, 8, None), (1024, 2048, 4096, 8, "dense"), (8192, 768, 3072, 8, None)):
        g = torch.Generator(device=dev).manual_seed(T + di)
        x = torch.randn(T, di, device=dev, generator=g).to(dtype)
        dy = torch.randn(T, do, device=dev, generator=g).to(dtype)
        A = (torch.randn(di, r, device=dev, generator=g) * 0.05).to(dtype)
        B = (torch.randn(r, do, device=dev, generator=g) * 0.05).to(dtype)
        W = (torch.randn(di, do, device=dev, generator=g) * 0.02).to(dtype) if acc else None
        ref = None
        for it in range(reps):
            y, h = ops.sow_forward(x, A, B, W, None, None, 0.5)
            dx, dA, dB, _ = ops.sow_backward(dy, x, h, A, B, W, None, 0.5, False)
            cur = [t.clone() for t in (y, h.view(T, -1)[:, :r], dx, dA, dB)]
            if ref is None:
                ref = cur
            else:
                for name, a, b in zip(("y", "h", "dx", "dA", "dB"), ref, cur):
                    if not torch.equal(a, b):
                        bad += 1
                        print(f"MISMATCH {dtype} T={T} {di}x{do} r={r} acc={acc} iter {it}: {name} differs in {(a != b).sum().item()} elements")
        torch.cuda.synchronize()
        print(f"ok    {str(dtype):15s} T={T:6d} {di:5d}x{do:<5d} r={r:2d} acc={acc}", flush=True)
print("race screen:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
