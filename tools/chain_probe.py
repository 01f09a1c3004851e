"""Per-shape kernel timing of one SoWLinear fwd+bwd (run under rocprofv3 --kernel-trace and summarise with
tools/kstats.py; host-side timing is launch-bound for kernels this short)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
T = 32768
dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.bfloat16
for (di, do) in ((512, 512), (512, 1376), (1376, 512), (768, 768)):
    xs = [torch.randn(T, di, device="cuda").to(dtype) for _ in range(6)]
    dys = [torch.randn(T, do, device="cuda").to(dtype) for _ in range(6)]
    A = (torch.randn(di, 50, device="cuda") * 0.04).to(dtype); B = (torch.randn(50, do, device="cuda") * 0.04).to(dtype)
    for it in range(12):
        y, h = ops.sow_forward(xs[it % 6], A, B, None, None, None, 1.0)
        ops.sow_backward(dys[it % 6], xs[it % 6], h, A, B, None, None, 1.0, False)
    torch.cuda.synchronize()
