"""Host-side cost of one SoWLinear forward + backward through torch.autograd (eager, no graph): tiny shapes so that the
GPU time is negligible.  Compares with nn.Linear on the same shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import SoWLinear
dev = "cuda:0"
for dtype in (torch.bfloat16, torch.float32):
    sow = SoWLinear(512, 512, bias=False, rank=50, init_method="normal", device=dev, dtype=dtype)
    lin = torch.nn.Linear(512, 512, bias=False, device=dev, dtype=dtype)
    x = torch.randn(64, 512, device=dev, dtype=dtype, requires_grad=True)
    dy = torch.randn(64, 512, device=dev, dtype=dtype)
    for name, mod in (("SoWLinear", sow), ("nn.Linear", lin)):
        for _ in range(20):
            mod(x).backward(dy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            mod(x).backward(dy)
        torch.cuda.synchronize()
        print(f"{name:10s} {str(dtype):15s} {1e6 * (time.perf_counter() - t0) / n:7.1f} us per fwd+bwd call (host-bound)")
