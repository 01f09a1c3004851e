import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import SoWLinear
dev = "cuda:0"
sow = SoWLinear(512, 512, bias=False, rank=50, init_method="normal", device=dev, dtype=torch.bfloat16)
x = torch.randn(64, 512, device=dev, dtype=torch.bfloat16, requires_grad=True)
dy = torch.randn(64, 512, device=dev, dtype=torch.bfloat16)
for _ in range(20): sow(x).backward(dy)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): sow(x).backward(dy)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
