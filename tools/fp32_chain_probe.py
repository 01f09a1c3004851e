"""Driver for rocprofv3 --kernel-trace / --pmc runs of the fp32 path at the north-star point (summarise with tools/kstats.py / tools/pmcstats.py).
usage: fp32_chain_probe.py [phases]   phases: 1 = data gradient only (default), 3 = data + weight gradients."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
T, d, r = 32768, 768, 50
x = [torch.randn(T, d, device="cuda") for _ in range(4)]
dy = [torch.randn(T, d, device="cuda") for _ in range(4)]
A = (torch.randn(d, r, device="cuda") * 0.04); B = (torch.randn(r, d, device="cuda") * 0.04)
for it in range(12):
    y, h = ops.sow_forward(x[it % 4], A, B, None, None, None, 1.0)
    ops.sow_backward(dy[it % 4], x[it % 4], h, A, B, None, None, 1.0, False, phases=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
torch.cuda.synchronize()
