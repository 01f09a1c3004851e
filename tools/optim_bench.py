#!/usr/bin/env python3
"""the optimizer step right after the hot path.
  f1  FactorAdamW (one fused kernel over the flat factor bucket) and its reset_state (one multi-tensor memset) against
      torch.optim.AdamW over the same 112 tensors (llama_60m, r = 50, bf16) and the reference's per-tensor reset loop;
  f4  TTAdam.step on dense fp32 parameters whose Adam moments live as tensor trains (reconstruct -> Adam -> decompose)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from sow_amd import SoWLinear
from sow_amd.dp import FactorBucket, factor_parameters
from sow_amd.optimizer import FactorAdamW, TTAdam

dev = torch.device("cuda:0")

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

shapes = [(512, 512)] * 32 + [(512, 1376)] * 16 + [(1376, 512)] * 8
model = nn.ModuleList([SoWLinear(i, o, bias=False, rank=50, init_method="normal", device=dev, dtype=torch.bfloat16) for i, o in shapes])
params = factor_parameters(model)
ref_params = [nn.Parameter(p.detach().clone()) for p in params]
for p in ref_params: p.grad = torch.randn_like(p) * 1e-2
bucket = FactorBucket(params)
bucket.flat_grad.normal_(std=1e-2)
opt = FactorAdamW(bucket, lr=1e-3, weight_decay=0.01)
ref = torch.optim.AdamW(ref_params, lr=1e-3, weight_decay=0.01)
print(f"factor group: {len(params)} tensors, {bucket.numel} elements (bf16)")
print(f"  FactorAdamW.step           {timeit(opt.step):8.1f} us")
print(f"  torch.optim.AdamW.step     {timeit(ref.step):8.1f} us  (foreach over {len(ref_params)} tensors)")
print(f"  FactorAdamW.reset_state    {timeit(opt.reset_state):8.1f} us")
def ref_reset():   # training_utils.py:257-277: per-tensor zeroing
    for p in ref_params:
        st = ref.state[p]
        st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"].zero_()
print(f"  per-tensor reset loop      {timeit(ref_reset):8.1f} us")

for (m, n, order, rank) in ((512, 512, 3, 8), (768, 768, 2, 16), (512, 1376, 3, 8)):
    p = nn.Parameter(torch.randn(m, n, device=dev) * 0.02)
    p.grad = torch.randn_like(p) * 1e-2
    o = TTAdam([{"params": [p], "ranks": [1] + [rank] * (order - 1) + [1]}], lr=1e-3)
    try:
        us = timeit(o.step, iters=10)
        print(f"TTAdam.step {m}x{n} order {order} rank {rank}: {us:9.1f} us")
    except Exception as e:   # parameter-group keys differ between revisions of the optimizer: report, do not hide
        print(f"TTAdam.step {m}x{n}: {type(e).__name__}: {e}")
