"""Latency of accumulate(model) over the 56 SoWLinear layers of llama_60m (dense accumulator branch as after
prepare_sow, normal_QR re-init), bf16."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn
from sow_amd import SoWLinear, accumulate
dev = "cuda:0"
shapes = [(512, 512)] * 32 + [(512, 1376)] * 16 + [(1376, 512)] * 8
model = nn.ModuleList([SoWLinear(i, o, bias=False, rank=50, init_method="normal_QR", device=dev, dtype=torch.bfloat16) for i, o in shapes])
for m in model:
    m.virtual_rank = min(m.in_features, m.out_features)   # prepare.py:120
for _ in range(2):
    accumulate(model)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    accumulate(model)
torch.cuda.synchronize()
print(f"accumulate(model), 56 layers: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(n):
    for m in model:
        m.accumulate()
torch.cuda.synchronize()
print(f"sequential per-layer loop  : {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
