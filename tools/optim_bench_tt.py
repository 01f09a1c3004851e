#!/usr/bin/env python3
"""TTAdam.step over 16 parameters (8 x 512x512, 4 x 512x1376, 4 x 768x768; order 3, ranks [1, 8, 8, 1]) --
the batched entry point (sow_ttadam_batch: reconstruct -> Adam -> re-decompose for all parameters in one launch sequence)
against the per-parameter path (TensorTrain.to_matrix / from_matrix per moment).  Wall time per step (synchronised) and,
under `rocprofv3 --kernel-trace`, launches per step = kernel rows / steps.  usage: optim_bench_tt.py [batched|single] [steps]"""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import TTAdam

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
shapes = [(512, 512)] * 8 + [(512, 1376)] * 4 + [(768, 768)] * 4


def run(batched):
    torch.manual_seed(1)
    ps = [nn.Parameter(torch.randn(*s, device=dev) * 0.02) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-2
    TTAdam.batched = batched
    opt = TTAdam([{"params": ps, "ranks": [1, 8, 8, 1]}], lr=1e-3)
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if mode in ("both", "batched"):
    print(f"TTAdam.step, 16 parameters, batched (sow_ttadam_batch): {run(True):8.3f} ms/step", flush=True)
if mode in ("both", "single"):
    print(f"TTAdam.step, 16 parameters, per parameter:              {run(False):8.3f} ms/step", flush=True)
