#!/usr/bin/env python3
"""Shape sweep of the bf16 streaming chain kernel (chain2): forward and backward-data launches timed by HIP-graph
replay over rotating buffers (8 launches per graph, distinct x / dY each, so nothing is served from the Infinity Cache).
Prints us per launch, the algorithmic GB/s  T*(D1 + D2 + r)*2 / t  and us per 64-wide K stage / output slice.

    python tools/chain_sweep.py [fwd|bwd|both] [d_in,d_out ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import _lib, ops  # noqa: E402

T, R, NB = 32768, 50, 8
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("fwd", "bwd", "both") else "both"
DT = torch.float32 if "f32" in sys.argv else torch.bfloat16
ES = 4 if DT == torch.float32 else 2
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:] if "," in a] or \
    [(512, 512), (1024, 512), (1280, 512), (1376, 512), (1408, 512), (1536, 512), (2048, 512), (512, 1024), (512, 1376),
     (512, 1408), (512, 2048), (768, 768)]


def timed(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / NB * 1e3


for (di, do) in shapes:
    xs = [torch.randn(T, di, device=dev).to(DT) for _ in range(NB)]
    dys = [torch.randn(T, do, device=dev).to(DT) for _ in range(NB)]
    A = torch.linalg.qr(torch.randn(di, R, device=dev) * 0.02)[0].to(DT).contiguous()
    B = (torch.randn(R, do, device=dev) * 0.02).to(DT)
    hs = [ops.sow_forward(x, A, B, None, None, None, 1.0)[1] for x in xs]
    dxs = [torch.empty_like(x) for x in xs]
    wss = [torch.empty(ops.workspace_bytes(T, di, do, R, 0, _lib.ACC_NONE, DT) + 256, dtype=torch.uint8, device=dev)
           for _ in range(NB)]
    dA, dB = torch.empty_like(A), torch.empty_like(B)
    line = f"{di:5d} -> {do:5d}"
    nbytes = T * (di + do + R) * ES
    if mode in ("fwd", "both"):
        us = timed(lambda: [ops.sow_forward(x, A, B, None, None, None, 1.0) for x in xs])
        line += f" | fwd {us:6.1f} us {nbytes / us / 1e3:6.0f} GB/s  ({us / ((di + 63) // 64 + (do + 63) // 64):5.2f} us/unit)"
    if mode in ("bwd", "both"):
        us = timed(lambda: [ops.sow_backward(dys[i], xs[i], hs[i], A, B, None, None, 1.0, False, out=(dA, dB, None),
                                             phases=_lib.BWD_DATA, dx=dxs[i], workspace=wss[i]) for i in range(NB)])
        line += f" | bwd-data {us:6.1f} us {nbytes / us / 1e3:6.0f} GB/s"
        us = timed(lambda: [ops.sow_backward(dys[i], xs[i], hs[i], A, B, None, None, 1.0, False, out=(dA, dB, None),
                                             phases=_lib.BWD_WEIGHTS_PARTIAL, dx=dxs[i], workspace=wss[i]) for i in range(NB)])
        line += f" | tn-partial {us:6.1f} us {T * (di + do + 128) * ES / us / 1e3:6.0f} GB/s"
    print(line, flush=True)
    del xs, dys, hs, dxs, wss
    torch.cuda.empty_cache()
