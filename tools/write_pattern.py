#!/usr/bin/env python3
"""Does HBM care about the width of the contiguous pieces a streaming kernel writes?  A [T, D] bf16 buffer filled
column slice by column slice (one torch kernel per slice: every row gets a `w`-column = 2w-byte piece, row pitch 2 D
bytes), for w = 64 (what one phase-2 step of the chain kernel writes), 128, 256, D; 8 rotating buffers, HIP-graph replay."""
import torch

dev = torch.device("cuda:0")
T, N = 32768, 8


def timed(fn, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for D in (512, 1376, 1408, 2048):
    bufs = [torch.empty(T, D, device=dev, dtype=torch.bfloat16) for _ in range(N)]
    src = [torch.randn(T, D, device=dev, dtype=torch.bfloat16) for _ in range(N)]
    mb = T * D * 2 / 1e6
    line = f"D={D:5d} ({mb:5.1f} MB):"
    for w in (64, 128, 256, D):
        def fill():
            for b in bufs:
                for c0 in range(0, D, w):
                    b[:, c0:c0 + w].fill_(1.0)
        def copy():
            for b, s_ in zip(bufs, src):
                for c0 in range(0, D, w):
                    b[:, c0:c0 + w].copy_(s_[:, c0:c0 + w])
        tf, tc = timed(fill) / N, timed(copy) / N
        nk = (D + w - 1) // w
        line += f" | w={w:4d} ({nk:2d} kernels) fill {tf:6.1f} us {mb / tf * 1e3:5.0f} GB/s copy {tc:6.1f} us {2 * mb / tc * 1e3:5.0f} GB/s"
    print(line, flush=True)
    del bufs, src
    torch.cuda.empty_cache()
