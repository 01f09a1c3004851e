#!/usr/bin/env python3
"""Practical HBM ceilings on this box with plain torch kernels (not the product path -- a yardstick for the roofline
fractions): write-only (fill), read-only (sum), copy, over rotating buffers larger than the 256 MiB Infinity Cache;
and the launch-to-launch cost of a HIP graph of small kernels."""
import torch

dev = torch.device("cuda:0")
N = 8


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
    return e0.elapsed_time(e1) / reps * 1e3   # us per graph


for mb in (32, 64, 128, 512):
    n = mb * 1024 * 1024 // 2
    a = [torch.randn(n, device=dev, dtype=torch.bfloat16) for _ in range(N)]
    b = [torch.empty(n, device=dev, dtype=torch.bfloat16) for _ in range(N)]
    out = torch.empty(N, device=dev, dtype=torch.float32)
    t_fill = timed(lambda: [x.zero_() for x in b]) / N
    t_copy = timed(lambda: [y.copy_(x) for x, y in zip(a, b)]) / N
    t_read = timed(lambda: [torch.sum(x.view(1, -1), dim=(1,), dtype=torch.float32, out=out[i:i + 1]) for i, x in enumerate(a)]) / N
    print(f"{mb:4d} MB/buffer: fill {t_fill:7.1f} us {mb * 1.048576 / t_fill * 1e3:6.0f} GB/s | copy {t_copy:7.1f} us {2 * mb * 1.048576 / t_copy * 1e3:6.0f} GB/s (r+w) | "
          f"read {t_read:7.1f} us {mb * 1.048576 / t_read * 1e3:6.0f} GB/s", flush=True)
    del a, b
    torch.cuda.empty_cache()
tiny = [torch.zeros(256, device=dev) for _ in range(64)]
print(f"graph of 64 tiny kernels: {timed(lambda: [x.add_(1.0) for x in tiny]) / 64:.2f} us per launch")
