#!/usr/bin/env python3
"""Idle time BETWEEN consecutive chain-kernel launches of one HIP graph (build `make STAMPS=1`, run with
SOW_AMD_LIB=sow_amd/lib/libsow_amd_stamps.so).

Eight launches back to back in one captured graph, every launch with its own stamp buffer (the pointer is a kernel
argument, fixed at capture); s_memrealtime is one 100-MHz clock for the whole device, so
    gap(i) = first block start of launch i+1  -  last block end of launch i
is the time the chip spends between two kernels: end-of-kernel cache write-back, the command processor's barrier, the
dispatch of the next grid.  Printed beside the launch-to-launch period of the replay (HIP events)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
set_stamps = lib.sow_debug_set_stamps
set_stamps.argtypes = [ctypes.c_void_p]
T, R, NL = 32768, 50, 8
dev = torch.device("cuda:0")
sw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
for k, v in sw.items():
    lib.sow_set_switch(k.encode(), int(v))

for name, shapes, bwd in (("fwd o 1x(512->512)", [(512, 512)], False), ("fwd q+k+v 3x(512->512)", [(512, 512)] * 3, False),
                          ("fwd gate+up 2x(512->1376)", [(512, 1376)] * 2, False), ("bwd q+k+v", [(512, 512)] * 3, True)):
    groups, bufs = [], []
    for _ in range(NL):
        calls = []
        for (di, do) in shapes:
            x = torch.randn(T, di, device=dev).bfloat16()
            dy = torch.randn(T, do, device=dev).bfloat16()
            A = (torch.randn(di, R, device=dev) * 0.04).bfloat16()
            B = (torch.randn(R, do, device=dev) * 0.04).bfloat16()
            calls.append(ops.LayerCall(x, A, B, dy2=dy, dx=torch.empty_like(x), out=(torch.zeros_like(A), torch.zeros_like(B), None)))
        groups.append(ops.LayerGroup(calls))
        bufs.append(torch.zeros(len(shapes) * (T // 64), 16, dtype=torch.int64, device=dev))

    def run_all():
        for g, b in zip(groups, bufs):
            set_stamps(b.data_ptr())
            g.backward(_lib.BWD_DATA) if bwd else g.forward()
        set_stamps(None)

    if bwd:
        for g in groups:
            g.forward()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run_all()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            run_all()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            graph.replay()
        e1.record(s)
        torch.cuda.synchronize()
    period = e0.elapsed_time(e1) / 5 / NL * 1e3
    st = [b.cpu().double() * 0.01 for b in bufs]   # us
    spans = [float(x[:, 6].max() - x[:, 0].min()) for x in st]
    gaps = [float(st[i + 1][:, 0].min() - st[i][:, 6].max()) for i in range(NL - 1)]
    skew = [float(x[:512, 0].max() - x[:512, 0].min()) for x in st]
    print(f"== {name} {sw}: launch-to-launch {period:.1f} us | in-kernel span {sum(spans) / NL:.1f} us | "
          f"gap between kernels {sum(gaps) / len(gaps):.2f} us (min {min(gaps):.2f} max {max(gaps):.2f}) | "
          f"first-round start skew {sum(skew) / NL:.2f} us", flush=True)
