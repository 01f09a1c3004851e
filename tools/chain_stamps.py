#!/usr/bin/env python3
"""In-kernel timeline of the persistent chain kernel (build `make STAMPS=1`, run with SOW_AMD_LIB=sow_amd/lib/libsow_amd_stamps.so).

One grouped forward (q, k, v: 3 x 512 -> 512, T = 32768) and one gate / up forward (2 x 512 -> 1376): per block the
s_memrealtime stamps (10 ns) of wave 0 at  0 block start | 1 first X stage + factor chunk landed | 2 end of phase 1 |
3 end of hand-off | 4 end of the phase-2 loop | 5 last flush issued | 6 end-of-block barrier.  Prints mean segment lengths
per round (round = blk // 512) and the spread of the start times, i.e. how far the resident workgroups drift apart."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import _lib, ops  # noqa: E402

lib = _lib.load()
set_stamps = lib.sow_debug_set_stamps
set_stamps.argtypes = [ctypes.c_void_p]
T, R = 32768, 50
dev = torch.device("cuda:0")
for name, shapes, bwd in (("fwd q+k+v", [(512, 512)] * 3, False), ("fwd gate+up", [(512, 1376)] * 2, False),
                          ("fwd down", [(1376, 512)], False), ("bwd q+k+v", [(512, 512)] * 3, True)):
    calls = []
    for (di, do) in shapes:
        x = torch.randn(T, di, device=dev).bfloat16()
        dy = torch.randn(T, do, device=dev).bfloat16()
        A = (torch.randn(di, R, device=dev) * 0.04).bfloat16()
        B = (torch.randn(R, do, device=dev) * 0.04).bfloat16()
        calls.append(ops.LayerCall(x, A, B, dy2=dy, dx=torch.empty_like(x), out=(torch.zeros_like(A), torch.zeros_like(B), None)))
    grp = ops.LayerGroup(calls)
    nblk = len(shapes) * (T // 64)
    buf = torch.zeros(nblk, 16, dtype=torch.int64, device=dev)
    run = (lambda: grp.backward(_lib.BWD_DATA)) if bwd else grp.forward
    if bwd:
        grp.forward()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    set_stamps(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    set_stamps(None)
    raw = buf.cpu().double() * 0.01    # us
    st = raw[:, :8] - raw[:, 0].min()
    waits = raw[:, 8:13]
    print(f"== {name}: kernel span {float(st[:, 6].max()):.1f} us (first block start -> last block end)")
    for rnd in range(len(shapes)):
        s = st[rnd * 512:(rnd + 1) * 512]
        seg = [float((s[:, i + 1] - s[:, i]).mean()) for i in range(6)]
        print(f"  round {rnd}: start {float(s[:, 0].mean()):6.2f} +- {float(s[:, 0].std()):4.2f} (min {float(s[:, 0].min()):6.2f} max {float(s[:, 0].max()):6.2f}) | "
              f"first-load {seg[0]:5.2f} | phase1 {seg[1]:5.2f} | hand-off {seg[2]:5.2f} | phase2 {seg[3]:5.2f} | last-flush {seg[4]:5.2f} | end-barrier {seg[5]:5.2f} | "
              f"block {float((s[:, 6] - s[:, 0]).mean()):5.2f} us")
        wv = waits[rnd * 512:(rnd + 1) * 512].mean(0)
        print(f"           waits per block: loader wave in its DMA wait {float(wv[0]):5.2f} | loader at barriers {float(wv[1]):5.2f} | "
              f"compute wave 0: X wait {float(wv[2]):5.2f}, phase-1 barriers {float(wv[3]):5.2f}, phase-2 barriers {float(wv[4]):5.2f} us")
