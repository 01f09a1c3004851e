"""North-star point of BASELINE.json: one SoWLinear fwd+bwd at r=50, d_in=d_out=768, T=32768 (bf16 and fp32).
Graph-replayed over 4 rotating buffer sets; reports GFLOP/s with the algorithmic count 6*T*r*(d_in+d_out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops
T, d, r = 32768, 768, 50
for dtype, peak in ((torch.bfloat16, 2500e12), (torch.float32, 157.3e12)):
    es = 2 if dtype == torch.bfloat16 else 4
    xs = [torch.randn(T, d, device="cuda").to(dtype) for _ in range(4)]
    dys = [torch.randn(T, d, device="cuda").to(dtype) for _ in range(4)]
    A = torch.linalg.qr(torch.randn(d, r, device="cuda") * 0.02)[0].to(dtype).contiguous()
    B = (torch.randn(r, d, device="cuda") * 0.02).to(dtype)
    def step():
        for i in range(4):
            _, h = ops.sow_forward(xs[i], A, B, None, None, None, 1.0)
            ops.sow_backward(dys[i], xs[i], h, A, B, None, None, 1.0, False)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): g.replay()
        e1.record(s); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 / 4 * 1e3
    flops = 6 * T * r * 2 * d
    nbytes = T * (3 * d + 2 * d) * es + 2 * T * r * es
    print(f"{str(dtype):16s} fwd+bwd {us:7.1f} us  {flops/us/1e6:8.1f} TFLOP/s = {flops/us*1e6/peak*100:5.1f}% of the {peak/1e12:.0f} TF MFMA peak | "
          f"{nbytes/us/1e6:6.2f} TB/s algorithmic = {nbytes/us*1e6/8e12*100:4.1f}% of 8 TB/s", flush=True)
