"""Layer-level sweep of SURVEY.md section 8(d): SoWLinear fwd+bwd at the seven (T, d_in, d_out, r) points, bf16 and
fp32, empty and dense accumulator, plus the periodic accumulate() latency at the cfg2 / cfg4 / cfg5 layer shapes.
Graph-replayed over rotating buffers; GFLOP/s uses 6*T*r*(d_in+d_out) (+ 4*T*d_in*d_out with a dense accumulator)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sow_amd import ops, SoWLinear

SHAPES = [(64, 256, 256, 8), (32768, 768, 768, 50), (32768, 512, 512, 50), (32768, 512, 1376, 50), (32768, 1376, 512, 50),
          (8192, 768, 3072, 8), (1024, 4096, 11008, 8)]

def bench(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us

print(f"{'T':>6} {'d_in':>5} {'d_out':>6} {'r':>3} {'dtype':>5} {'acc':>5} | {'us':>8} {'TFLOP/s':>8} {'alg TB/s':>8}")
for (T, di, do, r) in SHAPES:
    for dtype in (torch.bfloat16, torch.float32):
        es = 2 if dtype == torch.bfloat16 else 4
        nb = 4 if T * max(di, do) * es < 2e8 else 2
        xs = [torch.randn(T, di, device="cuda").to(dtype) for _ in range(nb)]
        dys = [torch.randn(T, do, device="cuda").to(dtype) for _ in range(nb)]
        A = torch.linalg.qr(torch.randn(di, r, device="cuda") * 0.02)[0].to(dtype).contiguous()
        B = (torch.randn(r, do, device="cuda") * 0.02).to(dtype)
        for acc in ("none", "dense"):
            W = (torch.randn(di, do, device="cuda") * 0.02).to(dtype) if acc == "dense" else None
            def step():
                for i in range(nb):
                    _, h = ops.sow_forward(xs[i], A, B, W, None, None, 1.0)
                    ops.sow_backward(dys[i], xs[i], h, A, B, W, None, 1.0, False)
            us = bench(step) / nb
            flops = 6 * T * r * (di + do) + (4 * T * di * do if acc == "dense" else 0)
            nbytes = T * (3 * di + 2 * do) * es + 2 * T * r * es
            print(f"{T:6d} {di:5d} {do:6d} {r:3d} {'bf16' if es == 2 else 'f32':>5} {acc:>5} | {us:8.1f} {flops/us/1e6:8.1f} {nbytes/us/1e6:8.2f}", flush=True)

print("\naccumulate() latency per layer (dense accumulator branch as after prepare_sow, normal_QR re-init), ms")
for name, (di, do, r, dtype) in {"cfg2 llama_60m attn 512x512 r50 bf16": (512, 512, 50, torch.bfloat16),
                                  "cfg2 llama_60m mlp 512x1376 r50 bf16": (512, 1376, 50, torch.bfloat16),
                                  "cfg4 roberta 768x3072 r8 f32": (768, 3072, 8, torch.float32),
                                  "cfg5 llama_7b 4096x11008 r8 bf16": (4096, 11008, 8, torch.bfloat16)}.items():
    layer = SoWLinear(di, do, bias=False, rank=r, init_method="normal_QR", device="cuda", dtype=dtype)
    layer.virtual_rank = min(di, do)
    layer.accumulate(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        layer.upscale_weights[0].data.normal_(0, 0.02)
        layer.accumulate()
    torch.cuda.synchronize()
    print(f"  {name:42s} {(time.perf_counter() - t0) / 5 * 1e3:8.2f}")
