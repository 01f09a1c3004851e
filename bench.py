#!/usr/bin/env python3
"""Benchmark of the SoW hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

Workload (BASELINE.json configs[1]): the 56 SoWLinear layers of llama_60m (`--architecture sow`,
rank 50; 32 x (512->512), 16 x (512->1376), 8 x (1376->512), reference scripts/configs/llama_60m.json)
at batch 128 x seq 256 = 32768 tokens per GPU, bf16.  One step = forward of all 56 layers in model
order, backward of all 56 in reverse order (every layer has its own x / dY buffers, so nothing is
re-read from cache that a real model would not have; the token-slab partial sums of the weight gradients
are reduced in one batched launch at the end of backward -- `--reduce layer` does it per layer, 3 % slower,
bit-identical gradients), and -- for N > 1 -- ONE RCCL all-reduce of the flat factor-gradient bucket.  Inputs are resident in HBM before the timed region.  The step is
captured in a HIP graph after warm-up (no host work in the timed region).

Prints ONE JSON line.  `value` = tokens/s over all ranks (T * N / step time); `gflops` = the
algorithmic 6*T*r*(d_in+d_out) count per second.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# HBM bytes per launch of the roofline kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
# WRITE_SIZE, see profiles/); None until collected for the current kernel version.
TRAFFIC_BYTES_PER_LAUNCH = 97.9e6  # profiles/r01_pmc_hbm_fetch_write_v3.txt (bf16, chain2_kernel<false>)
LLAMA_60M = dict(hidden=512, inter=1376, layers=8)


def layer_shapes():
    h, i = LLAMA_60M["hidden"], LLAMA_60M["inter"]
    block = [(h, h)] * 4 + [(h, i), (h, i), (i, h)]  # q k v o gate up down
    return block * LLAMA_60M["layers"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rank", type=int, default=50)
    ap.add_argument("--tokens", type=int, default=128 * 256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--acc", default="none", choices=["none", "dense"], help="accumulator state of the layers")
    ap.add_argument("--mode", default="stack", choices=["stack", "train"],
                    help="stack: the 56-layer SoWLinear hot path (headline); train: a full llama_60m training step "
                         "through the module-swap surface (prepare_sow, autograd, AdamW, accumulate), secondary figure")
    ap.add_argument("--accumulate-every", type=int, default=4, help="--mode train: SoW accumulation period in steps")
    ap.add_argument("--fused-factors", action="store_true",
                    help="--mode train, 1 GPU: factor group in a FactorBucket (gradients written straight into one flat buffer, "
                         "one batched reduction, one fused AdamW kernel) instead of torch.optim.AdamW's second param group")
    ap.add_argument("--reduce", choices=["batch", "layer"], default="batch",
                    help="weight-gradient reduction: one 5-us launch per layer, or deferred and batched into one launch at the end "
                         "of backward (sow_reduce_batch: same arithmetic, bit-identical gradients)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: weight-gradient kernels on a side stream (measured: -4 %%; only their small reduction on a side stream: -16 %% -- cross-stream edges of a HIP graph cost more than the 5-us kernel they hide)")
    return ap.parse_args()


class Stack:
    """The 56-layer SoWLinear stack with resident synthetic inputs."""

    def __init__(self, shapes, T, r, dtype, device, acc, streams=1, reduce="layer"):
        from sow_amd import ops
        from sow_amd.dp import FactorBucket
        self.shapes, self.T, self.r, self.dtype, self.acc = shapes, T, r, dtype, acc
        self.streams = streams
        self.deferred = ops.DeferredReduce() if reduce == "batch" else None
        self.side = torch.cuda.Stream(device=device) if streams > 1 else None
        kind = 2 if acc == "dense" else 0
        # per-layer workspaces (dh + slab partials): the split backward keeps them alive across two streams
        self.ws = [torch.empty(ops.workspace_bytes(T, di, do, r, 0, kind, dtype) + 256, dtype=torch.uint8, device=device)
                   for di, do in shapes]
        self.dx = {di: [torch.empty(T, di, dtype=dtype, device=device) for _ in range(4)] for di in {s[0] for s in shapes}}
        g = torch.Generator(device=device)
        self.x, self.dy, self.A, self.B, self.W = [], [], [], [], []
        params = []
        for li, (d_in, d_out) in enumerate(shapes):
            g.manual_seed(1234 + li)
            self.x.append(torch.randn(T, d_in, generator=g, device=device, dtype=torch.float32).to(dtype))
            self.dy.append(torch.randn(T, d_out, generator=g, device=device, dtype=torch.float32).to(dtype))
            a = torch.linalg.qr(torch.randn(d_in, r, generator=g, device=device) * 0.02)[0]  # orthonormal columns
            b = torch.randn(r, d_out, generator=g, device=device) * 0.02
            self.A.append(torch.nn.Parameter(a.to(dtype).contiguous()))
            self.B.append(torch.nn.Parameter(b.to(dtype).contiguous()))
            params += [self.A[-1], self.B[-1]]
            self.W.append((torch.randn(d_in, d_out, generator=g, device=device) * 0.02).to(dtype) if acc == "dense" else None)
        self.bucket = FactorBucket(params)  # grads are views into one flat buffer
        self.h = [None] * len(shapes)

    def forward_all(self):
        from sow_amd import ops
        for li in range(len(self.shapes)):
            _, self.h[li] = ops.sow_forward(self.x[li], self.A[li].data, self.B[li].data, self.W[li], None, None, 1.0)

    def backward_all(self):
        """Backward in reverse layer order.  The data-gradient kernel of a layer (dX: what the previous
        layer's backward waits for in a real model) stays on the main stream; with --streams 2 the
        weight-gradient kernels (skinny-TN + reduce), which nothing in backprop depends on, run on a side
        stream ordered by an event -- the true dependency structure of a training step."""
        from sow_amd import _lib, ops
        main = torch.cuda.current_stream()
        if self.side is not None:
            self.side.wait_stream(main)
        for n, li in enumerate(reversed(range(len(self.shapes)))):
            args = (self.dy[li], self.x[li], self.h[li], self.A[li].data, self.B[li].data, self.W[li], None, 1.0, False)
            kw = dict(out=(self.A[li].grad, self.B[li].grad, None), grad_beta=0.0, workspace=self.ws[li],
                      dx=self.dx[self.shapes[li][0]][n % 4])
            if self.deferred is not None:
                ops.sow_backward(*args, phases=_lib.BWD_DATA | _lib.BWD_WEIGHTS_PARTIAL, **kw)
                self.deferred.add(self.x[li], self.B[li].data, kw["out"], 0.0, self.ws[li], self.W[li], None)
            elif self.side is None:
                ops.sow_backward(*args, **kw)
            else:
                ops.sow_backward(*args, phases=_lib.BWD_DATA, **kw)
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ev)
                    ops.sow_backward(*args, phases=_lib.BWD_WEIGHTS, **kw)
        if self.deferred is not None:
            self.deferred.run()
        if self.side is not None:
            main.wait_stream(self.side)

    def step(self):
        self.forward_all()
        self.backward_all()


BWD_LABEL = {"layer": "backward: chain kernel + tn_partial + tn_reduce, x56 each",
             "batch": "backward: chain kernel + tn_partial x56 each, one batched tn_reduce"}


def algorithmic(shapes, T, r, es, acc):
    flops = sum(6 * T * r * (di + do) + (4 * T * di * do if acc == "dense" else 0) for di, do in shapes)
    nbytes = sum(T * (3 * di + 2 * do) * es + 2 * T * r * es for di, do in shapes)
    return flops, nbytes


def time_region(fn, iters, stream):
    """Average duration (ms) of fn() over `iters` calls, HIP events on the launch stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(shapes, T, r):
    """The CPU oracle (op-for-op restatement of the reference's SoWLinear, fp32) on this host's cores:
    one forward+backward pass over the same 56 layer shapes at the same T."""
    from oracle import sow_oracle as O
    # the GPU box gives one GPU's share of the host (16 cores) although os.cpu_count() reports the whole machine
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))
    gen = torch.Generator().manual_seed(1234)
    cache = {}
    for s in set(shapes):
        di, do = s
        cache[s] = (torch.randn(T, di, generator=gen), torch.randn(T, do, generator=gen),
                    torch.linalg.qr(torch.randn(di, r, generator=gen) * 0.02)[0].contiguous(), torch.randn(r, do, generator=gen) * 0.02)
    x, dy, A, B = cache[shapes[0]]
    O.sow_backward(dy, x, [A], [B], None, None, 1.0, False)  # warm-up
    t0 = time.perf_counter()
    for s in shapes:
        x, dy, A, B = cache[s]
        O.sow_forward(x, [A], [B], None, None, 1.0, None)
        O.sow_backward(dy, x, [A], [B], None, None, 1.0, False)
    dt = time.perf_counter() - t0
    return dict(value=T / dt, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 fwd+bwd pass over the 56 llama_60m SoWLinear shapes, T={T}, fp32, torch-CPU oracle, {dt:.2f} s")


def train_mode(args, world, rank, device):
    """Caller protocol of the reference's scripts/simple_train.py:316-333, 389-405, 425-428, 502-506, 566-572,
    596-650 with synthetic tokens: llama_60m from the JSON's numbers, prepare_sow(rank 50, normal_QR),
    bf16 cast after the swap, AdamW with two groups, accumulate + reset_optimizer between backward and step."""
    import transformers
    from sow_amd import SoWConfig, SoWLinear, accumulate, prepare_sow, reset_optimizer
    torch.manual_seed(42)
    cfg = transformers.LlamaConfig(hidden_size=512, intermediate_size=1376, num_hidden_layers=8, num_attention_heads=8,
                                   vocab_size=32000, max_position_embeddings=1024, rms_norm_eps=1e-6,
                                   tie_word_embeddings=False)
    model = transformers.AutoModelForCausalLM.from_config(cfg)
    targets = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]
    model = prepare_sow(model, SoWConfig(target_modules=targets, rank=args.rank, init_method="normal_QR", scale=1.0,
                                         decompose=None, device=str(device)))
    special, ids = [], set()
    for _, m in model.named_modules():
        if isinstance(m, SoWLinear):
            for wgt in list(m.downscale_weights) + list(m.upscale_weights):
                special.append(wgt)
                ids.add(id(wgt))
    model = model.to(device=device, dtype=torch.bfloat16)
    trainable = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    fused = args.fused_factors and world == 1
    if fused:
        from sow_amd.dp import FactorBucket
        from sow_amd.optimizer import FactorAdamW
        bucket = FactorBucket(special)
        bucket.attach(model)
        fopt = FactorAdamW(bucket, lr=1e-3, weight_decay=0.0)
        opt = torch.optim.AdamW([{"params": trainable, "lr": 1e-3, "weight_decay": 0.0}])
    else:
        opt = torch.optim.AdamW([{"params": trainable, "lr": 1e-3, "weight_decay": 0.0},
                                 {"params": special, "lr": 1e-3, "weight_decay": 0.0}])
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], output_device=device.index,
                                                          broadcast_buffers=False)
    batch, seq = 128, 256
    gen = torch.Generator(device=device).manual_seed(42 + rank)
    tokens = torch.randint(0, 32000, (batch, seq), generator=gen, device=device)
    step_no = [0]

    def step():
        loss = model(input_ids=tokens, labels=tokens.clone()).loss
        loss.backward()
        step_no[0] += 1
        if step_no[0] % args.accumulate_every == 0:      # simple_train.py:618-626 (GA = 1)
            if fused:
                bucket.finalize()
            accumulate(model.module if world > 1 else model)
            if fused:
                bucket.rebind()
                fopt.reset_state()
            else:
                reset_optimizer(opt, group_id=1)
        opt.step()
        opt.zero_grad()
        if fused:
            fopt.step()
            bucket.zero_grad()
        return loss

    for _ in range(max(args.warmup, 1)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms = elapsed / args.steps * 1e3
    if rank == 0:
        print(json.dumps({
            "metric": "llama_60m --architecture sow rank=50 training tokens/s (full step, synthetic tokens)",
            "value": batch * seq * world / (ms * 1e-3), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "llama_60m (HF LlamaForCausalLM from config) + prepare_sow rank 50, batch 128 x seq 256, "
                                   f"AdamW 2 groups, accumulate every {args.accumulate_every} steps",
                       "parallelism": f"ddp{world}", "fused_factors": bool(fused), "final_loss": float(loss.detach())}}))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; ranks beyond the visible devices (a rehearsal of the N > 1 path on a 1-GPU box) share them
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        backend = os.environ.get("SOW_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm (xGMI); "gloo" only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    es = 2 if args.dtype == "bf16" else 4
    shapes = layer_shapes()
    T = args.tokens

    from sow_amd import _lib
    _lib.load()  # fail loudly when the HIP library is missing
    if args.mode == "train":
        train_mode(args, world, rank, device)
        if world > 1:
            dist.destroy_process_group()
        return
    stack = Stack(shapes, T, args.rank, dtype, device, args.acc, args.streams, args.reduce)
    stream = torch.cuda.Stream(device=device)
    torch.cuda.synchronize()

    def comm():
        if world > 1:
            stack.bucket.all_reduce_async()
            stack.bucket.wait()

    graph = None
    with torch.cuda.stream(stream):
        for _ in range(max(args.warmup, 1)):   # W untimed warm-up steps (also sets kernel attributes)
            stack.step()
            comm()
        torch.cuda.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                stack.step()
            graph.replay()
            comm()
            torch.cuda.synchronize()

        def one_step():
            if graph is not None:
                graph.replay()
            else:
                stack.step()
            comm()

        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())

        # per-kernel timing with HIP events on the launch stream (rank 0, after the timed region).
        # forward_all launches exactly one kernel per layer (chain2_kernel<false>), so its average is a
        # single-kernel figure that can be checked against the rocprofv3 stats under profiles/.
        groups = {}
        if rank == 0:
            it = max(3, min(args.steps, 10))
            groups["forward: chain kernel x56"] = time_region(stack.forward_all, it, stream)
            groups[BWD_LABEL[args.reduce]] = time_region(stack.backward_all, it, stream)

    ms = elapsed / args.steps * 1e3
    flops, nbytes = algorithmic(shapes, T, args.rank, es, args.acc)
    if rank == 0:
        n_layers = len(shapes)
        fwd_ms = groups["forward: chain kernel x56"]
        # dominant kernel (largest total time in profiles/r01_bench_v2_kernel_stats.csv): the forward chain
        # kernel.  Algorithmic bytes per launch = T*(d_in + d_out + r)*s averaged over the 56 layers
        # (x read once, y written once, h saved once).
        kname = "sow::chain2_kernel<false> (fused forward chain)" if args.dtype == "bf16" else "sow::chain2f_kernel<false> (fused forward chain, fp32)"
        kbytes = sum(T * (di + do + args.rank) * es for di, do in shapes) / n_layers
        kms = fwd_ms / n_layers
        achieved = kbytes / (kms * 1e-3) / 1e9
        out = {
            "metric": "SoWLinear fwd+bwd tokens/s, llama_60m rank=50 (56-layer SoWLinear stack)",
            "value": T * world / (ms * 1e-3),
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "llama_60m --architecture sow: 56 SoWLinear layers (32x512->512, 16x512->1376, 8x1376->512), "
                                   f"rank {args.rank}, batch 128 x seq 256 = {T} tokens/GPU, fwd+bwd, acc={args.acc}",
                       "tokens_per_gpu": T, "rank": args.rank, "parallelism": f"dp{world}", "hip_graph": graph is not None, "streams": args.streams, "weight_grad_reduce": args.reduce},
            "gflops": flops * world / (ms * 1e-3) / 1e9,
            "algorithmic_gbytes_per_step": nbytes / 1e9,
            "step_hbm_gbs": nbytes / (ms * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": TRAFFIC_BYTES_PER_LAUNCH if (args.dtype == "bf16" and args.acc == "none") else None,
                         "avg_launch_ms": kms, "algorithmic_bytes_per_launch": kbytes},
            "kernel_groups_ms": groups,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(shapes, T, args.rank)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
