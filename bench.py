#!/usr/bin/env python3
"""Benchmark of the SoW hot path on MI355X (contract: see the task statement / DESIGN.md section 6).

Workload (BASELINE.json configs[1]): the 56 SoWLinear layers of llama_60m (`--architecture sow`,
rank 50; 32 x (512->512), 16 x (512->1376), 8 x (1376->512), reference scripts/configs/llama_60m.json)
at batch 128 x seq 256 = 32768 tokens per GPU, bf16.  One step = forward of all 56 layers in model
order, backward of all 56 in reverse order (every layer has its own x / dY buffers, so nothing is
re-read from cache that a real model would not have; the token-slab partial sums of the weight gradients
are reduced in one batched launch at the end of backward), and -- for N > 1 -- ONE RCCL all-reduce of the
flat factor-gradient bucket.  Inputs are resident in HBM before the timed region.  The step is captured in a
HIP graph after warm-up (no host work in the timed region).

Layers that are independent INSIDE a decoder block -- {q, k, v} and {gate, up}; o and down stand alone -- are issued
through the grouped C-ABI calls (sow_forward_group / sow_backward_group: one grid per kernel for the group; every layer
keeps its own x / dY / y, so the algorithmic bytes are unchanged): 4 launches per direction per block instead of 7.
`--group none` issues every layer on its own.

Prints ONE JSON line.  `value` = tokens/s over all ranks (T * N / step time); `gflops` = the
algorithmic 6*T*r*(d_in+d_out) count per second.  At N = 1 the same line also carries, as extra
keys measured in the same process after the headline: `dense` (the steady state after the first accumulate(): dense
frozen accumulator, MFMA-bound), `fp32` (the exact-fp32 parity path against the 157 TF fp32 matrix peak) and `train`
(a full llama_60m training step through the module-swap surface) -- skip them with --only-headline.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak
MFMA_F32_TFLOPS = 157.0    # fp32 matrix peak
# HBM bytes per launch of the roofline kernel, from the rocprofv3 PMC passes made at the commit named inside the file
# (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, tools/pmc_traffic.py); absent or for another kernel -> null in the line.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")
# the dense-accumulator layer as ONE launch per pass (N <= 512: 40 of the 56 layers): the section's roofline kernel
DENSE_KERNEL = "sow::gemm4_kernel<false, true, false> = gemm4h (projection pass + K-extended bf16 product in one launch, N <= 512)"
LLAMA_60M = dict(hidden=512, inter=1376, layers=8)
BLOCK_NAMES = ["q", "k", "v", "o", "gate", "up", "down"]
BLOCK_GROUPS = [[0, 1, 2], [3], [4, 5], [6]]   # model-independent layers inside one decoder block


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
    ap.add_argument("--only-headline", action="store_true", help="skip the dense / fp32 / train sections (N = 1 adds them by default)")
    ap.add_argument("--acc", default="none", choices=["none", "dense"], help="accumulator state of the layers")
    ap.add_argument("--mode", default="stack", choices=["stack", "train"],
                    help="stack: the 56-layer SoWLinear hot path (headline); train: a full llama_60m training step "
                         "through the module-swap surface (prepare_sow, autograd, AdamW, accumulate), secondary figure")
    ap.add_argument("--accumulate-every", type=int, default=4, help="--mode train: SoW accumulation period in steps")
    ap.add_argument("--fused-factors", action="store_true",
                    help="--mode train: factor group in a FactorBucket (gradients written straight into one flat buffer, one "
                         "batched reduction, one fused AdamW kernel, ONE all-reduce of the bucket at N > 1 with DDP covering "
                         "the other parameters) instead of torch.optim.AdamW's second param group")
    ap.add_argument("--no-group-siblings", action="store_true",
                    help="--mode train: do NOT group q/k/v and gate/up (sow_amd.group_siblings: one autograd node and one grid per "
                         "kernel for the siblings of a decoder block)")
    ap.add_argument("--reduce", choices=["batch", "block", "layer"], default="batch",
                    help="weight-gradient reduction: one 5-us launch per layer, or deferred and batched into one launch at the end "
                         "of backward (sow_reduce_batch: same arithmetic, bit-identical gradients), or one batched launch per "
                         "decoder block right after its partial sums")
    ap.add_argument("--tn-group", choices=["block", "group"], default="block",
                    help="weight-gradient partial sums: one launch per decoder block (7 layers, after the block's data-gradient "
                         "kernels) or one per launch group, right after that group's data-gradient kernel (dY still in the "
                         "Infinity Cache)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="2: the weight-gradient launches (one per decoder block) go to a side stream forked from the block's last "
                         "data-gradient kernel and joined before the batched reduction, so their start overlaps the write drain "
                         "of the data-gradient kernels")
    ap.add_argument("--group", choices=["block", "none"], default="block",
                    help="block: {q,k,v} and {gate,up} of a decoder block share launches (grouped C-ABI calls); none: one call per layer")
    ap.add_argument("--dry-run", action="store_true",
                    help="stop after torch.distributed is up: print {n_gpus, rccl} and exit (launcher / rendezvous rehearsal; with "
                         "SOW_BENCH_BACKEND=gloo it needs no GPU)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start N ranks (one process per GPU) of
    this same script under torch.distributed.run as a CHILD process -- before this process has made any GPU call; a process
    that has initialised the GPU must never exec another program -- relay rank 0's JSON line and exit with the child's
    code.  The driver's own `python -m torch.distributed.run ... bench.py --gpus N` form sets WORLD_SIZE and never gets here.
    Mirrors how the reference is launched (readme.md:6 torchrun --nproc-per-node, scripts/simple_train.py:229, 566-572)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


class Stack:
    """The 56-layer SoWLinear stack with resident synthetic inputs and static output buffers."""

    def __init__(self, shapes, T, r, dtype, device, acc, reduce="batch", group="block", tn_group="block", streams=1):
        from sow_amd import ops
        from sow_amd.dp import FactorBucket
        self.shapes, self.T, self.r, self.dtype, self.acc = shapes, T, r, dtype, acc
        self.deferred = ops.DeferredReduce() if reduce in ("batch", "block") else None
        # "block": one batched reduction per decoder block, right after its partial sums (they are still in the Infinity Cache)
        self.block_reduce = [ops.DeferredReduce() for _ in range(len(shapes) // len(BLOCK_NAMES))] if reduce == "block" else None
        self.side = torch.cuda.Stream(device=device) if streams > 1 else None
        kind = 2 if acc == "dense" else 0
        g = torch.Generator(device=device)
        self.x, self.dy, self.A, self.B, self.W = [], [], [], [], []
        params = []
        for li, (d_in, d_out) in enumerate(shapes):
            g.manual_seed(1234 + li)
            self.x.append(torch.randn(T, d_in, generator=g, device=device, dtype=torch.float32).to(dtype))
            self.dy.append(torch.randn(T, d_out, generator=g, device=device, dtype=torch.float32).to(dtype))
            # orthonormal columns; factored on the host: the GPU QR is ~300 tiny rocsolver launches per layer, which a
            # counter-collecting profiler run (rocprofv3 --pmc) spends minutes on before the first kernel of interest
            a = torch.linalg.qr(torch.randn(d_in, r, generator=g, device=device).cpu() * 0.02)[0].to(device)
            b = torch.randn(r, d_out, generator=g, device=device) * 0.02
            self.A.append(torch.nn.Parameter(a.to(dtype).contiguous()))
            self.B.append(torch.nn.Parameter(b.to(dtype).contiguous()))
            params += [self.A[-1], self.B[-1]]
            self.W.append((torch.randn(d_in, d_out, generator=g, device=device) * 0.02).to(dtype) if acc == "dense" else None)
        self.bucket = FactorBucket(params)  # params / grads are views into two flat buffers
        # every layer writes its OWN y and dX buffer (activations and their gradients are distinct live tensors in a
        # model; a buffer re-used by successive layers would have its writes absorbed by the 256 MiB Infinity Cache)
        self.calls = []
        for li, (d_in, d_out) in enumerate(shapes):
            ws = torch.empty(ops.workspace_bytes(T, d_in, d_out, r, 0, kind, dtype) + 256, dtype=torch.uint8, device=device)
            self.calls.append(ops.LayerCall(self.x[li], self.A[li].data, self.B[li].data, acc_down=self.W[li], scale=1.0,
                                            y=torch.empty(T, d_out, dtype=dtype, device=device), dy2=self.dy[li],
                                            dx=torch.empty(T, d_in, dtype=dtype, device=device),
                                            out=(self.A[li].grad, self.B[li].grad, None), grad_beta=0.0, workspace=ws))
        nb = len(BLOCK_NAMES)
        layout = BLOCK_GROUPS if group == "block" else [[i] for i in range(nb)]
        self.group_layers = [[blk * nb + i for i in idx] for blk in range(len(shapes) // nb) for idx in layout]
        self.groups = [ops.LayerGroup([self.calls[li] for li in ids]) for ids in self.group_layers]
        # weight-gradient partial sums: nothing consumes them before the optimizer, so ALL layers of a decoder block go
        # into one launch after the block's data-gradient kernels (dY / dh of 7 layers stay alive that long: ~0.4 GB)
        self.tn_layers = ([[blk * nb + i for i in range(nb)] for blk in range(len(shapes) // nb)]
                          if (group == "block" and tn_group == "block") else self.group_layers)
        self.tn_groups = [ops.LayerGroup([self.calls[li] for li in ids]) for ids in self.tn_layers]

    def forward_all(self, only=None):
        for gi, grp in enumerate(self.groups):
            if only is None or gi in only:
                grp.forward()

    def backward_all(self, only=None, phases=None, tn_only=None):
        """Backward in reverse order.  Per decoder block: the data-gradient kernels of its groups (dX: what the previous
        layer's backward waits for in a real model), then ONE launch of the weight-gradient partial sums of the whole
        block; ONE batched reduction at the very end.  `only` / `phases` / `tn_only` restrict the work for the per-launch
        timings."""
        from sow_amd import _lib
        full = phases is None and only is None and tn_only is None
        first_of_block = {min(gi for gi, ids in enumerate(self.group_layers) if ids[0] in set(tl)): ti
                          for ti, tl in enumerate(self.tn_layers)}
        for gi in reversed(range(len(self.groups))):
            if (full or (phases == _lib.BWD_DATA and (only is None or gi in only))):
                self.groups[gi].backward(_lib.BWD_DATA)
            ti = first_of_block.get(gi)        # the block's first group is its last in backward order: then its weights
            if ti is not None and (full or (phases == _lib.BWD_WEIGHTS_PARTIAL and (tn_only is None or ti in tn_only))):
                tn_ph = _lib.BWD_WEIGHTS_PARTIAL if (self.deferred is not None or not full) else _lib.BWD_WEIGHTS
                if self.deferred is not None:
                    tn_ph |= _lib.BWD_GROUP_SLABS      # slab counts planned over the block; reduction descriptors to match
                if self.side is not None and full:
                    main = torch.cuda.current_stream()
                    self.side.wait_stream(main)          # after this block's data-gradient kernels (they produce dh)
                    with torch.cuda.stream(self.side):
                        self.tn_groups[ti].backward(tn_ph)
                else:
                    self.tn_groups[ti].backward(tn_ph)
                if self.block_reduce is not None and full:
                    self.block_reduce[ti].add_group(self.tn_groups[ti], tn_ph)
                    self.block_reduce[ti].run()
                elif self.deferred is not None and full:
                    self.deferred.add_group(self.tn_groups[ti], tn_ph)
        if self.side is not None and full:
            torch.cuda.current_stream().wait_stream(self.side)
        if self.deferred is not None and self.block_reduce is None and full:
            self.deferred.run()

    def step(self):
        self.forward_all()
        self.backward_all()

    # ---- N > 1: backward cut per decoder block, so that a block's factor gradients can travel while the next block computes
    def backward_block(self, ti):
        """Data-gradient kernels of decoder block `ti` (groups in reverse), its weight-gradient partial sums and their
        reduction: after this the block's slice of the flat gradient buffer is final (needs --reduce block)."""
        from sow_amd import _lib
        assert self.block_reduce is not None and self.side is None
        for gi in reversed(range(len(self.groups))):
            if self.group_layers[gi][0] in self.tn_layers[ti]:
                self.groups[gi].backward(_lib.BWD_DATA)
        tn_ph = _lib.BWD_WEIGHTS_PARTIAL | _lib.BWD_GROUP_SLABS
        self.tn_groups[ti].backward(tn_ph)
        self.block_reduce[ti].add_group(self.tn_groups[ti], tn_ph)
        self.block_reduce[ti].run()

    def block_grad_range(self, ti):
        """[start, end) elements of the flat gradient buffer that hold the factor gradients of decoder block `ti`
        (the bucket's parameters are A, B per layer in layer order; the layers of a block are consecutive)."""
        ids = self.tn_layers[ti]
        lo, hi = 2 * min(ids), 2 * max(ids) + 2
        offs = self.bucket.offsets
        return offs[lo], (offs[hi] if hi < len(offs) else self.bucket.padded_numel)


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


def time_graph(fn, stream, reps=5):
    """ms per replay of a HIP graph of fn() (launch gaps as in the timed step), events on the launch stream."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


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
    out = dict(value=T / dt, unit="tokens/s", cores=torch.get_num_threads(), kind="port",
               sample=f"1 fwd+bwd pass over the 56 llama_60m SoWLinear shapes, T={T}, fp32, torch-CPU oracle, {dt:.2f} s")
    # the same oracle on ONE thread, bounded: the seven layers of one decoder block at T / 8 tokens, scaled to the stack
    # (SURVEY 8(d) asks for both thread counts; the work per token does not depend on T)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    Ts = max(T // 8, 64)
    t0 = time.perf_counter()
    for s in shapes[:len(BLOCK_NAMES)]:
        x, dy, A, B = cache[s]
        O.sow_forward(x[:Ts], [A], [B], None, None, 1.0, None)
        O.sow_backward(dy[:Ts], x[:Ts], [A], [B], None, None, 1.0, False)
    dt1 = (time.perf_counter() - t0) * (len(shapes) / len(BLOCK_NAMES))
    torch.set_num_threads(nthreads)
    out["single_thread"] = dict(value=Ts / dt1, unit="tokens/s", cores=1,
                                sample=f"one decoder block (7 layers) at T={Ts}, scaled x{len(shapes) // len(BLOCK_NAMES)} to the stack")
    return out


def train_mode(args, world, rank, device, steps=None, warmup=None, quiet=False):
    """Caller protocol of the reference's scripts/simple_train.py:316-333, 389-405, 425-428, 502-506, 566-572,
    596-650 with synthetic tokens: llama_60m from the JSON's numbers, prepare_sow(rank 50, normal_QR),
    bf16 cast after the swap, AdamW with two groups, accumulate + reset_optimizer between backward and step.

    --fused-factors (any world size): the factor group lives in a FactorBucket -- SoWLinear's backward writes the factor
    gradients straight into ONE flat buffer, DDP covers only the non-factor parameters (the factors are on its ignore
    list), the bucket's single all-reduce is issued from backward as soon as the last attached layer has queued its
    partial sums (overlapping the embedding / lm_head gradient work and DDP's own buckets), the re-initialised A is
    broadcast after accumulate(), and the factor group steps in one fused AdamW kernel."""
    import transformers
    from sow_amd import SoWConfig, SoWLinear, accumulate, prepare_sow, reset_optimizer
    steps = steps or args.steps
    warmup = warmup if warmup is not None else args.warmup
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
    fused = args.fused_factors
    grouped = 0
    if not args.no_group_siblings:    # composes with --fused-factors: grouped data-gradient launch, block-level weight gradients
        from sow_amd import group_siblings
        grouped = group_siblings(model)
    inner = model
    if fused:
        from sow_amd.dp import FactorBucket
        from sow_amd.optimizer import FactorAdamW
        bucket = FactorBucket(special)
        bucket.attach(model, auto_all_reduce=world > 1)
        fopt = FactorAdamW(bucket, lr=1e-3, weight_decay=0.0)
        opt = torch.optim.AdamW([{"params": trainable, "lr": 1e-3, "weight_decay": 0.0}])
    else:
        opt = torch.optim.AdamW([{"params": trainable, "lr": 1e-3, "weight_decay": 0.0},
                                 {"params": special, "lr": 1e-3, "weight_decay": 0.0}])
    if world > 1:
        if fused:
            bucket.exclude_from_ddp(model)
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
        gscale = bucket.wait() if fused else 1.0          # the bucket's all-reduce was issued inside backward
        if step_no[0] % args.accumulate_every == 0:      # simple_train.py:618-626 (GA = 1)
            accumulate(inner)                             # an attached bucket finalizes before and rebinds after
            if fused:
                bucket.broadcast_factors()
                fopt.reset_state()
            else:
                reset_optimizer(opt, group_id=1)
        opt.step()
        opt.zero_grad()
        if fused:
            fopt.step(grad_scale=gscale)
            bucket.zero_grad()
        return loss

    for _ in range(max(warmup, 1)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms = elapsed / steps * 1e3
    out = {
        "metric": "llama_60m --architecture sow rank=50 training tokens/s (full step, synthetic tokens)",
        "value": batch * seq * world / (ms * 1e-3), "unit": "tokens/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "llama_60m (HF LlamaForCausalLM from config) + prepare_sow rank 50, batch 128 x seq 256, "
                               f"AdamW 2 groups, accumulate every {args.accumulate_every} steps",
                   "parallelism": f"ddp{world}", "fused_factors": bool(fused), "sibling_groups": grouped,
                   "final_loss": float(loss.detach())}}
    if rank == 0 and not quiet:
        print(json.dumps(out))
    del model, inner, opt
    return out


def launch_kinds(stack):
    """Distinct launch kinds of one decoder block: (label, group index within block 0, layer shapes)."""
    ngroups = len(stack.groups) // LLAMA_60M["layers"]
    kinds = []
    for gi in range(ngroups):
        ids = stack.group_layers[gi]
        names = "+".join(BLOCK_NAMES[i % len(BLOCK_NAMES)] for i in ids)
        di, do = stack.shapes[ids[0]]
        kinds.append((f"{names} {len(ids)}x({di}->{do})", gi, [stack.shapes[i] for i in ids]))
    return kinds


def per_launch_table(stack, stream, T, r, es):
    """us / GB/s / fraction of the HBM peak per launch kind (HIP-graph replay of the 8 instances of the kind, one per
    decoder block, events on the launch stream).  Algorithmic bytes per launch: chain kernels T*(D1 + D2 + r)*s per
    layer; weight-gradient partial kernel T*(d_in + d_out + 2 r)*s per layer (x and dY once, h and dh once)."""
    from sow_amd import _lib
    nblk = LLAMA_60M["layers"]
    ngroups = len(stack.groups) // nblk
    table = {}
    for label, gi, shp in launch_kinds(stack):
        only = {blk * ngroups + gi for blk in range(nblk)}
        chain_bytes = sum(T * (di + do + r) * es for di, do in shp)
        for tag, fn in (("fwd chain", lambda: stack.forward_all(only)),
                        ("bwd chain (dX)", lambda: stack.backward_all(only, _lib.BWD_DATA))):
            us = time_graph(fn, stream) / nblk * 1e3
            gbs = chain_bytes / us / 1e3
            table[f"{tag}: {label}"] = {"us": round(us, 2), "GB/s": round(gbs), "frac": round(gbs / HBM_PEAK_GBS, 3),
                                        "algorithmic_MB": round(chain_bytes / 1e6, 2)}
    ntn = len(stack.tn_groups) // nblk
    for k in range(ntn):
        ids = stack.tn_layers[k]
        shp = [stack.shapes[i] for i in ids]
        tn_bytes = sum(T * (di + do + 2 * r) * es for di, do in shp)
        only = {blk * ntn + k for blk in range(nblk)}
        us = time_graph(lambda: stack.backward_all(None, _lib.BWD_WEIGHTS_PARTIAL, only), stream) / nblk * 1e3
        gbs = tn_bytes / us / 1e3
        names = "+".join(BLOCK_NAMES[i % len(BLOCK_NAMES)] for i in ids)
        table[f"bwd weight partials: {names} ({len(ids)} layers)"] = {
            "us": round(us, 2), "GB/s": round(gbs), "frac": round(gbs / HBM_PEAK_GBS, 3), "algorithmic_MB": round(tn_bytes / 1e6, 2)}
    return table


def _stats(ms_list):
    v = sorted(ms_list)
    n = len(v)
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"median": med, "min": v[0], "max": v[-1], "mean": sum(v) / n, "per_replay_ms": [round(x, 4) for x in ms_list]}


def measure_stack(args, dtype_name, acc, world, rank, device, steps, warmup, detail, warm_replays=5):
    """Build the stack, capture the step, time `steps` replays (barrier + synchronize on both sides, max over ranks).
    Every replay is also bracketed by HIP events on the launch stream (`replays`: median / min / max / per-replay list), so a
    record shows whether a slow figure was one stall or a uniform slowdown.

    N > 1: the step is cut into a forward graph and one backward graph per decoder block (last block first); after each
    block's graph the all-reduce of THAT block's slice of the flat factor-gradient bucket is issued on the bucket's side
    stream, so the exchange of block b overlaps the backward of blocks b-1 ... 0 (what DDP's gradient buckets do for the
    reference, simple_train.py:566-572); the step ends when the last slice has arrived."""
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    es = 2 if dtype_name == "bf16" else 4
    shapes = layer_shapes()
    T = args.tokens
    overlap = world > 1 and args.group == "block" and args.tn_group == "block" and args.streams == 1
    reduce_mode = "block" if overlap else args.reduce
    stack = Stack(shapes, T, args.rank, dtype, device, acc, reduce_mode, args.group, args.tn_group, args.streams)
    stream = torch.cuda.Stream(device=device)
    torch.cuda.synchronize()
    nblk = len(stack.tn_layers)

    def eager_step():
        if overlap:
            stack.forward_all()
            for ti in reversed(range(nblk)):
                stack.backward_block(ti)
                stack.bucket.all_reduce_async(start=stack.block_grad_range(ti)[0], end=stack.block_grad_range(ti)[1])
            stack.bucket.wait()
        else:
            stack.step()
            if world > 1:
                stack.bucket.all_reduce_async()
                stack.bucket.wait()

    graphs = None
    res = {}
    with torch.cuda.stream(stream):
        for _ in range(max(warmup, 1)):   # W untimed warm-up steps (also sets kernel attributes)
            eager_step()
        torch.cuda.synchronize()
        if not args.no_graph:
            def capture(fn):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    fn()
                return g
            if overlap:
                graphs = [capture(stack.forward_all)] + [capture(lambda ti=ti: stack.backward_block(ti)) for ti in reversed(range(nblk))]
            else:
                graphs = [capture(stack.step)]

        def one_step():
            if graphs is None:
                eager_step()
            elif overlap:
                graphs[0].replay()
                for k, ti in enumerate(reversed(range(nblk))):
                    graphs[1 + k].replay()
                    lo, hi = stack.block_grad_range(ti)
                    stack.bucket.all_reduce_async(start=lo, end=hi)
                stack.bucket.wait()
            else:
                graphs[0].replay()
                if world > 1:
                    stack.bucket.all_reduce_async()
                    stack.bucket.wait()

        # warm replays: graph upload, caches and the chip's clock / power state (an MFMA-heavy step needs ~10 replays to
        # settle: the first dense replays of a fresh process measured 7.9, 7.5, 7.1, 7.0, 6.9 ... 6.72 ms): at least
        # `warm_replays` and at least 0.25 s of them
        t_warm = time.perf_counter()
        n_warm = 0
        while graphs is not None and (n_warm < warm_replays or time.perf_counter() - t_warm < 0.25) and n_warm < 200:
            one_step()
            torch.cuda.synchronize()
            n_warm += 1
        res["warm_replays"] = n_warm
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record(stream)
        for i in range(steps):
            one_step()
            ev[i + 1].record(stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        ms = elapsed / steps * 1e3
        flops, nbytes = algorithmic(shapes, T, args.rank, es, acc)
        res.update(ms_per_step=ms, flops=flops, nbytes=nbytes, graph=graphs is not None, T=T, shapes=shapes, es=es,
                   replays=_stats([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]), overlap=overlap)
        if rank == 0 and detail:
            it = max(3, min(steps, 10))
            nl = len(stack.groups)
            res["fwd_ms"] = time_region(stack.forward_all, it, stream)
            res["bwd_ms"] = time_region(stack.backward_all if not overlap else
                                        (lambda: [stack.backward_block(ti) for ti in reversed(range(nblk))]), it, stream)
            res["n_fwd_launches"] = nl
            if acc == "none" and dtype_name == "bf16" and not overlap:
                res["per_launch"] = per_launch_table(stack, stream, T, args.rank, es)
            if acc == "dense":
                # layers whose forward is exactly ONE launch (gemm4h: projection pass + K-extended product, N <= 512: 40 of the
                # 56; gemm2h under the NO_GEMM4H switch); flops of one launch =
                # dense product + rank-r projection and extension
                one = [gi for gi, ids in enumerate(stack.group_layers) if all(shapes[i][1] <= 512 for i in ids)]
                n_l = sum(len(stack.group_layers[gi]) for gi in one)
                t = time_region(lambda: stack.forward_all(set(one)), it, stream)
                fl = sum(2 * T * shapes[i][0] * shapes[i][1] + 2 * T * args.rank * (shapes[i][0] + shapes[i][1])
                         for gi in one for i in stack.group_layers[gi])
                res["gemm2h"] = dict(avg_launch_ms=t / n_l, launches=n_l, tflops=fl / (t * 1e-3) / 1e12)
    del stack, graphs
    gc.collect()
    torch.cuda.empty_cache()
    return res


def northstar(device, reps=20, warm=5):
    """BASELINE.json north_star point: ONE SoWLinear forward + backward at r = 50, d_in = d_out = 768, T = 32768, in bf16 and
    in fp32, through the C-ABI layer calls (sow_forward_group / sow_backward_group with n = 1: the same kernels as sow_forward /
    sow_backward), 4 rotating buffer sets -- inputs AND outputs -- in one HIP graph so that no replay finds its inputs in the
    Infinity Cache or has its outputs absorbed by it; per-replay HIP events.  Flops = 6*T*r*(d_in+d_out);
    algorithmic bytes = T*(3 d_in + 2 d_out + 2 r)*s.  `frac` = TFLOP/s over the MFMA peak of the tensors' dtype (2.5 PF bf16,
    157 TF fp32); `frac_of_roofline` = over min(that peak, AI x 8 TB/s)."""
    from sow_amd import _lib, ops
    T, d, r, nset = 32768, 768, 50, 4
    out = {"what": f"one SoWLinear fwd+bwd, T={T}, d_in=d_out={d}, rank={r} (BASELINE.json north_star)"}
    g = torch.Generator(device=device).manual_seed(77)
    for name, dtype, peak in (("bf16", torch.bfloat16, MFMA_BF16_TFLOPS), ("fp32", torch.float32, MFMA_F32_TFLOPS)):
        es = 2 if dtype == torch.bfloat16 else 4
        xs = [torch.randn(T, d, device=device, generator=g).to(dtype) for _ in range(nset)]
        dys = [torch.randn(T, d, device=device, generator=g).to(dtype) for _ in range(nset)]
        A = torch.linalg.qr(torch.randn(d, r, generator=g, device=device).cpu() * 0.02)[0].to(device).to(dtype).contiguous()
        B = (torch.randn(r, d, device=device, generator=g) * 0.02).to(dtype)

        # every buffer set has its OWN outputs (y, dX, h, workspace), as every layer of the headline stack has: through
        # ops.sow_forward / sow_backward they would come out of the caching allocator -- the same blocks for every set -- and
        # their writes would be absorbed by the 256-MiB Infinity Cache (~3 us of 180 at the fp32 point)
        dA, dB = torch.empty_like(A), torch.empty_like(B)
        grps = [ops.LayerGroup([ops.LayerCall(xs[i], A, B, dy2=dys[i], dx=torch.empty_like(xs[i]), out=(dA, dB, None))])
                for i in range(nset)]

        def step():
            for grp in grps:
                grp.forward()
                grp.backward()

        s = torch.cuda.Stream(device=device)
        with torch.cuda.stream(s):
            step()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                step()
            for _ in range(warm):
                gr.replay()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            ev[0].record(s)
            for i in range(reps):
                gr.replay()
                ev[i + 1].record(s)
            torch.cuda.synchronize()
        st = _stats([ev[i].elapsed_time(ev[i + 1]) / nset for i in range(reps)])
        us = st["median"] * 1e3
        flops = 6 * T * r * 2 * d
        nbytes = T * (3 * d + 2 * d) * es + 2 * T * r * es
        tf = flops / us / 1e6
        ai = flops / nbytes
        roof = min(peak, ai * HBM_PEAK_GBS / 1e3)
        rec = {"us": us, "us_min": st["min"] * 1e3, "us_max": st["max"] * 1e3, "tflops": tf, "peak_tflops": peak, "frac": tf / peak,
               "algorithmic_MB": nbytes / 1e6, "hbm_gbs": nbytes / us / 1e3, "hbm_frac": nbytes / us / 1e3 / HBM_PEAK_GBS,
               "arithmetic_intensity": ai, "roofline_tflops": roof, "frac_of_roofline": tf / roof}
        if name == "fp32":
            exact = _lib.load().sow_get_switch(b"F32_EXACT") == 1
            rec["form"] = "exact (v_mfma_f32_32x32x2_f32)" if exact else "3xbf16 (fp32 operands split into three bf16 planes, 6 bf16 MFMAs per product, fp32 accumulate: fp32-equivalent results, runs on the bf16 matrix pipe)"
            rec["peak_note"] = "frac divides by the 157 TF fp32 matrix peak, the figure BASELINE.json's target is stated against"
        out[name] = rec
        del xs, dys, grps
    torch.cuda.empty_cache()
    return out


def traffic_record(kernel_name):
    try:
        with open(TRAFFIC_FILE) as f:
            rec = json.load(f)
        k = rec.get("kernels", {}).get(kernel_name)
        if k:
            return k["hbm_bytes_per_launch"], {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "commit": rec.get("commit"),
                                               "launches_profiled": k.get("launches"), "grouping": rec.get("group")}
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def init_distributed(args):
    """world / rank / device of this process and, for world > 1, the process group.  Returns (world, rank, device, rccl)
    where `rccl` records what actually came up: backend, world size and the ranks seen by an all-gather."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args)                 # never returns
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or "
                 f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SOW_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm (xGMI); "gloo" only for rehearsals
    cpu_only = args.dry_run and backend != "nccl"
    device = torch.device("cpu")
    if not cpu_only:
        # one process per GPU; ranks beyond the visible devices (a rehearsal of the N > 1 path on a 1-GPU box) share them
        dev_index = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    rccl = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        seen = [None] * world
        dist.all_gather_object(seen, rank)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": sorted(int(r) for r in seen)}
        assert dist.get_world_size() == args.gpus and rccl["ranks_seen"] == list(range(world)), rccl
    return world, rank, device, rccl


def main():
    args = parse()
    world, rank, device, rccl = init_distributed(args)
    if args.dry_run:
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl": rccl}))
        if world > 1:
            dist.destroy_process_group()
        return

    from sow_amd import _lib
    _lib.load()  # fail loudly when the HIP library is missing
    if args.mode == "train":
        train_mode(args, world, rank, device)
        if world > 1:
            dist.destroy_process_group()
        return

    head = measure_stack(args, args.dtype, args.acc, world, rank, device, args.steps, args.warmup, detail=True)
    ms, T, shapes, es = head["ms_per_step"], head["T"], head["shapes"], head["es"]
    if rank == 0:
        n_layers = len(shapes)
        if args.acc == "dense":
            # steady state after the first accumulate(): MFMA-bound; dominant kernel by total time = the one-launch forward
            kname = DENSE_KERNEL
            g2h = head["gemm2h"]
            roof = {"bound": "mfma", "kernel": kname, "achieved": g2h["tflops"], "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": g2h["tflops"] / MFMA_BF16_TFLOPS, "traffic": None, "avg_launch_ms": g2h["avg_launch_ms"],
                    "launches_timed": g2h["launches"]}
        else:
            # dominant kernel (largest total time in profiles/): the forward chain kernel.  Algorithmic bytes per launch =
            # sum over the launch's layers of T*(d_in + d_out + r)*s (x read once, y written once, h saved once).
            kernel_sym = "chain2_kernel<false, true>" if args.dtype == "bf16" else "chain3f_kernel<2, 4>"   # <backward, bf16 park tiles>
            kname = f"sow::{kernel_sym} (fused forward chain{', fp32' if args.dtype != 'bf16' else ''})"
            nl = head["n_fwd_launches"]
            kbytes = sum(T * (di + do + args.rank) * es for di, do in shapes) / nl
            kms = head["fwd_ms"] / nl
            achieved = kbytes / (kms * 1e-3) / 1e9
            traffic, tsrc = traffic_record(kernel_sym) if (args.dtype == "bf16" and args.group == "block") else (None, None)
            roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": kms,
                    "launches_per_step": nl, "layers_per_step": n_layers, "algorithmic_bytes_per_launch": kbytes}
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
                       "tokens_per_gpu": T, "rank": args.rank, "parallelism": f"dp{world}", "hip_graph": head["graph"],
                       "weight_grad_reduce": "block" if head["overlap"] else args.reduce, "streams": args.streams,
                       "launch_grouping": ("per decoder block: {q,k,v} {o} {gate,up} {down} for the chain kernels, all 7 layers for the "
                                           "weight-gradient partial sums; resident (persistent) workgroups") if args.group == "block" else "one call per layer"},
            "gflops": head["flops"] * world / (ms * 1e-3) / 1e9,
            "algorithmic_gbytes_per_step": head["nbytes"] / 1e9,
            "step_hbm_gbs": head["nbytes"] / (ms * 1e-3) / 1e9,
            "step_hbm_frac": head["nbytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "roofline": roof,
            "kernel_groups_ms": {"forward (chain kernels)": head["fwd_ms"],
                                 "backward (chain kernels + weight-gradient partials + one batched reduce)": head["bwd_ms"]},
        }
        if "per_launch" in head:
            out["per_launch"] = head["per_launch"]
    if rank == 0:
        out["replays"] = head["replays"]
        if rccl is not None:
            out["rccl"] = dict(rccl, overlap=("per-decoder-block slices of the factor-gradient bucket, each all-reduced on a side stream "
                                              "while the next block's backward runs") if head["overlap"] else "one all-reduce after backward")
    extras = world == 1 and not args.only_headline and args.dtype == "bf16" and args.acc == "none"
    if extras:
        k = max(20, args.steps)
        d = measure_stack(args, "bf16", "dense", 1, 0, device, k, max(args.warmup, 2), detail=True)
        g2h = d["gemm2h"]
        dms = d["replays"]["median"]
        out["dense"] = {
            "what": "steady state after the first accumulate(): dense frozen accumulator + live rank-50 factors (prepare.py:120)",
            "ms_per_step": dms, "ms_per_step_note": f"median of {k} graph replays after 5 warm replays, HIP events per replay",
            "steps": k, "replays": d["replays"], "wallclock_ms_per_step": d["ms_per_step"],
            "fwd_ms": d["fwd_ms"], "bwd_ms": d["bwd_ms"], "tokens_per_s": T / (dms * 1e-3),
            "tflops": d["flops"] / (dms * 1e-3) / 1e12,
            "frac_of_bf16_mfma_peak": d["flops"] / (dms * 1e-3) / 1e12 / MFMA_BF16_TFLOPS,
            "roofline": {"bound": "mfma", "kernel": DENSE_KERNEL,
                         "achieved": g2h["tflops"], "peak": MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": g2h["tflops"] / MFMA_BF16_TFLOPS, "traffic": None, "avg_launch_ms": g2h["avg_launch_ms"],
                         "launches_timed": g2h["launches"]}}
        f = measure_stack(args, "f32", "none", 1, 0, device, k, 2, detail=True)
        fms = f["replays"]["median"]
        x3 = _lib.load().sow_get_switch(b"F32_EXACT") != 1
        out["fp32"] = {"what": "the same stack on fp32 tensors (the 1e-5 parity path)",
                       "form": ("3xbf16: fp32 operands split into three bf16 planes, 6 bf16 MFMAs per product, fp32 accumulate "
                                "(fp32-equivalent; runs on the bf16 matrix pipe)") if x3 else "exact: v_mfma_f32_32x32x2_f32",
                       "ms_per_step": fms, "steps": k, "replays": f["replays"], "fwd_ms": f["fwd_ms"], "bwd_ms": f["bwd_ms"],
                       "tflops": f["flops"] / (fms * 1e-3) / 1e12,
                       "frac_of_fp32_mfma_peak": f["flops"] / (fms * 1e-3) / 1e12 / MFMA_F32_TFLOPS,
                       "peak_note": "flop count over the 157 TF fp32 matrix peak (the figure the target is stated against), "
                                    "whichever pipe the products run on"}
        out["northstar"] = northstar(device)
        try:
            targs = argparse.Namespace(**vars(args))
            targs.fused_factors = True     # FactorBucket path: block-level row-owner weight gradients, one fused AdamW kernel
            t = train_mode(targs, 1, 0, device, steps=max(3, min(args.steps, 10)), warmup=3, quiet=True)
            out["train"] = {"what": t["config"]["workload"], "ms_per_step": t["ms_per_step"], "tokens_per_s": t["value"],
                            "fused_factors": t["config"]["fused_factors"], "sibling_groups": t["config"]["sibling_groups"]}
        except Exception as e:  # transformers missing or too old on the box: the headline must still print
            out["train"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(shapes, T, args.rank)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
