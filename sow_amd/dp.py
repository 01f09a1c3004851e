"""Data-parallel plumbing for the SoW factor gradients.

The reference wraps the whole model in DistributedDataParallel (simple_train.py:566-572), which
all-reduces every trainable parameter in 25 MB buckets.  The SoW factors are 112 small tensors
(llama_60m r=50: 3.9 M elements); here they live in ONE flat parameter buffer and ONE flat gradient
buffer, so that
  * the data-parallel exchange is a single RCCL all-reduce over xGMI (backend "nccl" on ROCm),
    issued on a side stream as soon as the last factor gradient is written and overlapped with the
    rest of backward,
  * the optimizer step for the factor group is a single fused kernel (optimizer.FactorAdamW),
  * reset_optimizer is a single memset.
Ranks hold identical replicas; tokens are the sharded unit (weak scaling).  The periodic
accumulate() needs no reduction, but the re-initialised A must be identical on all ranks: either
identical seeds (what the reference relies on, simple_train.py:217) or `broadcast_factors()`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .layer import SoWLinear


def factor_parameters(model: nn.Module) -> List[nn.Parameter]:
    """All A_i then B_i of every SoWLinear, in module order (the `special_params` list of
    simple_train.py:389-405)."""
    out = []
    for _, m in model.named_modules():
        if isinstance(m, SoWLinear):
            out.extend(list(m.downscale_weights))
            out.extend(list(m.upscale_weights))
    return out


def _block_key(name: str) -> str:
    """Decoder-block prefix of a module name: everything up to the last numeric path component that is not the leaf
    (`model.layers.3.self_attn.q_proj` -> `model.layers.3`, `encoder.layer.7.attention.output.dense` -> `encoder.layer.7`);
    modules outside such a container share their parent's name."""
    parts = name.split(".")
    for i in range(len(parts) - 2, -1, -1):
        if parts[i].isdigit():
            return ".".join(parts[:i + 1])
    return ".".join(parts[:-1])


class _GradSink:
    """Per-layer hook used by SoWLinear's autograd backward after FactorBucket.attach(): runs the data gradient right away
    (dX is what the previous layer's backward waits for) and queues the layer's weight gradients with its decoder block.
    When every attached layer of the block has reported in, the block's token-slab partial sums run as ONE grouped launch
    (sow_backward_group: the row-owner kernel with slab counts planned over the block, the headline path of bench.py),
    accumulating into the layers' views of the flat gradient buffer; the reductions of all blocks are deferred to
    FactorBucket.finalize()."""

    def __init__(self, bucket, pA, pB, block=""):
        self.bucket, self.pA, self.pB, self.block = bucket, pA, pB, block
        self.ws = None
        self.pending = False      # partial sums launched, reduction not yet run
        self.queued = False       # data gradient done, waiting for the rest of the block

    def usable(self, A, B) -> bool:
        # n_iter == 1 layers only (A, B ARE the bucket's parameters), gradients bound to the flat buffer, r <= 64
        pA, pB = self.pA, self.pB
        return (A.data_ptr() == pA.data_ptr() and B.data_ptr() == pB.data_ptr() and B.shape[0] <= 64
                and pA.grad is not None and pB.grad is not None
                and pA.grad.data_ptr() == self.bucket.grad_ptr(pA) and pB.grad.data_ptr() == self.bucket.grad_ptr(pB))

    def prepare(self, x2, B, acc_down, acc_up):
        """Checks and workspace of one backward pass through this layer; returns (acc_kind, r_acc)."""
        from . import _lib, ops
        self.bucket._refuse_backward_during_collective()     # before anything is added to an already reduced buffer
        if self.queued:              # the same layer again before its block was complete (shared module, checkpoint replay)
            self.bucket._flush_block(self.block)
        if self.pending:             # second backward through this layer before finalize(): its partials are still needed
            self.bucket.finalize()
        T, d_in = x2.shape
        r, d_out = B.shape
        kind = ops.acc_kind(acc_down, acc_up)
        r_acc = acc_down.shape[1] if kind == _lib.ACC_LOWRANK else 0
        need = ops.workspace_bytes(T, d_in, d_out, r, r_acc, kind, x2.dtype) + 256
        if self.ws is None or self.ws.numel() < need or self.ws.device != x2.device:
            self.ws = torch.empty(need, dtype=torch.uint8, device=x2.device)
        return kind, r_acc

    def queue(self, dy2, x2, h, A, B, acc_down, acc_up, scale, kind, r_acc):
        """The data gradient of this pass has been enqueued (dh is in self.ws): hand the weight gradients to the block."""
        from . import _lib
        self.bucket._block_add(self, (dy2, x2, h, A, B, acc_down if kind != _lib.ACC_NONE else None,
                                      acc_up if kind == _lib.ACC_LOWRANK else None, float(scale), kind, r_acc))

    def backward(self, dy2, x2, h, A, B, acc_down, acc_up, scale):
        from . import _lib, ops
        kind, r_acc = self.prepare(x2, B, acc_down, acc_up)
        out = (self.pA.grad, self.pB.grad, None)
        dy2 = dy2.contiguous()
        dx, _, _, _ = ops.sow_backward(dy2, x2, h, A, B, acc_down, acc_up, scale, False, out=out, grad_beta=1.0,
                                       phases=_lib.BWD_DATA, workspace=self.ws)
        self.queue(dy2, x2, h, A, B, acc_down, acc_up, scale, kind, r_acc)
        return dx


class FactorBucket:
    """Re-homes the factor parameters (and their .grad) as views into two flat buffers."""

    def __init__(self, params: Iterable[nn.Parameter]):
        self.params = list(params)
        if not self.params:
            raise ValueError("FactorBucket needs at least one parameter")
        p0 = self.params[0]
        self.numel = sum(p.numel() for p in self.params)
        # 64-element alignment of every slot keeps the views 16-byte aligned for the vector paths
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != p0.dtype or p.device != p0.device:
                raise ValueError("FactorBucket parameters must share dtype and device")
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.padded_numel = off
        self.flat_param = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        self.flat_grad = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        for p, o in zip(self.params, self.offsets):
            view = self.flat_param[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self._works: list = []
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._armed = True
        self._reducer = None
        self._sinks_pending: list = []
        self._blocks: dict = {}          # decoder-block key -> {"n": attached layers, "queue": [(sink, record)]}
        self._off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        self._n_attached = 0
        self._arrived = 0
        self._auto = False
        self._auto_group = None
        self._average = True
        self._group = None

    # ------------------------------------------------------------------ deferred weight-gradient reduction
    def grad_ptr(self, p) -> int:
        return self.flat_grad.data_ptr() + self._off_of[id(p)] * self.flat_grad.element_size()

    def attach(self, model: nn.Module, auto_all_reduce: bool = False, group=None) -> int:
        """Let the SoWLinear layers of `model` (n_iter = 1, no bias) write their weight gradients straight into the
        flat buffer and defer the final reduction to finalize(): one launch per step instead of one per layer, and no
        per-parameter AccumulateGrad.  Gradients ACCUMULATE (zero_grad() between steps); call finalize() after
        backward, before the gradients are read (all_reduce_async() and FactorAdamW.step() do).  Returns the number of
        layers attached; layers with a bias or n_iter > 1 keep the ordinary autograd path.

        auto_all_reduce: issue the bucket's single all-reduce FROM BACKWARD, as soon as the last attached layer has
        queued its partial sums (the first SoW layer of the model: what is left of backward -- embedding gradients, DDP's
        own buckets for the non-factor parameters -- overlaps the collective); the step then only calls wait().
        With gradient accumulation (simple_train.py:596-650 runs `gradient_accumulation` micro-batches per update) only
        the LAST micro-batch's backward may reduce: run the others under `with bucket.no_sync():` (or
        `bucket.require_sync(False)` before them and `require_sync(True)` before the last), exactly as with
        DistributedDataParallel.no_sync().  A backward that arrives while a collective is pending raises.

        Autograd returns None for attached factors, so DistributedDataParallel must not manage them: attach BEFORE
        wrapping and call exclude_from_ddp(model) (a model that is already DDP-wrapped is refused)."""
        from . import ops
        if isinstance(model, torch.nn.parallel.DistributedDataParallel):
            raise RuntimeError("FactorBucket.attach: attach to the bare model and call exclude_from_ddp(model) BEFORE wrapping "
                               "it in DistributedDataParallel (attached factors get no autograd gradient, DDP's reducer "
                               "would wait for them forever)")
        if self._reducer is None:
            self._reducer = ops.DeferredReduce()
        mine = {id(p) for p in self.params}
        n = 0
        self._blocks = {}
        for name, m in model.named_modules():
            if isinstance(m, SoWLinear) and m.n_iter == 1 and m.bias is None:
                pA, pB = m.downscale_weights._parameters["0"], m.upscale_weights._parameters["0"]
                if id(pA) in mine and id(pB) in mine:
                    key = _block_key(name)
                    m._grad_sink = _GradSink(self, pA, pB, key)
                    self._blocks.setdefault(key, {"n": 0, "queue": []})["n"] += 1
                    n += 1
        self._n_attached = n
        self._auto, self._auto_group = bool(auto_all_reduce), group
        return n

    def exclude_from_ddp(self, model: nn.Module) -> List[str]:
        """Put the bucket's parameters on DistributedDataParallel's ignore list for `model` (call before wrapping):
        DDP then reduces only the non-factor parameters (embeddings, norms, lm_head) in its own buckets while the
        factors travel in this bucket's ONE all-reduce."""
        mine = {id(p) for p in self.params}
        names = [n for n, p in model.named_parameters() if id(p) in mine]
        torch.nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(model, names)
        return names

    def require_sync(self, flag: bool = True) -> None:
        """Arm (default) or disarm the all-reduce-from-backward of attach(auto_all_reduce=True): disarmed backward passes
        only accumulate into the local flat gradient buffer (non-final micro-batches of a gradient-accumulation step)."""
        self._armed = bool(flag)

    def no_sync(self):
        """Context manager: backward passes inside accumulate locally and issue no collective (DDP.no_sync() analogue)."""
        bucket = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.old = bucket._armed
                bucket._armed = False

            def __exit__(self_inner, *exc):
                bucket._armed = self_inner.old
                return False

        return _NoSync()

    def _refuse_backward_during_collective(self) -> None:
        if self._works:
            raise RuntimeError("FactorBucket: a backward pass reached an attached layer while the bucket's all-reduce is "
                               "pending -- its gradients would be added on top of an already reduced buffer.  With gradient "
                               "accumulation run the non-final micro-batches under `with bucket.no_sync():`; call wait() "
                               "before the next armed backward")

    def _layer_done(self) -> None:
        self._refuse_backward_during_collective()
        self._arrived += 1
        if self._auto and self._armed and self._n_attached and self._arrived % self._n_attached == 0:
            self.all_reduce_async(group=self._auto_group)

    # ------------------------------------------------------------------ block-level weight gradients
    def _block_add(self, sink, rec) -> None:
        blk = self._blocks.setdefault(sink.block, {"n": 1, "queue": []})
        blk["queue"].append((sink, rec))
        sink.queued = True
        if len(blk["queue"]) >= blk["n"]:
            self._flush_block(sink.block)
        self._layer_done()

    def _flush_block(self, key) -> None:
        """ONE grouped launch of the weight-gradient partial sums of the layers queued for decoder block `key`
        (reference: the seven SoWLinear backward passes of an HF decoder block, sow.py:107-126 under autograd)."""
        from . import _lib, ops
        blk = self._blocks.get(key)
        if not blk or not blk["queue"]:
            return
        q, blk["queue"] = blk["queue"], []
        # layers of one dtype / device go together (a model holds one of each; anything else flushes in runs)
        while q:
            dt, dev = q[0][1][1].dtype, q[0][1][1].device
            run = [e for e in q if e[1][1].dtype == dt and e[1][1].device == dev]
            q = [e for e in q if not (e[1][1].dtype == dt and e[1][1].device == dev)]
            arr = (_lib.LayerArgs * len(run))()
            stable = []
            for i, (sink, (dy2, x2, h, A, B, acc_down, acc_up, scale, kind, r_acc)) in enumerate(run):
                T, d_in = x2.shape
                r, d_out = B.shape
                a = arr[i]
                a.x, a.A, a.B = x2.data_ptr(), A.data_ptr(), B.data_ptr()
                a.acc_down = acc_down.data_ptr() if acc_down is not None else None
                a.acc_up = acc_up.data_ptr() if acc_up is not None else None
                a.bias, a.y, a.h_save, a.dy, a.dx = None, dy2.data_ptr(), h.data_ptr(), dy2.data_ptr(), x2.data_ptr()
                a.dA, a.dB, a.dbias = sink.pA.grad.data_ptr(), sink.pB.grad.data_ptr(), None
                a.T, a.d_in, a.d_out, a.r_live, a.r_acc, a.acc_kind = T, d_in, d_out, r, r_acc, kind
                a.scale, a.grad_beta = scale, 1.0
                a.workspace, a.workspace_bytes = sink.ws.data_ptr(), sink.ws.numel()
                stable.append((a.dA, a.dB, a.workspace, T, d_in, d_out, r, r_acc, kind))
            group = ops.LayerGroup.from_args(arr, len(run), ops._dt(run[0][1][1]), dev, keep=run)
            phases = _lib.BWD_WEIGHTS_PARTIAL | _lib.BWD_GROUP_SLABS
            group.backward(phases)
            self._reducer.add_group(group, phases, stable_key=tuple(stable))
            for sink, _ in run:
                sink.queued, sink.pending = False, True
                self._sinks_pending.append(sink)

    def finalize(self) -> None:
        """Launch the weight gradients of incomplete blocks (layers that did not all run backward) and sum the pending
        slab partials of every attached layer (no-op when nothing is pending)."""
        for key, blk in self._blocks.items():
            if blk["queue"]:
                self._flush_block(key)
        if self._reducer is not None and self._sinks_pending:
            self._reducer.run()
            for s in self._sinks_pending:
                s.pending = False
            self._sinks_pending.clear()

    def rebind(self) -> None:
        """Call after SoWLinear.accumulate(): `from_weights` rebinds .data to fresh tensors
        (reference sow.py:37-39); copy them back into the flat buffer and re-point the views."""
        for p, o in zip(self.params, self.offsets):
            view = self.flat_param[o:o + p.numel()].view_as(p)
            if p.data.data_ptr() != view.data_ptr():
                view.copy_(p.data)
                p.data = view
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[o:o + p.numel()].data_ptr():
                g = self.flat_grad[o:o + p.numel()].view_as(p)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g

    def zero_grad(self) -> None:
        if self.flat_grad.is_cuda:
            from . import ops
            ops.zero_([self.flat_grad])
        else:
            self.flat_grad.zero_()

    # ------------------------------------------------------------------ collectives
    def all_reduce_async(self, group=None, average: bool = True, start: Optional[int] = None, end: Optional[int] = None) -> None:
        """One sum all-reduce of the whole factor-gradient bucket (RCCL over xGMI on GPUs, gloo on CPU
        in the tests).  On GPU it runs on a side stream ordered after the current stream.  `start` / `end` (elements
        of the flat buffer) restrict it to a slice -- e.g. the gradients of one decoder block, issued as soon as that
        block's backward has been queued so that the exchange overlaps the rest of backward; several slices may be
        in flight, wait() waits for all of them.  Without a slice the pending deferred reductions are finalized first."""
        if start is None and end is None:
            self.finalize()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        buf = self.flat_grad if (start is None and end is None) else self.flat_grad[(start or 0):(end if end is not None else self.padded_numel)]
        op = dist.ReduceOp.SUM
        if self.flat_grad.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.flat_grad.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.flat_grad.device))
            with torch.cuda.stream(self._comm_stream):
                self._works.append(dist.all_reduce(buf, op=op, group=group, async_op=True))
        else:
            self._works.append(dist.all_reduce(buf, op=op, group=group, async_op=True))
        self._average = average
        self._group = group

    def wait(self) -> float:
        """Block the current stream on the collective(s); returns the scale the optimizer must apply to
        the summed gradient (1/world for averaging -- folded into FactorAdamW.step(grad_scale))."""
        self._arrived = 0
        if not self._works:
            if self._auto and dist.is_available() and dist.is_initialized() and dist.get_world_size(self._auto_group) > 1:
                # not every attached layer ran backward this step (unused branch), or every backward ran disarmed: reduce now
                self.all_reduce_async(group=self._auto_group)
            if not self._works:
                self.finalize()
                return 1.0
        for w in self._works:
            w.wait()
        if self.flat_grad.is_cuda and self._comm_stream is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._comm_stream)
        self._works = []
        return 1.0 / dist.get_world_size(self._group) if self._average else 1.0

    def broadcast_factors(self, src: int = 0, group=None) -> None:
        """Make the (re-initialised) factors identical on all ranks after accumulate()."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat_param, src=src, group=group)
