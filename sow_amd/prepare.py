"""Model surgery and the periodic step -- host-side mirror of tn_gradient/prepare.py and of
scripts/utils/training_utils.py:257-277 (reset_optimizer).  Same names and signatures as the
reference so that `from tn_gradient.prepare import prepare_sow, accumulate, load_sow, SoWConfig`
keeps working through the tn_gradient alias package.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .layer import SoWLinear

try:  # the reference derives SoWConfig from peft.PeftConfig (prepare.py:27); peft is optional here
    from peft import PeftConfig as _ConfigBase  # type: ignore
except Exception:  # pragma: no cover - peft absent in this image
    class _ConfigBase:  # minimal stand-in: keeps arbitrary keyword fields
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                setattr(self, k, v)


class SoWConfig(_ConfigBase):
    """prepare.py:27-38."""

    def __init__(self, target_modules, rank=16, scale=1.0, device="cpu", init_method="normal_QR", decompose="keep",
                 **kwargs):
        super().__init__(**kwargs)
        self.rank = rank
        self.scale = scale
        self.target_modules = target_modules
        self.device = device
        self.init_method = init_method
        self.decompose = decompose
        self.peft_type = "LORA"


def _is_target(name: str, module: nn.Module, targets, max_split: int) -> bool:
    """Suffix match of dotted module names (prepare.py:72-83).  The trailing-component loop stops
    one short of the full name, exactly as the reference's range() does."""
    if not isinstance(module, nn.Linear):
        return False
    parts = name.split(".")
    if len(parts) == 1 and parts[0] in targets:
        return True
    for i in range(1, min(max_split + 1, len(parts))):
        if ".".join(parts[-i:]) in targets:
            return True
    return False


def prepare_sow(model, config: SoWConfig):
    """Replace every targeted nn.Linear by a SoWLinear (prepare.py:41-179).

    decompose: None -> fresh factors, empty accumulator (pre-training);
               'keep' -> accumulator = W^T, fresh factors (fine-tuning, default);
               'qr'   -> Q, R = qr(W^T): accumulator = Q[:, :-r] R[:-r, :], factors = last r columns/rows.
    `virtual_rank` is forced to min(in, out) (prepare.py:120) so later accumulate() calls keep a dense
    accumulator."""
    targets = config.target_modules
    max_split = max(len(t.split(".")) for t in targets)
    # snapshot first: the reference iterates named_modules() lazily while swapping, which visits the
    # same (name, module) pairs because replaced modules are leaves
    todo = [(n, m) for n, m in model.named_modules() if _is_target(n, m, targets, max_split)]
    lookup = dict(model.named_modules())
    for name, module in todo:
        new_layer = SoWLinear(
            in_features=module.in_features,
            out_features=module.out_features,
            rank=config.rank,
            n_iter=1,
            scale=config.scale,
            init_method=config.init_method,
            bias=module.bias is not None,
            dtype=module.weight.data.dtype,
            device=config.device,
            init_params=config.decompose != "qr",
        )
        new_layer.virtual_rank = min(module.in_features, module.out_features)
        if config.decompose == "qr":
            r = config.rank
            wt = module.weight.data.t().to(config.device)
            if not wt.is_cuda:
                raise RuntimeError("prepare_sow(decompose='qr') factorises on the GPU (reference prepare.py:124 "
                                   "hard-codes 'cuda'); use SoWConfig(device='cuda')")
            k = min(wt.shape)
            q, rr = ops.qr_thin(wt.contiguous(), k, need_r=True, out_dtype=torch.float32)
            w_acc = ops.gemm(q[:, :-r].contiguous(), rr[:-r, :].contiguous())
            new_layer.downscale_weights.from_weights(list(torch.split(q[:, -r:].contiguous().to(wt.dtype), r, dim=1)))
            new_layer.upscale_weights.from_weights(list(torch.split(rr[-r:, :].contiguous().to(wt.dtype), r, dim=0)))
            new_layer.acc_downweight = nn.Parameter(w_acc.to(wt.dtype).contiguous(), requires_grad=False)
        elif config.decompose == "keep":
            new_layer.acc_downweight = nn.Parameter(module.weight.data.t().to(config.device).contiguous(),
                                                    requires_grad=False)
        if module.bias is not None:
            new_layer.bias = module.bias
        if "." in name:
            parent_name, child_name = name.rsplit(".", 1)
            setattr(lookup[parent_name], child_name, new_layer)
        else:
            setattr(model, name, new_layer)
    return model


def load_sow(model, checkpoint_path):
    """Load a safetensors checkpoint into a model with SoW layers (prepare.py:188-215): zero-numel
    parameters (the accumulator before it exists) are REPLACED by the checkpoint tensor, the others
    are copied into."""
    from safetensors.torch import load_file

    loaded = load_file(checkpoint_path)
    state_keys = set(model.state_dict().keys())
    modules = dict(model.named_modules())
    for name, tensor in loaded.items():
        if name not in state_keys:
            continue
        obj = model
        for part in name.split("."):
            obj = getattr(obj, part)
        if obj.numel() == 0:
            # the reference loads before model.to(device) (simple_train.py:357 vs :425), so its clone stays wherever
            # safetensors put it; here the replacement lands on the device the placeholder already lives on, so that
            # loading into a model that is already on the GPU works too
            new_param = nn.Parameter(tensor.clone().to(obj.device), requires_grad=False)
            if "." in name:
                parent, child = name.rsplit(".", 1)
                setattr(modules[parent], child, new_param)
            else:
                setattr(model, name, new_param)
        else:
            obj.data.copy_(tensor.data)


_ACC_STREAMS: dict = {}
_ACC_WIDTH = 8


def accumulate(model):
    """prepare.py:219-222: accumulate() on every SoWLinear.

    The layers are independent and each one's re-factorisation is a single-workgroup Householder panel (latency
    bound, ~0.5 ms), so on the GPU they are spread round-robin over a few side streams that fork from and join the
    current stream: 56 layers take ~7 QR latencies instead of 56.  The batched path (layers on the dense-accumulator
    branch) consumes the generator DIFFERENTLY from the sequential loop: it draws only the [in, rank] columns that reach
    Q[:, :rank], for all its layers in one normal_() per (device, dtype), where the per-layer path (and the reference,
    sow.py:163-165) draws [in, out] per layer -- so a seeded run is reproducible within a path, not across the two."""
    mods = [m for _, m in model.named_modules() if isinstance(m, SoWLinear)]
    # layers attached to a FactorBucket (sow_amd/dp.py): pending partial sums belong to the OLD factors -- reduce them
    # first; afterwards the rebound .data tensors are copied back into the flat buffer so the fused optimizer and the
    # all-reduce keep seeing them
    buckets = {id(s.bucket): s.bucket for s in (getattr(m, "_grad_sink", None) for m in mods) if s is not None}
    for b in buckets.values():
        b.finalize()
    fast = [m for m in mods if _batchable(m)]
    if len(fast) >= 2:
        _accumulate_batched(fast)
        slow = [m for m in mods if not _batchable_cached.pop(id(m), False)]
    else:
        _batchable_cached.clear()
        slow = mods
    _accumulate_layers(slow)
    for b in buckets.values():
        b.rebind()


_batchable_cached: dict = {}


def _batchable(m) -> bool:
    """Layers the batched entry point covers: one factor pair, r <= 64, on the GPU, on the dense-accumulator branch of
    sow.py:144-153 (what prepare_sow sets up, prepare.py:120) with no low-rank accumulator left to materialise."""
    A = m.downscale_weights._parameters["0"] if m.n_iter == 1 else None
    ok = (A is not None and A.is_cuda and m.rank <= 64 and m.virtual_rank >= min(m.in_features, m.out_features)
          and m.acc_upweight.numel() == 0 and A.dtype in ops._DT and A.is_contiguous()
          and m.upscale_weights._parameters["0"].is_contiguous()
          and (m.acc_downweight.numel() == 0 or (m.acc_downweight.dtype == A.dtype and m.acc_downweight.is_contiguous()
                                                 and m.acc_downweight.device == A.device)))
    _batchable_cached[id(m)] = ok
    return ok


def _accumulate_batched(mods):
    """sow.py:128-178 for many layers through ONE C call (sow_accumulate_batch: one launch per phase for all of them).

    In-place where the reference rebinds: the dense accumulator is updated where it lives, the new orthonormal A and the
    zeroed B are written into the parameters' own storage (after the update has consumed the old factors -- stream
    order), so optimizer references, flat-bucket views and allocator state are untouched.  Only the first `rank`
    columns of the [in, out] Gaussian of sow.py:163-171 reach Q[:, :rank], so unless a test has hooked the draw
    (`_fresh_gaussian`), only those columns are drawn -- one normal_() launch for the whole model."""
    import ctypes

    from . import _lib
    lib = _lib.load()
    # one C call per (device, dtype): a launch takes raw pointers of ONE device (a model spread over several GPUs gets one
    # call, with its own draws and workspace, per GPU)
    by_key = {}
    for m in mods:
        A0 = m.downscale_weights._parameters["0"]
        by_key.setdefault((A0.device, A0.dtype), []).append(m)
    for (dev, dtype), group in by_key.items():
        es = 2 if dtype == torch.bfloat16 else 4
        hooked = [("_fresh_gaussian" in m.__dict__) or (type(m)._fresh_gaussian is not SoWLinear._fresh_gaussian_default)
                  for m in group]
        qr = [m.init_method == "normal_QR" for m in group]
        # Gaussian draws for the re-initialisation, in the accumulator's dtype as in sow.py:163-165
        n_flat = sum(m.in_features * m.rank for m, h in zip(group, hooked) if not h)
        flat = torch.empty(n_flat, device=dev, dtype=dtype).normal_(mean=0.0, std=0.02) if n_flat else None
        draws, off = [], 0
        for m, h, q in zip(group, hooked, qr):
            if h:
                shape = (m.in_features, m.out_features) if q else (m.in_features, m.rank)
                draws.append(m._fresh_gaussian(shape, dev, dtype).to(dtype).contiguous())
            else:
                draws.append(flat[off:off + m.in_features * m.rank].view(m.in_features, m.rank))
                off += m.in_features * m.rank
        ws_sizes = [((int(lib.sow_qr_workspace_bytes(m.in_features, d.shape[1], m.rank, ops._DT[dtype], 0)) + 255) // 256 * 256) if q else 0
                    for m, d, q in zip(group, draws, qr)]
        ws = torch.empty(sum(ws_sizes) + 256, device=dev, dtype=torch.uint8)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        arr = (_lib.AccumulateArgs * len(group))()
        keep, off = [], 0
        for i, (m, d, q) in enumerate(zip(group, draws, qr)):
            A, B = m.downscale_weights._parameters["0"].data, m.upscale_weights._parameters["0"].data
            if m.acc_downweight.numel():
                acc, beta = m.acc_downweight.data, 1.0
            else:
                acc, beta = torch.empty(m.in_features, m.out_features, device=dev, dtype=dtype), 0.0
                m.acc_downweight = nn.Parameter(acc, requires_grad=False)
                m.acc_upweight = nn.Parameter(torch.empty(0, device=dev), requires_grad=False)
            keep.append((A, B, acc, d))
            a = arr[i]
            a.acc, a.A, a.B = acc.data_ptr(), A.data_ptr(), B.data_ptr()
            a.draw, a.ld_draw, a.draw_cols = (d.data_ptr(), d.stride(0), d.shape[1]) if q else (None, 0, 0)
            a.A_new, a.zero, a.zero_bytes = A.data_ptr(), B.data_ptr(), B.numel() * es
            a.d_in, a.d_out, a.r, a.r_new = m.in_features, m.out_features, m.rank, m.rank
            a.scale, a.acc_beta = float(m.scale), beta
            a.workspace, a.workspace_bytes = (ws_ptr + off, ws_sizes[i]) if q else (None, 0)
            off += ws_sizes[i]
        ops._launch(dev, "sow_accumulate_batch", lib.sow_accumulate_batch, arr, len(group), ops._DT[dtype])
        for (A, _, _, d), q in zip(keep, qr):
            if not q:                       # plain Gaussian re-initialisation (sow.py:174): after the update has read A
                A.copy_(d)


def _accumulate_layers(mods):
    dev = mods[0].downscale_weights[0].device if mods else None
    if len(mods) < 2 or dev is None or dev.type != "cuda":
        for m in mods:
            m.accumulate()
        return
    cur = torch.cuda.current_stream(dev)
    streams = _ACC_STREAMS.get(dev)
    if streams is None:
        streams = _ACC_STREAMS[dev] = [torch.cuda.Stream(device=dev) for _ in range(_ACC_WIDTH)]
    for s in streams:
        s.wait_stream(cur)
    for i, m in enumerate(mods):
        with torch.cuda.stream(streams[i % _ACC_WIDTH]):
            m.accumulate()
    for s in streams:
        cur.wait_stream(s)


def reset_optimizer(optimizer, group_id):
    """scripts/utils/training_utils.py:257-277: zero exp_avg / exp_avg_sq (/ max_exp_avg_sq) and the
    step counter of one param group.  The reference allocates fresh zero tensors per parameter; here
    the existing state buffers are zeroed by ONE multi-tensor launch (sow_zero_state)."""
    group = optimizer.param_groups[group_id]
    bufs = []
    for param in group["params"]:
        state = optimizer.state[param]
        for key in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if group.get("amsgrad", False) else ()):
            buf = state.get(key)
            if buf is None or buf.shape != param.shape or not buf.is_cuda or not buf.is_contiguous():
                state[key] = torch.zeros_like(param, memory_format=torch.preserve_format)
            else:
                bufs.append(buf)
        if "step" in state:
            step = state["step"]
            if torch.is_tensor(step) and step.is_cuda and step.is_contiguous():
                bufs.append(step)
            elif torch.is_tensor(step):
                state["step"] = torch.zeros_like(step)
            else:
                state["step"] = 0
    ops.zero_(bufs)


class SoWModel:
    """prepare.py:181-185: thin holder (the reference derives it from PeftModel but skips its __init__)."""

    def __init__(self, model, config: SoWConfig):
        self.config = config
        self.model = prepare_sow(model, config)


def export_alignment(module, export_name, out_dir=None):
    """prepare.py:224-245 (analysis dump, run_glue.py:55): percentage overlap between the left singular vectors of the
    accumulator and those of the live update sum_i A_i B_i.  The products run on the HIP GEMM; the SVD is
    torch.linalg.svd (utils.svd_weight).  The reference writes to a hard-coded home directory (:245); here the .npy goes
    to `out_dir` (default $SOW_ALIGN_DIR or ./align) and the array is also returned."""
    import os

    import numpy as np

    from .utils import svd_weight

    if not isinstance(module, SoWLinear):
        raise TypeError("Not a SoW layer")
    with torch.no_grad():
        A = torch.cat([w.data for w in module.downscale_weights], dim=1).contiguous()
        B = torch.cat([w.data for w in module.upscale_weights], dim=0).contiguous()
        update = ops.gemm(A, B)
        if module.acc_upweight.numel() != 0:
            weight = ops.gemm(module.acc_downweight.data, module.acc_upweight.data)
        else:
            weight = module.acc_downweight.data
        u_upd, _, _ = svd_weight(update, module.rank)
        u_w, _, _ = svd_weight(weight)
        grid = ops.gemm(u_w.float().contiguous(), u_upd.float().contiguous(), trans_a=True).abs()
        pct = (grid / grid.sum(dim=0)) * 100
    arr = pct.cpu().numpy()
    out_dir = out_dir or os.environ.get("SOW_ALIGN_DIR", "align")
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, export_name + ".npy"), arr)
    return arr
