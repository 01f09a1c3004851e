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
    current stream: 56 layers take ~7 QR latencies instead of 56.  Host order is unchanged, so the Gaussian
    re-initialisation draws come out of the generator exactly as in the sequential loop."""
    mods = [m for _, m in model.named_modules() if isinstance(m, SoWLinear)]
    # layers attached to a FactorBucket (sow_amd/dp.py): pending partial sums belong to the OLD factors -- reduce them
    # first; afterwards the rebound .data tensors are copied back into the flat buffer so the fused optimizer and the
    # all-reduce keep seeing them
    buckets = {id(s.bucket): s.bucket for s in (getattr(m, "_grad_sink", None) for m in mods) if s is not None}
    for b in buckets.values():
        b.finalize()
    _accumulate_layers(mods)
    for b in buckets.values():
        b.rebind()


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
