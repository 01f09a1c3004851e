#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz and *.json from the REFERENCE.

Run in the build container only (the reference lives at /root/reference and never
travels to the GPU box):

    python tests/golden/make_golden.py

The script imports the reference's tn_gradient package unmodified (through the
stub modules of _ref_import.py), drives it on seeded CPU fp32 inputs and stores
inputs + outputs.  The fixtures are DATA (inputs and expected outputs); no
reference source text is stored.  Random re-initialisation draws made inside
SoWLinear.accumulate are captured as inputs (RNG streams are not portable).

Rows of SURVEY.md section 8 covered: a3-a14 (see the `case_*` functions).
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

# driver-only third-party modules imported at the top of scripts/utils/training_utils.py
import importlib.machinery  # noqa: E402

for _name in ("wandb",):
    _m = types.ModuleType(_name)
    _m.__spec__ = importlib.machinery.ModuleSpec(_name, None)
    sys.modules.setdefault(_name, _m)
_loguru = types.ModuleType("loguru")
_loguru.__spec__ = importlib.machinery.ModuleSpec("loguru", None)
_loguru.logger = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
sys.modules.setdefault("loguru", _loguru)

from tn_gradient.layer.sow import SoWLinear  # noqa: E402
from tn_gradient.layer.tensor_linear import TensorTrainLinear  # noqa: E402
from tn_gradient.optimizer.ttadam import TTAdam  # noqa: E402
from tn_gradient.optimizer.ttsgd import TTSGD  # noqa: E402
from tn_gradient.prepare import SoWConfig, accumulate, prepare_sow  # noqa: E402
from tn_gradient.tt import TensorTrain  # noqa: E402
from tn_gradient.utils import closest_factorization, pad_matrix, qr_weight, svd_weight  # noqa: E402

REF_SCRIPTS = os.path.join(_ref_import.REFERENCE_ROOT, "scripts")


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **out)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


class DrawRecorder:
    """Records every tensor filled by torch.nn.init.normal_ while active."""

    def __enter__(self):
        self.draws = []
        self._orig = nn.init.normal_

        def rec(t, *a, **k):
            r = self._orig(t, *a, **k)
            self.draws.append(t.detach().clone())
            return r

        nn.init.normal_ = rec
        return self

    def __exit__(self, *exc):
        nn.init.normal_ = self._orig


def build_layer(d_in, d_out, rank, n_iter, bias, scale, acc, seed, vr_for_lowrank=None):
    """Reference SoWLinear on CPU fp32 with injected factors / accumulator."""
    g = torch.Generator().manual_seed(seed)
    layer = SoWLinear(d_in, d_out, bias=bias, rank=rank, n_iter=n_iter, scale=scale,
                      init_method="normal", init_params=False)
    for i in range(n_iter):
        layer.downscale_weights[i].data = torch.randn(d_in, rank, generator=g) * 0.05
        layer.upscale_weights[i].data = torch.randn(rank, d_out, generator=g) * 0.05
    if bias:
        layer.bias.data = torch.randn(d_out, generator=g) * 0.1
    if acc == "dense":
        layer.acc_downweight = nn.Parameter(torch.randn(d_in, d_out, generator=g) * 0.03, requires_grad=False)
    elif acc == "lowrank":
        vr = vr_for_lowrank or rank
        layer.acc_downweight = nn.Parameter(torch.randn(d_in, vr, generator=g) * 0.1, requires_grad=False)
        layer.acc_upweight = nn.Parameter(torch.randn(vr, d_out, generator=g) * 0.1, requires_grad=False)
    return layer, g


# ----------------------------------------------------------------------------
# a4 / a5: forward + backward
# ----------------------------------------------------------------------------
FWD_CASES = [
    # name, x_shape, d_out, rank, n_iter, bias, scale, acc
    ("cfg1_noacc", (64, 256), 256, 8, 1, False, 1.0, None),
    ("cfg1_bias_dense", (64, 256), 256, 8, 1, True, 0.5, "dense"),
    ("cfg1_lowrank", (64, 256), 256, 8, 1, True, 2.0, "lowrank"),
    ("r50_3d", (3, 37, 96), 160, 50, 1, True, 0.125, None),
    ("r50_3d_dense", (2, 33, 160), 96, 50, 1, False, 1.0, "dense"),
    ("niter2", (5, 16, 72), 40, 6, 2, True, 0.7, "lowrank"),
    ("niter3_odd", (129, 50), 70, 7, 3, False, 1.3, None),
    ("tiny_T1", (1, 24), 8, 4, 1, True, 1.0, "dense"),
]


def case_forward_backward():
    for idx, (name, xs, d_out, rank, n_iter, bias, scale, acc) in enumerate(FWD_CASES):
        d_in = xs[-1]
        layer, g = build_layer(d_in, d_out, rank, n_iter, bias, scale, acc, seed=100 + idx, vr_for_lowrank=2 * rank)
        x = torch.randn(*xs, generator=g, requires_grad=True)
        dy = torch.randn(*xs[:-1], d_out, generator=g)
        y = layer(x)
        y.backward(dy)
        arrs = dict(x=x, dy=dy, y=y, dx=x.grad, scale=np.float64(scale), rank=rank, n_iter=n_iter)
        for i in range(n_iter):
            arrs[f"A{i}"] = layer.downscale_weights[i]
            arrs[f"B{i}"] = layer.upscale_weights[i]
            arrs[f"dA{i}"] = layer.downscale_weights[i].grad
            arrs[f"dB{i}"] = layer.upscale_weights[i].grad
        if bias:
            arrs["bias"], arrs["dbias"] = layer.bias, layer.bias.grad
        if acc is not None:
            arrs["acc_down"] = layer.acc_downweight
            if acc == "lowrank":
                arrs["acc_up"] = layer.acc_upweight
        npz(f"fwdbwd_{name}", **arrs)


# ----------------------------------------------------------------------------
# a6: accumulate traces (both branches), draws captured
# ----------------------------------------------------------------------------
ACC_CASES = [
    # name, d_in, d_out, rank, n_iter, scale, init_method, force_dense, n_calls
    ("lowrank_grow", 48, 40, 8, 1, 1.0, "normal_QR", False, 6),   # vr 8,16,24,32,40 -> dense on the 5th/6th call
    ("niter2_normal", 36, 44, 5, 2, 0.5, "normal", False, 4),
    ("dense_prepare_style", 64, 96, 8, 1, 0.25, "normal_QR", True, 3),  # virtual_rank forced to min(in,out) (prepare.py:120)
    ("cfg1", 256, 256, 8, 1, 1.0, "normal_QR", False, 3),
]


def case_accumulate():
    for idx, (name, d_in, d_out, rank, n_iter, scale, init, force_dense, n_calls) in enumerate(ACC_CASES):
        g = torch.Generator().manual_seed(300 + idx)
        layer = SoWLinear(d_in, d_out, bias=False, rank=rank, n_iter=n_iter, scale=scale,
                          init_method=init, init_params=False)
        if force_dense:
            layer.virtual_rank = min(d_in, d_out)
        arrs = dict(scale=np.float64(scale), rank=rank, n_iter=n_iter, n_calls=n_calls,
                    vr0=layer.virtual_rank)
        torch.manual_seed(900 + idx)
        for c in range(n_calls):
            # fresh "trained" factors before every call (B is zero after accumulate, so inject)
            for i in range(n_iter):
                if c == 0:
                    layer.downscale_weights[i].data = torch.randn(d_in, rank, generator=g) * 0.1
                layer.upscale_weights[i].data = torch.randn(rank, d_out, generator=g) * 0.1
                arrs[f"c{c}_A{i}_in"] = layer.downscale_weights[i].data.clone()
                arrs[f"c{c}_B{i}_in"] = layer.upscale_weights[i].data.clone()
            with DrawRecorder() as rec:
                layer.accumulate()
            for i, d in enumerate(rec.draws):
                arrs[f"c{c}_draw{i}"] = d
            arrs[f"c{c}_vr"] = layer.virtual_rank
            arrs[f"c{c}_acc_down"] = layer.acc_downweight.data.clone()
            arrs[f"c{c}_acc_up"] = layer.acc_upweight.data.clone()
            for i in range(n_iter):
                arrs[f"c{c}_A{i}_out"] = layer.downscale_weights[i].data.clone()
                arrs[f"c{c}_B{i}_out"] = layer.upscale_weights[i].data.clone()
        npz(f"accumulate_{name}", **arrs)


# ----------------------------------------------------------------------------
# a7: qr_weight / svd_weight
# ----------------------------------------------------------------------------
def case_qr_svd():
    g = torch.Generator().manual_seed(7)
    arrs = {}
    mats = {
        "tall": torch.randn(96, 24, generator=g),
        "wide": torch.randn(20, 70, generator=g),
        "square": torch.randn(48, 48, generator=g) * 0.02,
        "gauss002": torch.randn(128, 80, generator=g) * 0.02,
    }
    # adversarial: rank-8 matrix whose leading columns are dependent (unpivoted QR is lossy)
    u = torch.randn(64, 8, generator=g)
    v = torch.randn(8, 64, generator=g)
    v[:, :8] = v[:, :1].repeat(1, 8) * torch.linspace(1.0, 2.0, 8)
    mats["rankdef"] = u @ v
    ranks = {"tall": 8, "wide": 5, "square": 16, "gauss002": 50, "rankdef": 8}
    for k, m in mats.items():
        q, r = qr_weight(m, ranks[k])
        qf, rf = qr_weight(m)
        arrs[f"{k}_in"], arrs[f"{k}_rank"] = m, ranks[k]
        arrs[f"{k}_q"], arrs[f"{k}_r"] = q, r
        arrs[f"{k}_qfull"], arrs[f"{k}_rfull"] = qf, rf
    # bf16 input goes through the fp32 up-cast and back
    mb = (torch.randn(40, 24, generator=g) * 0.02).to(torch.bfloat16)
    qb, rb = qr_weight(mb, 6)
    arrs["bf16_in"] = mb.float()
    arrs["bf16_q"], arrs["bf16_r"] = qb.float(), rb.float()
    u_, s_, v_ = svd_weight(mats["tall"], 6)
    arrs["svd_u"], arrs["svd_s"], arrs["svd_v"] = u_, s_, v_
    npz("qr_svd", **arrs)


# ----------------------------------------------------------------------------
# a8 / a9: prepare_sow replaced-module names, keep-mode tensors, state-dict keys
# ----------------------------------------------------------------------------
def case_prepare():
    from transformers import AutoConfig, AutoModelForCausalLM

    out = {}
    specs = {
        "llama_60m": ("llama_60m.json", ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]),
        "llama_7b": ("llama_7b.json", ["q_proj", "k_proj", "v_proj", "up_proj", "down_proj"]),
        "roberta": ("roberta.json", ["query", "key", "value", "output.dense", "intermediate.dense"]),
    }
    for key, (cfg_file, targets) in specs.items():
        cfg = AutoConfig.from_pretrained(os.path.join(REF_SCRIPTS, "configs", cfg_file))
        with torch.device("meta"):
            model = AutoModelForCausalLM.from_config(cfg)
        all_named = [(n, isinstance(m, nn.Linear)) for n, m in model.named_modules()]
        sow_cfg = SoWConfig(target_modules=targets, rank=8, init_method="normal", decompose=None, device="meta")
        model = prepare_sow(model, sow_cfg)
        replaced = [n for n, m in model.named_modules() if isinstance(m, SoWLinear)]
        shapes = {n: [m.in_features, m.out_features, m.virtual_rank] for n, m in model.named_modules()
                  if isinstance(m, SoWLinear)}
        out[key] = dict(targets=targets, named_modules=all_named, replaced=replaced, shapes=shapes)
        print(key, len(replaced), "layers replaced")
    # state-dict keys of one swapped block (llama_60m, first layer)
    cfg = AutoConfig.from_pretrained(os.path.join(REF_SCRIPTS, "configs", "llama_60m.json"))
    with torch.device("meta"):
        model = AutoModelForCausalLM.from_config(cfg)
    model = prepare_sow(model, SoWConfig(target_modules=specs["llama_60m"][1], rank=50, init_method="normal",
                                         decompose=None, device="meta"))
    out["llama_60m_state_keys_layer0"] = [k for k in model.state_dict().keys() if ".layers.0." in k]
    with open(os.path.join(HERE, "prepare_names.json"), "w") as f:
        json.dump(out, f)
    print("wrote prepare_names.json")

    # keep-mode on a small MLP (CPU): acc_down = W^T, fresh A, B; bias carried over
    torch.manual_seed(11)

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(20, 12, bias=True)
            self.act = nn.Tanh()
            self.fc2 = nn.Linear(12, 6, bias=False)
            self.head = nn.Linear(6, 3)

        def forward(self, x):
            return self.head(self.fc2(self.act(self.fc1(x))))

    m = Tiny()
    w1, b1, w2 = m.fc1.weight.data.clone(), m.fc1.bias.data.clone(), m.fc2.weight.data.clone()
    with DrawRecorder() as rec:
        m = prepare_sow(m, SoWConfig(target_modules=["fc1", "fc2"], rank=4, scale=0.5, init_method="normal",
                                     decompose="keep", device="cpu"))
    x = torch.randn(7, 20)
    npz("prepare_keep",
        w1=w1, b1=b1, w2=w2, x=x, y=m(x),
        head_w=m.head.weight, head_b=m.head.bias,
        fc1_acc_down=m.fc1.acc_downweight, fc1_A=m.fc1.downscale_weights[0], fc1_B=m.fc1.upscale_weights[0],
        fc2_acc_down=m.fc2.acc_downweight, fc2_A=m.fc2.downscale_weights[0], fc2_B=m.fc2.upscale_weights[0],
        fc1_vr=m.fc1.virtual_rank, fc2_vr=m.fc2.virtual_rank)


# ----------------------------------------------------------------------------
# a10: reset_optimizer
# ----------------------------------------------------------------------------
def case_reset_optimizer():
    sys.path.insert(0, REF_SCRIPTS)
    from utils.training_utils import reset_optimizer

    torch.manual_seed(5)
    p0 = nn.Parameter(torch.randn(6, 4))
    p1 = nn.Parameter(torch.randn(4, 3))
    p2 = nn.Parameter(torch.randn(3, 5))
    opt = torch.optim.AdamW([{"params": [p0], "lr": 1e-3}, {"params": [p1, p2], "lr": 1e-2}])
    for _ in range(3):
        for p in (p0, p1, p2):
            p.grad = torch.randn_like(p)
        opt.step()
    before = {f"p{i}_{k}": (v.clone() if torch.is_tensor(v) else torch.tensor(v))
              for i, p in enumerate((p0, p1, p2)) for k, v in opt.state[p].items()}
    reset_optimizer(opt, group_id=1)
    after = {f"p{i}_{k}": (v.clone() if torch.is_tensor(v) else torch.tensor(v))
             for i, p in enumerate((p0, p1, p2)) for k, v in opt.state[p].items()}
    arrs = {f"before_{k}": v for k, v in before.items()}
    arrs.update({f"after_{k}": v for k, v in after.items()})
    npz("reset_optimizer", **arrs)


# ----------------------------------------------------------------------------
# a11: TensorTrain
# ----------------------------------------------------------------------------
def case_tt():
    arrs = {}
    # the tensor of tests/tt_test.py:4,7
    a = torch.arange(2 * 2 * 2 * 3 * 3 * 3).reshape((2, 2, 2, 3, 3, 3)).float()
    tt = TensorTrain.from_tensor(a, [1, 4, 4, 1])
    arrs["t216_in"] = a
    for i, c in enumerate(tt.cores):
        arrs[f"t216_core{i}"] = c
    arrs["t216_rec"] = tt.reconstruct()

    g = torch.Generator().manual_seed(21)
    # padded matrices (from_matrix): 81x81 order 4 (tt_adam_update.py:100-147), 100x60 order 3, 50x37 order 2
    for name, (m, n, ranks) in {"m81": (81, 81, [1, 4, 4, 4, 1]), "m100x60": (100, 60, [1, 5, 5, 1]),
                                "m50x37": (50, 37, [1, 6, 1])}.items():
        mat = torch.randn(m, n, generator=g)
        t = TensorTrain.from_matrix(mat, ranks, padding=True)
        arrs[f"{name}_in"] = mat
        arrs[f"{name}_ranks"] = np.array(ranks)
        for i, c in enumerate(t.cores):
            arrs[f"{name}_core{i}"] = c
        arrs[f"{name}_tomatrix"] = t.to_matrix((m, n))
        arrs[f"{name}_in_shape"] = np.array(t.input_shape)
        arrs[f"{name}_out_shape"] = np.array(t.output_shape)

    # algebra on two order-3 trains with equal bond ranks
    x = torch.randn(3, 3, 3, 4, 4, 4, generator=g)
    y = torch.randn(3, 3, 3, 4, 4, 4, generator=g)
    tx = TensorTrain.from_tensor(x, [1, 5, 5, 1])
    ty = TensorTrain.from_tensor(y, [1, 5, 5, 1])
    arrs["alg_x"], arrs["alg_y"] = x, y
    for i in range(3):
        arrs[f"alg_tx_core{i}"], arrs[f"alg_ty_core{i}"] = tx.cores[i], ty.cores[i]
    s = tx + ty
    p = tx * ty
    d = tx - ty
    for i in range(3):
        arrs[f"alg_add_core{i}"], arrs[f"alg_mul_core{i}"], arrs[f"alg_sub_core{i}"] = s.cores[i], p.cores[i], d.cores[i]
    arrs["alg_add_rec"], arrs["alg_mul_rec"], arrs["alg_sub_rec"] = s.reconstruct(), p.reconstruct(), d.reconstruct()
    for cname, cval in {"pos": 2.5, "neg": -0.75}.items():
        sc = cval * tx
        for i in range(3):
            arrs[f"alg_scale_{cname}_core{i}"] = sc.cores[i]
        arrs[f"alg_scale_{cname}_rec"] = sc.reconstruct()
        arrs[f"alg_scale_{cname}_c"] = cval
    for cname, cval in {"pos": 1.5, "neg": -3.0}.items():
        ac = tx.clone().add_(cval)
        for i in range(3):
            arrs[f"alg_addc_{cname}_core{i}"] = ac.cores[i]
        arrs[f"alg_addc_{cname}_rec"] = ac.reconstruct()
        arrs[f"alg_addc_{cname}_c"] = cval
    arrs["alg_inner_full"] = tx.inner(ty, mode="full")
    arrs["alg_inner_right"] = tx.inner(ty, mode="right")
    arrs["alg_norm_full"] = tx.norm(mode="full")
    rounded = (tx + ty).round([1, 5, 5, 1])
    for i in range(3):
        arrs[f"alg_round_core{i}"] = rounded.cores[i]
    arrs["alg_round_rec"] = rounded.reconstruct()
    arrs["alg_round_ranks"] = np.array(rounded.ranks)
    ortho = tx.orthogonalize(mode="right")
    for i in range(3):
        arrs[f"alg_orthoR_core{i}"] = ortho.cores[i]
    npz("tt_algebra", **arrs)

    # integer work: ceil(n ** (1/d)) sweep and closest_factorization (bit-exact)
    sweep = {}
    for d in (2, 3, 4, 5, 6):
        sweep[str(d)] = [math.ceil(n ** (1 / d)) for n in range(1, 5001)]
    specials = [(3125, 5), (32768, 5), (4096, 3), (4096, 4), (11008, 3), (1376, 3), (512, 3), (768, 3), (3072, 3),
                (50265, 4), (32000, 3), (81, 4), (1024, 5), (243, 5), (7776, 5), (16807, 5), (59049, 5), (1000000, 6)]
    cf = {}
    for n, d in specials + [(n, d) for n in range(2, 400) for d in (2, 3, 4)]:
        r = closest_factorization(n, d)
        cf[f"{n},{d}"] = None if r is None else [list(map(int, r[0])), int(r[1])]
    with open(os.path.join(HERE, "tt_integer.json"), "w") as f:
        json.dump({"ceil_root": sweep,
                   "ceil_root_special": {f"{n},{d}": math.ceil(n ** (1 / d)) for n, d in specials},
                   "closest_factorization": cf}, f)
    print("wrote tt_integer.json")
    # pad_matrix
    pm = pad_matrix(torch.arange(6.0).reshape(2, 3), (4, 5))
    assert pm.shape == (4, 5)


# ----------------------------------------------------------------------------
# a12: TTAdam / TTSGD, a13: TensorTrainLinear
# ----------------------------------------------------------------------------
def case_tt_optim():
    arrs = {}
    g = torch.Generator().manual_seed(33)
    ranks = [1, 4, 4, 4, 1]
    for wd_name, wd in {"nowd": 0.0, "wd": 0.1}.items():
        p = nn.Parameter(torch.randn(81, 81, generator=g) * 0.1)
        arrs[f"adam_{wd_name}_p0"] = p.data.clone()
        opt = TTAdam([{"params": [p], "ranks": ranks}], lr=1e-2, weight_decay=wd)
        for s in range(3):
            grad = torch.randn(81, 81, generator=g)
            arrs[f"adam_{wd_name}_g{s}"] = grad
            p.grad = grad.clone()
            opt.step()
            arrs[f"adam_{wd_name}_p{s + 1}"] = p.data.clone()
        st = opt.state[p]
        for i, c in enumerate(st["exp_avg"].cores):
            arrs[f"adam_{wd_name}_m_core{i}"] = c
        for i, c in enumerate(st["exp_avg_sq"].cores):
            arrs[f"adam_{wd_name}_v_core{i}"] = c
    # dense (no "ranks") TTAdam group behaves like plain Adam w/ this bias correction
    p = nn.Parameter(torch.randn(10, 7, generator=g))
    arrs["adam_dense_p0"] = p.data.clone()
    opt = TTAdam([p], lr=5e-3)
    for s in range(2):
        grad = torch.randn(10, 7, generator=g)
        arrs[f"adam_dense_g{s}"] = grad
        p.grad = grad.clone()
        opt.step()
        arrs[f"adam_dense_p{s + 1}"] = p.data.clone()

    for mom_name, kw in {"mom": dict(momentum=0.9), "nomom": dict(momentum=0.0),
                         "nesterov": dict(momentum=0.8, nesterov=True, dampening=0.1)}.items():
        p = nn.Parameter(torch.randn(81, 81, generator=g) * 0.1)
        arrs[f"sgd_{mom_name}_p0"] = p.data.clone()
        opt = TTSGD([{"params": [p], "ranks": ranks}], lr=1e-2, **kw)
        for s in range(3):
            grad = torch.randn(81, 81, generator=g)
            arrs[f"sgd_{mom_name}_g{s}"] = grad
            p.grad = grad.clone()
            opt.step()
            arrs[f"sgd_{mom_name}_p{s + 1}"] = p.data.clone()
    npz("tt_optim", **arrs)

    torch.manual_seed(44)
    lin = TensorTrainLinear(100, 60, [1, 4, 4, 1], bias=False, type=torch.float32)  # type=None turns cores into strings (tensor_linear.py:30)
    x = torch.randn(5, 7, 100)
    y = lin(x)
    arrs = dict(x=x, y=y)
    for i, c in enumerate(lin.tt.cores):
        arrs[f"core{i}"] = c
    npz("tt_linear", **arrs)


# ----------------------------------------------------------------------------
# a14: caller protocol -- k-step loss trace on a tiny Llama with one accumulate
# ----------------------------------------------------------------------------
def case_train_trace():
    from transformers import AutoModelForCausalLM, LlamaConfig

    sys.path.insert(0, REF_SCRIPTS)
    from utils.training_utils import reset_optimizer

    torch.manual_seed(42)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=256, max_position_embeddings=64, rms_norm_eps=1e-6,
                      tie_word_embeddings=False, attn_implementation="eager")
    model = AutoModelForCausalLM.from_config(cfg)
    targets = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]
    rank = 6
    model = prepare_sow(model, SoWConfig(target_modules=targets, rank=rank, init_method="normal", scale=1.0,
                                         decompose=None, device="cpu"))
    # switch to normal_QR for accumulate()'s re-init (its CPU branch works; only reset_parameters hard-codes cuda)
    for m in model.modules():
        if isinstance(m, SoWLinear):
            m.init_method = "normal_QR"
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}

    special, ids = [], set()
    for n, m in model.named_modules():
        if isinstance(m, SoWLinear):
            for w in list(m.downscale_weights) + list(m.upscale_weights):
                special.append(w)
                ids.add(id(w))
    others = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    opt = torch.optim.AdamW([{"params": others, "lr": 1e-3, "weight_decay": 0.0},
                             {"params": special, "lr": 5e-3, "weight_decay": 0.0}])
    steps, acc_every = 6, 3
    tokens = torch.randint(0, 256, (steps, 4, 16), generator=torch.Generator().manual_seed(42))
    losses, draws, vr_trace = [], [], []
    update_step = 0
    for s in range(steps):
        batch = tokens[s]
        loss = model(input_ids=batch, labels=batch.clone()).loss
        loss.backward()
        losses.append(float(loss))
        # simple_train.py:618-626 with GA == 1, offset 0
        if update_step > 0 and update_step % acc_every == 0:
            with DrawRecorder() as rec:
                accumulate(model)
            draws.append(rec.draws)
            reset_optimizer(opt, group_id=1)
            vr_trace.append([m.virtual_rank for m in model.modules() if isinstance(m, SoWLinear)])
        opt.step()
        opt.zero_grad()
        update_step += 1
    arrs = {f"init::{k}": v for k, v in init_state.items() if v.numel() > 0}
    arrs["tokens"] = tokens
    arrs["losses"] = np.array(losses, dtype=np.float64)
    arrs["acc_every"], arrs["rank"] = acc_every, rank
    for ai, dl in enumerate(draws):
        for li, d in enumerate(dl):
            arrs[f"draw::{ai}::{li}"] = d
    arrs["vr_trace"] = np.array(vr_trace)
    final = model.state_dict()
    probe = "model.layers.1.mlp.down_proj"
    arrs["final::acc_down"] = final[probe + ".acc_downweight"]
    arrs["final::A"] = final[probe + ".downscale_weights.0"]
    arrs["final::B"] = final[probe + ".upscale_weights.0"]
    npz("train_trace", **arrs)
    print("losses", losses)

# ----------------------------------------------------------------------------
# round 2 -- BASELINE configs 4 / 5 caller protocols (run_glue.py:976-1002, finetune.py:39-77) on the RoBERTa-shaped
# trio of tests/protocols.py (768->768, 768->3072, 3072->768, r = 8, fp32, decompose='keep', 256 tokens per step)
# ----------------------------------------------------------------------------
def _protocol_backend():
    sys.path.insert(0, REF_SCRIPTS)
    from utils.training_utils import reset_optimizer
    return types.SimpleNamespace(SoWLinear=SoWLinear, SoWConfig=SoWConfig, prepare_sow=prepare_sow,
                                 reset_optimizer=reset_optimizer)


def case_protocol_traces():
    sys.path.insert(0, os.path.dirname(HERE))
    import protocols as P

    be = _protocol_backend()
    for proto, steps, kw in (("glue", P.TRIO_STEPS, {}), ("finetune", 10, {"ga": 2})):
        model = P.build_trio()
        arrs = {f"digest::{n}": P.tensor_digest(p) for n, p in model.named_parameters()}
        with DrawRecorder() as rec0:
            model = be.prepare_sow(model, be.SoWConfig(target_modules=P.TRIO_TARGETS, rank=P.TRIO_RANK, scale=1.0,
                                                       init_method="normal", decompose="keep", device="cpu"))
        layers = P.sow_layers(model, be)
        assert [n for n, _ in layers] == P.TRIO_NAMES, [n for n, _ in layers]
        arrs["replaced"] = np.array([n for n, _ in layers])
        for li, (_, m) in enumerate(layers):
            arrs[f"init::{li}::A"], arrs[f"init::{li}::B"] = m.downscale_weights[0].data.clone(), m.upscale_weights[0].data.clone()
            assert m.acc_downweight.shape == (m.in_features, m.out_features) and m.virtual_rank == min(m.in_features, m.out_features)
        xs, ts = P.trio_batches(steps)
        arrs["digest::xs"], arrs["digest::ts"] = P.tensor_digest(xs), P.tensor_digest(ts)
        opt = torch.optim.AdamW(P.param_groups(model, be))
        draws, scales = {}, []

        rec = DrawRecorder()

        def on_step(k, mdl, _arrs=arrs, _layers=layers, _steps=steps):
            if k in (0, _steps - 1):
                for li, (_, m) in enumerate(_layers):
                    _arrs[f"grad::{k}::{li}::dA"] = m.downscale_weights[0].grad.clone()
                    _arrs[f"grad::{k}::{li}::dB"] = m.upscale_weights[0].grad.clone()
                    _arrs[f"grad::{k}::{li}::dbias"] = m.bias.grad.clone()

        def on_acc(n, mdl, _arrs=arrs, _layers=layers):
            for li, d in enumerate(rec.draws):
                _arrs[f"draw::{n}::{li}"] = d
            rec.draws.clear()
            for li, (_, m) in enumerate(_layers):
                sample, rows, cols = P.matrix_probe(m.acc_downweight.data)
                _arrs[f"acc::{n}::{li}::sample"], _arrs[f"acc::{n}::{li}::rowsum"], _arrs[f"acc::{n}::{li}::colsum"] = sample, rows, cols
                assert m.acc_upweight.numel() == 0 and float(m.upscale_weights[0].abs().max()) == 0.0
            scales.append([float(m.scale) for _, m in _layers])

        run = P.glue_protocol if proto == "glue" else P.finetune_protocol
        rec.__enter__()              # records the re-initialisation draws of every accumulate() in the loop
        losses = run(model, opt, be, xs, ts, P.TRIO_RANK, P.TRIO_ACC_EVERY, on_accumulate=on_acc, on_step=on_step, **kw)
        rec.__exit__()
        arrs["losses"] = np.array(losses, dtype=np.float64)
        arrs["scales"] = np.array(scales, dtype=np.float64)
        arrs["n_acc"] = len(scales)
        with torch.no_grad():
            y = model(xs[-1])
        arrs["final_y_rows"] = y.reshape(-1, y.shape[-1])[::8].clone()
        for li, (_, m) in enumerate(layers):
            arrs[f"final::{li}::A"], arrs[f"final::{li}::B"] = m.downscale_weights[0].data.clone(), m.upscale_weights[0].data.clone()
            arrs[f"final::{li}::bias"] = m.bias.data.clone()
        print(proto, "losses", losses, "accumulations", len(scales), "scales", scales)
        npz(f"protocol_{proto}", **arrs)


# ----------------------------------------------------------------------------
# round 2 -- activation checkpointing around a `keep` layer (simple_train.py:423, run_glue.py:956, BASELINE config 5):
# forward re-runs inside backward; gradients must equal the plain ones
# ----------------------------------------------------------------------------
def case_checkpoint_layer():
    from torch.utils.checkpoint import checkpoint

    g = torch.Generator().manual_seed(515)
    d_in, d_out, r = 256, 688, 8
    lin = nn.Linear(d_in, d_out, bias=False)
    lin.weight.data = torch.randn(d_out, d_in, generator=g) * 0.03
    holder = nn.Sequential()
    holder.add_module("up_proj", lin)
    holder = prepare_sow(holder, SoWConfig(target_modules=["up_proj"], rank=r, scale=0.5, init_method="normal",
                                           decompose="keep", device="cpu"))
    layer = holder.up_proj
    layer.downscale_weights[0].data = torch.randn(d_in, r, generator=g) * 0.05
    layer.upscale_weights[0].data = torch.randn(r, d_out, generator=g) * 0.05
    x0 = torch.randn(4, 64, d_in, generator=g)
    dy = torch.randn(4, 64, d_out, generator=g)
    out = dict(W=lin.weight.data.clone(), A=layer.downscale_weights[0].data.clone(), B=layer.upscale_weights[0].data.clone(),
               x=x0, dy=dy, scale=np.float64(0.5), rank=r)
    for tag, fn in (("plain", lambda x: layer(torch.tanh(x))),
                    ("ckpt", lambda x: checkpoint(lambda t: layer(torch.tanh(t)), x, use_reentrant=False)),
                    ("ckpt_reentrant", lambda x: checkpoint(lambda t: layer(torch.tanh(t)), x, use_reentrant=True))):
        x = x0.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        y = fn(x)
        y.backward(dy)
        out[f"{tag}_y"], out[f"{tag}_dx"] = y.detach().clone(), x.grad.clone()
        out[f"{tag}_dA"], out[f"{tag}_dB"] = layer.downscale_weights[0].grad.clone(), layer.upscale_weights[0].grad.clone()
    for k in ("y", "dx", "dA", "dB"):
        assert torch.equal(out[f"plain_{k}"], out[f"ckpt_{k}"]) and torch.equal(out[f"plain_{k}"], out[f"ckpt_reentrant_{k}"]), k
    # the three runs are bit-identical in the reference: keep one copy
    for tag in ("ckpt", "ckpt_reentrant"):
        for k in ("y", "dx", "dA", "dB"):
            del out[f"{tag}_{k}"]
    npz("checkpoint_layer", **out)


# ----------------------------------------------------------------------------
# round 2 -- f3: checkpoint round trip (simple_train.py:167-203 save_pretrained; :357-386 + prepare.py:188-215 load_sow;
# commonsense_evaluate.py:268-287 load_state_dict(assign=True)) on the tiny Llama of case_train_trace
# ----------------------------------------------------------------------------
def case_load_sow():
    import shutil
    import tempfile

    from safetensors.torch import load_file
    from transformers import AutoModelForCausalLM, LlamaConfig

    from tn_gradient.prepare import load_sow

    def tiny(decompose=None):
        torch.manual_seed(42)
        cfg = LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=4, vocab_size=256, max_position_embeddings=64, rms_norm_eps=1e-6,
                          tie_word_embeddings=False, attn_implementation="eager")
        m = AutoModelForCausalLM.from_config(cfg)
        return prepare_sow(m, SoWConfig(target_modules=["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj",
                                                        "down_proj"], rank=6, init_method="normal", scale=1.0,
                                        decompose=decompose, device="cpu"))

    model = tiny()
    tokens = torch.randint(0, 256, (4, 4, 16), generator=torch.Generator().manual_seed(43))
    special = [w for m in model.modules() if isinstance(m, SoWLinear) for w in list(m.downscale_weights) + list(m.upscale_weights)]
    ids = {id(w) for w in special}
    opt = torch.optim.AdamW([{"params": [p for p in model.parameters() if p.requires_grad and id(p) not in ids], "lr": 1e-3},
                             {"params": special, "lr": 5e-3}], weight_decay=0.0)
    for s in range(2):
        model(input_ids=tokens[s], labels=tokens[s].clone()).loss.backward()
        opt.step()
        opt.zero_grad()
    accumulate(model)                      # acc_downweight: zero-numel -> [in, out]
    model(input_ids=tokens[2], labels=tokens[2].clone()).loss.backward()   # one more step so B != 0 in the checkpoint
    opt.step()
    opt.zero_grad()
    tmp = tempfile.mkdtemp()
    model.save_pretrained(tmp, max_shard_size="100GB")                      # simple_train.py:176-179
    dst = os.path.join(HERE, "load_sow_checkpoint.safetensors")
    shutil.copy(os.path.join(tmp, "model.safetensors"), dst)
    print(f"wrote load_sow_checkpoint.safetensors ({os.path.getsize(dst) / 1024:.1f} KiB)")
    with torch.no_grad():
        want_loss = float(model(input_ids=tokens[3], labels=tokens[3].clone()).loss)

    fresh = tiny()
    assert fresh.model.layers[0].mlp.up_proj.acc_downweight.numel() == 0
    load_sow(fresh, dst)
    saved = load_file(dst)
    sd = fresh.state_dict()
    shapes = {k: list(v.shape) for k, v in sd.items()}
    for k, v in saved.items():
        assert torch.equal(sd[k], v), k
    with torch.no_grad():
        got_loss = float(fresh(input_ids=tokens[3], labels=tokens[3].clone()).loss)
    assert got_loss == want_loss, (got_loss, want_loss)
    # load_state_dict(assign=True) path of commonsense_evaluate.py:268-282: the model is prepared with the DEFAULT
    # decompose='keep', so acc_downweight already is [in, out] (with zero-numel accumulators load_state_dict raises a size
    # mismatch even with assign=True -- recorded below)
    try:
        tiny().load_state_dict(saved, assign=True, strict=False)
        assign_on_empty = "ok"
    except RuntimeError as e:
        assign_on_empty = "RuntimeError: size mismatch" if "size mismatch" in str(e) else "RuntimeError"
    fresh2 = tiny(decompose="keep")
    res = fresh2.load_state_dict(saved, assign=True, strict=False)
    with torch.no_grad():
        loss2 = float(fresh2(input_ids=tokens[3], labels=tokens[3].clone()).loss)
    npz("load_sow", tokens=tokens, next_loss=np.float64(want_loss), assign_loss=np.float64(loss2),
        n_saved=len(saved))
    with open(os.path.join(HERE, "load_sow_meta.json"), "w") as f:
        json.dump({"state_shapes": shapes, "saved_keys": sorted(saved.keys()),
                   "assign_on_empty_accumulator": assign_on_empty, "assign_missing": list(res.missing_keys), "assign_unexpected": list(res.unexpected_keys),
                   "requires_grad_after_load": {k: bool(p.requires_grad) for k, p in fresh.named_parameters()}}, f)
    shutil.rmtree(tmp)
    print("load_sow: next-step loss", want_loss, "assign loss", loss2)


# ----------------------------------------------------------------------------
# round 2 -- a11: TensorTrain.sqrt / sqrtinv / reciprocal (tt.py:279-341, 480-494; tests/tt_test.py:1-13 prints exactly
# the first case)
# ----------------------------------------------------------------------------
def case_tt_newton():
    arrs = {}
    a = torch.arange(2 * 2 * 2 * 3 * 3 * 3).reshape((2, 2, 2, 3, 3, 3)).float()      # tests/tt_test.py:4
    tta = TensorTrain.from_tensor(a, [1, 4, 4, 1])
    arrs["t216_sqrt_rec"] = tta.sqrt().reconstruct()
    arrs["t216_dense_sqrt"] = a.sqrt()
    g = torch.Generator().manual_seed(61)
    # a positive, well-conditioned rank-2 tensor: elementwise values in [1, 3]
    u = [torch.rand(n, 2, generator=g) * 0.5 + 0.75 for n in (2, 2, 2, 3, 3, 3)]
    pos = torch.einsum("az,bz,cz,dz,ez,fz->abcdef", *u)
    arrs["pos_in"] = pos
    tp = TensorTrain.from_tensor(pos, [1, 3, 3, 1])
    for i, c in enumerate(tp.cores):
        arrs[f"pos_core{i}"] = c
    for name, fn in (("sqrt", lambda t: t.sqrt()), ("sqrt_it2", lambda t: t.sqrt(max_iter=2)),
                     ("sqrtinv", lambda t: t.sqrtinv()), ("sqrtinv_it2", lambda t: t.sqrtinv(threshold=None, max_iter=2))):
        out = fn(tp)
        arrs[f"pos_{name}_rec"] = out.reconstruct()
        arrs[f"pos_{name}_ranks"] = np.array(out.ranks)
        print(name, "ranks", out.ranks, "max", float(out.reconstruct().abs().max()))
    cores = [torch.randn(1, 2, 3, 4, generator=g), torch.randn(4, 2, 3, 4, generator=g) + 2 * torch.eye(4)[:, None, None, :],
             torch.randn(4, 2, 3, 1, generator=g)]
    rec = TensorTrain.from_cores([c.clone() for c in cores]).reciprocal()
    for i in range(3):
        arrs[f"recip_in{i}"], arrs[f"recip_out{i}"] = cores[i], rec.cores[i]
    npz("tt_newton", **arrs)




# ----------------------------------------------------------------------------
# round 2 -- a14: simple_train.py:596-650 with gradient accumulation 3 (the predicate of :618-626 fires on two micro-steps
# of every accumulation update and never on its last one)
# ----------------------------------------------------------------------------
def case_train_trace_ga():
    from transformers import AutoModelForCausalLM, LlamaConfig
    sys.path.insert(0, os.path.dirname(HERE))
    import protocols as P

    be = _protocol_backend()
    be.accumulate = accumulate
    torch.manual_seed(42)
    model = AutoModelForCausalLM.from_config(LlamaConfig(**P.LLAMA_TINY))
    rank = 6
    model = prepare_sow(model, SoWConfig(target_modules=P.LLAMA_TARGETS, rank=rank, init_method="normal", scale=1.0,
                                         decompose=None, device="cpu"))
    for m in model.modules():
        if isinstance(m, SoWLinear):
            m.init_method = "normal_QR"
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(P.llama_param_groups(model, be))
    ga, sow_acc, micro = 3, 1, 15
    tokens = torch.randint(0, 256, (micro, 4, 16), generator=torch.Generator().manual_seed(44))
    rec = DrawRecorder()
    draws = []

    def on_acc(n, mdl):
        draws.append(list(rec.draws))
        rec.draws.clear()

    rec.__enter__()
    losses, fired = P.pretrain_protocol(model, opt, be, tokens, ga, sow_acc, on_accumulate=on_acc)
    rec.__exit__()
    arrs = {f"init::{k}": v for k, v in init_state.items() if v.numel() > 0}
    arrs.update(tokens=tokens, losses=np.array(losses, dtype=np.float64), fired=np.array(fired), ga=ga, sow_accumulation=sow_acc,
                rank=rank)
    for ai, dl in enumerate(draws):
        for li, d in enumerate(dl):
            arrs[f"draw::{ai}::{li}"] = d
    final = model.state_dict()
    probe = "model.layers.1.mlp.down_proj"
    arrs["final::acc_down"] = final[probe + ".acc_downweight"]
    arrs["final::A"] = final[probe + ".downscale_weights.0"]
    arrs["final::B"] = final[probe + ".upscale_weights.0"]
    npz("train_trace_ga3", **arrs)
    print("ga3 losses", losses, "accumulate fired at global steps", fired)

CASES = dict(forward_backward=case_forward_backward, accumulate=case_accumulate, qr_svd=case_qr_svd, prepare=case_prepare,
             reset_optimizer=case_reset_optimizer, tt=case_tt, tt_optim=case_tt_optim, train_trace=case_train_trace,
             protocol_traces=case_protocol_traces, checkpoint_layer=case_checkpoint_layer, load_sow=case_load_sow,
             tt_newton=case_tt_newton, train_trace_ga=case_train_trace_ga)

if __name__ == "__main__":
    torch.set_num_threads(4)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        CASES[n]()
    print("done")
