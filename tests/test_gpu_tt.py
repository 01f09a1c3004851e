"""GPU parity of the TT path (TensorTrain, TTAdam, TTSGD, TensorTrainLinear) and of the caller
protocol (prepare_sow + accumulate + reset_optimizer on a tiny Llama) against the golden vectors."""
import json
import os

import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TT_TOL = 1e-4  # QR-based cores: GPU Householder vs LAPACK summation order


def _cores(g, prefix, n):
    return [g[f"{prefix}_core{i}"] for i in range(n)]


def test_tt_decompose_reconstruct():
    from sow_amd import TensorTrain
    g = load_golden("tt_algebra")
    # tests/tt_test.py's arange tensor has unfolding rank 2 < TT rank 4: Q columns 3-4 are spanned by
    # rounding noise in every implementation, so compare the two well-defined columns, the shapes and
    # the reconstruction (which does not depend on the arbitrary columns)
    tt = TensorTrain.from_tensor(g["t216_in"].to(DEV), [1, 4, 4, 1])
    for c, w in zip(tt.cores, _cores(g, "t216", 3)):
        assert tuple(c.shape) == tuple(w.shape)
    assert rel_err(tt.cores[0][..., :2].cpu(), g["t216_core0"][..., :2]) < TT_TOL
    assert rel_err(tt.reconstruct().cpu(), g["t216_rec"]) < TT_TOL
    assert rel_err(tt.reconstruct().cpu(), g["t216_in"]) < TT_TOL
    for name, order in (("m81", 4), ("m100x60", 3), ("m50x37", 2)):
        ranks = [int(r) for r in g[f"{name}_ranks"]]
        t = TensorTrain.from_matrix(g[f"{name}_in"].to(DEV), ranks, padding=True)
        assert list(t.input_shape) == [int(v) for v in g[f"{name}_in_shape"]]
        for c, w in zip(t.cores, _cores(g, name, order)):
            assert tuple(c.shape) == tuple(w.shape) and rel_err(c.cpu(), w) < TT_TOL
        assert rel_err(t.to_matrix(g[f"{name}_in"].shape).cpu(), g[f"{name}_tomatrix"]) < TT_TOL


def test_tt_algebra():
    from sow_amd import TensorTrain
    g = load_golden("tt_algebra")
    tx = TensorTrain.from_cores([c.to(DEV) for c in _cores(g, "alg_tx", 3)])
    ty = TensorTrain.from_cores([c.to(DEV) for c in _cores(g, "alg_ty", 3)])
    tx.ranks, ty.ranks = [1, 5, 5, 1], [1, 5, 5, 1]
    for c, w in zip((tx + ty).cores, _cores(g, "alg_add", 3)):
        assert torch.equal(c.cpu(), w)
    for c, w in zip((tx * ty).cores, _cores(g, "alg_mul", 3)):
        assert rel_err(c.cpu(), w) < 1e-5
    assert rel_err((tx * ty).reconstruct().cpu(), g["alg_mul_rec"]) < 1e-5
    for c, w in zip((tx - ty).cores, _cores(g, "alg_sub", 3)):
        assert rel_err(c.cpu(), w) < 1e-5
    for cname in ("pos", "neg"):
        cval = float(g[f"alg_scale_{cname}_c"])
        for a, w in zip((cval * tx).cores, _cores(g, f"alg_scale_{cname}", 3)):
            assert rel_err(a.cpu(), w) < 1e-5
        cval = float(g[f"alg_addc_{cname}_c"])
        for a, w in zip(tx.clone().add_(cval).cores, _cores(g, f"alg_addc_{cname}", 3)):
            assert rel_err(a.cpu(), w) < 1e-5
    for mode, key in (("full", "alg_inner_full"), ("right", "alg_inner_right")):
        want = float(g[key])
        assert abs(tx.inner(ty, mode=mode) - want) < 1e-4 * abs(want) + 1e-3
    assert abs(tx.norm(mode="full") - float(g["alg_norm_full"])) < 1e-4 * abs(float(g["alg_norm_full"]))
    rounded = (tx + ty).round([1, 5, 5, 1])
    assert list(rounded.ranks) == [int(r) for r in g["alg_round_ranks"]]
    assert rel_err(rounded.reconstruct().cpu(), g["alg_round_rec"]) < 5e-4
    ortho = tx.orthogonalize(mode="right")
    for a, w in zip(ortho.cores, _cores(g, "alg_orthoR", 3)):
        assert rel_err(a.cpu(), w) < TT_TOL


def test_tt_reciprocal_and_newton_helpers():
    """TensorTrain.reciprocal (tt.py:480-494) inverts the middle cores' [:, i, j, :] slices; sqrt/sqrtinv
    (tt.py:279-341, print-only in the reference's tests/tt_test.py) run on the device kernels."""
    from sow_amd import TensorTrain, ops
    gen = torch.Generator().manual_seed(8)
    cores = [torch.randn(1, 2, 3, 4, generator=gen), torch.randn(4, 2, 3, 4, generator=gen) + 2 * torch.eye(4)[:, None, None, :],
             torch.randn(4, 2, 3, 1, generator=gen)]
    tt = TensorTrain.from_cores([c.to(DEV) for c in cores])
    rec = tt.reciprocal()
    assert torch.equal(rec.cores[0].cpu(), cores[0]) and torch.equal(rec.cores[2].cpu(), cores[2])
    for i in range(2):
        for j in range(3):
            want = torch.linalg.inv(cores[1][:, i, j, :])
            assert rel_err(rec.cores[1][:, i, j, :].cpu(), want) < 1e-4
    assert abs(ops.absmax(cores[1].to(DEV)) - float(cores[1].abs().max())) < 1e-6
    a = (torch.arange(2 * 2 * 2 * 3 * 3 * 3).reshape(2, 2, 2, 3, 3, 3).float() + 1.0).to(DEV)   # tests/tt_test.py:4
    tta = TensorTrain.from_tensor(a, [1, 4, 4, 1])
    out = tta.sqrt().reconstruct()
    assert out.shape == a.shape and torch.isfinite(out).all()


def test_tt_integer_bit_exact():
    """ceil(n ** (1/d)) and closest_factorization are host integer work: bit-exact."""
    from sow_amd import closest_factorization
    from sow_amd.tensor_linear import TensorTrainLinear
    with open(os.path.join(GOLDEN, "tt_integer.json")) as f:
        J = json.load(f)
    for key, v in J["closest_factorization"].items():
        n, d = map(int, key.split(","))
        r = closest_factorization(n, d)
        assert (None if r is None else [list(r[0]), r[1]]) == v
    lin = TensorTrainLinear(3125, 32768, [1, 2, 2, 2, 2, 1], bias=False, device="cpu", type=torch.float32)
    assert lin.in_core_features == J["ceil_root_special"]["3125,5"] == 6
    assert lin.out_core_features == J["ceil_root_special"]["32768,5"] == 9


def test_tt_optimizers():
    from sow_amd import TTAdam, TTSGD
    g = load_golden("tt_optim")
    ranks = [1, 4, 4, 4, 1]
    for wd_name, wd in (("nowd", 0.0), ("wd", 0.1)):
        p = nn.Parameter(g[f"adam_{wd_name}_p0"].to(DEV))
        opt = TTAdam([{"params": [p], "ranks": ranks}], lr=1e-2, weight_decay=wd)
        for s in range(3):
            p.grad = g[f"adam_{wd_name}_g{s}"].to(DEV)
            opt.step()
            assert rel_err(p.data.cpu(), g[f"adam_{wd_name}_p{s + 1}"]) < 1e-4
        assert [tuple(c.shape) for c in opt.state[p]["exp_avg"].cores] == [(1, 3, 3, 4), (4, 3, 3, 4), (4, 3, 3, 4), (4, 3, 3, 1)]
    p = nn.Parameter(g["adam_dense_p0"].to(DEV))
    opt = TTAdam([p], lr=5e-3)
    for s in range(2):
        p.grad = g[f"adam_dense_g{s}"].to(DEV)
        opt.step()
        assert rel_err(p.data.cpu(), g[f"adam_dense_p{s + 1}"]) < 1e-5
    for name, kw in (("mom", dict(momentum=0.9)), ("nomom", dict(momentum=0.0)),
                     ("nesterov", dict(momentum=0.8, nesterov=True, dampening=0.1))):
        p = nn.Parameter(g[f"sgd_{name}_p0"].to(DEV))
        opt = TTSGD([{"params": [p], "ranks": ranks}], lr=1e-2, **kw)
        for s in range(3):
            p.grad = g[f"sgd_{name}_g{s}"].to(DEV)
            opt.step()
            assert rel_err(p.data.cpu(), g[f"sgd_{name}_p{s + 1}"]) < 1e-4


def test_tt_linear():
    from sow_amd import TensorTrainLinear
    g = load_golden("tt_linear")
    lin = TensorTrainLinear(100, 60, [1, 4, 4, 1], bias=False, device=DEV, type=torch.float32)
    for c, w in zip(lin.tt.cores, (g["core0"], g["core1"], g["core2"])):
        c.data = w.to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    y = lin(x)
    assert y.shape == g["y"].shape and rel_err(y.detach().cpu(), g["y"]) < 1e-5
    y.sum().backward()  # gradients flow to the cores through the GEMM autograd function
    assert all(c.grad is not None and torch.isfinite(c.grad).all() for c in lin.tt.cores)


def test_prepare_names_and_keys():
    """a8: the set of replaced module names and the state-dict keys are bit-exact (index work)."""
    transformers = pytest.importorskip("transformers")
    from sow_amd import SoWConfig, SoWLinear, prepare_sow
    with open(os.path.join(GOLDEN, "prepare_names.json")) as f:
        P = json.load(f)
    cfg = transformers.LlamaConfig(hidden_size=512, intermediate_size=1376, num_hidden_layers=8, num_attention_heads=8,
                                   vocab_size=32000, max_position_embeddings=1024, rms_norm_eps=1e-6,
                                   tie_word_embeddings=False)
    with torch.device("meta"):
        model = transformers.AutoModelForCausalLM.from_config(cfg)
    assert [[n, isinstance(m, nn.Linear)] for n, m in model.named_modules()] == P["llama_60m"]["named_modules"]
    model = prepare_sow(model, SoWConfig(target_modules=P["llama_60m"]["targets"], rank=50, init_method="normal",
                                         decompose=None, device="meta"))
    got = [n for n, m in model.named_modules() if isinstance(m, SoWLinear)]
    assert got == P["llama_60m"]["replaced"] and len(got) == 56
    assert [k for k in model.state_dict().keys() if ".layers.0." in k] == P["llama_60m_state_keys_layer0"]
    assert all(m.virtual_rank == min(m.in_features, m.out_features) for m in model.modules() if isinstance(m, SoWLinear))


def test_prepare_keep_and_qr_modes():
    from sow_amd import SoWConfig, prepare_sow
    g = load_golden("prepare_keep")

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
    m.fc1.weight.data, m.fc1.bias.data, m.fc2.weight.data = g["w1"], g["b1"], g["w2"]
    m.head.weight.data, m.head.bias.data = g["head_w"], g["head_b"]
    m = prepare_sow(m, SoWConfig(target_modules=["fc1", "fc2"], rank=4, scale=0.5, init_method="normal",
                                 decompose="keep", device=DEV))
    m.head.to(DEV)
    assert torch.equal(m.fc1.acc_downweight.cpu(), g["fc1_acc_down"])
    m.fc1.downscale_weights[0].data, m.fc1.upscale_weights[0].data = g["fc1_A"].to(DEV), g["fc1_B"].to(DEV)
    m.fc2.downscale_weights[0].data, m.fc2.upscale_weights[0].data = g["fc2_A"].to(DEV), g["fc2_B"].to(DEV)
    m.fc1.bias.data = m.fc1.bias.data.to(DEV)
    assert rel_err(m(g["x"].to(DEV)).detach().cpu(), g["y"]) < 1e-5
    assert m.fc1.virtual_rank == int(g["fc1_vr"]) and m.fc2.virtual_rank == int(g["fc2_vr"])
    # 'qr' mode: W_acc + A.B == W^T
    lin = nn.Sequential()
    lin.add_module("proj", nn.Linear(40, 24, bias=False))
    w = lin.proj.weight.data.clone()
    lin = prepare_sow(lin, SoWConfig(target_modules=["proj"], rank=4, init_method="normal", decompose="qr", device=DEV))
    rec = lin.proj.acc_downweight.data + lin.proj.downscale_weights[0].data @ lin.proj.upscale_weights[0].data
    assert rel_err(rec.cpu(), w.t()) < 1e-5


def test_train_trace_tiny_llama():
    """a14: caller protocol of simple_train.py:596-650 on a tiny Llama -- loss trace incl. one
    accumulate + reset_optimizer, with the reference's initial weights and re-init draws."""
    transformers = pytest.importorskip("transformers")
    from sow_amd import SoWConfig, SoWLinear, accumulate, prepare_sow, reset_optimizer
    g = load_golden("train_trace")
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=256, max_position_embeddings=64, rms_norm_eps=1e-6,
                                   tie_word_embeddings=False, attn_implementation="eager")
    model = transformers.AutoModelForCausalLM.from_config(cfg)
    rank = int(g["rank"])
    model = prepare_sow(model, SoWConfig(target_modules=["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj",
                                                         "down_proj"], rank=rank, init_method="normal", scale=1.0,
                                         decompose=None, device="cpu"))
    sd = {k[len("init::"):]: v for k, v in g.items() if k.startswith("init::")}
    missing = model.load_state_dict(sd, strict=False)
    assert all(k.endswith("acc_upweight") or k.endswith("acc_downweight") for k in missing.missing_keys)
    model.to(DEV)
    layers = [m for m in model.modules() if isinstance(m, SoWLinear)]
    for m in layers:
        m.init_method = "normal_QR"
    special, ids = [], set()
    for m in layers:
        for w in list(m.downscale_weights) + list(m.upscale_weights):
            special.append(w)
            ids.add(id(w))
    others = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    opt = torch.optim.AdamW([{"params": others, "lr": 1e-3, "weight_decay": 0.0},
                             {"params": special, "lr": 5e-3, "weight_decay": 0.0}])
    tokens = g["tokens"].to(DEV)
    acc_every = int(g["acc_every"])
    losses, update_step, acc_idx = [], 0, 0
    for s in range(tokens.shape[0]):
        loss = model(input_ids=tokens[s], labels=tokens[s].clone()).loss
        loss.backward()
        losses.append(float(loss))
        if update_step > 0 and update_step % acc_every == 0:
            draws = iter([g[f"draw::{acc_idx}::{li}"].to(DEV) for li in range(len(layers))])
            for m in layers:
                m._fresh_gaussian = lambda shape, device, dtype, _d=draws: next(_d).to(dtype)
            accumulate(model)
            reset_optimizer(opt, group_id=1)
            assert [m.virtual_rank for m in layers] == [int(v) for v in g["vr_trace"][acc_idx]]
            acc_idx += 1
        opt.step()
        opt.zero_grad()
        update_step += 1
    want = g["losses"]
    for got, w in zip(losses, want.tolist()):
        assert abs(got - w) < 2e-4 * abs(w), (losses, want)
    sdf = model.state_dict()
    probe = "model.layers.1.mlp.down_proj"
    assert rel_err(sdf[probe + ".acc_downweight"].cpu(), g["final::acc_down"]) < 1e-3
