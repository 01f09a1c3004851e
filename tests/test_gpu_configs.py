"""-m gpu: BASELINE configs 2, 4 and 5 at their OWN shapes and caller protocols, through the module-swap surface and the
C ABI, against reference-generated fixtures (tests/golden/protocol_*.npz, checkpoint_layer.npz, load_sow*.{npz,json,
safetensors}) and against the CPU oracle.

  config 4  roberta-base, r = 8, fp32, decompose='keep', periodic accumulate + scale -> 1/rank (run_glue.py:976-1002)
  config 5  llama-7b, r = 8, bf16, T = 4 * 256, 'keep', activation checkpointing (finetune.py:39-77, :292-312)
  config 2  llama_60m, r = 50, bf16, T = 32768: the three layer shapes, with and without the dense accumulator

Tolerances: fp32 1e-5 relative to the largest reference magnitude (north_star), 2e-5 for the T-long weight-gradient sums
(summation order); bf16 2e-2; names / counters / schedules bit-exact.
"""
import json
import os
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

import protocols as P
from conftest import GOLDEN, load_golden, rel_err
from oracle import sow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5


def _backend():
    import sow_amd
    return types.SimpleNamespace(SoWLinear=sow_amd.SoWLinear, SoWConfig=sow_amd.SoWConfig, prepare_sow=sow_amd.prepare_sow,
                                 reset_optimizer=sow_amd.reset_optimizer)


def _set_draw(m, draw):
    m._fresh_gaussian = lambda shape, device, dtype, _d=draw: _d.to(device, dtype)


@pytest.mark.parametrize("proto", ["glue", "finetune"])
def test_protocol_trace_config4(proto):
    """Config 4 / config 5 caller protocols on the RoBERTa-shaped trio (768 -> 768, 768 -> 3072, 3072 -> 768; r = 8; fp32;
    keep): loss trace, factor / bias gradients, dense accumulator after every accumulate(), scale -> 1/rank, final factors
    -- all against the reference's own run (tests/golden/make_golden.py::case_protocol_traces)."""
    g = load_golden("protocol_" + proto)
    errs = P.replay_and_check(_backend(), g, proto, DEV, _set_draw, dict(loss=2e-6, grad=TOL, acc=TOL, final=TOL))
    assert len(errs) > 20


def test_checkpointed_keep_layer():
    """Config 5 runs under gradient checkpointing (simple_train.py:423, run_glue.py:956): the forward is re-run inside
    backward.  Gradients match the reference fixture and are BIT-identical to the un-checkpointed ones (deterministic
    kernels, stateless autograd Function) in both checkpoint modes, fp32 and bf16."""
    from torch.utils.checkpoint import checkpoint

    from sow_amd import SoWConfig, prepare_sow
    g = load_golden("checkpoint_layer")
    for dtype in (torch.float32, torch.bfloat16):
        lin = nn.Linear(g["W"].shape[1], g["W"].shape[0], bias=False)
        lin.weight.data = g["W"].clone()
        holder = nn.Sequential()
        holder.add_module("up_proj", lin.to(dtype))
        holder = prepare_sow(holder, SoWConfig(target_modules=["up_proj"], rank=int(g["rank"]), scale=float(g["scale"]),
                                               init_method="normal", decompose="keep", device=DEV))
        layer = holder.up_proj
        layer.downscale_weights[0].data = g["A"].to(DEV, dtype)
        layer.upscale_weights[0].data = g["B"].to(DEV, dtype)
        runs = {}
        for tag, fn in (("plain", lambda x: layer(torch.tanh(x))),
                        ("ckpt", lambda x: checkpoint(lambda t: layer(torch.tanh(t)), x, use_reentrant=False)),
                        ("ckpt_reentrant", lambda x: checkpoint(lambda t: layer(torch.tanh(t)), x, use_reentrant=True))):
            x = g["x"].to(DEV, dtype).requires_grad_(True)
            for p in layer.parameters():
                p.grad = None
            y = fn(x)
            y.backward(g["dy"].to(DEV, dtype))
            runs[tag] = (y.detach().clone(), x.grad.clone(), layer.downscale_weights[0].grad.clone(),
                         layer.upscale_weights[0].grad.clone())
        for tag in ("ckpt", "ckpt_reentrant"):
            for a, b in zip(runs["plain"], runs[tag]):
                assert torch.equal(a, b), (tag, dtype)
        tol = TOL if dtype == torch.float32 else 2e-2
        for got, key in zip(runs["plain"], ("y", "dx", "dA", "dB")):
            assert rel_err(got.float().cpu(), g[f"plain_{key}"]) < tol, (key, dtype)


# ---------------------------------------------------------------------------------------------
# f3: checkpoint round trip
# ---------------------------------------------------------------------------------------------
def _tiny_llama(decompose=None):
    transformers = pytest.importorskip("transformers")
    from sow_amd import SoWConfig, prepare_sow
    torch.manual_seed(42)
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=256, max_position_embeddings=64, rms_norm_eps=1e-6,
                                   tie_word_embeddings=False, attn_implementation="eager")
    model = transformers.AutoModelForCausalLM.from_config(cfg)
    return prepare_sow(model, SoWConfig(target_modules=["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj",
                                                        "down_proj"], rank=6, init_method="normal", scale=1.0,
                                        decompose=decompose, device="cpu"))


def test_load_sow_reference_checkpoint():
    """A checkpoint WRITTEN BY THE REFERENCE (save_pretrained after accumulate(), simple_train.py:176-179) loads into a
    fresh sow_amd model through load_sow (prepare.py:188-215): zero-numel accumulators are replaced by [in, out] frozen
    Parameters, everything else copied; state dict, requires_grad flags and the next-step loss equal the reference's."""
    from safetensors.torch import load_file

    from sow_amd import load_sow
    path = os.path.join(GOLDEN, "load_sow_checkpoint.safetensors")
    g = load_golden("load_sow")
    with open(os.path.join(GOLDEN, "load_sow_meta.json")) as f:
        meta = json.load(f)
    model = _tiny_llama()
    assert model.model.layers[0].mlp.up_proj.acc_downweight.numel() == 0
    model.to(DEV)
    load_sow(model, path)
    saved = load_file(path)
    sd = model.state_dict()
    assert sorted(saved.keys()) == meta["saved_keys"] and {k: list(v.shape) for k, v in sd.items()} == meta["state_shapes"]
    for k, v in saved.items():
        assert sd[k].is_cuda and torch.equal(sd[k].cpu(), v), k
    assert {k: bool(p.requires_grad) for k, p in model.named_parameters()} == meta["requires_grad_after_load"]
    tokens = g["tokens"].to(DEV)
    with torch.no_grad():
        loss = float(model(input_ids=tokens[3], labels=tokens[3].clone()).loss)
    assert abs(loss - float(g["next_loss"])) < 2e-5 * abs(float(g["next_loss"]))
    # commonsense_evaluate.py:268-282: default decompose='keep' + load_state_dict(assign=True)
    with pytest.raises(RuntimeError, match="size mismatch"):      # the reference raises the same on empty accumulators
        _tiny_llama().load_state_dict(saved, assign=True, strict=False)
    m2 = _tiny_llama(decompose="keep")
    res = m2.load_state_dict(saved, assign=True, strict=False)
    assert list(res.missing_keys) == meta["assign_missing"] and list(res.unexpected_keys) == meta["assign_unexpected"]
    m2.to(DEV)
    with torch.no_grad():
        loss2 = float(m2(input_ids=tokens[3], labels=tokens[3].clone()).loss)
    assert abs(loss2 - float(g["assign_loss"])) < 2e-5 * abs(float(g["assign_loss"]))


def test_save_pretrained_round_trip_after_gpu_accumulate(tmp_path):
    """prepare_sow -> steps -> accumulate() on the GPU -> save_pretrained (safetensors) -> fresh model -> load_sow:
    identical state dict (zero-numel -> [in, out] transition included) and bit-identical next-step loss; resuming the
    optimizer from its state_dict reproduces the next parameter update exactly."""
    from safetensors.torch import load_file

    from sow_amd import SoWLinear, accumulate, load_sow, reset_optimizer

    def groups(model):
        special = [w for m in model.modules() if isinstance(m, SoWLinear) for w in list(m.downscale_weights) + list(m.upscale_weights)]
        ids = {id(w) for w in special}
        return [{"params": [p for p in model.parameters() if p.requires_grad and id(p) not in ids], "lr": 1e-3},
                {"params": special, "lr": 5e-3}]

    model = _tiny_llama().to(DEV)
    opt = torch.optim.AdamW(groups(model), weight_decay=0.0)
    tokens = torch.randint(0, 256, (5, 4, 16), generator=torch.Generator().manual_seed(9)).to(DEV)
    for s in range(3):
        model(input_ids=tokens[s], labels=tokens[s].clone()).loss.backward()
        if s == 1:
            accumulate(model)
            reset_optimizer(opt, group_id=1)
        opt.step()
        opt.zero_grad()
    model.save_pretrained(str(tmp_path), max_shard_size="100GB")
    torch.save(opt.state_dict(), str(tmp_path / "optimizer.pt"))
    saved = load_file(str(tmp_path / "model.safetensors"))
    assert saved["model.layers.0.mlp.up_proj.acc_downweight"].shape == (64, 176)
    assert saved["model.layers.0.mlp.up_proj.acc_upweight"].numel() == 0

    fresh = _tiny_llama().to(DEV)
    load_sow(fresh, str(tmp_path / "model.safetensors"))
    sd_a, sd_b = model.state_dict(), fresh.state_dict()
    assert list(sd_a.keys()) == list(sd_b.keys())
    for k in sd_a:
        assert sd_a[k].shape == sd_b[k].shape and sd_a[k].device == sd_b[k].device and torch.equal(sd_a[k], sd_b[k]), k
    opt2 = torch.optim.AdamW(groups(fresh), weight_decay=0.0)
    opt2.load_state_dict(torch.load(str(tmp_path / "optimizer.pt"), weights_only=True))
    for net, o in ((model, opt), (fresh, opt2)):
        loss = net(input_ids=tokens[3], labels=tokens[3].clone()).loss
        loss.backward()
        o.step()
        o.zero_grad()
        net._probe_loss = float(loss)
    assert model._probe_loss == fresh._probe_loss
    for (k, a), (_, b) in zip(model.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k


# ---------------------------------------------------------------------------------------------
# full config shapes against the oracle
# ---------------------------------------------------------------------------------------------
def _layer_case(T, d_in, d_out, r, dtype, dense, seed, scale):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(T, d_in, generator=gen).to(dtype)
    dy = torch.randn(T, d_out, generator=gen).to(dtype)
    A = torch.linalg.qr(torch.randn(d_in, r, generator=gen))[0].contiguous().to(dtype)
    B = (torch.randn(r, d_out, generator=gen) * 0.02).to(dtype)
    W = (torch.randn(d_in, d_out, generator=gen) * 0.02).to(dtype) if dense else None
    return x, dy, A, B, W


def _check_layer(T, d_in, d_out, r, dtype, dense, seed, scale=0.125, rows=None):
    """HIP forward + backward vs the oracle evaluated in fp32 on the same (dtype-rounded) inputs.  `rows`: compare y / dx on
    a row subset only (the oracle's dense products at T = 32768 would take minutes); dA / dB always in full."""
    from sow_amd import ops
    x, dy, A, B, W = _layer_case(T, d_in, d_out, r, dtype, dense, seed, scale)
    g = lambda t: None if t is None else t.to(DEV)
    y, h = ops.sow_forward(g(x), g(A), g(B), g(W), None, None, scale)
    dx, dA, dB, _ = ops.sow_backward(g(dy), g(x), h, g(A), g(B), g(W), None, scale, False)
    f = lambda t: None if t is None else t.float()
    tol = TOL if dtype == torch.float32 else 2e-2
    wtol = 2e-5 if dtype == torch.float32 else 2e-2
    idx = slice(None) if rows is None else torch.arange(3, T, T // rows)
    y_ref = O.sow_forward(f(x)[idx], [f(A)], [f(B)], f(W), None, scale, None)
    dx_ref, _, _, _ = O.sow_backward(f(dy)[idx], f(x)[idx], [f(A)], [f(B)], f(W), None, scale, False)
    assert rel_err(y.float().cpu()[idx], y_ref) < tol
    assert rel_err(dx.float().cpu()[idx], dx_ref) < tol
    # the weight gradients do not involve the frozen accumulator: full-T oracle without it
    _, dA_ref, dB_ref, _ = O.sow_backward(f(dy), f(x), [f(A)], [f(B)], None, None, scale, False)
    assert rel_err(dA.float().cpu(), dA_ref[0]) < wtol
    assert rel_err(dB.float().cpu(), dB_ref[0]) < wtol


CFG5 = [(4096, 4096), (4096, 11008), (11008, 4096)]     # llama_7b.json: q/k/v, up, down (finetune.py:294-298)
CFG4 = [(768, 768), (768, 3072), (3072, 768)]           # roberta.json: query/key/value/output.dense, intermediate, output
CFG2 = [(512, 512), (512, 1376), (1376, 512)]           # llama_60m.json


@pytest.mark.parametrize("shape", CFG5)
def test_config5_llama7b_shapes_bf16_dense(shape):
    """T = 4 x 256 = 1024 tokens, r = 8, bf16, dense frozen weight (`keep`)."""
    _check_layer(1024, shape[0], shape[1], 8, torch.bfloat16, True, 50 + shape[1] % 7)


@pytest.mark.parametrize("shape", CFG4)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_config4_roberta_shapes_dense(shape, dtype):
    """T = 16 x 512 = 8192 tokens, r = 8, dense frozen weight (`keep`), fp32 (run_glue default) and bf16."""
    _check_layer(8192, shape[0], shape[1], 8, dtype, True, 40 + shape[1] % 5)


@pytest.mark.parametrize("shape", CFG2)
@pytest.mark.parametrize("dense", [False, True])
def test_config2_llama60m_full_size_bf16(shape, dense):
    """T = 128 x 256 = 32768 tokens, r = 50, bf16: y / dX on 96 rows, dA / dB in full, before (no accumulator) and after
    (dense accumulator) the first accumulate()."""
    _check_layer(32768, shape[0], shape[1], 50, torch.bfloat16, dense, 20 + shape[1] % 3, scale=1.0, rows=96)


# ---------------------------------------------------------------------------------------------
# strided inputs: the saved activations must be the rows the forward read
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["last_dim_slice", "cls_token", "transposed"])
def test_non_contiguous_input(kind):
    from sow_amd import SoWLinear
    gen = torch.Generator().manual_seed(31)
    d_in, d_out, r = 96, 80, 8
    if kind == "last_dim_slice":
        big = torch.randn(6, 40, d_in + 32, generator=gen)
        view = lambda t: t[..., :d_in]
    elif kind == "cls_token":
        big = torch.randn(24, 7, d_in, generator=gen)
        view = lambda t: t[:, 0, :]
    else:
        big = torch.randn(d_in, 50, generator=gen)
        view = lambda t: t.t()
    layer = SoWLinear(d_in, d_out, bias=True, rank=r, scale=0.5, init_method="normal", device=DEV)
    W = torch.randn(d_in, d_out, generator=gen) * 0.05
    layer.acc_downweight = nn.Parameter(W.to(DEV), requires_grad=False)
    big_d = big.to(DEV).requires_grad_(True)
    x = view(big_d)
    assert not x.is_contiguous()
    y = layer(x)
    dy = torch.randn(*y.shape, generator=gen)
    y.backward(dy.to(DEV))
    A, B = layer.downscale_weights[0].data.cpu(), layer.upscale_weights[0].data.cpu()
    xc = view(big).contiguous()
    y_ref = O.sow_forward(xc, [A], [B], W, None, 0.5, layer.bias.data.cpu())
    dx_ref, dA_ref, dB_ref, db_ref = O.sow_backward(dy, xc, [A], [B], W, None, 0.5, True)
    assert rel_err(y.detach().cpu(), y_ref) < TOL
    big_ref = torch.zeros_like(big)
    view(big_ref).copy_(dx_ref)
    assert rel_err(big_d.grad.cpu(), big_ref) < TOL
    assert rel_err(layer.downscale_weights[0].grad.cpu(), dA_ref[0]) < TOL
    assert rel_err(layer.upscale_weights[0].grad.cpu(), dB_ref[0]) < TOL
    assert rel_err(layer.bias.grad.cpu(), db_ref) < TOL


def test_alias_package_builds_a_layer_on_the_gpu():
    """The drivers' import lines (simple_train.py:35-38, finetune.py:32-33) resolve to the HIP implementation."""
    from tn_gradient.layer.sow import SoWLinear
    from tn_gradient.prepare import SoWConfig, accumulate, prepare_sow

    import sow_amd
    assert SoWLinear is sow_amd.SoWLinear and prepare_sow is sow_amd.prepare_sow
    net = nn.Sequential()
    net.add_module("q_proj", nn.Linear(64, 48, bias=False))
    net = prepare_sow(net, SoWConfig(target_modules=["q_proj"], rank=4, init_method="normal_QR", decompose=None, device=DEV))
    assert isinstance(net.q_proj, SoWLinear) and net.q_proj.downscale_weights[0].is_cuda
    x = torch.randn(5, 64, device=DEV)
    A, B = net.q_proj.downscale_weights[0].data, net.q_proj.upscale_weights[0].data
    assert rel_err(net(x).detach().cpu(), O.sow_forward(x.cpu(), [A.cpu()], [B.cpu()], None, None, 1.0, None)) < TOL
    accumulate(net)
    assert tuple(net.q_proj.acc_downweight.shape) == (64, 48)
    assert "SoWLinear" in str(net)


# ---------------------------------------------------------------------------------------------
# f1: fused factor AdamW on the flat bucket (simple_train.py:502-506 factor group)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pdtype,sdtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                           (torch.bfloat16, torch.bfloat16)])
def test_adamw_flat_llama60m_bucket(pdtype, sdtype):
    """n = 3 904 000 (every factor of llama_60m r = 50), grad_scale = 0.5 (the data-parallel averaging of a 2-rank sum).
    Reference: torch.optim.AdamW in fp32 on the same (rounded) parameters and half the gradient."""
    from sow_amd import ops
    n = 3_904_000
    gen = torch.Generator(device=DEV).manual_seed(3)
    p0 = (torch.randn(n, generator=gen, device=DEV) * 0.05).to(pdtype)
    p = p0.clone()
    ref = nn.Parameter(p0.float().clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-6)
    m, v = torch.zeros(n, device=DEV, dtype=sdtype), torch.zeros(n, device=DEV, dtype=sdtype)
    for step in range(1, 4):
        g = (torch.randn(n, generator=gen, device=DEV) * 1e-2).to(pdtype)
        ref.grad = 0.5 * g.float()
        opt.step()
        ops.adamw_flat_(p, g, m, v, lr=1e-2, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.1, step=step, grad_scale=0.5)
        if pdtype == torch.bfloat16:
            # a bf16 parameter rounds after every step; compare the UPDATE against the fp32 trajectory restarted from it
            assert rel_err(p.float().cpu(), ref.data.cpu()) < 1e-2
        else:
            assert rel_err(p.cpu(), ref.data.cpu()) < 1e-5
    st = opt.state[ref]
    stol = 1e-5 if sdtype == torch.float32 else 1e-2
    if pdtype == torch.float32:
        assert rel_err(m.float().cpu(), st["exp_avg"].cpu()) < stol and rel_err(v.float().cpu(), st["exp_avg_sq"].cpu()) < stol


def test_tt_newton_and_reciprocal_vs_reference():
    """a11: TensorTrain.sqrt / sqrtinv / reciprocal (tt.py:279-341, 480-494) against the reference's outputs
    (tests/tt_test.py:1-13 prints the first case).  The Newton iterations round through chains of small QRs on
    rank-deficient unfoldings, so reconstructions -- not cores -- are compared, at 1e-4."""
    from sow_amd import TensorTrain
    g = load_golden("tt_newton")
    a = torch.arange(2 * 2 * 2 * 3 * 3 * 3).reshape(2, 2, 2, 3, 3, 3).float().to(DEV)
    got = TensorTrain.from_tensor(a, [1, 4, 4, 1]).sqrt().reconstruct()
    assert rel_err(got.cpu(), g["t216_sqrt_rec"]) < 1e-4
    tp = TensorTrain.from_cores([g[f"pos_core{i}"].to(DEV) for i in range(3)])
    for name, fn in (("sqrt", lambda t: t.sqrt()), ("sqrt_it2", lambda t: t.sqrt(max_iter=2)),
                     ("sqrtinv", lambda t: t.sqrtinv()), ("sqrtinv_it2", lambda t: t.sqrtinv(threshold=None, max_iter=2))):
        out = fn(tp)
        assert list(out.ranks) == [int(r) for r in g[f"pos_{name}_ranks"]], name
        assert rel_err(out.reconstruct().cpu(), g[f"pos_{name}_rec"]) < 1e-4, name
    rec = TensorTrain.from_cores([g[f"recip_in{i}"].to(DEV) for i in range(3)]).reciprocal()
    for i in range(3):
        assert rel_err(rec.cores[i].cpu(), g[f"recip_out{i}"]) < 1e-4


# ---------------------------------------------------------------------------------------------
# grouped C-ABI calls (sow_forward_group / sow_backward_group)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["qkv", "gate_up", "mixed", "dense_qkv"])
def test_grouped_calls_are_bit_identical_to_single_calls(case):
    """{q, k, v} / {gate, up} of a decoder block in ONE grid per kernel: outputs, saved h, dX, dA, dB bit-identical to the
    per-layer calls (every workgroup runs the single-layer code on its own layer).  `mixed`: an eligible layer next to a
    dense-accumulator layer, a short-T layer and an fp32-only shape -- those are forwarded to the single-layer path."""
    from sow_amd import _lib, ops
    T = 16400   # ragged last token block; > 8192 so the streaming kernels are not split
    if case == "qkv":
        specs = [(T, 512, 512, 50, False)] * 3
    elif case == "gate_up":
        specs = [(T, 512, 1376, 50, False), (T, 512, 1376, 50, False)]
    elif case == "dense_qkv":   # dense accumulator: the one-launch-per-layer GEMM (gemm2h) shares its grid across the group
        T = 20600
        specs = [(T, 512, 512, 50, True)] * 3 + [(T, 1376, 512, 50, True)]
    else:
        specs = [(T, 512, 512, 50, False), (T, 512, 512, 50, True), (4096, 512, 1376, 50, False), (T, 264, 72, 34, False),
                 (T, 1376, 512, 50, False)]
    gen = torch.Generator(device=DEV).manual_seed(77)
    calls, singles = [], []
    for (t, di, do, r, dense) in specs:
        x = torch.randn(t, di, generator=gen, device=DEV).bfloat16()
        dy = torch.randn(t, do, generator=gen, device=DEV).bfloat16()
        A = (torch.randn(di, r, generator=gen, device=DEV) * 0.05).bfloat16()
        B = (torch.randn(r, do, generator=gen, device=DEV) * 0.05).bfloat16()
        W = (torch.randn(di, do, generator=gen, device=DEV) * 0.02).bfloat16() if dense else None
        dA, dB = torch.zeros_like(A), torch.zeros_like(B)
        dx = torch.empty_like(x)
        calls.append(ops.LayerCall(x, A, B, acc_down=W, scale=0.75, dy2=dy, dx=dx, out=(dA, dB, None), grad_beta=0.0))
        y1, h1 = ops.sow_forward(x, A, B, W, None, None, 0.75)
        dx1, dA1, dB1, _ = ops.sow_backward(dy, x, h1, A, B, W, None, 0.75, False)
        singles.append((y1, h1, dx1, dA1, dB1))
    grp = ops.LayerGroup(calls)
    # NO_TN_ROWS: a group with enough layers may take the row-owner weight-gradient kernel, whose slab sums are cut
    # differently (test_block_weight_gradients_row_owner_kernel); the column-owner kernels are bit-identical to single calls
    with _lib.switch(NO_TN_ROWS=1):
        grp.forward()
        grp.backward(_lib.BWD_DATA | _lib.BWD_WEIGHTS)
    torch.cuda.synchronize()
    for c, (y1, h1, dx1, dA1, dB1) in zip(calls, singles):
        r = c.args.r_live
        hv, h1v = c.h.view(-1, 64), h1.view(-1, 64)
        assert torch.equal(c.y, y1) and torch.equal(hv[:, :r], h1v[:, :r]) and torch.equal(hv[:, 63], h1v[:, 63])
        assert torch.equal(c.dx, dx1)
        assert torch.equal(c._keep[7], dA1) and torch.equal(c._keep[8], dB1)
    # default switches: same outputs and input gradients; weight gradients to rounding
    grp.forward()
    grp.backward(_lib.BWD_DATA | _lib.BWD_WEIGHTS)
    for c, (y1, h1, dx1, dA1, dB1) in zip(calls, singles):
        assert torch.equal(c.y, y1) and torch.equal(c.dx, dx1)
        assert rel_err(c._keep[7].float().cpu(), dA1.float().cpu()) < 1e-2 and rel_err(c._keep[8].float().cpu(), dB1.float().cpu()) < 1e-2
    # the same with the grouping switched off (layer-by-layer through the group entry point)
    with _lib.switch(NO_GROUPED=1):
        for c in calls:
            c.y.zero_()
        grp.forward()
        grp.backward(_lib.BWD_DATA | _lib.BWD_WEIGHTS)
    for c, (y1, _, dx1, dA1, _) in zip(calls, singles):
        assert torch.equal(c.y, y1) and torch.equal(c.dx, dx1) and torch.equal(c._keep[7], dA1)


@pytest.mark.parametrize("T", [8192, 8200, 32768])
def test_block_weight_gradients_row_owner_kernel(T):
    """The 7 projections of a llama_60m decoder block in ONE weight-gradient launch: the row-owner kernel (a workgroup owns
    all columns of a token slab; slab counts planned over the group) against a float64 reference of dA = x^T dh,
    dB = h^T dY, dbias = colsum(dY) built from the kernel's own bf16 h / dh, and against the per-layer calls; the deferred
    form (PARTIAL with BWD_GROUP_SLABS + sow_backward_group_reduce_desc + sow_reduce_batch) is bit-identical to the
    one-call form.  A 768-wide layer (12 column groups: two per wave) and biases ride along."""
    from sow_amd import _lib, ops
    specs = [(512, 512, False)] * 3 + [(512, 512, True), (512, 1376, False), (512, 1376, True), (1376, 512, False)]
    if T == 8200:
        specs = specs[:5] + [(768, 768, True), (1376, 512, False)]
    r = 50
    gen = torch.Generator(device=DEV).manual_seed(5)
    calls, ref = [], []
    for (di, do, has_bias) in specs:
        x = torch.randn(T, di, generator=gen, device=DEV).bfloat16()
        dy = torch.randn(T, do, generator=gen, device=DEV).bfloat16()
        A = (torch.randn(di, r, generator=gen, device=DEV) * 0.05).bfloat16()
        B = (torch.randn(r, do, generator=gen, device=DEV) * 0.05).bfloat16()
        bias = torch.zeros(do, device=DEV, dtype=torch.bfloat16) if has_bias else None
        out = (torch.zeros_like(A), torch.zeros_like(B), torch.zeros_like(bias) if has_bias else None)
        calls.append(ops.LayerCall(x, A, B, bias=bias, scale=0.5, dy2=dy, dx=torch.empty_like(x), out=out, grad_beta=0.0))
        y1, h1 = ops.sow_forward(x, A, B, None, None, bias, 0.5)
        _, dA1, dB1, db1 = ops.sow_backward(dy, x, h1, A, B, None, None, 0.5, has_bias)
        ref.append((x, dy, A, B, dA1, dB1, db1))
    grp = ops.LayerGroup(calls)
    pinned = dict(NO_TN_ROWS=0, NO_GROUPED=0, TN_NARROW=0)   # the kernel under test, whatever the environment forces elsewhere
    with _lib.switch(**pinned):
        assert grp.weight_gradient_plan()[0]
        grp.forward()
        grp.backward(_lib.BWD_DATA | _lib.BWD_WEIGHTS)          # PARTIAL and REDUCE in one call: group-planned slabs
    torch.cuda.synchronize()
    one_call = [(c._keep[7].clone(), c._keep[8].clone(), None if c._keep[9] is None else c._keep[9].clone()) for c in calls]
    worst = 0.0
    for c, (x, dy, A, B, dA1, dB1, db1), (dA, dB, db) in zip(calls, ref, one_call):
        h = c.h.view(-1, 64)[:, :r].double()                                   # scaled bf16 h as saved by the forward kernel
        dh = (0.5 * (dy.double() @ B.double().t())).bfloat16().double()        # what the data-gradient kernel leaves for dA
        dA64, dB64 = x.double().t() @ dh, h.t() @ dy.double()
        for got, single, want in ((dA, dA1, dA64), (dB, dB1, dB64)):
            e_rows, e_single = rel_err(got.double().cpu(), want.cpu()), rel_err(single.double().cpu(), want.cpu())
            assert e_rows < 6e-3 and e_rows < 1.5 * e_single + 1e-3, (tuple(got.shape), e_rows, e_single)
            worst = max(worst, e_rows)
        if db is not None:
            assert rel_err(db.double().cpu(), dy.double().sum(0).cpu()) < 6e-3
            assert rel_err(db.double().cpu(), db1.double().cpu()) < 1e-2
    # deferred reduction built from the group's own descriptors
    for c in calls:
        for g_ in c._keep[7:10]:
            if g_ is not None:
                g_.zero_()
    ph = _lib.BWD_WEIGHTS_PARTIAL | _lib.BWD_GROUP_SLABS
    with _lib.switch(**pinned):
        grp.backward(ph)
        red = ops.DeferredReduce()
        red.add_group(grp, ph)
        red.run()
    torch.cuda.synchronize()
    for c, (dA, dB, db) in zip(calls, one_call):
        assert torch.equal(c._keep[7], dA) and torch.equal(c._keep[8], dB)
        assert db is None or torch.equal(c._keep[9], db)
    # and with the kernel switched off the group is bit-identical to the per-layer calls again
    with _lib.switch(NO_TN_ROWS=1):
        grp.backward(_lib.BWD_WEIGHTS)
    for c, (_, _, _, _, dA1, dB1, _) in zip(calls, ref):
        assert torch.equal(c._keep[7], dA1) and torch.equal(c._keep[8], dB1)



@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_row_owner_kernel_random_groups(seed):
    """Random groups that qualify for the row-owner kernel (checked through sow_backward_group_plan): widths that are not a
    multiple of 64, more than 16 column groups (two column ranges), ranks 4 .. 64, ragged token counts, layers with bias --
    dA, dB, dbias against float64 products of the operands the kernel reads."""
    from sow_amd import _lib, ops
    rng = np.random.default_rng(100 + seed)
    widths = [72, 264, 512, 768, 1000, 1376, 2048, 2752]
    for _attempt in range(20):
        T = int(rng.choice([8200, 12000, 16392]))
        n = int(rng.integers(5, 9))
        specs = [(int(rng.choice(widths)), int(rng.choice(widths)), int(rng.choice([4, 8, 34, 50, 64])), bool(rng.integers(0, 2)))
                 for _ in range(n)]
        specs = [(di, do, r, hb and r < 64) for (di, do, r, hb) in specs]
        gen = torch.Generator(device=DEV).manual_seed(seed)
        calls = []
        for (di, do, r, hb) in specs:
            x = torch.randn(T, di, generator=gen, device=DEV).bfloat16()
            dy = torch.randn(T, do, generator=gen, device=DEV).bfloat16()
            A = (torch.randn(di, r, generator=gen, device=DEV) * 0.05).bfloat16()
            B = (torch.randn(r, do, generator=gen, device=DEV) * 0.05).bfloat16()
            bias = torch.zeros(do, device=DEV, dtype=torch.bfloat16) if hb else None
            out = (torch.zeros_like(A), torch.zeros_like(B), torch.zeros_like(bias) if hb else None)
            calls.append(ops.LayerCall(x, A, B, bias=bias, scale=1.5, dy2=dy, dx=torch.empty_like(x), out=out, grad_beta=0.0))
        grp = ops.LayerGroup(calls)
        with _lib.switch(NO_TN_ROWS=0, NO_GROUPED=0, TN_NARROW=0):      # whatever the environment forces elsewhere
            rows, slabs = grp.weight_gradient_plan()
        if rows:
            break
        del calls, grp
        torch.cuda.empty_cache()
    assert rows, "no qualifying group drawn"
    with _lib.switch(NO_TN_ROWS=0, NO_GROUPED=0, TN_NARROW=0):
        grp.forward()
        grp.backward(_lib.BWD_DATA | _lib.BWD_WEIGHTS)
    torch.cuda.synchronize()
    for c, (di, do, r, hb) in zip(calls, specs):
        x, A, B, dy = c._keep[0], c._keep[1], c._keep[2], c._keep[6]
        h = c.h.view(-1, 64)[:, :r].double()
        dh = (1.5 * (dy.double() @ B.double().t())).bfloat16().double()
        assert rel_err(c._keep[7].double().cpu(), (x.double().t() @ dh).cpu()) < 8e-3, (T, di, do, r)
        assert rel_err(c._keep[8].double().cpu(), (h.t() @ dy.double()).cpu()) < 8e-3, (T, di, do, r)
        if hb:
            assert rel_err(c._keep[9].double().cpu(), dy.double().sum(0).cpu()) < 8e-3


# ---------------------------------------------------------------------------------------------
# batched periodic step (sow_accumulate_batch)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("init", ["normal_QR", "normal"])
def test_accumulate_model_batched_equals_per_layer(dtype, init):
    """accumulate(model) -- one C call, one launch per phase for all layers -- against SoWLinear.accumulate() layer by
    layer on an identical copy with the same re-initialisation draws: accumulator, new A (orthonormal, LAPACK signs),
    zeroed B; first call (no accumulator yet) and second call (in-place update); a layer the batch does not cover
    (n_iter = 2) takes the per-layer path inside the same call."""
    import copy

    from sow_amd import SoWLinear, accumulate
    torch.manual_seed(3)
    shapes = [(512, 512), (512, 1376), (1376, 512), (96, 40), (200, 264)]
    net = nn.ModuleList([SoWLinear(i, o, bias=False, rank=r, scale=0.5, init_method="normal", device=DEV, dtype=dtype)
                         for (i, o), r in zip(shapes, (50, 50, 50, 8, 34))])
    net.append(SoWLinear(64, 48, bias=False, rank=4, n_iter=2, scale=0.5, init_method="normal", device=DEV, dtype=dtype))
    for m in net:
        m.init_method = init
        m.virtual_rank = min(m.in_features, m.out_features)          # what prepare_sow sets (prepare.py:120)
    ref = copy.deepcopy(net)
    gen = torch.Generator().manual_seed(5)
    for call in range(2):
        for m, mr in zip(net, ref):
            for i in range(m.n_iter):
                b = (torch.randn(m.rank, m.out_features, generator=gen) * 0.1).to(DEV, dtype)
                m.upscale_weights[i].data.copy_(b)
                mr.upscale_weights[i].data.copy_(b)
            shape = (m.in_features, m.out_features) if init == "normal_QR" else (m.in_features, m.rank)
            ds = [torch.randn(*shape, generator=gen) * 0.02 for _ in range(m.n_iter)]
            for mod in (m, mr):
                mod._fresh_gaussian = lambda shape, device, dtype_, _it=iter(ds): next(_it).to(device, dtype_)
        pA = net[0].downscale_weights[0].data.data_ptr()
        accumulate(net)                       # batched (layers 0-4) + per-layer (layer 5)
        for mr in ref:
            mr.accumulate()
        torch.cuda.synchronize()
        assert net[0].downscale_weights[0].data.data_ptr() == pA       # factors rewritten in place
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        qtol = 5e-5 if dtype == torch.float32 else 2e-2
        for m, mr in zip(net, ref):
            assert tuple(m.acc_downweight.shape) == (m.in_features, m.out_features) and m.acc_upweight.numel() == 0
            assert m.virtual_rank == mr.virtual_rank
            assert rel_err(m.acc_downweight.float().cpu(), mr.acc_downweight.float().cpu()) < tol, (call, m.in_features)
            for i in range(m.n_iter):
                assert rel_err(m.downscale_weights[i].data.float().cpu(), mr.downscale_weights[i].data.float().cpu()) < qtol
                assert float(m.upscale_weights[i].data.abs().max()) == 0.0
        if init == "normal_QR" and dtype == torch.float32:
            q = net[1].downscale_weights[0].data.double()
            assert float((q.t() @ q - torch.eye(50, device=DEV, dtype=torch.float64)).abs().max()) < 1e-5


def test_accumulate_llama60m_unhooked_draws_are_orthonormal_and_fast():
    """The production path (no draw hook): one normal_() for all layers, only the first `rank` columns drawn."""
    import time

    from sow_amd import SoWLinear, accumulate
    torch.manual_seed(0)
    shapes = ([(512, 512)] * 4 + [(512, 1376)] * 2 + [(1376, 512)]) * 8
    net = nn.ModuleList([SoWLinear(i, o, bias=False, rank=50, init_method="normal", device=DEV, dtype=torch.bfloat16) for i, o in shapes])
    for m in net:
        m.init_method = "normal_QR"
        m.virtual_rank = min(m.in_features, m.out_features)
    accumulate(net)
    accumulate(net)
    ms = float("inf")
    for _ in range(3):                 # best of three: a wall-clock bound on a shared box
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        accumulate(net)
        torch.cuda.synchronize()
        ms = min(ms, (time.perf_counter() - t0) * 1e3)
    for m in (net[0], net[4], net[6], net[55]):
        q = m.downscale_weights[0].data.double()
        assert float((q.t() @ q - torch.eye(50, device=DEV, dtype=torch.float64)).abs().max()) < 2e-2
        assert float(m.upscale_weights[0].data.abs().max()) == 0.0
    different = float((net[0].downscale_weights[0].data.float() - net[1].downscale_weights[0].data.float()).abs().max())
    assert different > 1e-3          # independent draws per layer
    print(f"accumulate(llama_60m, 56 layers): {ms:.2f} ms")
    assert ms < 10.0


def test_pretrain_protocol_gradient_accumulation_3():
    """a14 with gradient accumulation: simple_train.py:596-650, GA = 3, on the HIP path through prepare_sow / accumulate(model)
    (batched) / reset_optimizer -- the predicate fires on micro-steps 10 and 11 exactly as in the reference's run."""
    import sow_amd
    be = types.SimpleNamespace(SoWLinear=sow_amd.SoWLinear, SoWConfig=sow_amd.SoWConfig, prepare_sow=sow_amd.prepare_sow,
                               reset_optimizer=sow_amd.reset_optimizer, accumulate=sow_amd.accumulate)
    g = load_golden("train_trace_ga3")
    losses, fired = P.replay_pretrain_ga(be, g, DEV, _set_draw, dict(loss=2e-4, acc=1e-3, final=1e-3))
    assert fired == [10, 11]


@pytest.mark.parametrize("shape", [(8200, 768, 768, 50), (4100, 512, 1376, 50), (2050, 264, 72, 34)])
def test_fp32_3xbf16_agrees_with_exact_fp32_mfma(shape):
    """fp32 tensors: the default 3 x bf16 form (products on the bf16 matrix pipe, dropped cross terms <= 2^-23 per product)
    against the exact v_mfma_f32_32x32x2_f32 form of the same kernels (F32_EXACT switch) and against the oracle."""
    from sow_amd import _lib, ops
    T, di, do, r = shape
    gen = torch.Generator().manual_seed(17)
    x, dy = torch.randn(T, di, generator=gen), torch.randn(T, do, generator=gen)
    A = torch.linalg.qr(torch.randn(di, r, generator=gen))[0].contiguous()
    B = torch.randn(r, do, generator=gen) * 0.05
    g = lambda t: t.to(DEV)
    y, h = ops.sow_forward(g(x), g(A), g(B), None, None, None, 0.5)
    dx, dA, dB, _ = ops.sow_backward(g(dy), g(x), h, g(A), g(B), None, None, 0.5, False)
    with _lib.switch(F32_EXACT=1):
        y0, h0 = ops.sow_forward(g(x), g(A), g(B), None, None, None, 0.5)
        dx0, dA0, dB0, _ = ops.sow_backward(g(dy), g(x), h0, g(A), g(B), None, None, 0.5, False)
    for a, b in ((y, y0), (dx, dx0), (dA, dA0), (dB, dB0)):
        assert rel_err(a.cpu(), b.cpu()) < 2e-6
    y_ref = O.sow_forward(x, [A], [B], None, None, 0.5, None)
    dx_ref, dA_ref, dB_ref, _ = O.sow_backward(dy, x, [A], [B], None, None, 0.5, False)
    assert rel_err(y.cpu(), y_ref) < TOL and rel_err(dx.cpu(), dx_ref) < TOL
    assert rel_err(dA.cpu(), dA_ref[0]) < TOL and rel_err(dB.cpu(), dB_ref[0]) < TOL


# ---------------------------------------------------------------------------------------------
# sibling grouping at the module level (sow_amd.group_siblings)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_group_siblings_in_a_llama_block(dtype):
    """q/k/v and gate/up of a tiny Llama run through ONE autograd node per group after group_siblings(model): the loss is
    bit-identical to the ungrouped model (same kernels per layer), every parameter gradient agrees to rounding (the sum of
    the siblings' input gradients is formed once instead of by autograd's accumulation), also under gradient checkpointing;
    a `keep` (dense accumulator) model groups too."""
    transformers = pytest.importorskip("transformers")
    import copy

    from sow_amd import SoWConfig, group_siblings, prepare_sow, ungroup_siblings
    torch.manual_seed(7)
    cfg = transformers.LlamaConfig(hidden_size=128, intermediate_size=344, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=256, max_position_embeddings=128, rms_norm_eps=1e-6,
                                   tie_word_embeddings=False, attn_implementation="eager")
    for decompose in (None, "keep"):
        base = transformers.AutoModelForCausalLM.from_config(cfg)
        base = prepare_sow(base, SoWConfig(target_modules=["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"],
                                           rank=8, init_method="normal", scale=0.5, decompose=decompose, device="cpu"))
        base = base.to(DEV, dtype)
        twin = copy.deepcopy(base)
        assert group_siblings(twin) == 4                       # {q,k,v} and {gate,up} in each of the two blocks
        tokens = torch.randint(0, 256, (8, 96), generator=torch.Generator().manual_seed(1)).to(DEV)   # 768 tokens
        for ckpt in (False, True):
            for net in (base, twin):
                net.zero_grad(set_to_none=True)
                if ckpt:
                    net.gradient_checkpointing_enable()
                    net.train()
                loss = net(input_ids=tokens, labels=tokens.clone()).loss
                loss.backward()
                net._loss = float(loss.detach())
            assert base._loss == twin._loss, (decompose, ckpt)
            for (n1, p1), (_, p2) in zip(base.named_parameters(), twin.named_parameters()):
                if p1.grad is None:
                    assert p2.grad is None, n1
                else:
                    assert rel_err(p2.grad.float().cpu(), p1.grad.float().cpu()) < (2e-2 if dtype == torch.bfloat16 else 1e-5), \
                        (n1, decompose, ckpt)
        ungroup_siblings(twin)
        assert not any(hasattr(m, "_sibling_group") for m in twin.modules())


def test_group_siblings_falls_back_when_inputs_differ():
    from sow_amd import SoWLinear, group_siblings

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = SoWLinear(64, 48, bias=True, rank=8, init_method="normal", device=DEV)
            self.k_proj = SoWLinear(64, 32, bias=False, rank=8, init_method="normal", device=DEV)
            self.v_proj = SoWLinear(64, 32, bias=False, rank=4, init_method="normal", device=DEV)

    blk = Block()
    assert group_siblings(blk) == 1
    x1, x2 = torch.randn(9, 64, device=DEV), torch.randn(9, 64, device=DEV)
    q = blk.q_proj(x1)               # computes the group on x1, parks k and v
    k_other = blk.k_proj(x2)         # different input: runs on its own
    v = blk.v_proj(x1)               # picks up its parked output
    A, B = blk.k_proj.downscale_weights[0].data.cpu(), blk.k_proj.upscale_weights[0].data.cpu()
    assert rel_err(k_other.detach().cpu(), O.sow_forward(x2.cpu(), [A], [B], None, None, 1.0, None)) < TOL
    A, B = blk.v_proj.downscale_weights[0].data.cpu(), blk.v_proj.upscale_weights[0].data.cpu()
    assert rel_err(v.detach().cpu(), O.sow_forward(x1.cpu(), [A], [B], None, None, 1.0, None)) < TOL
    A, B = blk.q_proj.downscale_weights[0].data.cpu(), blk.q_proj.upscale_weights[0].data.cpu()
    assert rel_err(q.detach().cpu(), O.sow_forward(x1.cpu(), [A], [B], None, None, 1.0, blk.q_proj.bias.data.cpu())) < TOL
    (q.sum() + v.sum() + k_other.sum()).backward()
    assert blk.k_proj.downscale_weights[0].grad is not None and blk.q_proj.bias.grad is not None
