"""Caller protocols of the reference's fine-tuning drivers, written once against a `backend` namespace
(SoWLinear, SoWConfig, prepare_sow, reset_optimizer) so that tests/golden/make_golden.py can run them on the REFERENCE
(CPU, in the build container) and the -m gpu tests on sow_amd with the same code.

  glue_protocol      scripts/run_glue.py:976-1002   optimizer.step() first, then accumulate; `scale = 1/rank` after the
                                                    FIRST accumulation only; reset_optimizer(group_id=2)
  finetune_protocol  scripts/finetune.py:39-77      accumulate between backward and optimizer.step(); `scale = 1/rank`
                                                    after EVERY accumulation; reset_optimizer(group_id=2)

The model is a RoBERTa-shaped trio (BASELINE config 4: 768->768, 768->3072, 3072->768, r = 8, fp32, decompose='keep')
whose module names exercise the multi-component suffix match of prepare_sow (`output.dense`, `intermediate.dense`,
run_glue.py:572).  Every random tensor is drawn on the CPU from seeded generators, so both sides see identical inputs;
the factor initialisations / re-initialisations (device RNG in the product) are passed in explicitly.
"""
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

TRIO_TARGETS = ["query", "output.dense", "intermediate.dense"]
TRIO_RANK = 8
TRIO_DIMS = (768, 3072)
TRIO_TOKENS = (2, 128)          # 256 tokens
TRIO_STEPS = 5
TRIO_ACC_EVERY = 2


class _Holder(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.dense = nn.Linear(d_in, d_out)


class Trio(nn.Module):
    """query (d->d), intermediate.dense (d->4d), output.dense (4d->d) with a residual, like one encoder block's linears."""

    def __init__(self, d=TRIO_DIMS[0], d_ff=TRIO_DIMS[1]):
        super().__init__()
        self.query = nn.Linear(d, d)
        self.intermediate = _Holder(d, d_ff)
        self.output = _Holder(d_ff, d)

    def forward(self, x):
        h = x + self.query(x)
        return h + self.output.dense(F.gelu(self.intermediate.dense(h)))


class TrioModel(nn.Module):
    """The trio one level down (`encoder.query`, `encoder.intermediate.dense`, ...): prepare_sow's suffix match never tests
    the FULL dotted name (prepare.py:79 stops one component short), so a top-level `intermediate.dense` would be skipped."""

    def __init__(self, d, d_ff):
        super().__init__()
        self.encoder = Trio(d, d_ff)

    def forward(self, x):
        return self.encoder(x)


def build_trio(seed=2024, d=TRIO_DIMS[0], d_ff=TRIO_DIMS[1]):
    """Dense model on the CPU with seeded nn.Linear default init (identical on both sides)."""
    torch.manual_seed(seed)
    return TrioModel(d, d_ff)


TRIO_NAMES = ["encoder.query", "encoder.intermediate.dense", "encoder.output.dense"]


def trio_batches(steps=TRIO_STEPS, d=TRIO_DIMS[0], seed=77, tokens=TRIO_TOKENS):
    g = torch.Generator().manual_seed(seed)
    xs = torch.randn(steps, *tokens, d, generator=g)
    ts = torch.randn(steps, *tokens, d, generator=g)
    return xs, ts


def tensor_digest(t):
    """Size-independent fingerprint used to check that seeded inputs were regenerated identically (numpy's pairwise sums:
    independent of torch's thread count)."""
    a = t.detach().double().flatten().numpy()
    idx = np.arange(0, a.size, max(a.size // 16, 1))
    return torch.from_numpy(np.concatenate([[a.sum()], [np.abs(a).sum()], a[idx]]))


def sow_layers(model, backend):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, backend.SoWLinear)]


def param_groups(model, backend, lr=1e-3, sow_lr=5e-3):
    """Three AdamW groups as run_glue.py:756-808 / finetune.py:389-422 build them: decay, no-decay, factors (group 2)."""
    factors, ids = [], set()
    for _, m in sow_layers(model, backend):
        for w in list(m.downscale_weights) + list(m.upscale_weights):
            factors.append(w)
            ids.add(id(w))
    rest = [(n, p) for n, p in model.named_parameters() if p.requires_grad and id(p) not in ids]
    decay = [p for n, p in rest if not n.endswith("bias")]
    no_decay = [p for n, p in rest if n.endswith("bias")]
    # eps: with the default 1e-8 the first AdamW step after reset_optimizer is sign(g) -- entries with |g| ~ eps amplify a
    # 1e-7 relative rounding difference into a 1e-3 difference of the factors, which would force loose tolerances on
    # everything downstream.  eps = 1e-4 (>> |g| ~ 1e-5 here) keeps the update linear in g, so the trace stays a sharp test
    # of the SoW arithmetic; the optimizer itself is torch.optim.AdamW on both sides and not under test.
    groups = [{"params": decay, "lr": lr, "weight_decay": 0.01, "eps": 1e-4},
              {"params": no_decay, "lr": lr, "weight_decay": 0.0, "eps": 1e-4},
              {"params": factors, "lr": sow_lr, "weight_decay": 0.0, "eps": 1e-4}]
    return groups


def _accumulate_all(model, backend, rank, set_scale, reinit):
    for i, (_, m) in enumerate(sow_layers(model, backend)):
        if reinit is not None:
            reinit(i, m)
        m.accumulate()
        if set_scale:
            m.scale = 1 / rank


def glue_protocol(model, opt, backend, xs, ts, rank, acc_every, reinit=None, on_accumulate=None, on_step=None):
    """run_glue.py:976-1002 with gradient_accumulation_steps = 1."""
    losses, completed = [], 0
    for step in range(xs.shape[0]):
        loss = F.mse_loss(model(xs[step]), ts[step])
        losses.append(float(loss.detach()))
        loss.backward()
        if on_step is not None:
            on_step(step, model)
        opt.step()
        opt.zero_grad()
        completed += 1
        if completed > 0 and completed % acc_every == 0:
            _accumulate_all(model, backend, rank, set_scale=(completed // acc_every == 1),
                            reinit=(lambda i, m, k=completed // acc_every - 1: reinit(k, i, m)) if reinit else None)
            backend.reset_optimizer(opt, group_id=2)
            if on_accumulate is not None:
                on_accumulate(completed // acc_every - 1, model)
    return losses


def finetune_protocol(model, opt, backend, xs, ts, rank, acc_every, ga=1, reinit=None, on_accumulate=None, on_step=None):
    """finetune.py:39-77 (SoWTrainer.training_step) inside the HF Trainer loop: `global_step` counts optimizer steps
    (one per `ga` micro-steps), `freq_step` counts micro-steps since the last accumulation.  The predicate is kept
    verbatim -- it needs BOTH counters divisible by accumulation_steps, which for ga = 1 never happens (quirk: the
    reference's fine-tune only accumulates when ga and accumulation_steps line up, and can then fire on consecutive
    micro-steps)."""
    st = types.SimpleNamespace(global_step=0, freq_step=None, n_acc=0)
    losses = []
    for k in range(xs.shape[0]):
        loss = F.mse_loss(model(xs[k]), ts[k])
        losses.append(float(loss.detach()))
        (loss / ga).backward()
        if on_step is not None:
            on_step(k, model)
        st.freq_step = st.freq_step + 1 if st.freq_step else 1
        if st.global_step > 0 and st.global_step % acc_every == 0 and st.freq_step % acc_every == 0 and st.freq_step > 0:
            _accumulate_all(model, backend, rank, set_scale=True,
                            reinit=(lambda i, m, n=st.n_acc: reinit(n, i, m)) if reinit else None)
            backend.reset_optimizer(opt, group_id=2)
            st.freq_step = 1
            if on_accumulate is not None:
                on_accumulate(st.n_acc, model)
            st.n_acc += 1
        if (k + 1) % ga == 0:
            opt.step()
            opt.zero_grad()
            st.global_step += 1
    return losses


def matrix_probe(w):
    """Compact but complete-coverage summary of a large matrix: every 64th row, plus all row and column sums (float64)."""
    w = w.detach().double().cpu()
    return w[::64].float().clone(), w.sum(dim=1), w.sum(dim=0)


def replay_and_check(be, g, proto, device, set_draw, tol):
    """Run `proto` ('glue' | 'finetune') on backend `be` with the inputs of fixture `g` (tests/golden/protocol_*.npz,
    generated from the reference) and compare losses, factor / bias gradients, accumulator probes, scales and the final
    factors.  `set_draw(module, tensor)` hands a recorded re-initialisation draw to the backend's layer.  `tol` = dict
    of relative tolerances (loss, grad, acc, final)."""
    steps = len(g["losses"])
    kw = {"ga": 2} if proto == "finetune" else {}
    model = build_trio()
    for n, p in model.named_parameters():
        assert torch.equal(tensor_digest(p), g[f"digest::{n}"]), f"seeded weight {n} was not regenerated identically"
    model = be.prepare_sow(model, be.SoWConfig(target_modules=TRIO_TARGETS, rank=TRIO_RANK, scale=1.0, init_method="normal",
                                               decompose="keep", device=device))
    layers = sow_layers(model, be)
    assert [n for n, _ in layers] == TRIO_NAMES == g["replaced"]          # index work: bit-exact
    model.to(device)
    for li, (_, m) in enumerate(layers):
        m.downscale_weights[0].data = g[f"init::{li}::A"].to(device)
        m.upscale_weights[0].data = g[f"init::{li}::B"].to(device)
        assert tuple(m.acc_downweight.shape) == (m.in_features, m.out_features) and m.virtual_rank == min(m.in_features, m.out_features)
    xs, ts = trio_batches(steps)
    assert torch.equal(tensor_digest(xs), g["digest::xs"]) and torch.equal(tensor_digest(ts), g["digest::ts"])
    xs, ts = xs.to(device), ts.to(device)
    opt = torch.optim.AdamW(param_groups(model, be))
    errs, scales = {}, []

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.double()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    def on_step(k, mdl):
        if k in (0, steps - 1):
            for li, (_, m) in enumerate(layers):
                for key, p in (("dA", m.downscale_weights[0]), ("dB", m.upscale_weights[0]), ("dbias", m.bias)):
                    errs[f"grad::{k}::{li}::{key}"] = (rel(p.grad, g[f"grad::{k}::{li}::{key}"]), tol["grad"])

    def reinit(n, li, m):
        set_draw(m, g[f"draw::{n}::{li}"])

    def on_acc(n, mdl):
        for li, (_, m) in enumerate(layers):
            sample, rows, cols = matrix_probe(m.acc_downweight.data)
            errs[f"acc::{n}::{li}::sample"] = (rel(sample, g[f"acc::{n}::{li}::sample"]), tol["acc"])
            errs[f"acc::{n}::{li}::rowsum"] = (rel(rows, g[f"acc::{n}::{li}::rowsum"]), tol["acc"])
            errs[f"acc::{n}::{li}::colsum"] = (rel(cols, g[f"acc::{n}::{li}::colsum"]), tol["acc"])
            assert m.acc_upweight.numel() == 0 and float(m.upscale_weights[0].detach().abs().max()) == 0.0
        scales.append([float(m.scale) for _, m in layers])

    run = glue_protocol if proto == "glue" else finetune_protocol
    losses = run(model, opt, be, xs, ts, TRIO_RANK, TRIO_ACC_EVERY, reinit=reinit, on_accumulate=on_acc, on_step=on_step, **kw)
    assert len(scales) == int(g["n_acc"]) and scales == g["scales"].tolist()           # protocol counters: exact
    for got, want in zip(losses, g["losses"].tolist()):
        assert abs(got - want) <= tol["loss"] * abs(want), (losses, g["losses"].tolist())
    with torch.no_grad():
        y = model(xs[-1])
    errs["final_y_rows"] = (rel(y.reshape(-1, y.shape[-1])[::8], g["final_y_rows"]), tol["final"])
    for li, (_, m) in enumerate(layers):
        errs[f"final::{li}::A"] = (rel(m.downscale_weights[0].data, g[f"final::{li}::A"]), tol["final"])
        errs[f"final::{li}::B"] = (rel(m.upscale_weights[0].data, g[f"final::{li}::B"]), tol["final"])
        errs[f"final::{li}::bias"] = (rel(m.bias.data, g[f"final::{li}::bias"]), tol["final"])
    bad = {k: v for k, v in errs.items() if not v[0] <= v[1]}
    assert not bad, bad
    return errs


# ---------------------------------------------------------------------------------------------
# scripts/simple_train.py:596-650 with gradient accumulation > 1 (the predicate of :618-626)
# ---------------------------------------------------------------------------------------------
LLAMA_TINY = dict(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                  vocab_size=256, max_position_embeddings=64, rms_norm_eps=1e-6, tie_word_embeddings=False,
                  attn_implementation="eager")
LLAMA_TARGETS = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]


def pretrain_protocol(model, opt, backend, tokens, ga, sow_accumulation, offset=0, reinit=None, on_accumulate=None):
    """simple_train.py:596-650, line for line where it touches the SoW path: loss / ga, backward, the accumulate predicate
    `(global_step % ga or ga == 1) and update_step > offset and (update_step - offset) % (ga * sow_accumulation) == 0`
    (quirk kept: with ga > 2 it is true on SEVERAL micro-steps of the same update, and never on the update's last one),
    accumulate(model) + reset_optimizer(group 1) between backward and the optimizer step, `continue` on non-boundary
    micro-steps.  Returns (losses per micro-step, list of global_steps at which accumulate fired)."""
    global_step = update_step = 0
    losses, fired = [], []
    accumulation_step = int(ga * sow_accumulation)
    for k in range(tokens.shape[0]):
        global_step += 1
        batch = tokens[k]
        loss = model(input_ids=batch, labels=batch.clone()).loss
        losses.append(float(loss.detach()))
        (loss / ga).backward()
        if (global_step % ga or ga == 1) and update_step > offset and (update_step - offset) % accumulation_step == 0:
            if reinit is not None:
                for i, (_, m) in enumerate(sow_layers(model, backend)):
                    reinit(len(fired), i, m)
            backend.accumulate(model)
            backend.reset_optimizer(opt, group_id=1)
            fired.append(global_step)
            if on_accumulate is not None:
                on_accumulate(len(fired) - 1, model)
        if global_step % ga != 0:
            continue
        opt.step()
        opt.zero_grad()
        update_step += 1
    return losses, fired


def llama_param_groups(model, backend):
    """Two AdamW groups as simple_train.py:389-405, 502-506: everything else, then the factors (group 1)."""
    special, ids = [], set()
    for _, m in sow_layers(model, backend):
        for w in list(m.downscale_weights) + list(m.upscale_weights):
            special.append(w)
            ids.add(id(w))
    others = [p for p in model.parameters() if p.requires_grad and id(p) not in ids]
    return [{"params": others, "lr": 1e-3, "weight_decay": 0.0, "eps": 1e-4},
            {"params": special, "lr": 5e-3, "weight_decay": 0.0, "eps": 1e-4}]


def replay_pretrain_ga(be, g, device, set_draw, tol):
    """Run pretrain_protocol on backend `be` with the inputs of tests/golden/train_trace_ga3.npz and compare the loss
    trace, the micro-steps at which accumulate fired (exact) and the final accumulator / factors of one layer."""
    import transformers
    torch.manual_seed(42)
    model = transformers.AutoModelForCausalLM.from_config(transformers.LlamaConfig(**LLAMA_TINY))
    model = be.prepare_sow(model, be.SoWConfig(target_modules=LLAMA_TARGETS, rank=int(g["rank"]), init_method="normal", scale=1.0,
                                               decompose=None, device="cpu"))
    sd = {k[len("init::"):]: v for k, v in g.items() if k.startswith("init::")}
    missing = model.load_state_dict(sd, strict=False)
    assert all(k.endswith("acc_upweight") or k.endswith("acc_downweight") for k in missing.missing_keys)
    model.to(device)
    layers = sow_layers(model, be)
    for _, m in layers:
        m.init_method = "normal_QR"
    opt = torch.optim.AdamW(llama_param_groups(model, be))
    tokens = g["tokens"].to(device)

    def reinit(n, li, m):
        set_draw(m, g[f"draw::{n}::{li}"])

    losses, fired = pretrain_protocol(model, opt, be, tokens, int(g["ga"]), int(g["sow_accumulation"]), reinit=reinit)
    assert fired == [int(v) for v in g["fired"]], (fired, g["fired"])
    for got, want in zip(losses, g["losses"].tolist()):
        assert abs(got - want) <= tol["loss"] * abs(want), (losses, g["losses"].tolist())
    probe = dict(model.named_modules())["model.layers.1.mlp.down_proj"]

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.double()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    assert rel(probe.acc_downweight.data, g["final::acc_down"]) < tol["acc"]
    assert rel(probe.downscale_weights[0].data, g["final::A"]) < tol["acc"]
    assert rel(probe.upscale_weights[0].data, g["final::B"]) < tol["final"]
    return losses, fired
