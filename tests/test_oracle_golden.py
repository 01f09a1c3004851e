"""Pins oracle/sow_oracle.py against the golden vectors generated FROM THE REFERENCE
(tests/golden/make_golden.py).  CPU only.  Tolerances: fp32 1e-5 relative (north_star);
integer / index work bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err
from oracle import sow_oracle as O

FWD = ["cfg1_noacc", "cfg1_bias_dense", "cfg1_lowrank", "r50_3d", "r50_3d_dense", "niter2", "niter3_odd", "tiny_T1"]
TOL = 1e-5


def _factors(g):
    n = int(g["n_iter"])
    return [g[f"A{i}"] for i in range(n)], [g[f"B{i}"] for i in range(n)]


@pytest.mark.parametrize("name", FWD)
def test_forward_backward(name):
    g = load_golden("fwdbwd_" + name)
    down, up = _factors(g)
    y = O.sow_forward(g["x"], down, up, g.get("acc_down"), g.get("acc_up"), g["scale"], g.get("bias"))
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < TOL
    dx, d_down, d_up, dbias = O.sow_backward(g["dy"], g["x"], down, up, g.get("acc_down"), g.get("acc_up"),
                                             g["scale"], "bias" in g)
    assert rel_err(dx, g["dx"]) < TOL
    for i in range(len(down)):
        assert rel_err(d_down[i], g[f"dA{i}"]) < TOL
        assert rel_err(d_up[i], g[f"dB{i}"]) < TOL
    if "bias" in g:
        assert rel_err(dbias, g["dbias"]) < TOL


ACC = {"lowrank_grow": (48, 40, "normal_QR"), "niter2_normal": (36, 44, "normal"),
       "dense_prepare_style": (64, 96, "normal_QR"), "cfg1": (256, 256, "normal_QR")}


@pytest.mark.parametrize("name", list(ACC))
def test_accumulate_trace(name):
    g = load_golden("accumulate_" + name)
    d_in, d_out, init = ACC[name]
    rank, n_iter, n_calls = int(g["rank"]), int(g["n_iter"]), int(g["n_calls"])
    vr = int(g["vr0"])
    acc_down, acc_up = None, None
    for c in range(n_calls):
        down = [g[f"c{c}_A{i}_in"] for i in range(n_iter)]
        up = [g[f"c{c}_B{i}_in"] for i in range(n_iter)]
        draws = [g[f"c{c}_draw{i}"] for i in range(n_iter)]
        nd, nu, acc_down, acc_up, vr = O.sow_accumulate(down, up, acc_down, acc_up, g["scale"], vr, rank, n_iter,
                                                        d_in, d_out, init, draws)
        assert vr == int(g[f"c{c}_vr"])  # integer schedule: bit-exact
        assert tuple(acc_down.shape) == tuple(g[f"c{c}_acc_down"].shape)
        assert tuple(acc_up.shape) == tuple(g[f"c{c}_acc_up"].shape)
        assert rel_err(acc_down, g[f"c{c}_acc_down"]) < TOL
        if acc_up.numel():
            assert rel_err(acc_up, g[f"c{c}_acc_up"]) < TOL
        for i in range(n_iter):
            assert rel_err(nd[i], g[f"c{c}_A{i}_out"]) < TOL
            assert float(nu[i].abs().max()) == 0.0 and float(g[f"c{c}_B{i}_out"].abs().max()) == 0.0


def test_qr_svd():
    g = load_golden("qr_svd")
    for k in ("tall", "wide", "square", "gauss002", "rankdef"):
        q, r = O.qr_weight(g[f"{k}_in"], int(g[f"{k}_rank"]))
        assert rel_err(q, g[f"{k}_q"]) < TOL and rel_err(r, g[f"{k}_r"]) < TOL
        qf, rf = O.qr_weight(g[f"{k}_in"])
        assert rel_err(qf, g[f"{k}_qfull"]) < TOL and rel_err(rf, g[f"{k}_rfull"]) < TOL
    qb, rb = O.qr_weight(g["bf16_in"].to(torch.bfloat16), 6)
    assert qb.dtype == torch.bfloat16
    assert torch.equal(qb.float(), g["bf16_q"]) and torch.equal(rb.float(), g["bf16_r"])
    u, s, v = O.svd_weight(g["tall_in"], 6)
    assert rel_err(s, g["svd_s"]) < TOL
    assert rel_err(u.abs(), g["svd_u"].abs()) < 1e-4 and rel_err(v.abs(), g["svd_v"].abs()) < 1e-4
    # the adversarial case really is lossy: keep, don't fix
    q, r = O.qr_weight(g["rankdef_in"], 8)
    assert rel_err(q @ r, g["rankdef_in"]) > 0.1


def test_prepare_names_bit_exact():
    with open(os.path.join(GOLDEN, "prepare_names.json")) as f:
        P = json.load(f)
    for key in ("llama_60m", "llama_7b", "roberta"):
        got = O.replaced_module_names([(n, bool(l)) for n, l in P[key]["named_modules"]], P[key]["targets"])
        assert got == P[key]["replaced"]
    assert len(P["llama_60m"]["replaced"]) == 56 and len(P["roberta"]["replaced"]) == 72 and len(P["llama_7b"]["replaced"]) == 160


def test_prepare_keep():
    g = load_golden("prepare_keep")
    assert torch.equal(O.decompose_keep(g["w1"]), g["fc1_acc_down"])
    assert torch.equal(O.decompose_keep(g["w2"]), g["fc2_acc_down"])
    h = torch.tanh(O.sow_forward(g["x"], [g["fc1_A"]], [g["fc1_B"]], g["fc1_acc_down"], None, 0.5, g["b1"]))
    h = O.sow_forward(h, [g["fc2_A"]], [g["fc2_B"]], g["fc2_acc_down"], None, 0.5, None)
    y = h @ g["head_w"].t() + g["head_b"]
    assert rel_err(y, g["y"]) < TOL
    assert int(g["fc1_vr"]) == 12 and int(g["fc2_vr"]) == 6  # prepare.py:120


def test_decompose_qr_identity():
    torch.manual_seed(0)
    w = torch.randn(24, 40)  # nn.Linear weight [out, in]
    w_acc, a, b = O.decompose_qr(w, 4)
    assert rel_err(w_acc + a @ b, w.t()) < TOL  # W_acc + A.B == W^T


def test_reset_optimizer():
    g = load_golden("reset_optimizer")
    for i in (0, 1, 2):
        st = {k: g[f"before_p{i}_{k}"] for k in ("step", "exp_avg", "exp_avg_sq")}
        st = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in st.items()}
        new = O.reset_optimizer_state(st, amsgrad=False) if i > 0 else st
        for k in ("step", "exp_avg", "exp_avg_sq"):
            want = g[f"after_p{i}_{k}"]
            want = torch.as_tensor(want) if not torch.is_tensor(want) else want
            assert torch.equal(new[k].float(), want.float())


def _cores(g, prefix, n):
    return [g[f"{prefix}_core{i}"] for i in range(n)]


def test_tt_decompose_reconstruct():
    g = load_golden("tt_algebra")
    cores = O.tt_from_tensor(g["t216_in"], [1, 4, 4, 1])
    for c, w in zip(cores, _cores(g, "t216", 3)):
        assert c.shape == w.shape and rel_err(c, w) < TOL
    assert rel_err(O.tt_reconstruct(cores), g["t216_rec"]) < TOL
    for name, order in (("m81", 4), ("m100x60", 3), ("m50x37", 2)):
        ranks = [int(r) for r in g[f"{name}_ranks"]]
        cores = O.tt_from_matrix(g[f"{name}_in"], ranks)
        for c, w in zip(cores, _cores(g, name, order)):
            assert c.shape == w.shape and rel_err(c, w) < 5e-5
        assert rel_err(O.tt_to_matrix(cores, g[f"{name}_in"].shape), g[f"{name}_tomatrix"]) < 5e-5


def test_tt_algebra():
    g = load_golden("tt_algebra")
    tx, ty = _cores(g, "alg_tx", 3), _cores(g, "alg_ty", 3)
    for c, w in zip(O.tt_add(tx, ty), _cores(g, "alg_add", 3)):
        assert torch.equal(c, w)
    for c, w in zip(O.tt_mul(tx, ty), _cores(g, "alg_mul", 3)):
        assert rel_err(c, w) < TOL
    assert rel_err(O.tt_reconstruct(O.tt_mul(tx, ty)), g["alg_mul_rec"]) < TOL
    sub = O.tt_add(tx, O.tt_scale(ty, -1))
    for c, w in zip(sub, _cores(g, "alg_sub", 3)):
        assert rel_err(c, w) < TOL
    for cname in ("pos", "neg"):
        c = float(g[f"alg_scale_{cname}_c"])
        for a, w in zip(O.tt_scale(tx, c), _cores(g, f"alg_scale_{cname}", 3)):
            assert rel_err(a, w) < TOL
        c = float(g[f"alg_addc_{cname}_c"])
        for a, w in zip(O.tt_add_constant(tx, [1, 5, 5, 1], c), _cores(g, f"alg_addc_{cname}", 3)):
            assert rel_err(a, w) < TOL
    assert abs(O.tt_inner(tx, ty, "full") - float(g["alg_inner_full"])) < 1e-4 * abs(float(g["alg_inner_full"])) + 1e-3
    assert abs(O.tt_inner(tx, ty, "right") - float(g["alg_inner_right"])) < 1e-4 * abs(float(g["alg_inner_right"])) + 1e-4
    assert abs(O.tt_inner(tx, tx, "full") - float(g["alg_norm_full"])) < 1e-4 * abs(float(g["alg_norm_full"]))
    oc, _ = O.tt_orthogonalize_right(tx, [1, 5, 5, 1], (3, 3, 3), (4, 4, 4))
    for a, w in zip(oc, _cores(g, "alg_orthoR", 3)):
        assert rel_err(a, w) < 5e-5
    s = O.tt_add(tx, ty)
    rc, rr = O.tt_round(s, [1, 10, 10, 1], (3, 3, 3), (4, 4, 4), [1, 5, 5, 1])
    assert rr == [int(r) for r in g["alg_round_ranks"]]
    assert rel_err(O.tt_reconstruct(rc), g["alg_round_rec"]) < 1e-4


def test_tt_integer_bit_exact():
    with open(os.path.join(GOLDEN, "tt_integer.json")) as f:
        J = json.load(f)
    for d, vals in J["ceil_root"].items():
        assert [O.tt_core_dim(n, int(d)) for n in range(1, 5001)] == vals
    for key, v in J["ceil_root_special"].items():
        n, d = map(int, key.split(","))
        assert O.tt_core_dim(n, d) == v
    assert O.tt_core_dim(3125, 5) == 6 and O.tt_core_dim(32768, 5) == 9  # pow rounding quirk
    for key, v in J["closest_factorization"].items():
        n, d = map(int, key.split(","))
        r = O.closest_factorization(n, d)
        assert (None if r is None else [list(r[0]), r[1]]) == v
    assert O.closest_factorization(1376, 3) == ([12, 11, 11], 1320)  # stale product quirk


def test_tt_optimizers():
    g = load_golden("tt_optim")
    ranks = [1, 4, 4, 4, 1]
    for wd_name, wd in (("nowd", 0.0), ("wd", 0.1)):
        p, st = g[f"adam_{wd_name}_p0"], {}
        for s in range(3):
            p, st = O.ttadam_step(p, g[f"adam_{wd_name}_g{s}"], st, lr=1e-2, weight_decay=wd, ranks=ranks)
            assert rel_err(p, g[f"adam_{wd_name}_p{s + 1}"]) < 2e-5
    p, st = g["adam_dense_p0"], {}
    for s in range(2):
        p, st = O.ttadam_step(p, g[f"adam_dense_g{s}"], st, lr=5e-3)
        assert rel_err(p, g[f"adam_dense_p{s + 1}"]) < TOL
    for name, kw in (("mom", dict(momentum=0.9)), ("nomom", dict(momentum=0.0)),
                     ("nesterov", dict(momentum=0.8, nesterov=True, dampening=0.1))):
        p, st = g[f"sgd_{name}_p0"], {}
        for s in range(3):
            p, st = O.ttsgd_step(p, g[f"sgd_{name}_g{s}"], st, lr=1e-2, ranks=ranks, **kw)
            assert rel_err(p, g[f"sgd_{name}_p{s + 1}"]) < 2e-5


def test_tt_linear():
    g = load_golden("tt_linear")
    y = O.tt_linear_forward(g["x"], _cores(g, "", 3) if False else [g["core0"], g["core1"], g["core2"]], 100, 60)
    assert y.shape == g["y"].shape and rel_err(y, g["y"]) < TOL


# ---------------------------------------------------------------------------------------------
# round 2: caller protocols of BASELINE configs 4 / 5 replayed on the oracle (tests/oracle_backend.py)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("proto", ["glue", "finetune"])
def test_protocol_trace_oracle(proto):
    """run_glue.py:976-1002 / finetune.py:39-77 on the RoBERTa-shaped trio (768/3072, r = 8, fp32, keep): losses, factor
    and bias gradients, dense-accumulator probes after every accumulate(), scale -> 1/rank schedule, final factors."""
    import oracle_backend
    import protocols as P
    g = load_golden("protocol_" + proto)

    def set_draw(m, draw):
        m.next_draws = [draw]

    torch.set_num_threads(min(8, os.cpu_count() or 1))
    errs = P.replay_and_check(oracle_backend.BACKEND, g, proto, "cpu", set_draw,
                              dict(loss=1e-6, grad=1e-5, acc=1e-5, final=1e-5))
    assert len(errs) > 20


def test_checkpointed_layer_oracle():
    """Activation checkpointing (BASELINE config 5; simple_train.py:423): the reference's gradients are bit-identical with
    and without torch.utils.checkpoint (asserted when the fixture was made); the oracle reproduces them."""
    g = load_golden("checkpoint_layer")
    acc = O.decompose_keep(g["W"])
    xt = torch.tanh(g["x"])
    y = O.sow_forward(xt, [g["A"]], [g["B"]], acc, None, g["scale"], None)
    assert rel_err(y, g["plain_y"]) < TOL
    dxt, dA, dB, _ = O.sow_backward(g["dy"], xt, [g["A"]], [g["B"]], acc, None, g["scale"], False)
    assert rel_err(dxt * (1 - xt * xt), g["plain_dx"]) < TOL
    assert rel_err(dA[0], g["plain_dA"]) < TOL and rel_err(dB[0], g["plain_dB"]) < TOL


def test_tt_newton_and_reciprocal():
    """a11: TensorTrain.sqrt / sqrtinv / reciprocal (tt.py:279-341, 480-494) against reference outputs; the first case is
    the one tests/tt_test.py:1-13 prints."""
    g = load_golden("tt_newton")
    a = torch.arange(2 * 2 * 2 * 3 * 3 * 3).reshape(2, 2, 2, 3, 3, 3).float()
    cores = O.tt_from_tensor(a, [1, 4, 4, 1])
    got = O.tt_reconstruct(O.tt_sqrt(cores))
    print("t216", rel_err(got, g["t216_sqrt_rec"]))
    pos = [g[f"pos_core{i}"] for i in range(3)]
    for name, fn in (("sqrt", lambda c: O.tt_sqrt(c)), ("sqrt_it2", lambda c: O.tt_sqrt(c, max_iter=2)),
                     ("sqrtinv", lambda c: O.tt_sqrtinv(c)), ("sqrtinv_it2", lambda c: O.tt_sqrtinv(c, threshold=None, max_iter=2))):
        out = fn(pos)
        assert O._tt_meta(out)[0] == [int(r) for r in g[f"pos_{name}_ranks"]], name
        e = rel_err(O.tt_reconstruct(out), g[f"pos_{name}_rec"])
        print(name, e)
        assert e < 1e-4, (name, e)
    rec = O.tt_reciprocal([g[f"recip_in{i}"] for i in range(3)])
    for i in range(3):
        assert rel_err(rec[i], g[f"recip_out{i}"]) < 1e-5


def test_pretrain_protocol_gradient_accumulation_3_oracle():
    """a14: simple_train.py:596-650 with gradient_accumulation = 3 on a tiny Llama -- the accumulate predicate (:618-626) fires on
    micro-steps 10 and 11 (two micro-steps of the same update, never its last one); loss trace and final state vs the reference."""
    pytest.importorskip("transformers")
    import oracle_backend
    import protocols as P
    g = load_golden("train_trace_ga3")

    def set_draw(m, draw):
        m.next_draws = [draw]

    losses, fired = P.replay_pretrain_ga(oracle_backend.BACKEND, g, "cpu", set_draw, dict(loss=2e-6, acc=1e-4, final=1e-4))
    assert fired == [10, 11] and len(losses) == 15
