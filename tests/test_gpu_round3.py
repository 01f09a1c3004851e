"""-m gpu: round-3 surface -- forward without the saved projection (no-grad callers), accumulator validation on the grouped
path, FactorAdamW hyper-parameters written through param_groups, and (further down) the round's new kernels.

Tolerances: `rel_err` = max |a - b| / max |b| (relative to the largest reference magnitude, not element-wise).
"""
import pytest
import torch
import torch.nn as nn

from conftest import rel_err
from oracle import sow_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _layer(d_in, d_out, r, dtype, bias=False, dense=False, seed=0):
    from sow_amd import SoWLinear
    torch.manual_seed(seed)
    m = SoWLinear(d_in, d_out, bias=bias, rank=r, init_method="normal", device=DEV, dtype=dtype)
    nn.init.normal_(m.downscale_weights[0], std=0.05)
    nn.init.normal_(m.upscale_weights[0], std=0.05)
    if bias:
        nn.init.normal_(m.bias, std=0.1)
    if dense:
        m.acc_downweight = nn.Parameter((torch.randn(d_in, d_out, device=DEV) * 0.02).to(dtype), requires_grad=False)
    return m


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(64, 256, 256, 8), (16384, 512, 1376, 50), (1024, 768, 3072, 8), (300, 96, 40, 5)])
@pytest.mark.parametrize("dense", [False, True])
def test_forward_without_saved_projection_is_bit_identical(shape, dtype, dense):
    """torch.no_grad() callers (eval / generate of scripts/commonsense_evaluate.py:268-287, the first pass of activation
    checkpointing) run sow_forward with h_save = NULL: same y bit for bit as the training forward, and the oracle's."""
    T, d_in, d_out, r = shape
    m = _layer(d_in, d_out, r, dtype, bias=True, dense=dense)
    x = torch.randn(T, d_in, device=DEV).to(dtype)
    y_train = m(x.clone().requires_grad_(True)).detach()
    with torch.no_grad():
        y_eval = m(x)
    assert not y_eval.requires_grad
    if dense and r <= 64:
        # with a dense accumulator and no h_save the low-rank term is added by a second kernel (one more rounding in bf16)
        tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
        assert rel_err(y_eval.float().cpu(), y_train.float().cpu()) < tol
    else:
        assert torch.equal(y_eval, y_train)
    acc = m.acc_downweight.data.float().cpu() if dense else None
    y_ref = O.sow_forward(x.float().cpu(), [m.downscale_weights[0].data.float().cpu()], [m.upscale_weights[0].data.float().cpu()],
                          acc, None, 1.0, m.bias.data.float().cpu())
    assert rel_err(y_eval.float().cpu(), y_ref) < (2e-2 if dtype == torch.bfloat16 else 1e-5)


def test_no_grad_forward_allocates_no_projection_buffer():
    from sow_amd import ops
    x = torch.randn(4096, 512, device=DEV, dtype=torch.bfloat16)
    A = torch.randn(512, 50, device=DEV, dtype=torch.bfloat16) * 0.05
    B = torch.randn(50, 512, device=DEV, dtype=torch.bfloat16) * 0.05
    y1, h1 = ops.sow_forward(x, A, B, None, None, None, 1.0)
    y0, h0 = ops.sow_forward(x, A, B, None, None, None, 1.0, save_h=False)
    assert h0 is None and h1 is not None and torch.equal(y0, y1)


def test_grouped_no_grad_forward_matches_single_calls():
    """group_siblings under torch.no_grad(): q / k / v share one launch, nothing is saved, outputs bit-identical."""
    from sow_amd import group_siblings, ungroup_siblings

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = _layer(512, 512, 50, torch.bfloat16, seed=1)
            self.k_proj = _layer(512, 512, 50, torch.bfloat16, seed=2)
            self.v_proj = _layer(512, 512, 50, torch.bfloat16, seed=3)

        def forward(self, h):
            return self.q_proj(h), self.k_proj(h), self.v_proj(h)

    net = Attn()
    x = torch.randn(128, 128, 512, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        single = net(x)
        assert group_siblings(net) == 1
        grouped = net(x)
        ungroup_siblings(net)
    for a, b in zip(single, grouped):
        assert torch.equal(a, b)


@pytest.mark.parametrize("bad", ["dtype", "shape"])
def test_grouped_path_validates_the_accumulator_like_the_single_call(bad):
    """A layer whose dense accumulator was loaded in another precision / shape (load_sow of a foreign checkpoint): the
    single call raises TypeError / ValueError; LayerCall raises the same, and a sibling group falls back to the single call
    (which raises) instead of handing a mistyped pointer to the grouped launch."""
    from sow_amd import group_siblings, ops
    x = torch.randn(16384, 512, device=DEV, dtype=torch.bfloat16)
    A = (torch.randn(512, 50, device=DEV) * 0.05).bfloat16()
    B = (torch.randn(50, 512, device=DEV) * 0.05).bfloat16()
    acc = torch.randn(512, 512, device=DEV) * 0.02 if bad == "dtype" else (torch.randn(512, 256, device=DEV) * 0.02).bfloat16()
    exc = TypeError if bad == "dtype" else ValueError
    with pytest.raises(exc):
        ops.sow_forward(x, A, B, acc, None, None, 1.0)
    with pytest.raises(exc):
        ops.LayerCall(x, A, B, acc_down=acc)

    class Mlp(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj = _layer(512, 512, 50, torch.bfloat16, seed=1)
            self.up_proj = _layer(512, 512, 50, torch.bfloat16, seed=2)

        def forward(self, h):
            return self.gate_proj(h) * self.up_proj(h)

    net = Mlp()
    net.up_proj.acc_downweight = nn.Parameter(acc, requires_grad=False)
    assert group_siblings(net) == 1
    with pytest.raises(exc):
        net(x.requires_grad_(True))


def test_factor_adamw_steps_with_the_lr_a_scheduler_wrote():
    from sow_amd.dp import FactorBucket
    from sow_amd.optimizer import FactorAdamW
    ps = [nn.Parameter(torch.randn(64, 16, device=DEV)), nn.Parameter(torch.randn(16, 64, device=DEV))]
    ref = [nn.Parameter(p.detach().clone()) for p in ps]
    bucket = FactorBucket(ps)
    opt = FactorAdamW(bucket, lr=3e-3, weight_decay=0.0)
    topt = torch.optim.AdamW(ref, lr=3e-3, weight_decay=0.0)
    for step_lr in (3e-3, 1e-3, 5e-4):
        for g in opt.param_groups:
            g["lr"] = step_lr                          # simple_train.py-style scheduler write, no read-back in between
        for g in topt.param_groups:
            g["lr"] = step_lr
        for p, q in zip(ps, ref):
            grad = torch.randn_like(q)
            p.grad.copy_(grad)
            q.grad = grad.clone()
        opt.step()
        topt.step()
        assert opt.state_dict()["lr"] == step_lr
    for p, q in zip(ps, ref):
        assert rel_err(p.data.cpu(), q.data.cpu()) < 1e-5
