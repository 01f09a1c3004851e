"""A module-shaped wrapper over the functional CPU oracle (oracle/sow_oracle.py), TEST INFRASTRUCTURE ONLY.

Gives tests/protocols.py a `backend` whose SoWLinear / prepare_sow / reset_optimizer run entirely on the oracle, so a whole
caller protocol (run_glue.py / finetune.py loops) can be replayed on the CPU and pinned against the reference-generated
fixtures (tests/test_oracle_golden.py), and then replayed on the HIP path against the same fixtures (-m gpu)."""
import types

import torch
import torch.nn as nn

from oracle import sow_oracle as O


class _OracleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, A, B, acc_down, acc_up, bias, scale):
        ctx.save_for_backward(x, A, B, acc_down, acc_up)
        ctx.scale, ctx.has_bias = scale, bias is not None
        return O.sow_forward(x, [A], [B], acc_down, acc_up, scale, bias)

    @staticmethod
    def backward(ctx, dy):
        x, A, B, acc_down, acc_up = ctx.saved_tensors
        dx, dA, dB, dbias = O.sow_backward(dy, x, [A], [B], acc_down, acc_up, ctx.scale, ctx.has_bias)
        return dx, dA[0], dB[0], None, None, dbias, None


class OracleSoWLinear(nn.Module):
    """n_iter = 1 layer with the reference's attribute names; arithmetic = oracle functions."""

    def __init__(self, in_features, out_features, bias, rank, scale, init_method):
        super().__init__()
        self.in_features, self.out_features, self.rank, self.n_iter = in_features, out_features, rank, 1
        self.scale, self.init_method = scale, init_method
        self.virtual_rank = min(rank, in_features, out_features)
        self.acc_downweight = nn.Parameter(torch.empty(0), requires_grad=False)
        self.acc_upweight = nn.Parameter(torch.empty(0), requires_grad=False)
        self.downscale_weights = nn.ParameterList([nn.Parameter(torch.zeros(in_features, rank))])
        self.upscale_weights = nn.ParameterList([nn.Parameter(torch.zeros(rank, out_features))])
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        self.next_draws = None       # set by the test before accumulate(): the re-initialisation draws (inputs)

    def forward(self, x):
        return _OracleFn.apply(x, self.downscale_weights[0], self.upscale_weights[0], self.acc_downweight,
                               self.acc_upweight, self.bias, float(self.scale))

    @torch.no_grad()
    def accumulate(self):
        nd, nu, acc_down, acc_up, vr = O.sow_accumulate(
            [self.downscale_weights[0].data], [self.upscale_weights[0].data], self.acc_downweight.data,
            self.acc_upweight.data, float(self.scale), self.virtual_rank, self.rank, 1, self.in_features,
            self.out_features, self.init_method, self.next_draws)
        self.acc_downweight = nn.Parameter(acc_down, requires_grad=False)
        self.acc_upweight = nn.Parameter(acc_up, requires_grad=False)
        self.virtual_rank = vr
        self.downscale_weights[0].data = nd[0]
        self.upscale_weights[0].data = nu[0]


class OracleConfig:
    def __init__(self, target_modules, rank=16, scale=1.0, device="cpu", init_method="normal_QR", decompose="keep"):
        self.target_modules, self.rank, self.scale, self.init_method, self.decompose = target_modules, rank, scale, init_method, decompose


def prepare_sow(model, config):
    flags = [(n, isinstance(m, nn.Linear)) for n, m in model.named_modules()]
    lookup = dict(model.named_modules())
    for name in O.replaced_module_names(flags, config.target_modules):
        lin = lookup[name]
        new = OracleSoWLinear(lin.in_features, lin.out_features, lin.bias is not None, config.rank, config.scale,
                              config.init_method)
        new.virtual_rank = min(lin.in_features, lin.out_features)                     # prepare.py:120
        if config.decompose == "keep":
            new.acc_downweight = nn.Parameter(O.decompose_keep(lin.weight.data), requires_grad=False)
        if lin.bias is not None:
            new.bias = lin.bias
        parent, _, child = name.rpartition(".")
        setattr(lookup[parent] if parent else model, child, new)
    return model


def reset_optimizer(optimizer, group_id):
    group = optimizer.param_groups[group_id]
    for p in group["params"]:
        st = optimizer.state[p]
        if st:
            st.update(O.reset_optimizer_state(st, group.get("amsgrad", False)))


def accumulate(model):
    for _, m in model.named_modules():
        if isinstance(m, OracleSoWLinear):
            m.accumulate()


BACKEND = types.SimpleNamespace(SoWLinear=OracleSoWLinear, SoWConfig=OracleConfig, prepare_sow=prepare_sow,
                                reset_optimizer=reset_optimizer, accumulate=accumulate)
