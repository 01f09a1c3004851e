"""Import shim used ONLY by tests/golden/make_golden.py (runs in the build container).

Makes the read-only reference at /root/reference importable by stubbing the
third-party modules it imports but that are absent offline (peft, opt_einsum,
termcolor, galore_torch).  None of the stubs carries SoW/TT arithmetic except
the opt_einsum shim, which forwards to torch.einsum.
Nothing under tests/ other than make_golden.py imports this file and it is
never used on the GPU box (the reference does not travel).
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _sym(names):
    table, out = {}, []
    for n in names:
        if n not in table:
            table[n] = chr(ord("a") + len(table)) if len(table) < 26 else chr(ord("A") + len(table) - 26)
        out.append(table[n])
    return table, out


def _interleaved_to_eq(args):
    """opt_einsum 'interleaved' call format -> (equation, operands)."""
    ops, subs = [], []
    i = 0
    table = {}

    def letters(names):
        s = ""
        for n in names:
            if n not in table:
                k = len(table)
                table[n] = chr(ord("a") + k) if k < 26 else chr(ord("A") + k - 26)
            s += table[n]
        return s

    out = None
    while i < len(args):
        if i + 1 < len(args) and not isinstance(args[i], (list, tuple)):
            ops.append(args[i])
            subs.append(letters(args[i + 1]))
            i += 2
        else:
            out = letters(args[i])
            i += 1
    if out is None:
        # implicit output: indices appearing once, sorted
        cnt = {}
        for s in subs:
            for c in s:
                cnt[c] = cnt.get(c, 0) + 1
        out = "".join(sorted(c for c, v in cnt.items() if v == 1))
    return ",".join(subs) + "->" + out, ops


def _contract(*args, **kw):
    if isinstance(args[0], str):
        return torch.einsum(args[0], *args[1:])
    eq, ops = _interleaved_to_eq(args)
    return torch.einsum(eq, *ops)


class _PathInfo:
    def __init__(self, eq):
        self.eq = eq


def _contract_path(*args, **kw):
    if isinstance(args[0], str):
        return [], _PathInfo(args[0])
    eq, _ = _interleaved_to_eq(args)
    return [], _PathInfo(eq)


class ContractExpression:
    def __init__(self, eq):
        self.eq = eq

    def __call__(self, *ops):
        return torch.einsum(self.eq, *ops)


def _contract_expression(eq, *shapes, **kw):
    return ContractExpression(eq)


def install():
    if "tn_gradient" in sys.modules:
        return
    peft = types.ModuleType("peft")

    class PeftConfig:  # base class only (prepare.py:27)
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

    class PeftModel:  # base class only (prepare.py:181)
        pass

    peft.PeftConfig, peft.PeftModel = PeftConfig, PeftModel
    sys.modules["peft"] = peft

    termcolor = types.ModuleType("termcolor")
    termcolor.colored = lambda s, *a, **k: s
    sys.modules["termcolor"] = termcolor

    galore = types.ModuleType("galore_torch")
    gp = types.ModuleType("galore_torch.galore_projector")

    class GaLoreProjector:
        pass

    gp.GaLoreProjector = GaLoreProjector
    galore.galore_projector = gp
    sys.modules["galore_torch"] = galore
    sys.modules["galore_torch.galore_projector"] = gp

    oe = types.ModuleType("opt_einsum")
    oe.contract = _contract
    oe.contract_path = _contract_path
    oe.contract_expression = _contract_expression
    oec = types.ModuleType("opt_einsum.contract")
    oec.ContractExpression = ContractExpression
    oe.contract_module = oec
    sys.modules["opt_einsum"] = oe
    sys.modules["opt_einsum.contract"] = oec

    sys.path.insert(0, REFERENCE_ROOT)
