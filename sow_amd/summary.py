"""Module summary printer -- the `__colorized_str__` hook `simple_train.py:45-46` installs as `nn.Module.__str__`
(reference tn_gradient/utils.py:155-242 describes the contract: children coloured by trainability, runs of identical
siblings folded into one "N x ..." line).  Own implementation: ANSI escapes instead of termcolor (absent in this image),
run-length folding with itertools.groupby; the text layout follows nn.Module.__repr__.
"""
from __future__ import annotations

import os
import sys
from itertools import groupby

_ANSI = {"frozen": "\033[31m", "trainable": "\033[32m", "mixed": "\033[33m", "none": ""}
_RESET = "\033[0m"


def _use_colour() -> bool:
    if os.environ.get("NO_COLOR"):
        return False
    return hasattr(sys.stdout, "isatty") and sys.stdout.isatty() or bool(os.environ.get("FORCE_COLOR"))


def trainability(module) -> str:
    """'none' (no parameters), 'frozen', 'trainable' or 'mixed' over all parameters below `module`."""
    flags = {bool(p.requires_grad) for p in module.parameters(recurse=True)}
    if not flags:
        return "none"
    if flags == {True}:
        return "trainable"
    if flags == {False}:
        return "frozen"
    return "mixed"


def _paint(text: str, kind: str, colour: bool) -> str:
    code = _ANSI[kind]
    return f"{code}{text}{_RESET}" if (colour and code) else text


def _indent_tail(block: str, n: int) -> str:
    head, *tail = block.split("\n")
    pad = " " * n
    return "\n".join([head] + [pad + t for t in tail])


def _fold(children):
    """children: list of (key, text, kind).  Consecutive numbered siblings (ModuleList entries) with the same text and
    kind collapse into ('first-last', 'N x text', kind)."""
    out = []
    for (text, kind, numbered), run in groupby(children, key=lambda c: (c[1], c[2], c[0].isdigit())):
        run = list(run)
        if numbered and len(run) > 1:
            out.append((f"{run[0][0]}-{run[-1][0]}", f"{len(run)} x {text}", kind))
        else:
            out.extend(run)
    return out


def module_summary(module, colour=None) -> str:
    """Multi-line summary of `module`; `colour=None` decides from the terminal (NO_COLOR / FORCE_COLOR respected)."""
    if colour is None:
        colour = _use_colour()
    own = module.extra_repr()
    own_lines = own.split("\n") if own else []
    children = [(key, module_summary(child, colour), trainability(child))
                for key, child in module._modules.items() if child is not None]
    child_lines = [_indent_tail(f"{_paint('(' + key + '):', kind, colour)} {text}", 2) for key, text, kind in _fold(children)]
    name = module._get_name()
    if not own_lines and not child_lines:
        return name + "()"
    if len(own_lines) == 1 and not child_lines:
        return f"{name}({own_lines[0]})"
    return name + "(\n  " + "\n  ".join(own_lines + child_lines) + "\n)"


def __colorized_str__(self, indent=2):  # noqa: N807 - the reference's name (simple_train.py:45)
    return module_summary(self)
