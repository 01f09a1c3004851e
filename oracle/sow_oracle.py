"""CPU ORACLE for the SoW hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, on torch-CPU tensors, the arithmetic of the reference
(antoine311200/sow, read-only at /root/reference) for the path named by
BASELINE.json:north_star.  Every function cites the reference file:line it
follows.  It is a *checker*:

  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import it;
  * nothing under sow_amd/ imports it, and the product path raises when the HIP
    library is missing instead of falling back to this file.

Parity pinning: the reference's own tests hold no golden vectors (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself,
generated in the build container by tests/golden/make_golden.py (which imports
/root/reference with stub modules for absent third-party packages) and
committed as tests/golden/*.npz.  tests/test_oracle_golden.py checks every
function below against those vectors.

The API is functional (tensors in, tensors out); module/optimizer state is
passed explicitly so the same calls can be replayed against the HIP library.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# L0 numerics helpers  (tn_gradient/utils.py)
# --------------------------------------------------------------------------


def qr_weight(weight: Tensor, rank: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """Reduced Householder QR in fp32, truncated to `rank` columns / rows.

    Follows tn_gradient/utils.py:8-30: non-fp32 input is up-cast (:13-17),
    torch.linalg.qr reduced mode (:19), Q[:, :rank] / R[:rank, :] (:20-22),
    cast back to the input dtype (:26-28).  `rank` falsy (None or 0) means
    no truncation, as in the reference's `if rank:`.
    """
    src_dtype = weight.dtype
    w = weight if src_dtype == torch.float32 else weight.to(torch.float32)
    q, r = torch.linalg.qr(w)
    if rank:
        q, r = q[:, :rank], r[:rank, :]
    if src_dtype != torch.float32:
        q, r = q.to(src_dtype), r.to(src_dtype)
    return q, r


def svd_weight(weight: Tensor, rank: Optional[int] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Truncated SVD with fp32 up-cast (tn_gradient/utils.py:32-57)."""
    src_dtype = weight.dtype
    w = weight if src_dtype == torch.float32 else weight.to(torch.float32)
    u, s, vh = torch.linalg.svd(w)
    if rank:
        u, s, vh = u[:, :rank], s[:rank], vh[:rank, :]
    if src_dtype != torch.float32:
        u, s, vh = u.to(src_dtype), s.to(src_dtype), vh.to(src_dtype)
    return u, s, vh


def pad_matrix(matrix: Tensor, new_shape: Sequence[int]) -> Tensor:
    """Zero-pad to `new_shape`, top-left aligned (utils.py:78-84).

    The reference allocates the padded matrix with torch.zeros default dtype
    (fp32) on the input's device, so a non-fp32 input is silently up-cast.
    """
    out = torch.zeros(tuple(new_shape), device=matrix.device)
    out[: matrix.shape[0], : matrix.shape[1]] = matrix
    return out


def unpad_matrix(matrix: Tensor, shape: Sequence[int]) -> Tensor:
    """Top-left crop (utils.py:86-87)."""
    return matrix[: shape[0], : shape[1]]


def tt_core_dim(n: int, order: int) -> int:
    """ceil(n ** (1/order)) in double precision (tt.py:53-54, tensor_linear.py:20-21).

    Integer result must be bit-exact with the reference, including the cases
    where pow rounds up (3125**(1/5) -> 6); hence the same float expression.
    """
    return math.ceil(n ** (1 / order))


def closest_factorization(n: int, d: int):
    """utils.py:89-99, including the stale product it returns.

    Loop: k = ceil(n**(1/d)); append k; n //= k; p *= k; d -= 1; when n hits
    1, if p < original n the last factor is bumped by n (=1) and (factors, p)
    is returned with p NOT recomputed.  Falls off the loop (returns None) when
    n starts <= 1.
    """
    factors: List[int] = []
    p, original = 1, n
    while n > 1:
        k = math.ceil(n ** (1 / d))
        factors.append(k)
        n, p, d = n // k, p * k, d - 1
        if n == 1:
            if p < original:
                factors[-1] += n
            return factors, p
    return None


# --------------------------------------------------------------------------
# L1 SoWLinear forward / backward / accumulate  (tn_gradient/layer/sow.py)
# --------------------------------------------------------------------------


def sow_forward(
    x: Tensor,
    down: Sequence[Tensor],
    up: Sequence[Tensor],
    acc_down: Optional[Tensor],
    acc_up: Optional[Tensor],
    scale: float,
    bias: Optional[Tensor],
) -> Tensor:
    """y = acc-term + sum_i (x @ A_i @ B_i) * scale + bias  (sow.py:107-126).

    acc-term: (x @ acc_down) @ acc_up when both are non-empty (:109-110),
    x @ acc_down when only acc_down is non-empty (:111-112), absent otherwise.
    The accumulator term is NOT multiplied by `scale`; each live pair is
    (:117-121).  Products are evaluated left to right, in the input dtype.
    """
    has_down = acc_down is not None and acc_down.numel() != 0
    has_up = acc_up is not None and acc_up.numel() != 0
    out = None
    if has_down and has_up:
        out = (x @ acc_down) @ acc_up
    elif has_down:
        out = x @ acc_down
    for a, b in zip(down, up):
        term = (x @ a @ b) * scale
        out = term if out is None else out + term
    if bias is not None:
        out = out + bias
    return out


def sow_backward(
    dy: Tensor,
    x: Tensor,
    down: Sequence[Tensor],
    up: Sequence[Tensor],
    acc_down: Optional[Tensor],
    acc_up: Optional[Tensor],
    scale: float,
    has_bias: bool,
):
    """Closed-form gradients of sow_forward (what autograd derives from sow.py:107-126).

    dh_i = scale * dY @ B_i^T ; dB_i = scale * (x A_i)^T @ dY ; dA_i = x^T @ dh_i ;
    dX = sum_i dh_i @ A_i^T + acc-term ; dbias = sum over tokens of dY.
    acc-term: (dY @ acc_up^T) @ acc_down^T (low-rank) or dY @ acc_down^T (dense).
    Returns (dX, [dA_i], [dB_i], dbias or None).  Leading dims are flattened
    for the reductions, as torch.matmul's broadcasting backward does.
    """
    d_in = x.shape[-1]
    d_out = dy.shape[-1]
    x2 = x.reshape(-1, d_in)
    dy2 = dy.reshape(-1, d_out)
    has_down = acc_down is not None and acc_down.numel() != 0
    has_up = acc_up is not None and acc_up.numel() != 0
    dx2 = None
    if has_down and has_up:
        dx2 = (dy2 @ acc_up.t()) @ acc_down.t()
    elif has_down:
        dx2 = dy2 @ acc_down.t()
    d_down, d_up = [], []
    for a, b in zip(down, up):
        h = x2 @ a
        dh = (dy2 * scale) @ b.t()
        d_up.append(h.t() @ (dy2 * scale))
        d_down.append(x2.t() @ dh)
        term = dh @ a.t()
        dx2 = term if dx2 is None else dx2 + term
    dbias = dy2.sum(dim=0) if has_bias else None
    return dx2.reshape(x.shape), d_down, d_up, dbias


def sow_accumulate(
    down: Sequence[Tensor],
    up: Sequence[Tensor],
    acc_down: Optional[Tensor],
    acc_up: Optional[Tensor],
    scale: float,
    virtual_rank: int,
    rank: int,
    n_iter: int,
    in_features: int,
    out_features: int,
    init_method: str,
    reinit_draws: Sequence[Tensor],
):
    """One call of SoWLinear.accumulate (sow.py:128-178) as a pure function.

    Steps: (i) acc = scale * sum_i A_i @ B_i via stack+sum (:131-134);
    (ii) add the previous accumulator, Q@R or dense (:137-140);
    (iii) if virtual_rank < min(in, out): truncated QR of acc at virtual_rank,
    store Q, R, then virtual_rank = min(vr + rank*n_iter, in, out) (:144-150);
    else store acc dense and an empty acc_up (:151-153);
    (v) new B_i = 0 (:159); new A_i = Q[:, :rank] of the QR of a fresh
    N(0, 0.02^2) [in, out] matrix when init_method == "normal_QR" (:168-172),
    else a fresh N(0, 0.02^2) [in, rank] matrix (:174).

    RNG streams are not portable between libraries, so the fresh Gaussian
    draws are INPUTS here: `reinit_draws[i]` is the [in, out] matrix (normal_QR)
    or the [in, rank] matrix (otherwise) for pair i, already in the dtype the
    reference would draw it in (the accumulator dtype for normal_QR, :163-165).

    Returns (new_down, new_up, new_acc_down, new_acc_up, new_virtual_rank).
    """
    acc = scale * torch.sum(torch.stack([a @ b for a, b in zip(down, up)]), dim=0)
    has_down = acc_down is not None and acc_down.numel() != 0
    has_up = acc_up is not None and acc_up.numel() != 0
    if has_down and has_up:
        acc = acc + acc_down @ acc_up
    elif has_down:
        acc = acc + acc_down
    if virtual_rank < min(in_features, out_features):
        q, r = qr_weight(acc, rank=virtual_rank)
        new_acc_down, new_acc_up = q.contiguous(), r.contiguous()
        virtual_rank = min(virtual_rank + rank * n_iter, in_features, out_features)
    else:
        new_acc_down, new_acc_up = acc.contiguous(), torch.empty(0)
    new_up = [torch.zeros_like(b) for b in up]
    new_down = []
    for i in range(n_iter):
        if init_method == "normal_QR":
            q, _ = qr_weight(reinit_draws[i], rank)
            new_down.append(q.contiguous())
        else:
            new_down.append(reinit_draws[i].clone())
    return new_down, new_up, new_acc_down, new_acc_up, virtual_rank


def sow_init_normal_qr(draw: Tensor, rank: int, dtype: torch.dtype) -> Tuple[Tensor, Tensor]:
    """reset_parameters, normal_QR branch (sow.py:94-99): A = Q[:, :r], B = R[:r, :]
    of the QR of an fp32 N(0, 0.02^2) [in, out] draw, cast to the parameter dtype."""
    q, r = qr_weight(draw, rank)
    return q.to(dtype).contiguous(), r.to(dtype).contiguous()


# --------------------------------------------------------------------------
# L2 model surgery  (tn_gradient/prepare.py)
# --------------------------------------------------------------------------


def match_target(name: str, is_linear: bool, target_modules: Sequence[str]) -> bool:
    """check_module (prepare.py:72-83): suffix match on dotted module names.

    max_split = longest target in dotted components; a Linear matches when its
    whole name (if it has one component) is a target, or when the last i
    components, i in [1, min(max_split + 1, n_components) ), joined by '.',
    are a target.  Note the exclusive upper bound: a name with n components is
    never compared through all n of them.
    """
    if not is_linear:
        return False
    max_split = max(len(t.split(".")) for t in target_modules)
    parts = name.split(".")
    if len(parts) == 1 and parts[0] in target_modules:
        return True
    for i in range(1, min(max_split + 1, len(parts))):
        if ".".join(parts[-i:]) in target_modules:
            return True
    return False


def replaced_module_names(named_linear_flags: Sequence[Tuple[str, bool]], target_modules: Sequence[str]) -> List[str]:
    """Names prepare_sow swaps, in model.named_modules() order (prepare.py:98-99)."""
    return [n for n, is_lin in named_linear_flags if match_target(n, is_lin, target_modules)]


def decompose_qr(weight: Tensor, rank: int):
    """prepare_sow decompose='qr' branch (prepare.py:122-139) on CPU.

    Q, R = qr(W^T) with W the nn.Linear weight [out, in]; the accumulator is
    Q[:, :-rank] @ R[:-rank, :], the live factors are the LAST `rank` columns of
    Q and rows of R.  (The reference hard-codes .to("cuda") at :124; the
    arithmetic is device independent.)  Returns (W_acc [in,out], A [in,rank], B [rank,out]).
    """
    q, r = torch.linalg.qr(weight.t())
    w_acc = q[:, :-rank] @ r[:-rank, :]
    return w_acc, q[:, -rank:], r[-rank:, :]


def decompose_keep(weight: Tensor) -> Tensor:
    """prepare_sow decompose='keep' (prepare.py:148-150): acc_down = W^T contiguous."""
    return weight.t().contiguous()


def reset_optimizer_state(state: Dict[str, Tensor], amsgrad: bool) -> Dict[str, Tensor]:
    """reset_optimizer for one parameter (scripts/utils/training_utils.py:257-277):
    exp_avg, exp_avg_sq (and max_exp_avg_sq under amsgrad) become zeros, `step`
    becomes a zero of its own type if present."""
    out = dict(state)
    out["exp_avg"] = torch.zeros_like(state["exp_avg"])
    out["exp_avg_sq"] = torch.zeros_like(state["exp_avg_sq"])
    if amsgrad:
        out["max_exp_avg_sq"] = torch.zeros_like(state["exp_avg"])
    if "step" in state:
        out["step"] = torch.zeros_like(state["step"])
    return out


# --------------------------------------------------------------------------
# L0 TensorTrain  (tn_gradient/tt.py) -- cores are [r_k, i_k, o_k, r_{k+1}]
# --------------------------------------------------------------------------


def tt_decompose(tensor: Tensor, ranks: Sequence[int], in_shape: Sequence[int], out_shape: Sequence[int]) -> List[Tensor]:
    """Sequential truncated QR (tt.py:111-140) of an INTERLEAVED tensor
    (i1,o1,i2,o2,...).  For k < order-1: L = reshape(r_k*i_k*o_k, -1);
    Q, R = qr(L, mode='complete'); keep Q[:, :r_{k+1}], R[:r_{k+1}, :]."""
    order = len(ranks) - 1
    cores = []
    rest = tensor
    for k in range(order - 1):
        rows = ranks[k] * in_shape[k] * out_shape[k]
        left = rest.reshape(rows, -1)
        q, r = torch.linalg.qr(left, mode="complete")
        q, r = q[:, : ranks[k + 1]], r[: ranks[k + 1], :]
        cores.append(q.reshape(ranks[k], in_shape[k], out_shape[k], ranks[k + 1]))
        rest = r
    cores.append(rest.reshape(ranks[-2], in_shape[-1], out_shape[-1], ranks[-1]))
    return cores


def tt_from_tensor(tensor: Tensor, ranks: Sequence[int]) -> List[Tensor]:
    """tt.py:26-35: tensor axes are (*in_shape, *out_shape); permute to
    interleaved order (:33) and decompose."""
    order = len(ranks) - 1
    half = tensor.dim() // 2
    in_shape, out_shape = tuple(tensor.shape[:half]), tuple(tensor.shape[half:])
    perm = [ax for pair in zip(range(order), range(order, 2 * order)) for ax in pair]
    return tt_decompose(tensor.permute(*perm), ranks, in_shape, out_shape)


def tt_from_matrix(matrix: Tensor, ranks: Sequence[int], padding: bool = True) -> List[Tensor]:
    """tt.py:48-67: mm = ceil(M**(1/order)), nn likewise; zero-pad to
    mm^order x nn^order; reshape to (mm,)*order + (nn,)*order; from_tensor."""
    order = len(ranks) - 1
    m, n = matrix.shape
    mm, nn = tt_core_dim(m, order), tt_core_dim(n, order)
    if padding:
        matrix = pad_matrix(matrix, (mm ** order, nn ** order))
    return tt_from_tensor(matrix.reshape((mm,) * order + (nn,) * order), ranks)


def tt_reconstruct(cores: Sequence[Tensor]) -> Tensor:
    """tt.py:213-237: contract the bond indices; output axes (i1..in, o1..on).

    Contraction order here is left-to-right; the reference lets opt_einsum
    choose, so fp32 results may differ at the 1e-7 level."""
    order = len(cores)
    acc = cores[0]  # [r0, i1, o1, r1]
    r0 = acc.shape[0]
    acc = acc.reshape(r0, -1, acc.shape[-1])  # [r0, (i1 o1), r1]
    dims = [(cores[0].shape[1], cores[0].shape[2])]
    for c in cores[1:]:
        rk, ik, ok, rn = c.shape
        acc = (acc.reshape(-1, rk) @ c.reshape(rk, -1)).reshape(r0, -1, rn)
        dims.append((ik, ok))
    # acc: [r0, i1 o1 i2 o2 ..., r_n] with r0 = r_n = 1 summed out like einsum does
    full = acc.sum(dim=0).sum(dim=-1) if (r0 != 1 or acc.shape[-1] != 1) else acc.reshape(-1)
    full = full.reshape([d for pair in dims for d in pair])
    perm = list(range(0, 2 * order, 2)) + list(range(1, 2 * order, 2))
    return full.permute(*perm)


def tt_to_matrix(cores: Sequence[Tensor], shape: Sequence[int]) -> Tensor:
    """tt.py:242-247: reshape [prod(in), prod(out)] then crop to `shape`."""
    t = tt_reconstruct(cores)
    order = len(cores)
    rows = int(math.prod(t.shape[:order]))
    cols = int(math.prod(t.shape[order:]))
    return unpad_matrix(t.reshape(rows, cols), shape)


def tt_add(a: Sequence[Tensor], b: Sequence[Tensor]) -> List[Tensor]:
    """tt.py:382-422: first core concatenated on the right bond, last core on
    the left bond, middle cores block-diagonal on both bonds."""
    order = len(a)
    out = []
    for i in range(order):
        ca, cb = a[i], b[i]
        if i == 0:
            out.append(torch.cat((ca, cb), dim=-1))
        elif i == order - 1:
            out.append(torch.cat((ca, cb), dim=0))
        else:
            ra0, _, _, ra1 = ca.shape
            rb0, _, _, rb1 = cb.shape
            core = torch.zeros(ra0 + rb0, ca.shape[1], ca.shape[2], ra1 + rb1, dtype=ca.dtype)
            core[:ra0, :, :, :ra1] = ca
            core[ra0:, :, :, ra1:] = cb
            out.append(core)
    return out


def tt_scale(cores: Sequence[Tensor], constant: float) -> List[Tensor]:
    """tt.py:428-447 (__rmul__): every core times sign(c) * |c|**(1/order).
    The sign lands on ALL cores, so an even order with c < 0 yields +|c| -- kept."""
    order = len(cores)
    sub = abs(constant) ** (1 / order)
    sgn = -1 if constant < 0 else 1
    return [c * (sgn * sub) for c in cores]


def tt_mul(a: Sequence[Tensor], b: Sequence[Tensor]) -> List[Tensor]:
    """tt.py:449-478: Hadamard product; per core einsum('aijb,cijd->acijbd')
    reshaped to [ra*rb, i, o, ra'*rb']."""
    out = []
    for ca, cb in zip(a, b):
        core = torch.einsum("aijb,cijd->acijbd", ca, cb)
        out.append(core.reshape(ca.shape[0] * cb.shape[0], ca.shape[1], ca.shape[2], ca.shape[3] * cb.shape[3]))
    return out


def tt_add_constant(cores: Sequence[Tensor], ranks: Sequence[int], constant: float) -> List[Tensor]:
    """tt.py:343-379 (add_): adds a rank-1 constant train whose every entry is
    sign * (|constant| / prod(ranks)) ** (1/order), block-concatenated like tt_add
    (the constant cores have the SAME shapes as the originals, :359)."""
    order = len(cores)
    sub = constant / math.prod(ranks)
    neg = sub < 0
    sub = abs(sub) ** (1 / order)
    fill = (-1 if neg else 1) * sub
    return tt_add(cores, [torch.full_like(c, fill) for c in cores])


def tt_inner(a: Sequence[Tensor], b: Sequence[Tensor], mode: str = "right") -> float:
    """tt.py:257-277.  mode='full': full contraction of both trains.
    mode='right': only the LAST cores, contracted over (left bond, in, out) with
    the right bonds left free, then squeezed (valid when those bonds are 1)."""
    if mode == "full":
        env = None
        for ca, cb in zip(a, b):
            if env is None:
                env = torch.einsum("aijb,cijd->acbd", ca, cb).sum(dim=(0, 1))
            else:
                env = torch.einsum("ac,aijb,cijd->bd", env, ca, cb)
        return float(env.sum().squeeze()) if env.numel() != 1 else float(env.squeeze())
    la, lb = a[-1], b[-1]
    return float(torch.einsum("aijb,aijd->bd", la, lb).squeeze())


def tt_orthogonalize_right(cores: List[Tensor], ranks: List[int], in_shape, out_shape, new_ranks=None):
    """tt.py:159-177 (mode='right', in place on copies): for k = order-1 .. 1:
    Q,S = qr(right_matrix(k)^T); W = left_matrix(k-1) @ S^T; optional truncation;
    core[k-1] = W, core[k] = Q^T; ranks[k] = W.shape[1]."""
    order = len(cores)
    cores, ranks = list(cores), list(ranks)
    for k in range(order - 1, 0, -1):
        left = cores[k - 1].reshape(ranks[k - 1] * in_shape[k - 1] * out_shape[k - 1], -1)
        right = cores[k].reshape(-1, in_shape[k] * out_shape[k] * ranks[k + 1])
        q, s = torch.linalg.qr(right.t())
        w = left @ s.t()
        if new_ranks:
            q, w = q[:, : new_ranks[k]], w[: new_ranks[k], :]
            ranks[k] = new_ranks[k]
        ranks[k] = w.shape[1]
        cores[k - 1] = w.reshape(ranks[k - 1], in_shape[k - 1], out_shape[k - 1], ranks[k])
        cores[k] = q.t().reshape(ranks[k], in_shape[k], out_shape[k], ranks[k + 1])
    return cores, ranks


def tt_round(cores: List[Tensor], ranks: List[int], in_shape, out_shape, new_ranks):
    """tt.py:182-211 (inplace=True body): right-orthogonalize, then a left sweep
    of complete-mode QRs truncated to new_ranks."""
    order = len(cores)
    if isinstance(new_ranks, int):
        new_ranks = [1] + [new_ranks] * (order - 1) + [1]
    elif not new_ranks:
        new_ranks = [1] + [i * o for i, o in zip(in_shape, out_shape)] + [1]
    cores, ranks = tt_orthogonalize_right(cores, ranks, in_shape, out_shape)
    for k in range(order - 1):
        left = cores[k].reshape(ranks[k] * in_shape[k] * out_shape[k], -1)
        right = cores[k + 1].reshape(-1, in_shape[k + 1] * out_shape[k + 1] * ranks[k + 2])
        q, s = torch.linalg.qr(left, mode="complete")
        q, s = q[:, : new_ranks[k + 1]], s[: new_ranks[k + 1], :]
        w = s @ right
        ranks[k], ranks[k + 1] = new_ranks[k], new_ranks[k + 1]
        cores[k] = q.reshape(ranks[k], in_shape[k], out_shape[k], ranks[k + 1])
        cores[k + 1] = w.reshape(ranks[k + 1], in_shape[k + 1], out_shape[k + 1], ranks[k + 2])
    return cores, ranks


def _tt_meta(cores: Sequence[Tensor]):
    """ranks, in_shape, out_shape as TensorTrain.from_cores derives them from the core shapes."""
    ranks = [c.shape[0] for c in cores] + [cores[-1].shape[-1]]
    return ranks, [c.shape[1] for c in cores], [c.shape[2] for c in cores]


def _tt_round(cores, new_ranks):
    ranks, ins, outs = _tt_meta(cores)
    return tt_round([c.clone() for c in cores], ranks, ins, outs, list(new_ranks))[0]


def _tt_sub(a, b):
    """tt.py:424-426: a + (-1) * b."""
    return tt_add(a, tt_scale(b, -1))


def tt_sqrt(cores: Sequence[Tensor], threshold: float = 1e-3, max_iter: int = 4) -> List[Tensor]:
    """TensorTrain.sqrt (tt.py:312-341): coupled Newton iteration on the train scaled by 4^-k, where
    k = floor(log_4(prod(ranks) * max|last core|)); every product is rounded back to the starting ranks.  `add_` is NOT
    in place in the reference (it returns a new train, :343-379), so `C.add_(-3)` leaves C untouched."""
    ranks, _, _ = _tt_meta(cores)
    max_value = math.prod(ranks) * float(cores[-1].abs().max())
    k = math.floor(math.log(max_value) / math.log(4))
    A = tt_scale(cores, 1 / (4 ** k))
    C = tt_add_constant(A, ranks, -1)
    while max_iter > 0 and tt_inner(_tt_sub(A, C), _tt_sub(A, C), mode="full") > threshold:
        B = _tt_round(_tt_sub(A, tt_scale(tt_mul(A, C), 1 / 2)), ranks)
        D = tt_mul(tt_scale(_tt_round(tt_mul(C, C), ranks), 1 / 4), tt_add_constant(C, _tt_meta(C)[0], -3))
        D = _tt_round(D, ranks)
        max_iter -= 1
        A, C = B, D
    return tt_scale(A, 2 ** k)


def tt_sqrtinv(cores: Sequence[Tensor], threshold: Optional[float] = 1e-8, max_iter: int = 4) -> List[Tensor]:
    """TensorTrain.sqrtinv (tt.py:279-310): Newton iteration for x^-1/2 with every product rounded to the maximal ranks
    [1, i_k * o_k, ..., 1]; scaling k from max|core| ** (order // 2) * prod(ranks)."""
    ranks, ins, outs = _tt_meta(cores)
    order = len(cores)
    max_value = float(max(float(c.abs().max()) for c in cores))
    max_value = math.prod(ranks) * (max_value ** (order // 2))
    k = math.floor(math.log(max_value) / math.log(4))
    c, revc = 1 / (4 ** k), 2 ** k
    A = tt_scale(cores, c)
    max_ranks = [1] + [i * o for i, o in zip(ins, outs)] + [1]
    while max_iter > 0:
        inner = tt_mul(cores, _tt_round(tt_mul(A, A), max_ranks))
        B = _tt_round(tt_scale(tt_add_constant(inner, _tt_meta(inner)[0], -3), -1 / 2), max_ranks)
        C = _tt_round(tt_mul(A, B), max_ranks)
        if threshold:
            diff = _tt_sub(C, A)
            if abs(tt_inner(diff, diff, mode="full")) < threshold:
                return tt_scale(C, revc)
        A = C
        max_iter -= 1
    return tt_scale(A, revc)


def tt_reciprocal(cores: Sequence[Tensor]) -> List[Tensor]:
    """TensorTrain.reciprocal (tt.py:480-494): first and last cores copied, every [:, i, o, :] slice of the middle cores
    replaced by its matrix inverse."""
    out = []
    for idx, c in enumerate(cores):
        if idx in (0, len(cores) - 1):
            out.append(c.clone())
        else:
            out.append(torch.linalg.inv(c.permute(1, 2, 0, 3)).permute(2, 0, 1, 3).contiguous())
    return out


# --------------------------------------------------------------------------
# L1' TT optimizers  (tn_gradient/optimizer/ttadam.py, ttsgd.py)
# --------------------------------------------------------------------------


def ttadam_step(p: Tensor, grad: Tensor, state: dict, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                weight_decay: float = 0.0, correct_bias: bool = True, ranks: Optional[Sequence[int]] = None):
    """One TTAdam.step for one parameter (ttadam.py:44-115), functional.

    `state` holds step / exp_avg / exp_avg_sq; when `ranks` is given the moments
    are lists of TT cores between steps: TT -> dense (:71-74, :79-84 with the
    v<0 -> 0 clamp), dense Adam update (:89-103), p += -step_size * m/(sqrt(v)+eps)
    (:108), decoupled decay p += -lr*wd*p AFTER the step (:110-111), dense -> TT
    (:113-115).  Returns (new_p, new_state)."""
    st = dict(state)
    if "step" not in st:
        st["step"] = 0
    if "exp_avg" not in st:
        m = torch.zeros_like(grad)
    elif ranks is not None:
        m = tt_to_matrix(st["exp_avg"], grad.shape).to(grad.dtype)
    else:
        m = st["exp_avg"].clone()
    if "exp_avg_sq" not in st:
        v = torch.zeros_like(grad)
    elif ranks is not None:
        v = tt_to_matrix(st["exp_avg_sq"], grad.shape).to(grad.dtype).clone()
        v[v < 0] = 0
    else:
        v = st["exp_avg_sq"].clone()
    st["step"] += 1
    b1, b2 = betas
    m = m * b1 + grad * (1.0 - b1)
    v = v * b2 + grad * grad * (1.0 - b2)
    denom = v.sqrt() + eps
    step_size = lr
    if correct_bias:
        step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
    p = p + (m / denom) * (-step_size)
    if weight_decay > 0.0:
        p = p + p * (-lr * weight_decay)
    if ranks is not None:
        st["exp_avg"] = tt_from_matrix(m, ranks, padding=True)
        st["exp_avg_sq"] = tt_from_matrix(v, ranks, padding=True)
    else:
        st["exp_avg"], st["exp_avg_sq"] = m, v
    return p, st


def ttsgd_step(p: Tensor, grad: Tensor, state: dict, lr: float, momentum: float = 0.9, dampening: float = 0.0,
               nesterov: bool = False, ranks: Optional[Sequence[int]] = None):
    """One TTSGD.step for one parameter with weight_decay == 0 (ttsgd.py:44-78).

    grad -> TT (:56-57); first call stores the TT as momentum_buffer and uses it
    (:65-66); later calls compute buf = momentum*buf + (1-dampening)*d_p into a
    LOCAL name only (:68-69) -- the stored buffer is never updated; d_p = buf
    (or d_p + momentum*buf under nesterov); TT -> dense (:75-76); p += -lr*d_p."""
    st = dict(state)
    if "step" not in st:
        st["step"] = 0
    if ranks is not None:
        d_p = tt_from_matrix(grad, ranks, padding=True)
    else:
        d_p = grad
    if momentum != 0:
        if "momentum_buffer" not in st:
            buf = st["momentum_buffer"] = [c.clone() for c in d_p] if ranks is not None else d_p.clone()
        else:
            buf = st["momentum_buffer"]
            if ranks is not None:
                buf = tt_add(tt_scale(buf, momentum), tt_scale(d_p, 1 - dampening))
            else:
                buf = momentum * buf + (1 - dampening) * d_p
        if nesterov:
            d_p = tt_add(d_p, tt_scale(buf, momentum)) if ranks is not None else d_p + momentum * buf
        else:
            d_p = buf
    if ranks is not None:
        d_p = tt_to_matrix(d_p, grad.shape)
    return p + (-lr * d_p), st


# --------------------------------------------------------------------------
# L1 TensorTrainLinear forward  (tn_gradient/layer/tensor_linear.py:54-84)
# --------------------------------------------------------------------------


def tt_linear_forward(x: Tensor, cores: Sequence[Tensor], in_features: int, out_features: int,
                      bias: Optional[Tensor] = None) -> Tensor:
    """Pad the last dim to i^order (:57), view [-1, i, ..., i] (:58), contract
    with the cores over bonds and input legs (:60-71), flatten, keep the first
    out_features columns (:74-76), restore leading dims, add bias (:78-79)."""
    order = len(cores)
    i_dim = cores[0].shape[1]
    lead = x.shape[:-1]
    pad = i_dim ** order - in_features
    xp = torch.nn.functional.pad(x, (0, pad)).reshape(-1, *([i_dim] * order))
    w = tt_reconstruct(cores)  # (i1..in, o1..on)
    rows = int(math.prod(w.shape[:order]))
    y = xp.reshape(xp.shape[0], rows) @ w.reshape(rows, -1)
    y = y[:, :out_features].reshape(*lead, out_features)
    if bias is not None:
        y = y + bias
    return y
