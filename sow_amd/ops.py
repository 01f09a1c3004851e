"""Tensor-level wrappers over the C ABI (raw device pointers + the current HIP stream).

PyTorch is used for device memory (the caching allocator owns every buffer), the
stream and the autograd plumbing only; all arithmetic runs in libsow_amd.so.
Every function requires CUDA(=HIP) tensors and raises otherwise.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"sow_amd supports float32 and bfloat16 tensors, got {t.dtype}") from None


def _need_gpu(*ts: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("sow_amd: the SoW hot path runs on MI355X only (got a CPU tensor); "
                               "there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"sow_amd: tensors on different devices ({dev} vs {t.device})")
    return dev


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device: Optional[torch.device] = None) -> int:
    """hipStream_t of the current torch stream.  torch.cuda.current_stream() builds a Stream object (~20 us of host
    time per call, as much as everything else in a layer call); the raw getter is ~1 us."""
    if _raw_stream is not None:
        idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def _launch(dev: torch.device, what: str, fn, *args) -> None:
    """Call a C-ABI launcher with the tensors' device current (the kernels run on the HIP device that is current on the
    calling thread; the stream argument is that device's current torch stream).  One process per GPU never takes the
    slow branch; a process that touches several GPUs gets the right device instead of an invalid-handle error."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx == torch.cuda.current_device():
        _lib.check(fn(*args, _stream(dev)), what)
    else:
        with torch.cuda.device(idx):
            _lib.check(fn(*args, _stream(dev)), what)


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_WS_BYTES: dict = {}
_GEMM_WS_BYTES: dict = {}
_FWD_WS_BYTES: dict = {}


def _forward_workspace_bytes(lib, T: int, d_in: int, d_out: int, r: int, r_acc: int, kind: int, dt: int) -> int:
    """sow_forward_workspace_bytes (0 for most shapes), memoised per shape."""
    key = (T, d_in, d_out, r, r_acc, kind, dt)
    n = _FWD_WS_BYTES.get(key)
    if n is None:
        n = _FWD_WS_BYTES[key] = int(lib.sow_forward_workspace_bytes(T, d_in, d_out, r, r_acc, kind, dt))
    return n


def _workspace_bytes(lib, T: int, d_in: int, d_out: int, r: int, r_acc: int, kind: int, dt: int) -> int:
    """sow_workspace_bytes, memoised per shape (a ctypes round trip per layer call otherwise)."""
    key = (T, d_in, d_out, r, r_acc, kind, dt)
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = int(lib.sow_workspace_bytes(T, d_in, d_out, r, r_acc, kind, dt))
    return n


def acc_kind(acc_down: Optional[torch.Tensor], acc_up: Optional[torch.Tensor]) -> int:
    """Accumulator kind as SoWLinear.forward decides it (reference sow.py:109-112)."""
    if acc_down is None or acc_down.numel() == 0:
        return _lib.ACC_NONE
    if acc_up is None or acc_up.numel() == 0:
        return _lib.ACC_DENSE
    return _lib.ACC_LOWRANK


def check_accumulator(x2: torch.Tensor, d_in: int, d_out: int, acc_down, acc_up, who: str = "sow_amd") -> Tuple[int, int]:
    """(acc_kind, r_acc) after the checks every entry point applies before raw pointers reach the kernels: the
    accumulator has the input's dtype (TypeError), the shapes of its kind (ValueError), lives on the input's device and is
    dense row-major.  A mismatch that got through would be read as the wrong type -- silent garbage for an fp32
    accumulator under bf16 inputs, an out-of-bounds device read the other way round."""
    kind = acc_kind(acc_down, acc_up)
    if kind == _lib.ACC_NONE:
        return kind, 0
    r_acc = 0
    if acc_down.dtype != x2.dtype or (kind == _lib.ACC_LOWRANK and acc_up.dtype != x2.dtype):
        bad = acc_down.dtype if acc_down.dtype != x2.dtype else acc_up.dtype
        raise TypeError(f"{who}: dtype mismatch, x is {x2.dtype} but the accumulator is {bad}")
    if kind == _lib.ACC_DENSE and tuple(acc_down.shape) != (d_in, d_out):
        raise ValueError(f"{who}: dense accumulator must be [in_features, out_features]")
    if kind == _lib.ACC_LOWRANK:
        r_acc = acc_down.shape[1] if acc_down.dim() == 2 else -1
        if acc_down.dim() != 2 or acc_down.shape[0] != d_in or tuple(acc_up.shape) != (r_acc, d_out):
            raise ValueError(f"{who}: low-rank accumulator shapes do not match")
    for t in (acc_down, acc_up if kind == _lib.ACC_LOWRANK else None):
        if t is not None and (not t.is_cuda or t.device != x2.device):
            raise RuntimeError(f"{who}: the accumulator is on {t.device}, the input on {x2.device}")
    return kind, r_acc


def sow_forward(x2: torch.Tensor, A: torch.Tensor, B: torch.Tensor, acc_down, acc_up, bias, scale: float,
                save_h: bool = True):
    """y, h_save = forward of the SoW contraction on a flattened [T, d_in] input.  save_h = False (no-grad / eval callers,
    e.g. the reload + generate loop of commonsense_evaluate.py:268-287): the projection is not written to HBM and None
    is returned in its place (r <= 64; wider ranks compose two GEMMs and need the buffer as their intermediate)."""
    lib = _lib.load()
    dev = _need_gpu(x2, A, B, acc_down if acc_down is not None and acc_down.numel() else None,
                    acc_up if acc_up is not None and acc_up.numel() else None, bias)
    dt = _dt(x2)
    for name, t in (("A", A), ("B", B), ("bias", bias)):
        if t is not None and t.dtype != x2.dtype:
            raise TypeError(f"sow_amd: dtype mismatch, x is {x2.dtype} but {name} is {t.dtype}")
    T, d_in = x2.shape
    r, d_out = B.shape
    if A.shape != (d_in, r):
        raise ValueError(f"sow_amd: A has shape {tuple(A.shape)}, expected {(d_in, r)}")
    kind, r_acc = check_accumulator(x2, d_in, d_out, acc_down, acc_up)
    x2 = x2.contiguous()
    A, B = A.contiguous(), B.contiguous()
    acc_down = acc_down.contiguous() if kind != _lib.ACC_NONE else None
    acc_up = acc_up.contiguous() if kind == _lib.ACC_LOWRANK else None
    bias = bias.contiguous() if bias is not None else None
    y = torch.empty((T, d_out), dtype=x2.dtype, device=dev)
    h = None
    if save_h or r > 64:
        h = torch.empty(T * (64 if r <= 64 else r), dtype=x2.dtype, device=dev)   # == sow_h_save_elems(T, r)
    # the forward touches a workspace only for some shapes (include/sow_amd.h: sow_forward_workspace_bytes)
    nws = _forward_workspace_bytes(lib, T, d_in, d_out, r, r_acc, kind, dt)
    ws = _ws(nws, dev) if nws else None
    _launch(dev, "sow_forward", lib.sow_forward, _ptr(x2), _ptr(A), _ptr(B), _ptr(acc_down), _ptr(acc_up), _ptr(bias), _ptr(y),
            _ptr(h), T, d_in, d_out, r, r_acc, kind, float(scale), dt, _ptr(ws), 0 if ws is None else ws.numel())
    return y, (h if save_h else None)


def workspace_bytes(T: int, d_in: int, d_out: int, r: int, r_acc: int, kind: int, dtype: torch.dtype) -> int:
    return _workspace_bytes(_lib.load(), T, d_in, d_out, r, r_acc, kind, _DT[dtype])


def sow_backward(dy2: torch.Tensor, x2: torch.Tensor, h: torch.Tensor, A: torch.Tensor, B: torch.Tensor, acc_down,
                 acc_up, scale: float, need_bias: bool,
                 out: Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]] = None,
                 grad_beta: float = 0.0, *, phases: int = _lib.BWD_DATA | _lib.BWD_WEIGHTS,
                 dx: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None):
    """dx, dA, dB, dbias.  `out` = (dA, dB, dbias) buffers to write/accumulate into (grad_beta).
    `phases` selects the data-gradient and / or weight-gradient kernels (see include/sow_amd.h); a split
    call must pass the same `workspace` (and `dx`) to both phases, each enqueued on the current stream."""
    lib = _lib.load()
    dev = _need_gpu(dy2, x2, h, A, B)
    dt = _dt(x2)
    if dy2.dtype != x2.dtype:
        raise TypeError(f"sow_amd: grad dtype {dy2.dtype} differs from input dtype {x2.dtype}")
    T, d_in = x2.shape
    r, d_out = B.shape
    kind = acc_kind(acc_down, acc_up)
    r_acc = acc_down.shape[1] if kind == _lib.ACC_LOWRANK else 0
    dy2 = dy2.contiguous()
    if dx is None:
        dx = torch.empty((T, d_in), dtype=x2.dtype, device=dev)
    if out is None:
        dA = torch.empty((d_in, r), dtype=x2.dtype, device=dev)
        dB = torch.empty((r, d_out), dtype=x2.dtype, device=dev)
        dbias = torch.empty((d_out,), dtype=x2.dtype, device=dev) if need_bias else None
        grad_beta = 0.0
    else:
        dA, dB, dbias = out
    nws = _workspace_bytes(lib, T, d_in, d_out, r, r_acc, kind, dt)
    ws = _ws(nws, dev) if workspace is None else workspace
    if ws.numel() < nws:
        raise ValueError("sow_amd: workspace too small")
    _launch(dev, "sow_backward", lib.sow_backward_ex, _ptr(dy2), _ptr(x2), _ptr(h), _ptr(A), _ptr(B),
            _ptr(acc_down) if kind != _lib.ACC_NONE else None, _ptr(acc_up) if kind == _lib.ACC_LOWRANK else None,
            _ptr(dx), _ptr(dA), _ptr(dB), _ptr(dbias), T, d_in, d_out, r, r_acc, kind, float(scale), float(grad_beta), dt,
            _ptr(ws), ws.numel(), int(phases))
    return dx, dA, dB, dbias


class LayerCall:
    """Arguments of one SoWLinear forward / backward inside a grouped call (include/sow_amd.h: sow_layer_args).  Static
    training buffers: build once, reuse every step.  `out` = (dA, dB, dbias) gradient buffers, `workspace` this layer's
    own workspace (workspace_bytes())."""

    def __init__(self, x2, A, B, *, acc_down=None, acc_up=None, bias=None, scale=1.0, y=None, h=None, dy2=None, dx=None,
                 out=None, grad_beta=0.0, workspace=None, forward_only=False, save_h=True):
        dev = _need_gpu(x2, A, B, bias, y, h, dy2, dx, workspace)
        self.dtype = _dt(x2)
        T, d_in = x2.shape
        r, d_out = B.shape
        # the same accumulator checks as the single-layer entry point (ops.sow_forward): same exceptions for the same input
        kind, _ = check_accumulator(x2, d_in, d_out, acc_down, acc_up, "sow_amd.LayerCall")
        for name, t in (("x", x2), ("A", A), ("B", B), ("bias", bias), ("y", y), ("dy", dy2), ("dx", dx),
                        ("acc_down", acc_down if kind != _lib.ACC_NONE else None),
                        ("acc_up", acc_up if kind == _lib.ACC_LOWRANK else None)):
            if t is not None and (not t.is_contiguous() or t.dtype != x2.dtype):
                raise ValueError(f"sow_amd.LayerCall: {name} must be contiguous and of the input dtype")
        if A.shape != (d_in, r):
            raise ValueError("sow_amd.LayerCall: factor shapes do not match the input")
        self.device, self.kind = dev, kind
        self.y = y if y is not None else torch.empty((T, d_out), dtype=x2.dtype, device=dev)
        # save_h = False (forward_only calls that no backward follows): h_save = NULL, the projection stays on chip
        self.h = h if h is not None else (torch.empty(T * (64 if r <= 64 else r), dtype=x2.dtype, device=dev)
                                          if (save_h or r > 64 or not forward_only) else None)
        self.dx = dx
        r_acc = acc_down.shape[1] if kind == _lib.ACC_LOWRANK else 0
        # a forward-only call needs scratch for a few shapes only (sow_forward_workspace_bytes), often none at all
        nws = (_forward_workspace_bytes(_lib.load(), T, d_in, d_out, r, r_acc, kind, self.dtype) if forward_only
               else workspace_bytes(T, d_in, d_out, r, r_acc, kind, x2.dtype))
        self.workspace = workspace if workspace is not None else (_ws(nws, dev) if nws else None)
        if self.workspace is not None and self.workspace.numel() < nws:
            raise ValueError("sow_amd.LayerCall: workspace too small")
        dA, dB, dbias = out if out is not None else (None, None, None)
        self._keep = (x2, A, B, acc_down, acc_up, bias, dy2, dA, dB, dbias)     # the struct holds raw pointers
        self.args = _lib.LayerArgs(
            x=_ptr(x2), A=_ptr(A), B=_ptr(B), acc_down=_ptr(acc_down) if kind != _lib.ACC_NONE else None,
            acc_up=_ptr(acc_up) if kind == _lib.ACC_LOWRANK else None, bias=_ptr(bias), y=_ptr(self.y), h_save=_ptr(self.h),
            dy=_ptr(dy2), dx=_ptr(dx), dA=_ptr(dA), dB=_ptr(dB), dbias=_ptr(dbias), T=T, d_in=d_in, d_out=d_out, r_live=r,
            r_acc=acc_down.shape[1] if kind == _lib.ACC_LOWRANK else 0, acc_kind=kind, scale=float(scale),
            grad_beta=float(grad_beta), workspace=_ptr(self.workspace),
            workspace_bytes=0 if self.workspace is None else self.workspace.numel())


class LayerGroup:
    """n independent layer calls issued through sow_forward_group / sow_backward_group: layers on the bf16 streaming
    kernels share launches (q / k / v; gate / up).  Outputs, input gradients and saved projections are bit-identical to n
    single calls; the weight gradients of a group large enough for the row-owner kernel (a whole decoder block) are summed
    over differently cut token slabs and agree to fp32 rounding of those sums."""

    def __init__(self, calls: Sequence[LayerCall]):
        if not calls:
            raise ValueError("empty group")
        if len({c.dtype for c in calls}) != 1 or len({c.device for c in calls}) != 1:
            raise ValueError("sow_amd.LayerGroup: all layers must share dtype and device")
        self.calls = list(calls)
        self.arr = (_lib.LayerArgs * len(calls))(*[c.args for c in calls])
        self.dtype, self.device = calls[0].dtype, calls[0].device

    @classmethod
    def from_args(cls, arr, n: int, dtype: int, device, keep=None) -> "LayerGroup":
        """A group over a ready-made sow_layer_args array (the caller has validated the tensors and keeps them alive --
        `keep` -- until the launches that read the raw pointers have been enqueued)."""
        g = cls.__new__(cls)
        g.calls, g.arr, g.dtype, g.device, g._keep = [None] * n, arr, dtype, device, keep
        return g

    def forward(self) -> None:
        _launch(self.device, "sow_forward_group", _lib.load().sow_forward_group, self.arr, len(self.calls), self.dtype)

    def backward(self, phases: int = _lib.BWD_DATA | _lib.BWD_WEIGHTS) -> None:
        _launch(self.device, "sow_backward_group", _lib.load().sow_backward_group, self.arr, len(self.calls), self.dtype,
                int(phases))

    def weight_gradient_plan(self, phases: int = _lib.BWD_DATA | _lib.BWD_WEIGHTS):
        """(row_owner_kernel: bool, [(slabs of x, slabs of dY) per layer]) for backward(phases) -- sow_backward_group_plan."""
        n = len(self.calls)
        slabs = (ctypes.c_int * (2 * n))()
        rc = _lib.load().sow_backward_group_plan(self.arr, n, self.dtype, int(phases), slabs)
        if rc < 0:
            _lib.check(rc, "sow_backward_group_plan")
        return bool(rc), [(slabs[2 * i], slabs[2 * i + 1]) for i in range(n)]

    def reduce_descs(self, phases: int):
        """Descriptors and block counts of the deferred weight-gradient reductions of this group (sow_reduce_batch), for a
        PARTIAL phase issued with the same `phases` flags (BWD_GROUP_SLABS included or not)."""
        lib = _lib.load()
        n, size = len(self.calls), lib.sow_reduce_desc_bytes()
        buf = ctypes.create_string_buffer(size * n)
        blocks = (ctypes.c_int * n)()
        _lib.check(lib.sow_backward_group_reduce_desc(self.arr, n, self.dtype, int(phases), buf, blocks),
                   "sow_backward_group_reduce_desc")
        return [buf.raw[i * size:(i + 1) * size] for i in range(n)], list(blocks)


class DeferredReduce:
    """The weight-gradient reductions of many layers in one launch (include/sow_amd.h: sow_reduce_batch).

    Usage per step: `sow_backward(..., phases=BWD_DATA | BWD_WEIGHTS_PARTIAL, out=..., workspace=ws_i)` for every layer
    (each layer its own workspace), then `run()` once before the gradients are consumed.  The descriptors are built on
    the first step from `add()` calls and reused while the same buffers are passed again (static training buffers, HIP
    graphs); `add()` with other pointers rebuilds them."""

    def __init__(self):
        self._keys, self._descs, self._blocks, self._dev, self._dt = [], [], [], None, None
        self._d_descs = self._d_starts = None
        self._total = 0
        self._pos = 0

    def add(self, x2, B, out, grad_beta, workspace, acc_down=None, acc_up=None):
        """Register (or re-validate) the layer whose PARTIAL phase was just enqueued."""
        lib = _lib.load()
        dA, dB, dbias = out
        T, d_in = x2.shape
        r, d_out = B.shape
        kind = acc_kind(acc_down, acc_up)
        r_acc = acc_down.shape[1] if kind == _lib.ACC_LOWRANK else 0
        key = (_ptr(dA), _ptr(dB), _ptr(dbias), T, d_in, d_out, r, r_acc, kind, float(grad_beta), _dt(x2), _ptr(workspace))
        i = self._pos
        self._pos += 1
        if i < len(self._keys) and self._keys[i] == key:
            return
        # new or changed layer: (re)build from here on
        del self._keys[i:], self._descs[i:], self._blocks[i:]
        self._d_descs = None
        buf = ctypes.create_string_buffer(lib.sow_reduce_desc_bytes())
        nb = ctypes.c_int(0)
        _lib.check(lib.sow_backward_reduce_desc(_ptr(dA), _ptr(dB), _ptr(dbias), T, d_in, d_out, r, r_acc, kind, float(grad_beta),
                                                _dt(x2), _ptr(workspace), workspace.numel(), buf, ctypes.byref(nb)),
                   "sow_backward_reduce_desc")
        self._keys.append(key)
        self._descs.append(buf.raw)
        self._blocks.append(nb.value)
        self._dev, self._dt = x2.device, _dt(x2)

    def add_group(self, group: "LayerGroup", phases: int, stable_key=None):
        """Register (or re-validate) every layer of a group whose PARTIAL phase was just enqueued with `phases`.
        `stable_key`: what the reduction descriptors of the group depend on (gradient buffers, workspaces, shapes) when the
        group object and its activation pointers change from step to step -- the cached descriptors are then reused."""
        n = len(group.calls)
        key = (("group", int(phases), stable_key) if stable_key is not None
               else ("group", id(group), int(phases), bytes(group.arr)))
        i = self._pos
        self._pos += n
        if i + n <= len(self._keys) and all(self._keys[i + k] == (key, k) for k in range(n)):
            return
        del self._keys[i:], self._descs[i:], self._blocks[i:]
        self._d_descs = None
        descs, blocks = group.reduce_descs(phases)
        for k in range(n):
            self._keys.append((key, k))
            self._descs.append(descs[k])
            self._blocks.append(blocks[k])
        self._dev, self._dt = group.device, group.dtype

    def run(self):
        """One launch for every layer added since the last run()."""
        n = self._pos
        self._pos = 0
        if n == 0:
            return
        if n != len(self._keys):           # fewer layers than last step
            del self._keys[n:], self._descs[n:], self._blocks[n:]
            self._d_descs = None
        if self._d_descs is None:
            raw = b"".join(self._descs)
            self._d_descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self._dev)
            starts, tot = [], 0
            for b in self._blocks:
                starts.append(tot)
                tot += b
            self._d_starts = torch.tensor(starts, dtype=torch.int32, device=self._dev)
            self._total = tot
        lib = _lib.load()
        _launch(self._dev, "sow_reduce_batch", lib.sow_reduce_batch, _ptr(self._d_descs), _ptr(self._d_starts), n, self._total,
                self._dt)


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
         out: Optional[torch.Tensor] = None, alpha: float = 1.0, beta: float = 0.0,
         bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = alpha * op(A) @ op(B) + beta * out (+ bias).  2-D row-major tensors with unit inner stride."""
    lib = _lib.load()
    dev = _need_gpu(A, B, out, bias)
    dt = _dt(A)
    if B.dtype != A.dtype:
        raise TypeError("sow_amd.gemm: operand dtypes differ")
    if A.dim() != 2 or B.dim() != 2:
        raise ValueError("sow_amd.gemm expects 2-D tensors")
    if A.stride(1) != 1:
        A = A.contiguous()
    if B.stride(1) != 1:
        B = B.contiguous()
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    Kb, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    if K != Kb:
        raise ValueError(f"sow_amd.gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=dev)
        beta = 0.0
    elif tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.dtype != A.dtype:
        raise ValueError("sow_amd.gemm: bad output tensor")
    if M == 0 or N == 0:
        return out
    if K == 0:
        if beta == 0.0:
            out.zero_()
        return out
    # short bf16 products split K over workgroups and need scratch for the partial sums (0 for every other shape)
    key = (M, N, K, bool(trans_a), dt)
    nws = _GEMM_WS_BYTES.get(key)
    if nws is None:
        nws = _GEMM_WS_BYTES[key] = int(lib.sow_gemm_workspace_bytes(M, N, K, int(trans_a), dt))
    ws = _ws(nws, dev) if nws else None
    _launch(dev, "sow_gemm_ex", lib.sow_gemm_ex, _ptr(A), max(A.stride(0), 1), int(trans_a), _ptr(B), max(B.stride(0), 1),
            int(trans_b), _ptr(out), max(out.stride(0), 1), _ptr(bias), M, N, K, float(alpha), float(beta), dt, _ptr(ws),
            0 if ws is None else ws.numel())
    return out


def qr_thin(W: torch.Tensor, k: int, need_r: bool = True, out_dtype: Optional[torch.dtype] = None):
    """Q[:, :k], R[:k, :] of the Householder QR of W (LAPACK sign convention), fp32 internals."""
    lib = _lib.load()
    dev = _need_gpu(W)
    if W.dim() != 2:
        raise ValueError("qr_thin expects a matrix")
    if W.stride(1) != 1:
        W = W.contiguous()
    m, n = W.shape
    out_dtype = out_dtype or W.dtype
    if k < 1 or k > m:
        raise ValueError(f"qr_thin: k={k} out of range for {m} rows")
    Q = torch.empty((m, k), dtype=out_dtype, device=dev)
    R = torch.empty((k, n), dtype=out_dtype, device=dev) if need_r else None
    nws = lib.sow_qr_workspace_bytes(m, n, k, _dt(W), int(need_r))
    ws = _ws(nws, dev)
    _launch(dev, "sow_qr_thin", lib.sow_qr_thin, _ptr(W), W.stride(0), m, n, _dt(W), k, _ptr(Q), k, _ptr(R), n, _DT[out_dtype],
            _ptr(ws), ws.numel())
    return Q, R


def zero_(tensors: Sequence[torch.Tensor]) -> None:
    """Zero a list of dense device tensors with one (or a few) multi-tensor launches."""
    lib = _lib.load()
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    if not ts:
        return
    dev = _need_gpu(*ts)
    for t in ts:
        if not t.is_contiguous():
            raise ValueError("sow_amd.zero_: tensors must be contiguous")
    n = len(ts)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in ts])
    _launch(dev, "sow_zero_state", lib.sow_zero_state, ptrs, sizes, n)


def adamw_flat_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float,
                betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01, step: int = 1,
                grad_scale: float = 1.0) -> None:
    lib = _lib.load()
    dev = _need_gpu(param, grad, exp_avg, exp_avg_sq)
    _launch(dev, "sow_adamw_flat", lib.sow_adamw_flat, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr,
            betas[0], betas[1], eps, weight_decay, int(step), grad_scale, _dt(param), _dt(exp_avg))


def ttadam_dense_(param, grad, exp_avg, exp_avg_sq, *, beta1, beta2, eps, step_size, lr_times_wd, clamp_v: bool) -> None:
    lib = _lib.load()
    dev = _need_gpu(param, grad, exp_avg, exp_avg_sq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("ttadam_dense_ expects contiguous float32 tensors")
    _launch(dev, "sow_ttadam_dense", lib.sow_ttadam_dense, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
            beta1, beta2, eps, step_size, lr_times_wd, int(clamp_v))


def tt_kron_core(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """einsum('aijb,cijd->acijbd') reshaped to [ra*rc, i, j, rb*rd] (reference tt.py:469-475)."""
    lib = _lib.load()
    dev = _need_gpu(a, b)
    a, b = a.contiguous().float(), b.contiguous().float()
    ra0, i, j, ra1 = a.shape
    rb0, i2, j2, rb1 = b.shape
    if (i, j) != (i2, j2):
        raise ValueError("tt_kron_core: physical dimensions differ")
    out = torch.empty((ra0 * rb0, i, j, ra1 * rb1), dtype=torch.float32, device=dev)
    _launch(dev, "sow_tt_kron_core", lib.sow_tt_kron_core, _ptr(a), _ptr(b), _ptr(out), ra0, rb0, i * j, ra1, rb1)
    return out


def absmax(x: torch.Tensor) -> float:
    """max |x| of an fp32 device tensor (one small kernel + a host read)."""
    lib = _lib.load()
    dev = _need_gpu(x)
    x = x.contiguous().float()
    out = torch.empty(1, dtype=torch.float32, device=dev)
    _launch(dev, "sow_absmax", lib.sow_absmax, _ptr(x), x.numel(), _ptr(out))
    return float(out.item())


def small_inverse(mats: torch.Tensor) -> torch.Tensor:
    """Inverse of a batch of [r, r] fp32 matrices (r <= 16)."""
    lib = _lib.load()
    dev = _need_gpu(mats)
    m = mats.contiguous().float()
    b, r, r2 = m.shape
    if r != r2:
        raise ValueError("small_inverse expects square matrices")
    out = torch.empty_like(m)
    _launch(dev, "sow_small_inverse", lib.sow_small_inverse, _ptr(m), _ptr(out), b, r)
    return out


def axpby_(x: torch.Tensor, y: torch.Tensor, a: float, b: float) -> torch.Tensor:
    """y <- a*x + b*y"""
    lib = _lib.load()
    dev = _need_gpu(x, y)
    if x.dtype != y.dtype or x.numel() != y.numel() or not (x.is_contiguous() and y.is_contiguous()):
        raise ValueError("axpby_: x and y must be contiguous, same dtype and size")
    _launch(dev, "sow_axpby", lib.sow_axpby, _ptr(x), _ptr(y), x.numel(), float(a), float(b), _dt(x))
    return y


class _MatMul(torch.autograd.Function):
    """2-D a @ b on the MFMA GEMM kernel with autograd (used by the TT layer's core contractions)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm(a, b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous()
        da = gemm(dc, b, trans_b=True) if ctx.needs_input_grad[0] else None   # dC @ B^T
        db = gemm(a, dc, trans_a=True) if ctx.needs_input_grad[1] else None   # A^T @ dC
        return da, db


def matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _MatMul.apply(a, b)
