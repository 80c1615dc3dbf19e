"""MI355X-native functional value estimators.

Drop-in for ``torchrl.objectives.value.functional`` (reference:
torchrl/objectives/value/functional.py): same names, argument order, keyword-only
``time_dim`` where the reference has it, same ``done`` / ``terminated`` semantics, same
``[*B, T, *F]`` conventions and the same exceptions.  Everything below the Python signature
is replaced by one call into the native library (include/trl_value.h): the gfx950 kernels for tensors on the GPU,
the library's own host entry points (``trl_*_cpu``: the SEQ family's row functions run by a few threads) for tensors
in host memory -- the reference, too, runs on whatever device its tensors live on (functional.py:163-170).  There is
no PyTorch fallback and the test oracle is never involved: a missing library raises.

Host-side work done here (and only here):
  * shape / time-dim validation with the reference's error types and messages
    (functional.py:44-46, 49-113);
  * canonicalisation to the contiguous ``(O, T, I)`` view the kernels use -- directly from
    the original layout, so a contiguous ``[*B, T, *F]`` input is never copied or transposed
    whatever ``time_dim`` is (the reference's ``_transpose_time`` would hand non-contiguous
    views to ATen);
  * reproducing the dtype / rounding of the reference's coefficient tensors
    (``gamma * int_mask`` rounds python floats to the default dtype even for float64 values,
    functional.py:172-174, 1322-1323; ``torch.full_like(next_state_value, gamma)`` keeps the
    value dtype, functional.py:850-857) so that the kernels receive already-rounded scalars.

The ``vec_*`` variants call the same kernels: the scan is exact, so the reference's truncated
geometric filter (functional.py:185-210) and its O(T^2) gamma tables (value/utils.py:131-182)
have no counterpart here.
"""
from __future__ import annotations

import contextlib
import ctypes
import struct
import math
import warnings
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch

from torchrl_amd import _extension

__all__ = [
    "generalized_advantage_estimate",
    "vec_generalized_advantage_estimate",
    "td0_advantage_estimate",
    "td0_return_estimate",
    "td1_return_estimate",
    "vec_td1_return_estimate",
    "td1_advantage_estimate",
    "vec_td1_advantage_estimate",
    "td_lambda_return_estimate",
    "vec_td_lambda_return_estimate",
    "td_lambda_advantage_estimate",
    "vec_td_lambda_advantage_estimate",
    "vtrace_advantage_estimate",
    "reward2go",
]

SHAPE_ERR = "All input tensors (value, reward and done states) must share a unique shape."
_TIME_ERR = "The tensor shape and the time dimension are not compatible: got {} and time_dim={}."

ALGO_AUTO, ALGO_SEQ, ALGO_SCAN, ALGO_FUSED, ALGO_LOCKSTEP = 0, 1, 2, 3, 4
_DT = {torch.float32: 0, torch.float64: 1}


# ---------------------------------------------------------------------------------------------
# time-dim handling
# ---------------------------------------------------------------------------------------------
def _canonical_dims(shape, time_dim: int) -> Tuple[int, int, int, int]:
    """``(O, T, I, td)`` for a ``[*B, T, *F]`` tensor with time at ``time_dim``; ``td`` is the
    non-negative index of the time dim.  Raises the reference's time-dim ``RuntimeError``."""
    nd = len(shape)
    td = time_dim - nd if time_dim >= 0 else time_dim
    if td < -nd or td >= 0:
        raise RuntimeError(_TIME_ERR.format(torch.Size(shape), td))
    if nd == 1:
        if td != -1:
            raise RuntimeError(_TIME_ERR.format(torch.Size(shape), td))
        return 1, int(shape[0]), 1, 0
    td += nd
    return math.prod(shape[:td]), int(shape[td]), math.prod(shape[td + 1 :]), td


def _expand_coef(x: torch.Tensor, shape, time_dim: int) -> torch.Tensor:
    """Broadcast a coefficient tensor (numel > 1) to ``shape`` the way the reference does:
    the decorator transposes it like every other tensor, then it broadcasts against the
    transposed operands (functional.py:64-91, 172-174)."""
    nd = len(shape)
    if tuple(x.shape) == tuple(shape):
        return x
    td = time_dim - nd if time_dim >= 0 else time_dim
    if td == -2 or nd < 2:
        return x.expand(shape)
    xtd = time_dim - x.ndim if time_dim >= 0 else time_dim
    if xtd < -x.ndim or xtd >= 0 or x.ndim < 2:
        raise RuntimeError(_TIME_ERR.format(x.shape, xtd))
    tshape = list(shape)
    tshape[td], tshape[-2] = tshape[-2], tshape[td]
    return x.transpose(xtd, -2).expand(tshape).transpose(td, -2)


# ---------------------------------------------------------------------------------------------
# coefficient rounding
# ---------------------------------------------------------------------------------------------
_F32 = struct.Struct("f")


def _round_to(value: float, dtype: torch.dtype) -> float:
    if dtype == torch.float64:
        return float(value)
    if dtype == torch.float32:
        try:  # round-to-nearest-even double -> float, as torch.tensor(value, dtype=float32) does
            return _F32.unpack(_F32.pack(value))[0]
        except OverflowError:
            pass  # beyond FLT_MAX: torch gives +-inf
    return float(torch.tensor(value, dtype=dtype))


def _is_full(x) -> bool:
    return isinstance(x, torch.Tensor) and x.numel() > 1


def _mask_coef_dtype(x) -> torch.dtype:
    """dtype of ``x * int32_mask`` for scalar-like or tensor x."""
    if isinstance(x, torch.Tensor) and x.dtype.is_floating_point:
        return x.dtype
    return torch.get_default_dtype()


def _scalar_mask_coef(x) -> Tuple[float, torch.dtype]:
    """value and dtype of ``x * int32_mask`` where the mask is 1, for scalar-like x
    (python number, 0-dim or 1-element tensor)."""
    dt = _mask_coef_dtype(x)
    if isinstance(x, torch.Tensor):
        val = float(x.detach())  # host read; device scalars synchronise (modules cache host copies)
        return (val if x.dtype.is_floating_point else _round_to(val, dt)), dt
    return _round_to(float(x), dt), dt


# ---------------------------------------------------------------------------------------------
# tensor plumbing
# ---------------------------------------------------------------------------------------------
def _check_value_dtype(*tensors) -> torch.dtype:
    dt = tensors[0].dtype
    for t in tensors[1:]:
        dt = torch.promote_types(dt, t.dtype)
    if dt not in _DT:
        raise NotImplementedError(
            f"torchrl_amd value estimators support float32 and float64 tensors, got {dt}"
        )
    return dt


def _require_hip(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError(
            "this torchrl_amd entry point (prepared steps, cross-GPU exchange, workspaces) runs on MI355X (HIP, device type "
            f"'cuda') tensors only; got a tensor on {t.device}."
        )


def _on_host(t: torch.Tensor) -> bool:
    """True for tensors in host memory (the library's ``trl_*_cpu`` entry points take them); HIP tensors give False, any
    other device raises."""
    kind = t.device.type
    if kind == "cpu":
        return True
    if kind != "cuda":
        raise RuntimeError(f"torchrl_amd value estimators run on HIP (device type 'cuda') and CPU tensors; got a tensor on {t.device}")
    return False


def _val(t: torch.Tensor, dtype, device=None) -> torch.Tensor:
    if t.requires_grad:
        t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)
    if device is not None and t.device != device:
        t = t.to(device)
    return t.contiguous()


def _shifted_pair(state_value: torch.Tensor, next_state_value: torch.Tensor, shape, time_dim: int, vdt, grads: bool) -> bool:
    """True when ``state_value`` / ``next_state_value`` are the ``[0, T)`` / ``[1, T + 1)`` time windows of ONE
    contiguous ``[*B, T + 1, *F]`` buffer with ``prod(F) == 1`` -- what the estimators' ``shifted=True`` single call
    of the value network yields (reference: advantages.py:440-490).  The kernels then read that buffer in place
    (include/trl_value.h: ``next_state_value == NULL``), the value bytes once, instead of two copies of it."""
    sv, nsv = state_value, next_state_value
    nd = len(shape)
    td = time_dim if time_dim >= 0 else nd + time_dim
    # cheapest test first (this runs on every call): the second window starts one time step behind the first
    if grads or nsv.data_ptr() - sv.data_ptr() != sv.element_size() * sv.stride(td):
        return False
    if sv.dtype != vdt or nsv.dtype != vdt or sv.device != nsv.device:
        return False
    if math.prod(shape[td + 1:]) != 1 or math.prod(shape) == 0 or sv.stride() != nsv.stride():
        return False
    if sv.untyped_storage().data_ptr() != nsv.untyped_storage().data_ptr() or nsv.storage_offset() != sv.storage_offset() + sv.stride(td):
        return False
    want = 1
    for d in range(td, -1, -1):  # contiguous strides of the [*B, T + 1] buffer (extent-1 dims are free)
        if shape[d] != 1 and sv.stride(d) != want:
            return False
        want *= shape[d] + 1 if d == td else shape[d]
    return True


def _mask(m: torch.Tensor, shape, device=None) -> torch.Tensor:
    if m.dtype != torch.bool:
        m = m != 0
    if tuple(m.shape) != tuple(shape):
        m = m.expand(shape)
    if device is not None and m.device != device:
        m = m.to(device)
    return m.contiguous().view(torch.uint8)  # materialises stride-0 expanded masks (DoneTransform)


def _masks(done: torch.Tensor, terminated: torch.Tensor, shape, time_dim: int, device, grads: bool):
    """``(done_u8, terminated_u8, mask_bcast)`` for the native call.  Masks that are broadcast over every dim
    after the time dim (stride 0 or extent 1: a per-env mask ``expand``ed over the agent dim, as
    ``DoneTransform`` produces) are handed over as compact ``[*B, T]`` arrays and read in place by the kernels
    (``mask_bcast = 1``); anything else is materialised in the canonical layout."""
    nd = len(shape)
    td = time_dim - nd if time_dim >= 0 else time_dim
    n_feat = -td - 1
    if n_feat > 0 and not grads and done.dtype == torch.bool and terminated.dtype == torch.bool:
        feat = range(nd - n_feat, nd)
        if math.prod(shape[d] for d in feat) > 1 and all(
                tuple(m.shape) == tuple(shape) and all(m.stride(d) == 0 or shape[d] == 1 for d in feat)
                for m in (done, terminated)):
            index = (Ellipsis,) + (0,) * n_feat
            compact = tuple(shape[:nd - n_feat])
            return _mask(done[index], compact, device), _mask(terminated[index], compact, device), 1
    return _mask(done, shape, device), _mask(terminated, shape, device), 0


def _ptr(t: Optional[torch.Tensor]):
    # a plain int (or None) is what a c_void_p parameter converts fastest
    return t.data_ptr() if t is not None else None


def _stream(device) -> int:
    """The caller's current hipStream_t on ``device`` as an integer handle."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


_NO_GUARD = contextlib.nullcontext()


def _on(device):
    """Device guard for the native call -- free when ``device`` is already current (the common case)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def _needs_grad(*xs) -> bool:
    return torch.is_grad_enabled() and any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs)


_GRAD_MSG = (
    "torchrl_amd: differentiating through {} is not implemented yet (GAE, TD(lambda) and V-trace are); "
    "call it under torch.no_grad() / with differentiable=False, or detach the inputs."
)


def last_kernel() -> str:
    """Name of the kernel family the last call on this thread dispatched to (for tests / profiling)."""
    return _extension.get_lib().trl_last_kernel().decode()


# ---------------------------------------------------------------------------------------------
# GAE
# ---------------------------------------------------------------------------------------------
@dataclass
class _GaeCall:
    shape: tuple
    O: int
    T: int
    I: int
    vdt: torch.dtype
    cdt: torch.dtype
    out_dtype: torch.dtype
    reward: torch.Tensor
    value: torch.Tensor
    next_value: Optional[torch.Tensor]  # None: `value` is a window of the shifted (O, T + 1) buffer (_shifted_pair)
    done: torch.Tensor
    terminated: torch.Tensor
    gamma_t: Optional[torch.Tensor]
    lg_t: Optional[torch.Tensor]
    gamma: float
    lmbda: float
    lg: float
    mask_bcast: int = 0


def _prepare_gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, time_dim) -> _GaeCall:
    """Host logic of the GAE entry points: validation, canonical view, coefficient rounding."""
    if terminated is None:
        terminated = done  # reference clones; the kernels never write their inputs
    if not (next_state_value.shape == state_value.shape == reward.shape == done.shape == terminated.shape):
        raise RuntimeError(SHAPE_ERR)
    shape = tuple(next_state_value.shape)
    O, T, I, _ = _canonical_dims(shape, time_dim)
    vdt = _check_value_dtype(reward, state_value, next_state_value)
    dev = next_state_value.device

    gamma_t = lg_t = None
    g_val = l_val = lg_val = 0.0
    if _is_full(gamma) or _is_full(lmbda):
        if _is_full(gamma) and not _is_full(lmbda):
            # lg = (coef)lmbda * gamma_t is formed in-kernel (what `lmbda * gamma` gives)
            cdt = _mask_coef_dtype(gamma)
            gamma_t = _expand_coef(gamma.detach(), shape, time_dim).to(device=dev, dtype=cdt).contiguous()
            l_val = float(lmbda)
        else:
            prod = lmbda * (gamma.detach() if isinstance(gamma, torch.Tensor) else gamma)
            g_full = gamma.detach() if _is_full(gamma) else torch.full((), float(gamma), dtype=_mask_coef_dtype(gamma))
            cdt = torch.promote_types(_mask_coef_dtype(g_full), _mask_coef_dtype(prod))
            lg_t = _expand_coef(prod.detach(), shape, time_dim).to(device=dev, dtype=cdt).contiguous()
            gamma_t = (_expand_coef(g_full, shape, time_dim) if g_full.numel() > 1 else g_full.expand(shape)).to(
                device=dev, dtype=cdt).contiguous()
    else:
        g_val, g_dt = _scalar_mask_coef(gamma)
        lg_val, lg_dt = _scalar_mask_coef(lmbda * gamma)
        cdt = torch.promote_types(g_dt, lg_dt)
    if cdt not in _DT:
        raise NotImplementedError(f"gamma / lmbda dtype {cdt} is not supported (float32 / float64 only)")
    # a differentiable call keeps both value tensors (the backward kernels read the dense layout); compact masks are fine
    grads = _needs_grad(state_value, next_state_value, reward, gamma, lmbda)
    d8, t8, bcast = _masks(done, terminated, shape, time_dim, dev, False)
    shifted = _shifted_pair(state_value, next_state_value, shape, time_dim, vdt, grads)
    return _GaeCall(
        shape=shape, O=O, T=T, I=I, vdt=vdt, cdt=cdt, out_dtype=next_state_value.dtype,
        reward=_val(reward, vdt, dev), value=state_value.detach() if shifted else _val(state_value, vdt, dev),
        next_value=None if shifted else _val(next_state_value, vdt, dev),
        done=d8, terminated=t8, gamma_t=gamma_t, lg_t=lg_t, gamma=g_val, lmbda=l_val, lg=lg_val, mask_bcast=bcast,
    )


_WORKSPACES = {}


def _workspace(lib, dev, O, T, I):
    """Per-(device, stream) workspace, zero-filled once (include/trl_value.h: it carries the
    epoch of the fused kernel's in-launch exchange); grown on demand."""
    need = lib.trl_workspace_bytes(O, T, I)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev))
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(max(need, 1 << 16), dtype=torch.uint8, device=dev)
        _WORKSPACES[key] = ws
        if _HOST:   # the host module mirrors this table: it must never keep the address of a replaced workspace
            _HOST.set_workspace(key[0], key[1], ws.data_ptr(), ws.numel())
    return ws


def workspace_error(device=None) -> bool:
    """True if a single-launch GAE + normalise on ``device``'s current stream ever timed out waiting for its
    other blocks (the persistent grid needs the whole GPU; such a call leaves the advantage un-normalised, the next
    estimator call warns once -- RuntimeWarning -- and GAE + normalise takes the two-launch route on that GPU from then
    on: :func:`fused_degraded`, :func:`fused_reset`).  Synchronises."""
    lib = _extension.get_lib()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ws = _WORKSPACES.get((dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev)))
    return ws is not None and lib.trl_workspace_error(_ptr(ws)) == 1


def fused_degraded(device=None) -> bool:
    """True once a single-launch GAE + normalise call on ``device`` (default: the current one) has given up: the library
    then runs the step as two launches (scan + scale pass), which need no co-resident grid (include/trl_value.h:
    trl_fused_degraded).  No synchronisation."""
    lib = _extension.get_lib()
    with (torch.cuda.device(device) if device is not None else _NO_GUARD):
        return lib.trl_fused_degraded() == 1


def fused_reset(device=None) -> None:
    """Re-arm the single-launch route on ``device`` after :func:`fused_degraded` (the caller knows that nothing else
    holds compute units any more)."""
    lib = _extension.get_lib()
    with (torch.cuda.device(device) if device is not None else _NO_GUARD):
        _extension.check(lib.trl_fused_reset(), lib)


def _lookback_workspace(lib, dev, O, T, I):
    """Workspace for the decoupled look-back scan of few, very long rows (trl_value.h:
    trl_workspace_bytes); None for every other shape (the library then needs none)."""
    if I != 1 or T < 2048 or O >= 4096:
        return None
    return _workspace(lib, dev, O, T, I)


def _launch_gae(c: _GaeCall, want_stats: bool = False, algo: int = ALGO_AUTO, out=None):
    lib = _extension.get_lib()
    host = _on_host(c.reward)
    dev = c.reward.device
    adv, vt = out if out is not None else (torch.empty(c.shape, dtype=c.vdt, device=dev), torch.empty(c.shape, dtype=c.vdt, device=dev))
    stats = ws = None
    ws_bytes = 0
    if c.O * c.T * c.I == 0:  # nothing to scan (the reference's loop runs zero steps): empty outputs, zero moments
        stats = torch.zeros(3, dtype=torch.float64, device=dev) if want_stats else None
        return adv.to(c.out_dtype), vt.to(c.out_dtype), stats
    if host:  # tensors in host memory: the library's row-sequential host path (one family; `algo` does not apply)
        stats = torch.empty(3, dtype=torch.float64) if want_stats else None
        rc = lib.trl_gae_fwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.value), _ptr(c.next_value), _ptr(c.done),
                                 _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lg_t), c.gamma, c.lmbda, c.lg,
                                 c.O, c.T, c.I, _ptr(adv), _ptr(vt), _ptr(stats))
        _extension.check(rc, lib)
        if c.out_dtype != c.vdt:
            adv, vt = adv.to(c.out_dtype), vt.to(c.out_dtype)
        return adv, vt, stats
    if want_stats:
        ws = _workspace(lib, dev, c.O, c.T, c.I)
        stats = torch.empty(3, dtype=torch.float64, device=dev)
    else:
        ws = _lookback_workspace(lib, dev, c.O, c.T, c.I)
    ws_bytes = ws.numel() if ws is not None else 0
    with _on(dev):
        rc = lib.trl_gae_fwd(
            _stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.value), _ptr(c.next_value),
            _ptr(c.done), _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lg_t), c.gamma, c.lmbda, c.lg,
            c.O, c.T, c.I, _ptr(adv), _ptr(vt), _ptr(stats), _ptr(ws), ws_bytes, algo)
    _extension.check(rc, lib)
    if c.out_dtype != c.vdt:
        adv, vt = adv.to(c.out_dtype), vt.to(c.out_dtype)
    return adv, vt, stats


def _launch_gae_normalize(c: _GaeCall, eps: float, algo: int = ALGO_AUTO, out=None, want_stats: bool = False, exchange=None):
    """GAE + normalisation over all elements in one C-ABI call (single launch when eligible).
    ``exchange`` (:class:`StatsExchange`): the batch is sharded over its ranks; the moments are
    summed across GPUs inside the launch (trl_gae_normalize_xchg_fwd)."""
    lib = _extension.get_lib()
    host = _on_host(c.reward)
    dev = c.reward.device
    adv, vt = out if out is not None else (torch.empty(c.shape, dtype=c.vdt, device=dev), torch.empty(c.shape, dtype=c.vdt, device=dev))
    if c.O * c.T * c.I == 0:
        if exchange is not None:
            raise NotImplementedError("gae_normalized: an empty shard cannot take part in the cross-GPU exchange")
        return adv.to(c.out_dtype), vt.to(c.out_dtype), (torch.zeros(3, dtype=torch.float64, device=dev) if want_stats else None)
    if host:
        if exchange is not None:
            _require_hip(c.reward)
        stats = torch.empty(3, dtype=torch.float64) if want_stats else None
        rc = lib.trl_gae_normalize_fwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.value), _ptr(c.next_value), _ptr(c.done),
                                           _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lg_t), c.gamma, c.lmbda, c.lg,
                                           c.O, c.T, c.I, _ptr(adv), _ptr(vt), float(eps), _ptr(stats))
        _extension.check(rc, lib)
        if c.out_dtype != c.vdt:
            adv, vt = adv.to(c.out_dtype), vt.to(c.out_dtype)
        return adv, vt, stats
    ws = _workspace(lib, dev, c.O, c.T, c.I)
    stats = torch.empty(3, dtype=torch.float64, device=dev) if want_stats else None
    head = (_stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.value), _ptr(c.next_value),
            _ptr(c.done), _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lg_t), c.gamma, c.lmbda, c.lg,
            c.O, c.T, c.I, _ptr(adv), _ptr(vt), float(eps), _ptr(stats), _ptr(ws), ws.numel())
    with _on(dev):
        if exchange is not None:
            rc = lib.trl_gae_normalize_xchg_fwd(*head, exchange.comm, algo)
        else:
            rc = lib.trl_gae_normalize_fwd(*head, algo)
    _extension.check(rc, lib)
    if c.out_dtype != c.vdt:
        adv, vt = adv.to(c.out_dtype), vt.to(c.out_dtype)
    return adv, vt, stats



# ---------------------------------------------------------------------------------------------
# autograd: the adjoint of a reverse affine scan is a forward scan with the same coefficients
# ---------------------------------------------------------------------------------------------
# Forward x_t = b_t + a_t x_{t+1}  =>  G_t = dL/db_t = gbar_t + a_{t-1} G_{t-1}, G_{-1} = 0.  The backward passes are
# HIP kernels behind the C ABI (trl_gae_bwd / trl_td_lambda_bwd / trl_td1_bwd / trl_vtrace_bwd, csrc/bwd_kernels.hip):
# they read the incoming gradients and the masks / coefficients once and write every gradient once.  Gradients flow to
# reward / state_value / next_state_value (reference behaviour under differentiable=True, test/test_cost.py:15521-15584),
# to V-trace's log-probabilities, and to gamma / lmbda tensors that require grad (per-element products of the adjoint G
# with the saved outputs -- elementwise torch ops on G; autograd sums them over the broadcast dims).
_BWD_ALGO = ALGO_AUTO  # kernel family of the backward calls (tests pin SEQ / SCAN)
_LAST_BWD_KERNEL = ""


def last_bwd_kernel() -> str:
    """Kernel family of the last backward call.  (autograd runs backward on its own thread and trl_last_kernel is
    per thread, so the backward Functions record the name themselves.)"""
    if _LAST_BWD_KERNEL is None:
        return _HOST.last_bwd_kernel()
    return _LAST_BWD_KERNEL


def _bwd_done(rc, lib, name=None):
    global _LAST_BWD_KERNEL
    _extension.check(rc, lib)
    _LAST_BWD_KERNEL = name or lib.trl_last_kernel().decode()


def _grad_in(g, c, dt):
    """Incoming gradient as a contiguous tensor of dtype ``dt`` (None stays None: the kernels read zeros)."""
    if g is None:
        return None
    g = g.detach()
    if g.dtype != dt:
        g = g.to(dt)
    return g.contiguous()


def _m3(c, m):
    """A mask of the call as a view that broadcasts against (O, T, I) (compact (O, T) masks: mask_bcast)."""
    return m.view(c.O, c.T, 1 if c.mask_bcast else c.I)


def _bwd_ws(lib, c):
    ws = _lookback_workspace(lib, c.reward.device, c.O, c.T, c.I)
    return _ptr(ws), (ws.numel() if ws is not None else 0), ws


def _coef_leaf(x, shape, time_dim):
    """A gamma / lmbda tensor that requires grad, as a differentiable view of the inputs' shape (torch's autograd
    then sums the per-element gradient the Functions below return over the broadcast dims); None otherwise."""
    if not (isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled()):
        return None
    if x.numel() == 1:
        return x.reshape(()).expand(shape)
    return x if tuple(x.shape) == tuple(shape) else _expand_coef(x, shape, time_dim)


def _coef_value(x, c, dt, time_dim) -> torch.Tensor:
    """Detached value of a coefficient in dtype ``dt``: 0-dim, or the canonical (O, T, I) view."""
    dev = c.reward.device
    if isinstance(x, torch.Tensor):
        x = x.detach().to(device=dev, dtype=dt)
        if x.numel() == 1:
            return x.reshape(())
        return (x if tuple(x.shape) == tuple(c.shape) else _expand_coef(x, c.shape, time_dim)).reshape(c.O, c.T, c.I)
    return torch.full((), float(x), dtype=dt, device=dev)


def _coef_grad(g3, leaf, c):
    return None if leaf is None else g3.reshape(c.shape).to(leaf.dtype)


def _next_in_time(x3, last):
    """x_{t+1} along dim 1 with ``last`` (broadcastable to one step) after the final step."""
    return torch.cat([x3[:, 1:], last.expand_as(x3[:, :1])], 1)


def _gae_bwd(c, g_adv, g_vt, need_v: bool, need_nv: bool, need_r: bool):
    """(d_state_value, d_next_state_value, d_reward) of GAE through trl_gae_bwd (None where not needed)."""
    global _LAST_BWD_KERNEL
    host = _HOST if _HOST_TRIED else _host_module()
    if host is not None and c.reward.device.type == "cuda":  # allocations + the C-ABI call in one C++ hop (csrc/host_fast.cpp)
        out = host.gae_bwd(g_adv, g_vt, c.reward, c.done, c.terminated, c.mask_bcast, c.gamma_t, c.lg_t, c.gamma, c.lmbda, c.lg,
                           c.O, c.T, c.I, _DT[c.cdt], need_v, need_nv, need_r, _BWD_ALGO)
        if out is not None:
            if out.__class__ is int:
                _extension.check(out, _extension.get_lib())
            else:
                _LAST_BWD_KERNEL = None  # asked of the host module on demand (last_bwd_kernel)
                return out
    lib = _extension.get_lib()
    dev = c.reward.device
    ga, gv = _grad_in(g_adv, c, c.vdt), _grad_in(g_vt, c, c.vdt)
    new = lambda on: torch.empty(c.shape, dtype=c.vdt, device=dev) if on else None
    d_v, d_nv, d_r = new(need_v), new(need_nv), new(need_r)
    if (d_v is not None or d_nv is not None or d_r is not None) and _on_host(c.reward):
        rc = lib.trl_gae_bwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(ga), _ptr(gv), _ptr(c.done), _ptr(c.terminated), c.mask_bcast,
                                 _ptr(c.gamma_t), _ptr(c.lg_t), c.gamma, c.lmbda, c.lg, c.O, c.T, c.I, _ptr(d_r), _ptr(d_v), _ptr(d_nv))
        _bwd_done(rc, lib, "gae_bwd_cpu")
    elif d_v is not None or d_nv is not None or d_r is not None:
        wsp, wsb, _ws = _bwd_ws(lib, c)
        with _on(dev):
            rc = lib.trl_gae_bwd(_stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(ga), _ptr(gv), _ptr(c.done), _ptr(c.terminated),
                                 c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lg_t), c.gamma, c.lmbda, c.lg, c.O, c.T, c.I,
                                 _ptr(d_r), _ptr(d_v), _ptr(d_nv), wsp, wsb, _BWD_ALGO)
        _bwd_done(rc, lib)
    return d_v, d_nv, d_r


def _keep_call(ctx, call):
    """Keep the prepared call for backward together with the version counters of its tensors.  The call holds detached
    aliases of the caller's tensors (same storage, same version counter when no conversion was needed) and the backward
    kernels re-read them: an in-place update of a rollout buffer between forward and backward must raise, as torch's own
    autograd does for saved tensors, instead of giving silently wrong gradients."""
    ctx.call = call
    ctx.versions = tuple((name, t, t._version) for name, t in vars(call).items() if isinstance(t, torch.Tensor))


def _kept_call(ctx):
    for name, t, version in ctx.versions:
        if t._version != version:
            raise RuntimeError(
                f"one of the variables needed for gradient computation has been modified by an inplace operation: the "
                f"estimator's `{name}` input is at version {t._version}; expected version {version} (the backward kernels "
                f"re-read the inputs of the forward call)")
    return ctx.call


class _GaeFunction(torch.autograd.Function):
    """``coefs = (gamma, lmbda, time_dim)`` as the caller passed them; ``gamma_e`` / ``lmbda_e`` are their
    differentiable full-shape views (:func:`_coef_leaf`) or None."""

    @staticmethod
    def forward(ctx, state_value, next_state_value, reward, gamma_e, lmbda_e, call, normalize_eps, coefs):
        if normalize_eps is None:
            adv, vt, _ = _launch_gae(call)
        else:
            raise NotImplementedError("differentiable normalised GAE: normalise the differentiable advantage with torch ops")
        _keep_call(ctx, call)
        ctx.coefs = coefs
        ctx.in_dtypes = (state_value.dtype, next_state_value.dtype, reward.dtype)
        ctx.in_shapes = (state_value.shape, next_state_value.shape, reward.shape)
        ctx.leaves = (gamma_e, lmbda_e)
        if gamma_e is not None or lmbda_e is not None:
            ctx.save_for_backward(adv)
        return adv, vt

    @staticmethod
    def backward(ctx, g_adv, g_vt):
        c = _kept_call(ctx)
        gamma_e, lmbda_e = ctx.leaves
        want_coef = gamma_e is not None or lmbda_e is not None
        none5 = (None, None, None, None, None)
        if g_adv is None and g_vt is None:
            return (None, None, None) + none5
        dev = c.reward.device
        need = ctx.needs_input_grad
        d_v, d_nv, d_r = _gae_bwd(c, g_adv, g_vt, need[0], need[1], need[2] or want_coef)
        d_gamma = d_lmbda = None
        if want_coef:
            # A_t = delta_t + (lmbda_t gamma_t (1 - done_t)) A_{t+1},  delta_t = r_t + gamma_t (1 - terminated_t) v'_t - v_t
            shp3 = (c.O, c.T, c.I)
            dt = torch.promote_types(c.vdt, c.cdt)
            (adv,) = ctx.saved_tensors
            A = adv.view(shp3).to(dt)
            G = d_r.view(shp3).to(dt)
            nd, nt = (1 - _m3(c, c.done)).to(dt), (1 - _m3(c, c.terminated)).to(dt)
            d_lg = G * nd * _next_in_time(A, torch.zeros((), dtype=dt, device=dev))
            gamma, lmbda, time_dim = ctx.coefs
            if gamma_e is not None:
                d_gamma = _coef_grad(G * nt * c.next_value.view(shp3).to(dt) + d_lg * _coef_value(lmbda, c, dt, time_dim), gamma_e, c)
            if lmbda_e is not None:
                d_lmbda = _coef_grad(d_lg * _coef_value(gamma, c, dt, time_dim), lmbda_e, c)
        outs = tuple(None if o is None or not n else o.to(d).view(s)
                     for o, n, d, s in zip((d_v, d_nv, d_r), need[:3], ctx.in_dtypes, ctx.in_shapes))
        return outs + (d_gamma, d_lmbda, None, None, None)


def _td_bwd(c, g_ret, td1: bool, need_nv: bool, need_r: bool):
    """(d_next_state_value, d_reward) of TD(lambda) / TD(1) through trl_td_lambda_bwd / trl_td1_bwd."""
    global _LAST_BWD_KERNEL
    host = _HOST if _HOST_TRIED else _host_module()
    if host is not None and c.reward.device.type == "cuda" and g_ret is not None:
        out = host.td_bwd(g_ret, c.reward, c.done, c.terminated, c.mask_bcast, c.gamma_t, None if td1 else c.lmbda_t, c.gamma,
                          1.0 if td1 else c.lmbda, c.O, c.T, c.I, _DT[c.cdt], td1, need_nv, need_r, _BWD_ALGO)
        if out is not None:
            if out.__class__ is int:
                _extension.check(out, _extension.get_lib())
            else:
                _LAST_BWD_KERNEL = None
                return out
    lib = _extension.get_lib()
    dev = c.reward.device
    gr = _grad_in(g_ret, c, c.vdt)
    d_nv = torch.empty(c.shape, dtype=c.vdt, device=dev) if need_nv else None
    d_r = torch.empty(c.shape, dtype=c.vdt, device=dev) if need_r else None
    if _on_host(c.reward):
        if td1:
            rc = lib.trl_td1_bwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(gr), _ptr(c.done), _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t),
                                     c.gamma, c.O, c.T, c.I, _ptr(d_r), _ptr(d_nv))
        else:
            rc = lib.trl_td_lambda_bwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(gr), _ptr(c.done), _ptr(c.terminated), c.mask_bcast,
                                           _ptr(c.gamma_t), _ptr(c.lmbda_t), c.gamma, c.lmbda, c.O, c.T, c.I, _ptr(d_r), _ptr(d_nv))
        _bwd_done(rc, lib, "td1_bwd_cpu" if td1 else "td_lambda_bwd_cpu")
        return d_nv, d_r
    wsp, wsb, _ws = _bwd_ws(lib, c)
    with _on(dev):
        if td1:
            rc = lib.trl_td1_bwd(_stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(gr), _ptr(c.done), _ptr(c.terminated), c.mask_bcast,
                                 _ptr(c.gamma_t), c.gamma, c.O, c.T, c.I, _ptr(d_r), _ptr(d_nv), wsp, wsb, _BWD_ALGO)
        else:
            rc = lib.trl_td_lambda_bwd(_stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(gr), _ptr(c.done), _ptr(c.terminated), c.mask_bcast,
                                       _ptr(c.gamma_t), _ptr(c.lmbda_t), c.gamma, c.lmbda, c.O, c.T, c.I, _ptr(d_r), _ptr(d_nv),
                                       wsp, wsb, _BWD_ALGO)
    _bwd_done(rc, lib)
    return d_nv, d_r


def _td_fixed_bwd(c, g_ret, td1: bool, need_nv: bool, need_r: bool, need_gamma: bool, need_lmbda: bool):
    """Gradients of the ``rolling_gamma=False`` forms through trl_td_fixed_bwd: ``(d_next_state_value, d_reward)`` in the value
    dtype and the per-element ``(d_gamma, d_lmbda)`` in the compute dtype (None where not needed)."""
    lib = _extension.get_lib()
    dev = c.reward.device
    if _on_host(c.reward):
        raise NotImplementedError("torchrl_amd: the backward of rolling_gamma=False runs on the GPU only (trl_td_fixed_bwd has no host "
                                  "counterpart); move the tensors to the HIP device or call under torch.no_grad()")
    gr = _grad_in(g_ret, c, c.vdt)
    cdt = torch.promote_types(c.vdt, c.cdt)
    new = lambda on, dt: torch.empty(c.shape, dtype=dt, device=dev) if on else None
    d_nv, d_r, d_g, d_l = new(need_nv, c.vdt), new(need_r, c.vdt), new(need_gamma, cdt), new(need_lmbda and not td1, cdt)
    if d_nv is None and d_r is None and d_g is None and d_l is None:
        return None, None, None, None
    with _on(dev):
        rc = lib.trl_td_fixed_bwd(_stream(dev), _DT[c.vdt], _DT[c.cdt], 1 if td1 else 0, _ptr(gr), _ptr(c.reward), _ptr(c.next_value),
                                  _ptr(c.done), _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lmbda_t), c.gamma, c.lmbda,
                                  c.O, c.T, c.I, _ptr(d_r), _ptr(d_nv), _ptr(d_g), _ptr(d_l))
    _bwd_done(rc, lib)
    return d_nv, d_r, d_g, d_l


class _TdLambdaFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, next_state_value, reward, gamma_e, lmbda_e, call, out_dtype, coefs):
        _keep_call(ctx, call)
        ctx.coefs = coefs
        ctx.in_dtypes = (next_state_value.dtype, reward.dtype)
        ctx.in_shapes = (next_state_value.shape, reward.shape)
        ctx.leaves = (gamma_e, lmbda_e)
        ret = _launch_td_lambda(call, out_dtype)
        if gamma_e is not None or lmbda_e is not None:
            ctx.save_for_backward(ret)
        return ret

    @staticmethod
    def backward(ctx, g_ret):
        c = _kept_call(ctx)
        gamma_e, lmbda_e = ctx.leaves
        want_coef = gamma_e is not None or lmbda_e is not None
        if g_ret is None:
            return (None,) * 7
        need = ctx.needs_input_grad
        if not c.rolling:   # O(T^2) form: its own kernels, coefficient gradients included (trl_td_fixed_bwd)
            d_nv, d_r, d_g, d_l = _td_fixed_bwd(c, g_ret, False, need[0], need[1], gamma_e is not None, lmbda_e is not None)
            outs = tuple(None if o is None or not n else o.to(d).view(s)
                         for o, n, d, s in zip((d_nv, d_r), need[:2], ctx.in_dtypes, ctx.in_shapes))
            return outs + (_coef_grad(d_g, gamma_e, c) if d_g is not None else None,
                           _coef_grad(d_l, lmbda_e, c) if d_l is not None else None, None, None, None)
        d_nv, d_r = _td_bwd(c, g_ret, False, need[0], need[1] or want_coef)
        d_gamma = d_lmbda = None
        if want_coef:
            # R_t = r_t + gamma_t((1-lmbda_t) v'_t + lmbda_t (g_{t+1} nd_t + v'_t d_t)); g_T := v'_{T-1}  (v' masked by nt)
            shp3 = (c.O, c.T, c.I)
            dt = torch.promote_types(c.vdt, c.cdt)
            dev = c.reward.device
            (ret,) = ctx.saved_tensors
            R = ret.view(shp3).to(dt)
            G = d_r.view(shp3).to(dt)
            dn = _m3(c, c.done).to(dt)
            nt = (1 - _m3(c, c.terminated)).to(dt)
            gam = c.gamma_t.view(shp3).to(dt) if c.gamma_t is not None else torch.full((), c.gamma, dtype=dt, device=dev)
            lam = c.lmbda_t.view(shp3).to(dt) if c.lmbda_t is not None else torch.full((), c.lmbda, dtype=dt, device=dev)
            nvm = c.next_value.view(shp3).to(dt) * nt
            g1 = _next_in_time(R, nvm[:, -1:]) * (1 - dn) + nvm * dn   # what lmbda_t multiplies at step t
            if gamma_e is not None:
                d_gamma = _coef_grad(G * ((1 - lam) * nvm + lam * g1), gamma_e, c)
            if lmbda_e is not None:
                d_lmbda = _coef_grad(G * gam * (g1 - nvm), lmbda_e, c)
        outs = tuple(None if o is None or not n else o.to(d).view(s)
                     for o, n, d, s in zip((d_nv, d_r), need[:2], ctx.in_dtypes, ctx.in_shapes))
        return outs + (d_gamma, d_lmbda, None, None, None)


class _Td1Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, next_state_value, reward, gamma_e, call, out_dtype):
        _keep_call(ctx, call)
        ctx.in_dtypes = (next_state_value.dtype, reward.dtype)
        ctx.in_shapes = (next_state_value.shape, reward.shape)
        ctx.gamma_e = gamma_e
        ret = _launch_td1(call, out_dtype)
        if gamma_e is not None:
            ctx.save_for_backward(ret)
        return ret

    @staticmethod
    def backward(ctx, g_ret):
        c = _kept_call(ctx)
        if g_ret is None:
            return (None,) * 5
        need = ctx.needs_input_grad
        want_coef = ctx.gamma_e is not None
        if not c.rolling:   # O(T^2) form (trl_td_fixed_bwd)
            d_nv, d_r, d_g, _ = _td_fixed_bwd(c, g_ret, True, need[0], need[1], want_coef, False)
            outs = tuple(None if o is None or not n else o.to(d).view(s)
                         for o, n, d, s in zip((d_nv, d_r), need[:2], ctx.in_dtypes, ctx.in_shapes))
            return outs + (_coef_grad(d_g, ctx.gamma_e, c) if d_g is not None else None, None, None)
        d_nv, d_r = _td_bwd(c, g_ret, True, need[0], need[1] or want_coef)
        d_gamma = None
        if want_coef:
            # R_t = r_t + gamma_t nt_t ((1 - dnt_t) R_{t+1} + dnt_t v'_t),  R_T := v'_{T-1}
            shp3 = (c.O, c.T, c.I)
            dt = torch.promote_types(c.vdt, c.cdt)
            (ret,) = ctx.saved_tensors
            R = ret.view(shp3).to(dt)
            G = d_r.view(shp3).to(dt)
            nv = c.next_value.view(shp3).to(dt)
            d8, t8 = _m3(c, c.done), _m3(c, c.terminated)
            dnt = (d8 & (1 - t8)).to(dt)
            nt = (1 - t8).to(dt)
            d_gamma = _coef_grad(G * nt * ((1 - dnt) * _next_in_time(R, nv[:, -1:]) + dnt * nv), ctx.gamma_e, c)
        outs = tuple(None if o is None or not n else o.to(d).view(s)
                     for o, n, d, s in zip((d_nv, d_r), need[:2], ctx.in_dtypes, ctx.in_shapes))
        return outs + (d_gamma, None, None)


def _vtrace_bwd(c, g_adv, g_vs, vs, need, need_gamma: bool):
    """Gradients w.r.t. (log_pi, log_mu, state_value, next_state_value, reward) -- None where ``need`` is False -- and the
    per-element gamma gradient (or None), through ONE launch of trl_vtrace_bwd."""
    lib = _extension.get_lib()
    dev = c.reward.device
    odt = torch.promote_types(c.vdt, c.cdt)
    ga, gs = _grad_in(g_adv, c, odt), _grad_in(g_vs, c, odt)
    new = lambda on, dt=c.vdt: torch.empty(c.shape, dtype=dt, device=dev) if on else None
    outs = [new(n) for n in need]
    d_gamma_t = new(need_gamma, odt)
    vs = vs.detach()
    if vs.dtype != odt:
        vs = vs.to(odt)
    vs = vs.contiguous()
    if _on_host(c.reward):
        rc = lib.trl_vtrace_bwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(ga), _ptr(gs), _ptr(c.log_pi), _ptr(c.log_mu), _ptr(c.value),
                                    _ptr(c.next_value), _ptr(c.reward), _ptr(vs), _ptr(c.done), _ptr(c.terminated), c.mask_bcast,
                                    _ptr(c.gamma_t), c.gamma, c.rho_thresh, c.c_thresh, _ptr(c.rho_t), _ptr(c.c_t), c.O, c.T, c.I,
                                    *(_ptr(o) for o in outs), _ptr(d_gamma_t))
        _bwd_done(rc, lib, "vtrace_bwd_cpu")
        return outs, d_gamma_t
    wsp, wsb, _ws = _bwd_ws(lib, c)
    with _on(dev):
        rc = lib.trl_vtrace_bwd(_stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(ga), _ptr(gs), _ptr(c.log_pi), _ptr(c.log_mu),
                                _ptr(c.value), _ptr(c.next_value), _ptr(c.reward), _ptr(vs), _ptr(c.done),
                                _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), c.gamma, c.rho_thresh, c.c_thresh,
                                _ptr(c.rho_t), _ptr(c.c_t), c.O, c.T, c.I, *(_ptr(o) for o in outs), _ptr(d_gamma_t), wsp, wsb, _BWD_ALGO)
    _bwd_done(rc, lib)
    return outs, d_gamma_t


class _VtraceFunction(torch.autograd.Function):
    """V-trace with gradients to log_pi / log_mu / state_value / next_state_value / reward (and a gamma tensor).
    Forward on trl_vtrace_fwd; backward = ONE launch of trl_vtrace_bwd."""

    @staticmethod
    def forward(ctx, log_pi, log_mu, state_value, next_state_value, reward, gamma_e, call):
        adv, vs = _launch_vtrace(call)
        _keep_call(ctx, call)
        ctx.gamma_e = gamma_e
        ctx.ins = tuple((t.dtype, t.shape) for t in (log_pi, log_mu, state_value, next_state_value, reward))
        ctx.save_for_backward(vs)
        return adv, vs

    @staticmethod
    def backward(ctx, g_adv, g_vs):
        c = _kept_call(ctx)
        if g_adv is None and g_vs is None:
            return (None,) * 7
        (vs,) = ctx.saved_tensors
        need = ctx.needs_input_grad
        outs, d_gamma_t = _vtrace_bwd(c, g_adv, g_vs, vs, need[:5], ctx.gamma_e is not None)
        res = tuple(None if o is None else o.to(d).view(s) for o, (d, s) in zip(outs, ctx.ins))
        return res + (_coef_grad(d_gamma_t, ctx.gamma_e, c) if d_gamma_t is not None else None, None)


# ---------------------------------------------------------------------------------------------
# GAE fast path: the common call (python-float gamma / lmbda, contiguous same-dtype tensors on the current
# device, time at dim -2, no autograd) goes from the public signature to ONE ctypes call without building the
# general call description -- the general path costs 2 - 4x the kernel at the small BASELINE shapes.  Anything
# else (and every error) falls through to the general path, which validates and raises as the reference does.
# ---------------------------------------------------------------------------------------------
_FAST_COEF = {}   # (gamma, lmbda, default dtype) -> (gamma rounded, lmbda*gamma rounded, coefficient dtype code)
_FAST_NEED = {}   # (O, T, I) -> trl_workspace_bytes
_T32, _T64, _TBOOL = torch.float32, torch.float64, torch.bool


_HOST = None   # the _trl_host module once _host_module() has initialised it


def scalar_gae_coef(gamma, lmbda):
    """``(gamma, lmbda*gamma, coefficient dtype code)`` as the reference's products with an int32 mask round them
    (functional.py:172-174), for scalar-like ``gamma`` / ``lmbda`` -- python numbers, 0-dim or 1-element HOST tensors (the
    modules' buffers: two float32 buffers multiply in float32, two python floats in double) -- exactly what
    :func:`_prepare_gae` computes; None for per-element coefficients.  The tuple can stand in for ``gamma`` (with
    ``lmbda=None``) in :func:`_gae_fast`."""
    if _is_full(gamma) or _is_full(lmbda):
        return None
    g_val, g_dt = _scalar_mask_coef(gamma)
    lg_val, lg_dt = _scalar_mask_coef(lmbda * gamma)
    cdt = torch.promote_types(g_dt, lg_dt)
    if cdt not in _DT:
        return None
    return g_val, lg_val, _DT[cdt]


_fast_coef = scalar_gae_coef


def _host_workspace(dev_index: int, O: int, T: int, I: int):
    dev = torch.device("cuda", dev_index)
    ws = _workspace(_extension.get_lib(), dev, O, T, I)
    _HOST.set_workspace(dev_index, _stream(dev), ws.data_ptr(), ws.numel())


def _host_grouped_scratch(dev_index: int, outer: int, G: int, inner: int):
    from ... import _utils
    dev = torch.device("cuda", dev_index)
    _utils._scratch(_extension.get_lib(), dev, _stream(dev), outer, G, inner)   # registers itself with the host module


def _host_module():
    """Load and initialise csrc/host_fast.cpp's module (once); None when it is not there."""
    global _HOST, _HOST_TRIED
    if not _HOST_TRIED:
        _HOST_TRIED = True
        mod = _extension.get_host()
        if mod is not None:
            lib = _extension.get_lib()
            mod.init(_extension.fn_address(lib.trl_gae_fwd), _extension.fn_address(lib.trl_gae_normalize_fwd),
                     _extension.fn_address(lib.trl_workspace_bytes), _extension.fn_address(lib.trl_td_lambda_fwd),
                     _extension.fn_address(lib.trl_vtrace_fwd), _extension.fn_address(lib.trl_standardize),
                     _extension.fn_address(lib.trl_standardize_grouped), _extension.fn_address(lib.trl_grouped_workspace_bytes),
                     _extension.fn_address(lib.trl_gae_bwd), _extension.fn_address(lib.trl_td_lambda_bwd),
                     _extension.fn_address(lib.trl_td1_bwd), _extension.fn_address(lib.trl_last_kernel),
                     _host_workspace, torch._C._cuda_getCurrentRawStream, _host_grouped_scratch)
            for (di, st), ws in _WORKSPACES.items():
                mod.set_workspace(di, st, ws.data_ptr(), ws.numel())
            from ... import _utils
            for (di, st, G), hit in _utils._SCRATCH.items():
                mod.set_grouped_scratch(di, st, G, hit[0].data_ptr(), hit[1], hit[2])
            _HOST = mod
    return _HOST


_HOST_TRIED = False


def host_fast_path() -> bool:
    """True when the entry points go through the C++ host module (bench.py reports it beside entry_point_ms)."""
    return _host_module() is not None


def _host_call(fn, *args):
    """One call into the host module: its outputs, or None when the call is not a fast-path one."""
    out = fn(*args)
    if out.__class__ is int:
        _extension.check(out, _extension.get_lib())
    return out


def _step_mask(m: torch.Tensor, shape, td: int) -> bool:
    """True when ``m`` is a per-step mask expanded (stride 0) over the dims after time: its ``[O, T]`` bytes are dense."""
    st = m.stride()
    acc = 1
    for k in range(len(shape) - 1, -1, -1):
        if k > td:
            if st[k] != 0 and shape[k] != 1:
                return False
        else:
            if st[k] != acc and shape[k] != 1:
                return False
            acc *= shape[k]
    return True


def _gae_fast(gamma, lmbda, state_value, next_state_value, reward, done, terminated, eps, algo=ALGO_AUTO, time_dim=-2):
    """``(advantage, value_target)`` through the fast path (``eps`` not None: normalised advantage), or None.
    With lib/_trl_host.so built the whole of it -- checks, output allocation, C-ABI call -- is one C++ call
    (csrc/host_fast.cpp); the python below is the same route for a tree without that module."""
    host = _HOST if _HOST_TRIED else _host_module()
    if host is not None:
        out = host.gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, eps, algo, time_dim)
        if out.__class__ is int:
            _extension.check(out, _extension.get_lib())
        return out
    pre = gamma if (gamma.__class__ is tuple and lmbda is None) else None
    if pre is None and (type(gamma) is not float or type(lmbda) is not float):
        return None
    if terminated is None:
        terminated = done
    nsv = next_state_value
    shape = nsv.shape
    if len(shape) < 2 or not (state_value.shape == shape == reward.shape == done.shape == terminated.shape):
        return None
    dt = nsv.dtype
    if (dt is not _T32 and dt is not _T64) or state_value.dtype is not dt or reward.dtype is not dt \
            or done.dtype is not _TBOOL or terminated.dtype is not _TBOOL:
        return None
    dev = nsv.device
    if dev.type != "cuda" or dev.index != torch.cuda.current_device() or state_value.device != dev or reward.device != dev \
            or done.device != dev or terminated.device != dev:
        return None
    if not (nsv.is_contiguous() and state_value.is_contiguous() and reward.is_contiguous()):
        return None
    nd = len(shape)
    td = time_dim + nd if time_dim < 0 else time_dim
    if not 0 <= td < nd:
        return None
    bcast = 0
    if not (done.is_contiguous() and terminated.is_contiguous()):
        if not (_step_mask(done, shape, td) and _step_mask(terminated, shape, td)):
            return None
        bcast = 1   # one byte per (outer row, step), shared by the feature elements (VMAS: done expanded over the agents)
    if torch.is_grad_enabled() and (nsv.requires_grad or state_value.requires_grad or reward.requires_grad):
        return None
    n = nsv.numel()
    if n == 0:
        return None
    key = (gamma, lmbda, torch.get_default_dtype())
    coef = pre if pre is not None else _FAST_COEF.get(key)
    if coef is None:
        g_val, g_dt = _scalar_mask_coef(gamma)
        lg_val, lg_dt = _scalar_mask_coef(lmbda * gamma)
        cdt = torch.promote_types(g_dt, lg_dt)
        if cdt not in _DT:
            return None
        coef = _FAST_COEF[key] = (g_val, lg_val, _DT[cdt])
    T = shape[td]
    I = 1
    for k in range(td + 1, nd):
        I *= shape[k]
    O = n // (T * I)
    lib = _extension.get_lib()
    stream = torch._C._cuda_getCurrentRawStream(dev.index)
    adv, vt = torch.empty_like(nsv), torch.empty_like(nsv)
    if eps is None:
        ws = _lookback_workspace(lib, dev, O, T, I) if (I == 1 and T >= 2048 and O < 4096) else None
        rc = lib.trl_gae_fwd(stream, _DT[dt], coef[2], reward.data_ptr(), state_value.data_ptr(), nsv.data_ptr(),
                             done.data_ptr(), terminated.data_ptr(), bcast, None, None, coef[0], 0.0, coef[1], O, T, I,
                             adv.data_ptr(), vt.data_ptr(), None, None if ws is None else ws.data_ptr(),
                             0 if ws is None else ws.numel(), algo)
    else:
        okey = (O, T, I)
        need = _FAST_NEED.get(okey)
        if need is None:
            need = _FAST_NEED[okey] = lib.trl_workspace_bytes(O, T, I)
        ws = _WORKSPACES.get((dev.index, stream))
        if ws is None or ws.numel() < need:
            ws = _workspace(lib, dev, O, T, I)
        rc = lib.trl_gae_normalize_fwd(stream, _DT[dt], coef[2], reward.data_ptr(), state_value.data_ptr(), nsv.data_ptr(),
                                       done.data_ptr(), terminated.data_ptr(), bcast, None, None, coef[0], 0.0, coef[1], O, T, I,
                                       adv.data_ptr(), vt.data_ptr(), eps, None, ws.data_ptr(), ws.numel(), algo)
    if rc:
        _extension.check(rc, lib)
    return adv, vt


def _geom_series_like(t: torch.Tensor, r, thr: float) -> torch.Tensor:
    """``[1, r, r**2, ...]`` as a ``[len, 1]`` column with ``t``'s dtype / device, cut where ``r**k`` falls
    under ``thr`` (reference: functional.py:185-210 -- the filter of the reference's conv1d estimators; the
    scan kernels here never truncate, this is kept for the module's surface).  ``r == 0`` gives zeros of
    ``t``'s shape, ``r >= 1`` the full length of ``t``."""
    if isinstance(r, torch.Tensor):
        r = r.item()
    if r == 0.0:
        return torch.zeros_like(t)
    lim = t.numel() if r >= 1.0 else int(math.log(thr) / math.log(r))
    ratios = torch.full_like(t[:lim], r)
    ratios[0] = 1.0
    return ratios.cumprod(0).unsqueeze(-1)   # cumulative products, as the reference forms them (not r**k)


def generalized_advantage_estimate(
    gamma: float,
    lmbda: float,
    state_value: torch.Tensor,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    *,
    time_dim: int = -2,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Generalized advantage estimate of a trajectory (reference: functional.py:121-182).

    Returns ``(advantage, value_target)``.  All tensors must have shape
    ``[*Batch x TimeSteps x *F]``; ``terminated`` defaults to ``done``.
    """
    if torch.compiler.is_compiling():
        (g, gt), (l, lt) = _ops._split_leaf(gamma, next_state_value.shape, time_dim), _ops._split_leaf(lmbda, next_state_value.shape, time_dim)
        return _ops.gae_op(state_value, next_state_value, reward, done, done if terminated is None else terminated, g, l, gt, lt,
                           time_dim, False)
    out = _gae_fast(gamma, lmbda, state_value, next_state_value, reward, done, terminated, None, ALGO_AUTO, time_dim)
    if out is not None:
        return out
    call = _prepare_gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, time_dim)
    if _needs_grad(state_value, next_state_value, reward, gamma, lmbda):
        return _GaeFunction.apply(state_value, next_state_value, reward, _coef_leaf(gamma, call.shape, time_dim),
                                  _coef_leaf(lmbda, call.shape, time_dim), call, None, (gamma, lmbda, time_dim))
    adv, vt, _ = _launch_gae(call)
    return adv, vt


def vec_generalized_advantage_estimate(
    gamma: Union[float, torch.Tensor],
    lmbda: Union[float, torch.Tensor],
    state_value: torch.Tensor,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    *,
    time_dim: int = -2,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Vectorized GAE (reference: functional.py:272-370).  Same kernels as the loop variant:
    the result is exact rather than filter-truncated."""
    if torch.compiler.is_compiling():
        (g, gt), (l, lt) = _ops._split_leaf(gamma, next_state_value.shape, time_dim), _ops._split_leaf(lmbda, next_state_value.shape, time_dim)
        return _ops.gae_op(state_value, next_state_value, reward, done, done if terminated is None else terminated, g, l, gt, lt,
                           time_dim, True)
    out = _gae_fast(gamma, lmbda, state_value, next_state_value, reward, done, terminated, None, ALGO_AUTO, time_dim)
    if out is not None:
        return out
    call = _prepare_gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, time_dim)
    call.out_dtype = _vec_gae_out_dtype(state_value.dtype, gamma, lmbda)
    if _needs_grad(state_value, next_state_value, reward, gamma, lmbda):
        return _GaeFunction.apply(state_value, next_state_value, reward, _coef_leaf(gamma, call.shape, time_dim),
                                  _coef_leaf(lmbda, call.shape, time_dim), call, None, (gamma, lmbda, time_dim))
    adv, vt, _ = _launch_gae(call)
    return adv, vt


def _vec_gae_out_dtype(vdtype: torch.dtype, gamma, lmbda) -> torch.dtype:
    """Output dtype of the reference's vec GAE (functional.py:314-370): scalar-like ``gamma * lmbda`` goes through
    ``_fast_vec_gae`` and keeps the value dtype; a ``gamma * lmbda`` tensor of more than one element multiplies into the
    advantage (``_custom_conv1d``), which then has ``promote(value dtype, that tensor's dtype)`` -- found with the vec
    fixtures of tools/gen_golden_vec.py (float32 values, float64 gamma tensor: the loop returns float32, vec float64)."""
    full = any(isinstance(c, torch.Tensor) and c.numel() > 1 for c in (gamma, lmbda))
    if not full:
        return vdtype
    cdt = torch.result_type(gamma, lmbda)
    return torch.promote_types(vdtype, cdt) if cdt.is_floating_point else vdtype


def gae_with_stats(gamma, lmbda, state_value, next_state_value, reward, done, terminated=None, *, time_dim=-2,
                   algo: int = ALGO_AUTO):
    """GAE plus the fp64 moments ``{N, sum(A), sum(A^2)}`` of the advantage (device tensor of 3
    doubles), produced by the same launch sequence; feeds :func:`normalize_` (after an
    all-reduce when the batch is sharded).  No reference counterpart: the reference computes
    ``adv.mean()`` / ``adv.std()`` as separate passes (advantages.py:1459-1463)."""
    call = _prepare_gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, time_dim)
    return _launch_gae(call, want_stats=True, algo=algo)


class PreparedStep:
    """A fully-bound C-ABI call sequence (fixed buffers, fixed stream) that can be re-issued with
    ``run()`` at the cost of the ctypes calls alone -- for callers that keep their rollout buffers
    in place (training loops, hipGraph capture, benchmarks).  Holds references to every tensor."""

    def __init__(self, calls, keep, outputs, stats=None):
        self._calls, self._keep, self.outputs, self.stats = calls, keep, outputs, stats

    def run(self):
        for fn, args in self._calls:
            rc = fn(*args)
            if rc:
                _extension.check(rc)
        return self.outputs


def prepare_gae_step(call: _GaeCall, eps: Optional[float] = None, *, normalize: bool = True, want_stats: bool = False,
                     algo: int = ALGO_AUTO, exchange=None) -> PreparedStep:
    """Bind GAE (+ normalisation) on ``call``'s tensors to the current stream.

    ``normalize=True``: one ``trl_gae_normalize_fwd`` call (single launch when eligible).
    ``normalize=False, want_stats=True``: ``trl_gae_fwd`` writing the fp64 moments to
    ``PreparedStep.stats`` -- the caller all-reduces them and runs :func:`prepare_normalize`'s step.
    ``exchange`` (:class:`StatsExchange`) with ``normalize=True``: ``trl_gae_normalize_xchg_fwd`` -- the
    moments are summed across the exchange's ranks inside the launch."""
    lib = _extension.get_lib()
    _require_hip(c := call.reward)
    dev = c.device
    adv = torch.empty(call.shape, dtype=call.vdt, device=dev)
    vt = torch.empty(call.shape, dtype=call.vdt, device=dev)
    ws = _workspace(lib, dev, call.O, call.T, call.I)
    stats = torch.empty(3, dtype=torch.float64, device=dev) if want_stats else None
    head = (_stream(dev), _DT[call.vdt], _DT[call.cdt], _ptr(call.reward), _ptr(call.value), _ptr(call.next_value),
            _ptr(call.done), _ptr(call.terminated), call.mask_bcast, _ptr(call.gamma_t), _ptr(call.lg_t), call.gamma, call.lmbda, call.lg,
            call.O, call.T, call.I, _ptr(adv), _ptr(vt))
    if normalize and exchange is not None:  # batch sharded over the exchange's ranks: still one C-ABI call
        calls = [(lib.trl_gae_normalize_xchg_fwd, head + (float(eps), _ptr(stats), _ptr(ws), ws.numel(), exchange.comm, algo))]
    elif normalize:
        calls = [(lib.trl_gae_normalize_fwd, head + (float(eps), _ptr(stats), _ptr(ws), ws.numel(), algo))]
    else:
        calls = [(lib.trl_gae_fwd, head + (_ptr(stats), _ptr(ws), ws.numel(), algo))]
    return PreparedStep(calls, (call, ws, exchange), (adv, vt), stats)


def prepare_normalize(x: torch.Tensor, stats: torch.Tensor, eps: float) -> PreparedStep:
    """Bind the in-place scale pass ``x <- (x - mean) / max(std, eps)`` (moments read from ``stats``)."""
    lib = _extension.get_lib()
    _require_hip(x)
    calls = [(lib.trl_normalize, (_stream(x.device), _DT[x.dtype], _ptr(x), x.numel(), _ptr(stats), float(eps), _ptr(x)))]
    return PreparedStep(calls, (x, stats), (x,), stats)


def gae_normalized(gamma, lmbda, state_value, next_state_value, reward, done, terminated=None, *, eps: float = 1e-4,
                   time_dim=-2, group=None, distributed: bool = False, algo: int = ALGO_AUTO, exchange=None):
    """GAE followed by ``(adv - adv.mean()) / adv.std().clamp_min(eps)`` over all elements: what
    ``GAE(average_gae=True)`` computes (reference: advantages.py:1436-1463).  Returns
    ``(normalised advantage, value_target)``.

    Single GPU: one C-ABI call, one kernel launch when the shape is eligible (FUSED family).
    ``distributed`` / ``group``: the batch is sharded over ranks along the env dimension; the
    fp64 moments are all-reduced (one 24-byte RCCL all-reduce) between the GAE kernel and the
    scale pass, so every rank normalises with the global statistics.  ``exchange``
    (:class:`StatsExchange`, intra-node): the moments cross the GPUs inside the launch instead
    (peer-to-peer mailbox stores over xGMI) and the sharded step stays one launch per GPU."""
    if torch.compiler.is_compiling():
        if exchange is not None or group is not None or distributed or algo != ALGO_AUTO:
            raise NotImplementedError("gae_normalized under torch.compile: single-GPU call only")
        (g, gt), (l, lt) = _ops._split(gamma), _ops._split(lmbda)
        return _ops.gae_normalized_op(state_value, next_state_value, reward, done, done if terminated is None else terminated, g, l,
                                      gt, lt, float(eps), time_dim)
    if exchange is None and group is None and not distributed:
        out = _gae_fast(gamma, lmbda, state_value, next_state_value, reward, done, terminated, float(eps), algo, time_dim)
        if out is not None:
            return out
    call = _prepare_gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, time_dim)
    if _needs_grad(state_value, next_state_value, reward, gamma, lmbda):
        # differentiable path: GAE through the autograd function, normalisation as the reference's torch ops
        adv, vt = _GaeFunction.apply(state_value, next_state_value, reward, _coef_leaf(gamma, call.shape, time_dim),
                                     _coef_leaf(lmbda, call.shape, time_dim), call, None, (gamma, lmbda, time_dim))
        return (adv - adv.mean()) / adv.std().clamp_min(eps), vt
    dist = torch.distributed
    if exchange is not None:
        adv, vt, _ = _launch_gae_normalize(call, eps, algo=algo, exchange=exchange)
        return adv, vt
    if (distributed or group is not None) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        adv, vt, stats = _launch_gae(call, want_stats=True, algo=ALGO_AUTO if algo == ALGO_FUSED else algo)
        dist.all_reduce(stats, group=group)
        adv = normalize_(adv if adv.dtype == call.vdt else adv.to(call.vdt), stats, eps)
        return adv, vt
    adv, vt, _ = _launch_gae_normalize(call, eps, algo=algo)
    return adv, vt


# ---------------------------------------------------------------------------------------------
# normalisation
# ---------------------------------------------------------------------------------------------
def moments(x: torch.Tensor) -> torch.Tensor:
    """``{N, sum, sumsq}`` of ``x`` in fp64 (device tensor of 3 doubles)."""
    lib = _extension.get_lib()
    xc = x.detach().contiguous()
    if xc.dtype not in _DT:
        raise NotImplementedError(f"moments: dtype {xc.dtype} unsupported")
    n = xc.numel()
    if _on_host(x):
        stats = torch.empty(3, dtype=torch.float64)
        _extension.check(lib.trl_moments_cpu(_DT[xc.dtype], _ptr(xc), n, _ptr(stats)), lib)
        return stats
    ws = _workspace(lib, x.device, n, 1, 1)
    stats = torch.empty(3, dtype=torch.float64, device=x.device)
    with _on(x.device):
        rc = lib.trl_moments(_stream(x.device), _DT[xc.dtype], _ptr(xc), n, _ptr(stats), _ptr(ws), ws.numel())
    _extension.check(rc, lib)
    return stats


def normalize_(x: torch.Tensor, stats: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``(x - mean) / std.clamp_min(eps)`` with mean / unbiased std taken from ``stats``
    (reference: advantages.py:1459-1463).  In place unless ``out`` is given."""
    lib = _extension.get_lib()
    if not x.is_contiguous() or x.dtype not in _DT:
        raise ValueError("normalize_: x must be a contiguous float32/float64 tensor")
    if stats.dtype != torch.float64 or stats.numel() != 3 or stats.device != x.device:
        raise ValueError("normalize_: stats must be 3 float64 values on x's device")
    out = x if out is None else out
    if out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError("normalize_: out must match x")
    if _on_host(x):
        _extension.check(lib.trl_normalize_cpu(_DT[x.dtype], _ptr(x), x.numel(), _ptr(stats), float(eps), _ptr(out)), lib)
        return out
    with _on(x.device):
        rc = lib.trl_normalize(_stream(x.device), _DT[x.dtype], _ptr(x), x.numel(), _ptr(stats), float(eps), _ptr(out))
    _extension.check(rc, lib)
    return out


def stats_to_mean_std(stats: torch.Tensor, dtype: torch.dtype, eps: float) -> Tuple[float, float]:
    """Host restatement of what ``trl_normalize`` derives from ``{N, sum, sumsq}``: mean and unbiased
    std rounded to ``dtype``, std clamped from below by ``eps`` (NaN kept).  For tests / logging; reads
    ``stats`` on the host."""
    n, s, ss = (float(v) for v in stats.detach().cpu().tolist())
    mean = s / n
    var = (ss - n * mean * mean) / (n - 1.0) if n > 1 else float("nan")
    if var < 0.0:
        var = 0.0
    mean_s = _round_to(mean, dtype)
    std_s = _round_to(math.sqrt(var), dtype) if var == var else float("nan")
    eps_s = _round_to(eps, dtype)
    if std_s < eps_s:
        std_s = eps_s
    return mean_s, std_s


def standardize(x: torch.Tensor, eps: float, group=None, distributed: bool = False) -> torch.Tensor:
    """Out-of-place ``(x - mean) / std.clamp_min(eps)`` over all elements.  With ``distributed``
    (or an explicit process ``group``) the fp64 moments are all-reduced first -- one 24-byte RCCL
    all-reduce -- so every rank normalises with the global statistics of the sharded batch."""
    if torch.compiler.is_compiling() and not distributed and group is None:
        return _ops.standardize_op(x, float(eps))
    if torch.is_grad_enabled() and x.requires_grad and not distributed and group is None:
        # differentiable input (differentiable=True estimators): the reference's own torch ops keep the graph
        # (advantages.py:1160-1165, 1828-1832); the single-launch kernel serves the no-grad case
        return (x - x.mean()) / x.std().clamp_min(eps)
    xc = x.detach().contiguous()
    if (distributed or group is not None) and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        st = moments(xc)
        torch.distributed.all_reduce(st, group=group)
        return normalize_(xc, st, eps, out=torch.empty_like(xc))
    # single GPU: one C-ABI call (one single-block launch up to 65536 elements -- a PPO minibatch)
    lib = _extension.get_lib()
    if xc.dtype not in _DT:
        raise NotImplementedError(f"standardize: dtype {xc.dtype} unsupported")
    n = xc.numel()
    out = torch.empty_like(xc)
    if n == 0:
        return out
    if _on_host(xc):
        return normalize_(xc, moments(xc), eps, out=out)
    dev = xc.device
    ws = _workspace(lib, dev, n, 1, 1) if n > 65536 else None
    with _on(dev):
        rc = lib.trl_standardize(_stream(dev), _DT[xc.dtype], _ptr(xc), n, float(eps), _ptr(out), None, _ptr(ws),
                                 ws.numel() if ws is not None else 0)
    _extension.check(rc, lib)
    return out


# ---------------------------------------------------------------------------------------------
# TD(lambda)
# ---------------------------------------------------------------------------------------------
@dataclass
class _TdLambdaCall:
    shape: tuple
    O: int
    T: int
    I: int
    vdt: torch.dtype
    cdt: torch.dtype
    reward: torch.Tensor
    next_value: torch.Tensor
    done: torch.Tensor
    terminated: torch.Tensor
    gamma_t: Optional[torch.Tensor]
    lmbda_t: Optional[torch.Tensor]
    gamma: float
    lmbda: float
    rolling: bool
    mask_bcast: int = 0


def _td_coef(x, done_shape, vdt, name):
    """gamma / lmbda of TD(lambda): a tensor of the inputs' shape is per-element, anything else
    goes through ``torch.full_like(next_state_value, x)`` (functional.py:850-857)."""
    if isinstance(x, torch.Tensor) and tuple(x.shape) == tuple(done_shape):
        return x.detach(), 0.0, (x.dtype if x.dtype.is_floating_point else vdt)
    if isinstance(x, torch.Tensor):
        if x.numel() != 1:
            raise RuntimeError(
                f"{name} must be a scalar or a tensor with the shape of the inputs {tuple(done_shape)}, got {tuple(x.shape)}")
        x = float(x.detach())
    return None, _round_to(float(x), vdt), vdt


def _prepare_td_lambda(gamma, lmbda, next_state_value, reward, done, terminated, rolling_gamma, time_dim,
                       vec: bool = False) -> _TdLambdaCall:
    if terminated is None:
        terminated = done
    if not (next_state_value.shape == reward.shape == done.shape == terminated.shape):
        raise RuntimeError(SHAPE_ERR)
    shape = tuple(next_state_value.shape)
    O, T, I, _ = _canonical_dims(shape, time_dim)
    vdt = _check_value_dtype(reward, next_state_value)
    dev = next_state_value.device
    if vec:
        # the vec variant broadcasts gamma / lmbda tensors ([B, T, 1] documented, functional.py:1045-1047)
        if _is_full(gamma) and tuple(gamma.shape) != shape:
            gamma = _expand_coef(gamma, shape, time_dim)
        if _is_full(lmbda) and tuple(lmbda.shape) != shape:
            lmbda = _expand_coef(lmbda, shape, time_dim)
    g_t, g_val, g_dt = _td_coef(gamma, shape, vdt, "gamma")
    l_t, l_val, l_dt = _td_coef(lmbda, shape, vdt, "lmbda")
    if rolling_gamma is None:
        rolling_gamma = True
    elif not rolling_gamma and g_t is None and l_t is None:
        raise RuntimeError("rolling_gamma=False is expected only with time-sensitive gamma or lambda values")
    cdt = torch.promote_types(g_dt, l_dt)
    if cdt not in _DT:
        raise NotImplementedError(f"gamma / lmbda dtype {cdt} is not supported (float32 / float64 only)")
    if g_t is not None:
        g_t = g_t.to(device=dev, dtype=cdt).contiguous()
    if l_t is not None:
        l_t = l_t.to(device=dev, dtype=cdt).contiguous()
    masks = _masks(done, terminated, shape, time_dim, dev, False)
    return _TdLambdaCall(shape=shape, O=O, T=T, I=I, vdt=vdt, cdt=cdt, reward=_val(reward, vdt, dev),
                         next_value=_val(next_state_value, vdt, dev), done=masks[0],
                         terminated=masks[1], mask_bcast=masks[2], gamma_t=g_t, lmbda_t=l_t, gamma=g_val,
                         lmbda=l_val, rolling=bool(rolling_gamma))


def _launch_td_lambda(c: _TdLambdaCall, out_dtype, algo: int = ALGO_AUTO) -> torch.Tensor:
    lib = _extension.get_lib()
    dev = c.reward.device
    out = torch.empty(c.shape, dtype=c.vdt, device=dev)
    if c.O * c.T * c.I == 0:
        return out.to(out_dtype)
    if _on_host(c.reward):
        rc = lib.trl_td_lambda_fwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.next_value), _ptr(c.done), _ptr(c.terminated),
                                       c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lmbda_t), c.gamma, c.lmbda, 1 if c.rolling else 0,
                                       c.O, c.T, c.I, _ptr(out))
        _extension.check(rc, lib)
        return out if out_dtype == c.vdt else out.to(out_dtype)
    ws = _lookback_workspace(lib, dev, c.O, c.T, c.I) if c.rolling else None
    with _on(dev):
        rc = lib.trl_td_lambda_fwd(
            _stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.next_value), _ptr(c.done), _ptr(c.terminated),
            c.mask_bcast, _ptr(c.gamma_t), _ptr(c.lmbda_t), c.gamma, c.lmbda, 1 if c.rolling else 0, c.O, c.T, c.I, _ptr(out),
            _ptr(ws), ws.numel() if ws is not None else 0, algo)
    _extension.check(rc, lib)
    return out if out_dtype == c.vdt else out.to(out_dtype)


def td_lambda_return_estimate(
    gamma: float,
    lmbda: float,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rolling_gamma: bool = None,
    *,
    time_dim: int = -2,
) -> torch.Tensor:
    r"""TD(:math:`\lambda`) return estimate (reference: functional.py:783-889).

    ``gamma`` / ``lmbda`` may be scalars or tensors with the shape of the inputs.
    ``rolling_gamma=False`` (each return uses the gamma / lmbda of its own first step for the
    whole tail) runs the O(T^2) kernel and needs a gamma or lmbda tensor.
    """
    if torch.compiler.is_compiling():
        (g, gt), (l, lt) = _ops._split_leaf(gamma, next_state_value.shape, time_dim), _ops._split_leaf(lmbda, next_state_value.shape, time_dim)
        return _ops.td_lambda_op(next_state_value, reward, done, done if terminated is None else terminated, g, l, gt, lt,
                                 -1 if rolling_gamma is None else int(bool(rolling_gamma)), time_dim, False)
    host = _HOST if _HOST_TRIED else _host_module()
    if host is not None and (rolling_gamma is None or rolling_gamma):
        out = _host_call(host.td_lambda, gamma, lmbda, next_state_value, reward, done, terminated, ALGO_AUTO, time_dim, True)
        if out is not None:
            return out
    call = _prepare_td_lambda(gamma, lmbda, next_state_value, reward, done, terminated, rolling_gamma, time_dim)
    if _needs_grad(next_state_value, reward, gamma, lmbda):
        return _TdLambdaFunction.apply(next_state_value, reward, _coef_leaf(gamma, call.shape, time_dim),
                                       _coef_leaf(lmbda, call.shape, time_dim), call, next_state_value.dtype,
                                       (gamma, lmbda, time_dim))
    return _launch_td_lambda(call, next_state_value.dtype)


def vec_td_lambda_return_estimate(
    gamma,
    lmbda,
    next_state_value,
    reward,
    done,
    terminated: torch.Tensor | None = None,
    rolling_gamma: Optional[bool] = None,
    *,
    time_dim: int = -2,
):
    r"""Vectorized TD(:math:`\lambda`) return estimate (reference: functional.py:1030-1183).
    Same kernels as the loop variant; keeps the reference's ``NotImplementedError`` for
    ``rolling_gamma=False`` with tensor coefficients."""
    if torch.compiler.is_compiling():
        (g, gt), (l, lt) = _ops._split_leaf(gamma, next_state_value.shape, time_dim), _ops._split_leaf(lmbda, next_state_value.shape, time_dim)
        return _ops.td_lambda_op(next_state_value, reward, done, done if terminated is None else terminated, g, l, gt, lt,
                                 -1 if rolling_gamma is None else int(bool(rolling_gamma)), time_dim, True)
    host = _HOST if _HOST_TRIED else _host_module()
    if host is not None:   # scalar coefficients: the reference's scalar path ignores rolling_gamma (functional.py:1101-1110)
        out = _host_call(host.td_lambda, gamma, lmbda, next_state_value, reward, done, terminated, ALGO_AUTO, time_dim,
                         rolling_gamma is None or bool(rolling_gamma))
        if out is not None:
            return out
    if terminated is None:
        terminated = done
    if not (next_state_value.shape == reward.shape == done.shape == terminated.shape):
        raise RuntimeError(SHAPE_ERR)
    def _is_scalar(t):
        return not isinstance(t, torch.Tensor) or t.numel() == 1

    if not (_is_scalar(gamma) and _is_scalar(lmbda)) and rolling_gamma is not None and not rolling_gamma:
        _, _, _, td = _canonical_dims(tuple(done.shape), time_dim)
        term = terminated.movedim(td, 0) if terminated.ndim > 1 else terminated
        terminated_follows_terminated = term[1:][term[:-1]].all()
        if not terminated_follows_terminated:
            raise NotImplementedError(
                "When using rolling_gamma=False and vectorized TD(lambda) with time-dependent gamma, "
                "make sure that conseducitve trajectories are separated as different batch "
                "items. Propagating a gamma value across trajectories is not permitted with "
                "this method. Check that you need to use rolling_gamma=False, and if so "
                "consider using the non-vectorized version of the return computation or splitting "
                "your trajectories."
            )
        raise NotImplementedError(
            "The vectorized version of TD(lambda) with rolling_gamma=False is currently not available. "
            "To use this feature, use the non-vectorized version of TD(lambda). You can expect "
            "good speed improvements by decorating the function with torch.compile!"
        )
    if _is_scalar(gamma) and _is_scalar(lmbda):
        rolling_gamma = None  # the reference's fast path ignores it (functional.py:1101-1110)
    call = _prepare_td_lambda(gamma, lmbda, next_state_value, reward, done, terminated, rolling_gamma, time_dim, vec=True)
    # the reference's vec variant returns promote(value dtype, dtype of a gamma / lmbda TENSOR), 0-dim ones included
    # (its conv1d path multiplies the values by the coefficient tensors; the loop variant assigns into a value-dtype buffer)
    out_dtype = next_state_value.dtype
    for coef in (gamma, lmbda):
        if isinstance(coef, torch.Tensor) and coef.dtype.is_floating_point:
            out_dtype = torch.promote_types(out_dtype, coef.dtype)
    if _needs_grad(next_state_value, reward, gamma, lmbda):
        return _TdLambdaFunction.apply(next_state_value, reward, _coef_leaf(gamma, call.shape, time_dim),
                                       _coef_leaf(lmbda, call.shape, time_dim), call, out_dtype,
                                       (gamma, lmbda, time_dim))
    return _launch_td_lambda(call, out_dtype)


def _check5(state_value, next_state_value, reward, done, terminated):
    if not (next_state_value.shape == state_value.shape == reward.shape == done.shape == terminated.shape):
        raise RuntimeError(SHAPE_ERR)


def td_lambda_advantage_estimate(
    gamma: float,
    lmbda: float,
    state_value: torch.Tensor,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rolling_gamma: bool = None,
    # not a kwarg because used directly
    time_dim: int = -2,
) -> torch.Tensor:
    r"""TD(:math:`\lambda`) advantage estimate (reference: functional.py:892-969)."""
    if terminated is None:
        terminated = done
    _check5(state_value, next_state_value, reward, done, terminated)
    returns = td_lambda_return_estimate(gamma, lmbda, next_state_value, reward, done, terminated=terminated,
                                        rolling_gamma=rolling_gamma, time_dim=time_dim)
    return returns - state_value


def vec_td_lambda_advantage_estimate(
    gamma,
    lmbda,
    state_value,
    next_state_value,
    reward,
    done,
    terminated: torch.Tensor | None = None,
    rolling_gamma: bool = None,
    # not a kwarg because used directly
    time_dim: int = -2,
):
    r"""Vectorized TD(:math:`\lambda`) advantage estimate (reference: functional.py:1186-1262)."""
    if terminated is None:
        terminated = done
    _check5(state_value, next_state_value, reward, done, terminated)
    return vec_td_lambda_return_estimate(gamma, lmbda, next_state_value, reward, done=done, terminated=terminated,
                                         rolling_gamma=rolling_gamma, time_dim=time_dim) - state_value


# ---------------------------------------------------------------------------------------------
# V-trace
# ---------------------------------------------------------------------------------------------
@dataclass
class _VtraceCall:
    shape: tuple
    O: int
    T: int
    I: int
    vdt: torch.dtype
    cdt: torch.dtype
    log_pi: torch.Tensor
    log_mu: torch.Tensor
    value: torch.Tensor
    next_value: Optional[torch.Tensor]  # None: shifted layout, as in _GaeCall
    reward: torch.Tensor
    done: torch.Tensor
    terminated: torch.Tensor
    gamma_t: Optional[torch.Tensor]
    gamma: float
    rho_thresh: float
    c_thresh: float
    mask_bcast: int = 0
    rho_t: Optional[torch.Tensor] = None  # per-element thresholds (a broadcastable threshold tensor was passed)
    c_t: Optional[torch.Tensor] = None


def _thresh(x, shape, time_dim, vdt, dev):
    """``(scalar, tensor)`` form of a V-trace clipping threshold.  functional.py:1311-1330: python thresholds become
    0-dim default-dtype tensors; a tensor goes to ``rho.clamp_max(tensor)`` as it is, i.e. it broadcasts against the
    (time-transposed) inputs -- handed to the kernels as a per-element tensor in the value dtype."""
    if isinstance(x, torch.Tensor):
        if x.numel() == 1:
            return float(x), None
        full = x.detach() if tuple(x.shape) == tuple(shape) else _expand_coef(x.detach(), shape, time_dim)
        return 0.0, full.to(device=dev, dtype=vdt).contiguous()
    return _round_to(float(x), torch.get_default_dtype()), None


def _prepare_vtrace(gamma, log_pi, log_mu, state_value, next_state_value, reward, done, terminated, rho_thresh,
                    c_thresh, time_dim) -> _VtraceCall:
    if not (next_state_value.shape == state_value.shape == reward.shape == done.shape):
        raise RuntimeError(SHAPE_ERR)
    if terminated is None:
        terminated = done  # functional.py:1320
    shape = tuple(next_state_value.shape)
    O, T, I, _ = _canonical_dims(shape, time_dim)
    vdt = _check_value_dtype(reward, state_value, next_state_value, log_pi, log_mu)
    dev = next_state_value.device
    gamma_t = None
    g_val = 0.0
    if _is_full(gamma):
        cdt = _mask_coef_dtype(gamma)
        gamma_t = _expand_coef(gamma.detach(), shape, time_dim).to(device=dev, dtype=cdt).contiguous()
    else:
        g_val, cdt = _scalar_mask_coef(gamma)
    if cdt not in _DT:
        raise NotImplementedError(f"gamma dtype {cdt} is not supported (float32 / float64 only)")

    def bc(t):
        return t if tuple(t.shape) == shape else _expand_coef(t, shape, time_dim)

    grads = _needs_grad(log_pi, log_mu, state_value, next_state_value, reward, gamma)
    masks = _masks(done, terminated, shape, time_dim, dev, False)
    shifted = _shifted_pair(state_value, next_state_value, shape, time_dim, vdt, grads)
    rho_s, rho_t = _thresh(rho_thresh, shape, time_dim, vdt, dev)
    c_s, c_t = _thresh(c_thresh, shape, time_dim, vdt, dev)
    if rho_t is not None or c_t is not None:  # SEQ family: dense layout, full masks
        shifted = False
        masks = _masks(done, terminated, shape, time_dim, dev, True)
    return _VtraceCall(shape=shape, O=O, T=T, I=I, vdt=vdt, cdt=cdt, log_pi=_val(bc(log_pi), vdt, dev),
                       log_mu=_val(bc(log_mu), vdt, dev), value=state_value.detach() if shifted else _val(state_value, vdt, dev),
                       next_value=None if shifted else _val(next_state_value, vdt, dev), reward=_val(reward, vdt, dev),
                       done=masks[0], terminated=masks[1], mask_bcast=masks[2], gamma_t=gamma_t,
                       gamma=g_val, rho_thresh=rho_s, c_thresh=c_s, rho_t=rho_t, c_t=c_t)


def _launch_vtrace(c: _VtraceCall, algo: int = ALGO_AUTO):
    lib = _extension.get_lib()
    dev = c.reward.device
    odt = torch.promote_types(c.vdt, c.cdt)
    adv = torch.empty(c.shape, dtype=odt, device=dev)
    vs = torch.empty(c.shape, dtype=odt, device=dev)
    if c.O * c.T * c.I == 0:
        return adv, vs
    if _on_host(c.reward):
        rc = lib.trl_vtrace_fwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(c.log_pi), _ptr(c.log_mu), _ptr(c.value), _ptr(c.next_value),
                                    _ptr(c.reward), _ptr(c.done), _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), c.gamma,
                                    c.rho_thresh, c.c_thresh, _ptr(c.rho_t), _ptr(c.c_t), c.O, c.T, c.I, _ptr(adv), _ptr(vs))
        _extension.check(rc, lib)
        return adv, vs
    ws = _lookback_workspace(lib, dev, c.O, c.T, c.I)
    with _on(dev):
        rc = lib.trl_vtrace_fwd(
            _stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(c.log_pi), _ptr(c.log_mu), _ptr(c.value), _ptr(c.next_value),
            _ptr(c.reward), _ptr(c.done), _ptr(c.terminated), c.mask_bcast, _ptr(c.gamma_t), c.gamma, c.rho_thresh, c.c_thresh,
            _ptr(c.rho_t), _ptr(c.c_t), c.O, c.T, c.I, _ptr(adv), _ptr(vs), _ptr(ws), ws.numel() if ws is not None else 0, algo)
    _extension.check(rc, lib)
    return adv, vs


def vtrace_advantage_estimate(
    gamma: float,
    log_pi: torch.Tensor,
    log_mu: torch.Tensor,
    state_value: torch.Tensor,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rho_thresh: Union[float, torch.Tensor] = 1.0,
    c_thresh: Union[float, torch.Tensor] = 1.0,
    # not a kwarg because used directly
    time_dim: int = -2,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """V-Trace off-policy actor critic targets (reference: functional.py:1270-1350).

    Returns ``(advantages, vs)``.  ``terminated=None`` makes ``not_terminated = not_done``.
    """
    if torch.compiler.is_compiling():
        (g, gt), (rs, rt), (cs, ct) = _ops._split_leaf(gamma, next_state_value.shape, time_dim), _ops._split(rho_thresh), _ops._split(c_thresh)
        return _ops.vtrace_op(log_pi, log_mu, state_value, next_state_value, reward, done, done if terminated is None else terminated,
                              g, gt, rs, cs, rt, ct, time_dim)
    host = _HOST if _HOST_TRIED else _host_module()
    if host is not None:
        out = _host_call(host.vtrace, gamma, log_pi, log_mu, state_value, next_state_value, reward, done, terminated, rho_thresh,
                         c_thresh, ALGO_AUTO, time_dim)
        if out is not None:
            return out
    call = _prepare_vtrace(gamma, log_pi, log_mu, state_value, next_state_value, reward, done, terminated,
                           rho_thresh, c_thresh, time_dim)
    if _needs_grad(log_pi, log_mu, state_value, next_state_value, reward, gamma):
        # broadcast log-probabilities enter as differentiable full-shape views: autograd sums their gradient back
        full = lambda t: t if tuple(t.shape) == call.shape else _expand_coef(t, call.shape, time_dim)
        return _VtraceFunction.apply(full(log_pi), full(log_mu), state_value, next_state_value, reward,
                                     _coef_leaf(gamma, call.shape, time_dim), call)
    return _launch_vtrace(call)


# ---------------------------------------------------------------------------------------------
# TD(0), TD(1), reward-to-go -- "next" rows of SURVEY.md 8(f).  TD(1) has its own kernels (trl_td1_fwd: the
# reference's TD(1) loops differ from TD(lambda) at lmbda = 1 in the start value and in how masks enter);
# vec TD(1) is vec TD(lambda) with lmbda = 1, exactly as functional.py:691-700 does.
# ---------------------------------------------------------------------------------------------
def td0_return_estimate(
    gamma: float,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    terminated: torch.Tensor | None = None,
    *,
    done: torch.Tensor | None = None,
) -> torch.Tensor:
    """TD(0) discounted return estimate (reference: functional.py:418-456); elementwise."""
    if done is not None and terminated is None:
        terminated = done.clone()
        warnings.warn("done for td0_return_estimate is deprecated. Pass ``terminated`` instead.")
    if not (next_state_value.shape == reward.shape == terminated.shape):
        raise RuntimeError(SHAPE_ERR)
    not_terminated = (~terminated).int()
    return reward + gamma * not_terminated * next_state_value


def td0_advantage_estimate(
    gamma: float,
    state_value: torch.Tensor,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
) -> torch.Tensor:
    """TD(0) advantage estimate (reference: functional.py:378-415); elementwise."""
    if terminated is None:
        terminated = done.clone()
    _check5(state_value, next_state_value, reward, done, terminated)
    return td0_return_estimate(gamma, next_state_value, reward, terminated) - state_value


def vec_td1_return_estimate(
    gamma,
    next_state_value,
    reward,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rolling_gamma: Optional[bool] = None,
    time_dim: int = -2,
):
    """Vectorized TD(1) return estimate (reference: functional.py:641-700)."""
    return vec_td_lambda_return_estimate(gamma=gamma, next_state_value=next_state_value, reward=reward, done=done,
                                         terminated=terminated, rolling_gamma=rolling_gamma, lmbda=1, time_dim=time_dim)


def _prepare_td1(gamma, next_state_value, reward, done, terminated, rolling_gamma, time_dim) -> _TdLambdaCall:
    """Host logic of TD(1) (functional.py:515-536): a gamma of the inputs' shape is per-element, anything else becomes
    ``torch.full_like(next_state_value, gamma)``."""
    if terminated is None:
        terminated = done
    if not (next_state_value.shape == reward.shape == done.shape == terminated.shape):
        raise RuntimeError(SHAPE_ERR)
    shape = tuple(next_state_value.shape)
    O, T, I, _ = _canonical_dims(shape, time_dim)
    vdt = _check_value_dtype(reward, next_state_value)
    dev = next_state_value.device
    g_t, g_val, cdt = _td_coef(gamma, shape, vdt, "gamma")
    if rolling_gamma is None:
        rolling_gamma = True
    elif not rolling_gamma and g_t is None:
        raise RuntimeError("rolling_gamma=False is expected only with time-sensitive gamma values")
    if cdt not in _DT:
        raise NotImplementedError(f"gamma dtype {cdt} is not supported (float32 / float64 only)")
    if g_t is not None:
        g_t = g_t.to(device=dev, dtype=cdt).contiguous()
    masks = _masks(done, terminated, shape, time_dim, dev, False)
    return _TdLambdaCall(shape=shape, O=O, T=T, I=I, vdt=vdt, cdt=cdt, reward=_val(reward, vdt, dev),
                         next_value=_val(next_state_value, vdt, dev), done=masks[0], terminated=masks[1],
                         mask_bcast=masks[2], gamma_t=g_t, lmbda_t=None, gamma=g_val, lmbda=1.0, rolling=bool(rolling_gamma))


def _launch_td1(c: _TdLambdaCall, out_dtype, algo: int = ALGO_AUTO) -> torch.Tensor:
    lib = _extension.get_lib()
    dev = c.reward.device
    out = torch.empty(c.shape, dtype=c.vdt, device=dev)
    if c.O * c.T * c.I == 0:
        return out.to(out_dtype)
    if _on_host(c.reward):
        rc = lib.trl_td1_fwd_cpu(_DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.next_value), _ptr(c.done), _ptr(c.terminated),
                                 c.mask_bcast, _ptr(c.gamma_t), c.gamma, 1 if c.rolling else 0, c.O, c.T, c.I, _ptr(out))
        _extension.check(rc, lib)
        return out if out_dtype == c.vdt else out.to(out_dtype)
    ws = _lookback_workspace(lib, dev, c.O, c.T, c.I) if c.rolling else None
    with _on(dev):
        rc = lib.trl_td1_fwd(
            _stream(dev), _DT[c.vdt], _DT[c.cdt], _ptr(c.reward), _ptr(c.next_value), _ptr(c.done), _ptr(c.terminated),
            c.mask_bcast, _ptr(c.gamma_t), c.gamma, 1 if c.rolling else 0, c.O, c.T, c.I, _ptr(out),
            _ptr(ws), ws.numel() if ws is not None else 0, algo)
    _extension.check(rc, lib)
    return out if out_dtype == c.vdt else out.to(out_dtype)


def td1_return_estimate(
    gamma: float,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rolling_gamma: bool = None,
    *,
    time_dim: int = -2,
) -> torch.Tensor:
    """TD(1) return estimate (reference: functional.py:464-562).

    ``g = r + gamma*nt*((1-dnt)*g + dnt*v')`` with ``dnt = done & ~terminated``, starting from the raw
    ``v'_{T-1}``; ``rolling_gamma=False`` (needs a gamma tensor of the inputs' shape) reruns every tail with
    its first step's gamma from ``g = 0``, as the reference's loop does (``trl_td1_fwd``)."""
    if torch.compiler.is_compiling():
        g, gt = _ops._split(gamma)
        return _ops.td1_op(next_state_value, reward, done, done if terminated is None else terminated, g, gt,
                           -1 if rolling_gamma is None else int(bool(rolling_gamma)), time_dim)
    call = _prepare_td1(gamma, next_state_value, reward, done, terminated, rolling_gamma, time_dim)
    if _needs_grad(next_state_value, reward, gamma):
        return _Td1Function.apply(next_state_value, reward, _coef_leaf(gamma, call.shape, time_dim), call,
                                  next_state_value.dtype)
    return _launch_td1(call, next_state_value.dtype)


def td1_advantage_estimate(
    gamma: float,
    state_value: torch.Tensor,
    next_state_value: torch.Tensor,
    reward: torch.Tensor,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rolling_gamma: bool = None,
    time_dim: int = -2,
) -> torch.Tensor:
    """TD(1) advantage estimate (reference: functional.py:565-638)."""
    if terminated is None:
        terminated = done
    _check5(state_value, next_state_value, reward, done, terminated)
    return td1_return_estimate(gamma, next_state_value, reward, done, terminated=terminated,
                               rolling_gamma=rolling_gamma, time_dim=time_dim) - state_value


def vec_td1_advantage_estimate(
    gamma,
    state_value,
    next_state_value,
    reward,
    done: torch.Tensor,
    terminated: torch.Tensor | None = None,
    rolling_gamma: bool = None,
    time_dim: int = -2,
):
    """Vectorized TD(1) advantage estimate (reference: functional.py:703-775)."""
    if terminated is None:
        terminated = done
    _check5(state_value, next_state_value, reward, done, terminated)
    return vec_td1_return_estimate(gamma, next_state_value, reward, done=done, terminated=terminated,
                                   rolling_gamma=rolling_gamma, time_dim=time_dim) - state_value


def reward2go(reward, done, gamma, *, time_dim: int = -2):
    """Discounted cumulative sum of rewards per trajectory (reference: functional.py:1358-1432).

    ``R_t = r_t + gamma * (1-done_t) * R_{t+1}`` restarting after every ``done`` and at the end
    of every row: GAE with ``v = v' = 0`` and ``lmbda*gamma -> gamma``.  Like the reference, a
    tensor with more than two dims is flattened so that trailing dims are independent rows."""
    if torch.compiler.is_compiling() and not isinstance(gamma, torch.Tensor):
        return _ops.reward2go_op(reward, done, float(gamma), time_dim)
    if reward.shape != done.shape:
        raise ValueError(f"reward and done must share the same shape, got {reward.shape} and {done.shape}")
    zeros = torch.zeros_like(reward)
    one = torch.ones((), dtype=reward.dtype if reward.dtype.is_floating_point else torch.get_default_dtype())
    g = one * gamma if not isinstance(gamma, torch.Tensor) else gamma
    call = _prepare_gae(1.0, g, zeros, zeros, reward, done, torch.ones_like(done, dtype=torch.bool), time_dim)
    adv, _, _ = _launch_gae(call)
    return adv


# ---------------------------------------------------------------------------------------------
# torch.compile: dynamo cannot look through ctypes, so while compiling every estimator above dispatches to a
# torch.library custom op with a fake implementation (_ops.py): torch.compile(fullgraph=True) traces a PPO-style
# function without graph breaks (the reference instead flips its estimators to the Python loop under dynamo,
# advantages.py:1303-1307).  Eager calls never pay the dispatcher: they go straight to the library.  The helpers
# below have no op of their own and stay opaque (a graph break).
# ---------------------------------------------------------------------------------------------
from . import _ops  # noqa: E402  (imports this module back; used only inside the functions above)

for _name in ("gae_with_stats", "moments", "normalize_"):
    globals()[_name] = torch.compiler.disable(globals()[_name], recursive=True)
del _name
