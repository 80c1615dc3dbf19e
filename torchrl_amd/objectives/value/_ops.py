"""``torch.library`` custom ops over the C-ABI entry points, so that the estimators are ``torch.compile``-friendly.

SURVEY.md 8(b): the reference flips its estimators to the Python loop under dynamo (advantages.py:1303-1307); here every
estimator is ONE opaque op with a fake (meta) implementation, so ``torch.compile(fullgraph=True)`` traces through a
PPO-style function without graph breaks.  The ops are used ONLY while compiling (``torch.compiler.is_compiling()``):
the eager entry points call the library directly -- the dispatcher's per-call cost is several times the small kernels.

Each op's implementation is the eager entry point itself (ctypes underneath); ``register_autograd`` wires the backward
kernels (``trl_gae_bwd`` / ``trl_td_lambda_bwd`` / ``trl_td1_bwd`` / ``trl_vtrace_bwd``) for the value / reward /
log-probability inputs.  Gradients w.r.t. gamma / lmbda TENSORS (round 3): GAE, TD(lambda) with rolling gamma and V-trace's
gamma -- the tensor enters the op as a differentiable view of the inputs' shape (``_split_leaf``) and the backward forms its
per-element gradient with torch ops on the kernel's adjoint, as the eager Functions do; TD(1) and the
``rolling_gamma=False`` forms keep coefficient gradients eager-only.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import functional as F

_NS = "torchrl_amd"


def _coef(scalar: float, tensor: Optional[Tensor]):
    return tensor if tensor is not None else scalar


def _split(x):
    """(float, tensor-or-None) form of a coefficient for an op schema."""
    if isinstance(x, torch.Tensor):
        return 0.0, x
    return float(x), None


def _split_leaf(x, shape, time_dim):
    """:func:`_split` for a coefficient that may require grad: such a tensor enters the op as a differentiable view of the
    inputs' shape (traced torch ops, outside the opaque op -- autograd then sums the op's per-element gradient back over the
    broadcast dims, exactly as for the eager Functions' ``_coef_leaf``)."""
    if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
        return 0.0, F._coef_leaf(x, tuple(shape), time_dim)
    return _split(x)


def _dims3(shape, time_dim):
    nd = len(shape)
    td = time_dim + nd if time_dim < 0 else time_dim
    O = 1
    for k in shape[:td]:
        O *= k
    I = 1
    for k in shape[td + 1:]:
        I *= k
    return O, shape[td], I


def _coef3(scalar, tensor, shape, time_dim, dt, dev):
    """Detached value of a coefficient in dtype ``dt``: 0-dim, or the canonical (O, T, I) view."""
    if tensor is None:
        return torch.full((), scalar, dtype=dt, device=dev)
    t = tensor.detach().to(device=dev, dtype=dt)
    if t.numel() == 1:
        return t.reshape(())
    t = t if tuple(t.shape) == tuple(shape) else F._expand_coef(t, tuple(shape), time_dim)
    return t.reshape(_dims3(shape, time_dim))


def _leaf_grad(g3, leaf):
    """Per-element gradient (O, T, I) -> the coefficient tensor's shape (full shape: what _split_leaf hands the op)."""
    if leaf.numel() != g3.numel():
        raise NotImplementedError("compiled gradient w.r.t. a broadcast coefficient tensor: pass it through the public functions "
                                  "(they expand it to the inputs' shape outside the op)")
    return g3.reshape(leaf.shape).to(leaf.dtype)


def _promote_out(*xs):
    dt = None
    for x in xs:
        if isinstance(x, torch.Tensor) and x.dtype.is_floating_point:
            dt = x.dtype if dt is None else torch.promote_types(dt, x.dtype)
    return dt


# ---- GAE ------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::gae", mutates_args=())
def gae_op(state_value: Tensor, next_state_value: Tensor, reward: Tensor, done: Tensor, terminated: Tensor, gamma: float,
           lmbda: float, gamma_t: Optional[Tensor], lmbda_t: Optional[Tensor], time_dim: int, vec: bool) -> Tuple[Tensor, Tensor]:
    fn = F.vec_generalized_advantage_estimate if vec else F.generalized_advantage_estimate
    return fn(_coef(gamma, gamma_t), _coef(lmbda, lmbda_t), state_value, next_state_value, reward, done, terminated, time_dim=time_dim)


@gae_op.register_fake
def _(state_value, next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim, vec):
    dt = F._vec_gae_out_dtype(state_value.dtype, _coef(gamma, gamma_t), _coef(lmbda, lmbda_t)) if vec else next_state_value.dtype
    return (torch.empty(next_state_value.shape, dtype=dt, device=next_state_value.device),
            torch.empty(next_state_value.shape, dtype=dt, device=next_state_value.device))


@torch.library.custom_op(f"{_NS}::gae_bwd", mutates_args=())
def gae_bwd_op(g_adv: Tensor, g_vt: Tensor, like: Tensor, done: Tensor, terminated: Tensor, gamma: float, lmbda: float,
               gamma_t: Optional[Tensor], lmbda_t: Optional[Tensor], time_dim: int) -> Tuple[Tensor, Tensor, Tensor]:
    c = F._prepare_gae(_coef(gamma, gamma_t), _coef(lmbda, lmbda_t), like, like, like, done, terminated, time_dim)
    d_v, d_nv, d_r = F._gae_bwd(c, g_adv.reshape(like.shape), g_vt.reshape(like.shape), True, True, True)
    return d_v.view(like.shape), d_nv.view(like.shape), d_r.view(like.shape)


@gae_bwd_op.register_fake
def _(g_adv, g_vt, like, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim):
    return torch.empty_like(like), torch.empty_like(like), torch.empty_like(like)


def _gae_setup(ctx, inputs, output):
    state_value, next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim, vec = inputs
    need = ctx.needs_input_grad
    ctx.coef_grads = (gamma_t is not None and need[7], lmbda_t is not None and need[8])
    if any(ctx.coef_grads):   # gradients w.r.t. gamma / lmbda tensors: torch ops on the kernel's adjoint (as the eager Function)
        ctx.save_for_backward(done, terminated, gamma_t, lmbda_t, output[0], next_state_value)
    else:
        ctx.save_for_backward(done, terminated, gamma_t, lmbda_t)
    ctx.cfg = (gamma, lmbda, time_dim)
    ctx.like = (reward.shape, F._check_value_dtype(reward, state_value, next_state_value), reward.device)
    ctx.dtypes = (state_value.dtype, next_state_value.dtype, reward.dtype)


def _gae_backward(ctx, g_adv, g_vt):
    done, terminated, gamma_t, lmbda_t = ctx.saved_tensors[:4]
    gamma, lmbda, time_dim = ctx.cfg
    shape, dt, dev = ctx.like
    like = torch.empty(shape, dtype=dt, device=dev)
    z = lambda g: torch.zeros(shape, dtype=dt, device=dev) if g is None else g.to(dt)
    d_v, d_nv, d_r = gae_bwd_op(z(g_adv), z(g_vt), like, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim)
    d_gamma = d_lmbda = None
    if any(ctx.coef_grads):
        # A_t = delta_t + (lmbda_t gamma_t (1 - done_t)) A_{t+1},  delta_t = r_t + gamma_t (1 - terminated_t) v'_t - v_t
        adv, nv = ctx.saved_tensors[4:6]
        shp3 = _dims3(shape, time_dim)
        cdt = _promote_out(like, gamma_t, lmbda_t)
        A, G = adv.reshape(shp3).to(cdt), d_r.reshape(shp3).to(cdt)
        nd = (~done).reshape(shp3).to(cdt)
        nt = (~terminated).reshape(shp3).to(cdt)
        d_lg = G * nd * F._next_in_time(A, torch.zeros((), dtype=cdt, device=dev))
        if ctx.coef_grads[0]:
            d_gamma = _leaf_grad(G * nt * nv.reshape(shp3).to(cdt) + d_lg * _coef3(lmbda, lmbda_t, shape, time_dim, cdt, dev), gamma_t)
        if ctx.coef_grads[1]:
            d_lmbda = _leaf_grad(d_lg * _coef3(gamma, gamma_t, shape, time_dim, cdt, dev), lmbda_t)
    return (d_v.to(ctx.dtypes[0]), d_nv.to(ctx.dtypes[1]), d_r.to(ctx.dtypes[2]), None, None, None, None, d_gamma, d_lmbda, None, None)


gae_op.register_autograd(_gae_backward, setup_context=_gae_setup)


@torch.library.custom_op(f"{_NS}::gae_normalized", mutates_args=())
def gae_normalized_op(state_value: Tensor, next_state_value: Tensor, reward: Tensor, done: Tensor, terminated: Tensor, gamma: float,
                      lmbda: float, gamma_t: Optional[Tensor], lmbda_t: Optional[Tensor], eps: float, time_dim: int) -> Tuple[Tensor, Tensor]:
    return F.gae_normalized(_coef(gamma, gamma_t), _coef(lmbda, lmbda_t), state_value, next_state_value, reward, done, terminated,
                            eps=eps, time_dim=time_dim)


@gae_normalized_op.register_fake
def _(state_value, next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t, eps, time_dim):
    return torch.empty_like(next_state_value), torch.empty_like(next_state_value)


# ---- TD(lambda) / TD(1) ---------------------------------------------------------------------------------------------------
def _rolling(code: int):
    return None if code < 0 else bool(code)


@torch.library.custom_op(f"{_NS}::td_lambda", mutates_args=())
def td_lambda_op(next_state_value: Tensor, reward: Tensor, done: Tensor, terminated: Tensor, gamma: float, lmbda: float,
                 gamma_t: Optional[Tensor], lmbda_t: Optional[Tensor], rolling: int, time_dim: int, vec: bool) -> Tensor:
    fn = F.vec_td_lambda_return_estimate if vec else F.td_lambda_return_estimate
    return fn(_coef(gamma, gamma_t), _coef(lmbda, lmbda_t), next_state_value, reward, done, terminated, _rolling(rolling), time_dim=time_dim)


@td_lambda_op.register_fake
def _(next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t, rolling, time_dim, vec):
    dt = next_state_value.dtype
    if vec:
        dt = _promote_out(next_state_value, gamma_t, lmbda_t)
    return torch.empty(next_state_value.shape, dtype=dt, device=next_state_value.device)


@torch.library.custom_op(f"{_NS}::td_bwd", mutates_args=())
def td_bwd_op(g_ret: Tensor, like: Tensor, done: Tensor, terminated: Tensor, gamma: float, lmbda: float, gamma_t: Optional[Tensor],
              lmbda_t: Optional[Tensor], time_dim: int, vec: bool, td1: bool) -> Tuple[Tensor, Tensor]:
    if td1:
        c = F._prepare_td1(_coef(gamma, gamma_t), like, like, done, terminated, None, time_dim)
    else:
        c = F._prepare_td_lambda(_coef(gamma, gamma_t), _coef(lmbda, lmbda_t), like, like, done, terminated, None, time_dim, vec=vec)
    d_nv, d_r = F._td_bwd(c, g_ret.reshape(like.shape), td1, True, True)
    return d_nv.view(like.shape), d_r.view(like.shape)


@td_bwd_op.register_fake
def _(g_ret, like, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim, vec, td1):
    return torch.empty_like(like), torch.empty_like(like)


@torch.library.custom_op(f"{_NS}::td_fixed_bwd", mutates_args=())
def td_fixed_bwd_op(g_ret: Tensor, next_state_value: Tensor, reward: Tensor, done: Tensor, terminated: Tensor, gamma: float,
                    lmbda: float, gamma_t: Optional[Tensor], lmbda_t: Optional[Tensor], time_dim: int, td1: bool) -> Tuple[Tensor, Tensor]:
    """rolling_gamma=False (trl_td_fixed_bwd): needs the forward inputs, every start k reruns its own tail."""
    if td1:
        c = F._prepare_td1(_coef(gamma, gamma_t), next_state_value, reward, done, terminated, False, time_dim)
    else:
        c = F._prepare_td_lambda(_coef(gamma, gamma_t), _coef(lmbda, lmbda_t), next_state_value, reward, done, terminated, False, time_dim)
    d_nv, d_r, _, _ = F._td_fixed_bwd(c, g_ret.reshape(reward.shape), td1, True, True, False, False)
    return d_nv.view(reward.shape), d_r.view(reward.shape)


@td_fixed_bwd_op.register_fake
def _(g_ret, next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim, td1):
    dt = F._check_value_dtype(reward, next_state_value)
    return (torch.empty(reward.shape, dtype=dt, device=reward.device), torch.empty(reward.shape, dtype=dt, device=reward.device))


def _td_setup(ctx, inputs, output):
    next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t, rolling, time_dim, vec = inputs
    ctx.fixed = rolling == 0 and (gamma_t is not None or lmbda_t is not None)  # rolling_gamma=False: trl_td_fixed_bwd
    need = ctx.needs_input_grad
    ctx.coef_grads = (gamma_t is not None and need[6], lmbda_t is not None and need[7])
    if any(ctx.coef_grads) and ctx.fixed:
        raise NotImplementedError("compiled gradients w.r.t. gamma / lmbda tensors with rolling_gamma=False: eager mode has them")
    if ctx.fixed:
        ctx.save_for_backward(done, terminated, gamma_t, lmbda_t, next_state_value, reward)
    elif any(ctx.coef_grads):
        ctx.save_for_backward(done, terminated, gamma_t, lmbda_t, next_state_value, output)
    else:
        ctx.save_for_backward(done, terminated, gamma_t, lmbda_t)
    ctx.cfg = (gamma, lmbda, time_dim, vec, False)
    ctx.like = (reward.shape, F._check_value_dtype(reward, next_state_value), reward.device)
    ctx.dtypes = (next_state_value.dtype, reward.dtype)


def _td_backward(ctx, g_ret):
    gamma, lmbda, time_dim, vec, td1 = ctx.cfg
    if ctx.fixed:
        done, terminated, gamma_t, lmbda_t, next_state_value, reward = ctx.saved_tensors
        d_nv, d_r = td_fixed_bwd_op(g_ret.to(ctx.like[1]), next_state_value, reward, done, terminated, gamma, lmbda, gamma_t, lmbda_t,
                                    time_dim, False)
        return (d_nv.to(ctx.dtypes[0]), d_r.to(ctx.dtypes[1])) + (None,) * 9
    done, terminated, gamma_t, lmbda_t = ctx.saved_tensors[:4]
    shape, dt, dev = ctx.like
    like = torch.empty(shape, dtype=dt, device=dev)
    d_nv, d_r = td_bwd_op(g_ret.to(dt), like, done, terminated, gamma, lmbda, gamma_t, lmbda_t, time_dim, vec, td1)
    d_gamma = d_lmbda = None
    if any(ctx.coef_grads):
        # R_t = r_t + gamma_t((1-lmbda_t) v'_t + lmbda_t (g_{t+1} nd_t + v'_t d_t)); g_T := v'_{T-1}  (v' masked by nt)
        nv, ret = ctx.saved_tensors[4:6]
        shp3 = _dims3(shape, time_dim)
        cdt = _promote_out(like, gamma_t, lmbda_t)
        R, G = ret.reshape(shp3).to(cdt), d_r.reshape(shp3).to(cdt)
        dn = done.reshape(shp3).to(cdt)
        nvm = nv.reshape(shp3).to(cdt) * (~terminated).reshape(shp3).to(cdt)
        # scalar coefficients of TD(lambda) take the value dtype (torch.full_like(next_state_value, gamma))
        gam = _coef3(F._round_to(gamma, dt), gamma_t, shape, time_dim, cdt, dev)
        lam = _coef3(F._round_to(lmbda, dt), lmbda_t, shape, time_dim, cdt, dev)
        g1 = F._next_in_time(R, nvm[:, -1:]) * (1 - dn) + nvm * dn   # what lmbda_t multiplies at step t
        if ctx.coef_grads[0]:
            d_gamma = _leaf_grad(G * ((1 - lam) * nvm + lam * g1), gamma_t)
        if ctx.coef_grads[1]:
            d_lmbda = _leaf_grad(G * gam * (g1 - nvm), lmbda_t)
    return (d_nv.to(ctx.dtypes[0]), d_r.to(ctx.dtypes[1]), None, None, None, None, d_gamma, d_lmbda, None, None, None)


td_lambda_op.register_autograd(_td_backward, setup_context=_td_setup)


@torch.library.custom_op(f"{_NS}::td1", mutates_args=())
def td1_op(next_state_value: Tensor, reward: Tensor, done: Tensor, terminated: Tensor, gamma: float, gamma_t: Optional[Tensor],
           rolling: int, time_dim: int) -> Tensor:
    return F.td1_return_estimate(_coef(gamma, gamma_t), next_state_value, reward, done, terminated, _rolling(rolling), time_dim=time_dim)


@td1_op.register_fake
def _(next_state_value, reward, done, terminated, gamma, gamma_t, rolling, time_dim):
    return torch.empty_like(next_state_value)


def _td1_setup(ctx, inputs, output):
    next_state_value, reward, done, terminated, gamma, gamma_t, rolling, time_dim = inputs
    ctx.fixed = rolling == 0 and gamma_t is not None  # as _td_setup
    if ctx.fixed:
        ctx.save_for_backward(done, terminated, gamma_t, next_state_value, reward)
    else:
        ctx.save_for_backward(done, terminated, gamma_t)
    ctx.cfg = (gamma, time_dim)
    ctx.like = (reward.shape, F._check_value_dtype(reward, next_state_value), reward.device)
    ctx.dtypes = (next_state_value.dtype, reward.dtype)


def _td1_backward(ctx, g_ret):
    gamma, time_dim = ctx.cfg
    if ctx.fixed:
        done, terminated, gamma_t, next_state_value, reward = ctx.saved_tensors
        d_nv, d_r = td_fixed_bwd_op(g_ret.to(ctx.like[1]), next_state_value, reward, done, terminated, gamma, 1.0, gamma_t, None,
                                    time_dim, True)
        return (d_nv.to(ctx.dtypes[0]), d_r.to(ctx.dtypes[1])) + (None,) * 6
    done, terminated, gamma_t = ctx.saved_tensors
    shape, dt, dev = ctx.like
    like = torch.empty(shape, dtype=dt, device=dev)
    d_nv, d_r = td_bwd_op(g_ret.to(dt), like, done, terminated, gamma, 1.0, gamma_t, None, time_dim, False, True)
    return (d_nv.to(ctx.dtypes[0]), d_r.to(ctx.dtypes[1])) + (None,) * 6


td1_op.register_autograd(_td1_backward, setup_context=_td1_setup)


# ---- V-trace --------------------------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::vtrace", mutates_args=())
def vtrace_op(log_pi: Tensor, log_mu: Tensor, state_value: Tensor, next_state_value: Tensor, reward: Tensor, done: Tensor,
              terminated: Tensor, gamma: float, gamma_t: Optional[Tensor], rho_thresh: float, c_thresh: float,
              rho_thresh_t: Optional[Tensor], c_thresh_t: Optional[Tensor], time_dim: int) -> Tuple[Tensor, Tensor]:
    return F.vtrace_advantage_estimate(_coef(gamma, gamma_t), log_pi, log_mu, state_value, next_state_value, reward, done, terminated,
                                       _coef(rho_thresh, rho_thresh_t), _coef(c_thresh, c_thresh_t), time_dim=time_dim)


def _vtrace_out_dtype(log_pi, log_mu, state_value, next_state_value, reward, gamma_t):
    vdt = F._check_value_dtype(reward, state_value, next_state_value, log_pi, log_mu)
    cdt = gamma_t.dtype if (gamma_t is not None and gamma_t.dtype.is_floating_point) else torch.get_default_dtype()
    return torch.promote_types(vdt, cdt)


@vtrace_op.register_fake
def _(log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma, gamma_t, rho_thresh, c_thresh, rho_thresh_t,
      c_thresh_t, time_dim):
    dt = _vtrace_out_dtype(log_pi, log_mu, state_value, next_state_value, reward, gamma_t)
    return (torch.empty(next_state_value.shape, dtype=dt, device=next_state_value.device),
            torch.empty(next_state_value.shape, dtype=dt, device=next_state_value.device))


@torch.library.custom_op(f"{_NS}::vtrace_bwd", mutates_args=())
def vtrace_bwd_op(g_adv: Tensor, g_vs: Tensor, vs: Tensor, log_pi: Tensor, log_mu: Tensor, state_value: Tensor, next_state_value: Tensor,
                  reward: Tensor, done: Tensor, terminated: Tensor, gamma: float, gamma_t: Optional[Tensor], rho_thresh: float,
                  c_thresh: float, rho_thresh_t: Optional[Tensor], c_thresh_t: Optional[Tensor],
                  time_dim: int, need_gamma: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    shape = next_state_value.shape
    full = lambda t: t if tuple(t.shape) == tuple(shape) else F._expand_coef(t, shape, time_dim)
    c = F._prepare_vtrace(_coef(gamma, gamma_t), full(log_pi), full(log_mu), state_value, next_state_value, reward, done, terminated,
                          _coef(rho_thresh, rho_thresh_t), _coef(c_thresh, c_thresh_t), time_dim)
    outs, d_gamma = F._vtrace_bwd(c, g_adv.reshape(shape), g_vs.reshape(shape), vs, (True,) * 5, need_gamma)
    # the per-element gamma gradient (promoted dtype) comes out of the same launch; an empty tensor when it is not wanted
    return tuple(o.view(shape) for o in outs) + (d_gamma.view(shape) if need_gamma else vs.new_empty(0),)


@vtrace_bwd_op.register_fake
def _(g_adv, g_vs, vs, log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma, gamma_t, rho_thresh, c_thresh,
      rho_thresh_t, c_thresh_t, time_dim, need_gamma):
    vdt = F._check_value_dtype(reward, state_value, next_state_value, log_pi, log_mu)
    return tuple(torch.empty(next_state_value.shape, dtype=vdt, device=next_state_value.device) for _ in range(5)) + (
        torch.empty(next_state_value.shape if need_gamma else (0,), dtype=vs.dtype, device=vs.device),)


def _vtrace_setup(ctx, inputs, output):
    (log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma, gamma_t, rho_thresh, c_thresh, rho_thresh_t,
     c_thresh_t, time_dim) = inputs
    ctx.save_for_backward(output[1], log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma_t, rho_thresh_t, c_thresh_t)
    ctx.cfg = (gamma, rho_thresh, c_thresh, time_dim)
    ctx.need_gamma = gamma_t is not None and ctx.needs_input_grad[8]


def _sum_to(g, shape):
    """Gradient of a broadcast input: summed back over the broadcast dims (what autograd does for expand)."""
    if tuple(g.shape) == tuple(shape):
        return g
    return g.sum_to_size(shape) if g.ndim == len(shape) else g.sum(tuple(range(g.ndim - len(shape)))).sum_to_size(shape)


def _vtrace_backward(ctx, g_adv, g_vs):
    vs, log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma_t, rho_t, c_t = ctx.saved_tensors
    gamma, rho_thresh, c_thresh, time_dim = ctx.cfg
    z = lambda g: torch.zeros_like(vs) if g is None else g.to(vs.dtype)
    if time_dim != -2 and (tuple(log_pi.shape) != tuple(reward.shape) or tuple(log_mu.shape) != tuple(reward.shape)):
        raise NotImplementedError("compiled backward of V-trace with broadcast log-probabilities needs time_dim=-2")
    outs = vtrace_bwd_op(z(g_adv), z(g_vs), vs, log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma, gamma_t,
                         rho_thresh, c_thresh, rho_t, c_t, time_dim, ctx.need_gamma)
    ins = (log_pi, log_mu, state_value, next_state_value, reward)
    d_gamma = _leaf_grad(outs[5], gamma_t) if ctx.need_gamma else None
    return tuple(_sum_to(o, t.shape).to(t.dtype) for o, t in zip(outs[:5], ins)) + (None, None, None, d_gamma, None, None, None, None, None)


vtrace_op.register_autograd(_vtrace_backward, setup_context=_vtrace_setup)


# ---- reward2go / standardize ------------------------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::reward2go", mutates_args=())
def reward2go_op(reward: Tensor, done: Tensor, gamma: float, time_dim: int) -> Tensor:
    return F.reward2go(reward, done, gamma, time_dim=time_dim)


@reward2go_op.register_fake
def _(reward, done, gamma, time_dim):
    return torch.empty_like(reward)


@torch.library.custom_op(f"{_NS}::standardize", mutates_args=())
def standardize_op(x: Tensor, eps: float) -> Tensor:
    return F.standardize(x, eps)


@standardize_op.register_fake
def _(x, eps):
    return torch.empty(x.shape, dtype=x.dtype, device=x.device)
