"""GPU (-m gpu): the reference's own TestValues suite (test/test_cost.py:13745-15315) restated on the HIP path:
the same cross-implementation identities at the reference's tolerance (rtol = atol = 1e-4, seeds via
torch.manual_seed(0), randn data, bernoulli(0.1) masks, done = done | terminated) -- and, since `vec == loop`
is trivially true here (one kernel serves both), every case is also compared with the CPU oracle."""
import pytest
import torch

from oracle import value_oracle as O
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(rtol=1e-4, atol=1e-4)


def _data(N, T, D=(1,), dtype=torch.float32, seed=0):
    torch.manual_seed(seed)
    done = torch.zeros(*N, T, *D, dtype=torch.bool)
    terminated = done.clone().bernoulli_(0.1)
    done = done.bernoulli_(0.1) | terminated
    reward = torch.randn(*N, T, *D, dtype=dtype)
    state_value = torch.randn(*N, T, *D, dtype=dtype)
    return done, terminated, reward, state_value


def cu(*xs):
    return tuple(x.to(DEV) if isinstance(x, torch.Tensor) else x for x in xs)


@pytest.mark.parametrize("gamma", [0.1, 0.5, 0.99])
@pytest.mark.parametrize("lmbda", [0.1, 0.5, 0.99])
@pytest.mark.parametrize("N", [(3,), (7, 3)])
@pytest.mark.parametrize("T", [200, 5, 3])
def test_tdlambda(gamma, lmbda, N, T):
    """test_cost.py:13753-13822: vec TD(lambda) adv == loop TD(lambda) adv == vec GAE when v' is the shifted v."""
    done, term, r, v = _data(N, T)
    nv = torch.cat([v[..., 1:, :], torch.randn_like(v[..., -1:, :])], -2)
    d_, t_, r_, v_, nv_ = cu(done, term, r, v, nv)
    r1 = F.vec_td_lambda_advantage_estimate(gamma, lmbda, v_, nv_, r_, done=d_, terminated=t_)
    r2 = F.td_lambda_advantage_estimate(gamma, lmbda, v_, nv_, r_, done=d_, terminated=t_)
    r3, *_ = F.vec_generalized_advantage_estimate(gamma, lmbda, v_, nv_, r_, done=d_, terminated=t_)
    torch.testing.assert_close(r3, r2, **TOL)
    torch.testing.assert_close(r1, r2, **TOL)
    torch.testing.assert_close(r2.cpu(), O.td_lambda(gamma, lmbda, nv, r, done, term) - v, **TOL)
    nv = torch.randn_like(nv)  # v' is not the next step's v (not a GAE identity any more)
    r1 = F.vec_td_lambda_advantage_estimate(gamma, lmbda, v_, nv.to(DEV), r_, done=d_, terminated=t_)
    torch.testing.assert_close(r1.cpu(), O.td_lambda(gamma, lmbda, nv, r, done, term) - v, **TOL)


@pytest.mark.parametrize("gamma", [0.1, 0.99])
@pytest.mark.parametrize("lmbda", [0.1, 0.99])
@pytest.mark.parametrize("N", [(3,), (7, 3)])
@pytest.mark.parametrize("T", [3, 100])
@pytest.mark.parametrize("feature_dim", [[5], [2, 5]])
def test_multidim_equals_per_feature(gamma, lmbda, N, T, feature_dim):
    """test_tdlambda_multi / test_td1_multi / test_gae_multidim / test_vtrace_multidim (:13831-14417): with feature
    dims [5] or [2, 5] and time_dim = -1 - len(D), every estimator equals the per-feature calls."""
    D = tuple(feature_dim)
    td = -1 - len(D)
    done, term, r, v = _data(N, T, D)
    nv = torch.randn_like(v)
    lp, lm = torch.rand_like(v).log(), torch.rand_like(v).log()
    d_, t_, r_, v_, nv_, lp_, lm_ = cu(done, term, r, v, nv, lp, lm)
    multi = dict(
        tdl=F.vec_td_lambda_advantage_estimate(gamma, lmbda, v_, nv_, r_, done=d_, terminated=t_, time_dim=td),
        td1=F.vec_td1_advantage_estimate(gamma, v_, nv_, r_, done=d_, terminated=t_, time_dim=td),
        gae=F.vec_generalized_advantage_estimate(gamma, lmbda, v_, nv_, r_, done=d_, terminated=t_, time_dim=td)[0],
        vtr=F.vtrace_advantage_estimate(gamma, lp_, lm_, v_, nv_, r_, d_, t_, time_dim=td)[0],
    )
    flat = lambda x: x.reshape(*N, T, -1)
    per = {k: [] for k in multi}
    K = flat(v_).shape[-1]
    for k in range(K):
        s = lambda x: flat(x)[..., k:k + 1]
        per["tdl"].append(F.td_lambda_advantage_estimate(gamma, lmbda, s(v_), s(nv_), s(r_), done=s(d_), terminated=s(t_)))
        per["td1"].append(F.td1_advantage_estimate(gamma, s(v_), s(nv_), s(r_), done=s(d_), terminated=s(t_)))
        per["gae"].append(F.generalized_advantage_estimate(gamma, lmbda, s(v_), s(nv_), s(r_), done=s(d_), terminated=s(t_))[0])
        per["vtr"].append(F.vtrace_advantage_estimate(gamma, s(lp_), s(lm_), s(v_), s(nv_), s(r_), s(d_), s(t_))[0])
    for name in multi:
        torch.testing.assert_close(flat(multi[name]), torch.cat(per[name], -1), **TOL)
    oadv, _ = O.gae(gamma, lmbda, v, nv, r, done, term, time_dim=td)
    torch.testing.assert_close(multi["gae"].cpu(), oadv, **TOL)
    assert torch.isfinite(multi["vtr"]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("N", [(1,), (3,), (7, 3)])
@pytest.mark.parametrize("T", [200, 5, 3])
@pytest.mark.parametrize("has_done", [True, False])
def test_gae(dtype, N, T, has_done):
    """test_cost.py:14104-14135 (vec vs loop, fp32 + fp64) and test_vtrace (:14318-14344: finite outputs)."""
    done, term, r, v = _data(N, T, dtype=dtype)
    if not has_done:
        done, term = torch.zeros_like(done), torch.zeros_like(term)
    nv = torch.randn_like(v)
    d_, t_, r_, v_, nv_ = cu(done, term, r, v, nv)
    a1, vt1 = F.vec_generalized_advantage_estimate(0.99, 0.95, v_, nv_, r_, done=d_, terminated=t_)
    a2, vt2 = F.generalized_advantage_estimate(0.99, 0.95, v_, nv_, r_, done=d_, terminated=t_)
    torch.testing.assert_close(a1, a2, **TOL)
    torch.testing.assert_close(vt1, vt2, **TOL)
    oadv, ovt = O.gae(0.99, 0.95, v, nv, r, done, term)
    torch.testing.assert_close(a2.cpu(), oadv, **TOL)
    lp = torch.rand_like(v).log().to(DEV)
    va, vs = F.vtrace_advantage_estimate(0.99, lp, lp.clone(), v_, nv_, r_, d_, t_)
    assert torch.isfinite(va).all() and torch.isfinite(vs).all()


@pytest.mark.parametrize("gamma_tensor", [False, True])
@pytest.mark.parametrize("lmbda_tensor", [False, True])
@pytest.mark.parametrize("one_element", [False, True])
def test_gae_param_as_tensor(gamma_tensor, lmbda_tensor, one_element):
    """test_cost.py:14147-14197: gamma / lmbda as floats, 1-element tensors or full tensors give the same GAE."""
    done, term, r, v = _data((3,), 30)
    nv = torch.randn_like(v)
    d_, t_, r_, v_, nv_ = cu(done, term, r, v, nv)
    mk = lambda x: torch.tensor([x], device=DEV) if one_element else torch.full_like(v_, x)
    g = mk(0.95) if gamma_tensor else 0.95
    l = mk(0.9) if lmbda_tensor else 0.9
    want = F.generalized_advantage_estimate(0.95, 0.9, v_, nv_, r_, done=d_, terminated=t_)
    for fn in (F.generalized_advantage_estimate, F.vec_generalized_advantage_estimate):
        got = fn(g, l, v_, nv_, r_, done=d_, terminated=t_)
        torch.testing.assert_close(got[0], want[0], **TOL)
        torch.testing.assert_close(got[1], want[1], **TOL)


@pytest.mark.parametrize("N", [(3,), (3, 7)])
@pytest.mark.parametrize("T", [3, 5, 50])
@pytest.mark.parametrize("rolling_gamma", [True, False, None])
def test_tdlambda_tensor_gamma(N, T, rolling_gamma):
    """test_cost.py:14425-14783: a constant gamma tensor equals the scalar gamma for the loop (both rolling modes), the
    vec form (rolling) and TD(1); rolling_gamma=False raises in the vec form with the reference's messages
    (:14792-14943)."""
    done, term, r, v = _data(N, T)
    nv = torch.randn_like(v)
    d_, t_, r_, v_, nv_ = cu(done, term, r, v, nv)
    gt = torch.full_like(v_, 0.9)
    want = F.td_lambda_advantage_estimate(0.9, 0.8, v_, nv_, r_, done=d_, terminated=t_)
    got = F.td_lambda_advantage_estimate(gt, 0.8, v_, nv_, r_, done=d_, terminated=t_, rolling_gamma=rolling_gamma)
    torch.testing.assert_close(got, want, **TOL)
    if rolling_gamma is False:
        with pytest.raises(NotImplementedError, match="When using rolling_gamma=False|The vectorized version of TD"):
            F.vec_td_lambda_advantage_estimate(gt, 0.8, v_, nv_, r_, done=d_, terminated=t_, rolling_gamma=False)
    else:
        got = F.vec_td_lambda_advantage_estimate(gt, 0.8, v_, nv_, r_, done=d_, terminated=t_, rolling_gamma=rolling_gamma)
        torch.testing.assert_close(got, want, **TOL)
    torch.testing.assert_close(F.vec_td1_advantage_estimate(gt, v_, nv_, r_, done=d_, terminated=t_),
                               F.td1_advantage_estimate(0.9, v_, nv_, r_, done=d_, terminated=t_), **TOL)
    # random gamma in [0.5, 1]: loop and vec agree when rolling (:14792-14868)
    rg = (0.5 + 0.5 * torch.rand_like(v_))
    if rolling_gamma is not False:
        torch.testing.assert_close(F.vec_td_lambda_advantage_estimate(rg, 0.8, v_, nv_, r_, done=d_, terminated=t_, rolling_gamma=rolling_gamma),
                                   F.td_lambda_advantage_estimate(rg, 0.8, v_, nv_, r_, done=d_, terminated=t_, rolling_gamma=rolling_gamma), **TOL)
        torch.testing.assert_close(F.td_lambda_return_estimate(rg, 0.8, nv_, r_, d_, t_).cpu(), O.td_lambda(rg.cpu(), 0.8, nv, r, done, term), **TOL)


@pytest.mark.parametrize("N", [(3,), (3, 7)])
@pytest.mark.parametrize("T", [3, 5, 200])
def test_successive_traj(N, T):
    """test_successive_traj_tdlambda / tdadv / gae (:14984-15135, 15232-15315): a batch with a `terminated` at T//2 - 1
    equals the two halves computed apart and concatenated."""
    torch.manual_seed(0)
    lmbda, gamma = torch.rand([]).item(), torch.rand([]).item()
    term = torch.zeros(*N, T, 1, dtype=torch.bool)
    term[..., T // 2 - 1, :] = True
    done = term.clone()
    done[..., -1, :] = True
    r, v, nv = (torch.randn(*N, T, 1) for _ in range(3))
    term_, done_, r_, v_, nv_ = cu(term, done, r, v, nv)
    h = T // 2
    cut = lambda x, a, b: x[..., a:b, :]
    for fn, args in ((F.td_lambda_return_estimate, (nv_, r_)), (F.vec_td_lambda_return_estimate, (nv_, r_))):
        whole = fn(gamma, lmbda, *args, done=done_, terminated=term_)
        parts = [fn(gamma, lmbda, *(cut(x, a, b) for x in args), done=cut(done_, a, b), terminated=cut(term_, a, b)) for a, b in ((0, h), (h, T))]
        torch.testing.assert_close(whole, torch.cat(parts, -2), **TOL)
    for fn in (F.td_lambda_advantage_estimate, F.vec_td_lambda_advantage_estimate, F.generalized_advantage_estimate, F.vec_generalized_advantage_estimate):
        whole = fn(gamma, lmbda, v_, nv_, r_, done=done_, terminated=term_)
        parts = [fn(gamma, lmbda, cut(v_, a, b), cut(nv_, a, b), cut(r_, a, b), done=cut(done_, a, b), terminated=cut(term_, a, b)) for a, b in ((0, h), (h, T))]
        if isinstance(whole, tuple):
            for k in range(2):
                torch.testing.assert_close(whole[k], torch.cat([p[k] for p in parts], -2), **TOL)
        else:
            torch.testing.assert_close(whole, torch.cat(parts, -2), **TOL)


def test_args_kwargs_timedim():
    """test_cost.py:15138-15227: time_dim is keyword-only where the reference has `*`; positional raises TypeError;
    time_dim=2 is the same as -3 on a 5-D input; transposed inputs give transposed outputs."""
    torch.manual_seed(0)
    v, nv, r = (torch.randn(2, 3, 10, 1, 1, device=DEV) for _ in range(3))
    d = torch.zeros(2, 3, 10, 1, 1, dtype=torch.bool, device=DEV).bernoulli_(0.1)
    for fn in (F.generalized_advantage_estimate, F.vec_generalized_advantage_estimate, F.td_lambda_advantage_estimate,
               F.vec_td_lambda_advantage_estimate, F.td1_advantage_estimate, F.vec_td1_advantage_estimate):
        args = (0.9, v, nv, r, d, d) if "td1" in fn.__name__ else (0.9, 0.9, v, nv, r, d, d)
        a = fn(*args, time_dim=2)
        b = fn(*args, time_dim=-3)
        a, b = (a[0], b[0]) if isinstance(a, tuple) else (a, b)
        torch.testing.assert_close(a, b, **TOL)
        tr = lambda x: x.transpose(2, -2) if isinstance(x, torch.Tensor) else x
        c = fn(*(tr(x) for x in args))  # time already at -2
        c = c[0] if isinstance(c, tuple) else c
        torch.testing.assert_close(tr(c), a, **TOL)
    with pytest.raises(TypeError):
        F.generalized_advantage_estimate(0.9, 0.9, v, nv, r, d, d, 2)
    with pytest.raises(TypeError):
        F.td_lambda_return_estimate(0.9, 0.9, nv, r, d, d, None, 2)
