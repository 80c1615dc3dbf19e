"""GPU (-m gpu): torch.compile(fullgraph=True) over the estimators (compile-friendly custom ops with fake
implementations, torchrl_amd/objectives/value/_ops.py).  Compiled results must equal the eager ones bit for bit (same
kernels), gradients included."""
import pytest
import torch

from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(shape=(64, 32, 1), seed=0, grad=False):
    g = torch.Generator().manual_seed(seed)
    mk = lambda: torch.randn(shape, generator=g).to(DEV).requires_grad_(grad)
    r, v, nv = mk(), mk(), mk()
    term = (torch.rand(shape, generator=g) < 0.05).to(DEV)
    done = term | (torch.rand(shape, generator=g) < 0.05).to(DEV)
    lp = (0.3 * torch.randn(shape, generator=g)).to(DEV).requires_grad_(grad)
    lmu = (0.3 * torch.randn(shape, generator=g)).to(DEV).requires_grad_(grad)
    return r, v, nv, done, term, lp, lmu


def test_ppo_style_function_compiles_fullgraph():
    """advantage (normalised GAE, no grad) -> clipped surrogate + value loss, as one graph."""
    r, v, nv, done, term, lp, lmu = _inputs()
    logits = torch.randn(64, 32, 1, device=DEV, requires_grad=True)

    def ppo_step(logits, v, nv, r, done, term, old_lp):
        with torch.no_grad():
            adv, target = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
        ratio = (logits - old_lp).exp()
        surr = torch.minimum(ratio * adv, ratio.clamp(0.8, 1.2) * adv)
        return -surr.mean() + 0.5 * (v - target).pow(2).mean()

    eager = ppo_step(logits, v, nv, r, done, term, lp.detach())
    compiled = torch.compile(ppo_step, fullgraph=True)(logits, v, nv, r, done, term, lp.detach())
    torch.testing.assert_close(compiled, eager, rtol=1e-6, atol=1e-6)
    compiled.backward()
    assert logits.grad is not None and torch.isfinite(logits.grad).all()


def test_every_estimator_compiles_and_matches_eager():
    r, v, nv, done, term, lp, lmu = _inputs((16, 40, 3), seed=1)
    gam = (0.9 + 0.1 * torch.rand(16, 40, 3)).to(DEV)

    def everything(v, nv, r, done, term, lp, lmu, gam):
        a1, t1 = F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)
        a2, t2 = F.vec_generalized_advantage_estimate(gam, 0.95, v, nv, r, done)
        r1 = F.td_lambda_return_estimate(0.99, 0.9, nv, r, done, term)
        r2 = F.vec_td_lambda_return_estimate(gam, 0.9, nv, r, done, term)
        r3 = F.td_lambda_advantage_estimate(0.99, 0.9, v, nv, r, done, term)
        r4 = F.td1_return_estimate(0.97, nv, r, done, term)
        r5 = F.td0_advantage_estimate(0.97, v, nv, r, done, term)
        a3, vs = F.vtrace_advantage_estimate(0.99, lp, lmu, v, nv, r, done, term, 0.9, 0.8)
        r6 = F.reward2go(r, done, 0.9)
        r7 = F.standardize(a1, 1e-6)
        return a1, t1, a2, t2, r1, r2, r3, r4, r5, a3, vs, r6, r7

    eager = everything(v, nv, r, done, term, lp, lmu, gam)
    compiled = torch.compile(everything, fullgraph=True)(v, nv, r, done, term, lp, lmu, gam)
    for k, (a, b) in enumerate(zip(compiled, eager)):
        assert a.dtype == b.dtype and a.shape == b.shape, k
        if k == 8:  # td0 is an elementwise torch-op composition: inductor contracts it into FMAs
            torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(a, b), (k, (a - b).abs().max().item())


def test_compiled_gradients_match_eager():
    def losses(v, nv, r, done, term, lp, lmu):
        adv, vt = F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)
        ret = F.td_lambda_return_estimate(0.99, 0.9, nv, r, done, term)
        ret1 = F.td1_return_estimate(0.97, nv, r, done, term)
        a, vs = F.vtrace_advantage_estimate(0.99, lp, lmu, v, nv, r, done, term, 0.9, 0.8)
        return (adv * vt).sum() + ret.pow(2).sum() + ret1.sum() + (a * vs).sum()

    grads = []
    for fn in (losses, torch.compile(losses, fullgraph=True)):
        r, v, nv, done, term, lp, lmu = _inputs((8, 50, 1), seed=2, grad=True)
        fn(v, nv, r, done, term, lp, lmu).backward()
        grads.append([t.grad.clone() for t in (r, v, nv, lp, lmu)])
    for a, b in zip(*grads):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)


def test_compiled_nonrolling_td_gradients_match_eager():
    """rolling_gamma=False under torch.compile: forward and backward (trl_td_fixed_bwd through its own custom op), as eager
    (the compiled op used to raise as soon as an input required grad)."""
    def loss(nv, r, gam, done, term):
        ret = F.td_lambda_return_estimate(gam, 0.9, nv, r, done, term, rolling_gamma=False)
        ret1 = F.td1_return_estimate(gam, nv, r, done, term, rolling_gamma=False)
        return ret.pow(2).sum() + (ret1 * ret).sum()

    grads = []
    for fn in (loss, torch.compile(loss, fullgraph=True)):
        r, v, nv, done, term, lp, lmu = _inputs((6, 30, 1), seed=4, grad=True)
        gam = 0.9 + 0.1 * torch.rand((6, 30, 1), generator=torch.Generator().manual_seed(1)).to(DEV)
        fn(nv, r, gam, done, term).backward()
        grads.append([nv.grad.clone(), r.grad.clone()])
    for a, b in zip(*grads):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("form", ["full", "zero_dim", "per_step"])
def test_compiled_gradients_to_coefficient_tensors_match_eager(form):
    """gamma / lmbda tensors that require grad under torch.compile (round 3; eager-only before): GAE, TD(lambda) with rolling
    gamma, V-trace's gamma -- full-shape, 0-dim and broadcast ([T, 1]) leaves."""
    shape = (6, 40, 1)

    def leaves(seed):
        g = torch.Generator().manual_seed(seed)
        if form == "full":
            mk = lambda lo: (lo + 0.09 * torch.rand(shape, generator=g)).to(DEV).requires_grad_(True)
        elif form == "zero_dim":
            mk = lambda lo: torch.tensor(lo + 0.05, device=DEV, requires_grad=True)
        else:
            mk = lambda lo: (lo + 0.09 * torch.rand(shape[1:], generator=g)).to(DEV).requires_grad_(True)
        return mk(0.9), mk(0.85)

    def loss(v, nv, r, done, term, lp, lmu, gam, lam):
        adv, vt = F.generalized_advantage_estimate(gam, lam, v, nv, r, done, term)
        a, vs = F.vtrace_advantage_estimate(gam, lp, lmu, v, nv, r, done, term)
        out = (adv * vt).sum() + (a * vs).sum()
        if form != "per_step":  # the loop variant of TD(lambda) takes scalars or tensors of the inputs' shape only (as the reference)
            out = out + F.td_lambda_return_estimate(gam, lam, nv, r, done, term).pow(2).sum()
        return out

    grads = []
    for fn in (loss, torch.compile(loss, fullgraph=True)):
        r, v, nv, done, term, lp, lmu = _inputs(shape, seed=6, grad=True)
        gam, lam = leaves(7)
        fn(v, nv, r, done, term, lp, lmu, gam, lam).backward()
        grads.append([t.grad.clone() for t in (gam, lam, v, nv, r, lp)])
    for a, b in zip(*grads):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)
