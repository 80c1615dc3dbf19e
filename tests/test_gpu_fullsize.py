"""GPU (-m gpu): BASELINE.json's full-size configs.  The reference outputs are pinned by
tests/golden/checksums.json (seeded CPU inputs -> float64 sums + 16 sampled elements produced
by the reference loop in the build container); plus size-independent properties."""
import pytest
import torch

import golden_util as G
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"f32": torch.float32, "f64": torch.float64}


def _check_summary(t, ref, rtol_sum, tol_elem):
    t64 = t.detach().double().flatten().cpu()
    idx = torch.tensor(ref["idx"])
    torch.testing.assert_close(t64[idx], torch.tensor(ref["samples"], dtype=torch.float64), rtol=tol_elem, atol=tol_elem)
    n = t64.numel()
    # sums of n values each within tol: allow n**0.5 * tol absolute slack on top of the relative bound
    assert abs(float(t64.sum()) - ref["sum"]) <= rtol_sum * abs(ref["sum"]) + tol_elem * n ** 0.5
    assert abs(float((t64 * t64).sum()) - ref["sumsq"]) <= 1e-5 * ref["sumsq"]


def test_c4_gae_normalise_65536x32():
    cs = G.checksums()["C4_gae_f32"]
    g = torch.Generator().manual_seed(cs["seed"])
    shape = tuple(cs["shape"])
    r = torch.randn(shape, generator=g); v = torch.randn(shape, generator=g); nv = torch.randn(shape, generator=g)
    term = torch.rand(shape, generator=g) < 0.01
    done = term | (torch.rand(shape, generator=g) < 0.01)
    for algo in (F.ALGO_SCAN, F.ALGO_SEQ):
        adv, vt, st = F.gae_with_stats(cs["gamma"], cs["lmbda"], v.to(DEV), nv.to(DEV), r.to(DEV), done.to(DEV), term.to(DEV), algo=algo)
        _check_summary(adv, cs["advantage"], 1e-5, 1e-5)
        _check_summary(vt, cs["value_target"], 1e-5, 1e-5)
        nadv = F.normalize_(adv.clone(), st, 1e-4)
        _check_summary(nadv, cs["normalised_advantage"], 1e-4, 1e-5)
        # size-independent properties: zero mean / unit (unbiased) std, value_target - advantage == value
        assert abs(float(nadv.double().mean())) < 1e-6
        assert abs(float(nadv.double().std()) - 1.0) < 1e-5
        torch.testing.assert_close(vt - adv, v.to(DEV), rtol=0, atol=2e-6)
        # done boundaries bit-exact: A_t == delta_t = r + gamma*nt*v' - v with the reference's rounding
        gnt = (0.99 * (~term).int())
        delta = (r + gnt * nv - v)
        assert torch.equal(adv.cpu()[done], delta[done])


def test_c3_vmas_4096x100x5_expanded_masks():
    cs = G.checksums()["C3_gae_f32"]
    g = torch.Generator().manual_seed(cs["seed"])
    shape = tuple(cs["shape"])
    r = torch.randn(shape, generator=g); v = torch.randn(shape, generator=g); nv = torch.randn(shape, generator=g)
    term = (torch.rand((4096, 100, 1, 1), generator=g) < 0.01)
    trunc = (torch.rand((4096, 100, 1, 1), generator=g) < 0.01)
    term_d = term.to(DEV).expand(shape)            # stride-0 views, as DoneTransform produces
    done_d = (term | trunc).to(DEV).expand(shape)
    adv, vt = F.generalized_advantage_estimate(cs["gamma"], cs["lmbda"], v.to(DEV), nv.to(DEV), r.to(DEV), done_d, term_d, time_dim=-3)
    _check_summary(adv, cs["advantage"], 1e-5, 1e-5)
    _check_summary(vt, cs["value_target"], 1e-5, 1e-5)


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_c5_long_horizon_1024x10000(dt):
    cs = G.checksums()
    g = torch.Generator().manual_seed(0)
    shape = (1024, 10000, 1)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(DT[dt])
    r, v, nv = mk(), mk(), mk()
    gamma = (0.9 + 0.1 * torch.rand(shape, generator=g, dtype=torch.float32)).to(DT[dt])
    term = torch.rand(shape, generator=g) < 0.001
    done = term | (torch.rand(shape, generator=g) < 0.001)
    log_pi = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(DT[dt])  # f64 log: f32 log differs by an ulp across hosts
    log_mu = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(DT[dt])
    te = 1e-5 if dt == "f32" else 1e-12
    c = lambda t: t.to(DEV)
    ret = F.td_lambda_return_estimate(c(gamma), 0.95, c(nv), c(r), c(done), c(term))
    assert "scan_long" in F.last_kernel()
    _check_summary(ret, cs[f"C5_tdlambda_{dt}"]["returns"], te, te)
    adv, vs = F.vtrace_advantage_estimate(c(gamma), c(log_pi), c(log_mu), c(v), c(nv), c(r), c(done), c(term))
    _check_summary(adv, cs[f"C5_vtrace_{dt}"]["advantage"], max(te, 1e-11), max(te, 1e-11))
    _check_summary(vs, cs[f"C5_vtrace_{dt}"]["vs"], max(te, 1e-11), max(te, 1e-11))
    adv, vt = F.generalized_advantage_estimate(c(gamma), 0.95, c(v), c(nv), c(r), c(done), c(term))
    _check_summary(adv, cs[f"C5_gae_{dt}"]["advantage"], te, te)
    _check_summary(vt, cs[f"C5_gae_{dt}"]["value_target"], te, te)
