"""Generate the golden fixtures under tests/golden/ from the imported reference.

Runs ONLY in the build container (needs /root/reference; see tools/ref_import.py).  Every
output stored here was produced by the reference's own functions
(torchrl/objectives/value/functional.py: generalized_advantage_estimate,
vec_generalized_advantage_estimate, td_lambda_return_estimate, vtrace_advantage_estimate) or,
for the normalisation cases, by the reference's expression
``(adv - adv.mean()) / adv.std().clamp_min(eps)`` (advantages.py:1459-1463, 1828-1832).

Fixtures are data only: ``<family>.npz`` (inputs + expected outputs, loadable with
``numpy.load(allow_pickle=False)``) and ``manifest.json`` (per-case parameters: how gamma /
lmbda were passed, time_dim, default dtype, ...).  ``checksums.json`` pins the reference
outputs at the full BASELINE sizes through seeds + float64 sums + sampled elements.

    python tools/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DT = {"f32": torch.float32, "f64": torch.float64}


def make_masks(shape, regime, gen, expand_dim=None):
    """done/terminated masks for the regimes of SURVEY.md Appendix B."""
    if expand_dim is not None:
        # per-env masks expanded (stride 0) over the agent dim, as DoneTransform does
        base = list(shape)
        base[expand_dim] = 1
        done, term = make_masks(tuple(base), regime, gen)
        return done.expand(shape), term.expand(shape)
    z = torch.zeros(shape, dtype=torch.bool)
    if regime == "none":
        return z.clone(), z.clone()
    if regime == "done_eq_term":
        t = torch.rand(shape, generator=gen) < 0.1
        return t.clone(), t
    if regime == "done_sup_term":
        t = torch.rand(shape, generator=gen) < 0.07
        tr = torch.rand(shape, generator=gen) < 0.07
        return t | tr, t
    if regime == "all_done":
        t = torch.rand(shape, generator=gen) < 0.5
        return ~z, t
    if regime == "done_last":
        d = z.clone()
        d.select(-2 if len(shape) > 1 else -1, shape[-2 if len(shape) > 1 else -1] - 1).fill_(True)
        return d, d.clone()
    raise ValueError(regime)


def coef_form(form, value, shape, gen):
    """Build gamma/lmbda in the requested form; returns (object passed to the reference, manifest entry, array or None)."""
    if form == "float":
        return value, {"kind": "float", "value": value}, None
    if form == "t0_f32":
        t = torch.tensor(value, dtype=torch.float32)
    elif form == "t0_f64":
        t = torch.tensor(value, dtype=torch.float64)
    elif form == "t1_f32":
        t = torch.tensor([value], dtype=torch.float32)
    elif form == "full_f32":
        t = (value - 0.1 * torch.rand(shape, generator=gen, dtype=torch.float64)).to(torch.float32)
    elif form == "full_f64":
        t = value - 0.1 * torch.rand(shape, generator=gen, dtype=torch.float64)
    else:
        raise ValueError(form)
    return t, {"kind": "tensor", "form": form}, t.numpy()


class Family:
    def __init__(self, name):
        self.name = name
        self.arrays = {}
        self.cases = []

    def add(self, meta, arrays):
        cid = f"c{len(self.cases):03d}"
        meta = dict(meta, id=cid)
        for k, v in arrays.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.detach().contiguous().numpy()
            self.arrays[f"{cid}.{k}"] = np.ascontiguousarray(v)
        self.cases.append(meta)

    def save(self):
        np.savez_compressed(os.path.join(OUT, self.name + ".npz"), **self.arrays)
        return self.cases


def gen_gae(F):
    fam = Family("gae")
    gen = torch.Generator().manual_seed(1234)
    specs = []
    # (shape, time_dim, expand_dim)
    shapes = [((1, 256, 1), -2, None), ((8, 128, 1), -2, None), ((64, 32, 1), -2, None),
              ((6, 20, 5, 1), -3, 2), ((6, 20, 5, 1), 1, None), ((3, 7, 2, 5), -3, None),
              ((4, 1, 1), -2, None), ((17,), -1, None), ((5, 9), -1, None)]
    for shape, td, ex in shapes:
        for dt in ("f32", "f64"):
            specs.append(dict(shape=shape, time_dim=td, expand=ex, dtype=dt, gl=(0.99, 0.95),
                              regime="done_sup_term", gform="float", lform="float", default="f32"))
    for regime in ("none", "done_eq_term", "all_done", "done_last"):
        for dt in ("f32", "f64"):
            specs.append(dict(shape=(8, 33, 1), time_dim=-2, expand=None, dtype=dt, gl=(0.9, 0.9),
                              regime=regime, gform="float", lform="float", default="f32"))
    for gl in ((0.5, 0.1), (1.0, 1.0), (0.0, 0.95)):
        specs.append(dict(shape=(4, 40, 1), time_dim=-2, expand=None, dtype="f32", gl=gl,
                          regime="done_sup_term", gform="float", lform="float", default="f32"))
    for gform, lform in (("t0_f32", "t0_f32"), ("t0_f64", "float"), ("t1_f32", "float"), ("full_f32", "float"),
                         ("full_f64", "float"), ("float", "full_f32"), ("full_f32", "full_f32"), ("full_f64", "full_f64")):
        for dt in ("f32", "f64"):
            specs.append(dict(shape=(5, 24, 1), time_dim=-2, expand=None, dtype=dt, gl=(0.99, 0.95),
                              regime="done_sup_term", gform=gform, lform=lform, default="f32"))
    for dt in ("f32", "f64"):  # reference tests' dtype_fixture: default dtype double
        specs.append(dict(shape=(7, 30, 3), time_dim=-2, expand=None, dtype=dt, gl=(0.99, 0.95),
                          regime="done_sup_term", gform="float", lform="float", default="f64"))
    # terminated=None
    specs.append(dict(shape=(4, 30, 1), time_dim=-2, expand=None, dtype="f32", gl=(0.99, 0.95),
                      regime="done_eq_term", gform="float", lform="float", default="f32", no_term=True))

    for s in specs:
        torch.set_default_dtype(DT[s["default"]])
        try:
            shape, dt = s["shape"], DT[s["dtype"]]
            r = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
            v = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
            nv = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
            done, term = make_masks(shape, s["regime"], gen, s["expand"])
            g, gm, ga = coef_form(s["gform"], s["gl"][0], shape, gen)
            l, lm, la = coef_form(s["lform"], s["gl"][1], shape, gen)
            kw = {} if s["time_dim"] == -2 and len(shape) > 1 else {"time_dim": s["time_dim"]}
            tm = None if s.get("no_term") else term
            adv, vt = F.generalized_advantage_estimate(g, l, v, nv, r, done, tm, **kw)
            arrays = dict(reward=r, state_value=v, next_state_value=nv, done=done, terminated=tm,
                          gamma=ga, lmbda=la, advantage=adv, value_target=vt)
            try:
                vadv, vvt = F.vec_generalized_advantage_estimate(g, l, v, nv, r, done, tm, **kw)
                arrays.update(vec_advantage=vadv, vec_value_target=vvt)
            except Exception as e:  # documented: the vec path does not support every form
                s = dict(s, vec_error=type(e).__name__)
            meta = dict(shape=list(shape), time_dim=s["time_dim"], dtype=s["dtype"], default_dtype=s["default"],
                        regime=s["regime"], expanded_masks=s["expand"] is not None, gamma=gm, lmbda=lm,
                        terminated_none=bool(s.get("no_term")), vec_error=s.get("vec_error"))
            fam.add(meta, arrays)
        finally:
            torch.set_default_dtype(torch.float32)
    return fam.save()


def gen_tdlambda(F):
    fam = Family("td_lambda")
    gen = torch.Generator().manual_seed(4321)
    specs = []
    for shape, td in (((1, 200, 1), -2), ((8, 64, 1), -2), ((6, 20, 5, 1), -3), ((3, 7, 2, 5), -3), ((4, 1, 1), -2), ((11,), -1)):
        for dt in ("f32", "f64"):
            specs.append(dict(shape=shape, time_dim=td, dtype=dt, gl=(0.99, 0.95), regime="done_sup_term",
                              gform="float", lform="float", rolling=None))
    for regime in ("none", "done_eq_term", "all_done", "done_last"):
        specs.append(dict(shape=(6, 21, 1), time_dim=-2, dtype="f32", gl=(0.9, 0.8), regime=regime,
                          gform="float", lform="float", rolling=None))
    for gform, lform in (("t0_f32", "t0_f32"), ("full_f32", "float"), ("full_f64", "float"), ("float", "full_f32"),
                         ("full_f32", "full_f32"), ("full_f64", "full_f64")):
        for dt in ("f32", "f64"):
            if (gform.endswith("f64") or lform.endswith("f64")) and dt == "f32":
                continue
            if (gform == "full_f32" or lform == "full_f32") and dt == "f64" and gform != lform and "full" in gform + lform:
                pass
            specs.append(dict(shape=(5, 24, 1), time_dim=-2, dtype=dt, gl=(0.99, 0.95), regime="done_sup_term",
                              gform=gform, lform=lform, rolling=None))
    for gform, lform in (("full_f32", "float"), ("full_f32", "full_f32"), ("float", "full_f32")):
        specs.append(dict(shape=(3, 12, 1), time_dim=-2, dtype="f32", gl=(0.95, 0.9), regime="done_sup_term",
                          gform=gform, lform=lform, rolling=False))
    specs.append(dict(shape=(3, 12, 1), time_dim=-2, dtype="f64", gl=(0.95, 0.9), regime="done_sup_term",
                      gform="full_f64", lform="float", rolling=False))
    for s in specs:
        shape, dt = s["shape"], DT[s["dtype"]]
        r = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        nv = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        done, term = make_masks(shape, s["regime"], gen)
        g, gm, ga = coef_form(s["gform"], s["gl"][0], shape, gen)
        l, lm, la = coef_form(s["lform"], s["gl"][1], shape, gen)
        kw = {} if s["time_dim"] == -2 and len(shape) > 1 else {"time_dim": s["time_dim"]}
        ret = F.td_lambda_return_estimate(g, l, nv, r, done, term, s["rolling"], **kw)
        meta = dict(shape=list(shape), time_dim=s["time_dim"], dtype=s["dtype"], regime=s["regime"],
                    gamma=gm, lmbda=lm, rolling_gamma=s["rolling"])
        fam.add(meta, dict(reward=r, next_state_value=nv, done=done, terminated=term, gamma=ga, lmbda=la, returns=ret))
    return fam.save()


def gen_vtrace(F):
    fam = Family("vtrace")
    gen = torch.Generator().manual_seed(777)
    specs = []
    for shape, td in (((1, 200, 1), -2), ((8, 64, 1), -2), ((6, 20, 5), -2), ((3, 7, 5), -2), ((3, 7, 2, 5), -3), ((4, 1, 1), -2)):
        for dt in ("f32", "f64"):
            specs.append(dict(shape=shape, time_dim=td, dtype=dt, gamma=0.99, regime="done_sup_term", gform="float",
                              rho=1.0, c=1.0, no_term=False))
    for rho, c in ((0.5, 1.0), (1.0, 0.5), (0.7, 0.3), ("t0", "t0")):
        specs.append(dict(shape=(5, 30, 1), time_dim=-2, dtype="f32", gamma=0.97, regime="done_sup_term", gform="float",
                          rho=rho, c=c, no_term=False))
    for regime in ("none", "done_eq_term", "all_done", "done_last"):
        specs.append(dict(shape=(6, 21, 1), time_dim=-2, dtype="f32", gamma=0.9, regime=regime, gform="float",
                          rho=1.0, c=1.0, no_term=False))
    specs.append(dict(shape=(6, 21, 1), time_dim=-2, dtype="f32", gamma=0.9, regime="done_sup_term", gform="float",
                      rho=1.0, c=1.0, no_term=True))
    for gform in ("t0_f32", "full_f32", "full_f64"):
        for dt in ("f32", "f64"):
            if gform == "full_f64" and dt == "f32":
                continue
            specs.append(dict(shape=(5, 24, 1), time_dim=-2, dtype=dt, gamma=0.99, regime="done_sup_term", gform=gform,
                              rho=1.0, c=1.0, no_term=False))
    for s in specs:
        shape, dt = s["shape"], DT[s["dtype"]]
        r = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        v = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        nv = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        # as test_cost.py:14329-14330: log-probabilities are logs of uniform samples
        log_pi = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-6).log().to(dt)
        log_mu = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-6).log().to(dt)
        done, term = make_masks(shape, s["regime"], gen)
        g, gm, ga = coef_form(s["gform"], s["gamma"], shape, gen)
        rho = torch.tensor(0.6) if s["rho"] == "t0" else s["rho"]
        c = torch.tensor(0.8) if s["c"] == "t0" else s["c"]
        kw = {} if s["time_dim"] == -2 else {"time_dim": s["time_dim"]}
        tm = None if s["no_term"] else term
        adv, vs = F.vtrace_advantage_estimate(g, log_pi, log_mu, v, nv, r, done, tm, rho, c, **kw)
        meta = dict(shape=list(shape), time_dim=s["time_dim"], dtype=s["dtype"], regime=s["regime"], gamma=gm,
                    rho_thresh=float(rho), c_thresh=float(c), thresh_as_tensor=s["rho"] == "t0",
                    terminated_none=s["no_term"])
        fam.add(meta, dict(log_pi=log_pi, log_mu=log_mu, state_value=v, next_state_value=nv, reward=r, done=done,
                           terminated=tm, gamma=ga, advantage=adv, vs=vs))
    # truncated-not-terminated boundary known answer
    r = torch.ones(1, 4, 1); v = torch.zeros(1, 4, 1); nv = torch.full((1, 4, 1), 10.0)
    lp = torch.zeros(1, 4, 1); d = torch.tensor([False, True, False, False]).view(1, 4, 1)
    for tm in (torch.zeros_like(d), None):
        adv, vs = F.vtrace_advantage_estimate(0.5, lp, lp, v, nv, r, d, tm)
        fam.add(dict(shape=[1, 4, 1], time_dim=-2, dtype="f32", regime="A8", gamma={"kind": "float", "value": 0.5},
                     rho_thresh=1.0, c_thresh=1.0, thresh_as_tensor=False, terminated_none=tm is None),
                dict(log_pi=lp, log_mu=lp, state_value=v, next_state_value=nv, reward=r, done=d, terminated=tm,
                     advantage=adv, vs=vs))
    return fam.save()


def gen_normalise():
    fam = Family("normalise")
    gen = torch.Generator().manual_seed(99)
    # (shape, dtype, scale, offset); the 1e-7 / zero-offset case has std < eps for every eps (clamp branch)
    for shape, dt, scale, off in (((64, 32, 1), "f32", 1.0, 0.3), ((8, 128, 1), "f32", 3.0, 0.3),
                                  ((16, 10, 5, 1), "f32", 1e-3, 0.3), ((16, 10, 5, 1), "f32", 1e-7, 0.0),
                                  ((64, 32, 1), "f64", 1.0, 0.3), ((1, 2, 1), "f32", 1.0, 0.3)):
        x = (torch.randn(shape, generator=gen, dtype=torch.float64) * scale + off).to(DT[dt])
        for eps in (1e-4, 1e-5, 1e-6):
            # advantages.py:1459-1463 / 1828-1832; _utils.py:901-912 (_standardize, eps 1e-6 via clamp_min)
            out = (x - x.mean()) / x.std().clamp_min(eps)
            fam.add(dict(shape=list(shape), dtype=dt, eps=eps), dict(x=x, out=out))
    return fam.save()


def gen_errors(F):
    """Exception type + message for the error behaviours of SURVEY.md 8(b)."""
    out = []
    r = torch.randn(3, 8, 1); d = torch.zeros(3, 8, 1, dtype=torch.bool)

    def rec(name, fn):
        try:
            fn()
            out.append(dict(name=name, type=None, message=None))
        except Exception as e:
            out.append(dict(name=name, type=type(e).__name__, message=str(e)))

    rec("gae_shape_mismatch", lambda: F.generalized_advantage_estimate(0.9, 0.9, r, r, r[:, :4], d))
    rec("gae_positional_time_dim", lambda: F.generalized_advantage_estimate(0.9, 0.9, r, r, r, d, d, -2))
    rec("vec_gae_shape_mismatch", lambda: F.vec_generalized_advantage_estimate(0.9, 0.9, r, r, r[:, :4], d))
    rec("gae_1d_time_dim_m2", lambda: F.generalized_advantage_estimate(0.9, 0.9, r[0, :, 0], r[0, :, 0], r[0, :, 0], d[0, :, 0]))
    rec("gae_1d_time_dim_m3", lambda: F.generalized_advantage_estimate(0.9, 0.9, r[0, :, 0], r[0, :, 0], r[0, :, 0], d[0, :, 0], time_dim=-3))
    rec("gae_bad_time_dim", lambda: F.generalized_advantage_estimate(0.9, 0.9, r, r, r, d, time_dim=-4))
    rec("tdl_shape_mismatch", lambda: F.td_lambda_return_estimate(0.9, 0.9, r, r[:, :4], d))
    rec("tdl_positional_time_dim", lambda: F.td_lambda_return_estimate(0.9, 0.9, r, r, d, d, None, -2))
    rec("tdl_not_rolling_scalar", lambda: F.td_lambda_return_estimate(0.9, 0.9, r, r, d, rolling_gamma=False))
    g = torch.full_like(r, 0.9)
    tm = d.clone(); tm[:, 3] = True
    rec("vec_tdl_not_rolling_after_term", lambda: F.vec_td_lambda_return_estimate(g, 0.9, r, r, tm, tm, rolling_gamma=False))
    rec("vec_tdl_not_rolling", lambda: F.vec_td_lambda_return_estimate(g, 0.9, r, r, d, d, rolling_gamma=False))
    rec("vtrace_shape_mismatch", lambda: F.vtrace_advantage_estimate(0.9, r, r, r, r, r[:, :4], d))
    rec("tdl_adv_shape_mismatch", lambda: F.td_lambda_advantage_estimate(0.9, 0.9, r[:, :4], r, r, d))
    return out


def gen_checksums(F):
    """Reference outputs at the full BASELINE sizes, pinned by seeds + sums + samples."""
    out = {}

    def summarize(t):
        t64 = t.detach().double().flatten()
        idx = torch.linspace(0, t64.numel() - 1, 16).long()
        return dict(sum=float(t64.sum()), sumsq=float((t64 * t64).sum()), idx=idx.tolist(), samples=[float(x) for x in t64[idx]])

    # C4 [65536, 32, 1] fp32, gamma .99 lambda .95 (+ normalised advantage, eps 1e-4)
    g = torch.Generator().manual_seed(0)
    shape = (65536, 32, 1)
    r = torch.randn(shape, generator=g); v = torch.randn(shape, generator=g); nv = torch.randn(shape, generator=g)
    term = torch.rand(shape, generator=g) < 0.01
    done = term | (torch.rand(shape, generator=g) < 0.01)
    adv, vt = F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)
    nadv = (adv - adv.mean()) / adv.std().clamp_min(1e-4)
    out["C4_gae_f32"] = dict(seed=0, shape=list(shape), gamma=0.99, lmbda=0.95, advantage=summarize(adv),
                             value_target=summarize(vt), normalised_advantage=summarize(nadv),
                             mean=float(adv.mean()), std=float(adv.std()))
    # C3 [4096, 100, 5, 1] time_dim=-3, expanded per-env masks, gamma = lambda = 0.9
    g = torch.Generator().manual_seed(0)
    shape = (4096, 100, 5, 1)
    r = torch.randn(shape, generator=g); v = torch.randn(shape, generator=g); nv = torch.randn(shape, generator=g)
    term = (torch.rand((4096, 100, 1, 1), generator=g) < 0.01).expand(shape)
    done = term | (torch.rand((4096, 100, 1, 1), generator=g) < 0.01).expand(shape)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, v, nv, r, done, term, time_dim=-3)
    out["C3_gae_f32"] = dict(seed=0, shape=list(shape), gamma=0.9, lmbda=0.9, time_dim=-3,
                             advantage=summarize(adv), value_target=summarize(vt))
    # C5 [1024, 10000, 1] per-element gamma, lambda .95: TD(lambda) and V-trace, fp32 and fp64
    for dt in ("f32", "f64"):
        g = torch.Generator().manual_seed(0)
        shape = (1024, 10000, 1)
        mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(DT[dt])
        r, v, nv = mk(), mk(), mk()
        gamma = (0.9 + 0.1 * torch.rand(shape, generator=g, dtype=torch.float32)).to(DT[dt])
        term = torch.rand(shape, generator=g) < 0.001
        done = term | (torch.rand(shape, generator=g) < 0.001)
        log_pi = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(DT[dt])  # f64 log: f32 log differs by an ulp across hosts
        log_mu = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(DT[dt])
        ret = F.td_lambda_return_estimate(gamma, 0.95, nv, r, done, term)
        out[f"C5_tdlambda_{dt}"] = dict(seed=0, shape=list(shape), lmbda=0.95, returns=summarize(ret))
        adv, vs = F.vtrace_advantage_estimate(gamma, log_pi, log_mu, v, nv, r, done, term)
        out[f"C5_vtrace_{dt}"] = dict(seed=0, shape=list(shape), advantage=summarize(adv), vs=summarize(vs))
        adv, vt = F.generalized_advantage_estimate(gamma, 0.95, v, nv, r, done, term)
        out[f"C5_gae_{dt}"] = dict(seed=0, shape=list(shape), lmbda=0.95, advantage=summarize(adv), value_target=summarize(vt))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    F = ref_import.load_functional()
    manifest = dict(
        generator="tools/gen_golden.py",
        reference="AsbjornLorenzen/torchrl v0.7.2 torchrl/objectives/value/functional.py",
        torch=torch.__version__,
        gae=gen_gae(F), td_lambda=gen_tdlambda(F), vtrace=gen_vtrace(F), normalise=gen_normalise(),
        errors=gen_errors(F),
    )
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    if "--no-checksums" not in sys.argv:
        with open(os.path.join(OUT, "checksums.json"), "w") as f:
            json.dump(gen_checksums(F), f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
