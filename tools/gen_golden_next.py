"""Golden fixtures for the "next" rows and the backward passes, from the imported reference.

Runs ONLY in the build container (needs /root/reference; tools/ref_import.py).  Every expected output is produced by
the reference's own functions (torchrl/objectives/value/functional.py):

  td.npz            td0_return_estimate / td0_advantage_estimate (:378-456), td1_return_estimate rolling and
                    rolling_gamma=False (:464-562), td1_advantage_estimate (:565-638), vec_td1_return_estimate (:641-700),
                    reward2go (:1358-1432; 2-D, 3-D, 4-D, the docstring case and test_cost.py:16287-16309's closed form)
  grads.npz         gradients of GAE / TD(lambda) / TD(1) / V-trace obtained by torch autograd THROUGH the reference's
                    loops (what differentiable=True estimators give, test/test_cost.py:15521-15584), w.r.t. reward,
                    state_value, next_state_value, log-probabilities and gamma / lmbda tensors; the scalar loss is
                    sum(out_k * W_k) with the random weights W_k stored in the fixture
  vtrace_thresh.npz vtrace_advantage_estimate with broadcastable threshold TENSORS (:1311-1330)

Fixtures are data only (npz arrays + manifest_next.json).

    python tools/gen_golden_next.py
"""
from __future__ import annotations

import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
from gen_golden import DT, OUT, Family, coef_form, make_masks  # noqa: E402


def gen_td(F):
    fam = Family("td")
    gen = torch.Generator().manual_seed(2468)

    def data(shape, dt, regime="done_sup_term"):
        r = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        v = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        nv = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        done, term = make_masks(shape, regime, gen)
        return r, v, nv, done, term

    # TD(0): elementwise
    for shape in ((4, 9, 1), (3, 7, 2, 5), (11,)):
        for dts in ("f32", "f64"):
            r, v, nv, done, term = data(shape, DT[dts])
            ret = F.td0_return_estimate(0.97, nv, r, term)
            adv = F.td0_advantage_estimate(0.97, v, nv, r, done, term)
            fam.add(dict(fn="td0", shape=list(shape), dtype=dts, gamma=0.97),
                    dict(reward=r, state_value=v, next_state_value=nv, done=done, terminated=term, returns=ret, advantage=adv))
    # TD(1): rolling (scalar / 0-dim / full gamma), rolling_gamma=False, regimes, time dims, masks where terminated is
    # NOT a subset of done (the loops treat them independently)
    specs = []
    for shape, td in (((1, 64, 1), -2), ((8, 33, 1), -2), ((6, 20, 5, 1), -3), ((3, 7, 2, 5), -3), ((4, 1, 1), -2), ((13,), -1)):
        for dts in ("f32", "f64"):
            specs.append(dict(shape=shape, time_dim=td, dtype=dts, gform="float", rolling=None, regime="done_sup_term"))
    for regime in ("none", "done_eq_term", "all_done", "done_last", "term_free"):
        specs.append(dict(shape=(6, 21, 1), time_dim=-2, dtype="f32", gform="float", rolling=None, regime=regime))
    for gform in ("t0_f32", "full_f32", "full_f64"):
        for dts in ("f32", "f64"):
            for rolling in (None, True, False):
                if gform == "t0_f32" and rolling is False:
                    continue
                specs.append(dict(shape=(3, 12, 1), time_dim=-2, dtype=dts, gform=gform, rolling=rolling, regime="done_sup_term"))
    specs.append(dict(shape=(4, 10, 3), time_dim=-2, dtype="f32", gform="full_f32", rolling=False, regime="done_sup_term"))
    specs.append(dict(shape=(2, 9, 4, 1), time_dim=-3, dtype="f32", gform="full_f32", rolling=False, regime="term_free"))
    for s in specs:
        shape, dt = s["shape"], DT[s["dtype"]]
        regime = s["regime"]
        r, v, nv, done, term = data(shape, dt, "done_sup_term" if regime == "term_free" else regime)
        if regime == "term_free":  # terminated drawn independently of done
            term = torch.rand(shape, generator=gen) < 0.15
        g, gm, ga = coef_form(s["gform"], 0.95, shape, gen)
        kw = {} if s["time_dim"] == -2 and len(shape) > 1 else {"time_dim": s["time_dim"]}
        ret = F.td1_return_estimate(g, nv, r, done, term, s["rolling"], **kw)
        arrays = dict(reward=r, state_value=v, next_state_value=nv, done=done, terminated=term, gamma=ga, returns=ret)
        if len(shape) > 1:
            arrays["advantage"] = F.td1_advantage_estimate(g, v, nv, r, done, term, s["rolling"], **({"time_dim": s["time_dim"]}))
        vec_error = None
        if s["rolling"] is not False and regime != "term_free":
            try:
                arrays["vec_returns"] = F.vec_td1_return_estimate(g, nv, r, done, term, s["rolling"], **kw)
            except Exception as e:  # noqa: BLE001
                vec_error = type(e).__name__
        fam.add(dict(fn="td1", shape=list(shape), time_dim=s["time_dim"], dtype=s["dtype"], regime=regime, gamma=gm,
                     rolling_gamma=s["rolling"], vec_error=vec_error), arrays)
    # reward2go
    for shape, td in (((4, 10), -1), ((5, 12, 1), -2), ((3, 9, 4), -2), ((2, 3, 8, 2), -2), ((2, 8, 3, 2), 1)):
        for dts in ("f32", "f64"):
            r = torch.randn(shape, generator=gen, dtype=torch.float64).to(DT[dts])
            done = torch.rand(shape, generator=gen) < 0.15
            kw = {} if td == -2 else {"time_dim": td}
            out = F.reward2go(r, done, 0.9, **kw)
            fam.add(dict(fn="reward2go", shape=list(shape), time_dim=td, dtype=dts, gamma=0.9), dict(reward=r, done=done, out=out))
    # the docstring example (functional.py:1381-1395) and the closed form of test_cost.py:16287-16309
    r = torch.ones(1, 10)
    d = torch.zeros(1, 10, dtype=torch.bool)
    d[:, [3, 7]] = True
    fam.add(dict(fn="reward2go", shape=[1, 10], time_dim=-1, dtype="f32", gamma=0.99, docstring=True),
            dict(reward=r, done=d, out=F.reward2go(r, d, 0.99, time_dim=-1)))
    return fam.save()


def gen_grads(F):
    fam = Family("grads")
    gen = torch.Generator().manual_seed(1357)

    def leaf(shape, dt, log=False):
        t = torch.randn(shape, generator=gen, dtype=torch.float64)
        if log:
            t = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-3).log()
        return t.to(dt).requires_grad_(True)

    def weights(shape, dt):
        return torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)

    def coef(form, value, shape, dt):
        if form == "float":
            return value, {"kind": "float", "value": value}, None
        t = (value - 0.1 * torch.rand(shape if form == "full" else (shape[0],) + (1,) * (len(shape) - 1), generator=gen,
                                      dtype=torch.float64)).to(dt).requires_grad_(True)
        return t, {"kind": "tensor", "form": form}, t

    cases = [((5, 24, 1), -2, "f64"), ((5, 24, 1), -2, "f32"), ((3, 40, 4), -2, "f64"), ((2, 6, 3, 2), -3, "f64"),
             ((1, 300, 1), -2, "f64"), ((64, 32, 1), -2, "f32")]
    for shape, td, dts in cases:
        dt = DT[dts]
        # with default dtype float32, python-float coefficients round to f32: the fixture keeps that
        for gform, lform in (("float", "float"), ("full", "float"), ("full", "full"), ("bcast", "float")):
            if gform != "float" and shape not in ((5, 24, 1), (3, 40, 4)):
                continue
            kw = {} if td == -2 else {"time_dim": td}
            done, term = make_masks(shape, "done_sup_term", gen)
            # --- GAE
            r, v, nv = leaf(shape, dt), leaf(shape, dt), leaf(shape, dt)
            g, gm, gleaf = coef(gform, 0.97, shape, dt)
            l, lm, lleaf = coef(lform, 0.9, shape, dt)
            w1, w2 = weights(shape, dt), weights(shape, dt)
            adv, vt = F.generalized_advantage_estimate(g, l, v, nv, r, done, term, **kw)
            ins = [r, v, nv] + [x for x in (gleaf, lleaf) if x is not None]
            grads = torch.autograd.grad((adv * w1).sum() + (vt * w2).sum(), ins)
            arrays = dict(reward=r, state_value=v, next_state_value=nv, done=done, terminated=term, w_advantage=w1,
                          w_value_target=w2, gamma=gleaf, lmbda=lleaf, d_reward=grads[0], d_state_value=grads[1],
                          d_next_state_value=grads[2])
            k = 3
            if gleaf is not None:
                arrays["d_gamma"] = grads[k]
                k += 1
            if lleaf is not None:
                arrays["d_lmbda"] = grads[k]
            fam.add(dict(fn="gae", shape=list(shape), time_dim=td, dtype=dts, gamma=gm, lmbda=lm), arrays)
            # --- TD(lambda), rolling
            if gform == "bcast":
                continue  # the loop variant takes scalars or full-shape tensors only
            r, nv = leaf(shape, dt), leaf(shape, dt)
            g, gm, gleaf = coef(gform, 0.97, shape, dt)
            l, lm, lleaf = coef(lform, 0.9, shape, dt)
            w1 = weights(shape, dt)
            ret = F.td_lambda_return_estimate(g, l, nv, r, done, term, **kw)
            ins = [r, nv] + [x for x in (gleaf, lleaf) if x is not None]
            grads = torch.autograd.grad((ret * w1).sum(), ins)
            arrays = dict(reward=r, next_state_value=nv, done=done, terminated=term, w_returns=w1, gamma=gleaf, lmbda=lleaf,
                          d_reward=grads[0], d_next_state_value=grads[1])
            k = 2
            if gleaf is not None:
                arrays["d_gamma"] = grads[k]
                k += 1
            if lleaf is not None:
                arrays["d_lmbda"] = grads[k]
            fam.add(dict(fn="td_lambda", shape=list(shape), time_dim=td, dtype=dts, gamma=gm, lmbda=lm), arrays)
            if lform != "float":
                continue
            # --- TD(1), rolling
            r, nv = leaf(shape, dt), leaf(shape, dt)
            g, gm, gleaf = coef(gform, 0.97, shape, dt)
            w1 = weights(shape, dt)
            ret = F.td1_return_estimate(g, nv, r, done, term, **kw)
            ins = [r, nv] + ([gleaf] if gleaf is not None else [])
            grads = torch.autograd.grad((ret * w1).sum(), ins)
            arrays = dict(reward=r, next_state_value=nv, done=done, terminated=term, w_returns=w1, gamma=gleaf,
                          d_reward=grads[0], d_next_state_value=grads[1])
            if gleaf is not None:
                arrays["d_gamma"] = grads[2]
            fam.add(dict(fn="td1", shape=list(shape), time_dim=td, dtype=dts, gamma=gm), arrays)
            # --- V-trace (thresholds below 1 so that both clipping branches occur)
            r, v, nv = leaf(shape, dt), leaf(shape, dt), leaf(shape, dt)
            lp, lmu = leaf(shape, dt, log=True), leaf(shape, dt, log=True)
            g, gm, gleaf = coef(gform, 0.97, shape, dt)
            w1, w2 = weights(shape, dt), weights(shape, dt)
            adv, vs = F.vtrace_advantage_estimate(g, lp, lmu, v, nv, r, done, term, 0.9, 0.8, **kw)
            ins = [lp, lmu, v, nv, r] + ([gleaf] if gleaf is not None else [])
            grads = torch.autograd.grad((adv * w1).sum() + (vs * w2).sum(), ins)
            arrays = dict(log_pi=lp, log_mu=lmu, state_value=v, next_state_value=nv, reward=r, done=done, terminated=term,
                          w_advantage=w1, w_vs=w2, gamma=gleaf, d_log_pi=grads[0], d_log_mu=grads[1], d_state_value=grads[2],
                          d_next_state_value=grads[3], d_reward=grads[4])
            if gleaf is not None:
                arrays["d_gamma"] = grads[5]
            fam.add(dict(fn="vtrace", shape=list(shape), time_dim=td, dtype=dts, gamma=gm, rho_thresh=0.9, c_thresh=0.8), arrays)
    return fam.save()


def gen_vtrace_thresh(F):
    fam = Family("vtrace_thresh")
    gen = torch.Generator().manual_seed(97531)
    for shape, td, tshape in (((5, 30, 1), -2, (5, 30, 1)), ((5, 30, 1), -2, (5, 1, 1)), ((4, 12, 3), -2, (3,)),
                              ((2, 6, 3, 2), -3, (2, 6, 3, 2))):
        for dts in ("f32", "f64"):
            dt = DT[dts]
            r = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
            v = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
            nv = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
            lp = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-6).log().to(dt)
            lmu = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-6).log().to(dt)
            done, term = make_masks(shape, "done_sup_term", gen)
            rho = (0.5 + torch.rand(tshape, generator=gen, dtype=torch.float64)).to(dt)
            c = (0.3 + torch.rand(tshape, generator=gen, dtype=torch.float64)).to(dt)
            kw = {} if td == -2 else {"time_dim": td}
            adv, vs = F.vtrace_advantage_estimate(0.98, lp, lmu, v, nv, r, done, term, rho, c, **kw)
            fam.add(dict(shape=list(shape), time_dim=td, dtype=dts, gamma=0.98, thresh_shape=list(tshape)),
                    dict(log_pi=lp, log_mu=lmu, state_value=v, next_state_value=nv, reward=r, done=done, terminated=term,
                         rho_thresh=rho, c_thresh=c, advantage=adv, vs=vs))
    return fam.save()


def main():
    os.makedirs(OUT, exist_ok=True)
    F = ref_import.load_functional()
    manifest = dict(generator="tools/gen_golden_next.py",
                    reference="AsbjornLorenzen/torchrl v0.7.2 torchrl/objectives/value/functional.py",
                    torch=torch.__version__, td=gen_td(F), grads=gen_grads(F), vtrace_thresh=gen_vtrace_thresh(F))
    with open(os.path.join(OUT, "manifest_next.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
