"""Golden gradients of the NON-rolling forms (rolling_gamma=False) of TD(lambda) and TD(1), from the imported reference.

Runs ONLY in the build container (needs /root/reference; tools/ref_import.py).  Expected values are torch autograd THROUGH the
reference's O(T^2) loops (functional.py:877-887, 550-561) w.r.t. reward, next_state_value and the gamma / lmbda tensors; the
scalar loss is sum(returns * W) with the random weights W stored in the fixture.  Data only: grads_nr.npz +
manifest_grads_nr.json.

    python tools/gen_golden_grads_nr.py
"""
from __future__ import annotations

import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
from gen_golden import DT, OUT, Family, make_masks  # noqa: E402


def main():
    F = ref_import.load_functional()
    fam = Family("grads_nr")
    gen = torch.Generator().manual_seed(8642)

    def leaf(shape, dt):
        return torch.randn(shape, generator=gen, dtype=torch.float64).to(dt).requires_grad_(True)

    def coef(value, shape, dt):
        return (value - 0.1 * torch.rand(shape, generator=gen, dtype=torch.float64)).to(dt).requires_grad_(True)

    for shape, td, dts, regime in (((3, 12, 1), -2, "f64", "done_sup_term"), ((3, 12, 1), -2, "f32", "done_sup_term"),
                                   ((4, 10, 3), -2, "f64", "done_sup_term"), ((2, 9, 4, 1), -3, "f64", "term_free"),
                                   ((2, 70, 1), -2, "f64", "done_sup_term"), ((5, 33, 1), -2, "f32", "none"),
                                   ((2, 300, 1), -2, "f64", "done_sup_term")):
        dt = DT[dts]
        kw = {} if td == -2 else {"time_dim": td}
        done, term = make_masks(shape, "done_sup_term" if regime == "term_free" else regime, gen)
        if regime == "term_free":
            term = torch.rand(shape, generator=gen) < 0.15
        w = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        # TD(lambda): gamma tensor with scalar lmbda, both tensors, lmbda tensor with scalar gamma
        for gten, lten in ((True, False), (True, True), (False, True)):
            r, nv = leaf(shape, dt), leaf(shape, dt)
            g = coef(0.97, shape, dt) if gten else 0.97
            l = coef(0.9, shape, dt) if lten else 0.9
            ret = F.td_lambda_return_estimate(g, l, nv, r, done, term, False, **kw)
            ins = [r, nv] + [x for x in (g, l) if isinstance(x, torch.Tensor)]
            grads = torch.autograd.grad((ret * w).sum(), ins)
            arrays = dict(reward=r, next_state_value=nv, done=done, terminated=term, w_returns=w, returns=ret,
                          gamma=g if gten else None, lmbda=l if lten else None, d_reward=grads[0], d_next_state_value=grads[1])
            k = 2
            if gten:
                arrays["d_gamma"] = grads[k]
                k += 1
            if lten:
                arrays["d_lmbda"] = grads[k]
            fam.add(dict(fn="td_lambda", shape=list(shape), time_dim=td, dtype=dts, regime=regime,
                         gamma={"kind": "tensor"} if gten else {"kind": "float", "value": 0.97},
                         lmbda={"kind": "tensor"} if lten else {"kind": "float", "value": 0.9}), arrays)
        # TD(1)
        r, nv = leaf(shape, dt), leaf(shape, dt)
        g = coef(0.97, shape, dt)
        ret = F.td1_return_estimate(g, nv, r, done, term, False, **kw)
        grads = torch.autograd.grad((ret * w).sum(), [r, nv, g])
        fam.add(dict(fn="td1", shape=list(shape), time_dim=td, dtype=dts, regime=regime, gamma={"kind": "tensor"}),
                dict(reward=r, next_state_value=nv, done=done, terminated=term, w_returns=w, returns=ret, gamma=g,
                     d_reward=grads[0], d_next_state_value=grads[1], d_gamma=grads[2]))
    cases = fam.save()
    with open(os.path.join(OUT, "manifest_grads_nr.json"), "w") as f:
        json.dump(dict(generator="tools/gen_golden_grads_nr.py",
                       reference="AsbjornLorenzen/torchrl v0.7.2 torchrl/objectives/value/functional.py",
                       torch=torch.__version__, grads_nr=cases), f, indent=1)
    print("grads_nr.npz", os.path.getsize(os.path.join(OUT, "grads_nr.npz")), "cases", len(cases))


if __name__ == "__main__":
    main()
