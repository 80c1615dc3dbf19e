"""Golden fixtures of the reference's own VEC variants, from the imported reference.

Runs ONLY in the build container (needs /root/reference; tools/ref_import.py).  For every case the fixture holds the
inputs, the reference LOOP outputs (the parity oracle) and the reference VEC outputs of the same call
(vec_generalized_advantage_estimate, functional.py:272-370; vec_td_lambda_return_estimate, :1030-1183), which differ
from the loop by the reference's own filter truncation (its tests compare the two at 1e-4).  The scan kernels here are
exact, so tests/test_gpu_next_rows.py checks them against the loop outputs at the SCAN bar and against the vec outputs
at the reference's 1e-4 -- and records the reference's own vec-vs-loop gap per case in the manifest.

Fixtures are data only (vec.npz + manifest_vec.json).

    python tools/gen_golden_vec.py
"""
from __future__ import annotations

import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
from gen_golden import DT, OUT, Family, coef_form, make_masks  # noqa: E402


def main():
    F = ref_import.load_functional()
    fam = Family("vec")
    gen = torch.Generator().manual_seed(13579)
    specs = []
    for shape, td, ex in (((1, 2048, 1), -2, None), ((8, 128, 1), -2, None), ((16, 100, 5, 1), -3, 2), ((32, 32, 1), -2, None),
                          ((3, 7, 2, 5), -3, None)):
        for dts in ("f32", "f64"):
            specs.append(dict(shape=shape, time_dim=td, expand=ex, dtype=dts, gform="float", lform="float", gl=(0.99, 0.95)))
    specs.append(dict(shape=(8, 128, 1), time_dim=-2, expand=None, dtype="f32", gform="float", lform="float", gl=(0.9, 0.9)))
    for gform, lform in (("full_f32", "float"), ("full_f32", "full_f32"), ("t0_f32", "t0_f32"), ("full_f64", "float")):
        for dts in ("f32", "f64"):
            specs.append(dict(shape=(6, 40, 1), time_dim=-2, expand=None, dtype=dts, gform=gform, lform=lform, gl=(0.99, 0.95)))
    for s in specs:
        shape, dt = s["shape"], DT[s["dtype"]]
        r = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        v = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        nv = torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
        if s["expand"] is not None:   # per-env masks expanded over the agent dim (stride 0), as DoneTransform produces
            mshape = tuple(1 if k == s["expand"] else n for k, n in enumerate(shape))
            done, term = make_masks(mshape, "done_sup_term", gen)
            done, term = done.expand(shape), term.expand(shape)
        else:
            done, term = make_masks(shape, "done_sup_term", gen)
        g, gm, ga = coef_form(s["gform"], s["gl"][0], shape, gen)
        l, lm, la = coef_form(s["lform"], s["gl"][1], shape, gen)
        kw = {} if s["time_dim"] == -2 else {"time_dim": s["time_dim"]}
        adv, vt = F.generalized_advantage_estimate(g, l, v, nv, r, done, term, **kw)
        ret = F.td_lambda_return_estimate(g, l, nv, r, done, term, **kw)
        arrays = dict(reward=r, state_value=v, next_state_value=nv, done=done.contiguous(), terminated=term.contiguous(), gamma=ga,
                      lmbda=la, advantage=adv.contiguous(), value_target=vt.contiguous(), returns=ret.contiguous())
        gap, vec_error = {}, {}
        # the reference's vec paths reject some inputs its loops accept (e.g. float64 values with a float32 gamma tensor:
        # einsum dtype mismatch in _custom_conv1d): the exception type is recorded, the loop outputs still pin the case
        try:
            vadv, vvt = F.vec_generalized_advantage_estimate(g, l, v, nv, r, done, term, **kw)
            arrays.update(vec_advantage=vadv.contiguous(), vec_value_target=vvt.contiguous())
            gap["gae"] = float((vadv.double() - adv.double()).abs().max())
        except Exception as e:  # noqa: BLE001
            vec_error["gae"] = type(e).__name__
        try:
            vret = F.vec_td_lambda_return_estimate(g, l, nv, r, done, term, **kw)
            arrays.update(vec_returns=vret.contiguous())
            gap["td_lambda"] = float((vret.double() - ret.double()).abs().max())
        except Exception as e:  # noqa: BLE001
            vec_error["td_lambda"] = type(e).__name__
        fam.add(dict(shape=list(shape), time_dim=s["time_dim"], dtype=s["dtype"], gamma=gm, lmbda=lm,
                     expanded_masks=s["expand"] is not None, reference_vec_minus_loop_max_abs=gap, reference_vec_error=vec_error),
                arrays)
    cases = fam.save()
    with open(os.path.join(OUT, "manifest_vec.json"), "w") as f:
        json.dump(dict(generator="tools/gen_golden_vec.py",
                       reference="AsbjornLorenzen/torchrl v0.7.2 torchrl/objectives/value/functional.py",
                       torch=torch.__version__, vec=cases), f, indent=1)
    print("vec.npz", os.path.getsize(os.path.join(OUT, "vec.npz")), "cases", len(cases),
          "max reference vec-loop gap", max(max(c["reference_vec_minus_loop_max_abs"].values(), default=0.0) for c in cases),
          "vec errors", [(c["id"], c["reference_vec_error"]) for c in cases if c["reference_vec_error"]])


if __name__ == "__main__":
    main()
