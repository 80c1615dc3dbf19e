"""Golden vectors for `_standardize` (reference: torchrl/_utils.py:875-936), generated in the build
container from the reference itself (imported with the shim of tools/ref_import.py plus stubs for the two
tensordict names _utils.py imports).  Writes tests/golden/standardize.npz + standardize_manifest.json.
    python tools/gen_golden_standardize.py"""
import importlib
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
DT = {"f32": torch.float32, "f64": torch.float64}


def load_standardize():
    ref_import.load_functional()  # seeds the tensordict stub and the _trl_ref parents
    sys.modules["tensordict"].unravel_key = lambda k: k
    sys.modules["tensordict.utils"].NestedKey = object
    saved = sys.modules.get("torchrl")
    try:
        sys.modules["torchrl"] = sys.modules["_trl_ref"]
        U = importlib.import_module("torchrl._utils")
    finally:
        if saved is None:
            sys.modules.pop("torchrl", None)
        else:
            sys.modules["torchrl"] = saved
    return U._standardize


def main():
    std = load_standardize()
    gen = torch.Generator().manual_seed(4242)
    arrays, cases = {}, []

    def add(meta, x, out, extra=None):
        cid = f"s{len(cases):03d}"
        arrays[cid + ".x"] = x.numpy()
        arrays[cid + ".out"] = out.numpy()
        for k, v in (extra or {}).items():
            arrays[f"{cid}.{k}"] = v.numpy()
        cases.append(dict(meta, id=cid))

    # the reference's own test tensor (test_cost.py:16166-16176)
    t = torch.arange(3 * 4 * 5 * 6, dtype=torch.float32).view(3, 4, 5, 6)
    add(dict(shape=[3, 4, 5, 6], dtype="f32", exclude_dims=[1, 3], eps=None, origin="test_cost.py:16168"), t, std(t, exclude_dims=(1, 3)))
    add(dict(shape=[3, 4, 5, 6], dtype="f32", exclude_dims=[], eps=None, origin="test_cost.py:16175"), t, std(t, ()))
    add(dict(shape=[3, 4, 5, 6], dtype="f32", exclude_dims=[], eps=None, mean=-1.0, std=2.0, origin="test_cost.py:16173"), t, std(t, (), -1, 2))
    specs = [((32, 10, 5, 1), (-2,)), ((32, 10, 5, 1), (2, 3)), ((32, 10, 5, 1), (0,)), ((32, 10, 5, 1), (1,)), ((32, 10, 5, 1), (0, 2)),
             ((64, 7), (1,)), ((64, 7), (0,)), ((6, 4, 3, 2, 5), (1, 3)), ((6, 4, 3, 2, 5), (-1,)), ((6, 4, 3, 2, 5), (0, 4)),
             ((128, 16, 3), (-1,)), ((2, 3), (0,)), ((50,), ())]
    for shape, ex in specs:
        for dt in ("f32", "f64"):
            for eps in (None, 1e-4):
                x = (torch.randn(shape, generator=gen, dtype=torch.float64) * 2.5 + 0.7).to(DT[dt])
                if len(shape) > 1:
                    x.select(-1, 0).mul_(1e-3)  # groups of very different scale
                out = std(x, exclude_dims=ex, eps=eps)
                add(dict(shape=list(shape), dtype=dt, exclude_dims=list(ex), eps=eps), x, out)
    # constant groups: std = 0 -> the eps clamp decides
    x = torch.ones(8, 4, 3) * torch.tensor([1.0, -2.0, 0.5])
    add(dict(shape=[8, 4, 3], dtype="f32", exclude_dims=[2], eps=1e-6, regime="constant groups"), x, std(x, exclude_dims=(2,), eps=1e-6))
    # caller-provided mean / std with excluded dims (broadcastable keepdim shapes)
    x = torch.randn(16, 5, 2, generator=gen)
    m, s = torch.randn(1, 5, 1, generator=gen), torch.rand(1, 5, 1, generator=gen) + 0.5
    add(dict(shape=[16, 5, 2], dtype="f32", exclude_dims=[1], eps=None, given="mean,std"), x, std(x, (1,), m, s), dict(mean=m, std=s))
    add(dict(shape=[16, 5, 2], dtype="f32", exclude_dims=[1], eps=None, given="mean"), x, std(x, (1,), m, None), dict(mean=m))
    # errors / warnings (type + message)
    errors = []
    for name, fn in (("repeated", lambda: std(x, (1, 1))), ("out_of_bounds", lambda: std(x, (3,))), ("negative_out_of_bounds", lambda: std(x, (-4,)))):
        try:
            fn()
            errors.append(dict(name=name, type=None))
        except Exception as e:  # noqa: BLE001
            errors.append(dict(name=name, type=type(e).__name__, message=str(e)))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = std(x, (0, 1, 2))
        errors.append(dict(name="all_excluded", type="warning", message=str(w[0].message), returns_input=bool(out is x)))
    np.savez_compressed(os.path.join(OUT, "standardize.npz"), **arrays)
    with open(os.path.join(OUT, "standardize_manifest.json"), "w") as f:
        json.dump(dict(generator="tools/gen_golden_standardize.py", reference="AsbjornLorenzen/torchrl v0.7.2 torchrl/_utils.py:875-936",
                       torch=torch.__version__, cases=cases, errors=errors), f, indent=1)
    print(len(cases), "cases;", os.path.getsize(os.path.join(OUT, "standardize.npz")), "bytes")
    print(errors)


if __name__ == "__main__":
    main()
