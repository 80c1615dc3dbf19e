"""Loader for the golden fixtures in tests/golden (data only; see tools/gen_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f32": torch.float32, "f64": torch.float64}

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)
with open(os.path.join(GOLDEN, "manifest_next.json")) as _f:  # tools/gen_golden_next.py: td0 / td1 / reward2go, gradients, ...
    MANIFEST.update({k: v for k, v in json.load(_f).items() if isinstance(v, list)})
with open(os.path.join(GOLDEN, "manifest_vec.json")) as _f:   # tools/gen_golden_vec.py: the reference's vec variants beside its loops
    MANIFEST.update({k: v for k, v in json.load(_f).items() if isinstance(v, list)})
with open(os.path.join(GOLDEN, "manifest_grads_nr.json")) as _f:   # tools/gen_golden_grads_nr.py: gradients of rolling_gamma=False
    MANIFEST.update({k: v for k, v in json.load(_f).items() if isinstance(v, list)})
_npz = {}


def cases(family, fn=None):
    """Cases of a fixture family; ``fn`` filters on the case's ``fn`` field (families that hold several functions)."""
    return [c for c in MANIFEST[family] if fn is None or c.get("fn") == fn]


def ids(family, fn=None):
    return [f"{family}-{c['id']}" for c in cases(family, fn)]


def arrays(family, case):
    if family not in _npz:
        _npz[family] = np.load(os.path.join(GOLDEN, family + ".npz"), allow_pickle=False)
    z = _npz[family]
    pre = case["id"] + "."
    return {k[len(pre):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(pre)}


def coef(meta, arrs, name):
    """Rebuild gamma / lmbda exactly as it was handed to the reference."""
    m = meta.get(name)
    if m is None:
        return None
    if m["kind"] == "float":
        return m["value"]
    return arrs[name]


def expand_masks(meta, arrs):
    """Fixtures store expanded masks materialised; re-create the stride-0 view when flagged."""
    done, term = arrs["done"], arrs.get("terminated")
    if meta.get("expanded_masks"):
        done = done[..., :1, :].expand(done.shape)
        if term is not None:
            term = term[..., :1, :].expand(term.shape)
    return done, term


def time_kw(meta):
    return {} if (meta["time_dim"] == -2 and len(meta["shape"]) > 1) else {"time_dim": meta["time_dim"]}


def checksums():
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        return json.load(f)
