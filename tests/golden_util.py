"""Loader for the golden fixtures in tests/golden (data only; see tools/gen_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"f32": torch.float32, "f64": torch.float64}

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)
_npz = {}


def cases(family):
    return MANIFEST[family]


def ids(family):
    return [f"{family}-{c['id']}" for c in MANIFEST[family]]


def arrays(family, case):
    if family not in _npz:
        _npz[family] = np.load(os.path.join(GOLDEN, family + ".npz"), allow_pickle=False)
    z = _npz[family]
    pre = case["id"] + "."
    return {k[len(pre):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith(pre)}


def coef(meta, arrs, name):
    """Rebuild gamma / lmbda exactly as it was handed to the reference."""
    m = meta[name]
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
