"""Import the reference's functional value estimators (build-container only).

The reference tree at /root/reference never travels to the GPU box; this helper is
used only by tools/gen_golden.py and by the oracle-vs-reference pin tests that skip
when /root/reference is absent.  `tensordict` is not installed here, so the parent
packages are pre-seeded; every arithmetic line executed is the
reference's own (torchrl/objectives/value/functional.py, value/utils.py).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("TRL_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "torchrl/objectives/value/functional.py"))


def load_functional():
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    if "tensordict" not in sys.modules:
        td = types.ModuleType("tensordict")
        td.TensorDictBase = type("TensorDictBase", (), {})
        tdu = types.ModuleType("tensordict.utils")

        def expand_right(t, shape):
            while t.ndimension() < len(shape):
                t = t.unsqueeze(-1)
            return t.expand(shape)

        tdu.expand_right = expand_right
        td.utils = tdu
        sys.modules["tensordict"] = td
        sys.modules["tensordict.utils"] = tdu
    ref = os.path.join(REF_ROOT, "torchrl")
    for name, path in [
        ("_trl_ref", ref),
        ("_trl_ref.objectives", ref + "/objectives"),
        ("_trl_ref.objectives.value", ref + "/objectives/value"),
    ]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    # functional.py does `from torchrl.objectives.value.utils import ...`: alias the
    # package name to the seeded parents for the duration of the import only, so that
    # an installed/in-repo package called torchrl is never shadowed afterwards.
    saved = {k: sys.modules.get(k) for k in ("torchrl", "torchrl.objectives", "torchrl.objectives.value")}
    try:
        sys.modules["torchrl"] = sys.modules["_trl_ref"]
        sys.modules["torchrl.objectives"] = sys.modules["_trl_ref.objectives"]
        sys.modules["torchrl.objectives.value"] = sys.modules["_trl_ref.objectives.value"]
        F = importlib.import_module("torchrl.objectives.value.functional")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return F
