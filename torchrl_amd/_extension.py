"""Discovery and loading of the gfx950 value-estimation library.

Mirrors the reference's extension convention (torchrl/_extension.py:17-31: probe for the
native module at import time, warn -- never hard-fail -- if it is absent; consumers raise
when they actually need it, torchrl/data/replay_buffers/samplers.py:23-37).  Here the native
module is a plain C-ABI shared library (include/trl_value.h) loaded with ctypes: no torch
headers, no pybind in the hot call.

There is NO fallback implementation: every estimator in
``torchrl_amd.objectives.value`` fails loudly (``ExtensionMissingError``) when the
library cannot be loaded.
"""
from __future__ import annotations

import ctypes
import os
import warnings

_LIB_NAME = "libtrl_value.so"
# TRL_VALUE_LIB: development switch (kernel variants built beside the product library); unset in normal use
_LIB_PATH = os.environ.get("TRL_VALUE_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", _LIB_NAME)
ABI_VERSION = 8

EXTENSION_WARNING = (
    f"Failed to load the torchrl_amd HIP library ({_LIB_PATH}). The value estimators "
    "(GAE / TD(lambda) / V-trace) will raise when called. Build it with "
    "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C torchrl_amd/csrc` "
    "(hipcc --offload-arch=gfx950)."
)


class ExtensionMissingError(RuntimeError):
    pass


_lib = None
_load_error = None

_vp, _i64, _dbl, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_size_t

# symbol -> (restype, argtypes); must list every entry point of include/trl_value.h
_PROTOTYPES = {
    "trl_abi_version": (_int, []),
    "trl_last_error": (ctypes.c_char_p, []),
    "trl_last_kernel": (ctypes.c_char_p, []),
    "trl_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "trl_workspace_error": (_int, [_vp]),
    "trl_fused_degraded": (_int, []),
    "trl_fused_reset": (_int, []),
    "trl_gae_fwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl,
                           _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _int]),
    "trl_gae_normalize_fwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl,
                                     _i64, _i64, _i64, _vp, _vp, _dbl, _vp, _vp, _sz, _int]),
    "trl_static_scratch_bytes": (_sz, []),
    "trl_gae_normalize_fwd_noalloc": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl,
                                             _i64, _i64, _i64, _vp, _vp, _dbl, _vp, _int]),
    "trl_td_lambda_fwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _int,
                                 _i64, _i64, _i64, _vp, _vp, _sz, _int]),
    "trl_td1_fwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _int, _vp, _dbl, _int,
                           _i64, _i64, _i64, _vp, _vp, _sz, _int]),
    "trl_vtrace_fwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _dbl, _dbl, _dbl, _vp, _vp,
                              _i64, _i64, _i64, _vp, _vp, _vp, _sz, _int]),
    "trl_gae_bwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl,
                           _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _int]),
    "trl_td_lambda_bwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl,
                                 _i64, _i64, _i64, _vp, _vp, _vp, _sz, _int]),
    "trl_td1_bwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _int, _vp, _dbl,
                           _i64, _i64, _i64, _vp, _vp, _vp, _sz, _int]),
    "trl_td_fixed_bwd": (_int, [_vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl,
                                _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "trl_vtrace_bwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _dbl, _dbl, _dbl, _vp, _vp,
                              _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int]),
    "trl_normalize": (_int, [_vp, _int, _vp, _i64, _vp, _dbl, _vp]),
    "trl_moments": (_int, [_vp, _int, _vp, _i64, _vp, _vp, _sz]),
    "trl_standardize": (_int, [_vp, _int, _vp, _i64, _dbl, _vp, _vp, _vp, _sz]),
    "trl_standardize_grouped": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _dbl, _vp, _vp, _vp, _sz]),
    "trl_grouped_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "trl_moments_grouped": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _vp, _vp, _sz]),
    "trl_normalize_grouped": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _vp, _dbl, _vp]),
    "trl_xchg_handle_bytes": (_sz, []),
    "trl_xchg_local_init": (_int, [ctypes.POINTER(_vp), _vp]),
    "trl_xchg_open": (_int, [_vp, _vp, ctypes.POINTER(_int), _int, _int, _dbl, ctypes.POINTER(_vp)]),
    "trl_xchg_allreduce_stats": (_int, [_vp, _vp, _vp]),
    "trl_xchg_set_timeout": (_int, [_vp, _dbl]),
    "trl_xchg_error": (_int, [_vp]),
    "trl_gae_normalize_xchg_fwd": (_int, [_vp, _int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl,
                                          _i64, _i64, _i64, _vp, _vp, _dbl, _vp, _vp, _sz, _vp, _int]),
    "trl_xchg_destroy": (_int, [_vp, _vp]),
    # host entry points (tensors in host memory)
    "trl_cpu_threads": (_int, []),
    "trl_set_cpu_threads": (_int, [_int]),
    "trl_gae_fwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl, _i64, _i64, _i64, _vp, _vp, _vp]),
    "trl_gae_normalize_fwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl, _i64, _i64, _i64, _vp, _vp,
                                         _dbl, _vp]),
    "trl_td_lambda_fwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _int, _i64, _i64, _i64, _vp]),
    "trl_td1_fwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _int, _vp, _dbl, _int, _i64, _i64, _i64, _vp]),
    "trl_vtrace_fwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _dbl, _dbl, _dbl, _vp, _vp,
                                  _i64, _i64, _i64, _vp, _vp]),
    "trl_gae_bwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _dbl, _i64, _i64, _i64, _vp, _vp, _vp]),
    "trl_td_lambda_bwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _int, _vp, _vp, _dbl, _dbl, _i64, _i64, _i64, _vp, _vp]),
    "trl_td1_bwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _int, _vp, _dbl, _i64, _i64, _i64, _vp, _vp]),
    "trl_vtrace_bwd_cpu": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _dbl, _dbl, _dbl, _vp, _vp,
                                  _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "trl_moments_cpu": (_int, [_int, _vp, _i64, _vp]),
    "trl_normalize_cpu": (_int, [_int, _vp, _i64, _vp, _dbl, _vp]),
}


def lib_path() -> str:
    return _LIB_PATH


def is_available() -> bool:
    return os.path.exists(_LIB_PATH)


def _load():
    global _lib, _load_error
    if _lib is not None:
        return _lib
    try:
        # torch first: the library must bind to the HIP runtime torch ships (same SONAME,
        # libamdhip64.so.7).  Loaded before torch, it would pull in the system copy through its
        # RUNPATH and the process would hold two HIP runtimes that do not see each other's state.
        import torch  # noqa: F401

        lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.trl_abi_version()
        if got != ABI_VERSION:
            raise OSError(f"ABI version mismatch: library {got}, python {ABI_VERSION}")
        _lib = lib
    except (OSError, AttributeError) as e:  # missing file, missing symbol, missing HIP runtime
        _load_error = e
        _lib = None
    return _lib


def _init_extension():
    """Called at package import (as torchrl/__init__.py:22,35 does): warn if the library is absent."""
    if _load() is None:
        warnings.warn(EXTENSION_WARNING + (f" [{_load_error}]" if _load_error else ""))


_host = False  # False: not tried yet; None: absent / disabled; else the _trl_host module


def get_host():
    """The host-side fast-path module (csrc/host_fast.cpp, built beside the library as lib/_trl_host.so) or None when it is
    not built or ``TRL_NO_HOST_SHIM=1``: the python entry points then make the same C-ABI call from python (same kernels, a
    few microseconds more host time per call)."""
    global _host
    if _host is False:
        _host = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_trl_host.so")
        if os.environ.get("TRL_NO_HOST_SHIM") != "1" and os.path.exists(path):
            import importlib.machinery
            import importlib.util
            try:
                loader = importlib.machinery.ExtensionFileLoader("_trl_host", path)
                spec = importlib.util.spec_from_file_location("_trl_host", path, loader=loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _host = mod
            except ImportError as e:  # built against another torch: say so once, keep the python route
                warnings.warn(f"torchrl_amd: host fast-path module {path} failed to load ({e}); using the python route")
    return _host


def fn_address(fn) -> int:
    """Address of a ctypes function of the loaded library (handed to the host module)."""
    return ctypes.cast(fn, ctypes.c_void_p).value


def get_lib():
    """Return the loaded library or raise loudly -- there is no fallback path."""
    lib = _load()
    if lib is None:
        raise ExtensionMissingError(EXTENSION_WARNING + (f" [{_load_error}]" if _load_error else ""))
    return lib


class TrlError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


WARN_FUSED_DEGRADED = 100  # include/trl_value.h: the call completed; the single-launch route was switched off on this GPU


def check(status: int, lib=None):
    if status != 0:
        lib = lib or get_lib()
        msg = lib.trl_last_error().decode("utf-8", "replace")
        if status == WARN_FUSED_DEGRADED:
            warnings.warn(f"torchrl_amd: {msg}", RuntimeWarning, stacklevel=3)
            return
        raise TrlError(f"torchrl_amd native call failed (status {status}): {msg}")
