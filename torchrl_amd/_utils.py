"""``_standardize`` -- the normalisation PPO applies to the advantage right after the value
estimator (reference: torchrl/_utils.py:875-936, called from torchrl/objectives/ppo.py:688, 975,
1282), on the HIP kernels of include/trl_value.h (trl_moments / trl_normalize and their *_grouped
forms).  Same signature, validation, warning and error messages as the reference; ``group``
(extension) all-reduces the per-group moments of a batch sharded over ranks."""
from __future__ import annotations

import functools
import math
import warnings
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _extension
from .objectives.value import functional as F

__all__ = ["_standardize"]

_SCRATCH = {}  # (device, stream, G) -> (per-group moments, workspace pointer, workspace bytes, buffer)
_NEED = {}     # (outer, G, inner) -> trl_grouped_workspace_bytes


def _grouped_pattern(shape, exclude_dims):
    return _grouped_pattern_cached(tuple(shape), tuple(exclude_dims))


@functools.lru_cache(maxsize=256)
def _grouped_pattern_cached(shape, exclude_dims):
    """(outer, G, inner, permutation or None): excluded dims that are adjacent merge into one group
    axis of a contiguous tensor; otherwise they are permuted to the front."""
    shape, ex = list(shape), sorted(exclude_dims)
    if ex == list(range(ex[0], ex[-1] + 1)):
        return math.prod(shape[:ex[0]]), math.prod(shape[ex[0]:ex[-1] + 1]), math.prod(shape[ex[-1] + 1:]), None
    perm = ex + [d for d in range(len(shape)) if d not in ex]
    G = math.prod(shape[d] for d in ex)
    return 1, G, math.prod(shape) // G, perm


def _standardize(input: Tensor, exclude_dims: Tuple[int] = (), mean: Optional[Tensor] = None, std: Optional[Tensor] = None,
                 eps: Optional[float] = None, *, group=None) -> Tensor:
    """Standardizes ``input``, optionally keeping some dims (e.g. the agent dim) independent.

    ``exclude_dims``: dims excluded from the statistics (may be negative); ``mean`` / ``std``: use
    these instead of the batch statistics (broadcastable to ``input``); ``eps``: lower clamp of the
    std (default: float32 resolution, 1e-6)."""
    if eps is None:
        eps = torch.finfo(torch.float).resolution if input.dtype.is_floating_point else 1e-6
    n_ex = len(exclude_dims)
    shape = input.shape
    if n_ex:
        exclude_dims = [d if d >= 0 else d + len(shape) for d in exclude_dims]
        if len(set(exclude_dims)) != n_ex:
            raise ValueError("Exclude dims has repeating elements")
        if any(d < 0 or d >= len(shape) for d in exclude_dims):
            raise ValueError(f"exclude_dims={exclude_dims} provided outside bounds for input of shape={shape}")
        if n_ex == len(shape):
            warnings.warn("_standardize called but all dims were excluded from the statistics, returning unprocessed input")
            return input
    if mean is not None or std is not None:
        # caller-provided statistics: nothing to reduce for that part; the reference's expression on the device
        included = tuple(d for d in range(len(shape)) if d not in exclude_dims) if n_ex else None
        if mean is None:
            mean = torch.mean(input, keepdim=True, dim=included) if n_ex else input.mean()
        else:
            mean = torch.as_tensor(mean, device=input.device)
        if std is None:
            std = torch.std(input, keepdim=True, dim=included) if n_ex else input.std()
        else:
            std = torch.as_tensor(std, device=input.device)
        return (input - mean) / std.clamp_min(eps)
    if group is None and input.__class__ is Tensor and not torch.compiler.is_compiling():
        host = F._HOST if F._HOST_TRIED else F._host_module()
        if host is not None:  # checks, output allocation and the C-ABI call in one C++ hop (csrc/host_fast.cpp)
            out = host.standardize(input, tuple(exclude_dims), float(eps))
            if out is not None:
                if out.__class__ is int:
                    _extension.check(out, _extension.get_lib())
                else:
                    return out
    if not n_ex:
        return F.standardize(input, eps, group=group).view(shape)

    lib = _extension.get_lib()
    F._require_hip(input)
    if input.dtype not in F._DT:
        raise NotImplementedError(f"_standardize: dtype {input.dtype} unsupported (float32 / float64)")
    if input.numel() == 0:
        return input.clone()
    outer, G, inner, perm = _grouped_pattern(tuple(shape), tuple(exclude_dims))
    x = input.detach()
    x = x.contiguous() if perm is None else x.permute(perm).contiguous()
    dev = x.device
    stream = F._stream(dev)
    stats, wsp, wsn = _scratch(lib, dev, stream, outer, G, inner)
    out = torch.empty_like(x)
    sharded = group is not None and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1
    dt = F._DT[x.dtype]
    with F._on(dev):
        if not sharded:
            # one C-ABI call; ONE launch for G <= 16 groups of period G * inner <= 1024 (values kept in registers across an
            # in-launch gather of the block moments), else moments -> per-group reduce -> scale pass
            rc = lib.trl_standardize_grouped(stream, dt, x.data_ptr(), outer, G, inner, float(eps), out.data_ptr(), stats.data_ptr(), wsp, wsn)
            if rc:
                _extension.check(rc, lib)
        else:
            _extension.check(lib.trl_moments_grouped(stream, dt, F._ptr(x), outer, G, inner, F._ptr(stats), wsp, wsn), lib)
            torch.distributed.all_reduce(stats, group=group)
            _extension.check(lib.trl_normalize_grouped(stream, dt, F._ptr(x), outer, G, inner, F._ptr(stats), float(eps), F._ptr(out)), lib)
    if perm is not None:
        inv = [0] * len(perm)
        for k, d in enumerate(perm):
            inv[d] = k
        out = out.permute(inv).contiguous()
    return out.view(shape)


def _scratch(lib, dev, stream, outer, G, inner):
    """(per-group moments tensor, workspace pointer, workspace bytes) on (device, stream): one buffer, ZERO-FILLED when
    created (include/trl_value.h: the grouped workspace carries the epoch of the one-launch route), the moments at its head."""
    key = (dev.index, stream, G)
    hit = _SCRATCH.get(key)
    need = _NEED.get((outer, G, inner))
    if need is None:
        need = _NEED[(outer, G, inner)] = lib.trl_grouped_workspace_bytes(outer, G, inner)
    if hit is None or hit[2] < need:
        head = ((G * 24 + 255) // 256) * 256
        buf = torch.zeros(head + max(need, 1 << 16), dtype=torch.uint8, device=dev)
        hit = _SCRATCH[key] = (buf[:G * 24].view(torch.float64), buf.data_ptr() + head, buf.numel() - head, buf)
        if F._HOST:  # the host module mirrors this table: it must never keep the address of a replaced buffer
            F._HOST.set_grouped_scratch(dev.index if dev.index is not None else torch.cuda.current_device(), stream, G,
                                        hit[0].data_ptr(), hit[1], hit[2])
    return hit[0], hit[1], hit[2]
