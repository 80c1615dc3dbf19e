"""Sequence utilities of the value-estimation package (reference: torchrl/objectives/value/utils.py).

The reference builds its vectorised estimators out of these (pad / conv1d / decay tables); the HIP
scan kernels need none of them, but they are part of the module's surface -- the recurrent modules
(``torchrl/modules/tensordict_module/rnn.py``) split rollouts with them and ``TestUtils``
(test/test_cost.py:16158-16338) pins them.  Pure tensor code, device-agnostic, no native call:
every function is a restatement with index arithmetic (one ``gather`` / boolean mask each) of what
the reference obtains with expand / roll / masked_scatter chains.
"""
from __future__ import annotations

from typing import Mapping, Union

import torch

__all__ = [
    "_custom_conv1d", "roll_by_gather", "_make_gammas_tensor", "_flatten_batch", "_get_num_per_traj",
    "_split_and_pad_sequence", "_inv_pad_sequence", "_get_num_per_traj_init",
]


def _custom_conv1d(tensor: torch.Tensor, filter: torch.Tensor) -> torch.Tensor:
    """One-sided, zero-padded correlation along time (reference: utils.py:14-88).

    ``tensor``: ``[B, 1, T]``; ``filter``: ``[F, 1]`` (one filter for everything) or
    ``[B, T, F, 1]`` (one filter per batch element and time step).
    ``out[b, 0, t] = sum_j filter[(b, t,) j] * tensor[b, 0, t + j]`` with ``tensor`` read as 0 beyond ``T``.
    """
    B, _, T = tensor.shape
    if filter.ndimension() > 2:
        filt = filter.reshape(*filter.shape[:-2], filter.shape[-2])          # [..., T, F]
    else:
        filt = filter.reshape(-1)                                            # [F]
    F_ = filt.shape[-1]
    pos = torch.arange(T, device=tensor.device).unsqueeze(-1) + torch.arange(F_, device=tensor.device)  # [T, F]: t + j
    inside = pos < T
    window = tensor.reshape(B, T)[:, pos.clamp_max(T - 1)] * inside.to(tensor.dtype)                    # [B, T, F]
    out = (window * filt).sum(-1).unsqueeze(-2)
    if out.shape != tensor.shape:
        raise RuntimeError(f"wrong output shape: input shape: {tensor.shape}, output shape: {out.shape}")
    return out


def roll_by_gather(mat: torch.Tensor, dim: int, shifts: torch.LongTensor) -> torch.Tensor:
    """Roll every column (``dim`` 0 / -2) or every row (``dim`` 1 / -1) of a batched matrix by its own
    shift (reference: utils.py:91-128).

        >>> x = torch.arange(12).view(3, 4)
        >>> roll_by_gather(x, 0, -torch.arange(4))
        tensor([[ 0,  5, 10,  3],
                [ 4,  9,  2,  7],
                [ 8,  1,  6, 11]])
        >>> roll_by_gather(x, 1, -torch.arange(3))
        tensor([[ 0,  1,  2,  3],
                [ 5,  6,  7,  4],
                [10, 11,  8,  9]])
    """
    *batch, n_rows, n_cols = mat.shape
    dev = mat.device
    if dim in (0, -2):
        src = (torch.arange(n_rows, device=dev).unsqueeze(-1) - shifts.reshape(1, n_cols)) % n_rows   # [rows, cols]
        return torch.gather(mat, -2, src.expand(*batch, n_rows, n_cols))
    if dim in (1, -1):
        src = (torch.arange(n_cols, device=dev).unsqueeze(0) - shifts.reshape(n_rows, 1)) % n_cols
        return torch.gather(mat, -1, src.expand(*batch, n_rows, n_cols))
    raise NotImplementedError(f"dim {dim} is not supported.")


def _make_gammas_tensor(gamma: torch.Tensor, T: int, rolling_gamma: bool) -> torch.Tensor:
    """Decay table ``[N, T, T + 1, 1]`` for ``gamma`` of shape ``[*B, T, D]``, ``N = prod(B) * D``
    (reference: utils.py:131-182).  Column 0 is 1; ``rolling_gamma``: entry ``(t, 1 + j)`` is the gamma of
    step ``t + j`` (wrapping past the end -- the values it multiplies are zero-padded there); otherwise the
    gamma of step ``t`` for every ``j``."""
    g = gamma.transpose(-2, -1).reshape(-1, T)                               # [N, T]
    ones = torch.ones(g.shape[0], T, 1, dtype=g.dtype, device=g.device)
    if rolling_gamma:
        step = (torch.arange(T, device=g.device).unsqueeze(-1) + torch.arange(T, device=g.device)) % T  # [t, j] -> t + j
        body = g[:, step]                                                    # [N, T, T]
    else:
        body = g.unsqueeze(-1).expand(g.shape[0], T, T)
    return torch.cat([ones, body], -1).unsqueeze(-1)


def _flatten_batch(tensor, time_dim: int = -1):
    """``[*B, T, *F] -> [prod(B) * T, *F]`` (reference: utils.py:185-193): rollouts are concatenated, every
    row's last step counting as a trajectory end."""
    return tensor.flatten(0, time_dim)


def _get_num_per_traj(done: torch.Tensor) -> torch.Tensor:
    """Lengths of the trajectories of a ``[*B, T]`` done mask once the rows are concatenated, the last step of
    every row closing a trajectory (reference: utils.py:196-211)."""
    flat = done.reshape(-1, done.shape[-1]).clone()
    flat[:, -1] = True
    ends = torch.nonzero(flat.reshape(-1), as_tuple=False).squeeze(-1) + 1   # exclusive end positions
    return torch.diff(ends, prepend=ends.new_zeros(1))


def _length_mask(splits: torch.Tensor, width, device) -> torch.Tensor:
    return torch.arange(int(width), device=device).unsqueeze(0) < splits.to(device).unsqueeze(1)


def _split_and_pad_sequence(tensor: Union[torch.Tensor, Mapping], splits: torch.Tensor, return_mask: bool = False,
                            time_dim: int = -1):
    """Cut the concatenated rollouts ``[*B, T, *F]`` into trajectories of lengths ``splits`` and zero-pad them
    to ``[len(splits), max(splits), *F]`` (reference: utils.py:214-317).  A mapping of tensors (what the
    reference accepts as a tensordict) is processed entry by entry.  ``return_mask`` also returns the
    ``[len(splits), max(splits)]`` validity mask."""
    if isinstance(tensor, Mapping):
        first = next(iter(tensor.values()))
        mask = _length_mask(splits, splits.max(), first.device)
        out = {k: _split_and_pad_sequence(v, splits, False, time_dim) for k, v in tensor.items()}
        return (out, mask) if return_mask else out
    mask = _length_mask(splits, splits.max(), tensor.device)
    flat = _flatten_batch(tensor, time_dim=time_dim)
    padded = flat.new_zeros(*mask.shape, *flat.shape[1:])
    padded[mask] = flat                                                      # row-major: trajectory after trajectory
    return (padded, mask) if return_mask else padded


def _inv_pad_sequence(tensor: Union[torch.Tensor, Mapping], splits: torch.Tensor, mask: torch.Tensor = None):
    """Inverse of :func:`_split_and_pad_sequence`: the valid steps, concatenated (reference: utils.py:320-354;
    a single trajectory is returned as is, as there)."""
    if splits.numel() == 1:
        return tensor
    if isinstance(tensor, Mapping):
        return {k: _inv_pad_sequence(v, splits, mask) for k, v in tensor.items()}
    if mask is None:
        mask = _length_mask(splits, tensor.shape[-1] if tensor.dim() == 2 else tensor.shape[1], tensor.device)
    return tensor[mask]


def _get_num_per_traj_init(is_init: torch.Tensor) -> torch.Tensor:
    """Like :func:`_get_num_per_traj` with an ``is_init`` mark: a trajectory ends where the next step is an
    initial one (reference: utils.py:357-361)."""
    done = torch.zeros_like(is_init)
    done[..., :-1] = is_init[..., 1:]
    return _get_num_per_traj(done)
