"""MI355X-native value-estimator modules: ``GAE``, ``TDLambdaEstimator``, ``VTrace`` (and
``TD0Estimator`` / ``TD1Estimator``).

Drop-in for ``torchrl.objectives.value.advantages`` (reference:
torchrl/objectives/value/advantages.py): same constructor keywords, key names and ``set_keys``
rules (:109-152, 303-336), time-dim resolution (:409-423), ``gamma ** steps_to_next_obs``
(:1405-1407), ``terminated`` defaulting to ``done`` (:1434), normalisation formulas and epsilons
(:1459-1463 GAE 1e-4, :1828-1832 V-trace 1e-5, :1160-1165 TD(lambda) rewards 1e-4), output keys
``advantage`` / ``value_target``, keyword dispatch (``module(obs=..., next_reward=...)`` returns
``(advantage, value_target)``).  The arithmetic is one C-ABI call into the HIP library
(include/trl_value.h) through :mod:`torchrl_amd.objectives.value.functional`.

``tensordict`` is a third-party dependency of the reference that is not vendored; these modules
accept a real ``TensorDict`` when that package is installed and otherwise the small
:class:`RolloutData` container below (nested-key ``get`` / ``set`` over plain dicts), which is also
what the keyword-dispatch path builds.

Differences that are deliberate:
  * ``vectorized`` is accepted and stored but both settings run the same exact scan kernels (the
    reference's "vec" path is a truncated-filter approximation of its loop, functional.py:185-210).
  * ``GAE(average_gae=True)`` runs GAE and the normalisation as ONE launch (FUSED kernel family);
    ``process_group`` (extension) all-reduces the fp64 moments when the batch is sharded over ranks.
  * module buffers (``gamma``, ``lmbda``, thresholds) are mirrored on the host, so a forward pass
    never synchronises the device to read a scalar.
"""
from __future__ import annotations

import abc
from collections.abc import Mapping
import warnings
from dataclasses import asdict, dataclass
from typing import Callable, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from torchrl_amd.objectives.value import functional as F

NestedKey = Union[str, Tuple[str, ...]]

try:  # optional: real tensordicts are accepted when the package exists
    from tensordict import TensorDictBase as _TensorDictBase  # type: ignore
except Exception:  # pragma: no cover - tensordict is absent in the build image
    _TensorDictBase = None


_UNRAVELLED = {}   # nested keys are a handful of constants per estimator: flattening them is memoised


def _unravel_key(key: NestedKey) -> NestedKey:
    if key.__class__ is str:
        return key
    try:
        hit = _UNRAVELLED.get(key)
    except TypeError:   # unhashable (a list): flatten without the table
        hit = None
        key = tuple(key)
    if hit is not None:
        return hit
    out = []
    for k in key:
        k = _unravel_key(k)
        out.extend(k if isinstance(k, tuple) else (k,))
    res = out[0] if len(out) == 1 else tuple(out)
    try:
        if len(_UNRAVELLED) < 4096:
            _UNRAVELLED[key] = res
    except TypeError:
        pass
    return res


class RolloutData:
    """Minimal stand-in for a ``TensorDict`` of batch size ``[*B, T]``: nested-key ``get`` / ``set``
    over plain dicts, ``batch_size`` / ``batch_dims`` / ``names``.  Only what the estimators use."""

    def __init__(self, source: Optional[dict] = None, batch_size=None, names=None):
        self._d = {}
        for k, v in (source or {}).items():
            self.set(k, v)
        self.batch_size = torch.Size(batch_size) if batch_size is not None else self._infer_batch_size()
        self.names = list(names) if names is not None else [None] * len(self.batch_size)

    # -- structure ---------------------------------------------------------------------------
    def _leaves(self, d=None):
        for v in (self._d if d is None else d).values():
            if isinstance(v, dict):
                yield from self._leaves(v)
            elif isinstance(v, torch.Tensor):
                yield v

    def _infer_batch_size(self):
        shapes = [tuple(t.shape) for t in self._leaves()]
        if not shapes:
            return torch.Size([])
        n = min(len(s) for s in shapes)
        common = []
        for i in range(n):
            if len({s[i] for s in shapes}) != 1:
                break
            common.append(shapes[0][i])
        return torch.Size(common)

    @property
    def batch_dims(self) -> int:
        return len(self.batch_size)

    ndim = batch_dims

    @property
    def shape(self):
        return self.batch_size

    def _has_names(self) -> bool:
        return any(n is not None for n in self.names)

    # -- access ------------------------------------------------------------------------------
    _MISSING = object()

    def get(self, key: NestedKey, default=_MISSING):
        key = _unravel_key(key)
        path = (key,) if isinstance(key, str) else key
        d = self._d
        for k in path:
            if not isinstance(d, dict) or k not in d:
                if default is RolloutData._MISSING:
                    raise KeyError(f"key {key} not found in RolloutData with keys {sorted(map(str, self.keys(True)))}")
                return default
            d = d[k]
        return RolloutData._wrap(d, self) if isinstance(d, dict) else d

    @staticmethod
    def _wrap(d, parent):
        sub = RolloutData.__new__(RolloutData)
        sub._d, sub.batch_size, sub.names = d, parent.batch_size, parent.names
        return sub

    def set(self, key: NestedKey, value):
        key = _unravel_key(key)
        path = (key,) if isinstance(key, str) else key
        d = self._d
        for k in path[:-1]:
            d = d.setdefault(k, {})
        if isinstance(value, RolloutData):
            value = value._d
        elif isinstance(value, dict):
            value = RolloutData(value, batch_size=())._d
        d[path[-1]] = value
        return self

    def keys(self, include_nested: bool = False):
        def walk(d, prefix):
            for k, v in d.items():
                if isinstance(v, dict) and include_nested:
                    yield from walk(v, prefix + (k,))
                else:
                    yield (prefix + (k,)) if prefix else k

        return list(walk(self._d, ()))

    def __contains__(self, key):
        return self.get(key, None) is not None

    def __getitem__(self, key):
        return self.get(key)

    def __setitem__(self, key, value):
        self.set(key, value)

    def select(self, *keys, strict: bool = True):
        out = RolloutData({}, batch_size=self.batch_size, names=self.names)
        for k in keys:
            v = self.get(k, None)
            if v is None:
                if strict:
                    raise KeyError(k)
                continue
            out.set(k, v)
        return out

    def to_dict(self):
        return self._d


class KeyedModule(nn.Module):
    """Reads ``in_keys`` from a :class:`RolloutData` / tensordict, calls ``module`` on the tensors and
    writes the result(s) under ``out_keys`` -- the role ``tensordict.nn.TensorDictModule`` plays for
    the reference's value networks (e.g. ``KeyedModule(nn.Linear(3, 1), ["obs"], ["state_value"])``)."""

    def __init__(self, module: Callable, in_keys, out_keys):
        super().__init__()
        self.module = module
        self.in_keys = [_unravel_key(k) for k in in_keys]
        self.out_keys = [_unravel_key(k) for k in out_keys]

    def forward(self, data):
        out = self.module(*[data.get(k) for k in self.in_keys])
        if isinstance(out, Tensor):
            out = (out,)
        for k, v in zip(self.out_keys, out):
            data.set(k, v)
        return data


def _is_collection(x) -> bool:
    return isinstance(x, RolloutData) or (_TensorDictBase is not None and isinstance(x, _TensorDictBase))


def _from_kwargs(kwargs: dict) -> RolloutData:
    """``module(obs=..., next_reward=..., next_done=...)``: ``next_<key>`` -> ``("next", <key>)``
    (the naming of ``tensordict.nn.dispatch``; reference test/test_cost.py:15501-15508)."""
    src = {}
    for k, v in kwargs.items():
        if k.startswith("next_"):
            src.setdefault("next", {})[k[len("next_"):]] = v
        else:
            src[k] = v
    return RolloutData(src)


class ValueEstimatorBase(nn.Module, metaclass=abc.ABCMeta):
    """Abstract parent of the value-estimator modules (reference: advantages.py:97-516)."""

    @dataclass
    class _AcceptedKeys:
        """Configurable keys and their defaults (reference: advantages.py:109-152)."""

        advantage: NestedKey = "advantage"
        value_target: NestedKey = "value_target"
        value: NestedKey = "state_value"
        reward: NestedKey = "reward"
        done: NestedKey = "done"
        terminated: NestedKey = "terminated"
        steps_to_next_obs: NestedKey = "steps_to_next_obs"
        sample_log_prob: NestedKey = "sample_log_prob"

    default_keys = _AcceptedKeys
    value_network: Optional[Callable]

    def __init__(self, *, value_network, shifted: bool = False, differentiable: bool = False,
                 skip_existing: Optional[bool] = None, advantage_key: NestedKey = None,
                 value_target_key: NestedKey = None, value_key: NestedKey = None, device=None):
        super().__init__()
        if device is None:
            device = getattr(torch, "get_default_device", lambda: torch.device("cpu"))()
        self._device = device
        self._tensor_keys = None
        self.differentiable = differentiable
        self.skip_existing = skip_existing
        self.__dict__["value_network"] = value_network  # not registered as a submodule (advantages.py:249)
        self.dep_keys = {}
        self.shifted = shifted
        self._host_cache = {}
        self._float_cache = {}
        self._coef_cache = None
        for name, val in (("advantage_key", advantage_key), ("value_target_key", value_target_key), ("value_key", value_key)):
            if val is not None:
                raise RuntimeError(
                    f"Setting '{name}' via constructor is deprecated, use .set_keys({name}='some_key') instead.",
                )

    # -- keys (advantages.py:266-336) ----------------------------------------------------------
    @property
    def tensor_keys(self):
        if self._tensor_keys is None:
            self.set_keys()
        return self._tensor_keys

    advantage_key = property(lambda self: self.tensor_keys.advantage)
    value_key = property(lambda self: self.tensor_keys.value)
    value_target_key = property(lambda self: self.tensor_keys.value_target)
    reward_key = property(lambda self: self.tensor_keys.reward)
    done_key = property(lambda self: self.tensor_keys.done)
    terminated_key = property(lambda self: self.tensor_keys.terminated)
    steps_to_next_obs_key = property(lambda self: self.tensor_keys.steps_to_next_obs)
    sample_log_prob_key = property(lambda self: self.tensor_keys.sample_log_prob)

    @property
    def in_keys(self):
        try:
            net_keys = list(self.value_network.in_keys)
        except AttributeError:
            return []
        return (net_keys + [("next", self.tensor_keys.reward), ("next", self.tensor_keys.done),
                            ("next", self.tensor_keys.terminated)] + [("next", k) for k in net_keys])

    @property
    def out_keys(self):
        return [self.tensor_keys.advantage, self.tensor_keys.value_target]

    @staticmethod
    def _normalise_key(value):
        """A key name in canonical form: str / tuple (nested) / list of those; anything else is rejected with the
        reference's messages (advantages.py:306-315)."""
        if isinstance(value, list):
            return [_unravel_key(k) for k in value]
        if isinstance(value, (str, tuple)):
            return _unravel_key(value)
        if value is None:
            raise ValueError("tensordict keys cannot be None")
        raise ValueError(f"key name must be of type NestedKey (Union[str, Tuple[str]]) but got {type(value)}")

    def set_keys(self, **kwargs) -> None:
        """Rename tensordict entries the estimator reads / writes (behaviour of the reference's ``set_keys``,
        advantages.py:303-336: unknown names raise ``KeyError``, a ``value`` key must be one of the value network's
        ``out_keys``; the new names are merged over the current configuration)."""
        accepted = self._AcceptedKeys.__dict__
        net_out = getattr(self.value_network, "out_keys", None)
        renamed = {}
        for name, raw in kwargs.items():
            key = self._normalise_key(raw)
            if name not in accepted:
                raise KeyError(f"{name} is not an accepted tensordict key for advantages")
            if name == "value" and net_out is not None and key not in net_out:
                raise KeyError(f"value key '{key}' not found in value network out_keys {net_out}")
            renamed[name] = key
        if self._tensor_keys is not None:
            current = asdict(self._tensor_keys)
        else:
            current = {**asdict(self.default_keys()), **self.dep_keys}
        self._tensor_keys = self._AcceptedKeys(**{**current, **renamed})

    is_functional = property(lambda self: False)
    is_stateless = property(lambda self: False)

    # -- helpers ---------------------------------------------------------------------------------
    def _host(self, t: Tensor) -> Tensor:
        """CPU mirror of a scalar buffer (same dtype, so the reference's rounding rules still apply);
        refreshed when the buffer is replaced or modified in place.  Avoids a device sync per call."""
        if not isinstance(t, Tensor) or t.device.type == "cpu" or t.numel() != 1:
            return t
        key = id(t)
        hit = self._host_cache.get(key)
        if hit is None or hit[0] != t._version or hit[1]() is not t:
            import weakref

            hit = (t._version, weakref.ref(t), t.detach().cpu())
            # one entry per buffer (gamma, lmbda, thresholds are looked up in turn); entries of dead tensors are dropped
            for k in [k for k, h in self._host_cache.items() if h[1]() is None]:
                del self._host_cache[k]
            self._host_cache[key] = hit
        return hit[2]

    def _host_float(self, t):
        """``float`` of a scalar buffer's host mirror, cached per mirror and version (``float(tensor)`` costs a microsecond);
        anything that is not a one-element tensor is returned unchanged."""
        if not isinstance(t, Tensor) or t.numel() != 1:
            return t
        h = self._host(t)
        c = self._float_cache.get(id(h))
        if c is not None and c[0] is h and c[1] == h._version:
            return c[2]
        if len(self._float_cache) > 16:
            self._float_cache.clear()
        v = float(h)
        self._float_cache[id(h)] = (h, h._version, v)   # holds h: its id cannot be reused while the entry lives
        return v

    def _gae_coef(self, gamma, lmbda):
        """Pre-rounded ``(gamma, lmbda*gamma, dtype code)`` of two scalar host mirrors (functional.scalar_gae_coef), cached
        until a buffer is replaced, modified or the default dtype changes; None for per-element coefficients."""
        if not (isinstance(gamma, Tensor) and isinstance(lmbda, Tensor)) or gamma.device.type != "cpu" or lmbda.device.type != "cpu":
            return None
        c = self._coef_cache
        ddt = torch.get_default_dtype()
        if c is not None and c[0] is gamma and c[1] is lmbda and c[2] == (gamma._version, lmbda._version, ddt):
            return c[3]
        coef = F.scalar_gae_coef(gamma, lmbda)
        self._coef_cache = (gamma, lmbda, (gamma._version, lmbda._version, ddt), coef)
        return coef

    def _get_time_dim(self, time_dim: Optional[int], data) -> int:
        """explicit -> attribute -> dim named "time" -> last batch dim (advantages.py:409-423)."""
        if time_dim is not None:
            return data.ndim + time_dim if time_dim < 0 else time_dim
        attr = getattr(self, "time_dim", None)
        if attr is not None:
            return data.ndim + attr if attr < 0 else attr
        if data._has_names():
            for i, name in enumerate(data.names):
                if name == "time":
                    return i
        return data.ndim - 1

    def _call_value_nets(self, data, params, next_params, single_call: bool, value_key, detach_next: bool):
        """Value at t and t+1 (advantages.py:425-516).  ``single_call`` (``shifted=True``) evaluates the
        network once on the T+1 steps ``[data, last step of data["next"]]``; otherwise it is called on
        ``data`` and on ``data["next"]`` (the reference vmaps the two calls; the result is the same)."""
        net = self.value_network
        in_keys = list(getattr(net, "in_keys", []))
        nxt = data.get("next")
        if single_call:
            if next_params is not None and next_params is not params:
                raise ValueError("the value at t and t+1 cannot be retrieved in a single call without recurring to vmap when both "
                                 "params and next params are passed.")
            ndim = None
            if data._has_names():
                for i, name in enumerate(data.names):
                    if name == "time":
                        ndim = i + 1
            if ndim is None:
                ndim = data.ndim
            tdim = ndim - 1
            T = data.batch_size[tdim]
            cat = RolloutData({}, batch_size=())
            for k in in_keys:
                a, b = data.get(k), nxt.get(k)
                cat.set(k, torch.cat([a, b.narrow(tdim, T - 1, 1)], tdim))
            cat.batch_size = torch.Size(list(data.batch_size[:tdim]) + [T + 1] + list(data.batch_size[tdim + 1:]))
            cat.names = data.names
            out = self._run_net(net, cat, params)
            value_est = out.get(value_key) if _is_collection(out) else out
            value, value_ = value_est.narrow(tdim, 0, T), value_est.narrow(tdim, 1, T)
        else:
            if (params is not None) ^ (next_params is not None):
                raise ValueError("params and next_params must be either both provided or not.")
            out = self._run_net(net, data, params)
            value = out.get(value_key) if _is_collection(out) else out
            nxt_in = nxt if _is_collection(nxt) else data.get("next")
            out_ = self._run_net(net, nxt_in, next_params)
            value_ = out_.get(value_key) if _is_collection(out_) else out_
        data.set(value_key, value)
        data.set(("next", value_key), value_)
        if detach_next:
            value_ = value_.detach()
        return value, value_

    @staticmethod
    def _run_net(net, data, params):
        """``net(data)`` under functional parameters: a tensordict of parameters (``params.to_module``, the reference's form,
        advantages.py:485-489) or -- without the tensordict package -- a mapping ``name -> tensor`` in ``net.named_parameters()``
        naming, applied with ``torch.func.functional_call``."""
        if params is None:
            return net(data)
        if hasattr(params, "to_module"):
            with params.to_module(net):
                return net(data)
        if isinstance(params, Mapping) and isinstance(net, nn.Module):
            return torch.func.functional_call(net, dict(params), (data,))
        raise NotImplementedError("functional parameters: a tensordict of parameters (params.to_module) or, for an nn.Module value "
                                  "network, a mapping name -> tensor")

    def _values(self, data, params, target_params):
        if self.value_network is not None:
            if params is not None:
                if hasattr(params, "to_module"):
                    params = params.detach()
                    if target_params is None:
                        target_params = params.clone(False)
                elif isinstance(params, Mapping):
                    params = {k: v.detach() for k, v in params.items()}
                    if target_params is None:
                        target_params = params
                else:
                    self._run_net(self.value_network, data, params)  # raises: not a form of functional parameters
            req = [p.requires_grad for p in self._net_params()] if (params is None and target_params is None) else None
            try:
                if req is not None:  # hold_out_net: no grads are assigned to the value net's params
                    for p in self._net_params():
                        p.requires_grad_(False)
                return self._call_value_nets(data, params, target_params, self.shifted, self.tensor_keys.value, True)
            finally:
                if req is not None:
                    for p, r in zip(self._net_params(), req):
                        p.requires_grad_(r)
        return data.get(self.tensor_keys.value), data.get(("next", self.tensor_keys.value))

    def _net_params(self):
        net = self.value_network
        return list(net.parameters()) if isinstance(net, nn.Module) else []

    def _gamma_for(self, data, reward):
        gamma = self.gamma
        if gamma.device != reward.device and gamma.numel() > 1:
            self.gamma = gamma = gamma.to(reward.device)
        steps = data.get(self.tensor_keys.steps_to_next_obs, None)
        if steps is not None:
            return gamma.to(reward.device) ** steps.view_as(reward)
        return self._host(gamma)

    def _entry(self, args, kwargs):
        """Resolve (data, dispatch_mode) from a tensordict-like positional arg or dispatch kwargs."""
        if args and _is_collection(args[0]):
            return args[0], False
        if args and isinstance(args[0], dict):
            return RolloutData(args[0]), False
        tensor_kwargs = {k: v for k, v in kwargs.items() if isinstance(v, Tensor)}
        for k in tensor_kwargs:
            kwargs.pop(k)
        return _from_kwargs(tensor_kwargs), True

    def _finish(self, data, dispatch_mode, adv, value_target):
        data.set(self.tensor_keys.advantage, adv)
        data.set(self.tensor_keys.value_target, value_target)
        return (adv, value_target) if dispatch_mode else data

    def _plain_inputs(self, data):
        """``(dict, value, next_value, reward, done, terminated, advantage key, value_target key)`` of a :class:`RolloutData` in the
        plain layout -- string keys, values already stored, unnamed dims, no ``steps_to_next_obs``, nothing that would record a
        graph -- read with plain dict look-ups; None when it is anything else (the caller then goes the long way, which also
        raises what is to be raised).  Lets the modules' ``forward`` cost less than the launch it issues."""
        tk = self._tensor_keys
        if tk is None:
            tk = self.tensor_keys
        vk, rk, dk, tmk, ak, vtk, sk = tk.value, tk.reward, tk.done, tk.terminated, tk.advantage, tk.value_target, tk.steps_to_next_obs
        d = data._d
        nxt = d.get("next")
        if (nxt.__class__ is not dict or vk.__class__ is not str or rk.__class__ is not str or dk.__class__ is not str
                or tmk.__class__ is not str or ak.__class__ is not str or vtk.__class__ is not str or sk.__class__ is not str or sk in d):
            return None
        names = data.names
        if len(data.batch_size) < 1 or names.count(None) != len(names):
            return None
        try:
            value, next_value, reward, done = d[vk], nxt[vk], nxt[rk], nxt[dk]
            if torch.is_grad_enabled() and (value.requires_grad or next_value.requires_grad or reward.requires_grad):
                return None
        except (KeyError, TypeError, AttributeError):
            return None
        return d, value, next_value, reward, done, nxt.get(tmk, done), ak, vtk

    def _skip(self, data) -> bool:
        return bool(self.skip_existing) and all(data.get(k, None) is not None for k in self.out_keys)

    @staticmethod
    def _check_dims(data):
        if data.batch_dims < 1:
            raise RuntimeError(
                f"Expected input tensordict to have at least one dimension, got tensordict.batch_size = {data.batch_size}")

    @abc.abstractmethod
    def forward(self, *args, **kwargs):
        ...

    def value_estimate(self, tensordict, target_params=None, next_value=None, **kwargs):
        raise NotImplementedError


def _scalar_buffer(x, device):
    return x.to(device) if isinstance(x, Tensor) else torch.tensor(x, device=device)


class GAE(ValueEstimatorBase):
    """Generalized advantage estimate module (reference: advantages.py:1196-1535).

    Keyword Args: ``gamma``, ``lmbda``, ``value_network`` (``None`` to read ``state_value`` and
    ``("next", "state_value")`` from the data), ``average_gae`` (normalise with eps 1e-4),
    ``differentiable``, ``vectorized``, ``skip_existing``, ``shifted``, ``device``, ``time_dim``;
    extensions for a batch sharded over ranks (one process per GPU): ``process_group`` (RCCL
    all-reduce of the fp64 moments) or ``stats_exchange`` (a :class:`StatsExchange`: the moments
    cross the GPUs inside the launch, intra-node).
    """

    def __init__(self, *, gamma, lmbda, value_network, average_gae: bool = False, differentiable: bool = False,
                 vectorized: Optional[bool] = None, skip_existing: Optional[bool] = None, advantage_key: NestedKey = None,
                 value_target_key: NestedKey = None, value_key: NestedKey = None, shifted: bool = False, device=None,
                 time_dim: Optional[int] = None, process_group=None, stats_exchange=None):
        super().__init__(shifted=shifted, value_network=value_network, differentiable=differentiable,
                         advantage_key=advantage_key, value_target_key=value_target_key, value_key=value_key,
                         skip_existing=skip_existing, device=device)
        self.register_buffer("gamma", _scalar_buffer(gamma, self._device))
        self.register_buffer("lmbda", _scalar_buffer(lmbda, self._device))
        self.average_gae = average_gae
        self.vectorized = vectorized
        self.time_dim = time_dim
        self.process_group = process_group
        self.stats_exchange = stats_exchange
        self._short_coef = None

    def _inputs(self, data, params, target_params, time_dim):
        self._check_dims(data)
        reward = data.get(("next", self.tensor_keys.reward))
        gamma = self._gamma_for(data, reward)
        lmbda = self._host(self.lmbda) if self.lmbda.numel() == 1 else self.lmbda.to(reward.device)
        value, next_value = self._values(data, params, target_params)
        done = data.get(("next", self.tensor_keys.done))
        terminated = data.get(("next", self.tensor_keys.terminated), done)
        return gamma, lmbda, value, next_value, reward, done, terminated, self._get_time_dim(time_dim, data)

    def _short_forward(self, data):
        """The whole of ``forward`` for the plain case (``_plain_inputs``) with scalar ``gamma`` / ``lmbda`` buffers: one hop to the
        C ABI.  None: not that case."""
        got = self._plain_inputs(data)
        if got is None:
            return None
        d, value, next_value, reward, done, terminated, ak, vtk = got
        buffers = self._buffers
        g, lm = buffers["gamma"], buffers["lmbda"]
        c = self._short_coef
        if c is None or c[0] is not g or c[1] is not lm or c[2] != g._version or c[3] != lm._version or c[4] is not torch.get_default_dtype():
            coef = self._gae_coef(self._host(g), self._host(lm)) if g.numel() == 1 and lm.numel() == 1 else None
            c = self._short_coef = (g, lm, g._version, lm._version, torch.get_default_dtype(), coef)
        coef = c[5]
        if coef is None:
            return None
        out = F._gae_fast(coef, None, value, next_value, reward, done, terminated, 1e-4 if self.average_gae else None, F.ALGO_AUTO,
                          len(data.batch_size) - 1)
        if out is None:
            return None
        d[ak], d[vtk] = out
        return data

    def forward(self, *args, params=None, target_params=None, time_dim: Optional[int] = None, **kwargs):
        if (len(args) == 1 and args[0].__class__ is RolloutData and not kwargs and params is None and target_params is None and time_dim is None
                and self.time_dim is None and self.value_network is None and not self.differentiable and not self.skip_existing
                and self.process_group is None and self.stats_exchange is None):
            done_data = self._short_forward(args[0])
            if done_data is not None:
                return done_data
        data, dispatch_mode = self._entry(args, kwargs)
        if self._skip(data):
            return data
        with torch.set_grad_enabled(self.differentiable):
            gamma, lmbda, value, next_value, reward, done, terminated, td = self._inputs(data, params, target_params, time_dim)
            fast = None
            if self.process_group is None and (self.stats_exchange is None or self.differentiable):
                coef = self._gae_coef(gamma, lmbda)
                if coef is not None:   # scalar buffers, dense tensors, no autograd: one hop to the C ABI (functional._gae_fast)
                    fast = F._gae_fast(coef, None, value, next_value, reward, done, terminated, 1e-4 if self.average_gae else None,
                                       F.ALGO_AUTO, td)
            if fast is not None:
                adv, value_target = fast
            elif self.average_gae:
                adv, value_target = F.gae_normalized(gamma, lmbda, value, next_value, reward, done, terminated, eps=1e-4,
                                                     time_dim=td, group=self.process_group,
                                                     exchange=None if self.differentiable else self.stats_exchange)
            else:
                fn = F.vec_generalized_advantage_estimate if self.vectorized else F.generalized_advantage_estimate
                adv, value_target = fn(gamma, lmbda, value, next_value, reward, done=done, terminated=terminated, time_dim=td)
        return self._finish(data, dispatch_mode, adv, value_target)

    def value_estimate(self, tensordict, params=None, target_params=None, time_dim: Optional[int] = None, **kwargs):
        data = tensordict if _is_collection(tensordict) else RolloutData(tensordict)
        gamma, lmbda, value, next_value, reward, done, terminated, td = self._inputs(data, params, target_params, time_dim)
        _, value_target = F.vec_generalized_advantage_estimate(gamma, lmbda, value, next_value, reward, done=done,
                                                               terminated=terminated, time_dim=td)
        return value_target


class TDLambdaEstimator(ValueEstimatorBase):
    r"""TD(:math:`\lambda`) advantage module (reference: advantages.py:953-1193)."""

    def __init__(self, *, gamma, lmbda, value_network, average_rewards: bool = False, differentiable: bool = False,
                 vectorized: bool = True, skip_existing: Optional[bool] = None, advantage_key: NestedKey = None,
                 value_target_key: NestedKey = None, value_key: NestedKey = None, shifted: bool = False, device=None,
                 time_dim: Optional[int] = None):
        super().__init__(value_network=value_network, differentiable=differentiable, advantage_key=advantage_key,
                         value_target_key=value_target_key, value_key=value_key, skip_existing=skip_existing,
                         shifted=shifted, device=device)
        self.register_buffer("gamma", torch.tensor(gamma, device=self._device))
        self.register_buffer("lmbda", torch.tensor(lmbda, device=self._device))
        self.average_rewards = average_rewards
        self.vectorized = vectorized
        self.time_dim = time_dim
        self._lmbda_one = False

    def _short_forward(self, data):
        """``forward`` for the plain case (``_plain_inputs``): the buffers as python floats where those act exactly like the
        buffers, the return estimate through the one-hop host path.  None: not that case."""
        got = self._plain_inputs(data)
        if got is None:
            return None
        d, value, next_value, reward, done, terminated, ak, vtk = got
        dt = next_value.dtype
        buffers = self._buffers
        gamma = buffers["gamma"]
        if gamma.numel() != 1 or (self.vectorized and torch.promote_types(dt, gamma.dtype) != dt):
            return None
        lmbda = 1.0
        if not self._lmbda_one:
            lmbda = buffers["lmbda"]
            if lmbda.numel() != 1 or (self.vectorized and torch.promote_types(dt, lmbda.dtype) != dt):
                return None
            lmbda = self._host_float(lmbda)
        fn = F.vec_td_lambda_return_estimate if self.vectorized else F.td_lambda_return_estimate
        value_target = fn(self._host_float(gamma), lmbda, next_value, reward, done=done, terminated=terminated,
                          time_dim=len(data.batch_size) - 1)
        d[ak] = value_target - value
        d[vtk] = value_target
        return data

    def forward(self, *args, params=None, target_params=None, **kwargs):
        if (len(args) == 1 and args[0].__class__ is RolloutData and not kwargs and params is None and target_params is None
                and self.time_dim is None and self.value_network is None and not self.differentiable and not self.skip_existing
                and not self.average_rewards):
            done_data = self._short_forward(args[0])
            if done_data is not None:
                return done_data
        data, dispatch_mode = self._entry(args, kwargs)
        if self._skip(data):
            return data
        with torch.set_grad_enabled(self.differentiable):
            self._check_dims(data)
            value, next_value = self._values(data, params, target_params)
            value_target = self.value_estimate(data, next_value=next_value)
            adv = value_target - value
        return self._finish(data, dispatch_mode, adv, value_target)

    def value_estimate(self, tensordict, target_params=None, next_value=None, time_dim: Optional[int] = None, **kwargs):
        data = tensordict if _is_collection(tensordict) else RolloutData(tensordict)
        reward = data.get(("next", self.tensor_keys.reward))
        gamma = self._gamma_for(data, reward)
        lmbda = 1.0 if self._lmbda_one else self._host(self.lmbda)
        if self.average_rewards:
            reward = F.standardize(reward, 1e-4)
            # reference quirk kept: the normalised reward is stored under ("next", steps_to_next_obs) (:1163-1165)
            data.set(("next", self.tensor_keys.steps_to_next_obs), reward)
        if next_value is None:
            if self.value_network is None:
                next_value = data.get(("next", self.tensor_keys.value))
            else:
                out = self._run_net(self.value_network, data.get("next"), target_params)
                next_value = out.get(self.tensor_keys.value) if _is_collection(out) else out
        done = data.get(("next", self.tensor_keys.done))
        terminated = data.get(("next", self.tensor_keys.terminated), done)
        td = self._get_time_dim(time_dim, data)
        fn = F.vec_td_lambda_return_estimate if self.vectorized else F.td_lambda_return_estimate
        # a one-element buffer acts exactly like its python float when it does not widen the result (functional._td_coef; the
        # vec variant's output dtype is promoted by coefficient TENSORS): floats take the one-hop host path
        if isinstance(gamma, Tensor) and gamma.numel() == 1 and (
                not self.vectorized or torch.promote_types(next_value.dtype, gamma.dtype) == next_value.dtype):
            gamma = self._host_float(gamma)
        if isinstance(lmbda, Tensor) and lmbda.numel() == 1 and (
                not self.vectorized or torch.promote_types(next_value.dtype, lmbda.dtype) == next_value.dtype):
            lmbda = self._host_float(lmbda)
        return fn(gamma, lmbda, next_value, reward, done=done, terminated=terminated, time_dim=td)


class TD1Estimator(TDLambdaEstimator):
    """TD(1) (bootstrapped discounted return) advantage module (reference: advantages.py:734-950):
    TD(lambda) at ``lmbda = 1``, which is how the reference's vectorized path computes it
    (functional.py:691-700)."""

    def __init__(self, *, gamma, value_network, average_rewards: bool = False, differentiable: bool = False,
                 skip_existing: Optional[bool] = None, advantage_key: NestedKey = None, value_target_key: NestedKey = None,
                 value_key: NestedKey = None, shifted: bool = False, device=None, time_dim: Optional[int] = None):
        super().__init__(gamma=gamma, lmbda=1.0, value_network=value_network, average_rewards=average_rewards,
                         differentiable=differentiable, vectorized=True, skip_existing=skip_existing,
                         advantage_key=advantage_key, value_target_key=value_target_key, value_key=value_key,
                         shifted=shifted, device=device, time_dim=time_dim)
        del self.lmbda  # TD1Estimator registers gamma only (advantages.py:800)
        self._lmbda_one = True


class TD0Estimator(ValueEstimatorBase):
    """TD(0) advantage module (reference: advantages.py:519-731); elementwise, no scan."""

    def __init__(self, *, gamma, value_network, shifted: bool = False, average_rewards: bool = False,
                 differentiable: bool = False, skip_existing: Optional[bool] = None, advantage_key: NestedKey = None,
                 value_target_key: NestedKey = None, value_key: NestedKey = None, device=None):
        super().__init__(value_network=value_network, differentiable=differentiable, shifted=shifted,
                         advantage_key=advantage_key, value_target_key=value_target_key, value_key=value_key,
                         skip_existing=skip_existing, device=device)
        self.register_buffer("gamma", torch.tensor(gamma, device=self._device))
        self.average_rewards = average_rewards

    def forward(self, *args, params=None, target_params=None, **kwargs):
        data, dispatch_mode = self._entry(args, kwargs)
        if self._skip(data):
            return data
        with torch.set_grad_enabled(self.differentiable):
            self._check_dims(data)
            value, next_value = self._values(data, params, target_params)
            value_target = self.value_estimate(data, next_value=next_value)
            adv = value_target - value
        return self._finish(data, dispatch_mode, adv, value_target)

    def value_estimate(self, tensordict, target_params=None, next_value=None, **kwargs):
        data = tensordict if _is_collection(tensordict) else RolloutData(tensordict)
        reward = data.get(("next", self.tensor_keys.reward))
        gamma = self.gamma.to(reward.device)
        steps = data.get(self.tensor_keys.steps_to_next_obs, None)
        if steps is not None:
            gamma = gamma ** steps.view_as(reward)
        if self.average_rewards:
            reward = reward - reward.mean()
            reward = reward / reward.std().clamp_min(1e-5)
            data.set(("next", self.tensor_keys.reward), reward)
        if next_value is None:
            next_value = data.get(("next", self.tensor_keys.value))
        done = data.get(("next", self.tensor_keys.done))
        terminated = data.get(("next", self.tensor_keys.terminated), done)
        return F.td0_return_estimate(gamma=gamma, next_state_value=next_value, reward=reward, terminated=terminated)


class VTrace(ValueEstimatorBase):
    """V-Trace module (reference: advantages.py:1538-1837).

    ``actor_network``: a ``ProbabilisticTensorDictModule``-like object (``get_dist`` + ``in_keys``) when
    the tensordict stack is installed, or any callable ``data -> log-prob of the stored action under the
    current policy`` (tensor broadcastable to the value's shape)."""

    def __init__(self, *, gamma, actor_network, value_network, rho_thresh=1.0, c_thresh=1.0, average_adv: bool = False,
                 differentiable: bool = False, skip_existing: Optional[bool] = None, advantage_key: NestedKey = None,
                 value_target_key: NestedKey = None, value_key: NestedKey = None, shifted: bool = False, device=None,
                 time_dim: Optional[int] = None):
        super().__init__(shifted=shifted, value_network=value_network, differentiable=differentiable,
                         advantage_key=advantage_key, value_target_key=value_target_key, value_key=value_key,
                         skip_existing=skip_existing, device=device)
        gamma = _scalar_buffer(gamma, self._device)
        self.register_buffer("gamma", gamma)
        self.register_buffer("rho_thresh", _scalar_buffer(rho_thresh, self._device))
        self.register_buffer("c_thresh", _scalar_buffer(c_thresh, self._device))
        self.average_adv = average_adv
        self.__dict__["actor_network"] = actor_network
        self.time_dim = time_dim
        if gamma.shape != ():
            raise NotImplementedError("Per-value gamma is not supported yet. Gamma must be a scalar.")

    @property
    def in_keys(self):
        return super().in_keys + [self.tensor_keys.sample_log_prob]

    def _log_pi(self, data, value):
        actor = self.actor_network
        if hasattr(actor, "get_dist") and hasattr(actor, "_dist_sample"):  # reference path (:85-94)
            from tensordict.nn import set_composite_lp_aggregate  # type: ignore
            from tensordict.nn.probabilistic import interaction_type  # type: ignore

            dist = actor.get_dist(data.select(*actor.in_keys, strict=False))
            s = actor._dist_sample(dist, interaction_type=interaction_type())
            with set_composite_lp_aggregate(True):
                lp = dist.log_prob(s)
        else:
            lp = actor(data)
            if _is_collection(lp):
                lp = lp.get(self.tensor_keys.sample_log_prob)
        return lp.view_as(value)

    def forward(self, *args, params=None, target_params=None, time_dim: Optional[int] = None, **kwargs):
        data, dispatch_mode = self._entry(args, kwargs)
        if self._skip(data):
            return data
        with torch.set_grad_enabled(self.differentiable):
            self._check_dims(data)
            reward = data.get(("next", self.tensor_keys.reward))
            gamma = self._gamma_for(data, reward)
            value, next_value = self._values(data, params, target_params)
            log_mu = data.get(self.tensor_keys.sample_log_prob).view_as(value)
            with torch.no_grad():
                log_pi = self._log_pi(data, value)
            done = data.get(("next", self.tensor_keys.done))
            terminated = data.get(("next", self.tensor_keys.terminated))  # no default here (:1811)
            td = self._get_time_dim(time_dim, data)
            rho_thresh, c_thresh = self._host(self.rho_thresh), self._host(self.c_thresh)
            ddt = torch.get_default_dtype()
            # default-dtype scalar buffers act exactly like their python floats (functional._scalar_mask_coef / _thresh round
            # python numbers to the default dtype, which these already have): floats take the one-hop host path
            if all(isinstance(t, Tensor) and t.numel() == 1 and t.dtype == ddt for t in (gamma, rho_thresh, c_thresh)):
                gamma, rho_thresh, c_thresh = self._host_float(gamma), self._host_float(rho_thresh), self._host_float(c_thresh)
            adv, value_target = F.vtrace_advantage_estimate(gamma, log_pi, log_mu, value, next_value, reward, done, terminated,
                                                            rho_thresh=rho_thresh, c_thresh=c_thresh, time_dim=td)
            if self.average_adv:
                adv = F.standardize(adv, 1e-5)
        return self._finish(data, dispatch_mode, adv, value_target)


def _deprecated(new_cls, name):
    """Deprecated aliases (reference: advantages.py:1840-1856).  The reference rebuilds the classes with
    ``type(name, bases, dict)``, which breaks their zero-argument ``super()`` calls; plain subclasses
    keep the same constructor and emit the same warning."""

    def __init__(self, *args, **kwargs):
        warnings.warn(f"class {name} is deprecated, please use {new_cls.__name__} instead.")
        new_cls.__init__(self, *args, **kwargs)

    return type(name, (new_cls,), {"__init__": __init__})


TD0Estimate = _deprecated(TD0Estimator, "TD0Estimate")
TD1Estimate = _deprecated(TD1Estimator, "TD1Estimate")
TDLambdaEstimate = _deprecated(TDLambdaEstimator, "TDLambdaEstimate")
