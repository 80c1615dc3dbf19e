"""Estimator modules (GAE / TDLambdaEstimator / VTrace / TD0 / TD1), modelled on the reference's
TestAdv (test/test_cost.py:15443-15929).  Host-side behaviour runs on CPU; everything that computes
runs on the GPU (-m gpu) and is checked against the CPU oracle."""
import pytest
import torch
from torch import nn

from torchrl_amd.objectives.value import (GAE, KeyedModule, RolloutData, TD0Estimator, TD1Estimator, TDLambdaEstimate,
                                          TDLambdaEstimator, VTrace)

DEV = "cuda:0"
ALL = [GAE, TDLambdaEstimator, TD1Estimator, TD0Estimator]


def _kw(cls):
    kw = dict(gamma=0.98, value_network=None)
    if cls in (GAE, TDLambdaEstimator):
        kw["lmbda"] = 0.94
    return kw


# ------------------------------------------------------------------------------------------- CPU
def _net_data(seed=0):
    g = torch.Generator().manual_seed(seed)
    obs, nobs = torch.randn(3, 6, 3, generator=g), torch.randn(3, 6, 3, generator=g)
    term = torch.rand(3, 6, 1, generator=g) < 0.1
    return RolloutData({"obs": obs, "next": {"obs": nobs, "reward": torch.randn(3, 6, 1, generator=g), "done": term.clone(),
                                             "terminated": term}}, batch_size=[3, 6])


@pytest.mark.parametrize("cls", [GAE, TDLambdaEstimator, TD0Estimator])
def test_functional_params_as_a_name_to_tensor_mapping(cls):
    """advantages.py:425-516 with ``params`` / ``target_params``: without the tensordict package a mapping in
    ``named_parameters()`` naming is applied with ``torch.func.functional_call``; the network's own parameters stay untouched and
    receive no graph; ``params`` alone also serves the t+1 call; the reference's two ValueErrors."""
    net = KeyedModule(nn.Linear(3, 1), ["obs"], ["state_value"])
    m = cls(**_kw(cls), value_network=net) if "value_network" not in _kw(cls) else cls(**{**_kw(cls), "value_network": net})
    p = {"module.weight": torch.full((1, 3), 0.5), "module.bias": torch.tensor([0.25], requires_grad=True)}
    tp = {"module.weight": torch.full((1, 3), -1.0), "module.bias": torch.tensor([2.0])}
    own = [q.detach().clone() for q in net.parameters()]
    d = _net_data()
    m(d, params=p)
    lin = lambda x, w: x @ w["module.weight"].T + w["module.bias"]
    torch.testing.assert_close(d.get("state_value"), lin(d.get("obs"), p).detach())
    torch.testing.assert_close(d.get(("next", "state_value")), lin(d.get(("next", "obs")), p).detach())
    assert not d.get("state_value").requires_grad and not d.get("advantage").requires_grad
    d2 = _net_data()
    m(d2, params=p, target_params=tp)
    torch.testing.assert_close(d2.get("state_value"), lin(d2.get("obs"), p).detach())
    torch.testing.assert_close(d2.get(("next", "state_value")), lin(d2.get(("next", "obs")), tp))
    assert not torch.equal(d2.get("advantage"), d.get("advantage"))
    d3 = _net_data()
    m(d3)                                                   # the network's own parameters: untouched by the calls above
    assert all(torch.equal(a, b) for a, b in zip(own, net.parameters()))
    assert not torch.equal(d3.get("state_value"), d.get("state_value"))
    with pytest.raises(ValueError, match="either both provided or not"):
        m(_net_data(), target_params=tp)
    sh = cls(**{**_kw(cls), "value_network": net, "shifted": True})
    with pytest.raises(ValueError, match="single call"):
        sh(_net_data(), params=p, target_params=tp)
    d4 = _net_data()
    sh(d4, params=p)                                        # one call on T + 1 steps with the same mapping
    torch.testing.assert_close(d4.get("state_value"), lin(d4.get("obs"), p).detach())
    with pytest.raises(NotImplementedError, match="functional parameters"):
        m(_net_data(), params=[1, 2])


@pytest.mark.parametrize("cls", ALL)
def test_set_keys_and_deprecated_ctor_keys(cls):
    m = cls(**_kw(cls))
    assert m.out_keys == ["advantage", "value_target"]
    m.set_keys(advantage="adv", value_target=("a", "vt"), value="sv")
    assert m.tensor_keys.advantage == "adv" and m.tensor_keys.value_target == ("a", "vt") and m.value_key == "sv"
    with pytest.raises(KeyError, match="not an accepted tensordict key"):
        m.set_keys(bogus="x")
    with pytest.raises(ValueError, match="cannot be None"):
        m.set_keys(advantage=None)
    for k in ("advantage_key", "value_target_key", "value_key"):
        with pytest.raises(RuntimeError, match="via constructor is deprecated"):
            cls(**_kw(cls), **{k: "x"})


def test_value_key_must_be_a_network_out_key():
    net = KeyedModule(nn.Linear(3, 1), ["obs"], ["state_value"])
    m = GAE(gamma=0.9, lmbda=0.9, value_network=net)
    assert ("next", "reward") in m.in_keys and "obs" in m.in_keys and ("next", "obs") in m.in_keys
    with pytest.raises(KeyError, match="not found in value network out_keys"):
        m.set_keys(value="other")


def test_time_dim_resolution_order():
    m = GAE(gamma=0.9, lmbda=0.9, value_network=None)
    data = RolloutData({"x": torch.zeros(2, 3, 4, 1)}, batch_size=[2, 3, 4])
    assert m._get_time_dim(None, data) == 2            # last batch dim
    assert m._get_time_dim(-3, data) == 0 and m._get_time_dim(1, data) == 1
    data.names = [None, "time", None]
    assert m._get_time_dim(None, data) == 1            # dim named "time"
    m.time_dim = 0
    assert m._get_time_dim(None, data) == 0            # attribute beats names
    assert m._get_time_dim(2, data) == 2               # explicit beats everything


def test_rollout_data_nested_keys_and_batch_size():
    d = RolloutData({"obs": torch.zeros(4, 10, 3), "next": {"reward": torch.zeros(4, 10, 1)}})
    assert d.batch_size == torch.Size([4, 10]) and d.batch_dims == 2
    d.set(("next", "done"), torch.zeros(4, 10, 1, dtype=torch.bool))
    assert d.get(("next", "done")).dtype == torch.bool and d.get("missing", None) is None
    assert d.get("next").get("reward").shape == (4, 10, 1)
    with pytest.raises(KeyError):
        d.get(("next", "nope"))


def test_state_dict_roundtrip_of_buffers():
    m = GAE(gamma=0.97, lmbda=0.93, value_network=None)
    sd = m.state_dict()
    assert set(sd) == {"gamma", "lmbda"}
    m2 = GAE(gamma=0.5, lmbda=0.5, value_network=None)
    m2.load_state_dict(sd)
    assert float(m2.gamma) == float(torch.tensor(0.97)) and float(m2.lmbda) == float(torch.tensor(0.93))
    v = VTrace(gamma=0.9, actor_network=lambda d: None, value_network=None, rho_thresh=0.7)
    assert set(v.state_dict()) == {"gamma", "rho_thresh", "c_thresh"}
    with pytest.raises(NotImplementedError, match="Per-value gamma"):
        VTrace(gamma=torch.full((3,), 0.9), actor_network=None, value_network=None)


def test_batch_dims_check():
    m = GAE(gamma=0.9, lmbda=0.9, value_network=None)
    with pytest.raises(RuntimeError, match="at least one dimension"):
        m(RolloutData({"state_value": torch.zeros(())}, batch_size=[]))


def test_deprecated_alias_warns():
    with pytest.warns(UserWarning, match="deprecated"):
        TDLambdaEstimate(gamma=0.9, lmbda=0.9, value_network=None)


# ------------------------------------------------------------------------------------------- GPU
def _data(shape=(4, 10), feat=(1,), seed=0, device=DEV):
    g = torch.Generator().manual_seed(seed)
    full = tuple(shape) + tuple(feat)
    mk = lambda: torch.randn(full, generator=g).to(device)
    term = (torch.rand(full, generator=g) < 0.1).to(device)
    done = term | (torch.rand(full, generator=g) < 0.1).to(device)
    return RolloutData({"state_value": mk(), "sample_log_prob": torch.rand(full, generator=g).log().to(device),
                        "next": {"state_value": mk(), "reward": mk(), "done": done, "terminated": term}}, batch_size=shape)


@pytest.mark.gpu
@pytest.mark.parametrize("cls", ALL)
def test_forward_writes_keys_and_matches_oracle(cls):
    from oracle import value_oracle as O

    d = _data()
    m = cls(**_kw(cls))
    out = m(d)
    assert out is d
    adv, vt = d.get("advantage"), d.get("value_target")
    assert adv.shape == (4, 10, 1) and vt.shape == (4, 10, 1) and adv.is_leaf and not adv.requires_grad
    v, nv, r = (d.get(k).cpu() for k in ("state_value", ("next", "state_value"), ("next", "reward")))
    dn, tm = d.get(("next", "done")).cpu(), d.get(("next", "terminated")).cpu()
    g32 = torch.tensor(0.98)
    if cls is GAE:
        eadv, evt = O.gae(g32, torch.tensor(0.94), v, nv, r, dn, tm)
    elif cls is TD0Estimator:
        evt = r + 0.98 * (~tm).int() * nv
        eadv = evt - v
    else:
        evt = O.td_lambda(g32, torch.tensor(0.94) if cls is TDLambdaEstimator else 1.0, nv, r, dn, tm)
        eadv = evt - v
    torch.testing.assert_close(adv.cpu(), eadv, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(vt.cpu(), evt, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_dispatch_kwargs_api():
    """test_cost.py:15501-15508: unpacked inputs, `next_` prefix, returns (advantage, value_target)."""
    net = KeyedModule(nn.Linear(3, 1).to(DEV), ["obs"], ["state_value"])
    m = GAE(gamma=0.98, lmbda=0.94, value_network=net, differentiable=False)
    obs, next_obs = torch.randn(2, 1, 10, 3, device=DEV)
    reward = torch.randn(1, 10, 1, device=DEV)
    done = torch.zeros(1, 10, 1, dtype=torch.bool, device=DEV)
    adv, vt = m(obs=obs, next_reward=reward, next_done=done, next_obs=next_obs, next_terminated=done)
    assert adv.shape == (1, 10, 1) and vt.shape == (1, 10, 1)
    # shifted=True (single call over T+1 steps) agrees when next_obs is the shifted obs
    seq = torch.randn(1, 11, 3, device=DEV)
    a1, _ = GAE(gamma=0.98, lmbda=0.94, value_network=net)(obs=seq[:, :-1], next_obs=seq[:, 1:], next_reward=reward, next_done=done)
    a2, _ = GAE(gamma=0.98, lmbda=0.94, value_network=net, shifted=True)(obs=seq[:, :-1], next_obs=seq[:, 1:], next_reward=reward, next_done=done)
    torch.testing.assert_close(a1, a2, rtol=1e-5, atol=1e-5)
    for p in net.parameters():
        assert p.grad is None and p.requires_grad


@pytest.mark.gpu
def test_time_dim_variants_agree():
    """test_cost.py test_time_dim: time_dim None / attribute / explicit on transposed data agree."""
    d = _data(shape=(4, 10))
    ref = GAE(gamma=0.98, lmbda=0.94, value_network=None)(d).get("advantage")
    tr = lambda t: t.transpose(0, 1).contiguous()
    d2 = RolloutData({"state_value": tr(d.get("state_value")),
                      "next": {k: tr(d.get(("next", k))) for k in ("state_value", "reward", "done", "terminated")}}, batch_size=(10, 4))
    a_explicit = GAE(gamma=0.98, lmbda=0.94, value_network=None)(d2, time_dim=0).get("advantage")
    a_attr = GAE(gamma=0.98, lmbda=0.94, value_network=None, time_dim=-2)(d2).get("advantage")
    d2.names = ["time", None]
    a_named = GAE(gamma=0.98, lmbda=0.94, value_network=None)(d2).get("advantage")
    for a in (a_explicit, a_attr, a_named):
        torch.testing.assert_close(tr(a), ref, rtol=1e-6, atol=1e-6)
    vt = GAE(gamma=0.98, lmbda=0.94, value_network=None).value_estimate(d2, time_dim=0)
    torch.testing.assert_close(tr(vt), d.get("value_target"), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_average_gae_is_fused_and_matches_reference_formula():
    from torchrl_amd.objectives.value import functional as F

    d = _data(shape=(64, 32), seed=3)
    plain = GAE(gamma=0.99, lmbda=0.95, value_network=None)(d).get("advantage").clone()
    normed = GAE(gamma=0.99, lmbda=0.95, value_network=None, average_gae=True)(d).get("advantage")
    assert F.last_kernel() == "gae_fused"
    exp = (plain - plain.mean()) / plain.std().clamp_min(1e-4)
    torch.testing.assert_close(normed, exp, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_skip_existing():
    d = _data()
    m = GAE(gamma=0.98, lmbda=0.94, value_network=None, skip_existing=True)
    d.set("advantage", torch.zeros(4, 10, 1, device=DEV))
    d.set("value_target", torch.zeros(4, 10, 1, device=DEV))
    m(d)
    assert float(d.get("advantage").abs().sum()) == 0.0
    m.skip_existing = False
    m(d)
    assert float(d.get("advantage").abs().sum()) > 0.0


@pytest.mark.gpu
def test_vtrace_module():
    from oracle import value_oracle as O

    d = _data(seed=5)
    g = torch.Generator().manual_seed(9)
    log_pi = torch.rand(4, 10, 1, generator=g).log().to(DEV)
    m = VTrace(gamma=0.98, actor_network=lambda data: log_pi, value_network=None, rho_thresh=0.9, c_thresh=0.8, average_adv=True)
    m(d)
    c = lambda k: d.get(k).cpu()
    eadv, evs = O.vtrace(torch.tensor(0.98), log_pi.cpu(), c("sample_log_prob"), c("state_value"), c(("next", "state_value")),
                         c(("next", "reward")), c(("next", "done")), c(("next", "terminated")), torch.tensor(0.9), torch.tensor(0.8))
    torch.testing.assert_close(d.get("value_target").cpu(), evs, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d.get("advantage").cpu(), O.normalize(eadv, 1e-5), rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_steps_to_next_obs_scales_gamma():
    from oracle import value_oracle as O

    d = _data(seed=6)
    steps = torch.randint(1, 4, (4, 10, 1)).to(DEV)
    d.set("steps_to_next_obs", steps)
    GAE(gamma=0.98, lmbda=0.94, value_network=None)(d)
    c = lambda k: d.get(k).cpu()
    gam = torch.tensor(0.98) ** steps.cpu()
    eadv, _ = O.gae(gam, torch.tensor(0.94), c("state_value"), c(("next", "state_value")), c(("next", "reward")),
                    c(("next", "done")), c(("next", "terminated")))
    torch.testing.assert_close(d.get("advantage").cpu(), eadv, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_differentiable_normalisation_keeps_the_graph():
    """VTrace(average_adv=True, differentiable=True) and TDLambdaEstimator(average_rewards=True, differentiable=True):
    the normalisation is the reference's differentiable torch ops (advantages.py:1828-1832, 1160-1165), so the gradient
    reaches the reward; with differentiable=False the outputs are leaves."""
    for diff in (True, False):
        d = _data(seed=7)
        reward = d.get(("next", "reward")).clone().requires_grad_(True)
        d.set(("next", "reward"), reward)
        g = torch.Generator().manual_seed(9)
        log_pi = torch.rand(4, 10, 1, generator=g).log().to(DEV)
        VTrace(gamma=0.98, actor_network=lambda data: log_pi, value_network=None, average_adv=True, differentiable=diff)(d)
        adv = d.get("advantage")
        assert adv.requires_grad == diff
        if diff:
            adv.pow(2).sum().backward()
            assert reward.grad is not None and reward.grad.norm() > 0
        else:
            assert adv.is_leaf


@pytest.mark.gpu
def test_host_mirror_keeps_one_entry_per_buffer():
    """looking gamma then lmbda up must not evict each other (a device buffer would be synchronised on
    every forward): the second look-up of a buffer returns the SAME host copy."""
    m = GAE(gamma=0.98, lmbda=0.94, value_network=None).to(DEV)
    g1, l1 = m._host(m.gamma), m._host(m.lmbda)
    assert g1.device.type == "cpu" and l1.device.type == "cpu"
    assert m._host(m.gamma) is g1 and m._host(m.lmbda) is l1 and m._host(m.gamma) is g1
    m.gamma.fill_(0.5)  # in-place update: the mirror is refreshed
    assert float(m._host(m.gamma)) == 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("default_dtype", [torch.float32, torch.float64], ids=["def32", "def64"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_module_one_hop_route_equals_general_route(dtype, default_dtype, monkeypatch):
    """The modules hand their scalar buffers to the one-hop host path as pre-rounded coefficients (GAE) or as python floats
    where those are equivalent (TD(lambda), V-trace).  Bit-equal to the general path -- which is what ran before -- for value
    dtypes and default dtypes that do and do not match the buffers', and after a buffer is modified in place."""
    from torchrl_amd.objectives.value import functional as F

    old = torch.get_default_dtype()
    torch.set_default_dtype(default_dtype)
    try:
        log_pi = torch.rand(4, 10, 1, generator=torch.Generator().manual_seed(9)).log().to(dtype).to(DEV)
        mods = {
            "gae": GAE(gamma=0.98, lmbda=0.94, value_network=None, device=DEV),
            "gae_avg": GAE(gamma=0.98, lmbda=0.94, value_network=None, average_gae=True, device=DEV),
            "gae_f32buf": GAE(gamma=torch.tensor(0.98, dtype=torch.float32), lmbda=torch.tensor(0.94, dtype=torch.float32),
                              value_network=None),
            "tdl": TDLambdaEstimator(gamma=0.98, lmbda=0.94, value_network=None, device=DEV),
            "tdl_loop": TDLambdaEstimator(gamma=0.98, lmbda=0.94, value_network=None, vectorized=False),
            "td1": TD1Estimator(gamma=0.98, value_network=None),
            "vtrace": VTrace(gamma=0.98, actor_network=lambda data: log_pi, value_network=None, rho_thresh=0.9, c_thresh=0.8),
        }

        def run(m):
            d = _data(seed=3)
            for k in ("state_value", "sample_log_prob", ("next", "state_value"), ("next", "reward")):
                d.set(k, d.get(k).to(dtype))
            m(d)
            return d.get("advantage"), d.get("value_target")

        fast = {k: run(m) for k, m in mods.items()}
        mods["gae"].gamma.fill_(0.5)       # in place: the cached coefficients must go
        fast["gae_changed"] = run(mods["gae"])
        assert not torch.equal(fast["gae"][0], fast["gae_changed"][0])
        assert F.host_fast_path()
        with monkeypatch.context() as mp:
            mp.setattr(F, "_HOST", None)
            mp.setattr(F, "_gae_fast", lambda *a, **k: None)
            mods["gae"].gamma.fill_(0.98)
            slow = {k: run(m) for k, m in mods.items()}
            mods["gae"].gamma.fill_(0.5)
            slow["gae_changed"] = run(mods["gae"])
        for k in fast:
            for a, b in zip(fast[k], slow[k]):
                assert a.dtype == b.dtype, k
                assert torch.equal(a, b), f"{k}: {(a - b).abs().max().item()}"
    finally:
        torch.set_default_dtype(old)


@pytest.mark.gpu
def test_gae_short_forward_covers_only_the_plain_case(monkeypatch):
    """``GAE.forward`` resolves the plain case (RolloutData, string keys, scalar buffers, nothing to differentiate) with a
    handful of dict look-ups and one hop to the C ABI; everything else goes the long way and gives the same numbers."""
    m = GAE(gamma=0.98, lmbda=0.94, value_network=None, average_gae=True, device=DEV)
    took = []
    orig = GAE._short_forward

    def spy(self, data):
        out = orig(self, data)
        took.append(out is not None)
        return out

    monkeypatch.setattr(GAE, "_short_forward", spy)
    d = _data(seed=5)
    assert m(d) is d and took == [True]
    ref = (d.get("advantage").clone(), d.get("value_target").clone())

    def same(data, adv_key="advantage"):
        assert torch.equal(data.get(adv_key), ref[0]) and torch.equal(data.get("value_target"), ref[1])

    d2 = _data(seed=5)
    m(d2, time_dim=-1)                       # explicit time_dim: not even tried
    assert took == [True]
    same(d2)
    d3 = _data(seed=5)
    d3.names = [None, "time"]                # named dims
    m(d3)
    assert took[-1] is False
    same(d3)
    m.set_keys(advantage=("out", "adv"))     # nested key
    d4 = _data(seed=5)
    m(d4)
    assert took[-1] is False
    same(d4, ("out", "adv"))
    m.set_keys(advantage="adv2")             # renamed string key: still the short way
    d4b = _data(seed=5)
    m(d4b)
    assert took[-1] is True
    same(d4b, "adv2")
    m.set_keys(advantage="advantage")
    d5 = _data(seed=5)
    d5.set("steps_to_next_obs", torch.ones(4, 10, 1, dtype=torch.long, device=DEV))
    m(d5)
    assert took[-1] is False
    torch.testing.assert_close(d5.get("advantage"), ref[0], rtol=1e-5, atol=1e-5)   # gamma ** 1 as a tensor: the per-element path
    d6 = _data(seed=5)
    d6.get("state_value").requires_grad_()
    m(d6)                                    # grad mode on, input requires grad: the long way (computed under no_grad)
    assert took[-1] is False and not d6.get("advantage").requires_grad
    same(d6)
    with torch.no_grad():
        d6b = _data(seed=5)
        d6b.get("state_value").requires_grad_()
        m(d6b)
        assert took[-1] is True
        same(d6b)
    d7 = _data(seed=5)
    del d7.to_dict()["next"]["terminated"]   # terminated defaults to done on both ways
    m(d7)
    assert took[-1] is True
    d7b = _data(seed=5)
    del d7b.to_dict()["next"]["terminated"]
    m(d7b, time_dim=-1)
    assert torch.equal(d7.get("advantage"), d7b.get("advantage"))
    with pytest.raises(KeyError):            # a missing entry raises what the long way raises
        m(RolloutData({"state_value": torch.zeros(4, 10, 1, device=DEV)}, batch_size=[4, 10]))
    m.gamma = torch.tensor(0.5, device=DEV)  # replaced buffer: the cached coefficients go
    d8 = _data(seed=5)
    m(d8)
    assert took[-1] is True and not torch.equal(d8.get("advantage"), ref[0])
    d8b = _data(seed=5)
    m(d8b, time_dim=-1)
    assert torch.equal(d8.get("advantage"), d8b.get("advantage"))
    m.gamma.fill_(0.98)                      # modified in place
    d9 = _data(seed=5)
    m(d9)
    same(d9)
    cpu = GAE(gamma=0.98, lmbda=0.94, value_network=None, average_gae=True)
    dc = _data(seed=5, device="cpu")
    cpu(dc)                                  # host tensors: declined, the CPU path answers
    assert took[-1] is False
    torch.testing.assert_close(dc.get("advantage"), ref[0].cpu(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cls", [TDLambdaEstimator, TD1Estimator])
@pytest.mark.parametrize("vectorized", [True, False])
def test_td_short_forward_equals_the_long_way(cls, vectorized, monkeypatch):
    if cls is TD1Estimator and not vectorized:
        pytest.skip("TD1Estimator has no loop variant")
    kw = dict(_kw(cls))
    if cls is TDLambdaEstimator:
        kw["vectorized"] = vectorized
    took = []
    orig = TDLambdaEstimator._short_forward

    def spy(self, data):
        out = orig(self, data)
        took.append(out is not None)
        return out

    monkeypatch.setattr(TDLambdaEstimator, "_short_forward", spy)
    m = cls(**kw, device=DEV)
    d = _data(seed=8)
    assert m(d) is d and took == [True]
    m.time_dim = -1                      # the attribute sends forward the long way
    d2 = _data(seed=8)
    m(d2)
    assert took == [True]
    for k in ("advantage", "value_target"):
        assert torch.equal(d.get(k), d2.get(k)), k
    m.time_dim = None
    d3 = _data(seed=8)
    d3.set(("next", "state_value"), d3.get(("next", "state_value")).double())   # fp64 values, fp32 buffers: still floats
    d3.set("state_value", d3.get("state_value").double())
    d3.set(("next", "reward"), d3.get(("next", "reward")).double())
    m(d3)
    assert took[-1] is True and d3.get("value_target").dtype == torch.float64
    m64 = cls(**{**kw, "gamma": torch.tensor(0.98, dtype=torch.float64)}, device=DEV)
    d4, d5 = _data(seed=8), _data(seed=8)
    m64(d4)                              # a wider buffer widens the vec variant's result: the long way decides
    assert took[-1] is (not vectorized)
    m64.time_dim = -1
    m64(d5)
    assert d4.get("value_target").dtype == d5.get("value_target").dtype and torch.equal(d4.get("value_target"), d5.get("value_target"))
    m.gamma.fill_(0.5)
    d6, d7 = _data(seed=8), _data(seed=8)
    m(d6)
    m.time_dim = -1
    m(d7)
    assert torch.equal(d6.get("advantage"), d7.get("advantage")) and not torch.equal(d6.get("advantage"), d.get("advantage"))
