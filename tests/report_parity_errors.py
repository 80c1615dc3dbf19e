"""Measured parity errors of the HIP path against the CPU oracle, per BASELINE config and estimator (GPU; test infrastructure:
it lives under tests/ because it calls the oracle -- not collected by pytest, run it by name on the GPU box):

    python tests/report_parity_errors.py > gpurun_out/parity_errors.txt

For every case it prints the largest absolute error and the largest MIXED error |e| / (1 + |ref|) -- the quantity a
`assert_close(rtol = atol = tol)` bounds -- of each output, for the kernel family AUTO picks and for SEQ where that differs.
The committed copy (profiles/r03_parity_errors.txt) is what the tolerances in tests/test_gpu_parity.py and
tests/test_gpu_fuzz.py are derived from: north_star's bar is 1e-5 (fp32) / 1e-12 (fp64) against the sequential CPU loop.
The normalised advantage is (A - mean) / std: an error e in A arrives as e / std, so its bar scales with 1 / std when
std < 1 (printed beside the case)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import value_oracle as O  # noqa: E402
from torchrl_amd.objectives.value import functional as F  # noqa: E402

DEV = "cuda:0"


def err(got, ref):
    g, r = got.detach().double().cpu().flatten(), ref.detach().double().cpu().flatten()
    fin = torch.isfinite(r)
    e = (g[fin] - r[fin]).abs()
    if e.numel() == 0:
        return 0.0, 0.0
    return float(e.max()), float((e / (1 + r[fin].abs())).max())


def line(cfg, what, kernel, pairs, extra=""):
    cells = "  ".join(f"{n}: abs {a:.2e} mixed {m:.2e}" for n, (a, m) in pairs)
    print(f"{cfg:34s} {what:30s} {kernel:26s} {cells} {extra}", flush=True)


def cu(t):
    return t.to(DEV)


def gae_case(cfg, shape, dtype, time_dim=-2, gamma=0.99, lmbda=0.95, p_term=0.01, p_trunc=0.01, expanded=False, gamma_tensor=False,
             normalise=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)
    r, v, nv = mk(), mk(), mk()
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g, dtype=torch.float32)).to(dtype) if gamma_tensor else gamma
    if expanded:   # per-env masks expanded over the agent dim (DoneTransform)
        mshape = shape[:-2] + (1, 1)
        term = (torch.rand(mshape, generator=g) < p_term).expand(shape)
        done = (term | (torch.rand(mshape, generator=g) < p_trunc).expand(shape))
    else:
        term = torch.rand(shape, generator=g) < p_term
        done = term | (torch.rand(shape, generator=g) < p_trunc)
    kw = {} if time_dim == -2 else {"time_dim": time_dim}
    oadv, ovt = O.gae(gam, lmbda, v, nv, r, done.contiguous(), term.contiguous(), **kw)
    dgam = cu(gam) if gamma_tensor else gam
    for algo, name in ((F.ALGO_AUTO, "AUTO"), (F.ALGO_SEQ, "SEQ")):
        call = F._prepare_gae(dgam, lmbda, cu(v), cu(nv), cu(r), cu(done), cu(term), time_dim)
        adv, vt, _ = F._launch_gae(call, algo=algo)
        k = F.last_kernel()
        dmask = done.contiguous()
        exact = bool(torch.equal(adv.cpu()[dmask], oadv[dmask]))
        line(cfg, f"GAE {'f32' if dtype == torch.float32 else 'f64'} {name}", k, [("adv", err(adv, oadv)), ("vt", err(vt, ovt))],
             f"done positions bit-exact: {exact}")
    if normalise:
        std = float(oadv.double().std())
        onorm = O.normalize(oadv, 1e-4)
        for algo, name in ((F.ALGO_FUSED, "FUSED"), (F.ALGO_SCAN, "two launches (SCAN)"), (F.ALGO_SEQ, "two launches (SEQ)")):
            call = F._prepare_gae(dgam, lmbda, cu(v), cu(nv), cu(r), cu(done), cu(term), time_dim)
            try:
                adv, vt, _ = F._launch_gae_normalize(call, 1e-4, algo=algo)
            except RuntimeError as e:
                line(cfg, f"GAE+normalise {name}", "-", [], f"not eligible: {str(e)[:60]}")
                continue
            line(cfg, f"GAE+normalise {name}", F.last_kernel(), [("adv_norm", err(adv, onorm)), ("vt", err(vt, ovt))], f"std(A) = {std:.3f}")


def c5_case(dtype):
    tag = "f32" if dtype == torch.float32 else "f64"
    cfg = f"C5 [1024x10000] {tag} gamma tensor"
    g = torch.Generator().manual_seed(0)
    shape = (1024, 10000, 1)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)
    r, v, nv = mk(), mk(), mk()
    gamma = (0.9 + 0.1 * torch.rand(shape, generator=g, dtype=torch.float32)).to(dtype)
    term = torch.rand(shape, generator=g) < 0.001
    done = term | (torch.rand(shape, generator=g) < 0.001)
    log_pi = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(dtype)
    log_mu = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(dtype)
    oret = O.td_lambda(gamma, 0.95, nv, r, done, term)
    oa, ovs = O.vtrace(gamma, log_pi, log_mu, v, nv, r, done, term)
    oadv, ovt = O.gae(gamma, 0.95, v, nv, r, done, term)
    for algo, name in ((F.ALGO_AUTO, "AUTO"), (F.ALGO_SEQ, "SEQ")):
        c = F._prepare_td_lambda(cu(gamma), 0.95, cu(nv), cu(r), cu(done), cu(term), None, -2)
        ret = F._launch_td_lambda(c, dtype, algo=algo)
        line(cfg, f"TD(lambda) {name}", F.last_kernel(), [("returns", err(ret, oret))])
        c = F._prepare_vtrace(cu(gamma), cu(log_pi), cu(log_mu), cu(v), cu(nv), cu(r), cu(done), cu(term), 1.0, 1.0, -2)
        a, vs = F._launch_vtrace(c, algo=algo)
        line(cfg, f"V-trace {name}", F.last_kernel(), [("adv", err(a, oa)), ("vs", err(vs, ovs))])
        c = F._prepare_gae(cu(gamma), 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term), -2)
        adv, vt, _ = F._launch_gae(c, algo=algo)
        line(cfg, f"GAE {name}", F.last_kernel(), [("adv", err(adv, oadv)), ("vt", err(vt, ovt))])


def c4_other(dtype):
    tag = "f32" if dtype == torch.float32 else "f64"
    cfg = f"C4 [65536x32] {tag}"
    g = torch.Generator().manual_seed(3)
    shape = (65536, 32, 1)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)
    r, v, nv = mk(), mk(), mk()
    term = torch.rand(shape, generator=g) < 0.01
    done = term | (torch.rand(shape, generator=g) < 0.01)
    log_pi = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(dtype)
    log_mu = torch.rand(shape, generator=g).clamp_min(1e-6).double().log().to(dtype)
    oret = O.td_lambda(0.99, 0.95, nv, r, done, term)
    oa, ovs = O.vtrace(0.99, log_pi, log_mu, v, nv, r, done, term)
    for algo, name in ((F.ALGO_AUTO, "AUTO"), (F.ALGO_SEQ, "SEQ")):
        c = F._prepare_td_lambda(0.99, 0.95, cu(nv), cu(r), cu(done), cu(term), None, -2)
        ret = F._launch_td_lambda(c, dtype, algo=algo)
        line(cfg, f"TD(lambda) {name}", F.last_kernel(), [("returns", err(ret, oret))])
        c = F._prepare_vtrace(0.99, cu(log_pi), cu(log_mu), cu(v), cu(nv), cu(r), cu(done), cu(term), 1.0, 1.0, -2)
        a, vs = F._launch_vtrace(c, algo=algo)
        line(cfg, f"V-trace {name}", F.last_kernel(), [("adv", err(a, oa)), ("vs", err(vs, ovs))])


def main():
    print(f"# HIP path vs CPU oracle; device {torch.cuda.get_device_name(0)}; torch {torch.__version__}")
    print("# mixed = max |e| / (1 + |ref|); the bar: 1e-5 (fp32), 1e-12 (fp64)")
    for dt in (torch.float32, torch.float64):
        tag = "f32" if dt == torch.float32 else "f64"
        gae_case(f"C1 [1x2048] {tag}", (1, 2048, 1), dt, p_term=1 / 200, p_trunc=0.0, normalise=True)
        gae_case(f"C2 [8x128] {tag}", (8, 128, 1), dt, p_term=0.005, p_trunc=0.005, normalise=True)
        gae_case(f"C3 [4096x100x5] {tag} expanded masks", (4096, 100, 5, 1), dt, time_dim=-3, gamma=0.9, lmbda=0.9, expanded=True, normalise=True)
        gae_case(f"C3 [4096x100x5] {tag} per-agent masks", (4096, 100, 5, 1), dt, time_dim=-3, gamma=0.9, lmbda=0.9)
        gae_case(f"C4 [65536x32] {tag}", (65536, 32, 1), dt, normalise=True)
        gae_case(f"C4 shard [8192x32] {tag}", (8192, 32, 1), dt, normalise=True)
        c4_other(dt)
        c5_case(dt)


if __name__ == "__main__":
    main()
