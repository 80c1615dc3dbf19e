// Host-side fast path of the python entry points (CPython module torchrl_amd/lib/_trl_host.so): the eligibility checks, the
// output allocations and the C-ABI call in one C++ function, so that generalized_advantage_estimate /
// vec_generalized_advantage_estimate / gae_normalized / (vec_)td_lambda_return_estimate / vtrace_advantage_estimate cost
// about the kernel launch itself (drop-in API overhead; measured numbers in DESIGN.md section 5).  Plumbing
// only: no arithmetic on the data happens here, the kernels are reached through the same entry points of
// include/trl_value.h, whose addresses python hands over at init -- the library this module drives is whichever one ctypes
// loaded.
//
// gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated|None, eps|None, algo, time_dim)
// td_lambda(gamma, lmbda, next_state_value, reward, done, terminated|None, algo, time_dim, tensors_ok)
// vtrace(gamma, log_pi, log_mu, state_value, next_state_value, reward, done, terminated|None, rho_thresh, c_thresh, algo, time_dim)
//   -> the outputs (tuple / tensor) the call was made
//   -> None                        not a fast-path call (python's general path takes it and raises the reference's errors)
//   -> int                         non-zero status of the C-ABI call (python raises TrlError with trl_last_error())
// Eligible: gamma / lmbda as python floats, 1-element tensors or tensors of the inputs' shape and dtype (parse_coef), python-float
// thresholds, dense same-shape float32 / float64 tensors of one dtype on the current HIP device, bool masks either dense or per-step masks expanded (stride 0) over the dims after time, no autograd.  Scalar
// coefficients are rounded here as functional._scalar_mask_coef / _round_to do (the reference's `gamma * int32_mask` takes
// the torch default dtype, functional.py:172-174, 1322-1323; torch.full_like(values, gamma) the value dtype, :850-857).
#include <torch/csrc/autograd/python_variable.h>
#include <ATen/ATen.h>
#include <c10/core/GradMode.h>
#include <c10/core/impl/DeviceGuardImplInterface.h>

#include <cstdint>
#include <vector>

#include "../../include/trl_value.h"

namespace {

decltype(&trl_gae_fwd) g_gae_fwd = nullptr;
decltype(&trl_gae_normalize_fwd) g_gae_norm = nullptr;
decltype(&trl_workspace_bytes) g_ws_bytes = nullptr;
decltype(&trl_td_lambda_fwd) g_td_lambda = nullptr;
decltype(&trl_vtrace_fwd) g_vtrace = nullptr;
decltype(&trl_gae_bwd) g_gae_bwd = nullptr;
decltype(&trl_td_lambda_bwd) g_td_lambda_bwd = nullptr;
decltype(&trl_td1_bwd) g_td1_bwd = nullptr;
decltype(&trl_last_kernel) g_last_kernel = nullptr;
const char *g_last_bwd = "";  // kernel family of the last backward call made here (autograd calls from its own thread)
decltype(&trl_standardize) g_standardize = nullptr;
decltype(&trl_standardize_grouped) g_standardize_grouped = nullptr;
decltype(&trl_grouped_workspace_bytes) g_grouped_bytes = nullptr;
PyObject *g_scratch_cb = nullptr;  // (device index, outer, G, inner) -> None; python allocates the grouped scratch and calls set_grouped_scratch
PyObject *g_ws_cb = nullptr;      // (device index, O, T, I) -> None; python allocates / grows the workspace and calls set_workspace
PyObject *g_stream_cb = nullptr;  // torch._C._cuda_getCurrentRawStream

struct Workspace {
    int dev;
    uintptr_t stream, ptr;
    size_t bytes;
};
std::vector<Workspace> g_ws;  // mirror of functional._WORKSPACES, kept current by set_workspace

struct GroupedScratch {  // mirror of _utils._SCRATCH: per (device, stream, G) the per-group moments and the zero-filled workspace
    int dev;
    uintptr_t stream;
    int64_t G;
    uintptr_t stats, ptr;
    size_t bytes;
};
std::vector<GroupedScratch> g_gs;
GroupedScratch *find_gs(int dev, uintptr_t stream, int64_t G) {
    for (auto &w : g_gs)
        if (w.dev == dev && w.stream == stream && w.G == G) return &w;
    return nullptr;
}

Workspace *find_ws(int dev, uintptr_t stream) {
    for (auto &w : g_ws)
        if (w.dev == dev && w.stream == stream) return &w;
    return nullptr;
}

bool dense_after_expand(const at::Tensor &m, c10::IntArrayRef shape, int64_t td) {
    // a per-step mask expanded over the dims after time: its [O, T] bytes are dense
    auto st = m.strides();
    int64_t acc = 1;
    for (int64_t k = (int64_t)shape.size() - 1; k >= 0; --k) {
        if (k > td) {
            if (st[k] != 0 && shape[k] != 1) return false;
        } else {
            if (st[k] != acc && shape[k] != 1) return false;
            acc *= shape[k];
        }
    }
    return true;
}

// What every estimator's fast path checks: `vals` dense, same shape / dtype / device; masks bool, dense or expanded per-step.
struct Call {
    int64_t O, T, I;
    int bcast, vdt;
    int dev;
    uintptr_t stream;
    at::ScalarType dt;
};
enum Status { kOk, kDecline, kError };

Status canonical(const at::Tensor *const *vals, int nvals, const at::Tensor &done, const at::Tensor &term, int64_t td, Call &c) {
    const at::Tensor &ref = *vals[0];
    auto shape = ref.sizes();
    const int64_t nd = (int64_t)shape.size();
    if (nd < 2 || done.sizes() != shape || term.sizes() != shape) return kDecline;
    const auto dt = ref.scalar_type();
    if ((dt != at::kFloat && dt != at::kDouble) || done.scalar_type() != at::kBool || term.scalar_type() != at::kBool) return kDecline;
    const auto dev = ref.device();
    if (!dev.is_cuda() || done.device() != dev || term.device() != dev) return kDecline;
    bool grad = false;
    for (int k = 0; k < nvals; ++k) {
        const at::Tensor &t = *vals[k];
        if (t.sizes() != shape || t.scalar_type() != dt || t.device() != dev || !t.is_contiguous()) return kDecline;
        grad = grad || t.requires_grad();
    }
    if (grad && c10::GradMode::is_enabled()) return kDecline;
    if (c10::impl::getDeviceGuardImpl(c10::DeviceType::CUDA)->getDevice().index() != dev.index()) return kDecline;
    if (td < 0) td += nd;
    if (td < 0 || td >= nd) return kDecline;
    c.bcast = 0;
    if (!done.is_contiguous() || !term.is_contiguous()) {
        if (!dense_after_expand(done, shape, td) || !dense_after_expand(term, shape, td)) return kDecline;
        c.bcast = 1;
    }
    const int64_t n = ref.numel();
    if (n == 0) return kDecline;
    c.T = shape[td];
    c.I = 1;
    for (int64_t k = td + 1; k < nd; ++k) c.I *= shape[k];
    c.O = n / (c.T * c.I);
    c.dt = dt;
    c.vdt = dt == at::kFloat ? TRL_F32 : TRL_F64;
    c.dev = dev.index();
    PyObject *dev_idx = PyLong_FromLong(c.dev);
    PyObject *sres = PyObject_CallFunctionObjArgs(g_stream_cb, dev_idx, nullptr);
    Py_DECREF(dev_idx);
    if (!sres) return kError;
    c.stream = (uintptr_t)PyLong_AsUnsignedLongLong(sres);
    Py_DECREF(sres);
    return PyErr_Occurred() ? kError : kOk;
}

// double -> the value a python float takes as a tensor of `dt` (round to nearest even; beyond FLT_MAX: inf, as torch)
bool round_to(double x, at::ScalarType dt, double &out) {
    if (dt == at::kDouble) { out = x; return true; }
    if (dt == at::kFloat) { out = (double)(float)x; return true; }
    return false;
}

// ---- gamma / lmbda / thresholds as the reference takes them -------------------------------------------------------------
// python float; 1-element tensor (0-dim or [1], host or device: reading a device scalar synchronises, as float(x) in python
// does); tensor of the inputs' shape (per-element coefficients, read in place).  Anything else declines.
enum CoefKind { kPyFloat, kOneElement, kFull };
struct Coef {
    CoefKind kind = kPyFloat;
    double val = 0.0;                     // python float as given / the tensor's one element
    at::ScalarType dt = at::kDouble;      // tensor dtype (kOneElement, kFull)
    const at::Tensor *t = nullptr;
};
bool parse_coef(PyObject *o, c10::IntArrayRef shape, const c10::Device &dev, Coef &c) {
    if (PyFloat_CheckExact(o)) {
        c.kind = kPyFloat;
        c.val = PyFloat_AS_DOUBLE(o);
        return true;
    }
    if (!THPVariable_Check(o)) return false;
    const at::Tensor &t = THPVariable_Unpack(o);
    const auto dt = t.scalar_type();
    if (dt != at::kFloat && dt != at::kDouble) return false;
    if (t.requires_grad() && c10::GradMode::is_enabled()) return false;
    c.t = &t;
    c.dt = dt;
    if (t.numel() == 1) {
        c.kind = kOneElement;
        c.val = t.item<double>();
        return true;
    }
    if (t.sizes() != shape || t.device() != dev || !t.is_contiguous()) return false;
    c.kind = kFull;
    return true;
}
int code_of(at::ScalarType dt) { return dt == at::kFloat ? TRL_F32 : TRL_F64; }
at::ScalarType promote2(at::ScalarType a, at::ScalarType b) { return (a == at::kDouble || b == at::kDouble) ? at::kDouble : at::kFloat; }

// the (device, stream) workspace when the call needs one: `always` (normalising call), else only for the look-back scan of few,
// very long rows (functional._lookback_workspace)
Status workspace_for(const Call &c, bool always, uintptr_t &ptr, size_t &bytes) {
    ptr = 0;
    bytes = 0;
    if (!always && !(c.I == 1 && c.T >= 2048 && c.O < 4096)) return kOk;
    const size_t need = g_ws_bytes(c.O, c.T, c.I);
    Workspace *w = find_ws(c.dev, c.stream);
    if (!w || w->bytes < need) {
        PyObject *res = PyObject_CallFunction(g_ws_cb, "iLLL", c.dev, (long long)c.O, (long long)c.T, (long long)c.I);
        if (!res) return kError;
        Py_DECREF(res);
        w = find_ws(c.dev, c.stream);
        if (!w || w->bytes < need) {
            PyErr_SetString(PyExc_RuntimeError, "_trl_host: workspace callback did not register a large enough workspace");
            return kError;
        }
    }
    ptr = w->ptr;
    bytes = w->bytes;
    return kOk;
}

bool empty_like(const at::Tensor &ref, at::ScalarType dt, at::Tensor &out) {
    try {
        out = dt == ref.scalar_type() ? at::empty_like(ref) : at::empty_like(ref, ref.options().dtype(dt));
        return true;
    } catch (const std::exception &e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return false;
    }
}

// status of a C-ABI call: 0 = go on; the degrade warning = emit a RuntimeWarning and go on (the outputs are valid); anything
// else is handed to python as an int (python raises TrlError with trl_last_error()).  Returns false when the caller must
// return `*ret` (an int object, or nullptr with a python error set because warnings are errors).
bool call_ok(int rc, PyObject **ret) {
    if (rc == 0) return true;
    if (rc == TRL_WARN_FUSED_DEGRADED) {
        if (PyErr_WarnEx(PyExc_RuntimeWarning,
                         "torchrl_amd: an earlier single-launch call (GAE + normalise, or standardize) on this GPU gave up waiting "
                         "for its other blocks and left its output UN-NORMALISED; those steps now take the multi-launch routes on "
                         "this GPU (functional.fused_reset() re-arms the single launch)", 1) < 0) {
            *ret = nullptr;
            return false;
        }
        return true;
    }
    *ret = PyLong_FromLong(rc);
    return false;
}

PyObject *pair(at::Tensor &&a, at::Tensor &&b) {
    PyObject *out = PyTuple_New(2);
    if (!out) return nullptr;
    PyTuple_SET_ITEM(out, 0, THPVariable_Wrap(std::move(a)));
    PyTuple_SET_ITEM(out, 1, THPVariable_Wrap(std::move(b)));
    return out;
}

bool ready(Py_ssize_t nargs, Py_ssize_t want, const char *name) {
    if (nargs != want) {
        PyErr_Format(PyExc_TypeError, "_trl_host.%s takes %zd positional arguments", name, want);
        return false;
    }
    if (!g_gae_fwd) {
        PyErr_SetString(PyExc_RuntimeError, "_trl_host called before init");
        return false;
    }
    return true;
}

PyObject *py_init(PyObject *, PyObject *args) {
    unsigned long long a, b, c, d, e, f, g, h, i, j, k, l;
    PyObject *ws_cb, *stream_cb, *scratch_cb;
    if (!PyArg_ParseTuple(args, "KKKKKKKKKKKKOOO", &a, &b, &c, &d, &e, &f, &g, &h, &i, &j, &k, &l, &ws_cb, &stream_cb, &scratch_cb))
        return nullptr;
    g_gae_bwd = (decltype(g_gae_bwd))(uintptr_t)i;
    g_td_lambda_bwd = (decltype(g_td_lambda_bwd))(uintptr_t)j;
    g_td1_bwd = (decltype(g_td1_bwd))(uintptr_t)k;
    g_last_kernel = (decltype(g_last_kernel))(uintptr_t)l;
    g_gae_fwd = (decltype(g_gae_fwd))(uintptr_t)a;
    g_gae_norm = (decltype(g_gae_norm))(uintptr_t)b;
    g_ws_bytes = (decltype(g_ws_bytes))(uintptr_t)c;
    g_td_lambda = (decltype(g_td_lambda))(uintptr_t)d;
    g_vtrace = (decltype(g_vtrace))(uintptr_t)e;
    g_standardize = (decltype(g_standardize))(uintptr_t)f;
    g_standardize_grouped = (decltype(g_standardize_grouped))(uintptr_t)g;
    g_grouped_bytes = (decltype(g_grouped_bytes))(uintptr_t)h;
    Py_XDECREF(g_ws_cb);
    Py_XDECREF(g_stream_cb);
    Py_XDECREF(g_scratch_cb);
    Py_INCREF(ws_cb);
    Py_INCREF(stream_cb);
    Py_INCREF(scratch_cb);
    g_ws_cb = ws_cb;
    g_stream_cb = stream_cb;
    g_scratch_cb = scratch_cb;
    g_ws.clear();
    g_gs.clear();
    Py_RETURN_NONE;
}

PyObject *py_set_grouped_scratch(PyObject *, PyObject *args) {
    int dev;
    unsigned long long stream, stats, ptr, bytes;
    long long G;
    if (!PyArg_ParseTuple(args, "iKLKKK", &dev, &stream, &G, &stats, &ptr, &bytes)) return nullptr;
    GroupedScratch *w = find_gs(dev, (uintptr_t)stream, G);
    if (w) {
        w->stats = (uintptr_t)stats;
        w->ptr = (uintptr_t)ptr;
        w->bytes = (size_t)bytes;
    } else {
        g_gs.push_back({dev, (uintptr_t)stream, (int64_t)G, (uintptr_t)stats, (uintptr_t)ptr, (size_t)bytes});
    }
    Py_RETURN_NONE;
}

PyObject *py_set_workspace(PyObject *, PyObject *args) {
    int dev;
    unsigned long long stream, ptr, bytes;
    if (!PyArg_ParseTuple(args, "iKKK", &dev, &stream, &ptr, &bytes)) return nullptr;
    Workspace *w = find_ws(dev, (uintptr_t)stream);
    if (w) {
        w->ptr = (uintptr_t)ptr;
        w->bytes = (size_t)bytes;
    } else {
        g_ws.push_back({dev, (uintptr_t)stream, (uintptr_t)ptr, (size_t)bytes});
    }
    Py_RETURN_NONE;
}

#define DECLINE_UNLESS_TENSOR(obj) \
    if (!THPVariable_Check(obj)) Py_RETURN_NONE

PyObject *py_gae(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (!ready(nargs, 10, "gae")) return nullptr;
    // coefficients: python floats / 1-element tensors / tensors of the inputs' shape (parse_coef), or (gamma, lmbda*gamma, coef
    // dtype code) already rounded by python (functional.scalar_gae_coef: what the modules pass for their 0-dim buffers) with
    // None in lmbda's place
    const bool pre = PyTuple_CheckExact(a[0]) && a[1] == Py_None;
    for (int k = 2; k <= 5; ++k) DECLINE_UNLESS_TENSOR(a[k]);
    const bool has_term = a[6] != Py_None;
    if (has_term) DECLINE_UNLESS_TENSOR(a[6]);
    const at::Tensor &sv = THPVariable_Unpack(a[2]);
    const at::Tensor &nsv = THPVariable_Unpack(a[3]);
    const at::Tensor &r = THPVariable_Unpack(a[4]);
    const at::Tensor &done = THPVariable_Unpack(a[5]);
    const at::Tensor &term = has_term ? THPVariable_Unpack(a[6]) : done;
    const bool normalise = a[7] != Py_None;
    const double eps = normalise ? PyFloat_AsDouble(a[7]) : 0.0;
    const long algo = PyLong_AsLong(a[8]);
    const int64_t td = PyLong_AsLongLong(a[9]);
    if (PyErr_Occurred()) return nullptr;

    // gamma * int32_mask and (lmbda * gamma) * int32_mask: python floats take the default dtype, floating tensors keep theirs
    // (functional.py:172-174); the products are formed by ATen itself so that their rounding is torch's by construction
    double g_val = 0.0, lg_val = 0.0, l_val = 0.0;
    int cdt;
    const void *gamma_t = nullptr, *lg_t = nullptr;
    at::Tensor keep_g, keep_lg;  // per-element coefficient tensors formed here
    if (pre) {
        if (!PyArg_ParseTuple(a[0], "ddi", &g_val, &lg_val, &cdt)) return nullptr;
        if (cdt != TRL_F32 && cdt != TRL_F64) Py_RETURN_NONE;
    } else {
        Coef g, l;
        if (!parse_coef(a[0], nsv.sizes(), nsv.device(), g) || !parse_coef(a[1], nsv.sizes(), nsv.device(), l)) {
            if (PyErr_Occurred()) return nullptr;
            Py_RETURN_NONE;
        }
        const auto ddt = c10::get_default_dtype_as_scalartype();
        if (ddt != at::kFloat && ddt != at::kDouble) Py_RETURN_NONE;
        const auto vdt = nsv.scalar_type();
        try {
            if (g.kind == kPyFloat && l.kind == kPyFloat) {
                round_to(g.val, ddt, g_val);
                round_to(l.val * g.val, ddt, lg_val);
                cdt = code_of(ddt);
            } else if (g.kind != kFull && l.kind != kFull) {  // scalar-like, at least one a tensor: `lmbda * gamma` as torch forms it
                at::ScalarType g_dt = ddt;
                if (g.kind == kPyFloat) round_to(g.val, ddt, g_val);
                else { g_val = g.val; g_dt = g.dt; }
                const at::Tensor prod = l.kind == kPyFloat ? at::mul(*g.t, at::Scalar(l.val))
                                                           : (g.kind == kPyFloat ? at::mul(*l.t, at::Scalar(g.val)) : at::mul(*l.t, *g.t));
                lg_val = prod.item<double>();
                cdt = code_of(promote2(g_dt, prod.scalar_type()));
            } else if (g.kind == kFull && l.kind != kFull) {  // lg = (coef)lmbda * gamma_t is formed in the kernel
                if (g.dt != vdt || (l.kind == kOneElement && l.dt != vdt)) Py_RETURN_NONE;
                gamma_t = g.t->data_ptr();
                l_val = l.val;
                cdt = code_of(g.dt);
            } else {  // per-element lmbda: lmbda * gamma as a tensor, gamma per element too
                keep_lg = g.kind == kPyFloat ? at::mul(*l.t, at::Scalar(g.val)) : at::mul(*l.t, *g.t);
                const at::ScalarType g_dt = g.kind == kPyFloat ? ddt : g.dt;
                if (keep_lg.scalar_type() != vdt || g_dt != vdt || keep_lg.sizes() != nsv.sizes()) Py_RETURN_NONE;
                if (g.kind == kFull) gamma_t = g.t->data_ptr();
                else {
                    keep_g = at::full(nsv.sizes(), at::Scalar(g.val), nsv.options().dtype(g_dt));
                    gamma_t = keep_g.data_ptr();
                }
                keep_lg = keep_lg.contiguous();
                lg_t = keep_lg.data_ptr();
                cdt = code_of(vdt);
            }
        } catch (const std::exception &) {  // mixed devices, ...: python's general path raises the reference's error
            Py_RETURN_NONE;
        }
    }

    const at::Tensor *vals[3] = {&nsv, &sv, &r};
    Call c;
    switch (canonical(vals, 3, done, term, td, c)) {
        case kDecline: Py_RETURN_NONE;
        case kError: return nullptr;
        default: break;
    }
    uintptr_t ws_ptr;
    size_t ws_bytes;
    if (workspace_for(c, normalise, ws_ptr, ws_bytes) != kOk) return nullptr;
    at::Tensor adv, vt;
    if (!empty_like(nsv, c.dt, adv) || !empty_like(nsv, c.dt, vt)) return nullptr;
    int rc;
    if (normalise)
        rc = g_gae_norm((void *)c.stream, c.vdt, cdt, r.data_ptr(), sv.data_ptr(), nsv.data_ptr(), (const uint8_t *)done.data_ptr(),
                        (const uint8_t *)term.data_ptr(), c.bcast, gamma_t, lg_t, g_val, l_val, lg_val, c.O, c.T, c.I, adv.data_ptr(),
                        vt.data_ptr(), eps, nullptr, (void *)ws_ptr, ws_bytes, (int)algo);
    else
        rc = g_gae_fwd((void *)c.stream, c.vdt, cdt, r.data_ptr(), sv.data_ptr(), nsv.data_ptr(), (const uint8_t *)done.data_ptr(),
                       (const uint8_t *)term.data_ptr(), c.bcast, gamma_t, lg_t, g_val, l_val, lg_val, c.O, c.T, c.I, adv.data_ptr(),
                       vt.data_ptr(), nullptr, (void *)ws_ptr, ws_bytes, (int)algo);
    PyObject *ret = nullptr;
    if (!call_ok(rc, &ret)) return ret;
    return pair(std::move(adv), std::move(vt));
}

PyObject *py_td_lambda(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (!ready(nargs, 9, "td_lambda")) return nullptr;
    for (int k = 2; k <= 4; ++k) DECLINE_UNLESS_TENSOR(a[k]);
    const bool has_term = a[5] != Py_None;
    if (has_term) DECLINE_UNLESS_TENSOR(a[5]);
    const at::Tensor &nsv = THPVariable_Unpack(a[2]);
    const at::Tensor &r = THPVariable_Unpack(a[3]);
    const at::Tensor &done = THPVariable_Unpack(a[4]);
    const at::Tensor &term = has_term ? THPVariable_Unpack(a[5]) : done;
    const long algo = PyLong_AsLong(a[6]);
    const int64_t td = PyLong_AsLongLong(a[7]);
    const bool tensors_ok = PyObject_IsTrue(a[8]) == 1;  // per-element coefficients: rolling gamma only (the caller knows)
    if (PyErr_Occurred()) return nullptr;
    Coef g, l;
    if (!parse_coef(a[0], nsv.sizes(), nsv.device(), g) || !parse_coef(a[1], nsv.sizes(), nsv.device(), l)) {
        if (PyErr_Occurred()) return nullptr;
        Py_RETURN_NONE;
    }
    if ((g.kind == kFull || l.kind == kFull) && !tensors_ok) Py_RETURN_NONE;
    const at::Tensor *vals[2] = {&nsv, &r};
    Call c;
    switch (canonical(vals, 2, done, term, td, c)) {
        case kDecline: Py_RETURN_NONE;
        case kError: return nullptr;
        default: break;
    }
    // torch.full_like(next_state_value, gamma): scalar-like coefficients take the value dtype (functional.py:850-857); a tensor
    // of the inputs' shape is read per element (here: when it has the value dtype)
    double g_val = 0.0, l_val = 0.0;
    const void *gamma_t = nullptr, *lmbda_t = nullptr;
    // (a coefficient tensor of another dtype -- 0-dim ones included -- changes the vec variant's output dtype: python's path)
    if ((g.kind != kPyFloat && g.dt != c.dt) || (l.kind != kPyFloat && l.dt != c.dt)) Py_RETURN_NONE;
    if (g.kind == kFull) gamma_t = g.t->data_ptr(); else round_to(g.val, c.dt, g_val);
    if (l.kind == kFull) lmbda_t = l.t->data_ptr(); else round_to(l.val, c.dt, l_val);
    uintptr_t ws_ptr;
    size_t ws_bytes;
    if (workspace_for(c, false, ws_ptr, ws_bytes) != kOk) return nullptr;
    at::Tensor out;
    if (!empty_like(nsv, c.dt, out)) return nullptr;
    const int rc = g_td_lambda((void *)c.stream, c.vdt, c.vdt, r.data_ptr(), nsv.data_ptr(), (const uint8_t *)done.data_ptr(),
                               (const uint8_t *)term.data_ptr(), c.bcast, gamma_t, lmbda_t, g_val, l_val, 1, c.O, c.T, c.I, out.data_ptr(),
                               (void *)ws_ptr, ws_bytes, (int)algo);
    PyObject *ret = nullptr;
    if (!call_ok(rc, &ret)) return ret;
    return THPVariable_Wrap(std::move(out));
}

PyObject *py_vtrace(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (!ready(nargs, 12, "vtrace")) return nullptr;
    if (!PyFloat_CheckExact(a[8]) || !PyFloat_CheckExact(a[9])) Py_RETURN_NONE;
    for (int k = 1; k <= 6; ++k) DECLINE_UNLESS_TENSOR(a[k]);
    const bool has_term = a[7] != Py_None;
    if (has_term) DECLINE_UNLESS_TENSOR(a[7]);
    const at::Tensor &lp = THPVariable_Unpack(a[1]);
    const at::Tensor &lmu = THPVariable_Unpack(a[2]);
    const at::Tensor &sv = THPVariable_Unpack(a[3]);
    const at::Tensor &nsv = THPVariable_Unpack(a[4]);
    const at::Tensor &r = THPVariable_Unpack(a[5]);
    const at::Tensor &done = THPVariable_Unpack(a[6]);
    const at::Tensor &term = has_term ? THPVariable_Unpack(a[7]) : done;
    const long algo = PyLong_AsLong(a[10]);
    const int64_t td = PyLong_AsLongLong(a[11]);
    if (PyErr_Occurred()) return nullptr;
    // gamma * int32_mask takes the default dtype; python thresholds become 0-dim default-dtype tensors (functional.py:1311-1323)
    // a gamma tensor keeps its dtype (1-element: the scalar; inputs' shape: per element, read in place)
    const auto ddt = c10::get_default_dtype_as_scalartype();
    Coef g;
    if (!parse_coef(a[0], nsv.sizes(), nsv.device(), g)) {
        if (PyErr_Occurred()) return nullptr;
        Py_RETURN_NONE;
    }
    double g_val = 0.0, rho_hi, c_hi;
    if (!round_to(PyFloat_AS_DOUBLE(a[8]), ddt, rho_hi) || !round_to(PyFloat_AS_DOUBLE(a[9]), ddt, c_hi)) Py_RETURN_NONE;
    at::ScalarType coef_dt = ddt;
    const void *gamma_t = nullptr;
    if (g.kind == kPyFloat) round_to(g.val, ddt, g_val);
    else if (g.kind == kOneElement) { g_val = g.val; coef_dt = g.dt; }
    else { gamma_t = g.t->data_ptr(); coef_dt = g.dt; }
    const int cdt = code_of(coef_dt);
    const at::Tensor *vals[5] = {&nsv, &sv, &r, &lp, &lmu};
    Call c;
    switch (canonical(vals, 5, done, term, td, c)) {
        case kDecline: Py_RETURN_NONE;
        case kError: return nullptr;
        default: break;
    }
    uintptr_t ws_ptr;
    size_t ws_bytes;
    if (workspace_for(c, false, ws_ptr, ws_bytes) != kOk) return nullptr;
    const auto odt = promote2(c.dt, coef_dt);  // promote(value, coefficient dtype)
    at::Tensor adv, vs;
    if (!empty_like(nsv, odt, adv) || !empty_like(nsv, odt, vs)) return nullptr;
    const int rc = g_vtrace((void *)c.stream, c.vdt, cdt, lp.data_ptr(), lmu.data_ptr(), sv.data_ptr(), nsv.data_ptr(), r.data_ptr(),
                            (const uint8_t *)done.data_ptr(), (const uint8_t *)term.data_ptr(), c.bcast, gamma_t, g_val, rho_hi, c_hi,
                            nullptr, nullptr, c.O, c.T, c.I, adv.data_ptr(), vs.data_ptr(), (void *)ws_ptr, ws_bytes, (int)algo);
    PyObject *ret = nullptr;
    if (!call_ok(rc, &ret)) return ret;
    return pair(std::move(adv), std::move(vs));
}

// ---- backward launchers (functional._gae_bwd / _td_bwd in one hop) -------------------------------------------------------
// incoming gradient as a dense tensor of dtype dt (None: the kernels read zeros)
bool grad_in(PyObject *o, at::ScalarType dt, at::Tensor &keep, const void *&ptr) {
    ptr = nullptr;
    if (o == Py_None) return true;
    if (!THPVariable_Check(o)) return false;
    const at::Tensor &g = THPVariable_Unpack(o);
    if (g.scalar_type() == dt && g.is_contiguous()) { ptr = g.data_ptr(); return true; }
    keep = g.detach().to(dt).contiguous();
    ptr = keep.data_ptr();
    return true;
}
const void *ptr_or_null(PyObject *o) { return o == Py_None ? nullptr : THPVariable_Unpack(o).data_ptr(); }
bool call_of(const at::Tensor &like, int64_t O, int64_t T, int64_t I, Call &c) {
    const auto dev = like.device();
    if (!dev.is_cuda() || c10::impl::getDeviceGuardImpl(c10::DeviceType::CUDA)->getDevice().index() != dev.index()) return false;
    c.O = O; c.T = T; c.I = I;
    c.dev = dev.index();
    PyObject *dev_idx = PyLong_FromLong(c.dev);
    PyObject *sres = PyObject_CallFunctionObjArgs(g_stream_cb, dev_idx, nullptr);
    Py_DECREF(dev_idx);
    if (!sres) return false;
    c.stream = (uintptr_t)PyLong_AsUnsignedLongLong(sres);
    Py_DECREF(sres);
    return !PyErr_Occurred();
}
PyObject *wrap_or_none(at::Tensor &t) {
    if (!t.defined()) Py_RETURN_NONE;
    return THPVariable_Wrap(std::move(t));
}

// gae_bwd(g_adv|None, g_vt|None, like, done_u8, terminated_u8, mask_bcast, gamma_t|None, lg_t|None, gamma, lmbda, lg, O, T, I,
//         coef dtype code, need_v, need_nv, need_r, algo) -> (d_state_value, d_next_state_value, d_reward) | None | int
// `like`: a dense value tensor of the call (shape / dtype / device of the outputs); masks and coefficient tensors as the
// prepared call holds them (functional._GaeCall)
PyObject *py_gae_bwd(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (!ready(nargs, 19, "gae_bwd")) return nullptr;
    for (int k : {2, 3, 4}) DECLINE_UNLESS_TENSOR(a[k]);
    const at::Tensor &like = THPVariable_Unpack(a[2]);
    const auto dt = like.scalar_type();
    if (dt != at::kFloat && dt != at::kDouble) Py_RETURN_NONE;
    const int64_t O = PyLong_AsLongLong(a[11]), T = PyLong_AsLongLong(a[12]), I = PyLong_AsLongLong(a[13]);
    const int bcast = (int)PyLong_AsLong(a[5]), cdt = (int)PyLong_AsLong(a[14]), algo = (int)PyLong_AsLong(a[18]);
    const double gamma = PyFloat_AsDouble(a[8]), lmbda = PyFloat_AsDouble(a[9]), lg = PyFloat_AsDouble(a[10]);
    const bool need_v = PyObject_IsTrue(a[15]) == 1, need_nv = PyObject_IsTrue(a[16]) == 1, need_r = PyObject_IsTrue(a[17]) == 1;
    if (PyErr_Occurred()) return nullptr;
    if (!need_v && !need_nv && !need_r) Py_RETURN_NONE;
    Call c;
    if (!call_of(like, O, T, I, c)) { if (PyErr_Occurred()) return nullptr; Py_RETURN_NONE; }
    at::Tensor keep_a, keep_v, d_v, d_nv, d_r;
    const void *ga, *gv;
    if (!grad_in(a[0], dt, keep_a, ga) || !grad_in(a[1], dt, keep_v, gv)) Py_RETURN_NONE;
    if ((need_v && !empty_like(like, dt, d_v)) || (need_nv && !empty_like(like, dt, d_nv)) || (need_r && !empty_like(like, dt, d_r))) return nullptr;
    uintptr_t ws_ptr;
    size_t ws_bytes;
    if (workspace_for(c, false, ws_ptr, ws_bytes) != kOk) return nullptr;
    const int rc = g_gae_bwd((void *)c.stream, code_of(dt), cdt, ga, gv, (const uint8_t *)THPVariable_Unpack(a[3]).data_ptr(),
                             (const uint8_t *)THPVariable_Unpack(a[4]).data_ptr(), bcast, ptr_or_null(a[6]), ptr_or_null(a[7]), gamma, lmbda, lg,
                             O, T, I, need_r ? d_r.data_ptr() : nullptr, need_v ? d_v.data_ptr() : nullptr,
                             need_nv ? d_nv.data_ptr() : nullptr, (void *)ws_ptr, ws_bytes, algo);
    PyObject *ret = nullptr;
    if (!call_ok(rc, &ret)) return ret;
    g_last_bwd = g_last_kernel();
    PyObject *out = PyTuple_New(3);
    if (!out) return nullptr;
    PyTuple_SET_ITEM(out, 0, wrap_or_none(d_v));
    PyTuple_SET_ITEM(out, 1, wrap_or_none(d_nv));
    PyTuple_SET_ITEM(out, 2, wrap_or_none(d_r));
    return out;
}

// td_bwd(g_ret, like, done_u8, terminated_u8, mask_bcast, gamma_t|None, lmbda_t|None, gamma, lmbda, O, T, I, coef dtype code,
//        td1, need_nv, need_r, algo) -> (d_next_state_value, d_reward) | None | int
PyObject *py_td_bwd(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (!ready(nargs, 17, "td_bwd")) return nullptr;
    for (int k : {0, 1, 2, 3}) DECLINE_UNLESS_TENSOR(a[k]);
    const at::Tensor &like = THPVariable_Unpack(a[1]);
    const auto dt = like.scalar_type();
    if (dt != at::kFloat && dt != at::kDouble) Py_RETURN_NONE;
    const int64_t O = PyLong_AsLongLong(a[9]), T = PyLong_AsLongLong(a[10]), I = PyLong_AsLongLong(a[11]);
    const int bcast = (int)PyLong_AsLong(a[4]), cdt = (int)PyLong_AsLong(a[12]), algo = (int)PyLong_AsLong(a[16]);
    const double gamma = PyFloat_AsDouble(a[7]), lmbda = PyFloat_AsDouble(a[8]);
    const bool td1 = PyObject_IsTrue(a[13]) == 1, need_nv = PyObject_IsTrue(a[14]) == 1, need_r = PyObject_IsTrue(a[15]) == 1;
    if (PyErr_Occurred()) return nullptr;
    if (!need_nv && !need_r) Py_RETURN_NONE;
    Call c;
    if (!call_of(like, O, T, I, c)) { if (PyErr_Occurred()) return nullptr; Py_RETURN_NONE; }
    at::Tensor keep, d_nv, d_r;
    const void *gr;
    if (!grad_in(a[0], dt, keep, gr)) Py_RETURN_NONE;
    if ((need_nv && !empty_like(like, dt, d_nv)) || (need_r && !empty_like(like, dt, d_r))) return nullptr;
    uintptr_t ws_ptr;
    size_t ws_bytes;
    if (workspace_for(c, false, ws_ptr, ws_bytes) != kOk) return nullptr;
    const uint8_t *done = (const uint8_t *)THPVariable_Unpack(a[2]).data_ptr(), *term = (const uint8_t *)THPVariable_Unpack(a[3]).data_ptr();
    const int rc = td1 ? g_td1_bwd((void *)c.stream, code_of(dt), cdt, gr, done, term, bcast, ptr_or_null(a[5]), gamma, O, T, I,
                                   need_r ? d_r.data_ptr() : nullptr, need_nv ? d_nv.data_ptr() : nullptr, (void *)ws_ptr, ws_bytes, algo)
                       : g_td_lambda_bwd((void *)c.stream, code_of(dt), cdt, gr, done, term, bcast, ptr_or_null(a[5]), ptr_or_null(a[6]),
                                         gamma, lmbda, O, T, I, need_r ? d_r.data_ptr() : nullptr, need_nv ? d_nv.data_ptr() : nullptr,
                                         (void *)ws_ptr, ws_bytes, algo);
    PyObject *ret = nullptr;
    if (!call_ok(rc, &ret)) return ret;
    g_last_bwd = g_last_kernel();
    PyObject *out = PyTuple_New(2);
    if (!out) return nullptr;
    PyTuple_SET_ITEM(out, 0, wrap_or_none(d_nv));
    PyTuple_SET_ITEM(out, 1, wrap_or_none(d_r));
    return out;
}

PyObject *py_last_bwd_kernel(PyObject *, PyObject *) { return PyUnicode_FromString(g_last_bwd); }

// standardize(x, exclude_dims tuple, eps) -> tensor | None | int: `_standardize(x, exclude_dims)` without caller statistics
// (reference: torchrl/_utils.py:875-936) for a dense float32 / float64 tensor on the current HIP device, no autograd; excluded
// dims must be valid, distinct and ADJACENT (they merge into one group axis); everything else -- validation errors, the
// all-dims-excluded warning, permuted group dims -- is python's.
PyObject *py_standardize(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (!ready(nargs, 3, "standardize")) return nullptr;
    DECLINE_UNLESS_TENSOR(a[0]);
    if (!PyTuple_CheckExact(a[1]) || !PyFloat_CheckExact(a[2])) Py_RETURN_NONE;
    const at::Tensor &x = THPVariable_Unpack(a[0]);
    const double eps = PyFloat_AS_DOUBLE(a[2]);
    const auto dt = x.scalar_type();
    const auto dev = x.device();
    if ((dt != at::kFloat && dt != at::kDouble) || !dev.is_cuda() || !x.is_contiguous()) Py_RETURN_NONE;
    if (x.requires_grad() && c10::GradMode::is_enabled()) Py_RETURN_NONE;
    if (c10::impl::getDeviceGuardImpl(c10::DeviceType::CUDA)->getDevice().index() != dev.index()) Py_RETURN_NONE;
    const int64_t n = x.numel(), nd = x.dim();
    if (n == 0) Py_RETURN_NONE;
    const Py_ssize_t n_ex = PyTuple_GET_SIZE(a[1]);
    int64_t lo = nd, hi = -1;
    uint64_t seen = 0;
    if (n_ex >= nd || nd > 60) Py_RETURN_NONE;
    for (Py_ssize_t k = 0; k < n_ex; ++k) {
        PyObject *o = PyTuple_GET_ITEM(a[1], k);
        if (!PyLong_CheckExact(o)) Py_RETURN_NONE;
        int64_t d = PyLong_AsLongLong(o);
        if (d < 0) d += nd;
        if (d < 0 || d >= nd || (seen >> d) & 1) Py_RETURN_NONE;
        seen |= 1ull << d;
        lo = d < lo ? d : lo;
        hi = d > hi ? d : hi;
    }
    if (n_ex > 0 && hi - lo + 1 != (int64_t)n_ex) Py_RETURN_NONE;  // not adjacent
    PyObject *dev_idx = PyLong_FromLong(dev.index());
    PyObject *sres = PyObject_CallFunctionObjArgs(g_stream_cb, dev_idx, nullptr);
    Py_DECREF(dev_idx);
    if (!sres) return nullptr;
    const uintptr_t stream = (uintptr_t)PyLong_AsUnsignedLongLong(sres);
    Py_DECREF(sres);
    if (PyErr_Occurred()) return nullptr;
    const int vdt = dt == at::kFloat ? TRL_F32 : TRL_F64;
    at::Tensor out;
    if (!empty_like(x, dt, out)) return nullptr;
    int rc;
    if (n_ex == 0) {
        Call c;
        c.O = n; c.T = 1; c.I = 1; c.dev = dev.index(); c.stream = stream;
        uintptr_t ws_ptr = 0;
        size_t ws_bytes = 0;
        if (n > 65536) {  // up to a minibatch: one single-block launch, no workspace (trl_value.h)
            // workspace_for's look-back rule does not apply here: ask for the workspace outright
            if (workspace_for(c, true, ws_ptr, ws_bytes) != kOk) return nullptr;
        }
        rc = g_standardize((void *)stream, vdt, x.data_ptr(), n, eps, out.data_ptr(), nullptr, (void *)ws_ptr, ws_bytes);
    } else {
        auto shape = x.sizes();
        int64_t outer = 1, G = 1, inner = 1;
        for (int64_t k = 0; k < lo; ++k) outer *= shape[k];
        for (int64_t k = lo; k <= hi; ++k) G *= shape[k];
        for (int64_t k = hi + 1; k < nd; ++k) inner *= shape[k];
        const size_t need = g_grouped_bytes(outer, G, inner);
        GroupedScratch *w = find_gs(dev.index(), stream, G);
        if (!w || w->bytes < need) {
            PyObject *res = PyObject_CallFunction(g_scratch_cb, "iLLL", (int)dev.index(), (long long)outer, (long long)G, (long long)inner);
            if (!res) return nullptr;
            Py_DECREF(res);
            w = find_gs(dev.index(), stream, G);
            if (!w || w->bytes < need) {
                PyErr_SetString(PyExc_RuntimeError, "_trl_host: scratch callback did not register a large enough grouped scratch");
                return nullptr;
            }
        }
        rc = g_standardize_grouped((void *)stream, vdt, x.data_ptr(), outer, G, inner, eps, out.data_ptr(), (double *)w->stats,
                                   (void *)w->ptr, w->bytes);
    }
    PyObject *ret = nullptr;
    if (!call_ok(rc, &ret)) return ret;
    return THPVariable_Wrap(std::move(out));
}

PyMethodDef methods[] = {
    {"init", py_init, METH_VARARGS,
     "init(trl_gae_fwd, trl_gae_normalize_fwd, trl_workspace_bytes, trl_td_lambda_fwd, trl_vtrace_fwd, trl_standardize, "
     "trl_standardize_grouped, trl_grouped_workspace_bytes, trl_gae_bwd, trl_td_lambda_bwd, trl_td1_bwd, trl_last_kernel, ws_cb, "
     "stream_cb, scratch_cb)"},
    {"set_workspace", py_set_workspace, METH_VARARGS, "set_workspace(device index, stream, pointer, bytes)"},
    {"set_grouped_scratch", py_set_grouped_scratch, METH_VARARGS,
     "set_grouped_scratch(device index, stream, G, moments pointer, workspace pointer, workspace bytes)"},
    {"standardize", (PyCFunction)(void (*)(void))py_standardize, METH_FASTCALL, "standardize(x, exclude_dims, eps)"},
    {"gae_bwd", (PyCFunction)(void (*)(void))py_gae_bwd, METH_FASTCALL, "gae_bwd(...): see host_fast.cpp"},
    {"td_bwd", (PyCFunction)(void (*)(void))py_td_bwd, METH_FASTCALL, "td_bwd(...): see host_fast.cpp"},
    {"last_bwd_kernel", py_last_bwd_kernel, METH_NOARGS, "kernel family of the last backward call made through this module"},
    {"gae", (PyCFunction)(void (*)(void))py_gae, METH_FASTCALL,
     "gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, eps, algo, time_dim)"},
    {"td_lambda", (PyCFunction)(void (*)(void))py_td_lambda, METH_FASTCALL,
     "td_lambda(gamma, lmbda, next_state_value, reward, done, terminated, algo, time_dim, tensors_ok)"},
    {"vtrace", (PyCFunction)(void (*)(void))py_vtrace, METH_FASTCALL,
     "vtrace(gamma, log_pi, log_mu, state_value, next_state_value, reward, done, terminated, rho_thresh, c_thresh, algo, time_dim)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_trl_host", "host-side fast path of the torchrl_amd value-estimator entry points", -1,
                      methods};

}  // namespace

PyMODINIT_FUNC PyInit__trl_host(void) { return PyModule_Create(&module); }
