// Host-side fast path of the GAE entry points (CPython module torchrl_amd.lib._trl_host): the eligibility checks, the two
// output allocations and the C-ABI call of functional._gae_fast in one C++ function, so that the public
// generalized_advantage_estimate / vec_generalized_advantage_estimate / gae_normalized cost about the kernel launch itself
// (drop-in API overhead; measured numbers in DESIGN.md section 5).  Plumbing only: no arithmetic on the data
// happens here, the kernels are reached through the same trl_gae_fwd / trl_gae_normalize_fwd entry points (include/trl_value.h)
// whose addresses python hands over at init -- the library this module drives is whichever one ctypes loaded.
//
// gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated|None, eps|None, algo, time_dim)
//   -> (advantage, value_target)   the call was made
//   -> None                        not a fast-path call (python's general path takes it and raises the reference's errors)
//   -> int                         non-zero status of the C-ABI call (python raises TrlError with trl_last_error())
// Eligible: python-float gamma / lmbda, dense same-shape float32 / float64 tensors on the current HIP device, bool masks either
// dense or per-step masks expanded (stride 0) over the dims after time, no autograd.
#include <torch/csrc/autograd/python_variable.h>
#include <ATen/ATen.h>
#include <c10/core/GradMode.h>
#include <c10/core/impl/DeviceGuardImplInterface.h>

#include <cstdint>
#include <vector>

#include "../../include/trl_value.h"

namespace {

typedef decltype(&trl_gae_fwd) gae_fwd_fn;
typedef decltype(&trl_gae_normalize_fwd) gae_norm_fn;
typedef decltype(&trl_workspace_bytes) ws_bytes_fn;

gae_fwd_fn g_gae_fwd = nullptr;
gae_norm_fn g_gae_norm = nullptr;
ws_bytes_fn g_ws_bytes = nullptr;
PyObject *g_coef_cb = nullptr;    // (gamma, lmbda) -> (gamma as rounded by the reference's mask product, lmbda*gamma likewise, coef dtype code) | None
PyObject *g_ws_cb = nullptr;      // (device index, O, T, I) -> None; python allocates / grows the workspace and calls set_workspace
PyObject *g_stream_cb = nullptr;  // torch._C._cuda_getCurrentRawStream

struct Coef {
    double gamma, lmbda, g_val, lg_val;
    int default_dtype, cdt;
};
Coef g_coef = {0, 0, 0, 0, -1, 0};

struct Workspace {
    int dev;
    uintptr_t stream, ptr;
    size_t bytes;
};
std::vector<Workspace> g_ws;  // mirror of functional._WORKSPACES, kept current by set_workspace

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

PyObject *py_init(PyObject *, PyObject *args) {
    unsigned long long a, b, c;
    PyObject *coef_cb, *ws_cb, *stream_cb;
    if (!PyArg_ParseTuple(args, "KKKOOO", &a, &b, &c, &coef_cb, &ws_cb, &stream_cb)) return nullptr;
    g_gae_fwd = (gae_fwd_fn)(uintptr_t)a;
    g_gae_norm = (gae_norm_fn)(uintptr_t)b;
    g_ws_bytes = (ws_bytes_fn)(uintptr_t)c;
    Py_XDECREF(g_coef_cb); Py_XDECREF(g_ws_cb); Py_XDECREF(g_stream_cb);
    Py_INCREF(coef_cb); Py_INCREF(ws_cb); Py_INCREF(stream_cb);
    g_coef_cb = coef_cb; g_ws_cb = ws_cb; g_stream_cb = stream_cb;
    g_ws.clear();
    g_coef.default_dtype = -1;
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

PyObject *py_gae(PyObject *, PyObject *const *a, Py_ssize_t nargs) {
    if (nargs != 10) {
        PyErr_SetString(PyExc_TypeError, "_trl_host.gae takes 10 positional arguments");
        return nullptr;
    }
    if (!g_gae_fwd) {
        PyErr_SetString(PyExc_RuntimeError, "_trl_host.gae called before init");
        return nullptr;
    }
    if (!PyFloat_CheckExact(a[0]) || !PyFloat_CheckExact(a[1])) Py_RETURN_NONE;
    for (int k = 2; k <= 5; ++k)
        if (!THPVariable_Check(a[k])) Py_RETURN_NONE;
    const bool has_term = a[6] != Py_None;
    if (has_term && !THPVariable_Check(a[6])) Py_RETURN_NONE;
    const at::Tensor &sv = THPVariable_Unpack(a[2]);
    const at::Tensor &nsv = THPVariable_Unpack(a[3]);
    const at::Tensor &r = THPVariable_Unpack(a[4]);
    const at::Tensor &done = THPVariable_Unpack(a[5]);
    const at::Tensor &term = has_term ? THPVariable_Unpack(a[6]) : done;
    const bool normalise = a[7] != Py_None;
    const double eps = normalise ? PyFloat_AsDouble(a[7]) : 0.0;
    const long algo = PyLong_AsLong(a[8]);
    int64_t td = PyLong_AsLongLong(a[9]);
    if (PyErr_Occurred()) return nullptr;

    auto shape = nsv.sizes();
    const int64_t nd = (int64_t)shape.size();
    if (nd < 2 || sv.sizes() != shape || r.sizes() != shape || done.sizes() != shape || term.sizes() != shape) Py_RETURN_NONE;
    const auto dt = nsv.scalar_type();
    if ((dt != at::kFloat && dt != at::kDouble) || sv.scalar_type() != dt || r.scalar_type() != dt || done.scalar_type() != at::kBool ||
        term.scalar_type() != at::kBool)
        Py_RETURN_NONE;
    const auto dev = nsv.device();
    if (!dev.is_cuda() || sv.device() != dev || r.device() != dev || done.device() != dev || term.device() != dev) Py_RETURN_NONE;
    if (c10::impl::getDeviceGuardImpl(c10::DeviceType::CUDA)->getDevice().index() != dev.index()) Py_RETURN_NONE;
    if (!nsv.is_contiguous() || !sv.is_contiguous() || !r.is_contiguous()) Py_RETURN_NONE;
    if (td < 0) td += nd;
    if (td < 0 || td >= nd) Py_RETURN_NONE;
    int bcast = 0;
    if (!done.is_contiguous() || !term.is_contiguous()) {
        if (!dense_after_expand(done, shape, td) || !dense_after_expand(term, shape, td)) Py_RETURN_NONE;
        bcast = 1;
    }
    if (c10::GradMode::is_enabled() && (nsv.requires_grad() || sv.requires_grad() || r.requires_grad())) Py_RETURN_NONE;
    const int64_t n = nsv.numel();
    if (n == 0) Py_RETURN_NONE;
    const int64_t T = shape[td];
    int64_t I = 1;
    for (int64_t k = td + 1; k < nd; ++k) I *= shape[k];
    const int64_t O = n / (T * I);

    // coefficients as the reference's int32-mask products round them (python owns that rule; one-entry cache)
    const double gamma = PyFloat_AS_DOUBLE(a[0]), lmbda = PyFloat_AS_DOUBLE(a[1]);
    const int ddt = (int)c10::get_default_dtype_as_scalartype();
    if (g_coef.default_dtype != ddt || g_coef.gamma != gamma || g_coef.lmbda != lmbda) {
        PyObject *res = PyObject_CallFunctionObjArgs(g_coef_cb, a[0], a[1], nullptr);
        if (!res) return nullptr;
        if (res == Py_None) return res;
        double gv, lgv;
        int cdt;
        if (!PyArg_ParseTuple(res, "ddi", &gv, &lgv, &cdt)) {
            Py_DECREF(res);
            return nullptr;
        }
        Py_DECREF(res);
        g_coef = {gamma, lmbda, gv, lgv, ddt, cdt};
    }

    PyObject *dev_idx = PyLong_FromLong(dev.index());
    PyObject *sres = PyObject_CallFunctionObjArgs(g_stream_cb, dev_idx, nullptr);
    Py_DECREF(dev_idx);
    if (!sres) return nullptr;
    const uintptr_t stream = (uintptr_t)PyLong_AsUnsignedLongLong(sres);
    Py_DECREF(sres);
    if (PyErr_Occurred()) return nullptr;

    // workspace: the normalising call always has one; the plain call only for the look-back scan of few, very long rows
    uintptr_t ws_ptr = 0;
    size_t ws_bytes = 0;
    if (normalise || (I == 1 && T >= 2048 && O < 4096)) {
        const size_t need = g_ws_bytes(O, T, I);
        Workspace *w = find_ws(dev.index(), stream);
        if (!w || w->bytes < need) {
            PyObject *res = PyObject_CallFunction(g_ws_cb, "iLLL", (int)dev.index(), (long long)O, (long long)T, (long long)I);
            if (!res) return nullptr;
            Py_DECREF(res);
            w = find_ws(dev.index(), stream);
            if (!w || w->bytes < need) {
                PyErr_SetString(PyExc_RuntimeError, "_trl_host: workspace callback did not register a large enough workspace");
                return nullptr;
            }
        }
        ws_ptr = w->ptr;
        ws_bytes = w->bytes;
    }

    at::Tensor adv, vt;
    try {
        adv = at::empty_like(nsv);
        vt = at::empty_like(nsv);
    } catch (const std::exception &e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    const int vdt = dt == at::kFloat ? TRL_F32 : TRL_F64;
    int rc;
    if (normalise)
        rc = g_gae_norm((void *)stream, vdt, g_coef.cdt, r.data_ptr(), sv.data_ptr(), nsv.data_ptr(), (const uint8_t *)done.data_ptr(),
                        (const uint8_t *)term.data_ptr(), bcast, nullptr, nullptr, g_coef.g_val, 0.0, g_coef.lg_val, O, T, I,
                        adv.data_ptr(), vt.data_ptr(), eps, nullptr, (void *)ws_ptr, ws_bytes, (int)algo);
    else
        rc = g_gae_fwd((void *)stream, vdt, g_coef.cdt, r.data_ptr(), sv.data_ptr(), nsv.data_ptr(), (const uint8_t *)done.data_ptr(),
                       (const uint8_t *)term.data_ptr(), bcast, nullptr, nullptr, g_coef.g_val, 0.0, g_coef.lg_val, O, T, I,
                       adv.data_ptr(), vt.data_ptr(), nullptr, (void *)ws_ptr, ws_bytes, (int)algo);
    if (rc) return PyLong_FromLong(rc);
    PyObject *out = PyTuple_New(2);
    if (!out) return nullptr;
    PyTuple_SET_ITEM(out, 0, THPVariable_Wrap(std::move(adv)));
    PyTuple_SET_ITEM(out, 1, THPVariable_Wrap(std::move(vt)));
    return out;
}

PyMethodDef methods[] = {
    {"init", py_init, METH_VARARGS, "init(trl_gae_fwd, trl_gae_normalize_fwd, trl_workspace_bytes, coef_cb, ws_cb, stream_cb)"},
    {"set_workspace", py_set_workspace, METH_VARARGS, "set_workspace(device index, stream, pointer, bytes)"},
    {"gae", (PyCFunction)(void (*)(void))py_gae, METH_FASTCALL,
     "gae(gamma, lmbda, state_value, next_state_value, reward, done, terminated, eps, algo, time_dim)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_trl_host", "host-side fast path of the torchrl_amd GAE entry points", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__trl_host(void) { return PyModule_Create(&module); }
