// Device-side core of the wavefront-scan kernels (shared by scan_kernels.hip and
// fused_kernels.hip).  See scan_kernels.hip for the algorithm description.
#pragma once
#include "common.h"

namespace scan {

constexpr int kWaves = 4;                  // waves that share one chunk (one "team")
constexpr int kTeam = kWaves * kWave;      // 256 lanes

template <typename TC>
struct Aff {
    TC a, b;
    int h;  // hard reset: the incoming value is ignored
};

// apply f2 first, then f1
template <typename TC>
__device__ __forceinline__ Aff<TC> compose(const Aff<TC> &f1, const Aff<TC> &f2) {
    Aff<TC> r;
    const TC t = f1.b + f1.a * f2.b;
    r.a = f1.a * f2.a;
    r.b = f1.h ? f1.b : t;
    r.h = f1.h | f2.h;
    return r;
}
template <typename TC>
__device__ __forceinline__ TC apply(const Aff<TC> &f, TC x) {
    const TC t = f.b + f.a * x;
    return f.h ? f.b : t;
}
// The same two for a chunk that holds no row end (HARD = false): no map carries a hard reset, h stays 0 and is neither
// moved between lanes nor selected on -- a quarter of the scan's instructions (interior chunks of long rows).
template <bool HARD, typename TC>
__device__ __forceinline__ Aff<TC> compose_t(const Aff<TC> &f1, const Aff<TC> &f2) {
    if constexpr (HARD) return compose(f1, f2);
    return Aff<TC>{f1.a * f2.a, f1.b + f1.a * f2.b, 0};
}
template <bool HARD, typename TC>
__device__ __forceinline__ TC apply_t(const Aff<TC> &f, TC x) {
    if constexpr (HARD) return apply(f, x);
    return f.b + f.a * x;
}
template <typename TC>
__device__ __forceinline__ Aff<TC> shfl_down_aff(const Aff<TC> &f, int off) {
    Aff<TC> r;
    r.a = __shfl_down(f.a, off, kWave);
    r.b = __shfl_down(f.b, off, kWave);
    r.h = __shfl_down(f.h, off, kWave);
    return r;
}

// ---- raw vector loads / stores (16 B per lane whenever the element type allows) -------------
template <int BYTES>
struct RawT;
template <>
struct RawT<1> { typedef unsigned char type; };
template <>
struct RawT<2> { typedef unsigned short type; };
template <>
struct RawT<4> { typedef unsigned int type; };
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <>
struct RawT<8> { typedef u32x2 type; };
template <>
struct RawT<16> { typedef u32x4 type; };

template <typename T, int V>
__device__ __forceinline__ void vload(const T *__restrict__ p, T (&out)[V]) {
    constexpr int B = V * (int)sizeof(T);
    if constexpr (B <= 16) {
        typename RawT<B>::type raw = *(const typename RawT<B>::type *)p;
        __builtin_memcpy(out, &raw, B);
    } else {
        static_assert(B % 16 == 0, "vector width");
#pragma unroll
        for (int k = 0; k < B / 16; ++k) {
            u32x4 raw = ((const u32x4 *)p)[k];
            __builtin_memcpy((char *)out + 16 * k, &raw, 16);
        }
    }
}
// Outputs are streamed with non-temporal stores: they are written once and not re-read by the
// kernel, and keeping them out of the L2's dirty set shortens the end-of-kernel write-back
// (measured on the C4 byte mix: 6.4 -> 5.0 us per launch for a pipelined copy).
template <typename T, int V>
__device__ __forceinline__ void vstore(T *__restrict__ p, const T (&in)[V]) {
    constexpr int B = V * (int)sizeof(T);
    if constexpr (B <= 16) {
        typename RawT<B>::type raw;
        __builtin_memcpy(&raw, in, B);
        __builtin_nontemporal_store(raw, (typename RawT<B>::type *)p);
    } else {
#pragma unroll
        for (int k = 0; k < B / 16; ++k) {
            u32x4 raw;
            __builtin_memcpy(&raw, (const char *)in + 16 * k, 16);
            __builtin_nontemporal_store(raw, (u32x4 *)p + k);
        }
    }
}
// The same two for a vector that is only ELEMENT-aligned (rows whose length breaks the 16-byte alignment of chunk / row
// starts: T = 9999, 999, 31 ...).  gfx950 serves global accesses of any width from dword- (byte-) aligned addresses in the
// unaligned access mode ROCm runs in, and the compiler emits one global_load_dwordx4 / global_store_dwordx4 for an
// under-aligned vector type: one instruction per lane instead of V, at the price of some lanes straddling a cache line.
// Measured (GAE, profiles/r03_unaligned_rows.txt): fp32 [1024x9999] 61.2 -> 38.3 us, [8192x999] 53.1 -> 41.3, [16384x301] 37.3 -> 30.0;
// fp64 [1024x9999] 96.8 -> 76.3, [8192x999] 74.2 -> 57.7 (aligned neighbours: 37.5 / 37.7 / 26.8; 75.7 / 56.7).
template <typename T, int V>
constexpr bool vec_width_ok() {  // widths one (or a few 16-byte) access instruction(s) can take
    constexpr int B = V * (int)sizeof(T);
    return B == 1 || B == 2 || B == 4 || B == 8 || (B >= 16 && B % 16 == 0);
}
template <typename R, int A>
struct UnderAligned;
template <typename R>
struct UnderAligned<R, 1> { typedef R type __attribute__((aligned(1))); };
template <typename R>
struct UnderAligned<R, 2> { typedef R type __attribute__((aligned(2))); };
template <typename R>
struct UnderAligned<R, 4> { typedef R type __attribute__((aligned(4))); };
template <typename R>
struct UnderAligned<R, 8> { typedef R type __attribute__((aligned(8))); };
template <typename T, int V>
__device__ __forceinline__ void vload_u(const T *__restrict__ p, T (&out)[V]) {
    constexpr int B = V * (int)sizeof(T);
    if constexpr (B <= 16) {
        typedef typename RawT<B>::type R;
        typedef typename UnderAligned<R, (int)alignof(T)>::type RU;
        const R raw = *(const RU *)p;
        __builtin_memcpy(out, &raw, B);
    } else {
        static_assert(B % 16 == 0, "vector width");
        typedef typename UnderAligned<u32x4, (int)alignof(T)>::type RU;
#pragma unroll
        for (int k = 0; k < B / 16; ++k) {
            const u32x4 raw = ((const RU *)p)[k];
            __builtin_memcpy((char *)out + 16 * k, &raw, 16);
        }
    }
}
template <typename T, int V>
__device__ __forceinline__ void vstore_u(T *__restrict__ p, const T (&in)[V]) {
    constexpr int B = V * (int)sizeof(T);
    if constexpr (B <= 16) {
        typedef typename RawT<B>::type R;
        typedef typename UnderAligned<R, (int)alignof(T)>::type RU;
        R raw;
        __builtin_memcpy(&raw, in, B);
        __builtin_nontemporal_store(raw, (RU *)p);
    } else {
        typedef typename UnderAligned<u32x4, (int)alignof(T)>::type RU;
#pragma unroll
        for (int k = 0; k < B / 16; ++k) {
            u32x4 raw;
            __builtin_memcpy(&raw, (const char *)in + 16 * k, 16);
            __builtin_nontemporal_store(raw, (RU *)p + k);
        }
    }
}
// VEC: the launcher established that every full vector is 16-byte aligned; otherwise full vectors are element-aligned
template <bool VEC, typename T, int V>
__device__ __forceinline__ void load_n(const T *__restrict__ p, i64 flat, int nvalid, T (&out)[V]) {
    if constexpr (VEC || vec_width_ok<T, V>()) {
        if (nvalid == V) {
            if constexpr (VEC) vload<T, V>(p + flat, out);
            else vload_u<T, V>(p + flat, out);
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = e < nvalid ? p[flat + e] : (T)0;
}
template <bool VEC, typename T, int V>
__device__ __forceinline__ void store_n(T *__restrict__ p, i64 flat, int nvalid, const T (&in)[V]) {
    if constexpr (VEC || vec_width_ok<T, V>()) {
        if (nvalid == V) {
            if constexpr (VEC) vstore<T, V>(p + flat, in);
            else vstore_u<T, V>(p + flat, in);
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e)
        if (e < nvalid) p[flat + e] = in[e];
}

// Reversed addressing (adjoint ops): the lane's element e of the REVERSED flat order, flat + e, sits at memory index
// n_total - 1 - (flat + e).  A lane's V elements are still one contiguous, naturally aligned vector (n_total % V == 0
// is the launcher's condition for VEC), read once and turned around in registers; a wave still touches 64*V
// consecutive elements per instruction.
template <bool VEC, typename T, int V>
__device__ __forceinline__ void load_n_rev(const T *__restrict__ p, i64 n_total, i64 flat, int nvalid, T (&out)[V]) {
    if constexpr (VEC || vec_width_ok<T, V>()) {
        if (nvalid == V) {
            T tmp[V];
            if constexpr (VEC) vload<T, V>(p + (n_total - flat - V), tmp);
            else vload_u<T, V>(p + (n_total - flat - V), tmp);
#pragma unroll
            for (int e = 0; e < V; ++e) out[e] = tmp[V - 1 - e];
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = e < nvalid ? p[n_total - 1 - flat - e] : (T)0;
}
template <bool VEC, typename T, int V>
__device__ __forceinline__ void store_n_rev(T *__restrict__ p, i64 n_total, i64 flat, int nvalid, const T (&in)[V]) {
    if constexpr (VEC || vec_width_ok<T, V>()) {
        if (nvalid == V) {
            T tmp[V];
#pragma unroll
            for (int e = 0; e < V; ++e) tmp[e] = in[V - 1 - e];
            if constexpr (VEC) vstore<T, V>(p + (n_total - flat - V), tmp);
            else vstore_u<T, V>(p + (n_total - flat - V), tmp);
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e)
        if (e < nvalid) p[n_total - 1 - flat - e] = in[e];
}

// 16-byte write-through store (global_store_dwordx4 sc1); other widths (ragged tails) as plain element stores.
// The s_nop: a store of more than 64 bits reads its data registers one wait state after issue, and the compiler
// does not know this asm is such a store (it may overwrite the registers with the very next instruction).
template <bool VEC, typename T, int V>
__device__ __forceinline__ void store_wt(T *__restrict__ p, i64 flat, int nvalid, const T (&in)[V]) {
    if constexpr (V * sizeof(T) == 16) {  // any element-aligned address (see vload_u)
        if (nvalid == V) {
            u32x4 raw;
            __builtin_memcpy(&raw, in, 16);
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + flat), "v"(raw) : "memory");
            return;
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e)
        if (e < nvalid) p[flat + e] = in[e];
}

template <typename T>
__device__ __forceinline__ T clamp_max_nan(T x, T hi) { return (x != x) ? x : (x > hi ? hi : x); }
__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }

// ---- shifted value layout (I == 1) -----------------------------------------------------------
// The value network was evaluated once on T + 1 steps per row (reference: advantages.py:440-490,
// `shifted=True`): state_value and next_state_value are the two T-step windows [0, T) and [1, T + 1)
// of ONE (O, T + 1) buffer.  The kernels then read that buffer in place: element `flat` of the (O, T)
// index space sits at flat + flat / T, its successor right behind it, so a lane that owns V steps of
// one row issues V + 1 scalar loads instead of two V-wide ones and the value bytes come from HBM once
// (GAE fp32: 22 -> 18 B per transition).  Rows are T + 1 elements apart, hence never 16-byte
// aligned as a whole: these loads stay scalar.
// flat / T without the 64-bit integer division (whose expansion kept the callers' register arrays in scratch):
// exact for flat < 2^52 after one correction step.
__device__ __forceinline__ i64 row_of(i64 flat, i64 T) {
    if ((((unsigned long long)flat | (unsigned long long)T) >> 32) == 0) return (i64)((unsigned)flat / (unsigned)T);
    i64 row = (i64)((double)flat / (double)T);
    const i64 rem = flat - row * T;
    row += rem >= T ? 1 : (rem < 0 ? -1 : 0);
    return row;
}
template <typename TS, int V>
__device__ __forceinline__ void load_shifted_pair(const TS *__restrict__ buf, i64 flat, i64 T, int nvalid, TS (&v)[V],
                                                  TS (&nv)[V]) {
    const i64 row = row_of(flat, T);
    unsigned t = (unsigned)(flat - row * T);
    const TS *__restrict__ p = buf + flat + row;
    if (nvalid == V && (i64)t + V <= T) {  // the lane's V steps share a row: V + 1 consecutive elements
        // (fetching the two 16-byte granules that hold the window with vector loads and picking it with selects
        // was measured slower than these V + 1 scalar loads: 9.69 vs 9.40 us for the fused kernel at [65536 x 32]; one
        // element-aligned vector load (vload_u) + 1 scalar: no difference, 8.23 vs 8.29 us)
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = p[e];
#pragma unroll
        for (int e = 0; e + 1 < V; ++e) nv[e] = v[e + 1];
        nv[V - 1] = p[V];
        return;
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {  // ragged: a row ends inside the lane's steps (one buffer element is skipped there)
        const bool ok = e < nvalid;
        v[e] = ok ? p[0] : (TS)0;
        nv[e] = ok ? p[1] : (TS)0;
        ++t;
        p += (t == (unsigned)T) ? 2 : 1;
        t = (t == (unsigned)T) ? 0u : t;
    }
}

// ---- per-estimator element logic -------------------------------------------------------------
// Each Op holds one lane's V consecutive time steps.  Interface used by the scan driver:
//   Raw, for_each_input(args, raw, f)  the input tensors and the lane's V raw elements of each
//                                   (load_inputs<Op, VEC> issues the global loads and nothing else,
//                                   which lets the caller prefetch the next chunk)
//   prepare(args, raw)              fuse the delta / coefficient math
//   init(e)                         value the recurrence starts from at a row end
//   step(e, x)                      x_t from x_{t+1}: the reference's exact per-step expression
//   affine(e, a, b)                 the same step as an affine map (used for aggregates only)
//   first_value()                   (V-trace) state_value of the lane's first step
//   TO, kNumOut, out_ptr(args, k)   the output tensors
//   outputs(...)                    epilogue: the lane's V output elements (+ fp64 moments for GAE)

template <typename TS_, typename TG_, int V_>
struct GaeOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef GaeArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false;
    static constexpr bool kShifted = false;  // see Shifted<> below
    static constexpr bool kReversed = false;    // see the adjoint ops (bwd_ops.h)
    static constexpr bool kCustomEmit = false;
    // look-back kernel (scan_impl.h): two 1024-thread blocks per CU (8 waves per SIMD) when the op fits 64 VGPRs (fp32: 70
    // without the hint)
    static constexpr int kLbWavesPerSimd = (sizeof(TS_) == 4 && sizeof(TG_) == 4) ? 8 : 1;
    struct Raw {
        TS r[V], v[V], nv[V];
        u8 d[V], tm[V];
        TG g[V], lg[V];
    };
    template <class F>
    static __device__ __forceinline__ void for_each_unshifted_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.reward, w.r);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);
        f((const TG *)a.lg_t, w.lg);
    }
    static __device__ __forceinline__ void load_shifted(const Args &a, i64 flat, int nvalid, Raw &w) {
        load_shifted_pair<TS, V>((const TS *)a.value, flat, a.T, nvalid, w.v, w.nv);
    }
    TS v[V];
    TC delta[V], disc[V];

    typedef TS TO;
    static constexpr int kNumOut = 2;
    static __device__ __forceinline__ TO *out_ptr(const Args &a, int k) { return (TO *)(k == 0 ? a.advantage : a.value_target); }
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.reward, w.r);
        f((const TS *)a.value, w.v);
        f((const TS *)a.next_value, w.nv);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);   // optional (null -> skipped)
        f((const TG *)a.lg_t, w.lg);     // optional
    }
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
        const TG gs = (TG)a.gamma, ls = (TG)a.lmbda, lgs = (TG)a.lg;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = a.gamma_t ? w.g[e] : gs;
            const TG ce = a.lg_t ? w.lg[e] : (a.gamma_t ? ls * ge : lgs);
            const TG g_nt = ge * (TG)(w.tm[e] ? 0 : 1);  // multiply, not select: as the reference
            v[e] = w.v[e];
            disc[e] = (TC)(ce * (TG)(w.d[e] ? 0 : 1));
            delta[e] = ((TC)w.r[e] + (TC)g_nt * (TC)w.nv[e]) - (TC)w.v[e];
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return delta[e] + x * disc[e]; }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = disc[e]; b = delta[e]; }
    __device__ __forceinline__ TS first_value() const { return v[0]; }

    // advantage / value_target of the lane's V steps + fp64 moments of the advantage
    __device__ __forceinline__ void finish(int nvalid, const TC (&x)[V], TS (&adv)[V], TS (&vt)[V], double &s,
                                           double &ss) const {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            adv[e] = (TS)x[e];
            vt[e] = adv[e] + v[e];
            if (e < nvalid) {
                s += (double)adv[e];
                ss += (double)adv[e] * (double)adv[e];
            }
        }
    }
    __device__ __forceinline__ void outputs(int nvalid, const TC (&x)[V], const bool (&)[V], TC, TS, TO (&o0)[V],
                                            TO (&o1)[V], double &s, double &ss) const {
        finish(nvalid, x, o0, o1, s, ss);
    }
    // lock-step layout (lockstep_impl.h): the lane's V elements are the V feature elements of ONE time step
    __device__ __forceinline__ TS value(int e) const { return v[e]; }
    __device__ __forceinline__ void outputs_ls(bool valid, const TC (&x)[V], bool, const TC (&)[V], const TS (&)[V], TO (&o0)[V],
                                               TO (&o1)[V], double &s, double &ss) const {
        finish(valid ? V : 0, x, o0, o1, s, ss);
    }
};

template <typename TS_, typename TG_, int V_>
struct TdLambdaOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TdLambdaArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false;
    static constexpr bool kShifted = false;
    static constexpr bool kReversed = false;
    static constexpr bool kCustomEmit = false;
    struct Raw {
        TS r[V], nv[V];
        u8 d[V], tm[V];
        TG g[V], l[V];
    };
    TS r[V], nvm[V], dn[V], omd[V];
    TG gm[V], lm[V];

    typedef TS TO;
    static constexpr int kNumOut = 1;
    static __device__ __forceinline__ TO *out_ptr(const Args &a, int) { return (TO *)a.returns; }
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.reward, w.r);
        f((const TS *)a.next_value, w.nv);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);   // optional
        f((const TG *)a.lmbda_t, w.l);   // optional
    }
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            r[e] = w.r[e];
            gm[e] = a.gamma_t ? w.g[e] : (TG)a.gamma;
            lm[e] = a.lmbda_t ? w.l[e] : (TG)a.lmbda;
            nvm[e] = w.nv[e] * (TS)(w.tm[e] ? 0 : 1);
            dn[e] = (TS)(w.d[e] ? 1 : 0);
            omd[e] = (TS)(w.d[e] ? 0 : 1);
        }
    }
    // g starts from v'_{T-1} (functional.py:866)
    __device__ __forceinline__ TC init(int e) const { return (TC)nvm[e]; }
    __device__ __forceinline__ TC step(int e, TC x) const {
        const TG oml = (TG)1 - lm[e];
        const TC g1 = x * (TC)omd[e] + (TC)(nvm[e] * dn[e]);
        return (TC)r[e] + (TC)gm[e] * ((TC)oml * (TC)nvm[e] + (TC)lm[e] * g1);
    }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const {
        const TG oml = (TG)1 - lm[e];
        a = ((TC)gm[e] * (TC)lm[e]) * (TC)omd[e];
        b = (TC)r[e] + (TC)gm[e] * ((TC)oml * (TC)nvm[e] + (TC)lm[e] * (TC)(nvm[e] * dn[e]));
    }
    __device__ __forceinline__ TS first_value() const { return (TS)0; }

    __device__ __forceinline__ void outputs(int, const TC (&x)[V], const bool (&)[V], TC, TS, TO (&o0)[V], TO (&)[V],
                                            double &, double &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) o0[e] = (TS)x[e];
    }
    __device__ __forceinline__ TS value(int) const { return (TS)0; }
    __device__ __forceinline__ void outputs_ls(bool, const TC (&x)[V], bool, const TC (&)[V], const TS (&)[V], TO (&o0)[V],
                                               TO (&)[V], double &, double &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) o0[e] = (TS)x[e];
    }
};

// TD(1) return, rolling gamma (functional.py:515-549): g starts from the RAW next value of the row's last step,
//   g = r + (gamma*nt) * ((1 - dnt)*g + dnt*v'),  dnt = done & ~terminated.
// Same tensors as TD(lambda) (TdLambdaArgs; lmbda unused).
template <typename TS_, typename TG_, int V_>
struct Td1Op {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TdLambdaArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false;
    static constexpr bool kShifted = false;
    static constexpr bool kReversed = false;
    static constexpr bool kCustomEmit = false;
    struct Raw {
        TS r[V], nv[V];
        u8 d[V], tm[V];
        TG g[V];
    };
    TS r[V], nv[V], dnt[V], omdnt[V];
    TG gnt[V];

    typedef TS TO;
    static constexpr int kNumOut = 1;
    static __device__ __forceinline__ TO *out_ptr(const Args &a, int) { return (TO *)a.returns; }
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.reward, w.r);
        f((const TS *)a.next_value, w.nv);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);   // optional
    }
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            r[e] = w.r[e];
            nv[e] = w.nv[e];
            gnt[e] = (a.gamma_t ? w.g[e] : (TG)a.gamma) * (TG)(w.tm[e] ? 0 : 1);
            const bool b = w.d[e] && !w.tm[e];
            dnt[e] = (TS)(b ? 1 : 0);
            omdnt[e] = (TS)(b ? 0 : 1);
        }
    }
    __device__ __forceinline__ TC init(int e) const { return (TC)nv[e]; }
    __device__ __forceinline__ TC step(int e, TC x) const {
        return (TC)r[e] + (TC)gnt[e] * ((TC)omdnt[e] * x + (TC)(dnt[e] * nv[e]));
    }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const {
        a = (TC)gnt[e] * (TC)omdnt[e];
        b = (TC)r[e] + (TC)gnt[e] * (TC)(dnt[e] * nv[e]);
    }
    __device__ __forceinline__ TS first_value() const { return (TS)0; }
    __device__ __forceinline__ void outputs(int, const TC (&x)[V], const bool (&)[V], TC, TS, TO (&o0)[V], TO (&)[V],
                                            double &, double &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) o0[e] = (TS)x[e];
    }
    __device__ __forceinline__ TS value(int) const { return (TS)0; }
    __device__ __forceinline__ void outputs_ls(bool, const TC (&x)[V], bool, const TC (&)[V], const TS (&)[V], TO (&o0)[V],
                                               TO (&)[V], double &, double &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) o0[e] = (TS)x[e];
    }
};

template <typename TS_, typename TG_, int V_>
struct VtraceOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef VtraceArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = true;
    static constexpr bool kShifted = false;
    static constexpr bool kReversed = false;
    static constexpr bool kCustomEmit = false;
    struct Raw {
        TS lp[V], lmu[V], v[V], nv[V], r[V];
        u8 d[V], tm[V];
        TG g[V];
    };
    template <class F>
    static __device__ __forceinline__ void for_each_unshifted_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.log_pi, w.lp);
        f((const TS *)a.log_mu, w.lmu);
        f((const TS *)a.reward, w.r);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);
    }
    static __device__ __forceinline__ void load_shifted(const Args &a, i64 flat, int nvalid, Raw &w) {
        load_shifted_pair<TS, V>((const TS *)a.value, flat, a.T, nvalid, w.v, w.nv);
    }
    TS v[V], nv[V], r[V], crho[V];
    TG td[V];
    TC coef[V], delta[V];

    typedef TC TO;
    static constexpr int kNumOut = 2;
    static __device__ __forceinline__ TO *out_ptr(const Args &a, int k) { return (TO *)(k == 0 ? a.advantage : a.vs); }
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.log_pi, w.lp);
        f((const TS *)a.log_mu, w.lmu);
        f((const TS *)a.value, w.v);
        f((const TS *)a.next_value, w.nv);
        f((const TS *)a.reward, w.r);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);   // optional
    }
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
        const TS rho_hi = (TS)a.rho_thresh, c_hi = (TS)a.c_thresh;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            v[e] = w.v[e];
            nv[e] = w.nv[e];
            r[e] = w.r[e];
            const TG ge = a.gamma_t ? w.g[e] : (TG)a.gamma;
            const TG dd = ge * (TG)(w.d[e] ? 0 : 1);
            td[e] = ge * (TG)(w.tm[e] ? 0 : 1);
            const TS rho = exp_t(w.lp[e] - w.lmu[e]);
            crho[e] = clamp_max_nan(rho, rho_hi);
            const TS cc = clamp_max_nan(rho, c_hi);
            delta[e] = (TC)crho[e] * (((TC)r[e] + (TC)td[e] * (TC)nv[e]) - (TC)v[e]);
            coef[e] = (TC)dd * (TC)cc;
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return delta[e] + coef[e] * x; }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = coef[e]; b = delta[e]; }
    __device__ __forceinline__ TS first_value() const { return v[0]; }

    // x_next / v_next: x and state_value of the step that follows the lane's last step
    // (only read when that step is not a row end).
    __device__ __forceinline__ void outputs(int nvalid, const TC (&x)[V], const bool (&row_end)[V], TC x_next, TS v_next,
                                            TO (&adv)[V], TO (&vs)[V], double &, double &) const {
        TC vs_next = x_next + (TC)v_next;
#pragma unroll
        for (int e = V - 1; e >= 0; --e) {
            vs[e] = x[e] + (TC)v[e];
            const TC nxt = row_end[e] ? (TC)nv[e] : vs_next;  // shift ignores done, functional.py:1343-1345
            adv[e] = (TC)crho[e] * (((TC)r[e] + (TC)td[e] * nxt) - (TC)v[e]);
            if (e < nvalid) vs_next = vs[e];
        }
    }
    // lock-step layout: x_next[e] / v_next[e] are x and state_value of element e of the NEXT time step
    __device__ __forceinline__ TS value(int e) const { return v[e]; }
    __device__ __forceinline__ void outputs_ls(bool, const TC (&x)[V], bool row_end, const TC (&x_next)[V], const TS (&v_next)[V],
                                               TO (&adv)[V], TO (&vs)[V], double &, double &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            vs[e] = x[e] + (TC)v[e];
            const TC nxt = row_end ? (TC)nv[e] : x_next[e] + (TC)v_next[e];
            adv[e] = (TC)crho[e] * (((TC)r[e] + (TC)td[e] * nxt) - (TC)v[e]);
        }
    }
};

// The same estimator reading the shifted value layout (load_shifted_pair): a separate kernel instantiation, so
// that the dense kernels keep their register budget (a run-time switch inside one kernel cost the fused
// GAE + normalise kernel its occupancy: 9.3 -> 20.7 us at [65536 x 32]).
template <class Base>
struct Shifted : Base {
    static constexpr bool kShifted = true;
};
template <typename TS, typename TG, int V>
using GaeShiftedOp = Shifted<GaeOp<TS, TG, V>>;
template <typename TS, typename TG, int V>
using VtraceShiftedOp = Shifted<VtraceOp<TS, TG, V>>;

// ---- generic load / store over an Op's tensor lists ------------------------------------------------
template <class Op, bool VEC>
__device__ __forceinline__ void load_inputs(const typename Op::Args &a, i64 flat, int nvalid, typename Op::Raw &w) {
    if constexpr (Op::kReversed) {
        const i64 n_total = a.O * a.T;  // reversed ops are I == 1 only
        Op::for_each_input(a, w, [&](auto *base, auto &dst) {
            if (base) load_n_rev<VEC>(base, n_total, flat, nvalid, dst);
        });
        Op::load_extra(a, n_total, flat, nvalid, w);
        return;
    }
    if constexpr (Op::kShifted) {
        Op::for_each_unshifted_input(a, w, [&](auto *base, auto &dst) {
            if (base) load_n<VEC>(base, flat, nvalid, dst);
        });
        Op::load_shifted(a, flat, nvalid, w);
        return;
    }
    Op::for_each_input(a, w, [&](auto *base, auto &dst) {
        if (base) load_n<VEC>(base, flat, nvalid, dst);
    });
}
template <class Op, bool VEC>
__device__ __forceinline__ void store_outputs(const typename Op::Args &a, i64 flat, int nvalid,
                                              const typename Op::TO (&o0)[Op::V], const typename Op::TO (&o1)[Op::V]) {
    store_n<VEC>(Op::out_ptr(a, 0), flat, nvalid, o0);
    if (Op::kNumOut > 1) store_n<VEC>(Op::out_ptr(a, 1), flat, nvalid, o1);
}

// epilogue of a lane's V steps: the estimator's outputs, stored (adjoint ops write their own, reversed, set)
template <class Op, bool VEC>
__device__ __forceinline__ void emit_outputs(const Op &op, const typename Op::Args &a, i64 flat, int nvalid,
                                             const typename Op::TC (&xs)[Op::V], const bool (&row_end)[Op::V],
                                             typename Op::TC x_in, typename Op::TS v_next, double &s, double &ss) {
    if constexpr (Op::kCustomEmit) {
        op.template emit<VEC>(a, flat, nvalid, xs, row_end, x_in, v_next);
    } else {
        typename Op::TO o0[Op::V], o1[Op::V];
        op.outputs(nvalid, xs, row_end, x_in, v_next, o0, o1, s, ss);
        store_outputs<Op, VEC>(a, flat, nvalid, o0, o1);
    }
}

// ---- scan of one chunk by one team (kWaves waves) ---------------------------------------------
template <class Op>
struct TeamShared {
    typename Op::TC a[2][kWaves], b[2][kWaves];
    int h[2][kWaves];
    typename Op::TS v0[2][kWaves];
};

// Geometry of one lane's slice of a chunk.
struct LaneGeom {
    i64 flat;      // flat index of the lane's first element
    int nvalid;    // how many of its V elements exist
    unsigned t0;   // time index of the lane's first element (mod T applied by the caller)
};

// Scan one chunk whose per-lane element data sits in `op` (already prepared).
//   X (in/out): carry.  On entry x (and V-trace's state_value) of the element that follows the
//   chunk; on exit those of the chunk's first element.  Ignored when WAVE_INDEP.
//   scan_lanes: upper bound on the number of consecutive lanes one row spans (64 = no bound).
//   emit(xs, row_end, x_in, v_next): consumes the per-step results.
//   carry_fn(sh, buf, Xc, Xv): called by every thread once the wave aggregates are in LDS; a kernel
//   that learns its carry from other blocks (decoupled look-back) resolves Xc / Xv there.
// MODE of scan_chunk_t:
//   kScanGeneral   rows may end anywhere: maps carry the hard-reset flag h
//   kScanNoRowEnd  the caller knows the chunk holds no row end (interior chunk of a long row): no h, no row-end logic
// (A third form -- rows as aligned power-of-two lane groups, steps bounded by the group instead of guarded by h -- was
// measured in the fused kernel at C4: no gain, 7.87 vs 7.77 us per step; dropped.)
constexpr int kScanGeneral = 0, kScanNoRowEnd = 1;

template <class Op, bool WAVE_INDEP, int MODE, class Emit, class CarryFn>
__device__ __forceinline__ void scan_chunk_t(const Op &op, TeamShared<Op> &sh, int buf, int nvalid, unsigned t0, unsigned T,
                                           int scan_lanes, typename Op::TC &Xc, typename Op::TS &Xv, Emit &&emit,
                                           CarryFn &&carry_fn) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr bool HARD = MODE == kScanGeneral;
    const int lane = threadIdx.x & (kWave - 1), w = (threadIdx.x / kWave) & (kWaves - 1);

    // time index of each element -> row-end flags
    bool row_end[V];
    {
        unsigned t = t0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            row_end[e] = MODE != kScanNoRowEnd && (t == T - 1);
            t = row_end[e] ? 0u : t + 1u;
        }
    }

    // 1. lane aggregate
    Aff<TC> agg{(TC)1, (TC)0, 0};
#pragma unroll
    for (int e = V - 1; e >= 0; --e) {
        if (e < nvalid) {
            Aff<TC> el;
            if (row_end[e]) {
                el.a = (TC)0;
                el.b = op.step(e, op.init(e));
                el.h = 1;
            } else {
                op.affine(e, el.a, el.b);
                el.h = 0;
            }
            agg = (e == V - 1) ? el : compose_t<HARD>(el, agg);
        }
    }

    // 2. inclusive suffix scan over the wave (only as far as a row can reach): offsets 1..8 move
    //    data with DPP row shifts inside each 16-lane row, then the rows are chained with two
    //    cross-row steps (the totals of the following rows sit in lanes 16/32/48).
    Aff<TC> S = agg;
    const int lir = lane & 15;
#define TRL_DPP_STEP(OFF)                                                          \
    if (scan_lanes > OFF) {                                                        \
        Aff<TC> p;                                                                 \
        p.a = dpp_mov<0x100 + OFF>(S.a);                                           \
        p.b = dpp_mov<0x100 + OFF>(S.b);                                           \
        p.h = HARD ? dpp_mov<0x100 + OFF>(S.h) : 0;                                \
        if (lir + OFF < 16) S = compose_t<HARD>(S, p);                             \
    }
    TRL_DPP_STEP(1)
    TRL_DPP_STEP(2)
    TRL_DPP_STEP(4)
    TRL_DPP_STEP(8)
#undef TRL_DPP_STEP
    if (scan_lanes > 16) {
        // rows 0 and 2 absorb the totals of rows 1 and 3 (lanes 16 and 48), then rows 0 and 1 absorb rows 2..3 (lane 32).
        // The sources are three fixed lanes: v_readlane + select instead of ds_bpermute shuffles (same values, same
        // composition order; ~2 x 120 cycles less latency per chunk on the chain scan -> carry -> replay).
        const int row = lane >> 4;
        const Aff<TC> p16{read_lane(S.a, 16), read_lane(S.b, 16), HARD ? read_lane(S.h, 16) : 0};
        const Aff<TC> p48{read_lane(S.a, 48), read_lane(S.b, 48), HARD ? read_lane(S.h, 48) : 0};
        if ((row & 1) == 0) S = compose_t<HARD>(S, row == 0 ? p16 : p48);
        const Aff<TC> p32{read_lane(S.a, 32), read_lane(S.b, 32), HARD ? read_lane(S.h, 32) : 0};
        if (row < 2) S = compose_t<HARD>(S, p32);
    }
    // exclusive: everything after this lane, within the wave (whole-wave shift by one lane)
    Aff<TC> E{dpp_mov<0x130>(S.a), dpp_mov<0x130>(S.b), HARD ? dpp_mov<0x130>(S.h) : 0};
    if (lane == kWave - 1) E = Aff<TC>{(TC)1, (TC)0, 0};
    TS v_next = Op::kNeedsNext ? dpp_mov<0x130>(op.first_value()) : (TS)0;

    // 3. carry into this wave
    TC Cw;
    if constexpr (WAVE_INDEP) {
        // every wave's range ends at a row end: the carry is never read
        Cw = (TC)0;
    } else {
        if (lane == 0) {
            sh.a[buf][w] = S.a;
            sh.b[buf][w] = S.b;
            if (HARD) sh.h[buf][w] = S.h;
            if (Op::kNeedsNext) sh.v0[buf][w] = op.first_value();
        }
        __syncthreads();
        carry_fn(sh, buf, Xc, Xv);
        TC x = Xc;
        TS xv = Xv;
        Cw = x;
        TS Cv = xv;
#pragma unroll
        for (int k = kWaves - 1; k >= 0; --k) {
            if (k == w) { Cw = x; Cv = xv; }
            const Aff<TC> g{sh.a[buf][k], sh.b[buf][k], HARD ? sh.h[buf][k] : 0};
            x = apply_t<HARD>(g, x);
            if (Op::kNeedsNext) xv = sh.v0[buf][k];
        }
        Xc = x;  // x of the chunk's first element
        Xv = xv;
        if (Op::kNeedsNext && lane == kWave - 1) v_next = Cv;
    }

    // 4. exact per-step replay from the lane's carry-in
    TC xs[V];
    const TC x_in = apply_t<HARD>(E, Cw);
    TC xp = x_in;
#pragma unroll
    for (int e = V - 1; e >= 0; --e) {
        const TC xi = row_end[e] ? op.init(e) : xp;
        xs[e] = op.step(e, xi);
        if (e < nvalid) xp = xs[e];
    }
    emit(xs, row_end, x_in, v_next);
}

template <class Op, bool WAVE_INDEP, class Emit, class CarryFn>
__device__ __forceinline__ void scan_chunk(const Op &op, TeamShared<Op> &sh, int buf, int nvalid, unsigned t0, unsigned T,
                                           int scan_lanes, typename Op::TC &Xc, typename Op::TS &Xv, Emit &&emit,
                                           CarryFn &&carry_fn) {
    scan_chunk_t<Op, WAVE_INDEP, kScanGeneral>(op, sh, buf, nvalid, t0, T, scan_lanes, Xc, Xv, emit, carry_fn);
}

template <class Op, bool WAVE_INDEP, class Emit>
__device__ __forceinline__ void scan_chunk(const Op &op, TeamShared<Op> &sh, int buf, int nvalid, unsigned t0, unsigned T,
                                           int scan_lanes, typename Op::TC &Xc, typename Op::TS &Xv, Emit &&emit) {
    scan_chunk<Op, WAVE_INDEP>(op, sh, buf, nvalid, t0, T, scan_lanes, Xc, Xv, emit,
                               [](TeamShared<Op> &, int, typename Op::TC &, typename Op::TS &) {});
}

// lanes one row can span inside a wave (bounds the shuffle steps); 64 when unknown
inline int scan_lanes_for(i64 T, int V, bool wave_indep) {
    if (!wave_indep || T % V != 0) return kWave;
    const i64 l = T / V;
    return l < kWave ? (int)l : kWave;
}

}  // namespace scan
