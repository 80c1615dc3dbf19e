// Round-2 ablation harness for the fused GAE + normalise kernel (development tool; not part of the product).
// Variants of the gather window at C4 [65536 x 32] fp32; every variant's outputs are checked against variant 0.
//   bit 0 (1): the LAST chunk's value_target is stored between publish and poll by waves 1..15 (wave 0 after the gather)
//   bit 1 (2): ALL chunks' value_target stored in that window
//   bit 2 (4): one 16-byte granule per block (one sc1 dwordx4 store, one dwordx4 poll load per block)
//   bit 3 (8): two-level gather: wave 0 of blocks b % 32 == 0 ... (not built)
#include "../../torchrl_amd/csrc/scan_core.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <math.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

using namespace scan;
constexpr i64 N = 65536LL * 32;
typedef GaeOp<float, float, 4> Op;
constexpr int kTeams = 4, kThreads = 1024, kMaxBlocks = 512;

struct Ws {
    unsigned epoch, error, pad[62];
    unsigned long long slots[1];
};
struct Geom {
    i64 O, T;
    int rows_per_chunk, scan_lanes;
    i64 nchunks;
    double n_total, eps;
    int slot_stride;  // 8-byte units between blocks' granules
    int replicas;
};

__device__ __forceinline__ unsigned long long tag_bits(double x, unsigned tag) {
    return ((unsigned long long)__double_as_longlong(x) & ~0xffull) | (unsigned long long)tag;
}
__device__ __forceinline__ double untag(unsigned long long b) { return __longlong_as_double((long long)(b & ~0xffull)); }

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_sc1(unsigned long long *p, unsigned long long a, unsigned long long b) {
    u64x2 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// store flavours: 0 nt (product), 1 plain, 2 sc1 (write-through), 3 sc0 sc1, 4 nt sc1
template <int FL>
__device__ __forceinline__ void fstore(float *p, const float (&in)[4]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {in[0], in[1], in[2], in[3]};
    if constexpr (FL == 0) __builtin_nontemporal_store(v, (f4 *)p);
    else if constexpr (FL == 1) *(f4 *)p = v;
    // s_nop: a store of more than 64 bits reads its data VGPRs one wait state after issue, and the compiler does not
    // know this asm is such a store (it may overwrite the registers in the next instruction)
    else if constexpr (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int K, int VAR, int SVT = 0, int SADV = 0>
__global__ __launch_bounds__(kThreads) void fused2(GaeArgs args, Geom geo, Ws *ws, unsigned long long *dbg) {
    typedef float TC;
    typedef float TS;
    constexpr int V = 4;
    constexpr int NW = kThreads / kWave;
    constexpr bool kDeferLast = (VAR & 1) != 0, kDeferAll = (VAR & 2) != 0, kWide = (VAR & 4) != 0;
    constexpr bool kStamp = (VAR & 64) != 0;
    __shared__ TeamShared<Op> sh[kTeams];
    __shared__ double red[2 * NW];
    __shared__ double bc[3];
    const int team = threadIdx.x / kTeam, tid = threadIdx.x & (kTeam - 1);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int lane_off = tid * V;
    const unsigned T = (unsigned)geo.T;
    const unsigned tag = (ws->epoch + 1u) & 0xffu;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (kStamp) st0 = wall_clock64();

    TS adv_keep[K][V];
    TS vt_keep[(kDeferAll ? K : 1)][V];
    i64 flat_k[K];
    auto geom_of = [&](int k, i64 &start) {
        const i64 c = ((i64)k * gridDim.x + blockIdx.x) * kTeams + team;
        start = c * geo.rows_per_chunk * geo.T;
    };
    double s = 0.0, ss = 0.0;
    typename Op::Raw raw[2];
    {
        i64 start;
        geom_of(0, start);
        load_inputs<Op, true>(args, start + lane_off, V, raw[0]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        i64 start;
        if (k + 1 < K) {
            geom_of(k + 1, start);
            load_inputs<Op, true>(args, start + lane_off, V, raw[(k + 1) & 1]);
        }
        geom_of(k, start);
        const i64 flat = start + lane_off;
        flat_k[k] = flat;
        Op op;
        op.prepare(args, raw[k & 1]);
        TC Xc = 0;
        TS Xv = 0;
        scan_chunk<Op, true>(op, sh[team], k & 1, V, (unsigned)lane_off % T, T, geo.scan_lanes, Xc, Xv,
                             [&](const TC(&xs)[V], const bool(&)[V], TC, TS) {
                                 if (kDeferAll || (kDeferLast && k == K - 1)) {
                                     op.finish(V, xs, adv_keep[k], vt_keep[kDeferAll ? k : 0], s, ss);
                                 } else {
                                     TS vt[V];
                                     op.finish(V, xs, adv_keep[k], vt, s, ss);
                                     fstore<SVT>((TS *)args.value_target + flat, vt);
                                 }
                             });
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) { red[2 * wv] = s; red[2 * wv + 1] = ss; }
    __syncthreads();
    auto store_deferred = [&] {
        if (kDeferAll) {
#pragma unroll
            for (int k = 0; k < K; ++k) fstore<SVT>((TS *)args.value_target + flat_k[k], vt_keep[k]);
        } else if (kDeferLast) {
            fstore<SVT>((TS *)args.value_target + flat_k[K - 1], vt_keep[0]);
        }
    };
    if (wv != 0) {
        store_deferred();
    } else {
        double bs = 0.0, bss = 0.0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { bs += red[2 * k]; bss += red[2 * k + 1]; }
        if (kStamp) st1 = wall_clock64();
        if constexpr ((VAR & 16) != 0) __builtin_amdgcn_s_setprio(3);
        const int R = geo.replicas;
        const size_t rep_off = (size_t)(blockIdx.x % R) * gridDim.x * geo.slot_stride;
        if (lane < R) {
            unsigned long long *sl = ws->slots + ((size_t)lane * gridDim.x + blockIdx.x) * geo.slot_stride;
            if (kWide) store16_sc1(sl, tag_bits(bs, tag), tag_bits(bss, tag));
            else {
                __hip_atomic_store(sl, tag_bits(bs, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sl + 1, tag_bits(bss, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        constexpr int kPer = kMaxBlocks / kWave;
        unsigned long long vs[kPer], vq[kPer];
        bool got[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) got[j] = lane + kWave * j >= (int)gridDim.x;
        const unsigned long long t_start = wall_clock64();
        bool ok;
        for (;;) {
            ok = true;
            if (kWide) {
#pragma unroll
                for (int j = 0; j < kPer; ++j) {
                    if (!got[j]) {
                        const unsigned long long *sl = ws->slots + rep_off + (size_t)(lane + kWave * j) * geo.slot_stride;
                        u64x2 v;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(sl) : "memory");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // harness only: serialises the loads of one round
                        vs[j] = v.x; vq[j] = v.y;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < kPer; ++j) {
                    if (!got[j]) {
                        const unsigned long long *sl = ws->slots + rep_off + (size_t)(lane + kWave * j) * geo.slot_stride;
                        vs[j] = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vq[j] = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                if (!got[j]) got[j] = ((unsigned)(vs[j] & 0xffull) == tag) && ((unsigned)(vq[j] & 0xffull) == tag);
                ok = ok && got[j];
            }
            if (__all(ok)) break;
            if (wall_clock64() - t_start > 100000000ull) break;
            if constexpr ((VAR & 8) == 0) __builtin_amdgcn_s_sleep((VAR & 32) ? 4 : 1);
        }
        if constexpr ((VAR & 16) != 0) __builtin_amdgcn_s_setprio(0);
        const bool all_ok = __all(ok);
        if (kStamp) st2 = wall_clock64();
        double acc = 0.0, acq = 0.0;
#pragma unroll
        for (int j = 0; j < kPer; ++j)
            if (lane + kWave * j < (int)gridDim.x) { acc += untag(vs[j]); acq += untag(vq[j]); }
        acc = wave_sum(acc);
        acq = wave_sum(acq);
        if (lane == 0) {
            const double nan = __longlong_as_double(0x7ff8000000000000LL);
            bc[0] = geo.n_total;
            bc[1] = all_ok ? acc : nan;
            bc[2] = all_ok ? acq : nan;
            if (!all_ok) ws->error = 1u;
            if (blockIdx.x == 0) ws->epoch = ws->epoch + 1u;
        }
    }
    __syncthreads();
    TS mean_s, std_s;
    mean_std<TS>(bc[0], bc[1], bc[2], geo.eps, mean_s, std_s);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        TS out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (adv_keep[k][e] - mean_s) / std_s;
        fstore<SADV>((TS *)args.advantage + flat_k[k], out);
    }
    if (wv == 0) store_deferred();
    if (kStamp && threadIdx.x == 0) {
        unsigned long long *q = dbg + 4 * blockIdx.x;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        q[0] = st0; q[1] = st1; q[2] = st2; q[3] = wall_clock64();
    }
}

static float *o_ref_adv, *o_ref_vt;

template <int K, int VAR, int SVT = 0, int SADV = 0>
double run(const char *name, GaeArgs a, Geom geo, Ws *ws, int G, int iters, bool check) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto f = [&] { hipLaunchKernelGGL((fused2<K, VAR, SVT, SADV>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, (unsigned long long *)nullptr); };
    CK(hipMemset(a.advantage, 0xff, N * 4)); CK(hipMemset(a.value_target, 0xff, N * 4));
    for (int i = 0; i < 50; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    const char *verdict = "";
    if (check) {
        std::vector<float> x(N), y(N);
        CK(hipMemcpy(x.data(), a.advantage, N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), o_ref_adv, N * 4, hipMemcpyDeviceToHost));
        bool same = memcmp(x.data(), y.data(), N * 4) == 0;
        CK(hipMemcpy(x.data(), a.value_target, N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), o_ref_vt, N * 4, hipMemcpyDeviceToHost));
        same = same && memcmp(x.data(), y.data(), N * 4) == 0;
        verdict = same ? "[outputs == variant 0]" : "[OUTPUTS DIFFER]";
    }
    printf("%-44s G=%d K=%d %8.2f us/launch  %8.1f GB/s %s\n", name, G, K, us, N * 22.0 / us * 1e-3, verdict);
    return us;
}

template <int K, int VAR, int SVT = 0, int SADV = 0>
void timeline(const char *name, GaeArgs a, Geom geo, Ws *ws, int G) {
    const int NL = 30;
    unsigned long long *dbg;
    CK(hipMalloc(&dbg, (size_t)NL * G * 32));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((fused2<K, VAR, SVT, SADV>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, (unsigned long long *)nullptr);
    for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((fused2<K, VAR | 64, SVT, SADV>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, dbg + (size_t)l * G * 4);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)NL * G * 4);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long prev_start = 0, prev_end = 0;
    int cnt = 0;
    for (int l = 0; l < NL; ++l) {
        unsigned long long s0 = ~0ull, s1 = 0, p0 = ~0ull, p1 = 0, g1 = 0, e1 = 0;
        for (int b = 0; b < G; ++b) {
            const unsigned long long *q = &h[((size_t)l * G + b) * 4];
            s0 = std::min(s0, q[0]); s1 = std::max(s1, q[0]); p0 = std::min(p0, q[1]); p1 = std::max(p1, q[1]); g1 = std::max(g1, q[2]); e1 = std::max(e1, q[3]);
        }
        if (l >= 5) {
            acc[0] += (s0 - prev_start) * 0.01; acc[1] += ((double)s0 - (double)prev_end) * 0.01; acc[2] += (p0 - s0) * 0.01;
            acc[3] += (p1 - s0) * 0.01; acc[4] += (g1 - s0) * 0.01; acc[5] += (e1 - s0) * 0.01;
            ++cnt;
        }
        prev_start = s0; prev_end = e1;
    }
    printf("%-44s timeline: period %5.2f | idle before %5.2f | first publish %5.2f  last publish %5.2f  gathered %5.2f  end %5.2f\n", name,
           acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt);
    CK(hipFree(dbg));
}

int main() {
    float *r, *v, *nv, *o1, *o2;
    unsigned char *d, *t;
    Ws *ws;
    CK(hipMalloc(&r, N * 4)); CK(hipMalloc(&v, N * 4)); CK(hipMalloc(&nv, N * 4));
    CK(hipMalloc(&o1, N * 4)); CK(hipMalloc(&o2, N * 4)); CK(hipMalloc(&o_ref_adv, N * 4)); CK(hipMalloc(&o_ref_vt, N * 4));
    CK(hipMalloc(&d, N)); CK(hipMalloc(&t, N)); CK(hipMalloc(&ws, 8 << 20));
    CK(hipMemset(ws, 0, 8 << 20));
    std::vector<float> h(N);
    for (int q = 0; q < 3; ++q) {
        for (i64 i = 0; i < N; ++i) h[i] = (float)rand() / (float)RAND_MAX - 0.5f;
        CK(hipMemcpy(q == 0 ? r : q == 1 ? v : nv, h.data(), N * 4, hipMemcpyHostToDevice));
    }
    std::vector<unsigned char> hm(N);
    for (i64 i = 0; i < N; ++i) hm[i] = (rand() % 100) == 0;
    CK(hipMemcpy(d, hm.data(), N, hipMemcpyHostToDevice)); CK(hipMemcpy(t, hm.data(), N, hipMemcpyHostToDevice));
    GaeArgs a{r, v, nv, d, t, nullptr, nullptr, 0.99f, 0.95f, 0.99f * 0.95f, 65536, 32, 1, o1, o2, nullptr, 0};
    const int G = 256;
    Geom geo{65536, 32, 32, 8, 65536 / 32, (double)N, 1e-4, 16, 4};
    // reference outputs: variant 0
    {
        GaeArgs ar = a;
        ar.advantage = o_ref_adv; ar.value_target = o_ref_vt;
        hipLaunchKernelGGL((fused2<2, 0>), dim3(G), dim3(kThreads), 0, 0, ar, geo, ws, (unsigned long long *)nullptr);
        CK(hipDeviceSynchronize());
    }
    for (int rep = 0; rep < 3; ++rep) {
        run<2, 0, 2, 2>("sc1/sc1", a, geo, ws, G, 1000, true);
        run<2, 8, 2, 2>("sc1/sc1 no sleep", a, geo, ws, G, 1000, true);
        run<2, 32, 2, 2>("sc1/sc1 sleep 4", a, geo, ws, G, 1000, true);
        run<2, 16, 2, 2>("sc1/sc1 setprio 3", a, geo, ws, G, 1000, true);
        run<2, 24, 2, 2>("sc1/sc1 setprio 3 no sleep", a, geo, ws, G, 1000, true);
    }
    for (int rp : {1, 2, 4, 8}) {
        geo.replicas = rp;
        char nm[64];
        snprintf(nm, sizeof nm, "sc1/sc1 replicas %d", rp);
        run<2, 0, 2, 2>(nm, a, geo, ws, G, 1000, true);
    }
    geo.replicas = 4;
    for (int it : {20, 50, 200, 1000, 5000}) {
        char nm[64];
        snprintf(nm, sizeof nm, "sc1/sc1 %d launches", it);
        run<2, 0, 2, 2>(nm, a, geo, ws, G, it, false);
    }
    return 0;
}
