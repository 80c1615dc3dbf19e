// Round-3 ablation harness for the fused GAE + normalise kernel (development tool; not part of the product).
// C4 [65536 x 32] fp32.  Every variant's outputs are checked against variant 0 (the product's scheme).
//   GATHER 0: product scheme -- two 8-byte tagged granules per block, one 128-byte line per block, R replicas, wave 0 of
//             every block polls all blocks' lines (only the missing ones again)
//   GATHER 1: one 16-byte granule per block (one sc1 dwordx4 store), all polls of a round issued together (one asm block)
//   GATHER 2: two-level: NG groups (group = block % NG); the group's leader (block < NG) polls its members, publishes the
//             group total; every block polls the NG group lines.  8-byte granules.
//   GATHER 3: two-level with 16-byte granules
//   GATHER 4: no gather at all (mean 0 / std 1): the floor without the exchange (outputs differ)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off kfused3.hip -o kfused3
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
    int ngroups;      // two-level gathers
    size_t l2_off;    // 8-byte units: start of the group lines
};

__device__ __forceinline__ unsigned long long tag_bits(double x, unsigned tag) {
    return ((unsigned long long)__double_as_longlong(x) & ~0xffull) | (unsigned long long)tag;
}
__device__ __forceinline__ double untag(unsigned long long b) { return __longlong_as_double((long long)(b & ~0xffull)); }

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_sc1(unsigned long long *p, unsigned long long a, unsigned long long b) {
    u64x2 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// four 16-byte sc1 loads issued together, one wait
__device__ __forceinline__ void load16x4_sc1(const unsigned long long *p0, const unsigned long long *p1, const unsigned long long *p2,
                                             const unsigned long long *p3, u64x2 &v0, u64x2 &v1, u64x2 &v2, u64x2 &v3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}
__device__ __forceinline__ void load16_sc1(const unsigned long long *p0, u64x2 &v0) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0) : "v"(p0) : "memory");
}

// store flavours: 0 nt, 1 plain, 2 sc1 (write-through; product), 3 sc0 sc1, 4 nt sc1
template <int FL>
__device__ __forceinline__ void fstore(float *p, const float (&in)[4]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {in[0], in[1], in[2], in[3]};
    if constexpr (FL == 0) __builtin_nontemporal_store(v, (f4 *)p);
    else if constexpr (FL == 1) *(f4 *)p = v;
    else if constexpr (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// DEFER: 0 value_target stored per chunk (product); 1 the LAST chunk's value_target stored after the gather; 2 all after
template <int K, int GATHER, int DEFER, bool STAMP, bool NOSCAN = false, bool NOSTORE = false>
__global__ __launch_bounds__(kThreads) void fused3(GaeArgs args, Geom geo, Ws *ws, unsigned long long *dbg) {
    typedef float TC;
    typedef float TS;
    constexpr int V = 4;
    constexpr int NW = kThreads / kWave;
    constexpr int SVT = 2, SADV = 2;
    __shared__ TeamShared<Op> sh[kTeams];
    __shared__ double red[2 * NW];
    __shared__ float bc_ms[2];
    const int team = threadIdx.x / kTeam, tid = threadIdx.x & (kTeam - 1);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int lane_off = tid * V;
    const unsigned T = (unsigned)geo.T;
    const unsigned tag = (ws->epoch + 1u) & 0xffu;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (STAMP) st0 = wall_clock64();

    TS adv_keep[K][V];
    TS vt_keep[(DEFER == 2 ? K : 1)][V];
    i64 flat_k[K];
    auto geom_of = [&](int k, i64 &start) {
        const i64 c = ((i64)k * gridDim.x + blockIdx.x) * kTeams + team;
        start = c * geo.rows_per_chunk * geo.T;
    };
    double s = 0.0, ss = 0.0;
    typename Op::Raw raw[2];
    if constexpr (!NOSCAN) {
        i64 start;
        geom_of(0, start);
        load_inputs<Op, true>(args, start + lane_off, V, raw[0]);
    }
    if constexpr (NOSCAN) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            i64 start;
            geom_of(k, start);
            flat_k[k] = start + lane_off;
#pragma unroll
            for (int e = 0; e < V; ++e) adv_keep[k][e] = (float)(threadIdx.x + e);
        }
        s = 1.0; ss = 2.0;
    }
#pragma unroll
    for (int k = 0; k < (NOSCAN ? 0 : K); ++k) {
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
                                 if (DEFER == 2 || (DEFER == 1 && k == K - 1)) {
                                     op.finish(V, xs, adv_keep[k], vt_keep[DEFER == 2 ? k : 0], s, ss);
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
    if (wv == 0) {
        double bs = 0.0, bss = 0.0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { bs += red[2 * k]; bss += red[2 * k + 1]; }
        if (STAMP) st1 = wall_clock64();
        __builtin_amdgcn_s_setprio(3);
        const int G = (int)gridDim.x;
        const unsigned long long t_start = wall_clock64();
        bool all_ok = true;
        double acc = 0.0, acq = 0.0;
        if constexpr (GATHER == 0 || GATHER == 1) {
            const int R = geo.replicas;
            const size_t rep_off = (size_t)(blockIdx.x % R) * G * geo.slot_stride;
            if (lane < R) {
                unsigned long long *sl = ws->slots + ((size_t)lane * G + blockIdx.x) * geo.slot_stride;
                if (GATHER == 1) store16_sc1(sl, tag_bits(bs, tag), tag_bits(bss, tag));
                else {
                    __hip_atomic_store(sl, tag_bits(bs, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sl + 1, tag_bits(bss, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            constexpr int kPer = kMaxBlocks / kWave;  // 8; G = 256 uses 4
            unsigned long long vs[kPer], vq[kPer];
            bool got[kPer];
#pragma unroll
            for (int j = 0; j < kPer; ++j) got[j] = lane + kWave * j >= G;
            bool ok;
            for (;;) {
                ok = true;
                if constexpr (GATHER == 1) {
                    // G <= 256 in this harness: four lines per lane
                    const unsigned long long *base = ws->slots + rep_off;
                    u64x2 v0, v1, v2, v3;
                    load16x4_sc1(base + (size_t)(lane) * geo.slot_stride, base + (size_t)(lane + 64) * geo.slot_stride,
                                 base + (size_t)(lane + 128) * geo.slot_stride, base + (size_t)(lane + 192) * geo.slot_stride, v0, v1, v2, v3);
                    vs[0] = v0.x; vq[0] = v0.y; vs[1] = v1.x; vq[1] = v1.y; vs[2] = v2.x; vq[2] = v2.y; vs[3] = v3.x; vq[3] = v3.y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        got[j] = got[j] || (((unsigned)(vs[j] & 0xffull) == tag) && ((unsigned)(vq[j] & 0xffull) == tag));
                        ok = ok && got[j];
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
#pragma unroll
                    for (int j = 0; j < kPer; ++j) {
                        if (!got[j]) got[j] = ((unsigned)(vs[j] & 0xffull) == tag) && ((unsigned)(vq[j] & 0xffull) == tag);
                        ok = ok && got[j];
                    }
                }
                if (__all(ok)) break;
                if (wall_clock64() - t_start > 100000000ull) break;
            }
            all_ok = __all(ok);
#pragma unroll
            for (int j = 0; j < kPer; ++j)
                if (lane + kWave * j < G) { acc += untag(vs[j]); acq += untag(vq[j]); }
            acc = wave_sum(acc);
            acq = wave_sum(acq);
        } else if constexpr (GATHER == 5) {
            const int R = geo.replicas;
            if (lane < R) store16_sc1(ws->slots + ((size_t)lane * G + blockIdx.x) * 2, tag_bits(bs, tag), tag_bits(bss, tag));
            const unsigned long long *base = ws->slots + (size_t)(blockIdx.x % R) * G * 2;
            u64x2 v0, v1, v2, v3;
            bool ok;
            for (;;) {
                load16x4_sc1(base + (size_t)lane * 2, base + (size_t)(lane + 64) * 2, base + (size_t)(lane + 128) * 2,
                             base + (size_t)(lane + 192) * 2, v0, v1, v2, v3);
                ok = ((unsigned)(v0.x & 0xff) == tag) && ((unsigned)(v0.y & 0xff) == tag) && ((unsigned)(v1.x & 0xff) == tag) &&
                     ((unsigned)(v1.y & 0xff) == tag) && ((unsigned)(v2.x & 0xff) == tag) && ((unsigned)(v2.y & 0xff) == tag) &&
                     ((unsigned)(v3.x & 0xff) == tag) && ((unsigned)(v3.y & 0xff) == tag);
                if (__all(ok)) break;
                if (wall_clock64() - t_start > 100000000ull) break;
            }
            all_ok = __all(ok);
            acc = ((untag(v0.x) + untag(v1.x)) + untag(v2.x)) + untag(v3.x);
            acq = ((untag(v0.y) + untag(v1.y)) + untag(v2.y)) + untag(v3.y);
            acc = wave_sum(acc);
            acq = wave_sum(acq);
        } else if constexpr (GATHER == 2 || GATHER == 3) {
            const int NG = geo.ngroups;           // groups; group of block b = b % NG; members G / NG (<= 64)
            const int g = (int)blockIdx.x % NG;
            const int members = G / NG;
            constexpr bool W = GATHER == 3;
            // level 1: own line (one copy: only the group's leader reads it)
            if (lane == 0) {
                unsigned long long *sl = ws->slots + (size_t)blockIdx.x * geo.slot_stride;
                if (W) store16_sc1(sl, tag_bits(bs, tag), tag_bits(bss, tag));
                else {
                    __hip_atomic_store(sl, tag_bits(bs, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sl + 1, tag_bits(bss, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            unsigned long long *l2 = ws->slots + geo.l2_off;
            const int R = geo.replicas;
            if ((int)blockIdx.x < NG) {  // leader of group g: gather the members' lines (lane m reads block g + m * NG)
                unsigned long long a = 0, q = 0;
                bool got = lane >= members;
                for (;;) {
                    if (!got) {
                        const unsigned long long *sl = ws->slots + (size_t)(g + lane * NG) * geo.slot_stride;
                        if (W) { u64x2 v; load16_sc1(sl, v); a = v.x; q = v.y; }
                        else {
                            a = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            q = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        got = ((unsigned)(a & 0xffull) == tag) && ((unsigned)(q & 0xffull) == tag);
                    }
                    if (__all(got)) break;
                    if (wall_clock64() - t_start > 100000000ull) break;
                }
                all_ok = __all(got);
                double ga = lane < members ? untag(a) : 0.0, gq = lane < members ? untag(q) : 0.0;
                ga = wave_sum(ga);
                gq = wave_sum(gq);
                if (lane < R) {
                    unsigned long long *sl = l2 + ((size_t)lane * NG + g) * geo.slot_stride;
                    if (W) store16_sc1(sl, tag_bits(ga, tag), tag_bits(gq, tag));
                    else {
                        __hip_atomic_store(sl, tag_bits(ga, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(sl + 1, tag_bits(gq, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            // level 2: every block polls the NG group lines of its replica
            {
                const size_t rep_off = (size_t)(blockIdx.x % R) * NG * geo.slot_stride;
                unsigned long long a = 0, q = 0;
                bool got = lane >= NG;
                for (;;) {
                    if (!got) {
                        const unsigned long long *sl = l2 + rep_off + (size_t)lane * geo.slot_stride;
                        if (W) { u64x2 v; load16_sc1(sl, v); a = v.x; q = v.y; }
                        else {
                            a = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            q = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        got = ((unsigned)(a & 0xffull) == tag) && ((unsigned)(q & 0xffull) == tag);
                    }
                    if (__all(got)) break;
                    if (wall_clock64() - t_start > 100000000ull) break;
                }
                all_ok = all_ok && __all(got);
                acc = wave_sum(lane < NG ? untag(a) : 0.0);
                acq = wave_sum(lane < NG ? untag(q) : 0.0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (STAMP) st2 = wall_clock64();
        if (lane == 0) {
            float mean_s = 0.f, std_s = 1.f;
            if (GATHER != 4 && all_ok) mean_std<float>(geo.n_total, acc, acq, geo.eps, mean_s, std_s);
            bc_ms[0] = mean_s;
            bc_ms[1] = std_s;
            if (!all_ok) ws->error = 1u;
            if (blockIdx.x == 0) ws->epoch = ws->epoch + 1u;
        }
    }
    __syncthreads();
    const TS mean_s = bc_ms[0], std_s = bc_ms[1];
    if constexpr (!NOSTORE) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        TS out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (adv_keep[k][e] - mean_s) / std_s;
        fstore<SADV>((TS *)args.advantage + flat_k[k], out);
    }
    } else if (mean_s == 123.f) ((TS *)args.advantage)[0] = adv_keep[0][0];
    if (DEFER == 2) {
#pragma unroll
        for (int k = 0; k < K; ++k) fstore<SVT>((TS *)args.value_target + flat_k[k], vt_keep[k]);
    } else if (DEFER == 1) {
        fstore<SVT>((TS *)args.value_target + flat_k[K - 1], vt_keep[0]);
    }
    if (STAMP && threadIdx.x == 0) {
        unsigned long long *q = dbg + 4 * blockIdx.x;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        q[0] = st0; q[1] = st1; q[2] = st2; q[3] = wall_clock64();
    }
}

static float *o_ref_adv, *o_ref_vt;

template <int K, int GATHER, int DEFER>
double run(const char *name, GaeArgs a, Geom geo, Ws *ws, int G, int iters, bool check) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto f = [&] { hipLaunchKernelGGL((fused3<K, GATHER, DEFER, false>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, (unsigned long long *)nullptr); };
    CK(hipMemset(a.advantage, 0xff, N * 4)); CK(hipMemset(a.value_target, 0xff, N * 4));
    CK(hipMemset(ws, 0, 64 << 20));  // epoch 0: no slot another variant left behind can carry a matching tag
    for (int i = 0; i < 50; ++i) f();
    CK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / iters;
        best = std::min(best, us);
        sum += us;
    }
    const char *verdict = "";
    if (check) {
        std::vector<float> x(N), y(N);
        CK(hipMemcpy(x.data(), a.advantage, N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), o_ref_adv, N * 4, hipMemcpyDeviceToHost));
        bool same = memcmp(x.data(), y.data(), N * 4) == 0;
        CK(hipMemcpy(x.data(), a.value_target, N * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y.data(), o_ref_vt, N * 4, hipMemcpyDeviceToHost));
        same = same && memcmp(x.data(), y.data(), N * 4) == 0;
        unsigned err = 0;
        CK(hipMemcpy(&err, &ws->error, 4, hipMemcpyDeviceToHost));
        verdict = err ? "[GAVE UP]" : same ? "[outputs == variant 0]" : "[OUTPUTS DIFFER]";
    }
    printf("%-50s G=%d K=%d  best %6.2f  mean %6.2f us/launch  %7.1f GB/s %s\n", name, G, K, best, sum / reps, N * 22.0 / (sum / reps) * 1e-3, verdict);
    fflush(stdout);
    return sum / reps;
}

template <int K, int GATHER, int DEFER, bool NOSCAN = false, bool NOSTORE = false>
void timeline(const char *name, GaeArgs a, Geom geo, Ws *ws, int G) {
    const int NL = 60;
    unsigned long long *dbg;
    CK(hipMalloc(&dbg, (size_t)NL * G * 32));
    CK(hipMemset(ws, 0, 64 << 20));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((fused3<K, GATHER, DEFER, false, NOSCAN, NOSTORE>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, (unsigned long long *)nullptr);
    for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((fused3<K, GATHER, DEFER, true, NOSCAN, NOSTORE>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, dbg + (size_t)l * G * 4);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)NL * G * 4);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prev_start = 0, prev_end = 0;
    int cnt = 0;
    for (int l = 0; l < NL; ++l) {
        unsigned long long s0 = ~0ull, s1 = 0, p0 = ~0ull, p1 = 0, g0 = ~0ull, g1 = 0, e1 = 0;
        for (int b = 0; b < G; ++b) {
            const unsigned long long *q = &h[((size_t)l * G + b) * 4];
            s0 = std::min(s0, q[0]); s1 = std::max(s1, q[0]); p0 = std::min(p0, q[1]); p1 = std::max(p1, q[1]);
            g0 = std::min(g0, q[2]); g1 = std::max(g1, q[2]); e1 = std::max(e1, q[3]);
        }
        if (l >= 10) {
            acc[0] += (s0 - prev_start) * 0.01; acc[1] += ((double)s0 - (double)prev_end) * 0.01; acc[2] += (p0 - s0) * 0.01;
            acc[3] += (p1 - s0) * 0.01; acc[4] += (g1 - s0) * 0.01; acc[5] += (e1 - s0) * 0.01; acc[6] += (s1 - s0) * 0.01; acc[7] += (g0 - s0) * 0.01;
            ++cnt;
        }
        prev_start = s0; prev_end = e1;
    }
    printf("%-50s timeline(us): period %5.2f | gap before first start %5.2f | last start %5.2f | first publish %5.2f  last publish %5.2f | first gathered %5.2f  last gathered %5.2f | end %5.2f\n",
           name, acc[0] / cnt, acc[1] / cnt, acc[6] / cnt, acc[2] / cnt, acc[3] / cnt, acc[7] / cnt, acc[4] / cnt, acc[5] / cnt);
    fflush(stdout);
    CK(hipFree(dbg));
}

int main() {
    float *r, *v, *nv, *o1, *o2;
    unsigned char *d, *t;
    Ws *ws;
    CK(hipMalloc(&r, N * 4)); CK(hipMalloc(&v, N * 4)); CK(hipMalloc(&nv, N * 4));
    CK(hipMalloc(&o1, N * 4)); CK(hipMalloc(&o2, N * 4)); CK(hipMalloc(&o_ref_adv, N * 4)); CK(hipMalloc(&o_ref_vt, N * 4));
    CK(hipMalloc(&d, N)); CK(hipMalloc(&t, N)); CK(hipMalloc(&ws, 64 << 20));
    CK(hipMemset(ws, 0, 64 << 20));
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
    // level-2 lines sit behind 4 replicas x 512 block lines
    Geom geo{65536, 32, 32, 8, 65536 / 32, (double)N, 1e-4, 16, 4, 16, (size_t)4 * 512 * 16};
    {
        GaeArgs ar = a;
        ar.advantage = o_ref_adv; ar.value_target = o_ref_vt;
        hipLaunchKernelGGL((fused3<2, 0, 0, false>), dim3(G), dim3(kThreads), 0, 0, ar, geo, ws, (unsigned long long *)nullptr);
        CK(hipDeviceSynchronize());
    }
    // clock settle
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL((fused3<2, 0, 0, false>), dim3(G), dim3(kThreads), 0, 0, a, geo, ws, (unsigned long long *)nullptr);
    CK(hipDeviceSynchronize());
    const int mode = getenv("KF_MODE") ? atoi(getenv("KF_MODE")) : 0;
    if (mode == 2) {
        geo.slot_stride = 16; geo.ngroups = 16; geo.l2_off = (size_t)4 * 512 * 16;
        for (int rep2 = 0; rep2 < 2; ++rep2) {
            geo.replicas = 4;
            run<2, 0, 0>("product", a, geo, ws, G, 1000, true);
            for (int rp : {1, 4, 8, 16, 32}) {
                geo.replicas = rp;
                char nm[64];
                snprintf(nm, sizeof nm, "packed 16B granules, %d replicas", rp);
                run<2, 5, 0>(nm, a, geo, ws, G, 1000, true);
            }
            geo.replicas = 16;
            run<2, 5, 1>("packed 16 replicas, last vt after gather", a, geo, ws, G, 1000, true);
        }
        geo.replicas = 16;
        timeline<2, 5, 0>("packed 16 replicas", a, geo, ws, G);
        timeline<2, 5, 0, true, true>("gather only: packed 16 replicas", a, geo, ws, G);
        geo.replicas = 4;
        timeline<2, 0, 0>("product", a, geo, ws, G);
    }
    if (mode == 1) {
        // the exchange alone: no loads, no scan (NOSCAN), with / without the final stores
        geo.slot_stride = 16; geo.replicas = 4; geo.ngroups = 16; geo.l2_off = (size_t)4 * 512 * 16;
        timeline<2, 0, 0, true, true>("gather only: product", a, geo, ws, G);
        timeline<2, 1, 0, true, true>("gather only: 16B granule", a, geo, ws, G);
        timeline<2, 3, 0, true, true>("gather only: two-level 16x16 16B", a, geo, ws, G);
        timeline<2, 4, 0, true, true>("gather only: none", a, geo, ws, G);
        timeline<2, 0, 0, true, false>("gather + adv stores: product", a, geo, ws, G);
        timeline<2, 0, 0, false, true>("scan + gather, no adv stores: product", a, geo, ws, G);
        timeline<2, 0, 2, false, true>("scan (no vt stores) + gather, no adv stores: product", a, geo, ws, G);
        geo.replicas = 1;
        timeline<2, 0, 0, true, true>("gather only: product, 1 replica", a, geo, ws, G);
        timeline<2, 1, 0, true, true>("gather only: 16B granule, 1 replica", a, geo, ws, G);
    }
    if (mode == 0) {
        // stride sweep: consecutive blocks' lines `stride` 8-byte units apart (16 = one 128-byte line)
        for (int stride : {16, 32, 64, 128, 256, 512, 528, 1040, 2064}) {
            for (int rp : {1, 4}) {
                geo.slot_stride = stride;
                geo.replicas = rp;
                geo.ngroups = 16;
                geo.l2_off = (size_t)4 * 512 * stride;
                char nm[96];
                snprintf(nm, sizeof nm, "stride %5d B rep %d: product", stride * 8, rp);
                run<2, 0, 0>(nm, a, geo, ws, G, 1000, true);
                snprintf(nm, sizeof nm, "stride %5d B rep %d: 16B granule", stride * 8, rp);
                run<2, 1, 0>(nm, a, geo, ws, G, 1000, true);
                snprintf(nm, sizeof nm, "stride %5d B rep %d: two-level 16x16 16B", stride * 8, rp);
                run<2, 3, 0>(nm, a, geo, ws, G, 1000, true);
            }
        }
        geo.slot_stride = 528; geo.replicas = 1; geo.l2_off = (size_t)4 * 512 * 528;
        timeline<2, 0, 0>("product stride 4224", a, geo, ws, G);
        timeline<2, 1, 0>("16B granule stride 4224", a, geo, ws, G);
        timeline<2, 3, 0>("two-level 16x16 16B stride 4224", a, geo, ws, G);
        geo.slot_stride = 16; geo.replicas = 4; geo.l2_off = (size_t)4 * 512 * 16;
        timeline<2, 0, 0>("product stride 128", a, geo, ws, G);
        run<2, 4, 0>("no gather (floor; outputs differ)", a, geo, ws, G, 1000, false);
    }
    return 0;
}
