// Exchange-only microbenchmark (development tool): 256 blocks x 1024 threads, nothing but the in-launch all-gather of one
// tagged 16-byte granule per block.  Reports, over launches, the time from the LAST publish to the first / last block that
// has gathered everything.  Runtime knobs: pollers (only blocks < pollers poll), delay before the first poll, sleep between
// rounds, replicas, line stride, loads per poll round (all lines at once or missing only).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 kxchg.hip -o kxchg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

struct P {
    int pollers, delay, sleep, replicas, stride;  // stride in 8-byte units
    int mode;  // 0: 16-B granules, 4 lines per lane per round; 1: 8-B granule pairs through atomic loads
               // 2: dense read-side layout: every block ALSO stores its granule into a packed array (16 B apart) and pollers
               //    read the packed array with coalesced 16-byte loads (4 KB for 256 blocks)
               // 3: as 0 but each poller publishes to / polls R replicas chosen by XCC id
};

__device__ __forceinline__ void store16_sc1(u64 *p, u64 a, u64 b) {
    u64x2 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load16x4_sc1(const u64 *p0, const u64 *p1, const u64 *p2, const u64 *p3, u64x2 &v0, u64x2 &v1, u64x2 &v2,
                                             u64x2 &v3) {
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

__device__ __forceinline__ void issue4(const u64 *p0, const u64 *p1, const u64 *p2, const u64 *p3, u64x2 &v0, u64x2 &v1, u64x2 &v2, u64x2 &v3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}
// wait until at most N vector-memory operations are outstanding; the four registers become readable
#define WAIT4(N, v0, v1, v2, v3) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory")

__device__ __forceinline__ bool tags_ok(const u64x2 &v0, const u64x2 &v1, const u64x2 &v2, const u64x2 &v3, unsigned tag) {
    return ((unsigned)(v0.x & 0xff) == tag) && ((unsigned)(v0.y & 0xff) == tag) && ((unsigned)(v1.x & 0xff) == tag) &&
           ((unsigned)(v1.y & 0xff) == tag) && ((unsigned)(v2.x & 0xff) == tag) && ((unsigned)(v2.y & 0xff) == tag) &&
           ((unsigned)(v3.x & 0xff) == tag) && ((unsigned)(v3.y & 0xff) == tag);
}

__global__ __launch_bounds__(1024) void xg(u64 *slots, unsigned *epoch, P p, u64 *dbg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int G = gridDim.x;
    const unsigned tag = (*epoch + 1u) & 0xffu;
    __shared__ int flag;
    u64 st0 = wall_clock64(), st1 = 0, st2 = 0;
    __syncthreads();
    if (wv == 0) {
        st1 = wall_clock64();
        __builtin_amdgcn_s_setprio(3);
        const u64 va = ((u64)blockIdx.x << 8) | tag, vb = ((u64)(blockIdx.x * 3) << 8) | tag;
        const int R = p.replicas;
        if (p.mode == 2 || p.mode == 3) {
            if (lane < R) store16_sc1(slots + ((size_t)lane * G + blockIdx.x) * 2, va, vb);
        } else if (lane < R) {
            u64 *sl = slots + ((size_t)lane * G + blockIdx.x) * p.stride;
            if (p.mode == 1) {
                __hip_atomic_store(sl, va, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sl + 1, vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else store16_sc1(sl, va, vb);
        }
        bool ok = true;
        if ((int)blockIdx.x < p.pollers) {
            for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(4);
            const size_t rep = (size_t)(blockIdx.x % R) * G;
            const u64 t_start = wall_clock64();
            u64x2 v0, v1, v2, v3;
            if (p.mode == 3) {
                const u64 *b = slots + rep * 2;
                const u64 *q0 = b + (size_t)lane * 2, *q1 = b + (size_t)(lane + 64) * 2, *q2 = b + (size_t)(lane + 128) * 2,
                          *q3 = b + (size_t)(lane + 192) * 2;
                u64x2 a0, a1, a2, a3, b0, b1, b2, b3;
                issue4(q0, q1, q2, q3, a0, a1, a2, a3);
                for (;;) {
                    for (int i = 0; i < p.sleep; ++i) __builtin_amdgcn_s_sleep(1);
                    issue4(q0, q1, q2, q3, b0, b1, b2, b3);
                    WAIT4(4, a0, a1, a2, a3);
                    if (__all(tags_ok(a0, a1, a2, a3, tag))) { WAIT4(0, b0, b1, b2, b3); v0 = a0; v1 = a1; v2 = a2; v3 = a3; ok = true; break; }
                    for (int i = 0; i < p.sleep; ++i) __builtin_amdgcn_s_sleep(1);
                    issue4(q0, q1, q2, q3, a0, a1, a2, a3);
                    WAIT4(4, b0, b1, b2, b3);
                    if (__all(tags_ok(b0, b1, b2, b3, tag))) { WAIT4(0, a0, a1, a2, a3); v0 = b0; v1 = b1; v2 = b2; v3 = b3; ok = true; break; }
                    if (wall_clock64() - t_start > 100000000ull) { WAIT4(0, a0, a1, a2, a3); ok = false; v0 = a0; v1 = a1; v2 = a2; v3 = a3; break; }
                }
            } else
            for (;;) {
                if (p.mode == 2) {
                    const u64 *b = slots + rep * 2;
                    load16x4_sc1(b + (size_t)lane * 2, b + (size_t)(lane + 64) * 2, b + (size_t)(lane + 128) * 2, b + (size_t)(lane + 192) * 2, v0,
                                 v1, v2, v3);
                } else if (p.mode == 1) {
                    const u64 *b = slots + rep * p.stride;
                    u64 a0 = __hip_atomic_load(b + (size_t)lane * p.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 b0 = __hip_atomic_load(b + (size_t)lane * p.stride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 a1 = __hip_atomic_load(b + (size_t)(lane + 64) * p.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 b1 = __hip_atomic_load(b + (size_t)(lane + 64) * p.stride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 a2 = __hip_atomic_load(b + (size_t)(lane + 128) * p.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 b2 = __hip_atomic_load(b + (size_t)(lane + 128) * p.stride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 a3 = __hip_atomic_load(b + (size_t)(lane + 192) * p.stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    u64 b3 = __hip_atomic_load(b + (size_t)(lane + 192) * p.stride + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v0 = u64x2{a0, b0}; v1 = u64x2{a1, b1}; v2 = u64x2{a2, b2}; v3 = u64x2{a3, b3};
                } else {
                    const u64 *b = slots + rep * p.stride;
                    load16x4_sc1(b + (size_t)lane * p.stride, b + (size_t)(lane + 64) * p.stride, b + (size_t)(lane + 128) * p.stride,
                                 b + (size_t)(lane + 192) * p.stride, v0, v1, v2, v3);
                }
                ok = ((unsigned)(v0.x & 0xff) == tag) && ((unsigned)(v0.y & 0xff) == tag) && ((unsigned)(v1.x & 0xff) == tag) &&
                     ((unsigned)(v1.y & 0xff) == tag) && ((unsigned)(v2.x & 0xff) == tag) && ((unsigned)(v2.y & 0xff) == tag) &&
                     ((unsigned)(v3.x & 0xff) == tag) && ((unsigned)(v3.y & 0xff) == tag);
                if (__all(ok)) break;
                if (wall_clock64() - t_start > 100000000ull) break;
                for (int i = 0; i < p.sleep; ++i) __builtin_amdgcn_s_sleep(1);
            }
            // check payloads
            ok = __all(ok) && (v0.x >> 8) == (u64)lane && (v1.x >> 8) == (u64)(lane + 64) && (v3.y >> 8) == (u64)((lane + 192) * 3);
        }
        __builtin_amdgcn_s_setprio(0);
        st2 = wall_clock64();
        if (lane == 0) {
            flag = __all(ok) ? 1 : 0;
            if (blockIdx.x == 0) *epoch = *epoch + 1u;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && dbg) {
        u64 *q = dbg + 4 * blockIdx.x;
        q[0] = st0; q[1] = st1; q[2] = st2; q[3] = (u64)flag;
    }
}

int main() {
    const int G = 256, NL = 80;
    u64 *slots, *dbg;
    unsigned *epoch;
    CK(hipMalloc(&slots, 64 << 20));
    CK(hipMalloc(&epoch, 256));
    CK(hipMalloc(&dbg, (size_t)NL * G * 32));
    auto exp = [&](const char *name, P p) {
        CK(hipMemset(slots, 0, 64 << 20));
        CK(hipMemset(epoch, 0, 256));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(xg, dim3(G), dim3(1024), 0, 0, slots, epoch, p, (u64 *)nullptr);
        for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(xg, dim3(G), dim3(1024), 0, 0, slots, epoch, p, dbg + (size_t)l * G * 4);
        CK(hipDeviceSynchronize());
        std::vector<u64> h((size_t)NL * G * 4);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double a_first = 0, a_last = 0, a_skew = 0;
        int bad = 0, cnt = 0;
        for (int l = 10; l < NL; ++l) {
            u64 p1 = 0, p0 = ~0ull, g0 = ~0ull, g1 = 0;
            for (int b = 0; b < G; ++b) {
                const u64 *q = &h[((size_t)l * G + b) * 4];
                p0 = std::min(p0, q[1]); p1 = std::max(p1, q[1]);
                if (b < p.pollers) { g0 = std::min(g0, q[2]); g1 = std::max(g1, q[2]); }
                bad += q[3] != 1;
            }
            a_first += ((double)g0 - (double)p1) * 0.01; a_last += ((double)g1 - (double)p1) * 0.01; a_skew += (p1 - p0) * 0.01;
            ++cnt;
        }
        printf("%-64s publish skew %5.2f | last publish -> first gathered %5.2f, last gathered %5.2f us %s\n", name, a_skew / cnt, a_first / cnt,
               a_last / cnt, bad ? "[BAD]" : "");
        fflush(stdout);
    };
    char nm[128];
    for (int mode : {1, 2}) {
        for (int pollers : {1, 8, 32, 256}) {
            P p{pollers, 0, 0, 4, 16, mode};
            snprintf(nm, sizeof nm, "mode %d pollers %3d delay 0 sleep 0 rep 4", mode, pollers);
            exp(nm, p);
        }
    }
    for (int rep : {4, 8, 16}) {
        for (int sleep : {0, 2, 4, 8}) {
            P q{256, 0, sleep, rep, 16, 3};
            snprintf(nm, sizeof nm, "mode 3 (packed, 2 rounds in flight) rep %d sleep %d", rep, sleep);
            exp(nm, q);
        }
        P q{256, 0, 0, rep, 16, 2};
        snprintf(nm, sizeof nm, "mode 2 (packed) rep %d", rep);
        exp(nm, q);
    }
    if (getenv("KX_ALL"))
    for (int delay : {2, 5, 10, 20}) {
        P p{256, delay, 0, 4, 16, 0};
        snprintf(nm, sizeof nm, "mode 0 pollers 256 delay %2d (x s_sleep 4) sleep 0 rep 4", delay);
        exp(nm, p);
    }
    if (getenv("KX_ALL"))
    for (int sleep : {1, 4, 16}) {
        P p{256, 0, sleep, 4, 16, 0};
        snprintf(nm, sizeof nm, "mode 0 pollers 256 delay 0 sleep %2d rep 4", sleep);
        exp(nm, p);
        P q{256, 5, sleep, 4, 16, 0};
        snprintf(nm, sizeof nm, "mode 0 pollers 256 delay 5 sleep %2d rep 4", sleep);
        exp(nm, q);
    }
    if (getenv("KX_ALL"))
    for (int rep : {1, 2, 8, 16}) {
        P p{256, 0, 0, rep, 16, 0};
        snprintf(nm, sizeof nm, "mode 0 pollers 256 rep %d", rep);
        exp(nm, p);
        P q{256, 0, 0, rep, 16, 2};
        snprintf(nm, sizeof nm, "mode 2 (packed) pollers 256 rep %d", rep);
        exp(nm, q);
    }
    return 0;
}
