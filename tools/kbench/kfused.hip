// Ablation harness for the fused GAE+normalise kernel (development tool; not part of the product).
#include "../../torchrl_amd/csrc/fused_kernels.hip"
#include "../../torchrl_amd/csrc/norm_kernels.hip"
xchg::Dev xchg_dev_of(void *) { return xchg::Dev{nullptr, nullptr, 0, 1, 0}; }  // the harness has no communicator
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
constexpr i64 N = 65536LL * 32;
typedef GaeOp<float, float, 4> Op;

template <int ABL, int K, int TEAMS = 4>
void run(const char* name, GaeArgs a, FusedGeom geo, FusedWs* ws, int G, int iters, double* sink = nullptr) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto f = [&] { hipLaunchKernelGGL((gae_fused_kernel<Op, true, true, K, ABL, TEAMS>), dim3(G), dim3(TEAMS * 256), 0, 0, a, geo, ws, sink); };
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-40s G=%d K=%d teams=%d %8.2f us/launch  %8.1f GB/s\n", name, G, K, TEAMS, us, N * 22.0 / us * 1e-3);
}

int main() {
    float *r, *v, *nv, *o1, *o2; unsigned char *d, *t; FusedWs* ws;
    CK(hipMalloc(&r, N * 4)); CK(hipMalloc(&v, N * 4)); CK(hipMalloc(&nv, N * 4));
    CK(hipMalloc(&o1, N * 4)); CK(hipMalloc(&o2, N * 4)); CK(hipMalloc(&d, N)); CK(hipMalloc(&t, N)); CK(hipMalloc(&ws, 64 << 20));  // 4 slot replicas x 256 blocks x the widest stride tried below (16.5 KB) = 16.9 MB
    CK(hipMemset(ws, 0, 64 << 20));
    std::vector<float> h(N); for (i64 i = 0; i < N; ++i) h[i] = (float)rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(r, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(nv, h.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<unsigned char> hm(N); for (i64 i = 0; i < N; ++i) hm[i] = (rand() % 100) == 0;
    CK(hipMemcpy(d, hm.data(), N, hipMemcpyHostToDevice)); CK(hipMemcpy(t, hm.data(), N, hipMemcpyHostToDevice));
    GaeArgs a{r, v, nv, d, t, nullptr, nullptr, 0.99f, 0.95f, 0.99f * 0.95f, 65536, 32, 1, o1, o2, nullptr};
    const int G = 256;
    FusedGeom geo{65536, 32, 32, 8, 65536 / 32, (double)N, 1e-4, 2};
    geo.slot_stride = 16;
    for (int rep = 0; rep < 3; ++rep) {
        run<0, 2>("full (product)", a, geo, ws, G, 500);
        run<4, 2>("last vt deferred", a, geo, ws, G, 500);
        run<1, 2>("no gather", a, geo, ws, G, 500);
        run<10, 2>("both chunks' loads up front", a, geo, ws, G, 500);
    }
    for (int stride : {16, 32, 64, 128, 512, 528, 1024, 2064}) {   // slot line spacing: 128 B .. 16 KB (memory-channel spreading)
        geo.slot_stride = stride;
        char nm[64]; snprintf(nm, sizeof nm, "slot stride %d B", stride * 8);
        run<0, 2>(nm, a, geo, ws, G, 1000);
    }
    geo.slot_stride = 16;
    {   // 40 stamped launches back to back: per-launch timeline on the device clock (100 MHz)
        geo.slot_stride = 16;
        const int NL = 40;
        unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)NL * G * 32));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((gae_fused_kernel<Op, true, true, 2, 0>), dim3(G), dim3(kFusedThreads), 0, 0, a, geo, ws, (double*)nullptr);
        for (int l = 0; l < NL; ++l)
            hipLaunchKernelGGL((gae_fused_kernel<Op, true, true, 2, 7>), dim3(G), dim3(kFusedThreads), 0, 0, a, geo, ws, (double*)(dbg + (size_t)l * G * 4));
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)NL * G * 4);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long prev_end = 0, prev_start = 0;
        for (int l = 0; l < NL; ++l) {
            unsigned long long s0 = ~0ull, s1 = 0, p0 = ~0ull, p1 = 0, g1 = 0, e1 = 0;
            for (int b = 0; b < G; ++b) {
                const unsigned long long* q = &h[((size_t)l * G + b) * 4];
                s0 = std::min(s0, q[0]); s1 = std::max(s1, q[0]); p0 = std::min(p0, q[1]); p1 = std::max(p1, q[1]); g1 = std::max(g1, q[2]); e1 = std::max(e1, q[3]);
            }
            printf("launch %2d: period %6.2f us | idle before %5.2f | start skew %4.2f  first publish %5.2f  last publish %5.2f  gathered %5.2f  end %5.2f\n", l,
                   prev_start ? (s0 - prev_start) * 0.01 : 0.0, prev_end ? ((double)s0 - (double)prev_end) * 0.01 : 0.0, (s1 - s0) * 0.01, (p0 - s0) * 0.01, (p1 - s0) * 0.01, (g1 - s0) * 0.01, (e1 - s0) * 0.01);
            prev_end = e1; prev_start = s0;
        }
    }
    for (int stride : {16}) {
      geo.slot_stride = stride;
      printf("slot stride %d bytes\n", stride * 8);
      run<0, 2>("  full", a, geo, ws, G, 300);
      // per-block stamps (100 MHz wall clock): start, before publish, gather done, end
        unsigned long long* dbg; CK(hipMalloc(&dbg, G * 32));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL((gae_fused_kernel<Op, true, true, 2, 7>), dim3(G), dim3(kFusedThreads), 0, 0, a, geo, ws, (double*)dbg);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hdbg(G * 4);
            CK(hipMemcpy(hdbg.data(), dbg, G * 32, hipMemcpyDeviceToHost));
            unsigned long long t0min = ~0ull, t0max = 0, t1min = ~0ull, t1max = 0, t2min = ~0ull, t2max = 0, t3max = 0;
            for (int b = 0; b < G; ++b) {
                t0min = std::min(t0min, hdbg[4*b]); t0max = std::max(t0max, hdbg[4*b]);
                t1min = std::min(t1min, hdbg[4*b+1]); t1max = std::max(t1max, hdbg[4*b+1]);
                t2min = std::min(t2min, hdbg[4*b+2]); t2max = std::max(t2max, hdbg[4*b+2]);
                t3max = std::max(t3max, hdbg[4*b+3]);
            }
            auto us = [&](unsigned long long t) { return (double)(t - t0min) * 0.01; };
            printf("stamps[us from first block start]: start<=%.2f  publish %.2f..%.2f  gathered %.2f..%.2f  end<=%.2f\n",
                   us(t0max), us(t1min), us(t1max), us(t2min), us(t2max), us(t3max));
        }
    }
    return 0;
}
