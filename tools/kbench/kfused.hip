// Ablation harness for the fused GAE+normalise kernel (development tool; not part of the product).
#include "../../torchrl_amd/csrc/fused_kernels.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
constexpr i64 N = 65536LL * 32;
typedef GaeOp<float, float, 4> Op;

template <int ABL, int K>
void run(const char* name, GaeArgs a, FusedGeom geo, FusedWs* ws, int G, int iters, double* sink = nullptr) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto f = [&] { hipLaunchKernelGGL((gae_fused_kernel<Op, true, true, K, ABL>), dim3(G), dim3(kFusedThreads), 0, 0, a, geo, ws, sink); };
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-40s G=%d K=%d %8.2f us/launch  %8.1f GB/s\n", name, G, K, us, N * 22.0 / us * 1e-3);
}

int main() {
    float *r, *v, *nv, *o1, *o2; unsigned char *d, *t; FusedWs* ws;
    CK(hipMalloc(&r, N * 4)); CK(hipMalloc(&v, N * 4)); CK(hipMalloc(&nv, N * 4));
    CK(hipMalloc(&o1, N * 4)); CK(hipMalloc(&o2, N * 4)); CK(hipMalloc(&d, N)); CK(hipMalloc(&t, N)); CK(hipMalloc(&ws, kFusedWsBytes));
    CK(hipMemset(ws, 0, kFusedWsBytes));
    std::vector<float> h(N); for (i64 i = 0; i < N; ++i) h[i] = (float)rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(r, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(nv, h.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<unsigned char> hm(N); for (i64 i = 0; i < N; ++i) hm[i] = (rand() % 100) == 0;
    CK(hipMemcpy(d, hm.data(), N, hipMemcpyHostToDevice)); CK(hipMemcpy(t, hm.data(), N, hipMemcpyHostToDevice));
    GaeArgs a{r, v, nv, d, t, nullptr, nullptr, 0.99f, 0.95f, 0.99f * 0.95f, 65536, 32, 1, o1, o2, nullptr};
    const int G = 256;
    FusedGeom geo{65536, 32, 32, 8, 65536 / 32, (double)N, 1e-4};
    run<0, 2>("full", a, geo, ws, G, 300);
    run<1, 2>("no gather", a, geo, ws, G, 300);
    run<2, 2>("no gather, no block reduce", a, geo, ws, G, 300);
    run<3, 2>("no gather/reduce/scan steps", a, geo, ws, G, 300);
    run<4, 2>("full, vt not deferred", a, geo, ws, G, 300);
    double* part; CK(hipMalloc(&part, 16 * 1024));
    run<5, 2>("plain scan (adv+vt stored at once)", a, geo, ws, G, 300);
    run<6, 2>("plain scan + block partials", a, geo, ws, G, 300, part);
    run<0, 2>("full (again)", a, geo, ws, G, 300);
    return 0;
}
