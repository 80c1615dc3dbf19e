// Micro-benchmarks used to size the kernels (development tool; not part of the product).
// Measures, at the C4 size (2,097,152 transitions), what plain streaming kernels of the same
// byte mix achieve on this chip, for several grid shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef long long i64;
constexpr i64 N = 65536LL * 32;

// read-only: 3 float4 + 2 uint per lane, grid-stride; result kept live via rare store
__global__ __launch_bounds__(256) void k_read(const float4* r, const float4* v, const float4* nv, const unsigned* d, const unsigned* t, float* sink, i64 n4) {
    float acc = 0.f;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (i64)gridDim.x * blockDim.x) {
        float4 a = r[k], b = v[k], c = nv[k]; unsigned x = d[k], y = t[k];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + (float)(x ^ y);
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(float4* o1, float4* o2, i64 n4) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (i64)gridDim.x * blockDim.x) {
        float4 a = make_float4(1.f, 2.f, 3.f, (float)k);
        o1[k] = a; o2[k] = a;
    }
}
__global__ __launch_bounds__(256) void k_copy(const float4* r, const float4* v, const float4* nv, const unsigned* d, const unsigned* t, float4* o1, float4* o2, i64 n4) {
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (i64)gridDim.x * blockDim.x) {
        float4 a = r[k], b = v[k], c = nv[k]; unsigned x = d[k], y = t[k];
        float m = (x ^ y) ? 1.f : 0.5f;
        o1[k] = make_float4(a.x + c.x * m, a.y + c.y * m, a.z + c.z * m, a.w + c.w * m);
        o2[k] = make_float4(b.x + a.x, b.y + a.y, b.z + a.z, b.w + a.w);
    }
}
// software-pipelined copy: prefetch next tile before storing the current one
__global__ __launch_bounds__(256) void k_copy_pipe(const float4* r, const float4* v, const float4* nv, const unsigned* d, const unsigned* t, float4* o1, float4* o2, i64 n4) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n4) return;
    float4 a = r[k], b = v[k], c = nv[k]; unsigned x = d[k], y = t[k];
    for (;;) {
        const i64 kn = k + stride;
        float4 an, bn, cn; unsigned xn = 0, yn = 0;
        const bool more = kn < n4;
        if (more) { an = r[kn]; bn = v[kn]; cn = nv[kn]; xn = d[kn]; yn = t[kn]; }
        float m = (x ^ y) ? 1.f : 0.5f;
        o1[k] = make_float4(a.x + c.x * m, a.y + c.y * m, a.z + c.z * m, a.w + c.w * m);
        o2[k] = make_float4(b.x + a.x, b.y + a.y, b.z + a.z, b.w + a.w);
        if (!more) break;
        a = an; b = bn; c = cn; x = xn; y = yn; k = kn;
    }
}

template <int MODE>  // 0 plain, 1 nontemporal stores, 2 nontemporal loads+stores
__global__ __launch_bounds__(1024) void k_copy_pipe2(const float4* r, const float4* v, const float4* nv, const unsigned* d, const unsigned* t, float4* o1, float4* o2, i64 n4) {
    const i64 stride = (i64)gridDim.x * blockDim.x;
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n4) return;
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ld4 = [](const float4* p) { if (MODE == 2) { f4 q = __builtin_nontemporal_load((const f4*)p); return make_float4(q.x, q.y, q.z, q.w); } return *p; };
    auto ldu = [](const unsigned* p) { if (MODE == 2) return __builtin_nontemporal_load(p); return *p; };
    auto st4 = [](float4* p, float4 x) { if (MODE >= 1) { f4 q = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(q, (f4*)p); } else *p = x; };
    float4 a = ld4(r + k), b = ld4(v + k), c = ld4(nv + k); unsigned x = ldu(d + k), y = ldu(t + k);
    for (;;) {
        const i64 kn = k + stride;
        float4 an, bn, cn; unsigned xn = 0, yn = 0;
        const bool more = kn < n4;
        if (more) { an = ld4(r + kn); bn = ld4(v + kn); cn = ld4(nv + kn); xn = ldu(d + kn); yn = ldu(t + kn); }
        float m = (x ^ y) ? 1.f : 0.5f;
        st4(o1 + k, make_float4(a.x + c.x * m, a.y + c.y * m, a.z + c.z * m, a.w + c.w * m));
        st4(o2 + k, make_float4(b.x + a.x, b.y + a.y, b.z + a.z, b.w + a.w));
        if (!more) break;
        a = an; b = bn; c = cn; x = xn; y = yn; k = kn;
    }
}

int main(int argc, char** argv) {
    int iters = 300;
    float *r, *v, *nv, *o1, *o2, *sink; unsigned char *d, *t;
    CK(hipMalloc(&r, N * 4)); CK(hipMalloc(&v, N * 4)); CK(hipMalloc(&nv, N * 4));
    CK(hipMalloc(&o1, N * 4)); CK(hipMalloc(&o2, N * 4)); CK(hipMalloc(&d, N)); CK(hipMalloc(&t, N)); CK(hipMalloc(&sink, 64));
    std::vector<float> h(N); for (i64 i = 0; i < N; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(r, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(nv, h.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<unsigned char> hm(N); for (i64 i = 0; i < N; ++i) hm[i] = (rand() % 100) == 0;
    CK(hipMemcpy(d, hm.data(), N, hipMemcpyHostToDevice)); CK(hipMemcpy(t, hm.data(), N, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const i64 n4 = N / 4;
    auto run = [&](const char* name, double bytes, std::function<void()> f) {
        for (int i = 0; i < 20; ++i) f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / iters;
        printf("%-34s %8.2f us/launch  %8.1f GB/s\n", name, us, bytes / us * 1e-3);
    };
    int grids[] = {256, 512, 1024, 2048};
    for (int g : grids) {
        char nm[64];
        snprintf(nm, 64, "read  29.4MB grid=%d", g);
        run(nm, N * 14.0, [&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const float4*)r, (const float4*)v, (const float4*)nv, (const unsigned*)d, (const unsigned*)t, sink, n4); });
    }
    for (int g : grids) {
        char nm[64];
        snprintf(nm, 64, "write 16.8MB grid=%d", g);
        run(nm, N * 8.0, [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, (float4*)o1, (float4*)o2, n4); });
    }
    for (int g : grids) {
        char nm[64];
        snprintf(nm, 64, "copy  46.1MB grid=%d", g);
        run(nm, N * 22.0, [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const float4*)r, (const float4*)v, (const float4*)nv, (const unsigned*)d, (const unsigned*)t, (float4*)o1, (float4*)o2, n4); });
    }
    for (int g : grids) {
        char nm[64];
        snprintf(nm, 64, "copy_pipe 46.1MB grid=%d", g);
        run(nm, N * 22.0, [&] { hipLaunchKernelGGL(k_copy_pipe, dim3(g), dim3(256), 0, 0, (const float4*)r, (const float4*)v, (const float4*)nv, (const unsigned*)d, (const unsigned*)t, (float4*)o1, (float4*)o2, n4); });
    }
    for (int thr : {256, 1024}) for (int g : {256, 512, 1024}) {
        if ((i64)g * thr > n4) continue;
        char nm[64];
        snprintf(nm, 64, "pipe2 plain thr=%d grid=%d", thr, g);
        run(nm, N * 22.0, [&] { hipLaunchKernelGGL(k_copy_pipe2<0>, dim3(g), dim3(thr), 0, 0, (const float4*)r, (const float4*)v, (const float4*)nv, (const unsigned*)d, (const unsigned*)t, (float4*)o1, (float4*)o2, n4); });
        snprintf(nm, 64, "pipe2 nt-store thr=%d grid=%d", thr, g);
        run(nm, N * 22.0, [&] { hipLaunchKernelGGL(k_copy_pipe2<1>, dim3(g), dim3(thr), 0, 0, (const float4*)r, (const float4*)v, (const float4*)nv, (const unsigned*)d, (const unsigned*)t, (float4*)o1, (float4*)o2, n4); });
        snprintf(nm, 64, "pipe2 nt-ld+st thr=%d grid=%d", thr, g);
        run(nm, N * 22.0, [&] { hipLaunchKernelGGL(k_copy_pipe2<2>, dim3(g), dim3(thr), 0, 0, (const float4*)r, (const float4*)v, (const float4*)nv, (const unsigned*)d, (const unsigned*)t, (float4*)o1, (float4*)o2, n4); });
    }
    // empty-kernel launch cadence
    run("k_write n4=0 (launch cadence)", 1.0, [&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, 0, (float4*)o1, (float4*)o2, (i64)0); });
    return 0;
}
