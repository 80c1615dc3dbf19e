// TEST INFRASTRUCTURE ONLY (tests/test_gpu_fused_timeout.py): occupy compute units for a while.
// Each block asks for (almost) all the LDS of a CU and sleeps until `ticks` of the 100 MHz wall clock have passed, so no
// other block that needs LDS can become resident on that CU meanwhile -- a deterministic way to make the persistent grid of
// the fused GAE + normalise kernel only PARTLY resident.  Built by __graft_entry__.build(); never part of the product library.
#include <hip/hip_runtime.h>

__global__ void hold_kernel(unsigned long long ticks, unsigned *sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
    if (sink && lds[threadIdx.x] == 0xffffffffu) *sink = 1;  // keeps the LDS live
}

extern "C" int hold_max_lds(void) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return -1;
    return v;
}

// launch `nblocks` holders of `lds_bytes` dynamic LDS each for `seconds`; returns 0 on success
extern "C" int hold_cus(void *stream, int nblocks, int lds_bytes, double seconds) {
    if (hipFuncSetAttribute((const void *)hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 1;
    const unsigned long long ticks = (unsigned long long)(seconds * 1e8);
    hipLaunchKernelGGL(hold_kernel, dim3(nblocks), dim3(64), lds_bytes, (hipStream_t)stream, ticks, (unsigned *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
