// Measurement aid, not part of the hot path: the rate at which a gfx950 SIMD retires back-to-back independent
// v_mfma_f32_16x16x4_f32 from one or two resident waves.  bench.py prints it next to the nominal 157.3 TF/s so that a
// kernel's roofline fraction can be read against what the matrix pipe sustains on the box the number was taken on
// (profiles/r02_mfma_valu_probe.md: ~0.77 of nominal with one wave per SIMD, ~0.86 with two; fp32 VALU work of the same
// wave is additive, not hidden).
#include "skp_common.h"
#include "skp_lab.h"

namespace {
__global__ __launch_bounds__(256) void skp_probe_mfma_kernel(float* out, int iters, float a, float b) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int skp_probe_mfma_f32(int waves_per_simd, int iters, float* scratch, float* tflops, void* stream) {
    if (!scratch || !tflops || iters <= 0) return SKP_E_BADARG;
    if (waves_per_simd < 1 || waves_per_simd > 4) return SKP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (cus > 256) cus = 256;                                    // scratch is sized for 256 CUs
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return (int)e; }
    hipLaunchKernelGGL(skp_probe_mfma_kernel, dim3(grid), dim3(256), 0, st, scratch, 64, 1.0001f, 0.5f);
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(skp_probe_mfma_kernel, dim3(grid), dim3(256), 0, st, scratch, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) return (int)e;
    // 8 MFMAs x 16*16*4*2 flops per wave and iteration, 4 waves per workgroup
    *tflops = (float)((double)grid * 4.0 * iters * 8.0 * 2048.0 / ((double)ms * 1e-3) / 1e12);
    return skp_launch_status();
}
