// Micro-probe: do fp32 VALU instructions issued by the SAME wave hide under its fp32 MFMAs on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_valu.hip -o tools/probe_mfma_valu.out && tools/probe_mfma_valu.out
// One wave per SIMD (256 threads per CU, one workgroup per CU), a loop of 8 independent v_mfma_f32_16x16x4_f32 per
// iteration with V independent VALU instructions (v_fma_f32 / v_pk_fma_f32 / v_exp_f32) placed between them.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V, int KIND, int NM>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float va[8]; f32x2 vp[8];
    for (int i = 0; i < 8; ++i) { va[i] = threadIdx.x * 0.001f + i; vp[i] = f32x2{va[i], va[i] + 1.f}; }
    const f32x2 b2 = {b, b}, a2 = {a, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < NM) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int j = (m * V + v) & 7;
                if (KIND == 0) va[j] = __builtin_fmaf(va[j], a, b);
                if (KIND == 1) vp[j] = vp[j] * a2 + b2;
                if (KIND == 2) va[j] = __builtin_amdgcn_exp2f(va[j]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + va[i] + vp[i][0] + vp[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V, int KIND, int NM>
static void run(const char* label, float* d, int waves_per_simd) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<V, KIND, NM>), dim3(grid), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<V, KIND, NM>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per iteration per wave at 2.4 GHz (waves_per_simd waves share a SIMD)
    printf("%-44s waves/SIMD %d: %8.3f ms  -> %7.1f clk / iteration / SIMD (8 MFMA = 256 clk when NM = 8)\n", label,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, 8>("8 MFMA only", d, w);
        run<4, 0, 0>("32 v_fma_f32 only", d, w);
        run<4, 0, 8>("8 MFMA + 32 v_fma_f32 interleaved", d, w);
        run<7, 0, 8>("8 MFMA + 56 v_fma_f32 interleaved", d, w);
        run<4, 1, 0>("32 v_pk_fma_f32 only", d, w);
        run<4, 1, 8>("8 MFMA + 32 v_pk_fma_f32 interleaved", d, w);
        run<2, 2, 0>("16 v_exp_f32 only", d, w);
        run<2, 2, 8>("8 MFMA + 16 v_exp_f32 interleaved", d, w);
    }
    return 0;
}
