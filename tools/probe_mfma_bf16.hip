// Micro-probe: sustained rate of back-to-back independent v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 on gfx950 (one / two
// waves per SIMD), alone and with fp32 VALU work of the same wave between them.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_bf16.hip -o /tmp/probe_bf16 && /tmp/probe_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int V>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
    f32x16 acc32[4];
    f32x4 acc16[8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc32[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc32[m], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < V; ++v) { const int j = (m * V + v) & 7; va[j] = __builtin_fmaf(va[j], a, b); }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc16[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc16[m], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < V; ++v) { const int j = (m * V + v) & 7; va[j] = __builtin_fmaf(va[j], a, b); }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc32[i][j];
    for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3] + va[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int V>
static void run(const char* label, float* d, int w) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * w;
    hipLaunchKernelGGL((probe<KIND, V>), dim3(grid), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, V>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * (KIND == 0 ? 4 * 32768.0 : 8 * 16384.0);
    printf("%-50s waves/SIMD %d: %8.3f ms  %8.1f TFLOP/s\n", label, w, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("4 x mfma_f32_32x32x16_bf16", d, w);
        run<0, 4>("4 x mfma_f32_32x32x16_bf16 + 16 v_fma_f32", d, w);
        run<0, 8>("4 x mfma_f32_32x32x16_bf16 + 32 v_fma_f32", d, w);
        run<1, 0>("8 x mfma_f32_16x16x32_bf16", d, w);
        run<1, 2>("8 x mfma_f32_16x16x32_bf16 + 16 v_fma_f32", d, w);
    }
    return 0;
}
