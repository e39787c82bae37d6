// Shared device helpers for libskp_hip.so (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/skp.h"

#define SKP_WAVE 64
#define SKP_LOG2E 1.4426950408889634f

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Developer overrides of launch plans (skp_tune_set in include/skp.h): tests and tools/ only, 0 = the library's own choice.
// The library reads no environment variables.
enum { SKP_TUNE_WINO_SPLIT = 0, SKP_TUNE_WINO_RAW_MAX_TILES, SKP_TUNE_MAP_BANDS, SKP_TUNE_FA2_TWO_KERNEL_BWD, SKP_TUNE_GN_FOLD_MAX_COUT, SKP_TUNE_CROSS_ATTN_TS,
       SKP_TUNE_COUNT };
int skp_tune(int key);                              // skp_select_loss.hip

static inline int skp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// PyTorch upsample_bicubic2d taps (align_corners=False, A=-0.75): source coordinate is NOT
// clamped for cubic mode; tap indices are clamped to [0, n-1] on access.
__device__ __forceinline__ void skp_cubic_taps(int dst, float ratio, int n, int idx[4], float w[4]) {
    const float A = -0.75f;
    const float src = ratio * ((float)dst + 0.5f) - 0.5f;
    const float fl = floorf(src);
    const float t = src - fl;
    const int i0 = (int)fl;
    float x = t + 1.0f;
    w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int v = i0 - 1 + i;
        idx[i] = v < 0 ? 0 : (v > n - 1 ? n - 1 : v);
    }
}

// 64-lane butterfly reductions (all lanes end with the result).
__device__ __forceinline__ float skp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float skp_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Deterministic block sum for 256-thread blocks; `red` = 4 floats of LDS. All threads get the result.
__device__ __forceinline__ float skp_block_sum_256(float v, float* red) {
    v = skp_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Raw buffer loads: wave-uniform base in SGPRs, 32-bit per-lane byte offset, scalar offset for the channel step;
// an offset with bit 31 set is out of range for the descriptor and returns 0 (used for the zero padding).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ float skp_buf_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ f32x2 skp_buf_load_f32x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ f32x4 skp_buf_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ i32x4 skp_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long v = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));      // stride 0
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}
__device__ void skp_buf_store_f32(float v, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void skp_buf_store_f32x2(f32x2 v, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void skp_buf_store_f32x4(f32x4 v, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
#define SKP_OOB ((int)0x80000000)      /* loads return 0, stores are dropped */
// LDS DMA: `size` bytes per lane (4, 12 or 16) from rsrc[voffset + soffset + offset] straight to LDS at (wave-uniform) lds + lane * size;
// counted by vmcnt, no staging registers.  size / offset / aux must be compile-time constants.
typedef __attribute__((address_space(3))) void* skp_lds_ptr;
__device__ void skp_buf_load_lds(i32x4 rsrc, skp_lds_ptr lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
