// EXPERIMENT (separate bench line, dtype "f32-emulated (bf16x3)"; the fp32-MFMA path stays the bench of record):
// fp32 GEMM of the frozen nn.Linear layers (to_q / to_k / to_v / to_out / GEGLU / proj_in / proj_out inside the hooked UNet
// forward, ptp_utils.py:227, and their input gradients) evaluated on the bf16 matrix cores by splitting every fp32 operand
// into three bf16 terms  x = h + m + l  (round-to-nearest at each step: 24 mantissa bits in total) and accumulating the six
// products  h.h + h.m + m.h + h.l + l.h + m.m  in fp32 (the dropped m.l, l.m, l.l terms are <= 2^-24 of |a||b| together).
// Measured basis (profiles/r02_mfma_valu_probe.md): v_mfma_f32_32x32x16_bf16 sustains 2.2-2.4 PF/s, i.e. 365-400 TF/s-equivalent
// for six products, against 121-135 TF/s for v_mfma_f32_16x16x4_f32.
//
//   C[M,N] = A[M,K] . B[N,K]^T (+ bias[N])       A fp32 (split in the kernel while it is staged), B pre-split planes
//   planes of a frozen weight: [3][N][K] bf16 (h, m, l), made once by skp_gemm_x3_split_f32 (optionally of the transpose, which
//   turns the same kernel into the input-gradient GEMM dX = dY . W).
//
// Tiling: workgroup 128 x 128 outputs, four waves in 2 x 2, wave 64 x 64 = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16, K stage 32
// (two MFMA k-steps, 48 MFMAs per wave and stage).  LDS: three A planes and three B planes of 128 rows x 32 bf16, rows padded to
// 80 bytes (the 16-byte operand reads of 16 consecutive rows then touch 64 distinct banks): 60 KB -> two workgroups per CU.
// The next stage's global loads are in flight under the stage's MFMAs (registers), split + LDS write between two barriers.
#include "skp_common.h"
#include "skp_lab.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int X3_TM = 128, X3_TN = 128, X3_KS = 32;
constexpr int X3_ROWB = 80;                                   // bytes per LDS row (64 used)
constexpr int X3_PLANE = X3_TM * X3_ROWB;                      // bytes per plane tile

__device__ __forceinline__ void x3_split4(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)v[e];
        const float r1 = v[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
    }
}

// planes[p][r][c] = term p of w[r][c] (transpose = 0) or of w[c][r] (transpose = 1); rows x cols is the shape of the planes
__global__ __launch_bounds__(256) void skp_x3_split_kernel(const float* __restrict__ w, __bf16* __restrict__ planes, int rows,
                                                          int cols, int transpose) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)rows * cols;
    if (i >= n) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const float v = transpose ? w[(size_t)c * rows + r] : w[i];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    planes[i] = h; planes[n + i] = m; planes[2 * n + i] = l;
}

__global__ __launch_bounds__(256, 2) void skp_gemm_x3_kernel(const float* __restrict__ A, const __bf16* __restrict__ Bp,
                                                            const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                            int K, long lda, long ldc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                                  // [3][128][80 B]
    unsigned char* Bs = smem + 3 * X3_PLANE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: consecutive workgroup ids share the A row panel (blockIdx.x walks N fastest)
    const int m0 = blockIdx.y * X3_TM, n0 = blockIdx.x * X3_TN;
    const int srow = tid >> 1, shalf = tid & 1;               // staging: one half row (16 k) per thread
    // raw buffer loads: rows past M (A) fall outside the descriptor and read as zero; rows past N (B planes) are sent there
    const i32x4 a_rs = skp_make_rsrc(A, (unsigned)((size_t)M * lda * sizeof(float)));
    const size_t plane_g = (size_t)N * K;
    const i32x4 b_rs = skp_make_rsrc(Bp, (unsigned)(3 * plane_g * sizeof(__bf16)));
    const int a_off = (int)(((size_t)(m0 + srow) * lda + shalf * 16) * sizeof(float));
    const bool b_ok = n0 + srow < N;
    const int b_off = b_ok ? (int)(((size_t)(n0 + srow) * K + shalf * 16) * sizeof(__bf16)) : (int)0x80000000;
    const bool a_ok = m0 + srow < M;

    f32x4 ar[4];
    f32x4 br[3][2];                                            // 8 bf16 each
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ar[q] = skp_buf_load_f32x4(a_rs, a_ok ? a_off + (k0 + 4 * q) * 4 : (int)0x80000000, 0, 0);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                br[p][q] = skp_buf_load_f32x4(b_rs, b_off + (int)((p * plane_g + k0 + 8 * q) * sizeof(__bf16)), 0, 0);
    };
    bf16x8 sp[3][2];                                           // the A half row, split
    auto split = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x4 h0, m0_, l0, h1, m1, l1;
            x3_split4(ar[2 * q], h0, m0_, l0);
            x3_split4(ar[2 * q + 1], h1, m1, l1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sp[0][q][e] = h0[e]; sp[0][q][4 + e] = h1[e];
                sp[1][q][e] = m0_[e]; sp[1][q][4 + e] = m1[e];
                sp[2][q][e] = l0[e]; sp[2][q][4 + e] = l1[e];
            }
        }
    };
    auto put = [&]() {
        unsigned char* ad = As + srow * X3_ROWB + shalf * 32;
        unsigned char* bd = Bs + srow * X3_ROWB + shalf * 32;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                *(bf16x8*)(ad + p * X3_PLANE + 16 * q) = sp[p][q];
                *(f32x4*)(bd + p * X3_PLANE + 16 * q) = br[p][q];
            }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    fetch(0);
    split();
    put();
    __syncthreads();
    const int r32 = lane & 31, kh = lane >> 5;
    const unsigned char* afrag = As + (wm * 64 + r32) * X3_ROWB + kh * 16;
    const unsigned char* bfrag = Bs + (wn * 64 + r32) * X3_ROWB + kh * 16;
    for (int k0 = 0; k0 < K; k0 += X3_KS) {
        const bool more = k0 + X3_KS < K;
        if (more) fetch(k0 + X3_KS);
        bf16x8 af[2][2][3], bf[2][2][3];                       // [k-step][block][term]: all operand reads before the MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    af[ks][i][p] = *(const bf16x8*)(afrag + p * X3_PLANE + i * 32 * X3_ROWB + ks * 32);
                    bf[ks][i][p] = *(const bf16x8*)(bfrag + p * X3_PLANE + i * 32 * X3_ROWB + ks * 32);
                }
        // six products per block, small terms first; term-major order: an accumulator is reused every fourth MFMA
        constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][TA[t]], bf[ks][j][TB[t]], acc[i][j], 0, 0, 0);
        if (more) split();                                     // VALU work of the next stage under the tail of the MFMAs
        __syncthreads();                                       // everyone is done with this stage's planes
        if (more) put();
        __syncthreads();
    }
    // D layout of the 32x32 tile: register e -> row 8*(e/4) + 4*(lane/32) + e%4, column lane%32
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + r32;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + 8 * (e >> 2) + 4 * kh + (e & 3);
                if (row < M) C[(size_t)row * ldc + col] = acc[i][j][e] + bv;
            }
    }
}

}  // namespace

extern "C" int skp_gemm_x3_split_f32(const void* w, void* planes, int rows, int cols, int transpose, void* stream) {
    if (!w || !planes || rows <= 0 || cols <= 0) return SKP_E_BADARG;
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(skp_x3_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                       (__bf16*)planes, rows, cols, transpose ? 1 : 0);
    return skp_launch_status();
}

extern "C" int skp_gemm_x3_nt_f32(const void* a, const void* b_planes, const void* bias, void* c, int M, int N, int K, int64_t lda,
                                  int64_t ldc, void* stream) {
    if (!a || !b_planes || !c || M <= 0 || N <= 0 || K <= 0) return SKP_E_BADARG;
    if (K % X3_KS || lda < K || ldc < N || (lda & 3) || (((uintptr_t)a) & 15) || (((uintptr_t)b_planes) & 15)) return SKP_E_RANGE;
    if ((size_t)M * lda * sizeof(float) >= (1ull << 31) || (size_t)3 * N * K * 2 >= (1ull << 31)) return SKP_E_RANGE;   // 32-bit buffer offsets
    const long gm = (M + X3_TM - 1) / X3_TM, gn = (N + X3_TN - 1) / X3_TN;
    if (gm > 65535 || gn > 65535) return SKP_E_RANGE;
    const size_t lds = 6 * (size_t)X3_PLANE;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_gemm_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(skp_gemm_x3_kernel, dim3((unsigned)gn, (unsigned)gm), dim3(256), lds, (hipStream_t)stream, (const float*)a,
                       (const __bf16*)b_planes, (const float*)bias, (float*)c, M, N, K, (long)lda, (long)ldc);
    return skp_launch_status();
}
