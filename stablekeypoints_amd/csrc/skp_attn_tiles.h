// Tile helpers shared by the cross- and self-attention kernels (fp32 MFMA 32x32x2, 32 rows per wave).
#pragma once
#include "skp_common.h"

#define SKP_LN2 0.6931471805599453f
template <int D8, int TT>
struct CAShape {
    static constexpr int D = D8 * 8, LDK = D + 4, CT = (D + 31) / 32, TP = TT * 32;
    static constexpr int LDS_FLOATS = TP * LDK + 64;
};

// token index of accumulator register r of t-tile tt for this lane half
__device__ __forceinline__ int ca_tok(int tt, int r, int hi) { return tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int D8, int TT>
__device__ __forceinline__ void ca_stage(float* smem, const float* __restrict__ src, int T, int C, int tid) {
    using S = CAShape<D8, TT>;
    constexpr int Q4 = S::D / 4;
    for (int idx = tid; idx < S::TP * Q4; idx += 256) {
        const int t = idx / Q4, c4 = idx - t * Q4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < T) v = *(const f32x4*)(src + (size_t)t * C + c4 * 4);
        *(f32x4*)(smem + t * S::LDK + c4 * 4) = v;
    }
}

// acc[tt] (+)= X . Y^T with X rows from LDS (tokens) and Y rows in registers (this lane's query row)
template <int D8, int TT>
__device__ __forceinline__ void ca_swapped_product(const float* smem, const f32x4 (&yv)[D8], f32x16 (&acc)[TT],
                                                   int i, int hi) {
    using S = CAShape<D8, TT>;
#pragma unroll
    for (int j = 0; j < D8; ++j) {
        f32x4 xa[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) xa[tt] = *(const f32x4*)(smem + (tt * 32 + i) * S::LDK + 8 * j + 4 * hi);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[tt][m], yv[j][m], acc[tt], 0, 0, 0);
    }
}

// o[ct] = P . X   (P in registers as A operand; X rows (tokens) from LDS as B operand)
// TRANS = false: o[n][c] += P[n][t] X[t][c]   (lane = column c, rows n in registers)
// TRANS = true : o[c][n] += X[t][c] P[n][t]   (lane = row n of P, channels c in registers) -- same reads, operands swapped
template <int D8, int TT, bool ZERO = true, bool TRANS = false>
__device__ __forceinline__ void ca_reg_product(const float* smem, const f32x16 (&p)[TT],
                                               f32x16 (&o)[CAShape<D8, TT>::CT], int i, int hi) {
    using S = CAShape<D8, TT>;
    if (ZERO) {
#pragma unroll
        for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* row = smem + ca_tok(tt, r, hi) * S::LDK + i;
#pragma unroll
            for (int ct = 0; ct < S::CT; ++ct)
                o[ct] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x2f32(row[ct * 32], p[tt][r], o[ct], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x2f32(p[tt][r], row[ct * 32], o[ct], 0, 0, 0);
        }
}

