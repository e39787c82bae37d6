// Flash-style self-attention on the fp32 matrix cores for the hooked UNet's long sequences (64^2 = 4096 and
// 32^2 = 1024 image tokens): out = softmax(scale q.k^T) v without materialising the [B*h, N, N] scores that the
// reference keeps three copies of (ptp_utils.py:493-506: sim, softmax, clone -- 4.3 GB each at B=8, 64^2).
//
// Same tile algebra as skp_cross_attn.hip (swapped products so the softmax row of a query lives in one lane):
//   forward   per 128-query workgroup, loop over 64-key tiles staged in LDS:
//             S^T = K.Q^T -> online softmax (running max m, sum l, exp2) -> O^T += V^T.P^T
//             O is accumulated TRANSPOSED (lane = query) so the online rescale by exp2(m_old - m_new) is lane-local.
//   backward  kernel dQ : per query tile, loop over key tiles:   P = exp2(S - lse), dP^T = V.dO^T, dS = P (dP - D),
//                         dQ^T += K^T.dS^T
//             kernel dKV: per 128-KEY workgroup (lane = key), loop over 64-query tiles staged in LDS:
//                         S = Q.K^T (non-swapped: lane = key), dP = dO.V^T, dV^T += dO^T.P, dK^T += Q^T.dS
//             D[n] = rowsum(dO * O) is computed by the dQ kernel and handed to the dKV kernel through HBM.
// Deterministic (no atomics).  Head dims 8/16/40/80/160 (d = 160 backward uses a 32-key tile to fit registers).
#include "skp_attn_tiles.h"
#include <stdlib.h>

// store a transposed accumulator (rows = channels, lane = query/key row) as out[row, c] * mul
template <int D8, int TT>
__device__ __forceinline__ void sa_store_t(const f32x16 (&o)[CAShape<D8, TT>::CT], float* __restrict__ rowptr, float mul,
                                           int hi) {
    using S = CAShape<D8, TT>;
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int c0 = ct * 32 + 8 * qd + 4 * hi;
            if (c0 < S::D)
                *(f32x4*)(rowptr + c0) = f32x4{o[ct][4 * qd] * mul, o[ct][4 * qd + 1] * mul, o[ct][4 * qd + 2] * mul,
                                               o[ct][4 * qd + 3] * mul};
        }
}

template <int D8, int KT32>
__global__ __launch_bounds__(256) void skp_self_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, float* __restrict__ out,
                                                                float* __restrict__ lse, int H, int N, int Nk, int kvb,
                                                                float scale) {
    // N queries, Nk keys; kvb = 1: k/v have a batch axis, 0: one k/v shared by every batch row (cross-attention with
    // the learned embedding, ptp_utils.py:229)
    using S = CAShape<D8, KT32>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + S::LDS_FLOATS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int n = blockIdx.x * 128 + wave * 32 + i;
    const bool nv = n < N;
    const size_t rowoff = ((size_t)b * N + (nv ? n : N - 1)) * C + h * S::D;
    const size_t hoff = (size_t)(kvb ? b : 0) * Nk * C + (size_t)h * S::D;
    f32x4 qv[D8];
#pragma unroll
    for (int j = 0; j < D8; ++j) qv[j] = *(const f32x4*)(q + rowoff + 8 * j + 4 * hi) * (scale * SKP_LOG2E);
    f32x16 o[S::CT];
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int kt0 = 0; kt0 < Nk; kt0 += S::TP) {
        __syncthreads();                                       // previous tile consumed
        ca_stage<D8, KT32>(Ks, k + hoff + (size_t)kt0 * C, Nk - kt0, C, tid);
        ca_stage<D8, KT32>(Vs, v + hoff + (size_t)kt0 * C, Nk - kt0, C, tid);
        __syncthreads();
        f32x16 acc[KT32];
#pragma unroll
        for (int tt = 0; tt < KT32; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
        ca_swapped_product<D8, KT32>(Ks, qv, acc, i, hi);
        float tm = -INFINITY;
        const int left = Nk - kt0;
#pragma unroll
        for (int tt = 0; tt < KT32; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ca_tok(tt, r, hi) >= left) acc[tt][r] = -INFINITY;
                tm = fmaxf(tm, acc[tt][r]);
            }
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);    // first tile: exp2(-inf) = 0
        float rs = 0.f;
#pragma unroll
        for (int tt = 0; tt < KT32; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tt][r] = __builtin_amdgcn_exp2f(acc[tt][r] - mn); rs += acc[tt][r]; }
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = mn;
#pragma unroll
        for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        ca_reg_product<D8, KT32, false, true>(Vs, acc, o, i, hi);    // O^T[c][n] += sum_t V[t][c] P[n][t]
    }
    if (nv) {
        sa_store_t<D8, KT32>(o, out + rowoff, 1.0f / l, hi);
        if (hi == 0) lse[((size_t)b * H + h) * N + n] = (m + __builtin_amdgcn_logf(l)) * SKP_LN2;
    }
}

// dQ kernel (also writes D[n] = rowsum(dO*O) for the dKV kernel)
template <int D8, int KT32>
__global__ __launch_bounds__(256) void skp_self_attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ out,
                                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ dq, float* __restrict__ Dbuf, int H,
                                                                   int N, int Nk, int kvb, float scale) {
    using S = CAShape<D8, KT32>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + S::LDS_FLOATS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int n = blockIdx.x * 128 + wave * 32 + i;
    const bool nv = n < N;
    const size_t rowoff = ((size_t)b * N + (nv ? n : N - 1)) * C + h * S::D;
    const size_t hoff = (size_t)(kvb ? b : 0) * Nk * C + (size_t)h * S::D;
    f32x4 qv[D8], dov[D8];
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < D8; ++j) {
        qv[j] = *(const f32x4*)(q + rowoff + 8 * j + 4 * hi) * (scale * SKP_LOG2E);
        dov[j] = *(const f32x4*)(dout + rowoff + 8 * j + 4 * hi);
        const f32x4 ov = *(const f32x4*)(out + rowoff + 8 * j + 4 * hi);
        dsum += dov[j][0] * ov[0] + dov[j][1] * ov[1] + dov[j][2] * ov[2] + dov[j][3] * ov[3];
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const size_t sidx = ((size_t)b * H + h) * N + (nv ? n : N - 1);
    const float lse2 = lse[sidx] * SKP_LOG2E;
    if (nv && hi == 0) Dbuf[sidx] = dsum;
    f32x16 dqa[S::CT];
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqa[ct][r] = 0.f;
    for (int kt0 = 0; kt0 < Nk; kt0 += S::TP) {
        __syncthreads();
        ca_stage<D8, KT32>(Ks, k + hoff + (size_t)kt0 * C, Nk - kt0, C, tid);
        ca_stage<D8, KT32>(Vs, v + hoff + (size_t)kt0 * C, Nk - kt0, C, tid);
        __syncthreads();
        f32x16 p[KT32], dp[KT32];
#pragma unroll
        for (int tt = 0; tt < KT32; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[tt][r] = 0.f; dp[tt][r] = 0.f; }
        ca_swapped_product<D8, KT32>(Ks, qv, p, i, hi);
        ca_swapped_product<D8, KT32>(Vs, dov, dp, i, hi);
        const int left = Nk - kt0;
#pragma unroll
        for (int tt = 0; tt < KT32; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = (ca_tok(tt, r, hi) < left) ? __builtin_amdgcn_exp2f(p[tt][r] - lse2) : 0.f;
                dp[tt][r] = pr * (dp[tt][r] - dsum);           // dS
            }
        ca_reg_product<D8, KT32, false, true>(Ks, dp, dqa, i, hi);   // dQ^T[c][n] += sum_t K[t][c] dS[n][t]
    }
    if (nv) sa_store_t<D8, KT32>(dqa, dq + rowoff, scale, hi);
}

// dK/dV kernel: lane = key.  Query tiles of QT32*32 rows (Q, dO, lse2, D) are staged in LDS.
template <int D8, int QT32>
__global__ __launch_bounds__(256) void skp_self_attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ v, const float* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                    float* __restrict__ dk, float* __restrict__ dv, int H, int N,
                                                                    int Nk, int kvb, float scale) {
    // dk, dv are written per batch row [B,Nk,C] (the caller sums over b when kvb == 0)
    using S = CAShape<D8, QT32>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* dOs = smem + S::LDS_FLOATS;
    float* Ls = dOs + S::LDS_FLOATS;                           // lse2[TP], then D[TP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int t = blockIdx.x * 128 + wave * 32 + i;            // this lane's key
    const bool tv = t < Nk;
    const size_t krow = ((size_t)(kvb ? b : 0) * Nk + (tv ? t : Nk - 1)) * C + h * S::D;
    const size_t rowoff = ((size_t)b * Nk + (tv ? t : Nk - 1)) * C + h * S::D;      // dk/dv row (always per batch row)
    const size_t hoff = (size_t)b * N * C + (size_t)h * S::D;
    const size_t soff = ((size_t)b * H + h) * N;
    f32x4 kv[D8], vv[D8];
#pragma unroll
    for (int j = 0; j < D8; ++j) {
        kv[j] = *(const f32x4*)(k + krow + 8 * j + 4 * hi) * (scale * SKP_LOG2E);
        vv[j] = *(const f32x4*)(v + krow + 8 * j + 4 * hi);
    }
    f32x16 dka[S::CT], dva[S::CT];
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[ct][r] = 0.f; dva[ct][r] = 0.f; }
    for (int q0 = 0; q0 < N; q0 += S::TP) {
        __syncthreads();
        ca_stage<D8, QT32>(Qs, q + hoff + (size_t)q0 * C, N - q0, C, tid);
        ca_stage<D8, QT32>(dOs, dout + hoff + (size_t)q0 * C, N - q0, C, tid);
        if (tid < S::TP) {
            const bool ok = q0 + tid < N;
            Ls[tid] = ok ? lse[soff + q0 + tid] * SKP_LOG2E : 0.f;
            Ls[S::TP + tid] = ok ? Dbuf[soff + q0 + tid] : 0.f;
        }
        __syncthreads();
        f32x16 p[QT32], dp[QT32];
#pragma unroll
        for (int nt = 0; nt < QT32; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[nt][r] = 0.f; dp[nt][r] = 0.f; }
        ca_swapped_product<D8, QT32>(Qs, kv, p, i, hi);        // S[n][t]: rows = staged queries, lane = key
        ca_swapped_product<D8, QT32>(dOs, vv, dp, i, hi);      // dP[n][t] = dO[n].V[t]
        const int left = N - q0;
#pragma unroll
        for (int nt = 0; nt < QT32; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nl = ca_tok(nt, r, hi);              // local query row of this register
                const float pr = (tv && nl < left) ? __builtin_amdgcn_exp2f(p[nt][r] - Ls[nl]) : 0.f;
                p[nt][r] = pr;
                dp[nt][r] = pr * (dp[nt][r] - Ls[S::TP + nl]);
            }
        ca_reg_product<D8, QT32, false, true>(dOs, p, dva, i, hi);   // dV^T[c][t] += sum_n dO[n][c] P[n][t]
        ca_reg_product<D8, QT32, false, true>(Qs, dp, dka, i, hi);   // dK^T[c][t] += sum_n Q[n][c] dS[n][t]
    }
    if (tv) {
        sa_store_t<D8, QT32>(dka, dk + rowoff, scale, hi);
        sa_store_t<D8, QT32>(dva, dv + rowoff, 1.0f, hi);
    }
}

static int sa_check(int B, int Bk, int H, int N, int Nk, int d) {
    if (B <= 0 || H <= 0 || N <= 0 || Nk <= 0 || d <= 0 || (Bk != 1 && Bk != B)) return SKP_E_BADARG;
    if (B > 65535 || H > 65535) return SKP_E_RANGE;
    if (d != 8 && d != 16 && d != 32 && d != 40 && d != 64 && d != 80 && d != 160) return SKP_E_RANGE;
    return 0;
}

#define SKP_SA_LAUNCH(KERNEL, D8V, T32V, NBUF, EXTRA, ...)                                               \
    {                                                                                                    \
        const size_t lds = ((size_t)NBUF * CAShape<D8V, T32V>::LDS_FLOATS + EXTRA) * sizeof(float);      \
        if (lds > 64 * 1024) {                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL<D8V, T32V>,                           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (e != hipSuccess) return (int)e;                                                          \
        }                                                                                                \
        hipLaunchKernelGGL((KERNEL<D8V, T32V>), grid, block, lds, st, __VA_ARGS__);                      \
    }

// second-generation kernels (skp_flash_attn.hip); -100 = head size not built there
int skp_fa2_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int Bk, int H, int N,
                int Nk, int d, float scale, void* stream);
int skp_fa2_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                float* dq, float* dk, float* dv, float* workspace, int B, int Bk, int H, int N, int Nk, int d,
                float scale, int allow_fused, int ldg, void* stream);

int64_t skp_fa2_bwd_workspace(int B, int Bk, int H, int N, int Nk, int d);

extern "C" int skp_flash_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                                      int B, int Bk, int H, int N, int Nk, int d, float scale, void* stream) {
    if (!q || !k || !v || !out || !lse) return SKP_E_BADARG;
    int rc = sa_check(B, Bk, H, N, Nk, d);
    if (rc) return rc;
    rc = skp_fa2_fwd(q, k, v, out, lse, B, Bk, H, N, Nk, d, scale, stream);      // head sizes 40 / 64 / 80 / 160
    if (rc != -100) return rc;
    const int kvb = Bk == 1 ? 0 : 1;
    dim3 grid((N + 127) / 128, H, B), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (d) {
        case 8: SKP_SA_LAUNCH(skp_self_attn_fwd_kernel, 1, 2, 2, 0, q, k, v, out, lse, H, N, Nk, kvb, scale) break;
        case 16: SKP_SA_LAUNCH(skp_self_attn_fwd_kernel, 2, 2, 2, 0, q, k, v, out, lse, H, N, Nk, kvb, scale) break;
        case 32: SKP_SA_LAUNCH(skp_self_attn_fwd_kernel, 4, 2, 2, 0, q, k, v, out, lse, H, N, Nk, kvb, scale) break;
        default: return SKP_E_RANGE;
    }
    return skp_launch_status();
}

extern "C" int64_t skp_flash_attn_bwd_workspace(int B, int Bk, int H, int N, int Nk, int d) {
    if (B <= 0 || H <= 0 || N <= 0 || Nk <= 0 || d <= 0 || (Bk != 1 && Bk != B)) return SKP_E_BADARG;
    if (d < 40) return (int64_t)B * H * N * (int64_t)sizeof(float);      // first-generation kernels (8 / 16 / 32-wide heads)
    return skp_fa2_bwd_workspace(B, Bk, H, N, Nk, d);
}

static int flash_bwd_impl(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                          float* dq, float* dk, float* dv, float* workspace, int B, int Bk, int H, int N, int Nk, int d,
                          float scale, int allow_fused, int ldg, void* stream);

/* workspace: skp_flash_attn_bwd_workspace() bytes */
extern "C" int skp_flash_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out,
                                      const float* dout, const float* lse, float* dq, float* dk, float* dv,
                                      float* workspace, int B, int Bk, int H, int N, int Nk, int d, float scale,
                                      void* stream) {
    return flash_bwd_impl(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, Bk, H, N, Nk, d, scale, 1, H * d, stream);
}

/* the same with dq, dk, dv as column bands of wider row-major buffers: row stride `ldg` floats (>= H*d, multiple of 4) -- the three
 * gradients of a self-attention block land side by side in one [B*N, 3*H*d] buffer, which makes the input gradient of its
 * frozen q / k / v projections ONE GEMM over the concatenated weights.  Second-generation head sizes only. */
extern "C" int skp_flash_attn_bwd_ld_f32(const float* q, const float* k, const float* v, const float* out,
                                         const float* dout, const float* lse, float* dq, float* dk, float* dv,
                                         float* workspace, int B, int Bk, int H, int N, int Nk, int d, float scale, int ldg,
                                         void* stream) {
    if (ldg < H * d || (ldg & 3)) return SKP_E_BADARG;
    return flash_bwd_impl(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, Bk, H, N, Nk, d, scale, 1, ldg, stream);
}

static int flash_bwd_impl(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                          float* dq, float* dk, float* dv, float* workspace, int B, int Bk, int H, int N, int Nk, int d,
                          float scale, int allow_fused, int ldg, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !workspace) return SKP_E_BADARG;
    int rc = sa_check(B, Bk, H, N, Nk, d);
    if (rc) return rc;
    rc = skp_fa2_bwd(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, Bk, H, N, Nk, d, scale, allow_fused, ldg, stream);
    if (rc != -100) return rc;
    if (ldg != H * d) return SKP_E_RANGE;                        // the first-generation kernels write dense rows only
    const int kvb = Bk == 1 ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    {
        dim3 grid((N + 127) / 128, H, B), block(256);
        switch (d) {
            case 8: SKP_SA_LAUNCH(skp_self_attn_bwd_dq_kernel, 1, 2, 2, 0, q, k, v, out, dout, lse, dq, workspace, H, N, Nk, kvb, scale) break;
            case 16: SKP_SA_LAUNCH(skp_self_attn_bwd_dq_kernel, 2, 2, 2, 0, q, k, v, out, dout, lse, dq, workspace, H, N, Nk, kvb, scale) break;
            case 32: SKP_SA_LAUNCH(skp_self_attn_bwd_dq_kernel, 4, 2, 2, 0, q, k, v, out, dout, lse, dq, workspace, H, N, Nk, kvb, scale) break;
        default: return SKP_E_RANGE;
        }
    }
    rc = skp_launch_status();
    if (rc) return rc;
    dim3 grid((Nk + 127) / 128, H, B), block(256);
    switch (d) {
        case 8: SKP_SA_LAUNCH(skp_self_attn_bwd_dkv_kernel, 1, 2, 2, 128, q, k, v, dout, lse, workspace, dk, dv, H, N, Nk, kvb, scale) break;
        case 16: SKP_SA_LAUNCH(skp_self_attn_bwd_dkv_kernel, 2, 2, 2, 128, q, k, v, dout, lse, workspace, dk, dv, H, N, Nk, kvb, scale) break;
        case 32: SKP_SA_LAUNCH(skp_self_attn_bwd_dkv_kernel, 4, 1, 2, 64, q, k, v, dout, lse, workspace, dk, dv, H, N, Nk, kvb, scale) break;
        default: return SKP_E_RANGE;
    }
    return skp_launch_status();
}

extern "C" int skp_self_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                                     int B, int H, int N, int d, float scale, void* stream) {
    return skp_flash_attn_fwd_f32(q, k, v, out, lse, B, B, H, N, N, d, scale, stream);
}

extern "C" int skp_self_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out,
                                     const float* dout, const float* lse, float* dq, float* dk, float* dv,
                                     float* workspace, int B, int H, int N, int d, float scale, void* stream) {
    // B*H*N-float workspace contract of this entry point: the two-kernel form only
    return flash_bwd_impl(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, B, H, N, N, d, scale, 0, H * d, stream);
}
