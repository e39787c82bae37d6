// Fused cross-attention core  out = softmax_t(scale q.k^T) v  for a short key axis (T <= 128 learned
// tokens) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, matches the reference's fp32
// einsum/softmax/matmul of ptp_utils.py:493-506 to rounding).
//
// One wave = 32 queries, one workgroup = 128 queries of one (batch, head).  K (then V) of the head is
// staged ONCE per workgroup in LDS ([t][d+4], float4 fills, ds_read_b128 fragment reads).
//   S^T = K.Q^T   "swapped" product: the MFMA C/D layout then puts query n = lane&31 in a lane and its
//                 tokens in the lane's accumulator registers (+ the partner lane^32), so the softmax over
//                 tokens is in-register: max/sum over 16*TT registers + ONE cross-half exchange.
//   O = P.V       the probability registers ARE the MFMA A operand of the second product (k-slot =
//                 lane>>5 pairs token t(r,0) with t(r,1) = t(r,0)+4; V rows are fetched to match), so P
//                 never leaves registers.
// Backward: recompute P from the saved log-sum-exp, dP^T = V.dO^T (same swapped product),
// dS = P (dP - rowsum(dO*O)), dQ = scale dS.K in-kernel; P and dS are staged token-major
// ([B,H,TP,N], n contiguous, coalesced) for the two reductions over queries dK = scale dS^T.Q and
// dV = P^T.dO, which run on skp_gemm_nt_f32.
#include "skp_attn_tiles.h"

#define SKP_CA_KSPLIT_MAX 16

int skp_gemm_nt_splitk(const float* A, const float* B, float* C, float* partial, int64_t c_elems, int ksplit,
                       int M, int N, int K, int Z0, int Z1,
                       int64_t sa0, int64_t sa1, int64_t sam, int64_t sak,
                       int64_t sb0, int64_t sb1, int64_t sbn, int64_t sbk,
                       int64_t sc0, int64_t sc1, int64_t scm, float alpha, void* stream);

static int ca_ksplit(int N) { int k = N / 256; return k < 1 ? 1 : (k > SKP_CA_KSPLIT_MAX ? SKP_CA_KSPLIT_MAX : k); }

template <int D8, int TT>
__global__ __launch_bounds__(256) void skp_cross_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, float* __restrict__ out,
                                                                 float* __restrict__ lse, int Bk, int H, int N, int T,
                                                                 float scale) {
    using S = CAShape<D8, TT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int n0 = blockIdx.x * 128 + wave * 32, n = n0 + i;
    const bool nv = n < N;
    const float* qrow = q + ((size_t)b * N + (nv ? n : N - 1)) * C + h * S::D;
    const float sl2 = scale * SKP_LOG2E;
    f32x4 qv[D8];
#pragma unroll
    for (int j = 0; j < D8; ++j) qv[j] = *(const f32x4*)(qrow + 8 * j + 4 * hi) * sl2;
    const size_t kvoff = (Bk == 1 ? 0 : (size_t)b * T * C) + (size_t)h * S::D;
    ca_stage<D8, TT>(smem, k + kvoff, T, C, tid);
    __syncthreads();
    f32x16 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
    ca_swapped_product<D8, TT>(smem, qv, acc, i, hi);
    float m = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (ca_tok(tt, r, hi) >= T) acc[tt][r] = -INFINITY;
            m = fmaxf(m, acc[tt][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[tt][r] = __builtin_amdgcn_exp2f(acc[tt][r] - m); l += acc[tt][r]; }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] *= inv;
    if (nv && hi == 0) lse[((size_t)b * H + h) * N + n] = (m + __builtin_amdgcn_logf(l)) * SKP_LN2;
    __syncthreads();                                           // everyone done with K
    ca_stage<D8, TT>(smem, v + kvoff, T, C, tid);
    __syncthreads();
    f32x16 o[S::CT];
    ca_reg_product<D8, TT>(smem, acc, o, i, hi);
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct) {
        const int c = ct * 32 + i;
        if (c < S::D) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (nn < N) out[((size_t)b * N + nn) * C + h * S::D + c] = o[ct][r];
            }
        }
    }
}

template <int D8, int TT>
__global__ __launch_bounds__(256) void skp_cross_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ out,
                                                                 const float* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ dq, float* __restrict__ Pst,
                                                                 float* __restrict__ dSst, int Bk, int H, int N, int T,
                                                                 float scale) {
    using S = CAShape<D8, TT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int n0 = blockIdx.x * 128 + wave * 32, n = n0 + i;
    const bool nv = n < N;
    const size_t rowoff = ((size_t)b * N + (nv ? n : N - 1)) * C + h * S::D;
    const float sl2 = scale * SKP_LOG2E;
    const size_t kvoff = (Bk == 1 ? 0 : (size_t)b * T * C) + (size_t)h * S::D;
    ca_stage<D8, TT>(smem, k + kvoff, T, C, tid);
    f32x16 p[TT], dp[TT];
    float dsum = 0.f;
    {
        f32x4 qv[D8];
#pragma unroll
        for (int j = 0; j < D8; ++j) qv[j] = *(const f32x4*)(q + rowoff + 8 * j + 4 * hi) * sl2;
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) p[tt][r] = 0.f;
        ca_swapped_product<D8, TT>(smem, qv, p, i, hi);
    }
    const float lse2 = lse[((size_t)b * H + h) * N + (nv ? n : N - 1)] * SKP_LOG2E;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            p[tt][r] = (ca_tok(tt, r, hi) < T) ? __builtin_amdgcn_exp2f(p[tt][r] - lse2) : 0.f;
    __syncthreads();                                           // done with K
    ca_stage<D8, TT>(smem, v + kvoff, T, C, tid);
    {
        f32x4 dov[D8];
#pragma unroll
        for (int j = 0; j < D8; ++j) {
            dov[j] = *(const f32x4*)(dout + rowoff + 8 * j + 4 * hi);
            const f32x4 ov = *(const f32x4*)(out + rowoff + 8 * j + 4 * hi);
            dsum += dov[j][0] * ov[0] + dov[j][1] * ov[1] + dov[j][2] * ov[2] + dov[j][3] * ov[3];
        }
        dsum += __shfl_xor(dsum, 32, 64);                      // rowsum(dO * O) of query n
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[tt][r] = 0.f;
        ca_swapped_product<D8, TT>(smem, dov, dp, i, hi);      // dP^T = V . dO^T
    }
    const size_t stb = ((size_t)b * H + h) * S::TP * N;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dp[tt][r] = p[tt][r] * (dp[tt][r] - dsum);         // dS
            if (nv) {
                const size_t o = stb + (size_t)ca_tok(tt, r, hi) * N + n;
                Pst[o] = p[tt][r];
                dSst[o] = dp[tt][r];
            }
        }
    __syncthreads();                                           // done with V
    ca_stage<D8, TT>(smem, k + kvoff, T, C, tid);
    __syncthreads();
    f32x16 o[S::CT];
    ca_reg_product<D8, TT>(smem, dp, o, i, hi);                // dQ = scale * dS . K
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct) {
        const int c = ct * 32 + i;
        if (c < S::D) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (nn < N) dq[((size_t)b * N + nn) * C + h * S::D + c] = scale * o[ct][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Token-split form for the small layers (16^2 / 8^2: N <= 256 queries per image).  There the grid above is 0.5 waves per
// SIMD and every wave is one latency chain: 32 queries x ALL padded tokens, 2 x 240 dependent-ish 32x32x2 MFMAs at d = 160
// (12.8 us of matrix time) between two staging round trips -- 33 us forward / 48 us backward whatever the batch
// (profiles/r06_cross_attn_route.txt).  Here a workgroup is 32 queries and NW waves, wave w owns the 32-token tile w: a third
// (quarter) of each product per wave, three (four) times the waves.  Each wave stages ITS rows of K, then V, into its own LDS
// region (no workgroup barrier around the staging); the softmax statistics (forward) and the partial outputs / dQ are
// merged through LDS in fixed order.
// ---------------------------------------------------------------------------------------------------------------------
template <int D8>
struct CATS {
    using S = CAShape<D8, 1>;
    static constexpr int OUT = S::CT * 1024;                                 // floats of a wave's partial output (CT tiles x 16 x 64)
    static constexpr int REG = (32 * S::LDK > OUT ? 32 * S::LDK : OUT);      // floats per wave region
};

// rows [t0, t0 + 32) of a [T, C] matrix (head slice at `src`) -> the wave's region, rows beyond T as zeros
template <int D8>
__device__ __forceinline__ void ca_stage_wave(float* reg, const float* __restrict__ src, int t0, int T, int C, int lane) {
    using S = CAShape<D8, 1>;
    constexpr int Q4 = S::D / 4, NIT = 32 * Q4 / 64;            // (D % 8 == 0: whole iterations)
    f32x4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {                          // every load in flight before the first LDS write
        const int idx = lane + 64 * it;
        const int t = idx / Q4, c4 = idx - t * Q4;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t0 + t < T) v[it] = *(const f32x4*)(src + (size_t)(t0 + t) * C + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = lane + 64 * it;
        const int t = idx / Q4, c4 = idx - t * Q4;
        *(f32x4*)(reg + t * S::LDK + c4 * 4) = v[it];
    }
}

// partial [32 rows x D] products of the NW waves (o[ct][r] in the MFMA result layout) -> summed in wave order, rows n0 .. n0 + 31
template <int D8, int NW>
__device__ __forceinline__ void ca_merge_store(float* smem, float* myreg, const f32x16 (&o)[CAShape<D8, 1>::CT], float* __restrict__ dst,
                                               float alpha, int b, int h, int H, int N, int n0, int wave, int lane) {
    using S = CAShape<D8, 1>;
    using X = CATS<D8>;
    const int i = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) myreg[(ct * 16 + r) * 64 + lane] = o[ct][r];
    __syncthreads();
    const int C = H * S::D;
    for (int ct = wave; ct < S::CT; ct += NW) {
        const int c = ct * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float acc = smem[(ct * 16 + r) * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) acc += smem[w * X::REG + (ct * 16 + r) * 64 + lane];
            const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (c < S::D && nn < N) dst[((size_t)b * N + nn) * C + h * S::D + c] = alpha * acc;
        }
    }
}

template <int D8, int NW>
__global__ __launch_bounds__(64 * NW) void skp_cross_attn_fwd_ts_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                      const float* __restrict__ v, float* __restrict__ out,
                                                                      float* __restrict__ lse, int Bk, int H, int N, int T,
                                                                      float scale) {
    using S = CAShape<D8, 1>;
    using X = CATS<D8>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int n0 = blockIdx.x * 32, n = n0 + i, t0 = wave * 32;
    const bool nv = n < N;
    float* reg = smem + wave * X::REG;
    float* st = smem + NW * X::REG;                              // [2][NW][32]: tile maxima, tile sums
    const size_t kvoff = (Bk == 1 ? 0 : (size_t)b * T * C) + (size_t)h * S::D;
    ca_stage_wave<D8>(reg, k + kvoff, t0, T, C, lane);
    const float* qrow = q + ((size_t)b * N + (nv ? n : N - 1)) * C + h * S::D;
    const float sl2 = scale * SKP_LOG2E;
    f32x4 qv[D8];
#pragma unroll
    for (int j = 0; j < D8; ++j) qv[j] = *(const f32x4*)(qrow + 8 * j + 4 * hi) * sl2;
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    ca_swapped_product<D8, 1>(reg, qv, acc, i, hi);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (t0 + ca_tok(0, r, hi) >= T) acc[0][r] = -INFINITY;
        m = fmaxf(m, acc[0][r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (hi == 0) st[wave * 32 + i] = m;
    ca_stage_wave<D8>(reg, v + kvoff, t0, T, C, lane);         // (this wave is done with its K rows; nobody else reads the region)
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) m = fmaxf(m, st[w * 32 + i]);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = __builtin_amdgcn_exp2f(acc[0][r] - m); l += acc[0][r]; }
    l += __shfl_xor(l, 32, 64);
    if (hi == 0) st[NW * 32 + wave * 32 + i] = l;
    __syncthreads();
    l = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) l += st[NW * 32 + w * 32 + i];
    const float inv = 1.0f / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] *= inv;
    if (nv && hi == 0 && wave == 0) lse[((size_t)b * H + h) * N + n] = (m + __builtin_amdgcn_logf(l)) * SKP_LN2;
    f32x16 o[S::CT];
    ca_reg_product<D8, 1>(reg, acc, o, i, hi);
    ca_merge_store<D8, NW>(smem, reg, o, out, 1.0f, b, h, H, N, n0, wave, lane);
}

template <int D8, int NW>
__global__ __launch_bounds__(64 * NW) void skp_cross_attn_bwd_ts_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                      const float* __restrict__ v, const float* __restrict__ out,
                                                                      const float* __restrict__ dout, const float* __restrict__ lse,
                                                                      float* __restrict__ dq, float* __restrict__ Pst,
                                                                      float* __restrict__ dSst, int Bk, int H, int N, int T, int TP,
                                                                      float scale) {
    using S = CAShape<D8, 1>;
    using X = CATS<D8>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, C = H * S::D;
    const int n0 = blockIdx.x * 32, n = n0 + i, t0 = wave * 32;
    const bool nv = n < N;
    float* reg = smem + wave * X::REG;
    const size_t rowoff = ((size_t)b * N + (nv ? n : N - 1)) * C + h * S::D;
    const float sl2 = scale * SKP_LOG2E;
    const size_t kvoff = (Bk == 1 ? 0 : (size_t)b * T * C) + (size_t)h * S::D;
    ca_stage_wave<D8>(reg, k + kvoff, t0, T, C, lane);
    f32x16 p[1], dp[1];
    float dsum = 0.f;
    {
        f32x4 qv[D8];
#pragma unroll
        for (int j = 0; j < D8; ++j) qv[j] = *(const f32x4*)(q + rowoff + 8 * j + 4 * hi) * sl2;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[0][r] = 0.f;
        ca_swapped_product<D8, 1>(reg, qv, p, i, hi);
    }
    const float lse2 = lse[((size_t)b * H + h) * N + (nv ? n : N - 1)] * SKP_LOG2E;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[0][r] = (t0 + ca_tok(0, r, hi) < T) ? __builtin_amdgcn_exp2f(p[0][r] - lse2) : 0.f;
    ca_stage_wave<D8>(reg, v + kvoff, t0, T, C, lane);
    {
        f32x4 dov[D8];
#pragma unroll
        for (int j = 0; j < D8; ++j) {
            dov[j] = *(const f32x4*)(dout + rowoff + 8 * j + 4 * hi);
            const f32x4 ov = *(const f32x4*)(out + rowoff + 8 * j + 4 * hi);
            dsum += dov[j][0] * ov[0] + dov[j][1] * ov[1] + dov[j][2] * ov[2] + dov[j][3] * ov[3];
        }
        dsum += __shfl_xor(dsum, 32, 64);                      // rowsum(dO * O) of query n
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[0][r] = 0.f;
        ca_swapped_product<D8, 1>(reg, dov, dp, i, hi);        // dP^T = V . dO^T (this wave's tokens)
    }
    const size_t stb = ((size_t)b * H + h) * TP * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dp[0][r] = p[0][r] * (dp[0][r] - dsum);                // dS
        if (nv) {
            const size_t o = stb + (size_t)(t0 + ca_tok(0, r, hi)) * N + n;
            Pst[o] = p[0][r];
            dSst[o] = dp[0][r];
        }
    }
    ca_stage_wave<D8>(reg, k + kvoff, t0, T, C, lane);
    f32x16 o[S::CT];
    ca_reg_product<D8, 1>(reg, dp, o, i, hi);                  // dQ (this wave's tokens) = dS . K
    ca_merge_store<D8, NW>(smem, reg, o, dq, scale, b, h, H, N, n0, wave, lane);
}

// The token-split form where the 128-query grid leaves the chip under one wave per SIMD (the 16^2 / 8^2 layers at any batch, the
// 32^2 layers at 1-2 images per rank): d = 80 / 160, 3 or 4 token tiles.
static bool ca_use_ts(int B, int H, int N, int T, int d) {
    if (skp_tune(SKP_TUNE_CROSS_ATTN_TS) == 1) return false;   // tests: the 128-query form at these shapes
    const int tt = (T + 31) / 32;
    if (tt < 2 || !(d == 80 || d == 160)) return false;
    return (long)((N + 127) / 128) * H * B * 4 < 1024;
}

static int ca_check(int B, int Bk, int H, int N, int T, int d) {
    if (B <= 0 || H <= 0 || N <= 0 || T <= 0 || d <= 0 || (Bk != 1 && Bk != B)) return SKP_E_BADARG;
    if (T > 128 || B > 65535 || H > 65535) return SKP_E_RANGE;
    if (d != 8 && d != 16 && d != 32 && d != 40 && d != 64 && d != 80 && d != 160) return SKP_E_RANGE;
    return 0;
}

#define SKP_CA_DISPATCH(KERNEL, ...)                                                                     \
    {                                                                                                    \
        const int tt = (T + 31) / 32;                                                                    \
        const int tsel = tt <= 1 ? 1 : (tt <= 3 ? 3 : 4);                                                \
        dim3 grid((N + 127) / 128, H, B), block(256);                                                    \
        hipStream_t st = (hipStream_t)stream;                                                            \
        int launched = 0;                                                                                \
        SKP_CA_CASE(KERNEL, 1, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 1, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 1, 4, __VA_ARGS__)     \
        SKP_CA_CASE(KERNEL, 2, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 2, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 2, 4, __VA_ARGS__)     \
        SKP_CA_CASE(KERNEL, 4, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 4, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 4, 4, __VA_ARGS__)     \
        SKP_CA_CASE(KERNEL, 5, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 5, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 5, 4, __VA_ARGS__)     \
        SKP_CA_CASE(KERNEL, 8, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 8, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 8, 4, __VA_ARGS__)     \
        SKP_CA_CASE(KERNEL, 10, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 10, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 10, 4, __VA_ARGS__)  \
        SKP_CA_CASE(KERNEL, 20, 1, __VA_ARGS__) SKP_CA_CASE(KERNEL, 20, 3, __VA_ARGS__) SKP_CA_CASE(KERNEL, 20, 4, __VA_ARGS__)  \
        if (!launched) return SKP_E_RANGE;                                                               \
    }

#define SKP_CA_CASE(KERNEL, D8V, TTV, ...)                                                               \
    if (!launched && d == D8V * 8 && tsel == TTV) {                                                      \
        const size_t lds = CAShape<D8V, TTV>::LDS_FLOATS * sizeof(float);                                \
        if (lds > 64 * 1024) {                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL<D8V, TTV>,                            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (e != hipSuccess) return (int)e;                                                          \
        }                                                                                                \
        hipLaunchKernelGGL((KERNEL<D8V, TTV>), grid, block, lds, st, __VA_ARGS__);                       \
        launched = 1;                                                                                    \
    }

#define SKP_CA_TS_CASE(KERNEL, D8V, NWV, ...)                                                            \
    if (!launched && d == D8V * 8 && nw == NWV) {                                                        \
        const size_t lds = ((size_t)NWV * CATS<D8V>::REG + 2 * NWV * 32) * sizeof(float);                \
        if (lds > 64 * 1024) {                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL<D8V, NWV>,                            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (e != hipSuccess) return (int)e;                                                          \
        }                                                                                                \
        hipLaunchKernelGGL((KERNEL<D8V, NWV>), dim3((N + 31) / 32, H, B), dim3(64 * NWV), lds, (hipStream_t)stream, __VA_ARGS__); \
        launched = 1;                                                                                    \
    }
#define SKP_CA_TS_DISPATCH(KERNEL, ...)                                                                  \
    {                                                                                                    \
        const int nw = (T + 31) / 32 <= 3 ? ((T + 31) / 32 <= 2 ? 2 : 3) : 4;                            \
        int launched = 0;                                                                                \
        SKP_CA_TS_CASE(KERNEL, 10, 2, __VA_ARGS__) SKP_CA_TS_CASE(KERNEL, 10, 3, __VA_ARGS__) SKP_CA_TS_CASE(KERNEL, 10, 4, __VA_ARGS__) \
        SKP_CA_TS_CASE(KERNEL, 20, 2, __VA_ARGS__) SKP_CA_TS_CASE(KERNEL, 20, 3, __VA_ARGS__) SKP_CA_TS_CASE(KERNEL, 20, 4, __VA_ARGS__) \
        if (!launched) return SKP_E_RANGE;                                                               \
    }

extern "C" int skp_cross_attn_tp(int T) {                      // padded token count of the staging buffers
    const int tt = (T + 31) / 32;
    return (tt <= 1 ? 1 : (tt <= 3 ? 3 : 4)) * 32;
}

extern "C" int64_t skp_cross_attn_bwd_workspace(int B, int H, int N, int T, int d) {
    if (B <= 0 || H <= 0 || N <= 0 || T <= 0 || T > 128 || d <= 0) return SKP_E_BADARG;
    // token-major staging of P and dS  +  split-K partials of dk (dv reuses them)
    return (2 * (int64_t)B * H * skp_cross_attn_tp(T) * N + (int64_t)ca_ksplit(N) * B * T * H * d) * (int64_t)sizeof(float);
}

extern "C" int skp_cross_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                                      int B, int Bk, int H, int N, int T, int d, float scale, void* stream) {
    if (!q || !k || !v || !out || !lse) return SKP_E_BADARG;
    int rc = ca_check(B, Bk, H, N, T, d);
    if (rc) return rc;
    if (ca_use_ts(B, H, N, T, d)) {
        SKP_CA_TS_DISPATCH(skp_cross_attn_fwd_ts_kernel, q, k, v, out, lse, Bk, H, N, T, scale)
        return skp_launch_status();
    }
    SKP_CA_DISPATCH(skp_cross_attn_fwd_kernel, q, k, v, out, lse, Bk, H, N, T, scale)
    return skp_launch_status();
}

extern "C" int skp_cross_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out,
                                      const float* dout, const float* lse, float* dq, float* dk, float* dv,
                                      float* workspace, int B, int Bk, int H, int N, int T, int d, float scale,
                                      void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !workspace) return SKP_E_BADARG;
    int rc = ca_check(B, Bk, H, N, T, d);
    if (rc) return rc;
    const int TP = skp_cross_attn_tp(T);
    float* Pst = workspace;
    float* dSst = workspace + (size_t)B * H * TP * N;
    if (ca_use_ts(B, H, N, T, d)) {
        SKP_CA_TS_DISPATCH(skp_cross_attn_bwd_ts_kernel, q, k, v, out, dout, lse, dq, Pst, dSst, Bk, H, N, T, TP, scale)
    } else {
        SKP_CA_DISPATCH(skp_cross_attn_bwd_kernel, q, k, v, out, dout, lse, dq, Pst, dSst, Bk, H, N, T, scale)
    }
    rc = skp_launch_status();
    if (rc) return rc;
    const int64_t C = (int64_t)H * d;
    // dk[b,t,h*d+c] = scale * sum_n dS[b,h,t,n] q[b,n,h*d+c] ;  dv[b,t,h*d+c] = sum_n P[b,h,t,n] dout[b,n,h*d+c]
    float* part = dSst + (size_t)B * H * TP * N;
    const int ksp = ca_ksplit(N);
    const int64_t celems = (int64_t)B * T * C;
    rc = skp_gemm_nt_splitk(dSst, q, dk, part, celems, ksp, T, d, N, B, H, (int64_t)H * TP * N, (int64_t)TP * N, N, 1,
                            (int64_t)N * C, d, 1, C, (int64_t)T * C, d, C, scale, stream);
    if (rc) return rc;
    return skp_gemm_nt_splitk(Pst, dout, dv, part, celems, ksp, T, d, N, B, H, (int64_t)H * TP * N, (int64_t)TP * N, N, 1,
                              (int64_t)N * C, d, 1, C, (int64_t)T * C, d, C, 1.0f, stream);
}
