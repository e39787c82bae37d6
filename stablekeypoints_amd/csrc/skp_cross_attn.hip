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
    SKP_CA_DISPATCH(skp_cross_attn_bwd_kernel, q, k, v, out, dout, lse, dq, Pst, dSst, Bk, H, N, T, scale)
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
