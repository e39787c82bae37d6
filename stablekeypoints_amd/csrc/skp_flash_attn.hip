// Flash attention, second generation: v_mfma_f32_16x16x4_f32 tiles (exact fp32), software-pipelined K/V staging.
//
// Replaces the 32x32x2 kernels of skp_self_attn.hip on the hot shapes of the hooked UNet (ptp_utils.py:493-506,540:
// softmax(scale q k^T) v per head) -- the 64^2 self-attention layers with 40-wide heads dominate: N = 4096 queries and
// keys, 8 heads.  Two things cost the first generation a third of the fp32 matrix peak there:
//   * 32-row MFMA tiles pad the 40 output channels of P.V (and of dQ / dK / dV) to 64: 37 % of those MFMAs multiplied
//     padding.  16-row tiles pad 40 -> 48.
//   * K/V tiles were fetched global -> registers -> LDS between two barriers per tile with nothing in flight behind them.
//     Here the next tile's global loads are issued before the current tile's MFMAs, land in registers meanwhile, and go
//     to the other LDS buffer afterwards: one barrier per tile and no exposed memory latency.
//
// Forward (swapped products, as before, so a query's scores live in the registers of four lanes):
//   S^T[t][n] = sum_c K[t][c] Q[n][c]      A = K tile (LDS, ds_read_b64: two k-steps per read), B = Q (registers)
//   D layout of 16x16x4: lane (n = lane & 15, g = lane >> 4) holds keys t = 4 g + r, r = 0..3, of a 16-key block
//   online softmax: running max over the lane's registers + two cross-lane steps (lane ^ 16, lane ^ 32); the running SUM
//   stays a per-lane partial until the end (it needs no agreement between the four lanes of a query)
//   O^T[c][n] += sum_t V[t][c] P[n][t]      A = V tile (LDS), B = P straight from the S registers: in k-step (kt, r)
//   lane group g contracts key 16 kt + 4 g + r, so the probability registers are the B operand as they are.
//
// Backward: see skp_fa2_bwd_kernel below.
#include "skp_attn_tiles.h"
#include <stdlib.h>

namespace {

template <int D>
struct FA2 {
    static constexpr int LDK = D + 4;           // LDS row stride (floats): conflict-free b64 K reads and b32 V reads for D = 40/64/80
    static constexpr int D8 = D / 8;            // ds_read_b64 per key row = two k-steps each
    static constexpr int CT = (D + 15) / 16;    // 16-channel output tiles
    static constexpr int KT = 64;               // keys per LDS tile
    static constexpr int TILE = KT * LDK;       // floats per staged matrix
    static constexpr int Q4 = D / 4;            // float4 per row
    static constexpr int U = (KT * Q4 + 255) / 256;   // float4 per thread and staged matrix
};

// global -> registers (rows beyond `rows_left` read as zero)
template <int D>
__device__ __forceinline__ void fa2_fetch(f32x4 (&r)[FA2<D>::U], const float* __restrict__ src, int rows_left, int C, int tid) {
    using F = FA2<D>;
#pragma unroll
    for (int u = 0; u < F::U; ++u) {
        const int idx = tid + 256 * u;
        const int t = idx / F::Q4, c4 = idx - t * F::Q4;
        r[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < F::KT * F::Q4 && t < rows_left) r[u] = *(const f32x4*)(src + (size_t)t * C + c4 * 4);
    }
}

template <int D>
__device__ __forceinline__ void fa2_put(float* __restrict__ dst, const f32x4 (&r)[FA2<D>::U], int tid) {
    using F = FA2<D>;
#pragma unroll
    for (int u = 0; u < F::U; ++u) {
        const int idx = tid + 256 * u;
        const int t = idx / F::Q4, c4 = idx - t * F::Q4;
        if (idx < F::KT * F::Q4) *(f32x4*)(dst + t * F::LDK + c4 * 4) = r[u];
    }
}

// acc[kt][nt] += X[16 kt + i16][:] . Y_nt[:]   (X rows from LDS, Y in registers as f32x2 fragments [D8])
template <int D, int NKT, int NQT>
__device__ __forceinline__ void fa2_rowdot(const float* __restrict__ X, const f32x2 (&y)[NQT][FA2<D>::D8],
                                           f32x4 (&acc)[NKT][NQT], int i16, int g) {
    using F = FA2<D>;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const float* row = X + (16 * kt + i16) * F::LDK + 2 * g;
#pragma unroll
        for (int jj = 0; jj < F::D8; ++jj) {
            const f32x2 a = *(const f32x2*)(row + 8 * jj);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt)
                    acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], y[nt][jj][m], acc[kt][nt], 0, 0, 0);
        }
    }
}

// o[ct][nt] += sum over the tile's rows t of X[t][16 ct + i16] * p[kt][nt][r]   (t = 16 kt + 4 g + r)
template <int D, int NKT, int NQT>
__device__ __forceinline__ void fa2_colacc(const float* __restrict__ X, const f32x4 (&p)[NKT][NQT],
                                           f32x4 (&o)[FA2<D>::CT][NQT], int i16, int g) {
    using F = FA2<D>;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* row = X + (16 * kt + 4 * g + r) * F::LDK + i16;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const float a = row[16 * ct];
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt)
                    o[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[kt][nt][r], o[ct][nt], 0, 0, 0);
            }
        }
}

// Reductions over the four lanes (g = 0..3) that share a column.  v_permlane16_swap / v_permlane32_swap (gfx950) exchange
// 16-lane rows / 32-lane halves between two registers on the VALU: with both operands = v the two results hold
// {v[lane], v[lane ^ 16]} (resp. ^ 32) in every lane -- no LDS round trip as with ds_bpermute.
typedef unsigned fa2_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fa2_max4(float v) {
    fa2_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fa2_sum4(float v) {
    fa2_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace

// grid (ceil(N / (64 NQT)), H, B), 256 threads; wave w owns queries [blk*64*NQT + w*16*NQT, +16 NQT)
template <int D, int NQT, int MINW>
__global__ __launch_bounds__(256, MINW) void skp_fa2_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, float* __restrict__ out,
                                                          float* __restrict__ lse, int H, int N, int Nk, int kvb, float scale) {
    using F = FA2<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 buffers][K | V][TILE]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int nbase = blockIdx.x * (64 * NQT) + wave * (16 * NQT);
    const size_t hoff = (size_t)(kvb ? b : 0) * Nk * C + (size_t)h * D;
    const float* kg = k + hoff;
    const float* vg = v + hoff;
    const float sl2 = scale * SKP_LOG2E;

    f32x2 qf[NQT][F::D8];
    int nrow[NQT];
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nbase + 16 * nt + i16;
        nrow[nt] = n;
        const float* qrow = q + ((size_t)b * N + (n < N ? n : N - 1)) * C + h * D + 2 * g;
#pragma unroll
        for (int jj = 0; jj < F::D8; ++jj) qf[nt][jj] = *(const f32x2*)(qrow + 8 * jj) * sl2;
    }
    f32x4 o[F::CT][NQT];
#pragma unroll
    for (int ct = 0; ct < F::CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) o[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[NQT], lpart[NQT];
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) { mrun[nt] = -INFINITY; lpart[nt] = 0.f; }

    f32x4 kr[F::U], vr[F::U];
    fa2_fetch<D>(kr, kg, Nk, C, tid);
    fa2_fetch<D>(vr, vg, Nk, C, tid);
    fa2_put<D>(smem, kr, tid);
    fa2_put<D>(smem + F::TILE, vr, tid);
    __syncthreads();

    int cur = 0;
    for (int kt0 = 0; kt0 < Nk; kt0 += F::KT) {
        const float* Ks = smem + cur * 2 * F::TILE;
        const float* Vs = Ks + F::TILE;
        const bool more = kt0 + F::KT < Nk;
        if (more) {                                             // next tile: global loads in flight under this tile's MFMAs
            fa2_fetch<D>(kr, kg + (size_t)(kt0 + F::KT) * C, Nk - kt0 - F::KT, C, tid);
            fa2_fetch<D>(vr, vg + (size_t)(kt0 + F::KT) * C, Nk - kt0 - F::KT, C, tid);
        }
        f32x4 s[4][NQT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) s[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        fa2_rowdot<D, 4, NQT>(Ks, qf, s, i16, g);
        if (kt0 + F::KT > Nk) {                                 // ragged last tile (uniform branch)
            const int left = Nk - kt0;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * kt + 4 * g + r >= left) {
#pragma unroll
                        for (int nt = 0; nt < NQT; ++nt) s[kt][nt][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) {
            float tm = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) tm = fmaxf(tm, fmaxf(fmaxf(s[kt][nt][0], s[kt][nt][1]), fmaxf(s[kt][nt][2], s[kt][nt][3])));
            tm = fa2_max4(tm);
            const float mn = fmaxf(mrun[nt], tm);
            const float alpha = __builtin_amdgcn_exp2f(mrun[nt] - mn);     // first tile: exp2(-inf) = 0
            mrun[nt] = mn;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][nt][r] = __builtin_amdgcn_exp2f(s[kt][nt][r] - mn);
                    rs += s[kt][nt][r];
                }
            lpart[nt] = lpart[nt] * alpha + rs;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) o[ct][nt] *= alpha;
        }
        fa2_colacc<D, 4, NQT>(Vs, s, o, i16, g);
        if (more) {
            float* nb = smem + (cur ^ 1) * 2 * F::TILE;
            fa2_put<D>(nb, kr, tid);
            fa2_put<D>(nb + F::TILE, vr, tid);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const float l = fa2_sum4(lpart[nt]);
        const float inv = 1.0f / l;
        const int n = nrow[nt];
        if (n < N) {
            float* orow = out + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) *(f32x4*)(orow + c0) = o[ct][nt] * inv;
            }
            if (g == 0) lse[((size_t)b * H + h) * N + n] = (mrun[nt] + __builtin_amdgcn_logf(l)) * SKP_LN2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward: two kernels built from the same three tile products (deterministic, no atomics, no transposes).
//   dQ kernel   lane = query (as the forward): per key tile  S^T = K.Q^T, dP^T = V.dO^T, dS = P (dP - D),
//               dQ^T[c][n] += sum_t K[t][c] dS[n][t]; also writes D[n] = rowsum(dO * O) for the second kernel.
//   dK/dV kernel lane = key: per 64-query tile (Q, dO, lse, D staged with the same pipeline)  S = Q.K^T, dP = dO.V^T,
//               dV^T[c][t] += sum_n dO[n][c] P[n][t],  dK^T[c][t] += sum_n Q[n][c] dS[n][t].
// In both, the score registers are the B operand of the accumulating product as they are (k-slot g <-> row 4 g + r).
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NQT, int MINW>
__global__ __launch_bounds__(256, MINW) void skp_fa2_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ out,
                                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ dq, float* __restrict__ Dbuf, int H, int N,
                                                                   int Nk, int kvb, float scale) {
    using F = FA2<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int nbase = blockIdx.x * (64 * NQT) + wave * (16 * NQT);
    const size_t hoff = (size_t)(kvb ? b : 0) * Nk * C + (size_t)h * D;
    const float* kg = k + hoff;
    const float* vg = v + hoff;
    const float sl2 = scale * SKP_LOG2E;

    f32x2 qf[NQT][F::D8], dof[NQT][F::D8];
    float lse2[NQT], dsum[NQT];
    int nrow[NQT];
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nbase + 16 * nt + i16;
        nrow[nt] = n;
        const int nc = n < N ? n : N - 1;
        const size_t ro = ((size_t)b * N + nc) * C + h * D + 2 * g;
        float ds = 0.f;
#pragma unroll
        for (int jj = 0; jj < F::D8; ++jj) {
            qf[nt][jj] = *(const f32x2*)(q + ro + 8 * jj) * sl2;
            dof[nt][jj] = *(const f32x2*)(dout + ro + 8 * jj);
            const f32x2 ov = *(const f32x2*)(out + ro + 8 * jj);
            ds += dof[nt][jj][0] * ov[0] + dof[nt][jj][1] * ov[1];
        }
        dsum[nt] = fa2_sum4(ds);                                // rowsum(dO * O) of query n
        const size_t si = ((size_t)b * H + h) * N + nc;
        lse2[nt] = lse[si] * SKP_LOG2E;
        if (n < N && g == 0) Dbuf[si] = dsum[nt];
    }
    f32x4 dqa[F::CT][NQT];
#pragma unroll
    for (int ct = 0; ct < F::CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) dqa[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 kr[F::U], vr[F::U];
    fa2_fetch<D>(kr, kg, Nk, C, tid);
    fa2_fetch<D>(vr, vg, Nk, C, tid);
    fa2_put<D>(smem, kr, tid);
    fa2_put<D>(smem + F::TILE, vr, tid);
    __syncthreads();

    int cur = 0;
    for (int kt0 = 0; kt0 < Nk; kt0 += F::KT) {
        const float* Ks = smem + cur * 2 * F::TILE;
        const float* Vs = Ks + F::TILE;
        const bool more = kt0 + F::KT < Nk;
        if (more) {
            fa2_fetch<D>(kr, kg + (size_t)(kt0 + F::KT) * C, Nk - kt0 - F::KT, C, tid);
            fa2_fetch<D>(vr, vg + (size_t)(kt0 + F::KT) * C, Nk - kt0 - F::KT, C, tid);
        }
        f32x4 s[4][NQT], dp[4][NQT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) { s[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        fa2_rowdot<D, 4, NQT>(Ks, qf, s, i16, g);
        fa2_rowdot<D, 4, NQT>(Vs, dof, dp, i16, g);
        const int left = Nk - kt0;                              // keys beyond the end contribute nothing
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(s[kt][nt][r] - lse2[nt]);
                    if (16 * kt + 4 * g + r >= left) pr = 0.f;
                    s[kt][nt][r] = pr * (dp[kt][nt][r] - dsum[nt]);     // dS
                }
        fa2_colacc<D, 4, NQT>(Ks, s, dqa, i16, g);
        if (more) {
            float* nb = smem + (cur ^ 1) * 2 * F::TILE;
            fa2_put<D>(nb, kr, tid);
            fa2_put<D>(nb + F::TILE, vr, tid);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nrow[nt];
        if (n < N) {
            float* drow = dq + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) *(f32x4*)(drow + c0) = dqa[ct][nt] * scale;
            }
        }
    }
}

// grid (ceil(Nk / (64 NTT)), H, B); wave w owns keys [blk*64*NTT + w*16*NTT, +16 NTT).  LDS per buffer: Q | dO | lse2[64] | D[64]
template <int D, int NTT, int MINW>
__global__ __launch_bounds__(256, MINW) void skp_fa2_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ v, const float* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                    float* __restrict__ dk, float* __restrict__ dv, int H, int N,
                                                                    int Nk, int kvb, float scale) {
    using F = FA2<D>;
    constexpr int BUF = 2 * F::TILE + 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int tbase = blockIdx.x * (64 * NTT) + wave * (16 * NTT);
    const size_t hoff = (size_t)b * N * C + (size_t)h * D;
    const float* qg = q + hoff;
    const float* dog = dout + hoff;
    const size_t soff = ((size_t)b * H + h) * N;
    const float sl2 = scale * SKP_LOG2E;

    f32x2 kf[NTT][F::D8], vf[NTT][F::D8];
    int trow[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int t = tbase + 16 * tt + i16;
        trow[tt] = t;
        const size_t ro = ((size_t)(kvb ? b : 0) * Nk + (t < Nk ? t : Nk - 1)) * C + h * D + 2 * g;
#pragma unroll
        for (int jj = 0; jj < F::D8; ++jj) {
            kf[tt][jj] = *(const f32x2*)(k + ro + 8 * jj) * sl2;
            vf[tt][jj] = *(const f32x2*)(v + ro + 8 * jj);
        }
    }
    f32x4 dka[F::CT][NTT], dva[F::CT][NTT];
#pragma unroll
    for (int ct = 0; ct < F::CT; ++ct)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) { dka[ct][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[ct][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // per-row statistics of a query tile: thread tid < 64 carries lse2, 64 <= tid < 128 carries D
    auto fetch_stats = [&](int q0) -> float {
        const int i = tid & 63, n = q0 + i;
        if (tid >= 128 || n >= N) return 0.f;
        return tid < 64 ? lse[soff + n] * SKP_LOG2E : Dbuf[soff + n];
    };
    f32x4 qr[F::U], dr[F::U];
    float st = fetch_stats(0);
    fa2_fetch<D>(qr, qg, N, C, tid);
    fa2_fetch<D>(dr, dog, N, C, tid);
    fa2_put<D>(smem, qr, tid);
    fa2_put<D>(smem + F::TILE, dr, tid);
    if (tid < 128) smem[2 * F::TILE + tid] = st;
    __syncthreads();

    int cur = 0;
    for (int q0 = 0; q0 < N; q0 += F::KT) {
        const float* Qs = smem + cur * BUF;
        const float* dOs = Qs + F::TILE;
        const float* Ls = dOs + F::TILE;                        // lse2[64] | D[64]
        const bool more = q0 + F::KT < N;
        if (more) {
            st = fetch_stats(q0 + F::KT);
            fa2_fetch<D>(qr, qg + (size_t)(q0 + F::KT) * C, N - q0 - F::KT, C, tid);
            fa2_fetch<D>(dr, dog + (size_t)(q0 + F::KT) * C, N - q0 - F::KT, C, tid);
        }
        f32x4 s[4][NTT], dp[4][NTT];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) { s[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        fa2_rowdot<D, 4, NTT>(Qs, kf, s, i16, g);               // S[n][t]: rows = staged queries, lane = key
        fa2_rowdot<D, 4, NTT>(dOs, vf, dp, i16, g);             // dP[n][t] = dO[n] . V[t]
        const int left = N - q0;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 l4 = *(const f32x4*)(Ls + 16 * nt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(Ls + 64 + 16 * nt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 16 * nt + 4 * g + r < left;
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    float pr = __builtin_amdgcn_exp2f(s[nt][tt][r] - l4[r]);
                    if (!ok) pr = 0.f;
                    s[nt][tt][r] = pr;                                   // P
                    dp[nt][tt][r] = pr * (dp[nt][tt][r] - d4[r]);        // dS
                }
            }
        }
        fa2_colacc<D, 4, NTT>(dOs, s, dva, i16, g);             // dV^T[c][t] += sum_n dO[n][c] P[n][t]
        fa2_colacc<D, 4, NTT>(Qs, dp, dka, i16, g);             // dK^T[c][t] += sum_n Q[n][c] dS[n][t]
        if (more) {
            float* nb = smem + (cur ^ 1) * BUF;
            fa2_put<D>(nb, qr, tid);
            fa2_put<D>(nb + F::TILE, dr, tid);
            if (tid < 128) nb[2 * F::TILE + tid] = st;
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int t = trow[tt];
        if (t < Nk) {
            const size_t ro = ((size_t)b * Nk + t) * C + h * D;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) {
                    *(f32x4*)(dk + ro + c0) = dka[ct][tt] * scale;
                    *(f32x4*)(dv + ro + c0) = dva[ct][tt];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NQT, int MINW>
static int fa2_launch_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int N,
                          int Nk, int kvb, float scale, hipStream_t st) {
    using F = FA2<D>;
    const size_t lds = (size_t)4 * F::TILE * sizeof(float);
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fa2_fwd_kernel<D, NQT, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    dim3 grid((N + 64 * NQT - 1) / (64 * NQT), H, B), block(256);
    hipLaunchKernelGGL((skp_fa2_fwd_kernel<D, NQT, MINW>), grid, block, lds, st, q, k, v, out, lse, H, N, Nk, kvb, scale);
    return skp_launch_status();
}

// returns -100 when this head size is not built here (the caller falls back to the first-generation kernels)
int skp_fa2_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int Bk, int H, int N,
                int Nk, int d, float scale, void* stream) {
    const int kvb = Bk == 1 ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    const char* ev = getenv("SKP_FA2_VARIANT");                 // tile-shape A/B switch (tools/fa_bench.py)
    const int variant = ev ? atoi(ev) : 0;
    switch (d) {
        case 40:
            if (variant == 1) return fa2_launch_fwd<40, 2, 3>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
            if (variant == 2) return fa2_launch_fwd<40, 4, 2>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
            if (variant == 3) return fa2_launch_fwd<40, 1, 4>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
            return fa2_launch_fwd<40, 2, 2>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
        case 64: return fa2_launch_fwd<64, 2, 2>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
        case 80: return fa2_launch_fwd<80, 2, 2>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
        default: return -100;
    }
}

template <int D, int NQ, int MINWQ, int NT, int MINWT>
static int fa2_launch_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                          const float* lse, float* dq, float* dk, float* dv, float* ws, int B, int H, int N, int Nk, int kvb,
                          float scale, hipStream_t st) {
    using F = FA2<D>;
    const size_t lds_q = (size_t)4 * F::TILE * sizeof(float), lds_kv = (size_t)2 * (2 * F::TILE + 128) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fa2_bwd_dq_kernel<D, NQ, MINWQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_fa2_bwd_dkv_kernel<D, NT, MINWT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    dim3 block(256);
    hipLaunchKernelGGL((skp_fa2_bwd_dq_kernel<D, NQ, MINWQ>), dim3((N + 64 * NQ - 1) / (64 * NQ), H, B), block, lds_q, st,
                       q, k, v, out, dout, lse, dq, ws, H, N, Nk, kvb, scale);
    int rc = skp_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL((skp_fa2_bwd_dkv_kernel<D, NT, MINWT>), dim3((Nk + 64 * NT - 1) / (64 * NT), H, B), block, lds_kv, st,
                       q, k, v, dout, lse, ws, dk, dv, H, N, Nk, kvb, scale);
    return skp_launch_status();
}

// workspace: B*H*N floats (D = rowsum(dO * O)); -100 when the head size is not built here
int skp_fa2_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                float* dq, float* dk, float* dv, float* workspace, int B, int Bk, int H, int N, int Nk, int d, float scale,
                void* stream) {
    const int kvb = Bk == 1 ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    const char* ev = getenv("SKP_FA2_VARIANT");                 // tile-shape A/B switch (tools/fa_bench.py)
    const int variant = ev ? atoi(ev) : 0;
#define FA2_BWD(DV, NQ, WQ, NT, WT) \
    return fa2_launch_bwd<DV, NQ, WQ, NT, WT>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, Nk, kvb, scale, st)
    switch (d) {
        case 40:
            if (variant == 1) FA2_BWD(40, 1, 3, 1, 3);
            if (variant == 2) FA2_BWD(40, 2, 2, 1, 3);
            FA2_BWD(40, 2, 2, 2, 2);
        case 64:
            if (variant == 1) FA2_BWD(64, 2, 2, 2, 1);
            if (variant == 2) FA2_BWD(64, 1, 2, 1, 2);
            FA2_BWD(64, 2, 2, 1, 2);
        case 80:
            if (variant == 1) FA2_BWD(80, 2, 1, 2, 1);
            FA2_BWD(80, 1, 2, 1, 2);
        default: return -100;
    }
#undef FA2_BWD
}
