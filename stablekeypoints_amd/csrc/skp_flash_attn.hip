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
    static constexpr int KT = D > 128 ? 32 : 64;   // keys per LDS tile (160-wide heads: four double-buffered 64-row tiles exceed the LDS)
    static constexpr int NKT = KT / 16;         // 16-row blocks per tile
    static constexpr int TILE = KT * LDK;       // floats per staged matrix
    static constexpr int Q4 = D / 4;            // float4 per row
    static constexpr int U = (KT * Q4 + 255) / 256;   // float4 per thread and staged matrix
};

// Staging of a 64-row tile: thread `tid` moves float4 number tid + 256 u of the tile.  The row / column split of that
// index is loop-invariant, so it is done once (FA2Stage) -- inside the tile loop a fetch is one load per float4.
template <int D>
struct FA2Stage {
    int goff[FA2<D>::U];      // element offset inside the global tile (row * C + 4 * c4), -1 = this thread has no slot u
    int loff[FA2<D>::U];      // float offset inside the LDS tile
    __device__ __forceinline__ void init(int C, int tid, int rows = FA2<D>::KT) {   // rows: tiles shorter than 64 rows
        using F = FA2<D>;
#pragma unroll
        for (int u = 0; u < F::U; ++u) {
            const int idx = tid + 256 * u;
            const int t = idx / F::Q4, c4 = idx - t * F::Q4;
            const bool ok = idx < rows * F::Q4;
            goff[u] = t * C + c4 * 4;
            loff[u] = ok ? t * F::LDK + c4 * 4 : -1;
        }
    }
};

// global -> registers.  FULL: every row of the tile exists (no per-row test); else rows >= rows_left read as zero.
template <int D, bool FULL>
__device__ __forceinline__ void fa2_fetch(f32x4 (&r)[FA2<D>::U], const float* __restrict__ src, const FA2Stage<D>& st, int rows_left,
                                          int tid) {
    using F = FA2<D>;
#pragma unroll
    for (int u = 0; u < F::U; ++u) {
        if (FULL) {
            if (st.loff[u] >= 0) r[u] = *(const f32x4*)(src + st.goff[u]);
        } else {                                                // ragged tile (once per kernel): redo the row split here
            r[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (st.loff[u] >= 0 && (tid + 256 * u) / F::Q4 < rows_left) r[u] = *(const f32x4*)(src + st.goff[u]);
        }
    }
}

template <int D>
__device__ __forceinline__ void fa2_put(float* __restrict__ dst, const f32x4 (&r)[FA2<D>::U], const FA2Stage<D>& st) {
    using F = FA2<D>;
#pragma unroll
    for (int u = 0; u < F::U; ++u)
        if (st.loff[u] >= 0) *(f32x4*)(dst + st.loff[u]) = r[u];
}

// on-the-fly variant (index split recomputed per tile) kept for A/B timing
template <int D>
__device__ __forceinline__ void fa2_fetch_otf(f32x4 (&r)[FA2<D>::U], const float* __restrict__ src, int rows_left, int C, int tid) {
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
__device__ __forceinline__ void fa2_put_otf(float* __restrict__ dst, const f32x4 (&r)[FA2<D>::U], int tid) {
    using F = FA2<D>;
#pragma unroll
    for (int u = 0; u < F::U; ++u) {
        const int idx = tid + 256 * u;
        const int t = idx / F::Q4, c4 = idx - t * F::Q4;
        if (idx < F::KT * F::Q4) *(f32x4*)(dst + t * F::LDK + c4 * 4) = r[u];
    }
}

// next tile (rows [r0, r0 + 64) of a matrix with `total` rows): full tiles take the test-free path
template <int D>
__device__ __forceinline__ void fa2_fetch_tile(f32x4 (&r)[FA2<D>::U], const float* __restrict__ base, int C, int r0, int total,
                                               const FA2Stage<D>& st, int tid) {
    const float* src = base + (size_t)r0 * C;
    if (r0 + FA2<D>::KT <= total) fa2_fetch<D, true>(r, src, st, 0, tid);
    else fa2_fetch<D, false>(r, src, st, total - r0, tid);
}

template <int D, bool PRE>
__device__ __forceinline__ void fa2_stage_in(f32x4 (&r)[FA2<D>::U], const float* __restrict__ base, int C, int r0, int total,
                                             const FA2Stage<D>& st, int tid) {
    if (PRE) fa2_fetch_tile<D>(r, base, C, r0, total, st, tid);
    else fa2_fetch_otf<D>(r, base + (size_t)r0 * C, total - r0, C, tid);
}
template <int D, bool PRE>
__device__ __forceinline__ void fa2_stage_out(float* __restrict__ dst, const f32x4 (&r)[FA2<D>::U], const FA2Stage<D>& st, int tid) {
    if (PRE) fa2_put<D>(dst, r, st);
    else fa2_put_otf<D>(dst, r, tid);
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

// the same with the Y fragments read from LDS rows (Y + (16 nt + i16) * LDK): saves the D8 * NQT register pairs
template <int D, int NKT, int NQT>
__device__ __forceinline__ void fa2_rowdot_lds(const float* __restrict__ X, const float* __restrict__ Y, f32x4 (&acc)[NKT][NQT],
                                               int i16, int g) {
    using F = FA2<D>;
#pragma unroll
    for (int jj = 0; jj < F::D8; ++jj) {
        f32x2 y[NQT];
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) y[nt] = *(const f32x2*)(Y + (16 * nt + i16) * F::LDK + 2 * g + 8 * jj);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const f32x2 a = *(const f32x2*)(X + (16 * kt + i16) * F::LDK + 2 * g + 8 * jj);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt)
                    acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], y[nt][m], acc[kt][nt], 0, 0, 0);
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

// Reductions over the four lanes (g = 0..3) that share a column.
// (v_permlane16_swap / v_permlane32_swap do the same exchange on the VALU; measured 1 % slower here than ds_bpermute)
__device__ __forceinline__ float fa2_max4(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float fa2_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

}  // namespace

// grid (ceil(N / (64 NQT)), H, B), 256 threads; wave w owns queries [blk*64*NQT + w*16*NQT, +16 NQT)
template <int D, int NQT, int MINW, int OPT>
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

    // Row sums through the matrix pipe: when D is not a multiple of 16 the last 16-channel tile of P.V has idle rows.
    // Column D of the staged V tile (LDS padding, never written by the staging) is set to 1, so row D of O^T accumulates
    // sum_t P[n][t] -- rescaled by alpha together with O -- and the softmax needs no VALU adds for the running sum.
    constexpr bool ONES = (D % 16) != 0 && (OPT & 2);
    constexpr int LCT = D / 16, LG = (D % 16) / 4, LR = D % 4;   // where row D of O^T lives: tile, lane group, register
    constexpr bool PRE = (OPT & 4) != 0;                        // loop-invariant staging offsets precomputed
    FA2Stage<D> stg;
    if (PRE) stg.init(C, tid);
    f32x4 kr[F::U], vr[F::U];
    if (PRE) {
        fa2_fetch_tile<D>(kr, kg, C, 0, Nk, stg, tid);
        fa2_fetch_tile<D>(vr, vg, C, 0, Nk, stg, tid);
        fa2_put<D>(smem, kr, stg);
        fa2_put<D>(smem + F::TILE, vr, stg);
    } else {
        fa2_fetch_otf<D>(kr, kg, Nk, C, tid);
        fa2_fetch_otf<D>(vr, vg, Nk, C, tid);
        fa2_put_otf<D>(smem, kr, tid);
        fa2_put_otf<D>(smem + F::TILE, vr, tid);
    }
    if (ONES && tid < 128) smem[(tid >> 6) * 2 * F::TILE + F::TILE + (tid & 63) * F::LDK + D] = 1.0f;
    __syncthreads();

    int cur = 0;
    for (int kt0 = 0; kt0 < Nk; kt0 += F::KT) {
        const float* Ks = smem + cur * 2 * F::TILE;
        const float* Vs = Ks + F::TILE;
        const bool more = kt0 + F::KT < Nk;
        if (more) {                                             // next tile: global loads in flight under this tile's MFMAs
            if (PRE) {
                fa2_fetch_tile<D>(kr, kg, C, kt0 + F::KT, Nk, stg, tid);
                fa2_fetch_tile<D>(vr, vg, C, kt0 + F::KT, Nk, stg, tid);
            } else {
                fa2_fetch_otf<D>(kr, kg + (size_t)(kt0 + F::KT) * C, Nk - kt0 - F::KT, C, tid);
                fa2_fetch_otf<D>(vr, vg + (size_t)(kt0 + F::KT) * C, Nk - kt0 - F::KT, C, tid);
            }
        }
        f32x4 s[F::NKT][NQT];
#pragma unroll
        for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) s[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        fa2_rowdot<D, F::NKT, NQT>(Ks, qf, s, i16, g);
        if (kt0 + F::KT > Nk) {                                 // ragged last tile (uniform branch)
            const int left = Nk - kt0;
#pragma unroll
            for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * kt + 4 * g + r >= left) {
#pragma unroll
                        for (int nt = 0; nt < NQT; ++nt) s[kt][nt][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) {                      // one query block after the other: the next block's max
            float tm = fmaxf(fmaxf(s[0][nt][0], s[0][nt][1]), fmaxf(s[0][nt][2], s[0][nt][3]));   // reduction overlaps this block's exps
#pragma unroll
            for (int kt = 1; kt < F::NKT; ++kt) tm = fmaxf(tm, fmaxf(fmaxf(s[kt][nt][0], s[kt][nt][1]), fmaxf(s[kt][nt][2], s[kt][nt][3])));
            tm = fa2_max4(tm);
            const float mn = fmaxf(mrun[nt], tm);
            const float alpha = __builtin_amdgcn_exp2f(mrun[nt] - mn);     // first tile: exp2(-inf) = 0
            mrun[nt] = mn;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][nt][r] = __builtin_amdgcn_exp2f(s[kt][nt][r] - mn);
                    if (!ONES) rs += s[kt][nt][r];
                }
            if (!ONES) lpart[nt] = lpart[nt] * alpha + rs;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) o[ct][nt] *= alpha;
        }
        fa2_colacc<D, F::NKT, NQT>(Vs, s, o, i16, g);
        if (more) {
            float* nb = smem + (cur ^ 1) * 2 * F::TILE;
            if (PRE) { fa2_put<D>(nb, kr, stg); fa2_put<D>(nb + F::TILE, vr, stg); }
            else { fa2_put_otf<D>(nb, kr, tid); fa2_put_otf<D>(nb + F::TILE, vr, tid); }
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        float l;
        if (ONES) l = __shfl(o[LCT][nt][LR], 16 * LG + i16, 64);          // row D of O^T = sum_t P[n][t]
        else l = fa2_sum4(lpart[nt]);
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
// Forward for SHORT key axes and wide heads (the 16^2 layers: N = 256, d = 160): the grid above is one four-wave workgroup
// per CU there (64 (batch row, head) pairs x 4 query blocks = 256), one wave per SIMD, and a workgroup's eight 32-key tiles are a
// serial chain of ~5 us each (2.4 us of MFMAs, then softmax, staging and a barrier that nothing overlaps): 47 us for 17 us of
// matrix work.  Here the workgroup has EIGHT waves: waves 0-3 take the first half of the key tiles, waves 4-7 the second half,
// for the SAME 64 queries (two waves per SIMD: one half's softmax / staging runs beside the other's MFMAs); each half keeps
// its own running max / sum / output and its own single-buffered K | V tile (the next tile travels through registers
// meanwhile), and the two partial results are merged through LDS at the end (flash-decoding's split over the key axis, inside
// the workgroup: no second launch, fixed order, bit-reproducible).
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512, 1) void skp_fa2_fwd_halves_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, float* __restrict__ out,
                                                                   float* __restrict__ lse, int H, int N, int Nk, int kvb, float scale) {
    using F = FA2<D>;
    static_assert(D % 16 == 0, "the row sums are lane partials here");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 halves][K | V][TILE]; the merge buffer lies over it
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int half = wave >> 2, wq = wave & 3, htid = tid & 255;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int nbase = blockIdx.x * 64 + wq * 16;
    const size_t hoff = (size_t)(kvb ? b : 0) * Nk * C + (size_t)h * D;
    const float* kg = k + hoff;
    const float* vg = v + hoff;
    const float sl2 = scale * SKP_LOG2E;
    const int ntiles = (Nk + F::KT - 1) / F::KT, nt0 = (ntiles + 1) / 2;
    const int tfirst = half ? nt0 : 0, tcount = half ? ntiles - nt0 : nt0;     // this half's tiles; both halves loop nt0 times

    f32x2 qf[1][F::D8];
    const int n = nbase + i16;
    {
        const float* qrow = q + ((size_t)b * N + (n < N ? n : N - 1)) * C + h * D + 2 * g;
#pragma unroll
        for (int jj = 0; jj < F::D8; ++jj) qf[0][jj] = *(const f32x2*)(qrow + 8 * jj) * sl2;
    }
    f32x4 o[F::CT][1];
#pragma unroll
    for (int ct = 0; ct < F::CT; ++ct) o[ct][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lpart = 0.f;

    FA2Stage<D> stg;
    stg.init(C, htid);
    float* Ks = smem + half * 2 * F::TILE;
    float* Vs = Ks + F::TILE;
    f32x4 kr[F::U], vr[F::U];
    if (tcount > 0) {
        fa2_fetch_tile<D>(kr, kg, C, tfirst * F::KT, Nk, stg, htid);
        fa2_fetch_tile<D>(vr, vg, C, tfirst * F::KT, Nk, stg, htid);
        fa2_put<D>(Ks, kr, stg);
        fa2_put<D>(Vs, vr, stg);
    }
    __syncthreads();
    for (int it = 0; it < nt0; ++it) {
        const bool have = it < tcount, more = it + 1 < tcount;
        const int kt0 = (tfirst + it) * F::KT;
        if (more) {                                             // next tile of this half: in flight under this tile's MFMAs
            fa2_fetch_tile<D>(kr, kg, C, kt0 + F::KT, Nk, stg, htid);
            fa2_fetch_tile<D>(vr, vg, C, kt0 + F::KT, Nk, stg, htid);
        }
        if (have) {
            f32x4 s[F::NKT][1];
#pragma unroll
            for (int kt = 0; kt < F::NKT; ++kt) s[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            fa2_rowdot<D, F::NKT, 1>(Ks, qf, s, i16, g);
            if (kt0 + F::KT > Nk) {                             // ragged last tile (wave-uniform branch)
                const int left = Nk - kt0;
#pragma unroll
                for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * kt + 4 * g + r >= left) s[kt][0][r] = -INFINITY;
            }
            float tm = fmaxf(fmaxf(s[0][0][0], s[0][0][1]), fmaxf(s[0][0][2], s[0][0][3]));
#pragma unroll
            for (int kt = 1; kt < F::NKT; ++kt) tm = fmaxf(tm, fmaxf(fmaxf(s[kt][0][0], s[kt][0][1]), fmaxf(s[kt][0][2], s[kt][0][3])));
            tm = fa2_max4(tm);
            const float mn = fmaxf(mrun, tm);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mn);
            mrun = mn;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][0][r] = __builtin_amdgcn_exp2f(s[kt][0][r] - mn);
                    rs += s[kt][0][r];
                }
            lpart = lpart * alpha + rs;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) o[ct][0] *= alpha;
            fa2_colacc<D, F::NKT, 1>(Vs, s, o, i16, g);
        }
        __syncthreads();                                        // everyone is done reading the tiles
        if (more) { fa2_put<D>(Ks, kr, stg); fa2_put<D>(Vs, vr, stg); }
        __syncthreads();
    }
    // merge: the second half parks (o, m, l partial) in LDS, the first half combines in fixed order and writes
    constexpr int MST = 4 * F::CT + 4;                          // floats per lane (o tuples, m, l, pad: 16-byte rows for ds_*_b128)
    static_assert(MST % 4 == 0, "the per-lane record is read and written with 128-bit LDS operations");
    float* mb = smem + (size_t)(wq * 64 + lane) * MST;
    if (half == 1) {
#pragma unroll
        for (int ct = 0; ct < F::CT; ++ct) *(f32x4*)(mb + 4 * ct) = o[ct][0];
        mb[4 * F::CT] = mrun;
        mb[4 * F::CT + 1] = lpart;
    }
    __syncthreads();
    if (half == 0) {
        const float m1 = mb[4 * F::CT], l1 = mb[4 * F::CT + 1];
        const float mn = fmaxf(mrun, m1);
        const float a0 = __builtin_amdgcn_exp2f(mrun - mn), a1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - mn);
        const float l = fa2_sum4(lpart * a0 + l1 * a1);
        const float inv = 1.0f / l;
        if (n < N) {
            float* orow = out + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const f32x4 o1 = *(const f32x4*)(mb + 4 * ct);
                *(f32x4*)(orow + 16 * ct + 4 * g) = (o[ct][0] * a0 + o1 * a1) * inv;
            }
            if (g == 0) lse[((size_t)b * H + h) * N + n] = (mn + __builtin_amdgcn_logf(l)) * SKP_LN2;
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
template <int D, int NQT, int MINW, bool PRE>
__global__ __launch_bounds__(256, MINW) void skp_fa2_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ out,
                                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                                   float* __restrict__ dq, float* __restrict__ Dbuf, int H, int N,
                                                                   int Nk, int kvb, float scale, int ldg, int nsplit,
                                                                   float* __restrict__ part) {
    using F = FA2<D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    // nsplit > 1 (few workgroups): blockIdx.x = query block + blocks * split; a split walks its range of key tiles and leaves a
    // partial dQ in `part` [split][B][N][C] (summed in fixed order by skp_fa2_split_reduce_kernel)
    const int nqb = gridDim.x / nsplit, split = blockIdx.x / nqb;
    const int kper = ((Nk + F::KT - 1) / F::KT + nsplit - 1) / nsplit * F::KT;
    const int k_lo = split * kper, k_hi = min(Nk, k_lo + kper);
    const int nbase = (blockIdx.x - split * nqb) * (64 * NQT) + wave * (16 * NQT);
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
        if (n < N && g == 0 && split == 0) Dbuf[si] = dsum[nt];
    }
    f32x4 dqa[F::CT][NQT];
#pragma unroll
    for (int ct = 0; ct < F::CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) dqa[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    FA2Stage<D> stg;
    if (PRE) stg.init(C, tid);
    f32x4 kr[F::U], vr[F::U];
    fa2_stage_in<D, PRE>(kr, kg, C, k_lo, Nk, stg, tid);
    fa2_stage_in<D, PRE>(vr, vg, C, k_lo, Nk, stg, tid);
    fa2_stage_out<D, PRE>(smem, kr, stg, tid);
    fa2_stage_out<D, PRE>(smem + F::TILE, vr, stg, tid);
    __syncthreads();

    int cur = 0;
    for (int kt0 = k_lo; kt0 < k_hi; kt0 += F::KT) {
        const float* Ks = smem + cur * 2 * F::TILE;
        const float* Vs = Ks + F::TILE;
        const bool more = kt0 + F::KT < k_hi;
        if (more) {
            fa2_stage_in<D, PRE>(kr, kg, C, kt0 + F::KT, Nk, stg, tid);
            fa2_stage_in<D, PRE>(vr, vg, C, kt0 + F::KT, Nk, stg, tid);
        }
        f32x4 s[F::NKT][NQT], dp[F::NKT][NQT];
#pragma unroll
        for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) { s[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        fa2_rowdot<D, F::NKT, NQT>(Ks, qf, s, i16, g);
        fa2_rowdot<D, F::NKT, NQT>(Vs, dof, dp, i16, g);
        if (kt0 + F::KT > Nk) {                                 // ragged last tile: keys beyond the end contribute nothing
            const int left = Nk - kt0;
#pragma unroll
            for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * kt + 4 * g + r >= left) {
#pragma unroll
                        for (int nt = 0; nt < NQT; ++nt) s[kt][nt][r] = -INFINITY;      // P = exp2(-inf) = 0
                    }
        }
#pragma unroll
        for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[kt][nt][r] = __builtin_amdgcn_exp2f(s[kt][nt][r] - lse2[nt]) * (dp[kt][nt][r] - dsum[nt]);     // dS
        fa2_colacc<D, F::NKT, NQT>(Ks, s, dqa, i16, g);
        if (more) {
            float* nb = smem + (cur ^ 1) * 2 * F::TILE;
            fa2_stage_out<D, PRE>(nb, kr, stg, tid);
            fa2_stage_out<D, PRE>(nb + F::TILE, vr, stg, tid);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nrow[nt];
        if (n < N) {
            float* drow = nsplit > 1 ? part + (((size_t)split * gridDim.z + b) * N + n) * C + h * D
                                     : dq + ((size_t)b * N + n) * ldg + h * D;      // ldg: row stride of the gradient outputs
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) *(f32x4*)(drow + c0) = dqa[ct][nt] * scale;
            }
        }
    }
}

// grid (ceil(Nk / (64 NTT)), H, B); wave w owns keys [blk*64*NTT + w*16*NTT, +16 NTT).  LDS per buffer: Q | dO | lse2[64] | D[64]
template <int D, int NTT, int MINW, bool PRE>
__global__ __launch_bounds__(256, MINW) void skp_fa2_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                    const float* __restrict__ v, const float* __restrict__ dout,
                                                                    const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                    float* __restrict__ dk, float* __restrict__ dv, int H, int N,
                                                                    int Nk, int kvb, float scale, int ldg, int nsplit,
                                                                    float* __restrict__ part) {
    using F = FA2<D>;
    constexpr int BUF = 2 * F::TILE + 2 * F::KT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    // nsplit > 1: blockIdx.x = key block + blocks * split; a split walks its range of query tiles; partial dK | dV in `part`
    // [split][{dK, dV}][B][Nk][C]
    const int nkb = gridDim.x / nsplit, split = blockIdx.x / nkb;
    const int qper = ((N + F::KT - 1) / F::KT + nsplit - 1) / nsplit * F::KT;
    const int q_lo = split * qper, q_hi = min(N, q_lo + qper);
    const int tbase = (blockIdx.x - split * nkb) * (64 * NTT) + wave * (16 * NTT);
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

    // per-row statistics of a query tile: thread tid < KT carries lse2, KT <= tid < 2 KT carries D
    auto fetch_stats = [&](int q0) -> float {
        const int i = tid & (F::KT - 1), n = q0 + i;
        if (tid >= 2 * F::KT) return 0.f;
        if (n >= N) return tid < F::KT ? INFINITY : 0.f;         // missing rows: lse2 = +inf => their P is exactly 0
        return tid < F::KT ? lse[soff + n] * SKP_LOG2E : Dbuf[soff + n];
    };
    FA2Stage<D> stg;
    if (PRE) stg.init(C, tid);
    f32x4 qr[F::U], dr[F::U];
    float st = fetch_stats(q_lo);
    fa2_stage_in<D, PRE>(qr, qg, C, q_lo, N, stg, tid);
    fa2_stage_in<D, PRE>(dr, dog, C, q_lo, N, stg, tid);
    fa2_stage_out<D, PRE>(smem, qr, stg, tid);
    fa2_stage_out<D, PRE>(smem + F::TILE, dr, stg, tid);
    if (tid < 2 * F::KT) smem[2 * F::TILE + tid] = st;
    __syncthreads();

    int cur = 0;
    for (int q0 = q_lo; q0 < q_hi; q0 += F::KT) {
        const float* Qs = smem + cur * BUF;
        const float* dOs = Qs + F::TILE;
        const float* Ls = dOs + F::TILE;                        // lse2[KT] | D[KT]
        const bool more = q0 + F::KT < q_hi;
        if (more) {
            st = fetch_stats(q0 + F::KT);
            fa2_stage_in<D, PRE>(qr, qg, C, q0 + F::KT, N, stg, tid);
            fa2_stage_in<D, PRE>(dr, dog, C, q0 + F::KT, N, stg, tid);
        }
        f32x4 s[F::NKT][NTT], dp[F::NKT][NTT];
#pragma unroll
        for (int nt = 0; nt < F::NKT; ++nt)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) { s[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        fa2_rowdot<D, F::NKT, NTT>(Qs, kf, s, i16, g);               // S[n][t]: rows = staged queries, lane = key
        fa2_rowdot<D, F::NKT, NTT>(dOs, vf, dp, i16, g);             // dP[n][t] = dO[n] . V[t]
        // rows beyond N (ragged last tile) were staged as zeros with lse2 = +inf  =>  P = exp2(0 - inf) = 0, dS = 0
#pragma unroll
        for (int nt = 0; nt < F::NKT; ++nt) {
            const f32x4 l4 = *(const f32x4*)(Ls + 16 * nt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(Ls + F::KT + 16 * nt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const float pr = __builtin_amdgcn_exp2f(s[nt][tt][r] - l4[r]);
                    s[nt][tt][r] = pr;                                   // P
                    dp[nt][tt][r] = pr * (dp[nt][tt][r] - d4[r]);        // dS
                }
        }
        fa2_colacc<D, F::NKT, NTT>(dOs, s, dva, i16, g);             // dV^T[c][t] += sum_n dO[n][c] P[n][t]
        fa2_colacc<D, F::NKT, NTT>(Qs, dp, dka, i16, g);             // dK^T[c][t] += sum_n Q[n][c] dS[n][t]
        if (more) {
            float* nb = smem + (cur ^ 1) * BUF;
            fa2_stage_out<D, PRE>(nb, qr, stg, tid);
            fa2_stage_out<D, PRE>(nb + F::TILE, dr, stg, tid);
            if (tid < 2 * F::KT) nb[2 * F::TILE + tid] = st;
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const int t = trow[tt];
        if (t < Nk) {
            const size_t rows = (size_t)gridDim.z * Nk;
            float* dkr = nsplit > 1 ? part + (((size_t)split * 2) * rows + (size_t)b * Nk + t) * C + h * D
                                    : dk + ((size_t)b * Nk + t) * ldg + h * D;
            float* dvr = nsplit > 1 ? dkr + rows * C : dv + ((size_t)b * Nk + t) * ldg + h * D;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) {
                    *(f32x4*)(dkr + c0) = dka[ct][tt] * scale;
                    *(f32x4*)(dvr + c0) = dva[ct][tt];
                }
            }
        }
    }
}

// out[row][0..C) (row stride ldg) = sum_s part[s * sstride + row * C ...]   (fixed order; blockIdx.y = which of up to three tensors)
struct FA2SplitReduce {
    const float* part[3];
    float* out[3];
    long n4[3];          // float4 per tensor
    long sstride[3];     // floats between consecutive splits of a tensor
};
__global__ __launch_bounds__(256) void skp_fa2_split_reduce_kernel(FA2SplitReduce a, int nsplit, int c4, int ldg4) {
    const int z = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n4[z]) return;
    f32x4 acc = ((const f32x4*)a.part[z])[i];
    for (int s = 1; s < nsplit; ++s) acc += ((const f32x4*)(a.part[z] + (size_t)s * a.sstride[z]))[i];
    ((f32x4*)a.out[z])[(i / c4) * ldg4 + (i % c4)] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward (one pass over the (query tile, key block) pairs; S and dP are computed ONCE).
//   Workgroup = 128 keys (wave w: keys 32 w .. 32 w + 31, K / V fragments in registers, lane = key), loop over 64-query tiles
//   (Q, dO, lse2, D staged in LDS, next tile prefetched in registers):
//     S = Q.K^T, dP = dO.V^T, P = exp2(S - lse2), dS = P (dP - D)                   (as in the dK/dV kernel)
//     dV^T += dO^T.P, dK^T += Q^T.dS                                                (score registers are the B operand)
//     dS is written to LDS as [query][key of the block] -- the transpose the dQ product needs --, then wave w computes
//     dQ^T[c][n] = sum_{t < 128} K[t][c] dS[n][t] for ITS 16 queries over all 128 keys of the block (A = the block's K
//     rows kept in LDS, B = one ds_read_b128 per 16 keys), complete for this key block, and stores it as a partial.
//   dQ = scale * sum over key blocks of the partials: a second, memory-bound kernel adds them in fixed order (no atomics:
//   bit-reproducible).  Issued MFMA tiles per 2048 (query, key) pairs: 448 (two-kernel form: 608; useful: 400).
//   LDS: Q | dO tile 22.5 KB, K block 22.5 KB, dS exchange 33.8 KB, statistics 0.5 KB = 79.3 KB -> two workgroups per CU (d = 40);
//   d = 80: 117 KB and 380 registers -> one workgroup per CU (MINB = 1).
// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool OVL = false, int NQ = 4>
struct FA2F {
    using F = FA2<D>;
    static constexpr int KB = 128, QT = 16 * NQ, LDX = KB + 4;    // NQ = 3: 48-query tiles (80-wide heads: 75.8 KB with OVL)
    // OVL: the dS exchange buffer lies over the Q | dO tiles (two more barriers per tile, 34 KB less LDS: two workgroups per CU
    // at D = 64)
    static constexpr int QTILE = QT * F::LDK;                   // floats per staged Q / dO tile
    static constexpr int OFF_Q = 0, OFF_DO = QTILE, OFF_K = 2 * QTILE, OFF_X = OVL ? 0 : OFF_K + KB * F::LDK;
    static constexpr int OFF_S = OVL ? OFF_K + KB * F::LDK : OFF_X + QT * LDX;
    static constexpr int LDS_FLOATS = OFF_S + 128;
    static_assert(!OVL || QT * LDX <= 2 * QTILE, "the exchange buffer must fit the Q | dO tiles");
};

// D[b,h,n] = sum_c dO[b,n,h,c] * O[b,n,h,c]   (one wave per 64 rows of a head would waste lanes: one thread per (row, head))
__global__ __launch_bounds__(256) void skp_fa2_rowdot_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                            float* __restrict__ Dbuf, int H, int N, int D, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;             // i = (b*H + h)*N + n
    if (i >= total) return;
    const long n = i % N, bh = i / N;
    const int h = (int)(bh % H);
    const long b = bh / H;
    const size_t ro = ((size_t)b * N + n) * (size_t)(H * D) + (size_t)h * D;
    float acc = 0.f;
    for (int c = 0; c < D; c += 4) {
        const f32x4 a = *(const f32x4*)(o + ro + c), d4 = *(const f32x4*)(dout + ro + c);
        acc += (a[0] * d4[0] + a[1] * d4[1]) + (a[2] * d4[2] + a[3] * d4[3]);
    }
    Dbuf[i] = acc;
}

// NW = 4: a wave owns 32 keys of the block (two 16-key tiles); NW = 8: 16 keys (one tile) -- half the K / V fragments and dK / dV
// accumulators per wave, so the 80-wide heads fit two waves per SIMD without spilled registers at 64-query tiles.
template <int D, int MINB, bool OVL, int NQ, int NW = 4>
__global__ __launch_bounds__(64 * NW, MINB) void skp_fa2_bwd_fused_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                   float* __restrict__ dqp, float* __restrict__ dk,
                                                                   float* __restrict__ dv, int H, int N, int Nk, float scale, int ldg,
                                                                   int nsplit, float* __restrict__ kvpart) {
    using F = FA2<D>;
    using X = FA2F<D, OVL, NQ>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem + X::OFF_Q;
    float* dOs = smem + X::OFF_DO;
    float* Ks = smem + X::OFF_K;
    float* Xs = smem + X::OFF_X;
    float* Ls = smem + X::OFF_S;                              // lse2[QT] | D[QT] (at 0 and 64)
    constexpr int TT = 8 / NW, KW = 16 * TT, NTH = 64 * NW;    // key tiles per wave, keys per wave, threads
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    const int stid = tid < 256 ? tid : (1 << 24);              // tile staging is dealt to the first 256 threads
    // nsplit > 1 (fewer key blocks than CUs): blockIdx.x = key block + blocks * split; a split walks its range of query tiles and
    // leaves partial dK | dV in `kvpart` [split][{dK, dV}][B][Nk][C] (skp_fa2_split_reduce_kernel); the dQ partials are per
    // (key block, query tile) either way
    const int nkbx = gridDim.x / nsplit, split = blockIdx.x / nkbx;
    const int kb = blockIdx.x - split * nkbx, b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int qper = ((N + X::QT - 1) / X::QT + nsplit - 1) / nsplit * X::QT;
    const int q_lo = split * qper, q_hi = min(N, q_lo + qper);
    const int t0 = kb * X::KB;
    const size_t hoff = (size_t)b * N * C + (size_t)h * D;     // self-attention: q, k, v, dout share [B, N, C]
    const float* qg = q + hoff;
    const float* dog = dout + hoff;
    const size_t soff = ((size_t)b * H + h) * N;
    const float sl2 = scale * SKP_LOG2E;

    // the block's K rows into LDS (raw, for the dQ product); K / V fragments of this wave's 32 keys into registers
    for (int idx = tid; idx < X::KB * F::Q4; idx += NTH) {
        const int t = idx / F::Q4, c4 = idx - t * F::Q4;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (t0 + t < Nk) val = *(const f32x4*)(k + ((size_t)b * Nk + t0 + t) * C + h * D + c4 * 4);
        *(f32x4*)(Ks + t * F::LDK + c4 * 4) = val;
    }
    // OVL (64-wide heads): the K fragments of the S product are read from the block's LDS copy (raw; the scale moves into
    // the exp2 argument) instead of living in registers -- with them the kernel does not fit two waves per SIMD
    f32x2 kf[OVL ? 1 : TT][OVL ? 1 : F::D8], vf[TT][F::D8];
    int trow[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int t = t0 + KW * wave + 16 * tt + i16;
        trow[tt] = t;
        const size_t ro = ((size_t)b * Nk + (t < Nk ? t : Nk - 1)) * C + h * D + 2 * g;
#pragma unroll
        for (int jj = 0; jj < F::D8; ++jj) {
            if (!OVL) kf[OVL ? 0 : tt][OVL ? 0 : jj] = *(const f32x2*)(k + ro + 8 * jj) * sl2;
            vf[tt][jj] = *(const f32x2*)(v + ro + 8 * jj);
        }
    }
    f32x4 dka[F::CT][TT], dva[F::CT][TT];
#pragma unroll
    for (int ct = 0; ct < F::CT; ++ct)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { dka[ct][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[ct][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto fetch_stats = [&](int q0) -> float {
        const int i = tid & 63, n = q0 + i;
        if (tid >= 128) return 0.f;
        if (n >= N || i >= X::QT) return tid < 64 ? INFINITY : 0.f;   // missing rows: lse2 = +inf => P = 0
        return tid < 64 ? lse[soff + n] * SKP_LOG2E : Dbuf[soff + n];
    };
    FA2Stage<D> stg;
    stg.init(C, stid, X::QT);
    f32x4 qr[F::U], dr[F::U];
    float st = fetch_stats(q_lo);
    fa2_fetch_tile<D>(qr, qg, C, q_lo, N, stg, stid);
    fa2_fetch_tile<D>(dr, dog, C, q_lo, N, stg, stid);
    fa2_put<D>(Qs, qr, stg);
    fa2_put<D>(dOs, dr, stg);
    if (tid < 128) Ls[tid] = st;
    __syncthreads();

    const size_t pstride = (size_t)gridDim.z * N * C;          // floats per key-block partial of dQ
    float* dqb = dqp + (size_t)kb * pstride + hoff;
    for (int q0 = q_lo; q0 < q_hi; q0 += X::QT) {
        const bool more = q0 + X::QT < q_hi;
        if (!OVL && more) {                                     // next query tile: in flight under this tile's MFMAs
            st = fetch_stats(q0 + X::QT);
            fa2_fetch_tile<D>(qr, qg, C, q0 + X::QT, N, stg, stid);
            fa2_fetch_tile<D>(dr, dog, C, q0 + X::QT, N, stg, stid);
        }
        f32x4 s[NQ][TT], dp[NQ][TT];
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) { s[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nt][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (OVL) fa2_rowdot_lds<D, NQ, TT>(Qs, Ks + KW * wave * F::LDK, s, i16, g);   // raw S[n][t]
        else fa2_rowdot<D, NQ, TT>(Qs, (const f32x2 (&)[TT][F::D8])kf, s, i16, g);    // S[n][t] (scaled, log2 units)
        fa2_rowdot<D, NQ, TT>(dOs, vf, dp, i16, g);              // dP[n][t]
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) {
            const f32x4 l4 = *(const f32x4*)(Ls + 16 * nt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(Ls + 64 + 16 * nt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const float pr = __builtin_amdgcn_exp2f(OVL ? __builtin_fmaf(s[nt][tt][r], sl2, -l4[r]) : s[nt][tt][r] - l4[r]);
                    s[nt][tt][r] = pr;                                   // P
                    dp[nt][tt][r] = pr * (dp[nt][tt][r] - d4[r]);        // dS
                }
        }
        fa2_colacc<D, NQ, TT>(dOs, s, dva, i16, g);               // dV^T[c][t] += sum_n dO[n][c] P[n][t]
        fa2_colacc<D, NQ, TT>(Qs, dp, dka, i16, g);               // dK^T[c][t] += sum_n Q[n][c] dS[n][t]
        // dS -> LDS, [query n][key of the block]
        if (OVL) __syncthreads();                              // the exchange buffer lies over Q | dO: everyone is done reading them
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
                    Xs[(16 * nt + 4 * g + r) * X::LDX + KW * wave + 16 * tt + i16] = dp[nt][tt][r];
        if (OVL && more) {                                     // the score registers are free now: fetch the next tile into them
            st = fetch_stats(q0 + X::QT);
            fa2_fetch_tile<D>(qr, qg, C, q0 + X::QT, N, stg, stid);
            fa2_fetch_tile<D>(dr, dog, C, q0 + X::QT, N, stg, stid);
        }
        __syncthreads();                                       // dS complete; everyone is done with this tile's Q / dO
        if (!OVL && more) {
            fa2_put<D>(Qs, qr, stg);
            fa2_put<D>(dOs, dr, stg);
            if (tid < 128) Ls[tid] = st;
        }
        // dQ^T[c][n] over the block's 128 keys
        if (NQ == 4 && NW == 4) {                              // wave w: queries 16 w .. 16 w + 15, all channel tiles (one dS read per 16 keys)
            f32x4 dqa[F::CT];
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) dqa[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xrow = Xs + (16 * wave + i16) * X::LDX + 4 * g;
#pragma unroll
            for (int kt = 0; kt < X::KB / 16; ++kt) {
                const f32x4 bx = *(const f32x4*)(xrow + 16 * kt);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* krow = Ks + (16 * kt + 4 * g + r) * F::LDK + i16;
#pragma unroll
                    for (int ct = 0; ct < F::CT; ++ct)
                        dqa[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[16 * ct], bx[r], dqa[ct], 0, 0, 0);
                }
            }
            const int n = q0 + 16 * wave + i16;
            if (n < N) {
                float* drow = dqb + (size_t)n * C;
#pragma unroll
                for (int ct = 0; ct < F::CT; ++ct) {
                    const int c0 = 16 * ct + 4 * g;
                    if (c0 < D) *(f32x4*)(drow + c0) = dqa[ct];
                }
            }
        } else {                                               // NQ query tiles x CT channel tiles dealt round-robin to the waves
            constexpr int NPAIR = NQ * F::CT;
#pragma unroll
            for (int j = 0; j < (NPAIR + NW - 1) / NW; ++j) {
                const int pair = wave + NW * j;                 // wave-uniform
                if (pair < NPAIR) {
                    const int ct = pair / NQ, qt = pair - ct * NQ;
                    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
                    const float* xrow = Xs + (16 * qt + i16) * X::LDX + 4 * g;
                    const float* kcol = Ks + (4 * g) * F::LDK + 16 * ct + i16;
#pragma unroll
                    for (int kt = 0; kt < X::KB / 16; ++kt) {
                        const f32x4 bx = *(const f32x4*)(xrow + 16 * kt);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kcol[(16 * kt + r) * F::LDK], bx[r], acc1, 0, 0, 0);
                    }
                    const int n = q0 + 16 * qt + i16, c0 = 16 * ct + 4 * g;
                    if (n < N && c0 < D) *(f32x4*)(dqb + (size_t)n * C + c0) = acc1;
                }
            }
        }
        __syncthreads();                                       // next tile staged; the exchange buffer is free again
        if (OVL) {
            if (more) {
                fa2_put<D>(Qs, qr, stg);
                fa2_put<D>(dOs, dr, stg);
                if (tid < 128) Ls[tid] = st;
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int t = trow[tt];
        if (t < Nk) {
            const size_t rows = (size_t)gridDim.z * Nk;
            float* dkr = nsplit > 1 ? kvpart + (((size_t)split * 2) * rows + (size_t)b * Nk + t) * C + h * D
                                    : dk + ((size_t)b * Nk + t) * ldg + h * D;
            float* dvr = nsplit > 1 ? dkr + rows * C : dv + ((size_t)b * Nk + t) * ldg + h * D;
#pragma unroll
            for (int ct = 0; ct < F::CT; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) {
                    *(f32x4*)(dkr + c0) = dka[ct][tt] * scale;
                    *(f32x4*)(dvr + c0) = dva[ct][tt];
                }
            }
        }
    }
}

// dq = scale * sum_kb part[kb]   (fixed order)
// (c4 = float4 per row of the partials, ldg4 = float4 per row of dq: equal unless dq is a column band of a wider buffer)
__global__ __launch_bounds__(256) void skp_fa2_dq_reduce_kernel(const float* __restrict__ part, float* __restrict__ dq, long n4,
                                                               long stride, int nkb, float scale, int c4, int ldg4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = ((const f32x4*)part)[i];
    for (int z = 1; z < nkb; ++z) acc += ((const f32x4*)(part + (size_t)z * stride))[i];
    const long o = c4 == ldg4 ? i : (i / c4) * ldg4 + (i % c4);
    ((f32x4*)dq)[o] = acc * scale;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NQT, int MINW, int OPT>
static int fa2_launch_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int N,
                          int Nk, int kvb, float scale, hipStream_t st) {
    using F = FA2<D>;
    const size_t lds = (size_t)4 * F::TILE * sizeof(float);
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fa2_fwd_kernel<D, NQT, MINW, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    dim3 grid((N + 64 * NQT - 1) / (64 * NQT), H, B), block(256);
    hipLaunchKernelGGL((skp_fa2_fwd_kernel<D, NQT, MINW, OPT>), grid, block, lds, st, q, k, v, out, lse, H, N, Nk, kvb, scale);
    return skp_launch_status();
}

template <int D>
static int fa2_launch_fwd_halves(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int N, int Nk,
                                 int kvb, float scale, hipStream_t st) {
    using F = FA2<D>;
    constexpr size_t tiles = (size_t)4 * F::TILE * sizeof(float), merge = (size_t)256 * (4 * F::CT + 4) * sizeof(float);
    const size_t lds = tiles > merge ? tiles : merge;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fa2_fwd_halves_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(skp_fa2_fwd_halves_kernel<D>, dim3((N + 63) / 64, H, B), dim3(512), lds, st, q, k, v, out, lse, H, N, Nk, kvb, scale);
    return skp_launch_status();
}

// returns -100 when this head size is not built here (the caller falls back to the first-generation kernels)
int skp_fa2_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int Bk, int H, int N,
                int Nk, int d, float scale, void* stream) {
    const int kvb = Bk == 1 ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
#define FA2_FWD(DV, NQ, W, O) return fa2_launch_fwd<DV, NQ, W, O>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st)
    // OPT bits: 2 = row sums through the idle MFMA rows (D % 16 != 0), 4 = precomputed staging offsets.  Measured
    // (profiles/r02_flash_attn.md): 64 queries per wave wins where the grid still fills the chip twice; the precomputed
    // offsets win at D = 40 / 64 and lose at D = 80.
    const bool big = (long)((N + 255) / 256) * H * B >= 512;
    switch (d) {
        case 40:
            if (!big) FA2_FWD(40, 2, 2, 6);
            FA2_FWD(40, 4, 2, 6);
        case 64: FA2_FWD(64, 2, 2, 4);                          // (64 queries per wave spills at this head size)
        case 80: FA2_FWD(80, 2, 2, 0);
        case 160:                                               // 16^2 layers (N = 256): 16 queries per wave, 32-key tiles
            // few workgroups, short key axis: the key tiles split over two wave sets of one workgroup
            if (Nk >= 4 * FA2<160>::KT && (long)((N + 63) / 64) * H * B <= 1024)
                return fa2_launch_fwd_halves<160>(q, k, v, out, lse, B, H, N, Nk, kvb, scale, st);
            FA2_FWD(160, 1, 1, 4);
        default: return -100;
    }
#undef FA2_FWD
}

// Range splits of the backward at few workgroups (1 image per rank: the 16^2 layers are 64 workgroups of the two-kernel form,
// the 32^2 layers 128 of the fused one -- a quarter / half of the CUs).  Each split must own at least two tiles.
static int fa2_two_kernel_splits(int B, int H, int N, int Nk, int d) {
    if (d != 160) return 1;                                  // (64 queries / keys per workgroup, 32-row tiles at this head size)
    const long wgs = (long)((std::max(N, Nk) + 63) / 64) * H * B;
    const int tiles = std::min(N, Nk) / 32;
    int ns = 1;
    while (ns < 4 && wgs * ns * 2 <= 256 && tiles / (ns * 2) >= 2) ns *= 2;
    return ns;
}
static int fa2_fused_splits(int B, int H, int N, int Nk, int d) {
    if (d != 80) return 1;                                   // one 120 KB workgroup per CU: a split helps only below 256 of them
    const long wgs = (long)((Nk + 127) / 128) * H * B;
    return (wgs * 2 <= 256 && N >= 256) ? 2 : 1;
}

template <int D, int NQ, int MINWQ, int NT, int MINWT, bool PRE>
static int fa2_launch_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                          const float* lse, float* dq, float* dk, float* dv, float* ws, int B, int H, int N, int Nk, int kvb,
                          float scale, int ldg, hipStream_t st) {
    using F = FA2<D>;
    const size_t lds_q = (size_t)4 * F::TILE * sizeof(float), lds_kv = (size_t)2 * (2 * F::TILE + 2 * F::KT) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fa2_bwd_dq_kernel<D, NQ, MINWQ, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_fa2_bwd_dkv_kernel<D, NT, MINWT, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    dim3 block(256);
    const int nqb = (N + 64 * NQ - 1) / (64 * NQ), nkb = (Nk + 64 * NT - 1) / (64 * NT);
    const int ns = fa2_two_kernel_splits(B, H, N, Nk, D);
    float* part = ws + (size_t)B * H * N;                   // [ns][B][N][C] dQ | [ns][2][B][Nk][C] dK, dV
    float* kvpart = part + (size_t)ns * B * N * H * D;
    hipLaunchKernelGGL((skp_fa2_bwd_dq_kernel<D, NQ, MINWQ, PRE>), dim3(nqb * ns, H, B), block, lds_q, st,
                       q, k, v, out, dout, lse, dq, ws, H, N, Nk, kvb, scale, ldg, ns, part);
    int rc = skp_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL((skp_fa2_bwd_dkv_kernel<D, NT, MINWT, PRE>), dim3(nkb * ns, H, B), block, lds_kv, st,
                       q, k, v, dout, lse, ws, dk, dv, H, N, Nk, kvb, scale, ldg, ns, kvpart);
    rc = skp_launch_status();
    if (rc || ns == 1) return rc;
    const long C = (long)H * D, nq4 = (long)B * N * C / 4, nk4 = (long)B * Nk * C / 4;
    FA2SplitReduce a;
    a.part[0] = part; a.out[0] = dq; a.n4[0] = nq4; a.sstride[0] = (long)B * N * C;
    a.part[1] = kvpart; a.out[1] = dk; a.n4[1] = nk4; a.sstride[1] = 2 * (long)B * Nk * C;
    a.part[2] = kvpart + (size_t)B * Nk * C; a.out[2] = dv; a.n4[2] = nk4; a.sstride[2] = 2 * (long)B * Nk * C;
    hipLaunchKernelGGL(skp_fa2_split_reduce_kernel, dim3((unsigned)((std::max(nq4, nk4) + 255) / 256), 3), dim3(256), 0, st, a, ns,
                       (int)(C / 4), ldg / 4);
    return skp_launch_status();
}

static bool fa2_fused_ok(int Bk, int B, int H, int N, int Nk, int d) {
    if (skp_tune(SKP_TUNE_FA2_TWO_KERNEL_BWD)) return false;     // tests: the two-kernel backward at the fused form's shapes
    // 40-wide heads: two workgroups per CU (79 KB LDS, 254 registers); 64-wide: two per CU with the dS exchange laid over the
    // Q | dO tiles and the K fragments read from LDS (70 KB, 256 registers): 0.58 -> 0.75 of peak; 80-wide: 64-query tiles on
    // eight waves (round 3): 0.42 (two kernels) -> 0.53 (48-query four-wave form, round 2) -> 0.59 at N = 1024, 0.72 at N = 4096.
    if (!((d == 40 || d == 64 || d == 80) && Bk == B && N == Nk && N >= 1024)) return false;   // the big self-attention layers
    // the single-pass form keeps ceil(Nk / 128) copies of dQ as partials: O(B N^2 C / 128) floats (13 GB at SD-2.1 768^2,
    // B = 16).  Past 2 GiB the two-kernel form (B*H*N floats of scratch) takes over.
    const long long part = (long long)((Nk + 127) / 128) * B * N * (long long)H * d * 4;
    return part <= (2ll << 30);
}

// bytes of scratch the backward needs: D = rowsum(dO * O) [B,H,N], plus the per-key-block dQ partials of the fused form
int64_t skp_fa2_bwd_workspace(int B, int Bk, int H, int N, int Nk, int d) {
    int64_t fl = (int64_t)B * H * N;
    if (fa2_fused_ok(Bk, B, H, N, Nk, d)) {
        fl += (int64_t)((Nk + 127) / 128) * B * N * H * d;
        const int ns = fa2_fused_splits(B, H, N, Nk, d);
        if (ns > 1) fl += (int64_t)ns * 2 * B * Nk * H * d;
    } else {
        const int ns = fa2_two_kernel_splits(B, H, N, Nk, d);
        if (ns > 1) fl += (int64_t)ns * B * ((int64_t)N + 2 * Nk) * H * d;
    }
    return fl * (int64_t)sizeof(float);
}

template <int D, int MINB, bool OVL, int NQ, int NW = 4>
static int fa2_launch_bwd_fused(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                const float* lse, float* dq, float* dk, float* dv, float* ws, int B, int H, int N, int Nk,
                                float scale, int ldg, hipStream_t st) {
    using X = FA2F<D, OVL, NQ>;
    const size_t lds = (size_t)X::LDS_FLOATS * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fa2_bwd_fused_kernel<D, MINB, OVL, NQ, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    float* Dbuf = ws;
    float* part = ws + (size_t)B * H * N;
    const long rows = (long)B * H * N;
    hipLaunchKernelGGL(skp_fa2_rowdot_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, out, dout, Dbuf, H, N, D, rows);
    int rc = skp_launch_status();
    if (rc) return rc;
    const int nkb = (Nk + X::KB - 1) / X::KB;
    const int ns = fa2_fused_splits(B, H, N, Nk, D);
    float* kvpart = part + (size_t)nkb * B * N * H * D;
    hipLaunchKernelGGL((skp_fa2_bwd_fused_kernel<D, MINB, OVL, NQ, NW>), dim3(nkb * ns, H, B), dim3(64 * NW), lds, st, q, k, v, dout, lse, Dbuf, part, dk, dv, H,
                       N, Nk, scale, ldg, ns, kvpart);
    rc = skp_launch_status();
    if (rc) return rc;
    const long n4 = (long)B * N * H * D / 4;
    hipLaunchKernelGGL(skp_fa2_dq_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, part, dq, n4,
                       (long)B * N * H * D, nkb, scale, H * D / 4, ldg / 4);
    rc = skp_launch_status();
    if (rc || ns == 1) return rc;
    const long nk4 = (long)B * Nk * H * D / 4;
    FA2SplitReduce a;
    a.part[0] = kvpart; a.out[0] = dk; a.n4[0] = nk4; a.sstride[0] = 2 * (long)B * Nk * H * D;
    a.part[1] = kvpart + (size_t)B * Nk * H * D; a.out[1] = dv; a.n4[1] = nk4; a.sstride[1] = 2 * (long)B * Nk * H * D;
    a.part[2] = nullptr; a.out[2] = nullptr; a.n4[2] = 0; a.sstride[2] = 0;
    hipLaunchKernelGGL(skp_fa2_split_reduce_kernel, dim3((unsigned)((nk4 + 255) / 256), 2), dim3(256), 0, st, a, ns, H * D / 4, ldg / 4);
    return skp_launch_status();
}

// workspace: skp_fa2_bwd_workspace() bytes; -100 when the head size is not built here
int skp_fa2_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                float* dq, float* dk, float* dv, float* workspace, int B, int Bk, int H, int N, int Nk, int d, float scale,
                int allow_fused, int ldg, void* stream) {
    const int kvb = Bk == 1 ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (allow_fused && fa2_fused_ok(Bk, B, H, N, Nk, d)) {
        if (d == 40) return fa2_launch_bwd_fused<40, 2, false, 4>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, Nk, scale, ldg, st);
        if (d == 64) return fa2_launch_bwd_fused<64, 2, true, 4>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, Nk, scale, ldg, st);
        // 80-wide heads: 64-query tiles on EIGHT waves (16 keys each: half the K / V fragments and dK / dV accumulators per wave,
        // 224 registers, nothing spilled, two waves per SIMD at one 120 KB workgroup per CU)
        return fa2_launch_bwd_fused<80, 1, false, 4, 8>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, Nk, scale, ldg, st);
    }
#define FA2_BWD(DV, NQ, WQ, NT, WT, PRE) \
    return fa2_launch_bwd<DV, NQ, WQ, NT, WT, PRE>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, Nk, kvb, scale, ldg, st)
    switch (d) {
        case 40: FA2_BWD(40, 2, 2, 2, 2, true);
        case 64: FA2_BWD(64, 2, 2, 1, 2, true);
        case 80: FA2_BWD(80, 2, 1, 2, 1, false);                // one wave per SIMD, 32 rows per wave (measured best)
        case 160: FA2_BWD(160, 1, 1, 1, 1, true);
        default: return -100;
    }
#undef FA2_BWD
}
