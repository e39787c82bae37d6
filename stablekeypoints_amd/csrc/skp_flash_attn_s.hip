// Flash attention FORWARD on the bf16 matrix cores with three-term operand splits (fp32 in / out, fp32 accumulate and softmax):
// the second kernel family of the round-5 experiment (skp_conv_wino4s.hip is the first).  Every fp32 operand of the two tile
// products  S^T = K.Q^T  and  O^T += V^T.P^T  is the exact sum of three bf16 terms h + m + l; the six products that matter
// run on v_mfma_f32_16x16x32_bf16, three instructions per (16 x 16 outputs, 16 contraction elements):
//       tuples  MH = [m01 h01 m23 h23]   HL = [h01 l01 h23 l23]        A.MH.B.HL + A.HL.B.MH + A.MH.B.MH
// (K = 32 = two slots of 16 contraction elements; lane (i16, kq) holds elements 4 kq .. 4 kq + 3 as two pairs per slot).
// Unlike the convolution this kernel has no operand stream that outgrows the L2 port and little VALU work per MFMA:
//   * K and V are split ONCE per call by a pre-pass (skp_fas_split_kernel) into tile images that are the LDS layout byte for
//     byte: [tile][16-key block][16-channel group][MH | HL][64 lanes x 16 bytes]; a 64-key tile is 24 KB (d = 40) and arrives
//     by LDS DMA (no staging registers, no ds_write), double-buffered;
//   * Q is split once per wave into registers; P is split in the lanes where the softmax left it -- the probability registers
//     s[kt][nt][r] (keys 16 kt + 4 g + r of query n) ARE the four contraction elements of lane (n, g): two pairs, 22 VALU
//     operations per f32x4;
//   * the row sums come out of the matrix pipe where d % 16 != 0: row d of V^T is the constant 1 (h = 1, m = l = 0: exact).
// Workgroup = 8 waves (two per SIMD: one wave's softmax / split runs beside the other's MFMAs) x 32 queries on the same K/V
// tiles.  Same contract as skp_flash_attn_fwd_f32 (lse for the fp32 backward kernels); d in {40, 80}; opt-in.
#include "skp_attn_tiles.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

template <int D>
struct FAS {
    static constexpr int G = (D + 15) / 16;                 // 16-channel groups of the QK^T contraction = 16-channel tiles of O^T
    static constexpr int KT = 64, NKT = 4;                  // keys per tile, 16-key blocks per tile
    static constexpr int MAT_B = NKT * G * 2 * 1024;        // bytes of one tile image (K or V^T): [kt][g][MH | HL][64 x 16 B]
    static constexpr int CHUNKS = MAT_B / 1024;             // 1 KB pieces (one LDS-DMA wave-instruction each)
    static constexpr bool ONES = (D % 16) != 0;             // a spare row of the last O^T tile carries the row sums
};

__device__ __forceinline__ unsigned fas_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (a0, a1, a2, a3) -> MH = [m01 h01 m23 h23], HL = [h01 l01 h23 l23]
__device__ __forceinline__ void fas_split4(const f32x4 a, f32x4& mh, f32x4& hl) {
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float x = a[2 * p], y = a[2 * p + 1];
        h[p] = fas_cvt_pk(x, y);
        const float rx = x - __builtin_bit_cast(float, h[p] << 16), ry = y - __builtin_bit_cast(float, h[p] & 0xffff0000u);
        m[p] = fas_cvt_pk(rx, ry);
        const float sx = rx - __builtin_bit_cast(float, m[p] << 16), sy = ry - __builtin_bit_cast(float, m[p] & 0xffff0000u);
        l[p] = fas_cvt_pk(sx, sy);
    }
    mh = f32x4{__builtin_bit_cast(float, m[0]), __builtin_bit_cast(float, h[0]), __builtin_bit_cast(float, m[1]), __builtin_bit_cast(float, h[1])};
    hl = f32x4{__builtin_bit_cast(float, h[0]), __builtin_bit_cast(float, l[0]), __builtin_bit_cast(float, h[1]), __builtin_bit_cast(float, l[1])};
}
__device__ __forceinline__ f32x4 fas_mfma(const f32x4& a, const f32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Tile images of K and V^T for one (batch row of k/v, head):  Kt / Vt [bk][h][tile][MAT_B bytes]
//   K  image: A operand of S^T = K.Q^T : lane (i16 = key 16 kt + i16, kq) <- channels 16 g + 4 kq .. + 3 of that key
//   V^T image: A operand of O^T += V^T.P^T : lane (i16 = channel 16 g + i16, kq) <- keys 16 kt + 4 kq .. + 3 of that channel
// one thread = one (kt, g, lane) cell of both images; keys past Nk and channels past d are zero; channel d of V^T is 1.
template <int D>
__global__ __launch_bounds__(256) void skp_fas_split_kernel(const float* __restrict__ k, const float* __restrict__ v, f32x4* __restrict__ Kt,
                                                           f32x4* __restrict__ Vt, int H, int Nk, int ntiles) {
    using F = FAS<D>;
    const int cell = blockIdx.x * 256 + threadIdx.x;             // (tile, kt, g, lane)
    const int lane = cell & 63, rest = cell >> 6;
    const int g = rest % F::G, kt = (rest / F::G) % F::NKT, tile = rest / (F::G * F::NKT);
    if (tile >= ntiles) return;
    const int h = blockIdx.y, bk = blockIdx.z, C = H * D;
    const int i16 = lane & 15, kq = lane >> 4;
    const float* kb = k + (size_t)bk * Nk * C + h * D;
    const float* vb = v + (size_t)bk * Nk * C + h * D;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    {   // K: one key, four channels
        const int t = tile * F::KT + 16 * kt + i16, c = 16 * g + 4 * kq;
        if (t < Nk && c < D) a = *(const f32x4*)(kb + (size_t)t * C + c);          // d % 4 == 0
    }
    {   // V^T: one channel, four keys
        const int c = 16 * g + i16, t0 = tile * F::KT + 16 * kt + 4 * kq;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (t0 + e < Nk) b[e] = c < D ? vb[(size_t)(t0 + e) * C + c] : (F::ONES && c == D ? 1.0f : 0.f);
    }
    f32x4 mh, hl;
    const size_t base = (((size_t)(bk * H + h) * ntiles + tile) * (F::MAT_B / 16)) + ((kt * F::G + g) * 2) * 64 + lane;
    fas_split4(a, mh, hl);
    Kt[base] = mh; Kt[base + 64] = hl;
    fas_split4(b, mh, hl);
    Vt[base] = mh; Vt[base + 64] = hl;
}

// grid (ceil(N / (128 NQT)), H, B), 512 threads; wave w owns queries [blk * 128 NQT + 16 NQT w, + 16 NQT)
template <int D, int NQT>
__global__ __launch_bounds__(512, 2) void skp_fas_fwd_kernel(const float* __restrict__ q, const float* __restrict__ Kt, const float* __restrict__ Vt,
                                                            float* __restrict__ out, float* __restrict__ lse, int H, int N, int Nk,
                                                            int ntiles, int kvb, float scale) {
    using F = FAS<D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][K image | V^T image]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int nbase = blockIdx.x * (128 * NQT) + wave * (16 * NQT);
    const float sl2 = scale * SKP_LOG2E;
    const size_t img = (size_t)((kvb ? b : 0) * H + h) * ntiles * F::MAT_B;
    const i32x4 krs = skp_make_rsrc((const char*)Kt + img, (unsigned)((size_t)ntiles * F::MAT_B));
    const i32x4 vrs = skp_make_rsrc((const char*)Vt + img, (unsigned)((size_t)ntiles * F::MAT_B));

    // Q tuples of this wave's 2 x 16 queries, pre-scaled
    f32x4 qmh[NQT][F::G], qhl[NQT][F::G];
    int nrow[NQT];
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nbase + 16 * nt + i16;
        nrow[nt] = n;
        const float* qrow = q + ((size_t)b * N + (n < N ? n : N - 1)) * C + h * D;
#pragma unroll
        for (int gg = 0; gg < F::G; ++gg) {
            const int c = 16 * gg + 4 * g;
            const f32x4 x = c < D ? *(const f32x4*)(qrow + c) * sl2 : f32x4{0.f, 0.f, 0.f, 0.f};
            fas_split4(x, qmh[nt][gg], qhl[nt][gg]);
        }
    }
    f32x4 o[F::G][NQT];
#pragma unroll
    for (int ct = 0; ct < F::G; ++ct)
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) o[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[NQT], lpart[NQT];
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) { mrun[nt] = -INFINITY; lpart[nt] = 0.f; }
    constexpr int LCT = D / 16, LG = (D % 16) / 4, LR = D % 4;   // where row d of O^T lives: tile, lane group, register

    // LDS DMA of one tile (both images): 1 KB pieces, piece p of a matrix by wave p % 8
    auto stage = [&](int tile, int buf) {
        unsigned char* dst = smem + buf * 2 * F::MAT_B;
#pragma unroll
        for (int j = 0; j < (F::CHUNKS + 7) / 8; ++j) {
            const int p = j * 8 + wave;
            if (p < F::CHUNKS) {
                skp_buf_load_lds(krs, (skp_lds_ptr)(dst + p * 1024), 16, lane * 16, tile * F::MAT_B + p * 1024, 0, 0);
                skp_buf_load_lds(vrs, (skp_lds_ptr)(dst + F::MAT_B + p * 1024), 16, lane * 16, tile * F::MAT_B + p * 1024, 0, 0);
            }
        }
    };
    stage(0, 0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile `tile` have landed ...
        __syncthreads();                                          // ... and everybody's; nobody reads the other buffer any more
        if (tile + 1 < ntiles) stage(tile + 1, cur ^ 1);
        const unsigned char* Ks = smem + cur * 2 * F::MAT_B + lane * 16;
        const unsigned char* Vs = Ks + F::MAT_B;
        const int kt0 = tile * F::KT;

        f32x4 s[F::NKT][NQT];
#pragma unroll
        for (int kt = 0; kt < F::NKT; ++kt) {
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) s[kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gg = 0; gg < F::G; ++gg) {
                const f32x4 kmh = *(const f32x4*)(Ks + ((kt * F::G + gg) * 2) * 1024);
                const f32x4 khl = *(const f32x4*)(Ks + ((kt * F::G + gg) * 2 + 1) * 1024);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) s[kt][nt] = fas_mfma(kmh, qhl[nt][gg], s[kt][nt]);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) s[kt][nt] = fas_mfma(khl, qmh[nt][gg], s[kt][nt]);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) s[kt][nt] = fas_mfma(kmh, qmh[nt][gg], s[kt][nt]);
            }
        }
        if (kt0 + F::KT > Nk) {                                   // ragged last tile (uniform branch)
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
        for (int nt = 0; nt < NQT; ++nt) {
            float tm = fmaxf(fmaxf(s[0][nt][0], s[0][nt][1]), fmaxf(s[0][nt][2], s[0][nt][3]));
#pragma unroll
            for (int kt = 1; kt < F::NKT; ++kt) tm = fmaxf(tm, fmaxf(fmaxf(s[kt][nt][0], s[kt][nt][1]), fmaxf(s[kt][nt][2], s[kt][nt][3])));
            tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float mn = fmaxf(mrun[nt], tm);
            const float alpha = __builtin_amdgcn_exp2f(mrun[nt] - mn);          // first tile: exp2(-inf) = 0
            mrun[nt] = mn;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < F::NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][nt][r] = __builtin_amdgcn_exp2f(s[kt][nt][r] - mn);
                    if (!F::ONES) rs += s[kt][nt][r];
                }
            if (!F::ONES) lpart[nt] = lpart[nt] * alpha + rs;
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) o[ct][nt] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < F::NKT; ++kt) {
            f32x4 pmh[NQT], phl[NQT];
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) fas_split4(s[kt][nt], pmh[nt], phl[nt]);
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) {
                const f32x4 vmh = *(const f32x4*)(Vs + ((kt * F::G + ct) * 2) * 1024);
                const f32x4 vhl = *(const f32x4*)(Vs + ((kt * F::G + ct) * 2 + 1) * 1024);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) o[ct][nt] = fas_mfma(vmh, phl[nt], o[ct][nt]);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) o[ct][nt] = fas_mfma(vhl, pmh[nt], o[ct][nt]);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) o[ct][nt] = fas_mfma(vmh, pmh[nt], o[ct][nt]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        float l;
        if (F::ONES) l = __shfl(o[LCT][nt][LR], 16 * LG + i16, 64);          // row d of O^T = sum_t P[n][t]
        else { l = lpart[nt]; l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64); }
        const float inv = 1.0f / l;
        const int n = nrow[nt];
        if (n < N) {
            float* orow = out + ((size_t)b * N + n) * C + h * D;
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) *(f32x4*)(orow + c0) = o[ct][nt] * inv;
            }
            if (g == 0) lse[((size_t)b * H + h) * N + n] = (mrun[nt] + __builtin_amdgcn_logf(l)) * SKP_LN2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward (self-attention, Bk == B), two kernels on the same tuples:
//   dQ kernel   (lane = query, as the forward): per 64-key tile  S^T = K.Q^T, P = exp2(S - lse), dP^T = V.dO^T,
//               dS = P (dP - D), dQ^T += K^T.dS^T;  also writes D[n] = rowsum(dO * O) for the second kernel.
//               Images per key tile: K and V row-style (lane (key, kq) <- 4 channels), K^T (lane (channel, kq) <- 4 keys).
//   dK/dV kernel (lane = key): per 32-query tile  S = Q.K^T, dP = dO.V^T, dV^T += dO^T.P, dK^T += Q^T.dS.
//               Images per query tile: Q (pre-scaled) and dO row-style, Q^T and dO^T.
// The score registers are the B operand of the accumulating products as they are (k-slot g <-> row 4 g + r).
// ---------------------------------------------------------------------------------------------------------------------
// Row-style and transposed images of X[rows][D] per (batch row, head), NB 16-row blocks per tile; x * mult is what is split.
template <int D, int NB>
__global__ __launch_bounds__(256) void skp_fas_images_kernel(const float* __restrict__ x, f32x4* __restrict__ Xr, f32x4* __restrict__ Xt, int H,
                                                            int Nr, int ntiles, float mult) {
    using F = FAS<D>;
    constexpr int MAT16 = NB * F::G * 2 * 64;                   // f32x4 per tile image
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int lane = cell & 63, rest = cell >> 6;
    const int g = rest % F::G, nb = (rest / F::G) % NB, tile = rest / (F::G * NB);
    if (tile >= ntiles) return;
    const int h = blockIdx.y, b = blockIdx.z, C = H * D;
    const int i16 = lane & 15, kq = lane >> 4;
    const float* xb = x + (size_t)b * Nr * C + h * D;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, t4 = {0.f, 0.f, 0.f, 0.f};
    {
        const int r = tile * 16 * NB + 16 * nb + i16, c = 16 * g + 4 * kq;
        if (r < Nr && c < D) a = *(const f32x4*)(xb + (size_t)r * C + c) * mult;
    }
    {
        const int c = 16 * g + i16, r0 = tile * 16 * NB + 16 * nb + 4 * kq;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (r0 + e < Nr && c < D) t4[e] = xb[(size_t)(r0 + e) * C + c] * mult;
    }
    f32x4 mh, hl;
    const size_t base = ((size_t)(b * H + h) * ntiles + tile) * MAT16 + ((nb * F::G + g) * 2) * 64 + lane;
    fas_split4(a, mh, hl);
    Xr[base] = mh; Xr[base + 64] = hl;
    fas_split4(t4, mh, hl);
    Xt[base] = mh; Xt[base + 64] = hl;
}

// grid (ceil(N / 256), H, B), 512 threads; wave w owns queries [blk * 256 + 32 w, + 32)
template <int D>
__global__ __launch_bounds__(512, 2) void skp_fas_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ o, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ Kr,
                                                               const float* __restrict__ Vr, const float* __restrict__ Kt,
                                                               float* __restrict__ dq, float* __restrict__ Dn, int H, int N, int Nk,
                                                               int ntiles, float scale, int ldg) {
    using F = FAS<D>;
    constexpr int NQT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][K | V | K^T images]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int nbase = blockIdx.x * 256 + wave * (16 * NQT);
    const float sl2 = scale * SKP_LOG2E;
    const size_t img = (size_t)(b * H + h) * ntiles * F::MAT_B;
    const i32x4 krs = skp_make_rsrc((const char*)Kr + img, (unsigned)((size_t)ntiles * F::MAT_B));
    const i32x4 vrs = skp_make_rsrc((const char*)Vr + img, (unsigned)((size_t)ntiles * F::MAT_B));
    const i32x4 trs = skp_make_rsrc((const char*)Kt + img, (unsigned)((size_t)ntiles * F::MAT_B));

    f32x4 qmh[NQT][F::G], qhl[NQT][F::G], gmh[NQT][F::G], ghl[NQT][F::G];
    float lse2[NQT], dsum[NQT];
    int nrow[NQT];
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nbase + 16 * nt + i16;
        nrow[nt] = n;
        const size_t ro = ((size_t)b * N + (n < N ? n : N - 1)) * C + h * D;
        float dpart = 0.f;
#pragma unroll
        for (int gg = 0; gg < F::G; ++gg) {
            const int c = 16 * gg + 4 * g;
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f}, ov = {0.f, 0.f, 0.f, 0.f};
            if (c < D) { x = *(const f32x4*)(q + ro + c) * sl2; gv = *(const f32x4*)(dout + ro + c); ov = *(const f32x4*)(o + ro + c); }
            fas_split4(x, qmh[nt][gg], qhl[nt][gg]);
            fas_split4(gv, gmh[nt][gg], ghl[nt][gg]);
            dpart += gv[0] * ov[0] + gv[1] * ov[1] + gv[2] * ov[2] + gv[3] * ov[3];
        }
        dpart += __shfl_xor(dpart, 16, 64);
        dpart += __shfl_xor(dpart, 32, 64);
        dsum[nt] = dpart;
        lse2[nt] = lse[((size_t)b * H + h) * N + (n < N ? n : N - 1)] * SKP_LOG2E;
        if (g == 0 && n < N) Dn[((size_t)b * H + h) * N + n] = dpart;
    }
    f32x4 acc[F::G][NQT];
#pragma unroll
    for (int ct = 0; ct < F::G; ++ct)
#pragma unroll
        for (int nt = 0; nt < NQT; ++nt) acc[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int tile, int buf) {
        unsigned char* dst = smem + buf * 3 * F::MAT_B;
#pragma unroll
        for (int j = 0; j < (F::CHUNKS + 7) / 8; ++j) {
            const int p = j * 8 + wave;
            if (p < F::CHUNKS) {
                skp_buf_load_lds(krs, (skp_lds_ptr)(dst + p * 1024), 16, lane * 16, tile * F::MAT_B + p * 1024, 0, 0);
                skp_buf_load_lds(vrs, (skp_lds_ptr)(dst + F::MAT_B + p * 1024), 16, lane * 16, tile * F::MAT_B + p * 1024, 0, 0);
                skp_buf_load_lds(trs, (skp_lds_ptr)(dst + 2 * F::MAT_B + p * 1024), 16, lane * 16, tile * F::MAT_B + p * 1024, 0, 0);
            }
        }
    };
    stage(0, 0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + 1 < ntiles) stage(tile + 1, cur ^ 1);
        const unsigned char* Ks = smem + cur * 3 * F::MAT_B + lane * 16;
        const unsigned char* Vs = Ks + F::MAT_B;
        const unsigned char* Ts = Ks + 2 * F::MAT_B;
        const int left = Nk - tile * F::KT;
#pragma unroll
        for (int kt = 0; kt < F::NKT; ++kt) {
            f32x4 s[NQT], dp[NQT];
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) { s[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int gg = 0; gg < F::G; ++gg) {
                const f32x4 kmh = *(const f32x4*)(Ks + ((kt * F::G + gg) * 2) * 1024);
                const f32x4 khl = *(const f32x4*)(Ks + ((kt * F::G + gg) * 2 + 1) * 1024);
                const f32x4 vmh = *(const f32x4*)(Vs + ((kt * F::G + gg) * 2) * 1024);
                const f32x4 vhl = *(const f32x4*)(Vs + ((kt * F::G + gg) * 2 + 1) * 1024);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) { s[nt] = fas_mfma(kmh, qhl[nt][gg], s[nt]); dp[nt] = fas_mfma(vmh, ghl[nt][gg], dp[nt]); }
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) { s[nt] = fas_mfma(khl, qmh[nt][gg], s[nt]); dp[nt] = fas_mfma(vhl, gmh[nt][gg], dp[nt]); }
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) { s[nt] = fas_mfma(kmh, qmh[nt][gg], s[nt]); dp[nt] = fas_mfma(vmh, gmh[nt][gg], dp[nt]); }
            }
            f32x4 dmh[NQT], dhl[NQT];
#pragma unroll
            for (int nt = 0; nt < NQT; ++nt) {
                f32x4 ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (16 * kt + 4 * g + r < left) ? __builtin_amdgcn_exp2f(s[nt][r] - lse2[nt]) : 0.f;
                    ds[r] = p * (dp[nt][r] - dsum[nt]);
                }
                fas_split4(ds, dmh[nt], dhl[nt]);
            }
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) {
                const f32x4 tmh = *(const f32x4*)(Ts + ((kt * F::G + ct) * 2) * 1024);
                const f32x4 thl = *(const f32x4*)(Ts + ((kt * F::G + ct) * 2 + 1) * 1024);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) acc[ct][nt] = fas_mfma(tmh, dhl[nt], acc[ct][nt]);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) acc[ct][nt] = fas_mfma(thl, dmh[nt], acc[ct][nt]);
#pragma unroll
                for (int nt = 0; nt < NQT; ++nt) acc[ct][nt] = fas_mfma(tmh, dmh[nt], acc[ct][nt]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NQT; ++nt) {
        const int n = nrow[nt];
        if (n < N) {
            float* drow = dq + ((size_t)b * N + n) * ldg + h * D;
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) *(f32x4*)(drow + c0) = acc[ct][nt] * scale;
            }
        }
    }
}

// grid (ceil(Nk / 256), H, B), 512 threads; wave w owns keys [blk * 256 + 32 w, + 32); query tiles of 32
template <int D>
__global__ __launch_bounds__(512, 2) void skp_fas_bwd_dkv_kernel(const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ lse,
                                                                const float* __restrict__ Dn, const float* __restrict__ Qr,
                                                                const float* __restrict__ Gr, const float* __restrict__ Qt,
                                                                const float* __restrict__ Gt, float* __restrict__ dk, float* __restrict__ dv,
                                                                int H, int N, int Nk, int nqtiles, float scale, int ldg) {
    using F = FAS<D>;
    constexpr int NKB = 2, NB = 2;                              // 16-key blocks per wave, 16-query blocks per tile
    constexpr int QMAT = NB * F::G * 2 * 1024;                  // bytes of one query-tile image
    constexpr int QCH = QMAT / 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][Q | dO | Q^T | dO^T images]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int tbase = blockIdx.x * 256 + wave * (16 * NKB);
    const size_t img = (size_t)(b * H + h) * nqtiles * QMAT;
    const i32x4 qrs = skp_make_rsrc((const char*)Qr + img, (unsigned)((size_t)nqtiles * QMAT));
    const i32x4 grs = skp_make_rsrc((const char*)Gr + img, (unsigned)((size_t)nqtiles * QMAT));
    const i32x4 qts = skp_make_rsrc((const char*)Qt + img, (unsigned)((size_t)nqtiles * QMAT));
    const i32x4 gts = skp_make_rsrc((const char*)Gt + img, (unsigned)((size_t)nqtiles * QMAT));
    const float* lrow = lse + ((size_t)b * H + h) * N;
    const float* drow = Dn + ((size_t)b * H + h) * N;

    // this wave's keys as B operands (lane (key, g) <- channels 16 gg + 4 g ..)
    f32x4 kmh[NKB][F::G], khl[NKB][F::G], vmh[NKB][F::G], vhl[NKB][F::G];
    int trow[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int t = tbase + 16 * kb + i16;
        trow[kb] = t;
        const size_t ro = ((size_t)b * Nk + (t < Nk ? t : Nk - 1)) * C + h * D;
#pragma unroll
        for (int gg = 0; gg < F::G; ++gg) {
            const int c = 16 * gg + 4 * g;
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
            if (c < D && t < Nk) { x = *(const f32x4*)(k + ro + c); y = *(const f32x4*)(v + ro + c); }
            fas_split4(x, kmh[kb][gg], khl[kb][gg]);
            fas_split4(y, vmh[kb][gg], vhl[kb][gg]);
        }
    }
    f32x4 ak[F::G][NKB], av[F::G][NKB];
#pragma unroll
    for (int ct = 0; ct < F::G; ++ct)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { ak[ct][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; av[ct][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    auto stage = [&](int tile, int buf) {
        unsigned char* dst = smem + buf * 4 * QMAT;
#pragma unroll
        for (int j = 0; j < (QCH + 7) / 8; ++j) {
            const int p = j * 8 + wave;
            if (p < QCH) {
                skp_buf_load_lds(qrs, (skp_lds_ptr)(dst + p * 1024), 16, lane * 16, tile * QMAT + p * 1024, 0, 0);
                skp_buf_load_lds(grs, (skp_lds_ptr)(dst + QMAT + p * 1024), 16, lane * 16, tile * QMAT + p * 1024, 0, 0);
                skp_buf_load_lds(qts, (skp_lds_ptr)(dst + 2 * QMAT + p * 1024), 16, lane * 16, tile * QMAT + p * 1024, 0, 0);
                skp_buf_load_lds(gts, (skp_lds_ptr)(dst + 3 * QMAT + p * 1024), 16, lane * 16, tile * QMAT + p * 1024, 0, 0);
            }
        }
    };
    stage(0, 0);
    for (int tile = 0; tile < nqtiles; ++tile) {
        const int cur = tile & 1;
        // lse / D of this tile's queries: rows 16 nb + 4 g + r of the score blocks
        f32x4 l2[NB], dn[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n0 = tile * 16 * NB + 16 * nb + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + r;
                l2[nb][r] = n < N ? lrow[n] * SKP_LOG2E : INFINITY;      // rows past N: P = exp2(-inf) = 0
                dn[nb][r] = n < N ? drow[n] : 0.f;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + 1 < nqtiles) stage(tile + 1, cur ^ 1);
        const unsigned char* Qs = smem + cur * 4 * QMAT + lane * 16;
        const unsigned char* Gs = Qs + QMAT;
        const unsigned char* Qts = Qs + 2 * QMAT;
        const unsigned char* Gts = Qs + 3 * QMAT;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 s[NKB], dp[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) { s[kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int gg = 0; gg < F::G; ++gg) {
                const f32x4 qm = *(const f32x4*)(Qs + ((nb * F::G + gg) * 2) * 1024);
                const f32x4 qh = *(const f32x4*)(Qs + ((nb * F::G + gg) * 2 + 1) * 1024);
                const f32x4 gm = *(const f32x4*)(Gs + ((nb * F::G + gg) * 2) * 1024);
                const f32x4 gh = *(const f32x4*)(Gs + ((nb * F::G + gg) * 2 + 1) * 1024);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { s[kb] = fas_mfma(qm, khl[kb][gg], s[kb]); dp[kb] = fas_mfma(gm, vhl[kb][gg], dp[kb]); }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { s[kb] = fas_mfma(qh, kmh[kb][gg], s[kb]); dp[kb] = fas_mfma(gh, vmh[kb][gg], dp[kb]); }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { s[kb] = fas_mfma(qm, kmh[kb][gg], s[kb]); dp[kb] = fas_mfma(gm, vmh[kb][gg], dp[kb]); }
            }
            f32x4 pmh[NKB], phl[NKB], dmh[NKB], dhl[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(s[kb][r] - l2[nb][r]);
                    ds[r] = p[r] * (dp[kb][r] - dn[nb][r]);
                }
                fas_split4(p, pmh[kb], phl[kb]);
                fas_split4(ds, dmh[kb], dhl[kb]);
            }
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) {
                const f32x4 tm = *(const f32x4*)(Qts + ((nb * F::G + ct) * 2) * 1024);
                const f32x4 th = *(const f32x4*)(Qts + ((nb * F::G + ct) * 2 + 1) * 1024);
                const f32x4 um = *(const f32x4*)(Gts + ((nb * F::G + ct) * 2) * 1024);
                const f32x4 uh = *(const f32x4*)(Gts + ((nb * F::G + ct) * 2 + 1) * 1024);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { ak[ct][kb] = fas_mfma(tm, dhl[kb], ak[ct][kb]); av[ct][kb] = fas_mfma(um, phl[kb], av[ct][kb]); }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { ak[ct][kb] = fas_mfma(th, dmh[kb], ak[ct][kb]); av[ct][kb] = fas_mfma(uh, pmh[kb], av[ct][kb]); }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) { ak[ct][kb] = fas_mfma(tm, dmh[kb], ak[ct][kb]); av[ct][kb] = fas_mfma(um, pmh[kb], av[ct][kb]); }
            }
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int t = trow[kb];
        if (t < Nk) {
            float* krow = dk + ((size_t)b * Nk + t) * ldg + h * D;
            float* vrow = dv + ((size_t)b * Nk + t) * ldg + h * D;
#pragma unroll
            for (int ct = 0; ct < F::G; ++ct) {
                const int c0 = 16 * ct + 4 * g;
                if (c0 < D) {
                    *(f32x4*)(krow + c0) = ak[ct][kb] * SKP_LN2;        // the Q images carry scale * log2(e)
                    *(f32x4*)(vrow + c0) = av[ct][kb];
                }
            }
        }
    }
}

template <int D>
static int fas_bwd_run(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse, float* dq,
                       float* dk, float* dv, void* workspace, int B, int H, int N, float scale, int ldg, hipStream_t st) {
    using F = FAS<D>;
    const int ntiles = (N + F::KT - 1) / F::KT, nqtiles = (N + 31) / 32;
    const size_t kimg = (size_t)B * H * ntiles * F::MAT_B;                  // 64-row tiles: K, V (row), K^T
    const size_t qimg = (size_t)B * H * nqtiles * (F::MAT_B / 2);           // 32-row tiles: Q, dO (row), Q^T, dO^T
    char* w = (char*)workspace;
    f32x4 *Kr = (f32x4*)w, *Kt = (f32x4*)(w + kimg), *Vr = (f32x4*)(w + 2 * kimg), *Vt = (f32x4*)(w + 3 * kimg);
    f32x4 *Qr = (f32x4*)(w + 4 * kimg), *Qt = (f32x4*)(w + 4 * kimg + qimg), *Gr = (f32x4*)(w + 4 * kimg + 2 * qimg),
          *Gt = (f32x4*)(w + 4 * kimg + 3 * qimg);
    float* Dn = (float*)(w + 4 * kimg + 4 * qimg);
    const int kcells = ntiles * F::NKT * F::G * 64, qcells = nqtiles * 2 * F::G * 64;
    hipLaunchKernelGGL((skp_fas_images_kernel<D, 4>), dim3((kcells + 255) / 256, H, B), dim3(256), 0, st, k, Kr, Kt, H, N, ntiles, 1.0f);
    hipLaunchKernelGGL((skp_fas_images_kernel<D, 4>), dim3((kcells + 255) / 256, H, B), dim3(256), 0, st, v, Vr, Vt, H, N, ntiles, 1.0f);
    hipLaunchKernelGGL((skp_fas_images_kernel<D, 2>), dim3((qcells + 255) / 256, H, B), dim3(256), 0, st, q, Qr, Qt, H, N, nqtiles, scale * SKP_LOG2E);
    hipLaunchKernelGGL((skp_fas_images_kernel<D, 2>), dim3((qcells + 255) / 256, H, B), dim3(256), 0, st, dout, Gr, Gt, H, N, nqtiles, 1.0f);
    const size_t lds_q = (size_t)6 * F::MAT_B, lds_k = (size_t)8 * (F::MAT_B / 2);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fas_bwd_dq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_fas_bwd_dkv_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(skp_fas_bwd_dq_kernel<D>, dim3((N + 255) / 256, H, B), dim3(512), lds_q, st, q, o, dout, lse, (const float*)Kr,
                       (const float*)Vr, (const float*)Kt, dq, Dn, H, N, N, ntiles, scale, ldg);
    hipLaunchKernelGGL(skp_fas_bwd_dkv_kernel<D>, dim3((N + 255) / 256, H, B), dim3(512), lds_k, st, k, v, lse, (const float*)Dn,
                       (const float*)Qr, (const float*)Gr, (const float*)Qt, (const float*)Gt, dk, dv, H, N, N, nqtiles, scale, ldg);
    return skp_launch_status();
}

template <int D, int NQT>
static int fas_run(const float* q, const float* k, const float* v, float* out, float* lse, void* workspace, int B, int Bk, int H, int N,
                   int Nk, float scale, hipStream_t st) {
    using F = FAS<D>;
    const int ntiles = (Nk + F::KT - 1) / F::KT;
    const size_t img = (size_t)Bk * H * ntiles * F::MAT_B;
    f32x4* Kt = (f32x4*)workspace;
    f32x4* Vt = (f32x4*)((char*)workspace + img);
    const int cells = ntiles * F::NKT * F::G * 64;
    hipLaunchKernelGGL(skp_fas_split_kernel<D>, dim3((cells + 255) / 256, H, Bk), dim3(256), 0, st, k, v, Kt, Vt, H, Nk, ntiles);
    const size_t lds = (size_t)4 * F::MAT_B;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_fas_fwd_kernel<D, NQT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((skp_fas_fwd_kernel<D, NQT>), dim3((N + 128 * NQT - 1) / (128 * NQT), H, B), dim3(512), lds, st, q, (const float*)Kt, (const float*)Vt, out, lse,
                       H, N, Nk, ntiles, Bk == B && B > 1 ? 1 : 0, scale);
    return skp_launch_status();
}

}  // namespace

// 1 where the split forward runs: d in {40, 80}, 32-bit offsets of the tile images.
extern "C" int skp_flash_attn_fwd_split_ok(int B, int Bk, int H, int N, int Nk, int d) {
    if (B <= 0 || H <= 0 || N <= 0 || Nk <= 0 || (Bk != 1 && Bk != B)) return 0;
    if (d != 40 && d != 80) return 0;
    const long long ntiles = (Nk + 63) / 64, mat = 4ll * ((d + 15) / 16) * 2 * 1024;
    return ntiles * mat < 0x7fffffffll ? 1 : 0;
}

// bytes of the K / V^T tile images (2.4x the fp32 tensors at d = 40)
extern "C" int64_t skp_flash_attn_fwd_split_workspace(int B, int Bk, int H, int N, int Nk, int d) {
    if (!skp_flash_attn_fwd_split_ok(B, Bk, H, N, Nk, d)) return 0;
    const int64_t ntiles = (Nk + 63) / 64, mat = (int64_t)4 * ((d + 15) / 16) * 2 * 1024;
    return 2 * (int64_t)Bk * H * ntiles * mat;
}

// out / lse as skp_flash_attn_fwd_f32; the two tile products on the bf16 matrix cores with three-term operand splits.
extern "C" int skp_flash_attn_fwd_split_f32(const float* q, const float* k, const float* v, float* out, float* lse, void* workspace,
                                            int B, int Bk, int H, int N, int Nk, int d, float scale, void* stream) {
    if (!q || !k || !v || !out || !lse || !workspace) return SKP_E_BADARG;
    if (B <= 0 || H <= 0 || N <= 0 || Nk <= 0 || (Bk != 1 && Bk != B)) return SKP_E_BADARG;
    if (!skp_flash_attn_fwd_split_ok(B, Bk, H, N, Nk, d)) return SKP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    // (64 queries per wave measured no faster at d = 40 and spills: 32 it is); 16 per wave where 32 would leave CUs without a
    // workgroup (the 32^2 layers of a 1- or 2-image step: 64 workgroups of 256 queries)
    const bool small = (long)((N + 255) / 256) * H * B < 256;
    if (d == 40) return small ? fas_run<40, 1>(q, k, v, out, lse, workspace, B, Bk, H, N, Nk, scale, st)
                              : fas_run<40, 2>(q, k, v, out, lse, workspace, B, Bk, H, N, Nk, scale, st);
    return small ? fas_run<80, 1>(q, k, v, out, lse, workspace, B, Bk, H, N, Nk, scale, st)
                 : fas_run<80, 2>(q, k, v, out, lse, workspace, B, Bk, H, N, Nk, scale, st);
}

// ---- backward (self-attention: Bk == B, Nk == N), d = 40 ----
extern "C" int skp_flash_attn_bwd_split_ok(int B, int Bk, int H, int N, int Nk, int d) {
    if (B <= 0 || H <= 0 || N <= 0 || Bk != B || Nk != N || d != 40) return 0;
    const long long ntiles = (N + 63) / 64, mat = 4ll * ((d + 15) / 16) * 2 * 1024;
    return ntiles * mat < 0x7fffffffll ? 1 : 0;
}
// bytes: four 64-row tile images of K / V, four 32-row tile images of Q / dO, D [B,H,N]
extern "C" int64_t skp_flash_attn_bwd_split_workspace(int B, int Bk, int H, int N, int Nk, int d) {
    if (!skp_flash_attn_bwd_split_ok(B, Bk, H, N, Nk, d)) return 0;
    const int64_t mat = (int64_t)4 * ((d + 15) / 16) * 2 * 1024;
    return 4 * (int64_t)B * H * ((N + 63) / 64) * mat + 4 * (int64_t)B * H * ((N + 31) / 32) * (mat / 2) + (int64_t)B * H * N * 4;
}
// dq, dk, dv as skp_flash_attn_bwd_f32 (out / lse from either forward); five tile products on the bf16 matrix cores.
extern "C" int skp_flash_attn_bwd_split_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                            const float* lse, float* dq, float* dk, float* dv, void* workspace, int B, int Bk, int H,
                                            int N, int Nk, int d, float scale, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !workspace) return SKP_E_BADARG;
    if (!skp_flash_attn_bwd_split_ok(B, Bk, H, N, Nk, d)) return SKP_E_RANGE;
    return fas_bwd_run<40>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, scale, H * d, (hipStream_t)stream);
}
// the same with dq, dk, dv as column bands of wider row-major buffers (row stride ldg floats >= H * d, multiple of 4): see
// skp_flash_attn_bwd_ld_f32
extern "C" int skp_flash_attn_bwd_split_ld_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                               const float* lse, float* dq, float* dk, float* dv, void* workspace, int B, int Bk, int H,
                                               int N, int Nk, int d, float scale, int ldg, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !workspace) return SKP_E_BADARG;
    if (ldg < H * d || (ldg & 3)) return SKP_E_BADARG;
    if (!skp_flash_attn_bwd_split_ok(B, Bk, H, N, Nk, d)) return SKP_E_RANGE;
    return fas_bwd_run<40>(q, k, v, out, dout, lse, dq, dk, dv, workspace, B, H, N, scale, ldg, (hipStream_t)stream);
}
