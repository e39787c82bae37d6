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

// grid (ceil(N / 256), H, B), 512 threads; wave w owns queries [blk * 256 + 32 w, + 32)
template <int D>
__global__ __launch_bounds__(512, 2) void skp_fas_fwd_kernel(const float* __restrict__ q, const float* __restrict__ Kt, const float* __restrict__ Vt,
                                                            float* __restrict__ out, float* __restrict__ lse, int H, int N, int Nk,
                                                            int ntiles, int kvb, float scale) {
    using F = FAS<D>;
    constexpr int NQT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 buffers][K image | V^T image]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, C = H * D;
    const int nbase = blockIdx.x * 256 + wave * (16 * NQT);
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

template <int D>
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
        hipError_t e = hipFuncSetAttribute((const void*)skp_fas_fwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(skp_fas_fwd_kernel<D>, dim3((N + 255) / 256, H, B), dim3(512), lds, st, q, (const float*)Kt, (const float*)Vt, out, lse,
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
    if (d == 40) return fas_run<40>(q, k, v, out, lse, workspace, B, Bk, H, N, Nk, scale, st);
    return fas_run<80>(q, k, v, out, lse, workspace, B, Bk, H, N, Nk, scale, st);
}
