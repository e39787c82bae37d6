// 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4,3x3) on the BF16 matrix cores with three-term operand splits:
// fp32 in, fp32 out, fp32 accumulate, every fp32 operand of the transform-domain products represented EXACTLY as the sum of
// three bf16 terms  x = h + m + l  (round-to-nearest at each step, 8 + 8 + 8 significand bits) and the six products that
// matter  h.h + h.m + m.h + h.l + l.h + m.m  evaluated on v_mfma_f32_16x16x32_bf16 (the dropped m.l, l.m, l.l are below
// 2^-24 of |a||b|).  bf16 keeps the fp32 exponent range, so the result is as accurate as the fp32-instruction kernels of
// skp_conv_wino4.hip at any input scale (measured: 0.4-0.6x their error against fp64, profiles/r05_conv_split.md) while the
// matrix pipe does 6 / 16 of the work: v_mfma_f32_16x16x4_f32 retires 32 MACs per cycle and SIMD, the bf16 form 512.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          6x6 transform domain, as skp_conv_wino4.hip
//
// What bounds the kernel once the matrix pipe is 2.7x faster: the transformed filter streams from L2 at
// 6 bytes per (cin, cout, position) and a CU's L2 port moves 64 bytes per clock, so a workgroup must reuse every filter byte on
// >= 32 tiles; the 36 x (channels x tiles) fp32 accumulators bound channels x tiles per CU at 2048.  Hence
//   workgroup = 64 output channels x 32 tiles, wave = 16 channels x 32 tiles x 36 positions (288 accumulator registers),
// i.e. the geometry of skp_wino4_conv_kernel, not of the 128-channel form.
//
// Operand packing.  One MFMA contracts K = 32 = two slots of 16 input channels; lane (i16, kq) of an operand holds channels
// 4 kq .. 4 kq + 3, as channel PAIRS in dwords: registers (0, 2) carry the first slot's pairs (c0 c1), (c2 c3), registers (1, 3)
// the second slot's (any K order is fine as long as both operands use it).  Two operand tuples per side give all six products
// in three instructions:      MH = [m01 h01 m23 h23]      HL = [h01 l01 h23 l23]
//       A.MH . B.HL = m.h + h.l        A.HL . B.MH = h.m + l.h        A.MH . B.MH = m.m + h.h
// Filter in memory:  Us[cin / 16][cout / 16][p][ 64 lanes x (m01 h01 m23 h23) | 64 lanes x (l01 l23) ]   (24 bytes per lane and position:
//                    MH is one fully coalesced 16-byte load, HL is built from it and the 8-byte l load by four register moves; the
//                    36 positions of a wave's (stage, 16 channels) block are one contiguous 54 KB stream)
// Input in LDS:      [half][18 positions][MH | HL][kq][32 tiles][16 bytes]   (h is stored twice: both tuples are single aligned
//                    16-byte reads, and the thread that produced a channel pair writes (m, h) and (h, l) as two 8-byte stores)
//
// LDS.  A 16-channel stage of 32 tiles is 36 x 4 x 32 x 16 x 2 = 144 KB: no room for two.  The stage is double-buffered by
// POSITION HALVES instead: rows 0-2 of the 6x6 tile (positions 0-17) live in half A, rows 3-5 in half B.  While the MFMAs walk
// half A of stage s the side jobs write rows 3-5 of stage s (the patch is still in registers) into half B and request the patches
// of stage s + 1; while they walk half B, the side jobs run the column pass of stage s + 1 and write its rows 0-2 into half A.
// One barrier per half.  The input transform + split is fp32 VALU work of the same single wave per SIMD (per stage and thread:
// 2 patches, ~190 transform + ~400 split + 108 LDS stores against 216 MFMAs = 3456 matrix cycles): it does not hide, the
// kernel is VALU-issue bound at ~5.5k cycles per stage -- still 2x the fp32 form's 11k for the same products.
#include <algorithm>
#include <type_traits>
#include <utility>
#include "skp_common.h"
#include "skp_wino4_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned short w4s_bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ float w4s_bf16_float(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// ---- filter transform + split: Us[c16][cb][p][ 64 lanes x (m01 h01 m23 h23) | 64 lanes x (l01 l23) ], lane = 16 kq + cout % 16 ----
__global__ void skp_wino4s_filter_kernel(const float* __restrict__ w, unsigned short* __restrict__ Us, int Cout, int Cin, int flip_t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    double g[3][3];
    if (!flip_t) {
        const float* p = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = p[a * 3 + b];
    } else {
        const float* p = w + ((size_t)ci * Cout + co) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = p[(2 - a) * 3 + (2 - b)];
    }
    const double G[6][3] = {{0.25, 0., 0.},
                            {-1. / 6, -1. / 6, -1. / 6},
                            {-1. / 6, 1. / 6, -1. / 6},
                            {1. / 24, 1. / 12, 1. / 6},
                            {1. / 24, -1. / 12, 1. / 6},
                            {0., 0., 1.}};
    double t[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
    const int c16 = ci >> 4, kq = (ci >> 2) & 3, e = ci & 3, cb = co >> 4, i16 = co & 15;
    const int CB = Cout >> 4;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float u = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);   // one rounding, from fp64
            const unsigned short h = w4s_bf16_bits(u);
            const float r1 = u - w4s_bf16_float(h);
            const unsigned short m = w4s_bf16_bits(r1);
            const unsigned short l = w4s_bf16_bits(r1 - w4s_bf16_float(m));
            const int p = i * 6 + j;
            unsigned short* blk = Us + ((size_t)(c16 * CB + cb) * 36 + p) * 768;          // 1536 bytes: MH tuples of the 64 lanes, then their l pairs
            const int ln = kq * 16 + i16;
            blk[ln * 8 + (e >> 1) * 4 + (e & 1)] = m; blk[ln * 8 + (e >> 1) * 4 + 2 + (e & 1)] = h; blk[512 + ln * 4 + e] = l;
        }
}

constexpr int W4S_POS_B = 2 * 4 * 32 * 16;         // one position: [MH | HL][4 kq][32 tiles][16 bytes] = 4 KB
constexpr int W4S_HALF_B = 18 * W4S_POS_B;         // one half-stage buffer: 72 KB
constexpr int W4S_SEG_B = 12 * W4S_POS_B;          // LDS offsets are 16-bit immediates: three base registers, 12 positions (48 KB) apart
#ifndef W4S_ABL                                    // lab builds only (tools/split_ablate.sh): 1 no side jobs, 2 no filter loads, 4 no MFMAs, 8 no LDS operand reads,
                                                   // 16 no patch loads, 32 no column pass, 64 no row transform, 128 no split arithmetic, 256 no LDS stores, 512 truncating split
#define W4S_ABL 0
#endif
constexpr int W4S_RING = 9;                        // filter ring slots (divides 36): prefetch distance 8 positions.  VMEM returns in order, so
                                                   // a filter load queued behind the next stage's patch loads (HBM) inherits their latency

// three bf16 terms of a pair of fp32 values (the two input channels of this thread): packed dwords h, m, l
__device__ __forceinline__ unsigned w4s_cvt_pk(float lo, float hi) {       // {bf16(lo), bf16(hi)}, round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void w4s_split(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = w4s_cvt_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);   // exact
    m = w4s_cvt_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    l = w4s_cvt_pk(sa, sb);
}

// ---- bf16 MFMAs on the named accumulators (skp_wino4_common.h) ----
// NOP: the A tuple was assembled by register moves just before the statement: the matrix instruction must not read a VGPR
// in the two cycles after a VALU wrote it, and nothing pads hazards around / inside an asm statement (measured: NaNs without)
template <int T, bool NOP>   // accumulator tuple T (0..63): a[4T : 4T + 3] += A . B
__device__ __forceinline__ void w4s_mfma_named(const f32x4& A, const f32x4& B) {
    if (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(4 * T), "n"(4 * T + 3), "v"(A), "v"(B) : W4_AGPR_CLOBBERS);
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(4 * T), "n"(4 * T + 3), "v"(A), "v"(B) : W4_AGPR_CLOBBERS);
}
// the same instruction with a VGPR accumulator (positions 32-35): also assembly, so that the compiler never allocates an
// AGPR temporary for an MFMA of its own -- it would pick one of the named registers
__device__ __forceinline__ void w4s_mfma_vgpr(f32x4& acc, const f32x4& A, const f32x4& B) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B) : W4_AGPR_CLOBBERS);
}
template <bool STATS>
__global__ __launch_bounds__(256, 1) void skp_wino4s_conv_kernel(Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][W4S_HALF_B] input tuples; the epilogue parks statistics there
    f32x2* const sst = (f32x2*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    int tblock, cg, zsplit;
    if (!w4_work(a, blockIdx.x + blockIdx.z * gridDim.x, tblock, cg, zsplit)) return;
    const int tile0 = tblock * 32;
    const int n0 = (cg * 4 + wave) * 16;
    const int HW = a.H * a.W;
    const int nsteps = min(a.steps, a.total_steps - zsplit * a.steps);
    const int cin_begin = zsplit * a.steps * 16;

    // ---- transform role: the 6x6 patches of 2 consecutive channels of one tile ----
    const int tl = tid & 31, cp = tid >> 5;
    int roff[6];
    bool lok, rok;
    {
        const int tg = tile0 + tl;
        const bool tv = tg < a.nTiles;
        const int tgc = tv ? tg : 0;
        const int b = tgc / a.tilesPerImg, rem = tgc - b * a.tilesPerImg;
        const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        const int base = (b * a.Cin + 2 * cp) * HW + 4 * tx;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int r = 4 * ty - 1 + i;
            roff[i] = (tv && r >= 0 && r < a.H) ? (base + r * a.W) * 4 : SKP_OOB;
        }
        lok = tx > 0;
        rok = tx + 1 < a.tilesX;
    }
    const i32x4 xrs = skp_make_rsrc(a.x, a.x_bytes);
    const i32x4 urs = skp_make_rsrc(a.U, a.u_bytes);
    f32x2 d[2][6][3];                                // [channel][row][column pair]: pairs (c0,c5), (c1,c2), (c3,c4)
    f32x4 ldmid;
    auto load_part = [&](int cin0, int e, int i, int part) {     // one of the three loads of a patch row
        const int so = (cin0 + e) * HW * 4;
        if (part == 0) {
            ldmid = skp_buf_load_f32x4(xrs, roff[i], so, 0);
            d[e][i][1] = f32x2{ldmid[0], ldmid[1]};
            d[e][i][2] = f32x2{ldmid[2], ldmid[3]};
        } else if (part == 1) {
            d[e][i][0][0] = skp_buf_load_f32(xrs, lok ? roff[i] - 4 : SKP_OOB, so, 0);
        } else {
            d[e][i][0][1] = skp_buf_load_f32(xrs, rok ? roff[i] + 16 : SKP_OOB, so, 0);
        }
    };
    auto col_pass = [&](int e, int k) {
        f32x2 v[6], t[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = d[e][i][k];
        w4_in1d(v, t);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[e][i][k] = t[i];
    };
    // row i of B^T d B for the two channels, cut into steps that ride between the MFMAs of three positions:
    // row_in1d (the row transform of one channel), split_a (first term + residuals of one column), split_b (the other two terms
    // + the two 8-byte LDS stores: (m, h) into the MH tuple array, (h, l) into the HL one)
    float tr[2][6];
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned char* wseg[3];                          // write bases of LDS segments 0-2 (rows 0-1, rows 2-3, rows 4-5)
#pragma unroll
    for (int g = 0; g < 3; ++g) wseg[g] = smem + g * W4S_SEG_B + ((cp >> 1) * 32 + tl) * 16 + (cp & 1) * 8;
    auto row_in1d = [&](int e, int i) {
        if (W4S_ABL & 64) { tr[e][0] = d[e][i][0][0]; tr[e][1] = d[e][i][1][0]; tr[e][2] = d[e][i][1][1]; tr[e][3] = d[e][i][2][0]; tr[e][4] = d[e][i][2][1]; tr[e][5] = d[e][i][0][1]; return; }
        float r[6] = {d[e][i][0][0], d[e][i][1][0], d[e][i][1][1], d[e][i][2][0], d[e][i][2][1], d[e][i][0][1]};
        w4_in1d(r, tr[e]);
    };
    // work items of one row (both channels), ~12 VALU operations each with 2-4 independent chains (one wave per SIMD: a chain
    // of dependent operations issues one instruction per ~5 cycles): I0 / I1 = row transform of channel 0 / 1, A(c) = first
    // bf16 term + residuals of columns 2c, 2c + 1, B(c) = their other two terms + four 8-byte LDS stores
    unsigned sp_h[2];
    float sp_ra[2], sp_rb[2];
    auto split_a2 = [&](int c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 2 * c + q;
            if (W4S_ABL & 128) { sp_h[q] = __builtin_bit_cast(unsigned, tr[0][j]); sp_ra[q] = tr[0][j]; sp_rb[q] = tr[1][j]; continue; }
            if (W4S_ABL & 512) {
                sp_h[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, tr[1][j]), __builtin_bit_cast(unsigned, tr[0][j]), 0x07060302u);
                sp_ra[q] = tr[0][j] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, tr[0][j]) & 0xffff0000u);
                sp_rb[q] = tr[1][j] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, tr[1][j]) & 0xffff0000u);
                continue;
            }
            sp_h[q] = w4s_cvt_pk(tr[0][j], tr[1][j]);
            sp_ra[q] = tr[0][j] - __builtin_bit_cast(float, sp_h[q] << 16);           // exact
            sp_rb[q] = tr[1][j] - __builtin_bit_cast(float, sp_h[q] & 0xffff0000u);
        }
    };
    auto split_b2 = [&](int i, int c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = 2 * c + q;
            unsigned m, l;
            if (W4S_ABL & 128) { m = __builtin_bit_cast(unsigned, sp_ra[q]); l = __builtin_bit_cast(unsigned, sp_rb[q]); }
            else if (W4S_ABL & 512) {
                m = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sp_rb[q]), __builtin_bit_cast(unsigned, sp_ra[q]), 0x07060302u);
                const float sa = sp_ra[q] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sp_ra[q]) & 0xffff0000u);
                const float sb = sp_rb[q] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, sp_rb[q]) & 0xffff0000u);
                l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
            } else {
                m = w4s_cvt_pk(sp_ra[q], sp_rb[q]);
                const float sa = sp_ra[q] - __builtin_bit_cast(float, m << 16), sb = sp_rb[q] - __builtin_bit_cast(float, m & 0xffff0000u);
                l = w4s_cvt_pk(sa, sb);
            }
            if (W4S_ABL & 256) { asm volatile("" : : "v"(m), "v"(l), "v"(sp_h[q])); continue; }
            unsigned char* dst = wseg[i >> 1] + ((i & 1) * 6 + j) * W4S_POS_B;
            *(u32x2*)dst = u32x2{m, sp_h[q]};
            *(u32x2*)(dst + W4S_POS_B / 2) = u32x2{sp_h[q], l};
        }
    };
    // step k (0..5) of the third `t` (0..2) of row i: the eight items sit behind every other MFMA of the row's three positions
    //   t = 0: I0 I1 . A0 . B0      t = 1: . A1 . B1 . A2      t = 2: . B2 . . . .
    auto row_step = [&](int i, int t, int k) {
        if (t == 0) {
            if (k < 2) row_in1d(k, i);
            else if (k == 3) split_a2(0);
            else if (k == 5) split_b2(i, 0);
        } else if (t == 1) {
            if (k == 1) split_a2(1);
            else if (k == 3) split_b2(i, 1);
            else if (k == 5) split_a2(2);
        } else if (k == 1) split_b2(i, 2);
    };

    const unsigned char* vseg[3];                    // read bases of the three LDS segments
#pragma unroll
    for (int g = 0; g < 3; ++g) vseg[g] = smem + g * W4S_SEG_B + (kq * 32 + i16) * 16;

    // zero the named accumulators; positions 32-35 are variables
    w4_unroll([&](auto rc) { asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(decltype(rc)::value) : W4_AGPR_CLOBBERS); },
               std::make_integer_sequence<int, 256>{});
    f32x4 accv[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) accv[p][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // filter operand of (position p, stage c16): 24 bytes per lane, [m01 h01 m23 h23] by one 16-byte load, [l01 l23] by an 8-byte one
    const int u_lane = lane * 16;
    const int u_c16 = (a.Cout >> 4) * (36 * 1536);                  // bytes between stages
    const int u_cb = (cg * 4 + wave) * (36 * 1536);
    f32x4 ua_mh[W4S_RING];
    f32x2 ua_l[W4S_RING];
    auto load_u = [&](int slot, int ub, int p) {                    // ub: scalar byte offset of the (stage, channel block) stream
        ua_mh[slot] = skp_buf_load_f32x4(urs, u_lane, ub + p * 1536, 0);
        ua_l[slot] = skp_buf_load_f32x2(urs, (u_lane >> 1) + 1024, ub + p * 1536, 0);
    };

    // prologue: stage 0 patches -> column pass -> rows 0-2 into half A; first ring slots
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int part = 0; part < 3; ++part) load_part(cin_begin, e, i, part);
#pragma unroll
    for (int q = 0; q < W4S_RING - 1; ++q) load_u(q, (cin_begin >> 4) * u_c16 + u_cb, q);
#pragma unroll
    for (int k = 0; k < 3; ++k) { col_pass(0, k); col_pass(1, k); }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 6; ++k) row_step(i, t, k);
    __syncthreads();

    f32x4 bmh[2], bhl[2];                            // B tuples of the two tile blocks (single-buffered: re-read right after their last use)
    auto read_mh = [&](int p) {
        if (W4S_ABL & 8) return;
        const unsigned char* src = vseg[p / 12] + (p % 12) * W4S_POS_B;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) bmh[tb] = *(const f32x4*)(src + tb * 256);
    };
    auto read_hl = [&](int p) {
        if (W4S_ABL & 8) return;
        const unsigned char* src = vseg[p / 12] + (p % 12) * W4S_POS_B + W4S_POS_B / 2;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) bhl[tb] = *(const f32x4*)(src + tb * 256);
    };

    // One position = six MFMAs (16 cycles each on the matrix pipe) with one step of the side jobs behind each: with one wave per
    // SIMD nothing else hides them.  HALF 0: positions 0-17 (half A); side jobs: rows 3-5 of THIS stage -> half B, then (MODE 0)
    // the patch loads of the next stage.  HALF 1: positions 18-35 (half B); side jobs (MODE 0): column pass of the next stage,
    // its rows 0-2 -> half A.  MODE 1 = the last stage of the workgroup.
    auto run_half = [&](int s, auto half_c, auto mode_c) {
        constexpr int HALF = decltype(half_c)::value, MODE = decltype(mode_c)::value;
        const int ub = ((cin_begin >> 4) + s) * u_c16 + u_cb;
        const int cin_next = cin_begin + (s + 1) * 16;
        read_hl(HALF * 18);
        read_mh(HALF * 18);
        w4_unroll([&](auto plc) {
            constexpr int PL = decltype(plc)::value, P = HALF * 18 + PL, D = W4S_RING - 1;
            // Patch loads of the next stage go out as early as their registers are free (VMEM returns in order: the filter loads
            // queued behind them inherit their HBM latency, so they need a whole half-stage of slack before the column pass):
            // rows 0-2 were consumed in the previous half, row 3 / 4 / 5 are dead once their row transform has run (PL 0 / 3 / 6).
            auto side = [&](int k) {
                if (W4S_ABL & 1) return;
                if (HALF == 0) {
                    if (PL < 9) row_step(3 + PL / 3, PL % 3, k);
                    if (MODE == 0 && !(W4S_ABL & 16)) {
                        constexpr int LR = PL < 5 ? PL : (PL == 7 ? 5 : -1);       // the patch row loaded beside this position
                        if (LR >= 0) load_part(cin_next, k / 3, LR, k % 3);
                    }
                } else if (MODE == 0) {
                    if (PL >= 6 && PL < 9) { if (W4S_ABL & 32) return; if (k == 0) col_pass(0, PL - 6); else if (k == 3) col_pass(1, PL - 6); }
                    else if (PL >= 9) row_step((PL - 9) / 3, (PL - 9) % 3, k);
                }
            };
            {   // filter operand of position P + D (wrapping into the next stage)
                constexpr int Q = P + D;
                if (!(W4S_ABL & 2) && (MODE == 0 || Q < 36)) load_u(Q % W4S_RING, Q < 36 ? ub : ub + u_c16, Q < 36 ? Q : Q - 36);
            }
            const f32x4 umh = ua_mh[P % W4S_RING];
            const f32x2 ul = ua_l[P % W4S_RING];
            const f32x4 uhl = {umh[1], ul[0], umh[3], ul[1]};
            // small terms first: m.h + h.l, h.m + l.h, then m.m + h.h; the two tile blocks alternate
            auto mfma = [&](auto tbc, auto nopc, const f32x4& A, const f32x4& B) {
                constexpr int TB = decltype(tbc)::value;
                if (W4S_ABL & 4) return;
                if constexpr (P < 32) w4s_mfma_named<2 * P + TB, decltype(nopc)::value != 0>(A, B);
                else w4s_mfma_vgpr(accv[P - 32][TB], A, B);
            };
            using T0 = std::integral_constant<int, 0>;
            using T1 = std::integral_constant<int, 1>;
            mfma(T0{}, T0{}, umh, bhl[0]); side(0); __builtin_amdgcn_sched_barrier(0);
            mfma(T1{}, T0{}, umh, bhl[1]); side(1); if (PL + 1 < 18) read_hl(P + 1); __builtin_amdgcn_sched_barrier(0);
            mfma(T0{}, T1{}, uhl, bmh[0]); side(2); __builtin_amdgcn_sched_barrier(0);
            mfma(T1{}, T1{}, uhl, bmh[1]); side(3); __builtin_amdgcn_sched_barrier(0);
            mfma(T0{}, T0{}, umh, bmh[0]); side(4); __builtin_amdgcn_sched_barrier(0);
            mfma(T1{}, T0{}, umh, bmh[1]); side(5); if (PL + 1 < 18) read_mh(P + 1); __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 18>{});
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    for (int s = 0; s + 1 < nsteps; ++s) {
        run_half(s, I0{}, I0{});
        __syncthreads();
        run_half(s, I1{}, I0{});
        __syncthreads();
    }
    run_half(nsteps - 1, I0{}, I1{});
    __syncthreads();
    run_half(nsteps - 1, I1{}, I1{});
    if (STATS) __syncthreads();                      // the epilogue parks statistics in the stage buffers

    // ---- output transform (in-lane) + store, as skp_wino4_conv_kernel ----
    // ---- output role (lane = tile within a 16-block, registers = 4 output channels) ----
    int o_base[2];
    bool t_ok[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int tg = tile0 + tb * 16 + i16;
        t_ok[tb] = tg < a.nTiles;
        const int tgc = t_ok[tb] ? tg : 0;
        const int b = tgc / a.tilesPerImg, rem = tgc - b * a.tilesPerImg;
        const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        o_base[tb] = ((b * a.Cout) * a.H + 4 * ty) * a.W + 4 * tx;
    }
    const i32x4 yrs = skp_make_rsrc(a.y + zsplit * a.y_split_stride, a.y_bytes);
    const i32x4 rrs = skp_make_rsrc(a.res, a.res ? a.y_bytes : 0u);
    const i32x4 brs = skp_make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
    f32x4 rr[4][2][4];
    float bvs[4];
    auto load_res = [&](int r) {
        const int co = n0 + 4 * kq + r;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const bool ok = t_ok[tb] && co < a.Cout;
            const int vo = (o_base[tb] + co * HW) * 4;
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) rr[r][tb][oy] = skp_buf_load_f32x4(rrs, ok ? vo + oy * a.W * 4 : SKP_OOB, 0, 0);
        }
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = n0 + 4 * kq + r;
        bvs[r] = skp_buf_load_f32(brs, co < a.Cout ? co * 4 : SKP_OOB, 0, 0);
    }
    load_res(0);
    load_res(1);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs' results are in the register file before the first read
    __builtin_amdgcn_sched_barrier(0);
    w4_unroll([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int co = n0 + 4 * kq + r;
        const float bv = bvs[r];
        if (r + 2 < 4) load_res(r + 2);
        w4_unroll([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value;
            const bool ok = t_ok[tb] && co < a.Cout;
            const int vo = (o_base[tb] + co * HW) * 4;
            float t[6][4];                           // T = M A : rows of the 6x6 tile -> 4 columns
            w4_unroll([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float m[6];
                w4_unroll([&](auto jc) {
                    constexpr int j = decltype(jc)::value, p = i * 6 + j;
                    if constexpr (p < 32) m[j] = w4_acc_read<(2 * p + tb) * 4 + r>();
                    else m[j] = accv[p - 32][tb][r];
                }, std::make_integer_sequence<int, 6>{});
                w4_out1d(m, t[i]);
            }, std::make_integer_sequence<int, 6>{});
#pragma unroll
            for (int ox = 0; ox < 4; ++ox) {
                float m[6], yv[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) m[i] = t[i][ox];
                w4_out1d(m, yv);
#pragma unroll
                for (int oy = 0; oy < 4; ++oy) rr[r][tb][oy][ox] += yv[oy] + bv;
            }
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) skp_buf_store_f32x4(rr[r][tb][oy], yrs, ok ? vo + oy * a.W * 4 : SKP_OOB, 0, 0);
            if (STATS) w4_park_stats(sst, wave * 8 + r * 2 + tb, lane, rr[r][tb], ok);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 2>{});
    }, std::make_integer_sequence<int, 4>{});
    if (STATS) {                                     // 64 channels x 2 tile blocks = 128 (channel, block) pairs
        __syncthreads();
        if (tid < 128) {
            const int wv = tid >> 5, tb = (tid >> 4) & 1, lc = tid & 15;
            w4_store_stats(a, sst, wv * 8 + (lc & 3) * 2 + tb, lc >> 2, tile0 + tb * 16, (cg * 4 + wv) * 16 + lc);
        }
    }
}

// ---- geometry (always 64 channels x 32 tiles), K-split plan, gate ----
static Wino4Grid wino4s_grid(int Cout, int tiles, int S) {
    Wino4Grid g;
    g.ntb = (tiles + 31) / 32;
    g.ncg = Cout / 64;
    g.tb_per_xcd = g.ntb >= 32 ? (g.ntb + 7) / 8 : 0;
    if (g.tb_per_xcd) {
        g.gx = 8u * g.tb_per_xcd * g.ncg;
        g.rounds = (int)((g.gx * (unsigned)S + 255) / 256);
    } else {
        const int upx = (g.ncg * S + 7) / 8;
        g.gx = 8u * upx * g.ntb;
        g.rounds = (upx * g.ntb + 31) / 32;
    }
    return g;
}
static bool wino4s_layout_ok(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return false;
    if ((Cin % 16) || (Cout % 64) || (H % 4) || (W % 4)) return false;
    return (long long)B * Cin * H * W * 4 < 0x7fffffffll && 36ll * Cin * Cout * 6 < 0x7fffffffll && (long long)B * Cout * H * W * 4 < 0x7fffffffll;
}
static int wino4s_plan(int B, int Cin, int Cout, int H, int W) {
    if (!wino4s_layout_ok(B, Cin, Cout, H, W)) return 0;
    const int tiles = B * (H / 4) * (W / 4);
    const int nsteps = Cin / 16;
    const double out_bytes = (double)B * Cout * H * W * 4;
    if (const char* e = getenv("SKP_WINO_SPLIT")) {
        const int S = atoi(e);
        if (S >= 1 && S <= 16 && (S - 1) * ((nsteps + S - 1) / S) < nsteps) return S;
    }
    int best = 1;
    double best_cost = 1e30;
    const double stage_us = 2.8;                   // ~5.5k cycles at ~2 GHz
    for (int S = 1; S <= 16; ++S) {
        const int per = (nsteps + S - 1) / S;
        if ((S - 1) * per >= nsteps) continue;
        const Wino4Grid g = wino4s_grid(Cout, tiles, S);
        double cost = g.rounds * (per + 3.0) * stage_us;
        if (S > 1) cost += 6.0 + (S + 1) * out_bytes / 8.0e6;
        if (cost < best_cost * (S > 1 ? 0.92 : 1.0)) { best_cost = cost; best = S; }
    }
    return best;
}

}  // namespace

// 1 where the split kernel RUNS (layout): Cin % 16 == 0, Cout % 64 == 0, H, W % 4 == 0, 32-bit byte offsets.
extern "C" int skp_conv3x3_f4s_ok(int B, int Cin, int Cout, int H, int W) {
    return wino4s_layout_ok(B, Cin, Cout, H, W) ? 1 : 0;
}

// Us: 36 * Cin * Cout * 3 bf16 (6 bytes per filter value of the transform domain).
extern "C" int skp_conv3x3_f4s_filter_f32(const void* w, void* Us, int Cout, int Cin, int flip_transpose, void* stream) {
    if (!w || !Us || Cout <= 0 || Cin <= 0) return SKP_E_BADARG;
    if ((Cin & 15) || (Cout & 15)) return SKP_E_RANGE;
    const int n = Cout * Cin;
    hipLaunchKernelGGL(skp_wino4s_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                       (unsigned short*)Us, Cout, Cin, flip_transpose);
    return skp_launch_status();
}

extern "C" int64_t skp_conv3x3_f4s_workspace(int B, int Cin, int Cout, int H, int W) {
    const int S = wino4s_plan(B, Cin, Cout, H, W);
    return S > 1 ? (int64_t)S * B * Cout * H * W * (int64_t)sizeof(float) : 0;
}

// 16-tile blocks per image when the launch can emit output statistics (unsplit, blocks do not straddle images), else 0
extern "C" int skp_conv3x3_f4s_stats_blocks(int B, int Cin, int Cout, int H, int W) {
    if (wino4s_plan(B, Cin, Cout, H, W) != 1) return 0;
    const int tpi = (H / 4) * (W / 4);
    return (tpi % 32 == 0) ? tpi / 16 : 0;
}

// y = conv3x3(x) (+ bias) (+ residual), fp32 in / out, products on the bf16 matrix cores (three-term splits, six products).
// workspace: skp_conv3x3_f4s_workspace() bytes (NULL forces an unsplit launch); stats: optional
// [B][Cout][skp_conv3x3_f4s_stats_blocks()][2] = {mean, sum (y - mean)^2} per 16-tile block (unsplit launches only).
extern "C" int skp_conv3x3_f4s_f32(const void* x, const void* Us, const void* bias, const void* residual, void* y, void* workspace,
                                   float* stats, int B, int Cin, int Cout, int H, int W, void* stream) {
    if (!x || !Us || !y) return SKP_E_BADARG;
    if (!wino4s_layout_ok(B, Cin, Cout, H, W)) return (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) ? SKP_E_BADARG : SKP_E_RANGE;
    int S = wino4s_plan(B, Cin, Cout, H, W);
    if (!workspace) S = 1;
    if (stats && (S != 1 || skp_conv3x3_f4s_stats_blocks(B, Cin, Cout, H, W) == 0)) return SKP_E_RANGE;
    Wino4Args a;
    a.x = (const float*)x; a.U = (const float*)Us;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.tilesX = W / 4;
    a.tilesPerImg = a.tilesX * (H / 4);
    a.nTiles = B * a.tilesPerImg;
    a.x_bytes = (unsigned)((size_t)B * Cin * H * W * 4);
    a.u_bytes = (unsigned)((size_t)36 * Cin * Cout * 6);
    const size_t out_elems = (size_t)B * Cout * H * W;
    a.y_bytes = (unsigned)(out_elems * 4);
    a.total_steps = Cin / 16;
    a.steps = (a.total_steps + S - 1) / S;
    a.splits = S;
    a.y_split_stride = out_elems;
    a.y = S > 1 ? (float*)workspace : (float*)y;
    a.bias = S > 1 ? nullptr : (const float*)bias;
    a.res = S > 1 ? nullptr : (const float*)residual;
    a.stats = S > 1 ? nullptr : stats;
    a.sblk = a.tilesPerImg / 16;
    a.gncoef = nullptr; a.vpad = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * W4S_HALF_B;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_wino4s_conv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4s_conv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const Wino4Grid g = wino4s_grid(Cout, a.nTiles, S);
    a.ntb = g.ntb; a.ncg = g.ncg; a.tb_per_xcd = g.tb_per_xcd;
    a.gx = (int)g.gx;
    a.vtotal = g.tb_per_xcd ? (int)g.gx * S : (int)g.gx;
    const dim3 grid = g.tb_per_xcd ? dim3(g.gx, 1, S) : dim3(g.gx, 1, 1);
    if (a.stats) hipLaunchKernelGGL(skp_wino4s_conv_kernel<true>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(skp_wino4s_conv_kernel<false>, grid, dim3(256), lds, st, a);
    int rc = skp_launch_status();
    if (rc || S == 1) return rc;
    const size_t n4 = out_elems / 4;
    hipLaunchKernelGGL(skp_wino4_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)workspace,
                       (const float*)bias, (const float*)residual, (float*)y, n4, out_elems, S, (H * W) / 4, Cout);
    return skp_launch_status();
}
