// 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4,3x3) on the fp32 matrix cores, for layers whose height and
// width are multiples of 4 (all VAE-encoder and UNet resolutions of the 512^2 path): 36 multiplies per 4x4 output
// tile and channel pair, i.e. 4x fewer than the direct form (F(2x2,3x3) in skp_conv_wino.hip: 2.25x).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          6x6 transform domain, interpolation points 0, +-1, +-2, inf
//
// Work split (v_mfma_f32_16x16x4_f32), two workgroup forms with the same 288 accumulator registers per wave (of the
// 512 registers of its SIMD: one wave per SIMD), so that the 6x6 -> 4x4 output transform stays in-lane:
//   * skp_wino4_conv_kernel:      wave = 16 output channels x 32 tiles x 36 positions, workgroup = 64 channels x 32 tiles
//   * skp_wino4_conv_c128_kernel: wave = 32 output channels x 16 tiles x 36 positions, workgroup = 128 channels x 16 tiles
//     (>= 128 tiles, Cout a multiple of 128 or its last group >= 64): half the input-transform work per MFMA, two filter
//     loads per position.
//   * skp_wino4r_conv_kernel:     the first form's tile with the filter kept as its 9 taps and transformed in the lanes, the input
//     transformed once per launch by its own kernel (small-spatial layers with >= 1280 channels: see the section further down)
// The workgroup transforms the input patches of a 16-channel stage into LDS once (double buffered, 2 x 72 / 36 KB) in
// MFMA operand order.  The transformed filter is streamed from L2 through a register ring (12 / 6 positions deep; VMEM
// returns in order: everything queued behind a patch load inherits its latency, so the ring has to cover it).  The
// patch loads of the next stage are issued in the first positions of the MFMA loop and their 6x6 transform + LDS writes
// are spread over the last positions.  With one wave per SIMD every non-MFMA instruction is exposed, so the loop is
// kept to: one (two) filter load(s), two (one) LDS operand reads and 8 MFMAs per position, plus those side jobs.
// Zero padding, ragged tile blocks, idle channel rows, bias and residual all go through raw buffer loads / stores whose
// out-of-range offsets return 0 / are dropped: no divergent branch anywhere near the accumulators (a branch around
// them costs hundreds of spilled registers).  fp32 throughout; relative error ~1e-5 of the output maximum.
#include <algorithm>
#include <type_traits>
#include "skp_common.h"
#include "skp_wino4_common.h"
#include <stdlib.h>

namespace {

// ---- filter transform: U'[p][ci/16][(ci%16)/4][co][ci%4] = (G g G^T)[p], p = 6*i + j ----------------------------
__global__ void skp_wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int flip_t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    float g[3][3];
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
    const float G[6][3] = {{0.25f, 0.f, 0.f},
                           {-1.f / 6, -1.f / 6, -1.f / 6},
                           {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6},
                           {1.f / 24, -1.f / 12, 1.f / 6},
                           {0.f, 0.f, 1.f}};
    float t[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
    const int c16 = ci >> 4, kq = (ci >> 2) & 3, m = ci & 3;
    const int C16 = Cin >> 4;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
            const int p = i * 6 + j;
            U[((((size_t)p * C16 + c16) * 4 + kq) * Cout + co) * 4 + m] = u;
        }
}

// fp32 MFMAs on NAMED accumulators (skp_wino4_common.h): accumulator tuple T = 2 * position + channel block for positions 0-31
// lives in a[4T : 4T + 3]; positions 32-35 are VGPR tuples.  Operands come straight from buffer loads / LDS reads (no VALU write
// in front of the statement: no wait states needed).
template <int P, int CB>
__device__ __forceinline__ void w4c_mfma(f32x4 (&accv)[4][2], float a, float b) {
    if constexpr (P < 32)
        asm volatile("v_mfma_f32_16x16x4_f32 a[%0:%1], %2, %3, a[%0:%1]" : : "n"(4 * (2 * P + CB)), "n"(4 * (2 * P + CB) + 3), "v"(a), "v"(b) : W4_AGPR_CLOBBERS);
    else
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv[P - 32][CB]) : "v"(a), "v"(b) : W4_AGPR_CLOBBERS);
}
template <int P, int CB, int R>
__device__ __forceinline__ float w4c_acc(const f32x4 (&accv)[4][2]) {
    if constexpr (P < 32) return w4_acc_read<4 * (2 * P + CB) + R>();
    else return accv[P - 32][CB][R];
}

template <bool STATS>
__global__ __launch_bounds__(256, 1) void skp_wino4_conv_kernel(Wino4Args a) {
    extern __shared__ f32x4 vst[];                   // [2][36][4][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    int roff[6];                                     // byte offset of (row i, column 4*tx) for channel pair 0, or SKP_OOB
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
    // patch registers [channel][row][column pair]: pairs (c0,c5), (c1,c2), (c3,c4) -- the aligned 4-column load lands in
    // two pairs, the two halo scalars share the third -- so the vertical pass runs on v_pk_* ops, two columns per op
    f32x2 d[2][6][3];
    auto load_row = [&](int cin0, int e, int i) {    // one patch row: left scalar, 4 aligned columns, right scalar
        const int so = (cin0 + e) * HW * 4;
        const f32x4 mid = skp_buf_load_f32x4(xrs, roff[i], so, 0);
        d[e][i][0][0] = skp_buf_load_f32(xrs, lok ? roff[i] - 4 : SKP_OOB, so, 0);
        d[e][i][1] = f32x2{mid[0], mid[1]};
        d[e][i][2] = f32x2{mid[2], mid[3]};
        d[e][i][0][1] = skp_buf_load_f32(xrs, rok ? roff[i] + 16 : SKP_OOB, so, 0);
    };
    auto col_pass = [&](int e, int k) {              // d[e][:, pair k] <- B^T d[e][:, pair k]   (vertical pass)
        f32x2 v[6], t[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = d[e][i][k];
        w4_in1d(v, t);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[e][i][k] = t[i];
    };
    auto row_pass_store = [&](int buf, int i) {      // V[i][:] = (row i) B, both channels, written as float2 (channels 2cp, 2cp+1)
        float r0[6] = {d[0][i][0][0], d[0][i][1][0], d[0][i][1][1], d[0][i][2][0], d[0][i][2][1], d[0][i][0][1]};
        float r1[6] = {d[1][i][0][0], d[1][i][1][0], d[1][i][1][1], d[1][i][2][0], d[1][i][2][1], d[1][i][0][1]};
        float t0[6], t1[6];
        w4_in1d(r0, t0);
        w4_in1d(r1, t1);
        float* dst = (float*)(vst + buf * W4_STAGE_F4) + (((i * 6) * 4 + (cp >> 1)) * 32 + tl) * 4 + 2 * (cp & 1);
#pragma unroll
        for (int j = 0; j < 6; ++j) *(f32x2*)(dst + j * (4 * 32 * 4)) = f32x2{t0[j], t1[j]};
    };

    // accumulators [position][tile block]: positions 0-31 by NAME in a[0:255] (w4c_mfma), 32-35 in VGPR tuples
    w4_unroll([&](auto rc) { asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(decltype(rc)::value) : W4_AGPR_CLOBBERS); },
              std::make_integer_sequence<int, 256>{});
    f32x4 accv[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) accv[p][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int C16 = a.Cin >> 4;
    const int co_l = min(n0 + i16, a.Cout - 1);
    const int uvo = (kq * a.Cout + co_l) * 16;
    const int u_c16 = 4 * a.Cout * 16, u_p = C16 * u_c16;

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

    // prologue: stage 0 patches, first ring slots
#pragma unroll
    for (int i = 0; i < 6; ++i) { load_row(cin_begin, 0, i); load_row(cin_begin, 1, i); }
#pragma unroll
    for (int k = 0; k < 3; ++k) { col_pass(0, k); col_pass(1, k); }
#pragma unroll
    for (int i = 0; i < 6; ++i) row_pass_store(0, i);
    f32x4 ua[W4_RING];
#pragma unroll
    for (int q = 0; q < W4_RING - 1; ++q) ua[q] = skp_buf_load_f32x4(urs, uvo, (cin_begin >> 4) * u_c16 + q * u_p, 0);
    __syncthreads();

    // MODE 0: a further stage follows (its patch loads + transform ride along); MODE 1: last stage
    auto run_stage = [&](int s, auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        const f32x4* vb = vst + (s & 1) * W4_STAGE_F4 + kq * 32 + i16;
        const int ub = ((cin_begin >> 4) + s) * u_c16;
        f32x4 va[3][2];                              // LDS operands two positions ahead (LDS latency > one position's MFMAs)
        va[0][0] = vb[0];
        va[0][1] = vb[16];
        va[1][0] = vb[128];
        va[1][1] = vb[128 + 16];
        w4_unroll([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            {   // filter operand for position p + RING-1 (wrapping into the next stage)
                constexpr int D = W4_RING - 1;
                constexpr int q = p + D;
                if (MODE == 0 || q < 36) {
                    const int uo = q < 36 ? ub + q * u_p : ub + u_c16 + (q - 36) * u_p;
                    ua[q % W4_RING] = skp_buf_load_f32x4(urs, uvo, uo, 0);
                }
            }
            if (MODE == 0) {
                if (p < 6) {                         // next stage's patch rows, both channels
                    load_row(cin_begin + (s + 1) * 16, 0, p);
                    load_row(cin_begin + (s + 1) * 16, 1, p);
                } else if (p >= 24 && p < 30) {
                    col_pass((p - 24) & 1, (p - 24) >> 1);
                } else if (p >= 30) {
                    row_pass_store((s + 1) & 1, p - 30);
                }
            }
            if (p + 2 < 36) {
                va[(p + 2) % 3][0] = vb[(p + 2) * 128];
                va[(p + 2) % 3][1] = vb[(p + 2) * 128 + 16];
            }
            w4_unroll([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                w4c_mfma<p, 0>(accv, ua[p % W4_RING][m], va[p % 3][0][m]);
                w4c_mfma<p, 1>(accv, ua[p % W4_RING][m], va[p % 3][1][m]);
            }, std::make_integer_sequence<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 36>{});
    };
    for (int s = 0; s + 1 < nsteps; ++s) {
        run_stage(s, std::integral_constant<int, 0>{});
        __syncthreads();
    }
    run_stage(nsteps - 1, std::integral_constant<int, 1>{});
    if (STATS) __syncthreads();                      // the epilogue parks statistics in the stage buffers

    // ---- output transform (in-lane) + store.  Branch-free: invalid tiles/channels store out of range (dropped). ----
    const i32x4 yrs = skp_make_rsrc(a.y + zsplit * a.y_split_stride, a.y_bytes);
    const i32x4 rrs = skp_make_rsrc(a.res, a.res ? a.y_bytes : 0u);
    const i32x4 brs = skp_make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
    // residual rows are requested two output channels ahead of their use (one exposed latency, bounded registers)
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
                w4_unroll([&](auto jc) { constexpr int j = decltype(jc)::value; m[j] = w4c_acc<i * 6 + j, tb, r>(accv); },
                          std::make_integer_sequence<int, 6>{});
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
            if (STATS) w4_park_stats((f32x2*)vst, wave * 8 + r * 2 + tb, lane, rr[r][tb], ok);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 2>{});
    }, std::make_integer_sequence<int, 4>{});
    if (STATS) {                                     // 64 channels x 2 tile blocks = 128 (channel, block) pairs
        __syncthreads();
        if (tid < 128) {
            const int wv = tid >> 5, tb = (tid >> 4) & 1, lc = tid & 15;
            w4_store_stats(a, (const f32x2*)vst, wv * 8 + (lc & 3) * 2 + tb, lc >> 2, tile0 + tb * 16, (cg * 4 + wv) * 16 + lc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Raw-filter form for the small-spatial UNet layers (8^2 .. 32^2 at 8 rows: 32 .. 512 tiles against 640 .. 2560 channels).
// There the kernels above are bound by their FILTER stream, not by the matrix cores: every workgroup pulls 36 transformed
// values per (cin, cout) pair for at most a few tile blocks (1280 -> 1280 at 8^2: 236 MB per launch from HBM for 3.8 GF).
// Here the filter stays as the 9 taps it is (4x fewer bytes) and G g G^T is applied by the lane that owns the pair, on its
// way into the MFMA A operand; the INPUT transform moves out of the kernel instead (skp_wino4r_input_kernel, once per launch
// for all channel groups: with 10 - 40 channel groups the in-kernel transform was redone that many times), so the stage loop
// has no patch loads, no transform role and no LDS writes: the transformed tiles of the next stage arrive by LDS DMA
// (buffer_load ... lds, no registers), the next stage's taps in 9 registers quads.
//   U = (D G') g (D G')^T with G' = [1 0 0; 1 1 1; 1 -1 1; 1 2 4; 1 -2 4; 0 0 1], D = diag(1/4, -1/6, -1/6, 1/24, 1/24, 1):
//   the scales d_i d_j are folded into the pre-transformed input, the lanes apply the small-integer G' only
//   (60 four-wide VALU operations per stage and lane against 288 MFMAs).
// Layouts:  Rw[Cin/16][Cout/16][9 taps][kq][i16][m]   (cin = 16 c16 + 4 kq + m, cout = 16 cb + i16): one 1 KB row per load
//           Vg[Cin/16][36][kq][vpad tiles][m]          scaled B^T d B, zero in the padding tiles
// Workgroup = 64 output channels x 32 tiles (wave: 16 x 32, 288 accumulators), K splits and work order as skp_wino4_conv_kernel.
__global__ void skp_wino4r_filter_kernel(const float* __restrict__ w, float* __restrict__ R, int Cout, int Cin, int flip_t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    const int c16 = ci >> 4, kq = (ci >> 2) & 3, m = ci & 3, cb = co >> 4, i16 = co & 15, CB = Cout >> 4;
    float* dst = R + ((size_t)(c16 * CB + cb) * 9 * 64 + kq * 16 + i16) * 4 + m;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float v = flip_t ? w[((size_t)ci * Cout + co) * 9 + (8 - t)] : w[((size_t)co * Cin + ci) * 9 + t];
        dst[t * 256] = v;
    }
}

// Vg = scaled input transform; thread = (tile, channel), the 4 channels of an operand quad in adjacent lanes (16-byte rows of Vg)
__global__ __launch_bounds__(256) void skp_wino4r_input_kernel(const float* __restrict__ x, float* __restrict__ Vg, int B, int Cin, int H,
                                                               int W, int tilesX, int tilesPerImg, int nTiles, int vpad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int m = idx & 3, rest = idx >> 2;
    const int tg = rest % vpad, q4 = rest / vpad;
    if (q4 >= (Cin >> 2)) return;
    const bool tv = tg < nTiles;
    const int tgc = tv ? tg : 0;
    const int b = tgc / tilesPerImg, rem = tgc - b * tilesPerImg;
    const int ty = rem / tilesX, tx = rem - ty * tilesX;
    const float dsc[6] = {0.25f, -1.f / 6, -1.f / 6, 1.f / 24, 1.f / 24, 1.f};
    const float* xc = x + ((size_t)b * Cin + q4 * 4 + m) * H * W;
    float d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int r = 4 * ty - 1 + i;
        const bool rv = tv && r >= 0 && r < H;
        const float* row = xc + r * W + 4 * tx;
        const f32x4 mid = rv ? *(const f32x4*)row : f32x4{0.f, 0.f, 0.f, 0.f};
        d[i][0] = (rv && tx > 0) ? row[-1] : 0.f;
        d[i][1] = mid[0]; d[i][2] = mid[1]; d[i][3] = mid[2]; d[i][4] = mid[3];
        d[i][5] = (rv && tx + 1 < tilesX) ? row[4] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float v[6], t[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = d[i][j];
        w4_in1d(v, t);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i][j] = t[i];
    }
    const int c16 = q4 >> 2, kq = q4 & 3;
    float* dst = Vg + (((size_t)c16 * 36 * 4 + kq) * vpad + tg) * 4 + m;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float t[6];
        w4_in1d(d[i], t);
#pragma unroll
        for (int j = 0; j < 6; ++j) dst[(size_t)(i * 6 + j) * 16 * vpad] = t[j] * (dsc[i] * dsc[j]);
    }
}

#ifdef W4R_STAMPS                                   // lab builds only (tools/): cycle stamps of workgroups 0 and 100
__device__ unsigned long long w4r_stamps[2][64];
#define W4R_STAMP(k) do { if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && (k) < 64) w4r_stamps[blockIdx.x ? 1 : 0][k] = __builtin_readcyclecounter(); } while (0)
#else
#define W4R_STAMP(k) do { } while (0)
#endif
// PART: the launch is K-split -- the output is a partial sum for skp_wino4_reduce_kernel (which adds bias / residual): the
// epilogue carries no bias / residual prefetch (128 registers less at the end of the stage loop).
// NTB: tile blocks of 16 the wave multiplies (2: the 32 tiles of the workgroup; 1: launches with <= 16 tiles -- the 8^2 layers of a
// 1- or 2-image step, config 3's per-rank shape -- where the second block would be all padding: half the MFMAs per stage).
template <bool PART, int NTB = 2>
__global__ __launch_bounds__(256, 1) void skp_wino4r_conv_kernel(Wino4Args a) {
    extern __shared__ f32x4 vst[];                   // [2][36][4][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    int tblock, cg, zsplit;
    if (!w4_work(a, blockIdx.x + blockIdx.z * gridDim.x, tblock, cg, zsplit)) return;
    W4R_STAMP(0);
    const int tile0 = tblock * 32;
    const int n0 = (cg * 4 + wave) * 16;
    const int HW = a.H * a.W;
    const int nsteps = min(a.steps, a.total_steps - zsplit * a.steps);
    const int c16_0 = zsplit * a.steps;
    const i32x4 vrs = skp_make_rsrc(a.x, a.x_bytes);
    const i32x4 wrs = skp_make_rsrc(a.U, a.u_bytes);

    // transformed tiles of a stage -> LDS: f32x4 number r * 256 + tid of the stage image (row = 8 r + tid / 32 of its 144 (p, kq) rows)
    const int v_vo = ((tid >> 5) * a.vpad + tile0 + (tid & 31)) * 16;
    const int v_row8 = a.vpad * 128;                 // bytes between rows 8 apart
    auto load_v = [&](int c16, int buf, int r) {
        skp_buf_load_lds(vrs, (skp_lds_ptr)(vst + buf * W4_STAGE_F4 + r * 256 + wave * 64), 16, v_vo, c16 * 18 * v_row8 + r * v_row8, 0, 0);
    };
    // taps of the lane's four (cin, cout) pairs, one f32x4 (over cin % 4) per tap
    const int CB = a.Cout >> 4;
    const int w_cb = ((cg * 4 + wave) * 9) * 1024;
    const int w_c16 = CB * 9 * 1024;
    f32x4 g[9], gn[9];
    auto load_g = [&](f32x4 (&dst)[9], int c16) {
#pragma unroll
        for (int t = 0; t < 9; ++t) dst[t] = skp_buf_load_f32x4(wrs, lane * 16, c16 * w_c16 + w_cb + t * 1024, 0);
    };
    // row i of G' g G'^T in six pieces (one per position of the previous row, so that a few VALU operations sit between the MFMAs
    // of every position instead of forty in front of one): piece 0..2 = the column combination t[b] of row i, 3..5 = the row
    // combinations (u[0] = t[0] and u[5] = t[2] are aliases)
    f32x4 ut[3];
    auto u_piece = [&](const f32x4 (&gg)[9], int i, int piece, f32x4 (&u)[6]) {
        if (piece < 3) {
            const int b = piece;
            const f32x4 g0 = gg[b], g1 = gg[3 + b], g2 = gg[6 + b];
            ut[b] = i == 0 ? g0 : i == 5 ? g2 : i == 1 ? (g0 + g2) + g1 : i == 2 ? (g0 + g2) - g1
                  : i == 3 ? (g0 + 4.f * g2) + 2.f * g1 : (g0 + 4.f * g2) - 2.f * g1;
        } else if (piece == 3) {
            const f32x4 s2 = ut[0] + ut[2];
            u[0] = ut[0]; u[5] = ut[2];
            u[1] = s2 + ut[1]; u[2] = s2 - ut[1];
        } else if (piece == 4) {
            const f32x4 e = ut[0] + 4.f * ut[2];
            u[3] = e + 2.f * ut[1];
        } else {
            const f32x4 e = ut[0] + 4.f * ut[2];
            u[4] = e - 2.f * ut[1];
        }
    };
    auto u_row = [&](const f32x4 (&gg)[9], int i, f32x4 (&u)[6]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) u_piece(gg, i, k, u);
    };

    // accumulators [position][tile block]: positions 0-31 by NAME in a[0:255] (w4c_mfma), 32-35 in VGPR tuples.  The A operand
    // here is computed by VALU operations (u_piece) -- of the NEXT row, at least one position before its first MFMA, far
    // outside the two wait states a VALU write needs in front of an MFMA read.
    w4_unroll([&](auto rc) { asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(decltype(rc)::value) : W4_AGPR_CLOBBERS); },
              std::make_integer_sequence<int, 256>{});
    f32x4 accv[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) accv[p][tb] = f32x4{0.f, 0.f, 0.f, 0.f};

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

    // prologue: stage 0 tiles and taps
#pragma unroll
    for (int r = 0; r < 18; ++r) load_v(c16_0, 0, r);
    load_g(g, c16_0);
    f32x4 ur[2][6];
    u_row(g, 0, ur[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    W4R_STAMP(1);

    // One code path for every stage (a second copy of the MFMA loop for the last stage cost spilled accumulators): the side
    // requests of the last stage re-fetch that stage (clamped index; nothing reads them).
    for (int s = 0; s < nsteps; ++s) {
        const f32x4* vb = vst + (s & 1) * W4_STAGE_F4 + kq * 32 + i16;
        const int c16n = c16_0 + min(s + 1, nsteps - 1);
        f32x4 va[3][2];                              // LDS operands two positions ahead (three: measured no faster)
        va[0][0] = vb[0];
        va[1][0] = vb[128];
        if (NTB == 2) { va[0][1] = vb[16]; va[1][1] = vb[128 + 16]; }
        w4_unroll([&](auto pc) {
            constexpr int p = decltype(pc)::value, i = p / 6, j = p - 6 * i;
            if (p == 0) load_g(gn, c16n);                // the taps first: they come from HBM, the tiles (L2 / MALL) queue behind them
            else if (p <= 18) load_v(c16n, (s + 1) & 1, p - 1);
            if (i < 5) u_piece(g, i + 1, j, ur[(i + 1) & 1]);
            if (p + 2 < 36) {
                va[(p + 2) % 3][0] = vb[(p + 2) * 128];
                if (NTB == 2) va[(p + 2) % 3][1] = vb[(p + 2) * 128 + 16];
            }
            w4_unroll([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                w4c_mfma<p, 0>(accv, ur[i & 1][j][m], va[p % 3][0][m]);
                if constexpr (NTB == 2) w4c_mfma<p, 1>(accv, ur[i & 1][j][m], va[p % 3][1][m]);
            }, std::make_integer_sequence<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 36>{});
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t] = gn[t];
        u_row(g, 0, ur[0]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the LDS DMA of the next stage's tiles has landed
        __syncthreads();
        W4R_STAMP(2 + s);
    }

    // ---- output transform (in-lane) + store, as skp_wino4_conv_kernel ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs' results are in the register file before the first read
    __builtin_amdgcn_sched_barrier(0);
    const i32x4 yrs = skp_make_rsrc(a.y + zsplit * a.y_split_stride, a.y_bytes);
    if (PART) {
        w4_unroll([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int co = n0 + 4 * kq + r;
            w4_unroll([&](auto tbc) {
                constexpr int tb = decltype(tbc)::value;
                const bool ok = t_ok[tb] && co < a.Cout;
                const int vo = (o_base[tb] + co * HW) * 4;
                float t[6][4];
                w4_unroll([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    float m[6];
                    w4_unroll([&](auto jc) { constexpr int j = decltype(jc)::value; m[j] = w4c_acc<i * 6 + j, tb, r>(accv); },
                              std::make_integer_sequence<int, 6>{});
                    w4_out1d(m, t[i]);
                }, std::make_integer_sequence<int, 6>{});
                f32x4 o[4];
#pragma unroll
                for (int ox = 0; ox < 4; ++ox) {
                    float m[6], yv[4];
#pragma unroll
                    for (int i = 0; i < 6; ++i) m[i] = t[i][ox];
                    w4_out1d(m, yv);
#pragma unroll
                    for (int oy = 0; oy < 4; ++oy) o[oy][ox] = yv[oy];
                }
#pragma unroll
                for (int oy = 0; oy < 4; ++oy) skp_buf_store_f32x4(o[oy], yrs, ok ? vo + oy * a.W * 4 : SKP_OOB, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, NTB>{});
        }, std::make_integer_sequence<int, 4>{});
        W4R_STAMP(40);
#ifdef W4R_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4R_STAMP(41);
#endif
        return;
    }
    const i32x4 rrs = skp_make_rsrc(a.res, a.res ? a.y_bytes : 0u);
    const i32x4 brs = skp_make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
    f32x4 rr[4][2][4];
    float bvs[4];
    auto load_res = [&](int r) {
        const int co = n0 + 4 * kq + r;
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
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
    __builtin_amdgcn_sched_barrier(0);
    w4_unroll([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int co = n0 + 4 * kq + r;
        const float bv = bvs[r];
        if (r + 1 < 4) load_res(r + 1);
        w4_unroll([&](auto tbc) {
            constexpr int tb = decltype(tbc)::value;
            const bool ok = t_ok[tb] && co < a.Cout;
            const int vo = (o_base[tb] + co * HW) * 4;
            float t[6][4];
            w4_unroll([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float m[6];
                w4_unroll([&](auto jc) { constexpr int j = decltype(jc)::value; m[j] = w4c_acc<i * 6 + j, tb, r>(accv); },
                          std::make_integer_sequence<int, 6>{});
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
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, NTB>{});
    }, std::make_integer_sequence<int, 4>{});
}

// ---- 128-channel form (Cout % 128 == 0): a wave owns 32 output channels x 16 tiles x 36 positions (same 288
// accumulator registers), the workgroup 128 channels on 16 tiles.  The input transform -- exposed VALU work with one
// wave per SIMD -- is done for half as many tiles per unit of MFMA work (one patch per thread and stage instead of
// two), at the price of two filter operand loads per position instead of one.  LDS stage: [36][4][16] f32x4 = 36 KB.
constexpr int W4C_STAGE_F4 = 36 * 4 * 16;
constexpr int W4C_RING = 6;                      // ring slots per channel block (divides 36)
// GNF: the input is x, not silu(GroupNorm(x)): every patch value goes through v = x * scale[b,c] + shift[b,c], v / (1 + e^-v)
// on its way into the input transform (one pass over the activation saved per norm: the GroupNorm apply kernel disappears).
// Zero padding stays zero: out-of-image rows / columns use (scale, shift) = (0, 0) and silu(0) = 0.
// Persistent: a workgroup walks the work ids  blockIdx.x, + gridDim.x, ...  (w4_work's order, so the co-resident workgroups of
// an XCD stay on one band).  While the LAST stage of a unit runs, its side jobs load and transform the first patches of the
// NEXT unit (the transform role is retargeted before that stage), so per unit only the epilogue and the filter ring refill
// are left outside the MFMA loop -- with 8 stages per unit (128 input channels) the prologue + dispatch gap was ~25 % of it.
template <bool STATS, bool GNF = false>
__global__ __launch_bounds__(256, 1) void skp_wino4_conv_c128_kernel(Wino4Args a) {
    extern __shared__ f32x4 vst[];                   // [2][36][4][16] stage buffers, then [32][64] float2 statistics slots
    f32x2* const sst = (f32x2*)(vst + 2 * W4C_STAGE_F4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int HW = a.H * a.W;
    int tblock, cg, zsplit;
    int wid = blockIdx.x;
    while (wid < a.vtotal && !w4_work(a, wid, tblock, cg, zsplit)) wid += gridDim.x;
    if (wid >= a.vtotal) return;

    // ---- transform role: the 6x6 patch of one channel of one tile ----
    const int tl = tid & 15, tc = tid >> 4;          // tile in the block, channel in the stage
    int roff[6];
    bool lok, rok;
    int edelta = 0;                                  // byte offset of the halo value an END lane of the block loads (0: none)
    int coff = SKP_OOB;                              // GNF: per-(image, channel) coefficients of the stage's channel `tc`
    f32x2 mrow = {1.f, 1.f};                         // GNF: shift masks of patch rows 0 / 5 (0 outside the image) ...
    f32x2 gh_edge = {0.f, 0.f};                      // per stage: shift of the (c0, c5) pair in rows 1..4
    unsigned rowmask = 0;
    auto aim_transform = [&](int tb, bool valid) {   // point the transform role at tile block tb (nothing: every load returns 0)
        const int tg = tb * 16 + tl;
        const bool tv = valid && tg < a.nTiles;
        const int tgc = tv ? tg : 0;
        const int b = tgc / a.tilesPerImg, rem = tgc - b * a.tilesPerImg;
        const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        const int base = (b * a.Cin + tc) * HW + 4 * tx;
        rowmask = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int r = 4 * ty - 1 + i;
            roff[i] = (tv && r >= 0 && r < a.H) ? (base + r * a.W) * 4 : SKP_OOB;
            rowmask |= (roff[i] != SKP_OOB ? 1u : 0u) << i;
        }
        lok = tx > 0;
        rok = tx + 1 < a.tilesX;
        // halo columns: the neighbour tiles of a patch row sit in the neighbour lanes (16 consecutive tiles of a tile row per wave
        // row), so columns 0 / 5 come from their aligned loads by DPP; only the block's two END lanes load a halo value (one
        // masked load per patch row instead of two full ones)
        edelta = tl == 0 ? (lok ? -4 : 0) : (tl == 15 ? (rok ? 16 : 0) : 0);
        coff = valid ? (b * a.Cin + tc) * 8 : SKP_OOB;
        if (GNF) {
            mrow = f32x2{(rowmask & 1u) ? 1.f : 0.f, (rowmask & 32u) ? 1.f : 0.f};
        }
    };
    aim_transform(tblock, true);
    int cin_ld = zsplit * a.steps * 16;              // first input channel of the unit whose patches are being loaded
    const i32x4 xrs = skp_make_rsrc(a.x, a.x_bytes);
    const i32x4 urs = skp_make_rsrc(a.U, a.u_bytes);
    const i32x4 crs = skp_make_rsrc(a.gncoef, GNF ? (unsigned)a.B * a.Cin * 8u : 0u);
    f32x2 gcoef = {0.f, 0.f};
    f32x2 d[6][3];                                   // [row][column pair]: (c0,c5), (c1,c2), (c3,c4)
    auto gn_fetch = [&](int cin0) { if (GNF) gcoef = skp_buf_load_f32x2(crs, coff, cin0 * 8, 0); };
    auto gn_prep = [&]() { if (GNF) gh_edge = f32x2{edelta != 0 ? gcoef[1] : 0.f, 0.f}; };
    // GNF: v = x * s + h;  silu(v) = v * rcp(1 + 2^(-v log2 e)) on value PAIRS (the patch registers are column pairs): two
    // packed fmas / muls, a packed add and a packed multiply per pair plus the four quarter-rate transcendentals -- an IEEE
    // division here is ten VALU instructions per value, all of them added to the MFMA time.  Zero padding must stay zero
    // although silu(0 * s + h) is not: a loaded 0 of an out-of-image row / column gets shift 0 as well.  Only patch rows 0 and 5
    // and patch columns 0 and 5 can lie outside the image, so the shift is masked per (row class, column class) with float
    // masks that are fixed per unit (aim_transform), four multiplies per stage instead of selects per row.
    // columns 0 / 5 of patch row i from the neighbour lanes' columns 4 / 1 (row_shr:1 / row_shl:1 inside the 16-lane tile group); the
    // end lanes keep what they loaded (d[i][0][0]); a tile at the image's left / right border gets 0 (zero padding)
    auto halo_row = [&](int i, float edge) {
        const float c4 = d[i][2][1], c1 = d[i][1][0];       // (by value: __builtin_bit_cast of a vector-element lvalue reads element 0)
        const float fromL = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c4), 0x111, 0xf, 0xf, true));
        const float fromR = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c1), 0x101, 0xf, 0xf, true));
        d[i][0][0] = tl == 0 ? edge : (lok ? fromL : 0.f);
        d[i][0][1] = tl == 15 ? edge : (rok ? fromR : 0.f);
    };
    auto gn_row = [&](int i) {                       // normalise + SiLU row i of the freshly loaded patch
        if (GNF) {
            const float rm = i == 0 ? mrow[0] : (i == 5 ? mrow[1] : 1.0f);
            const f32x2 S2 = {gcoef[0], gcoef[0]};
            const f32x2 Hm = {gcoef[1] * rm, gcoef[1] * rm};                 // rows 1..4: rm == 1 folds away
            auto act2 = [&](f32x2 x, f32x2 h) {
                const f32x2 v = x * S2 + h;
                const f32x2 z = v * (-SKP_LOG2E);
                const f32x2 w = f32x2{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + 1.0f;
                return v * f32x2{__builtin_amdgcn_rcpf(w[0]), __builtin_amdgcn_rcpf(w[1])};
            };
            d[i][1] = act2(d[i][1], Hm);
            d[i][2] = act2(d[i][2], Hm);
            // the end lanes' own halo value: one activation (the other lanes take their neighbours' activated columns); a value
            // outside the image is a loaded 0 and must stay 0: its shift is masked (gh_edge: shift x (lane has an in-image halo))
            const float he = (i == 0 || i == 5) ? gh_edge[0] * rm : gh_edge[0];
            const float v = d[i][0][0] * gcoef[0] + he;
            const float e = v * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v * (-SKP_LOG2E)) + 1.0f);
            halo_row(i, e);
        }
    };
    auto load_row = [&](int cin0, int i) {
        const int so = cin0 * HW * 4;
        const f32x4 mid = skp_buf_load_f32x4(xrs, roff[i], so, 0);
        d[i][0][0] = skp_buf_load_f32(xrs, (edelta != 0 && roff[i] != SKP_OOB) ? roff[i] + edelta : SKP_OOB, so, 0);
        d[i][1] = f32x2{mid[0], mid[1]};
        d[i][2] = f32x2{mid[2], mid[3]};
    };
    auto col_pass = [&](int k) {
        if (!GNF && k == 0) {                        // plain form: the halo pair is put together here, right before its first use
#pragma unroll
            for (int i = 0; i < 6; ++i) halo_row(i, d[i][0][0]);
        }
        f32x2 v[6], t[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = d[i][k];
        w4_in1d(v, t);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i][k] = t[i];
    };
    auto row_pass_store = [&](int buf, int i) {
        float r0[6] = {d[i][0][0], d[i][1][0], d[i][1][1], d[i][2][0], d[i][2][1], d[i][0][1]};
        float t[6];
        w4_in1d(r0, t);
        float* dst = (float*)(vst + buf * W4C_STAGE_F4) + (((i * 6) * 4 + (tc >> 2)) * 16 + tl) * 4 + (tc & 3);
#pragma unroll
        for (int j = 0; j < 6; ++j) dst[j * (4 * 16 * 4)] = t[j];
    };

    // first unit: its first patches the plain way
    gn_fetch(cin_ld);
#pragma unroll
    for (int i = 0; i < 6; ++i) load_row(cin_ld, i);
    gn_prep();
#pragma unroll
    for (int i = 0; i < 6; ++i) gn_row(i);
#pragma unroll
    for (int k = 0; k < 3; ++k) col_pass(k);
#pragma unroll
    for (int i = 0; i < 6; ++i) row_pass_store(0, i);
    int bpar = 0;                                    // stage s of the current unit lives in buffer (bpar + s) & 1

    const int C16 = a.Cin >> 4;
    const int u_c16 = 4 * a.Cout * 16, u_p = C16 * u_c16;
    const i32x4 rrs = skp_make_rsrc(a.res, a.res ? a.y_bytes : 0u);
    const i32x4 brs = skp_make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);

    for (;;) {
        const int tile0 = tblock * 16;
        const int n0 = (cg * 4 + wave) * 32;         // this wave's 32 output channels (two 16-row MFMA blocks)
        const int nsteps = min(a.steps, a.total_steps - zsplit * a.steps);
        const int cin_begin = zsplit * a.steps * 16;
        int uvo[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) uvo[cb] = (kq * a.Cout + min(n0 + cb * 16 + i16, a.Cout - 1)) * 16;
        f32x4 ua[W4C_RING][2];
#pragma unroll
        for (int q = 0; q < W4C_RING - 1; ++q)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) ua[q][cb] = skp_buf_load_f32x4(urs, uvo[cb], (cin_begin >> 4) * u_c16 + q * u_p, 0);

        // the unit after this one
        int wnext = wid + gridDim.x, tb_n = 0, cg_n = 0, z_n = 0;
        while (wnext < a.vtotal && !w4_work(a, wnext, tb_n, cg_n, z_n)) wnext += gridDim.x;
        const bool has_next = wnext < a.vtotal;

        // accumulators [position][channel block]: positions 0-31 by NAME in a[0:255] (w4c_mfma), 32-35 in VGPR tuples
        w4_unroll([&](auto rc) { asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(decltype(rc)::value) : W4_AGPR_CLOBBERS); },
                  std::make_integer_sequence<int, 256>{});
        f32x4 accv[4][2];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) accv[p][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();

        // MODE 0: a stage of the unit, loading / transforming the unit's next stage on the side; MODE 1: the unit's last stage,
        // the side jobs work on the first stage of the NEXT unit (the transform role already points there; nothing -> zeros)
        auto run_stage = [&](int s, auto mode_c) {
            constexpr int MODE = decltype(mode_c)::value;
            const f32x4* vb = vst + ((bpar + s) & 1) * W4C_STAGE_F4 + kq * 16 + i16;
            const int ub = ((cin_begin >> 4) + s) * u_c16;
            const int cin_side = MODE == 0 ? cin_ld + (s + 1) * 16 : cin_ld;
            f32x4 va[3];
            va[0] = vb[0];
            va[1] = vb[64];
            w4_unroll([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                {
                    constexpr int D = W4C_RING - 1;
                    constexpr int q = p + D;
                    if (MODE == 0 || q < 36) {
                        const int uo = q < 36 ? ub + q * u_p : ub + u_c16 + (q - 36) * u_p;
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) ua[q % W4C_RING][cb] = skp_buf_load_f32x4(urs, uvo[cb], uo, 0);
                    }
                }
                if (p == 0) gn_fetch(cin_side);
                if (p < 6) load_row(cin_side, p);
                else if (GNF && p == 20) gn_prep();
                else if (GNF && p >= 21 && p < 27) gn_row(p - 21);
                else if (p >= 27 && p < 30) col_pass(p - 27);
                else if (p >= 30) row_pass_store((bpar + s + 1) & 1, p - 30);
                if (p + 2 < 36) va[(p + 2) % 3] = vb[(p + 2) * 64];
                w4_unroll([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    w4c_mfma<p, 0>(accv, ua[p % W4C_RING][0][m], va[p % 3][m]);
                    w4c_mfma<p, 1>(accv, ua[p % W4C_RING][1][m], va[p % 3][m]);
                }, std::make_integer_sequence<int, 4>{});
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, 36>{});
        };
        for (int s = 0; s + 1 < nsteps; ++s) {
            run_stage(s, std::integral_constant<int, 0>{});
            __syncthreads();
        }
        aim_transform(tb_n, has_next);               // the unit's own patches are all loaded: retarget the transform role
        cin_ld = z_n * a.steps * 16;
        run_stage(nsteps - 1, std::integral_constant<int, 1>{});

        // ---- epilogue: output role (lane = tile of the block, registers = 4 output channels per channel block) ----
        int o_base;
        bool t_ok;
        {
            const int tg = tile0 + i16;
            t_ok = tg < a.nTiles;
            const int tgc = t_ok ? tg : 0;
            const int b = tgc / a.tilesPerImg, rem = tgc - b * a.tilesPerImg;
            const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
            o_base = ((b * a.Cout) * a.H + 4 * ty) * a.W + 4 * tx;
        }
        const i32x4 yrs = skp_make_rsrc(a.y + zsplit * a.y_split_stride, a.y_bytes);
        f32x4 rr[8][4];                              // [channel block * 4 + r][output row]
        float bvs[8];
        auto load_res = [&](int e) {
            const int co = n0 + (e >> 2) * 16 + 4 * kq + (e & 3);
            const bool ok = t_ok && co < a.Cout;
            const int vo = (o_base + co * HW) * 4;
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) rr[e][oy] = skp_buf_load_f32x4(rrs, ok ? vo + oy * a.W * 4 : SKP_OOB, 0, 0);
        };
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = n0 + (e >> 2) * 16 + 4 * kq + (e & 3);
            bvs[e] = skp_buf_load_f32(brs, co < a.Cout ? co * 4 : SKP_OOB, 0, 0);
        }
        constexpr int RD = GNF ? 2 : 3;              // residual rows in flight (the GNF forms carry more state across the epilogue)
#pragma unroll
        for (int e = 0; e < RD; ++e) load_res(e);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results are in the register file before the first read
        __builtin_amdgcn_sched_barrier(0);
        w4_unroll([&](auto ec) {
            constexpr int e = decltype(ec)::value, cb = e >> 2, r = e & 3;
            const int co = n0 + cb * 16 + 4 * kq + r;
            const bool ok = t_ok && co < a.Cout;
            const int vo = (o_base + co * HW) * 4;
            const float bv = bvs[e];
            if (e + RD < 8) load_res(e + RD);
            float t[6][4];
            w4_unroll([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float m[6];
                w4_unroll([&](auto jc) { constexpr int j = decltype(jc)::value; m[j] = w4c_acc<i * 6 + j, cb, r>(accv); },
                          std::make_integer_sequence<int, 6>{});
                w4_out1d(m, t[i]);
            }, std::make_integer_sequence<int, 6>{});
#pragma unroll
            for (int ox = 0; ox < 4; ++ox) {
                float m[6], yv[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) m[i] = t[i][ox];
                w4_out1d(m, yv);
#pragma unroll
                for (int oy = 0; oy < 4; ++oy) rr[e][oy][ox] += yv[oy] + bv;
            }
#pragma unroll
            for (int oy = 0; oy < 4; ++oy) skp_buf_store_f32x4(rr[e][oy], yrs, ok ? vo + oy * a.W * 4 : SKP_OOB, 0, 0);
            if (STATS) w4_park_stats(sst, wave * 8 + e, lane, rr[e], ok);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 8>{});
        if (STATS) {                                 // 128 channels of one tile block
            __syncthreads();
            if (tid < 128) {
                const int wv = tid >> 5, lc = tid & 31;
                w4_store_stats(a, sst, wv * 8 + ((lc >> 4) << 2) + (lc & 3), (lc >> 2) & 3, tile0, (cg * 4 + wv) * 32 + lc);
            }
        }
        if (!has_next) break;
        bpar = (bpar + nsteps) & 1;
        wid = wnext; tblock = tb_n; cg = cg_n; zsplit = z_n;
    }
}

// 128-channel workgroup form where there are enough tiles (measured: 12-16 % faster on the VAE / 32^2 UNet layers; at 16^2
// (128 tiles at 8 rows) 5-18 % faster once the K splits fill the CUs evenly, at 8^2 no better than the 64-channel form:
// profiles/r03_conv_splits.md).  A ragged last channel group is fine (clamped filter
// rows, guarded stores): the 320-channel UNet layers run 3 groups (17 % idle MFMA rows) and still gain from the form's
// shorter stages (one patch per thread instead of two).
static bool wino4_use_c128(int Cout, int tiles) {
    return tiles >= 128 && (Cout % 128 == 0 || (Cout > 128 && Cout % 128 >= 64));
}

// Workgroup grid of a launch with S K-splits (the order is explained at w4_work) and the number of waves of workgroups it
// makes on the 256 CUs (one workgroup per CU: LDS).  In the unit-grouped order an XCD's 32 CUs take ceil(units / 8) units.
static Wino4Grid wino4_grid(int Cout, int tiles, int S) {
    Wino4Grid g;
    const bool c128 = wino4_use_c128(Cout, tiles);
    g.ntb = c128 ? (tiles + 15) / 16 : (tiles + 31) / 32;
    g.ncg = c128 ? (Cout + 127) / 128 : (Cout + 63) / 64;
    g.tb_per_xcd = g.ntb >= (c128 ? 64 : 32) ? (g.ntb + 7) / 8 : 0;
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

// K splits of a launch: S workgroups share the Cin / 16 stages of one (tile block, channel group) unit, ceil(stages / S) each
// (the last one takes what is left), partial outputs summed by skp_wino4_reduce_kernel in split order.
int wino4_plan(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    if ((Cin % 16) || (Cout % 16) || (H % 4) || (W % 4)) return 0;
    const int tiles = B * (H / 4) * (W / 4);
    const bool c128 = wino4_use_c128(Cout, tiles);
    const int nsteps = Cin / 16;
    const double out_bytes = (double)B * Cout * H * W * 4;
    if (const int S = skp_tune(SKP_TUNE_WINO_SPLIT)) {           // tests / tools: force the K split
        if (S >= 1 && S <= 16 && (S - 1) * ((nsteps + S - 1) / S) < nsteps) return S;
    }
    int best = 1;
    double best_cost = 1e30;
    // measured stage times (us): the 128-channel form ~3.9, the 64-channel form ~5.6 (tools/conv_bench.py with SKP_WINO_SPLIT
    // forced); the reduce pass streams (S + 1) x the output at ~8 TB/s (the partials are L2 / MALL resident)
    // (ragged channel counts -- 320 = 2.5 groups of 128 -- pay a full stage for the half-idle group: 4.8 us per stage and a K split
    // already at 5 % gain, tools/split_sweep.py, round 6: 320 -> 320 @64^2 at 8 rows S = 1 215 us, S = 2 195 us)
    const bool ragged_c128 = c128 && (Cout % 128) != 0;
    const double stage_us = c128 ? (ragged_c128 ? 4.8 : 3.9) : 5.6;
    for (int S = 1; S <= 16; ++S) {
        const int per = (nsteps + S - 1) / S;
        if ((S - 1) * per >= nsteps) continue;                   // an empty last split
        const Wino4Grid g = wino4_grid(Cout, tiles, S);
        double cost = g.rounds * (per + 2.0) * stage_us;       // ~2 stages of prologue + epilogue per workgroup
        if (S > 1) cost += 6.0 + (S + 1) * out_bytes / 8.0e6;
        if (cost < best_cost * (S > 1 ? (ragged_c128 ? 0.95 : 0.92) : 1.0)) { best_cost = cost; best = S; }
    }
    return best;
}

// ---- raw-filter form: geometry (always 64 channels x 32 tiles), K-split plan, gate ----
static Wino4Grid wino4r_grid(int Cout, int tiles, int S) {
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
// Where the form pays (tools/conv_raw_bench.py, profiles/r04_conv_raw.md): >= 1280 channels on both sides at up to 1536 tiles
// (the UNet's 8^2 / 16^2 levels and 1280 -> 1280 at 32^2, measured at 4 .. 40 rows: -12 .. -26 %), and up to 512 tiles when one
// side has >= 1920 channels (the 32^2 skip-connection layers: -7 %); the 640-channel layers stay on the transformed-filter
// kernels (equal or faster there at the step's 8 rows).
// skp_tune_set("wino_raw_max_tiles", n): every qualifying shape up to n tiles (tools/conv_raw_bench.py).
static int wino4r_max_tiles() { return skp_tune(SKP_TUNE_WINO_RAW_MAX_TILES); }
static bool wino4r_layout_ok(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return false;
    if ((Cin % 16) || (Cout % 64) || (H % 4) || (W % 4)) return false;
    const long long tiles = (long long)B * (H / 4) * (W / 4);
    const long long vpad = (tiles + 31) / 32 * 32;
    return 36ll * Cin * vpad * 4 < 0x7fffffffll && 9ll * Cin * Cout * 4 < 0x7fffffffll && (long long)B * Cout * H * W * 4 < 0x7fffffffll;
}
static bool wino4r_shape_ok(int B, int Cin, int Cout, int H, int W) {
    if (!wino4r_layout_ok(B, Cin, Cout, H, W)) return false;
    const long long tiles = (long long)B * (H / 4) * (W / 4);
    if (wino4r_max_tiles() > 0) return tiles <= wino4r_max_tiles() && Cin >= 256;
    if (Cin >= 1280 && Cout >= 1280) return tiles <= 1536;
    return tiles <= 512 && (Cin >= 1920 || Cout >= 1920) && Cin >= 640 && Cout >= 640;
}
constexpr double W4R_STAGE1_US = 2.7;
static int wino4r_plan(int B, int Cin, int Cout, int H, int W) {
    const int tiles = B * (H / 4) * (W / 4);
    const int nsteps = Cin / 16;
    const double out_bytes = (double)B * Cout * H * W * 4;
    if (const int S = skp_tune(SKP_TUNE_WINO_SPLIT)) {
        if (S >= 1 && S <= 16 && (S - 1) * ((nsteps + S - 1) / S) < nsteps) return S;
    }
    int best = 1;
    double best_cost = 1e30;
    // measured (cycle stamps, W4R_STAMPS builds): 4.8 us per stage, ~2.1 us prologue + ~5.3 us epilogue + dispatch per workgroup
    // (one 16-tile block per wave, <= 16 tiles: half the MFMAs per stage, half the epilogue)
    const double stage_us = tiles <= 16 ? W4R_STAGE1_US : 4.8, over = tiles <= 16 ? 2.0 : 2.5;
    for (int S = 1; S <= 16; ++S) {
        const int per = (nsteps + S - 1) / S;
        if ((S - 1) * per >= nsteps) continue;
        const Wino4Grid g = wino4r_grid(Cout, tiles, S);
        double cost = g.rounds * (per + over) * stage_us;
        if (S > 1) cost += 6.0 + (S + 1) * out_bytes / 8.0e6;
        if (cost < best_cost * (S > 1 ? 0.97 : 1.0)) { best_cost = cost; best = S; }
    }
    return best;
}

}  // namespace

extern "C" int skp_conv3x3_f4_filter_f32(const void* w, void* U, int Cout, int Cin, int flip_transpose, void* stream) {
    if (!w || !U || Cout <= 0 || Cin <= 0) return SKP_E_BADARG;
    if (Cin & 15) return SKP_E_RANGE;
    const int n = Cout * Cin;
    hipLaunchKernelGGL(skp_wino4_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                       (float*)U, Cout, Cin, flip_transpose);
    return skp_launch_status();
}

extern "C" int64_t skp_conv3x3_f4_workspace(int B, int Cin, int Cout, int H, int W) {
    const int S = wino4_plan(B, Cin, Cout, H, W);
    return S > 1 ? (int64_t)S * B * Cout * H * W * (int64_t)sizeof(float) : 0;
}

// 16-tile blocks per image when the kernels can emit output statistics for this launch (unsplit, blocks do not straddle
// images), else 0
extern "C" int skp_conv3x3_f4_stats_blocks(int B, int Cin, int Cout, int H, int W) {
    const int S = wino4_plan(B, Cin, Cout, H, W);
    if (S != 1) return 0;
    const int tpi = (H / 4) * (W / 4);
    return (tpi % 32 == 0) ? tpi / 16 : 0;          // 32: the 64-channel form walks two 16-tile blocks per workgroup
}

static int wino4_run(const void* x, const void* U, const void* bias, const void* residual, void* y, void* workspace, float* stats,
                     int B, int Cin, int Cout, int H, int W, void* stream, const float* gncoef = nullptr);

// 1 when skp_conv3x3_f4_gn_f32 serves this launch: 128-channel workgroup form, unsplit, at most four channel groups.  Every
// channel group redoes the SiLU of its input patches; since the accumulators are named (no spills in the folded kernel) that
// costs less than the separate GroupNorm apply pass (one read + one write of the activation) up to the VAE's 512-channel levels:
// 256 -> 256 @256^2 1 794 -> 1 657 us, 512 -> 512 @128^2 1 580 -> 1 493, 512 -> 512 @64^2 436 -> 401 (tools/gn_fold_bench.py;
// rounds 3-4 gated at ONE group: profiles/r03_conv_gn_fold.md).  skp_tune_set("gn_fold_max_cout", n): A/B runs.
extern "C" int skp_conv3x3_f4_gn_ok(int B, int Cin, int Cout, int H, int W) {
    if (wino4_plan(B, Cin, Cout, H, W) != 1) return 0;
    const int tiles = B * (H / 4) * (W / 4);
    const int max_cout = skp_tune(SKP_TUNE_GN_FOLD_MAX_COUT) ? skp_tune(SKP_TUNE_GN_FOLD_MAX_COUT) : 512;
    return (wino4_use_c128(Cout, tiles) && Cout <= max_cout) ? 1 : 0;
}

// y = conv3x3(silu(x * scale[b,c] + shift[b,c])) (+ bias) (+ residual): the GroupNorm(+offset)+SiLU in front of the
// convolution applied inside the patch load (coef: [B,Cin,2] from skp_group_norm_coef_*); optional output block sums as
// skp_conv3x3_f4_stats_f32.  Forward only (the VAE encoder runs without autograd).
extern "C" int skp_conv3x3_f4_gn_f32(const void* x, const void* U, const void* bias, const void* residual, void* y,
                                     float* stats, const float* coef, int B, int Cin, int Cout, int H, int W, void* stream) {
    if (!coef) return SKP_E_BADARG;
    if (!skp_conv3x3_f4_gn_ok(B, Cin, Cout, H, W)) return SKP_E_RANGE;
    if (stats && skp_conv3x3_f4_stats_blocks(B, Cin, Cout, H, W) == 0) return SKP_E_RANGE;
    return wino4_run(x, U, bias, residual, y, nullptr, stats, B, Cin, Cout, H, W, stream, coef);
}

extern "C" int skp_conv3x3_f4_f32(const void* x, const void* U, const void* bias, const void* residual, void* y,
                                  void* workspace, int B, int Cin, int Cout, int H, int W, void* stream) {
    return wino4_run(x, U, bias, residual, y, workspace, nullptr, B, Cin, Cout, H, W, stream);
}

extern "C" int skp_conv3x3_f4_stats_f32(const void* x, const void* U, const void* bias, const void* residual, void* y,
                                        float* stats, int B, int Cin, int Cout, int H, int W, void* stream) {
    if (!stats || skp_conv3x3_f4_stats_blocks(B, Cin, Cout, H, W) == 0) return SKP_E_RANGE;
    return wino4_run(x, U, bias, residual, y, nullptr, stats, B, Cin, Cout, H, W, stream);
}

static int wino4_run(const void* x, const void* U, const void* bias, const void* residual, void* y, void* workspace, float* stats,
                     int B, int Cin, int Cout, int H, int W, void* stream, const float* gncoef) {
    if (!x || !U || !y) return SKP_E_BADARG;
    int S = wino4_plan(B, Cin, Cout, H, W);
    if (S == 0) return (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) ? SKP_E_BADARG : SKP_E_RANGE;
    if (!workspace) S = 1;
    const unsigned long long xb = (unsigned long long)B * Cin * H * W * 4, ub = (unsigned long long)36 * Cin * Cout * 4,
                             yb = (unsigned long long)B * Cout * H * W * 4;
    if (xb >= 0x80000000ull || ub >= 0x80000000ull || yb >= 0x80000000ull) return SKP_E_RANGE;
    Wino4Args a;
    a.x = (const float*)x; a.U = (const float*)U;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.tilesX = W / 4;
    a.tilesPerImg = a.tilesX * (H / 4);
    a.nTiles = B * a.tilesPerImg;
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb;
    a.total_steps = Cin / 16;
    a.steps = (a.total_steps + S - 1) / S;
    a.splits = S;
    const size_t out_elems = (size_t)B * Cout * H * W;
    a.y_split_stride = out_elems;
    a.y = S > 1 ? (float*)workspace : (float*)y;
    a.bias = S > 1 ? nullptr : (const float*)bias;
    a.res = S > 1 ? nullptr : (const float*)residual;
    a.stats = S > 1 ? nullptr : stats;
    a.sblk = a.tilesPerImg / 16;
    a.gncoef = gncoef;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * W4_STAGE_F4 * sizeof(f32x4), lds_c = (size_t)2 * W4C_STAGE_F4 * sizeof(f32x4) + 32 * 64 * sizeof(f32x2);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_wino4_conv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4_conv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4_conv_c128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4_conv_c128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4_conv_c128_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4_conv_c128_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const bool c128 = wino4_use_c128(Cout, a.nTiles);
    const Wino4Grid g = wino4_grid(Cout, a.nTiles, S);
    a.ntb = g.ntb; a.ncg = g.ncg; a.tb_per_xcd = g.tb_per_xcd;
    a.gx = (int)g.gx;
    a.vtotal = g.tb_per_xcd ? (int)g.gx * S : (int)g.gx;
    // 128-channel form: persistent workgroups, one per CU (the ids are multiples of 8 apart, so a workgroup stays on its XCD's band)
    static const int persist = [] {                 // one workgroup per CU of this device
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
        return ncu;
    }();
    const dim3 grid = c128 ? dim3((unsigned)(persist >= 8 ? std::min(a.vtotal, persist & ~7) : a.vtotal), 1, 1)
                           : (g.tb_per_xcd ? dim3(g.gx, 1, S) : dim3(g.gx, 1, 1));
    if (c128) {                                     // 128 channels x 16 tiles per workgroup
        if (gncoef) {
            if (a.stats) hipLaunchKernelGGL((skp_wino4_conv_c128_kernel<true, true>), grid, dim3(256), lds_c, st, a);
            else hipLaunchKernelGGL((skp_wino4_conv_c128_kernel<false, true>), grid, dim3(256), lds_c, st, a);
        } else if (a.stats) hipLaunchKernelGGL((skp_wino4_conv_c128_kernel<true>), grid, dim3(256), lds_c, st, a);
        else hipLaunchKernelGGL((skp_wino4_conv_c128_kernel<false>), grid, dim3(256), lds_c, st, a);
    } else if (gncoef) {
        return SKP_E_RANGE;
    } else {                                        // 64 channels x 32 tiles per workgroup
        if (a.stats) hipLaunchKernelGGL(skp_wino4_conv_kernel<true>, grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(skp_wino4_conv_kernel<false>, grid, dim3(256), lds, st, a);
    }
    int rc = skp_launch_status();
    if (rc || S == 1) return rc;
    const size_t n4 = out_elems / 4;
    hipLaunchKernelGGL(skp_wino4_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)workspace,
                       (const float*)bias, (const float*)residual, (float*)y, n4, out_elems, S, (H * W) / 4, Cout);
    return skp_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Raw-filter form (small-spatial layers): see skp_wino4r_conv_kernel.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int skp_conv3x3_f4r_ok(int B, int Cin, int Cout, int H, int W) {
    return wino4r_shape_ok(B, Cin, Cout, H, W) ? 1 : 0;
}

// R: 9 * Cin * Cout floats.  flip_transpose as skp_conv3x3_f4_filter_f32 (the backward-data filter of w[Cin][Cout][3][3]).
extern "C" int skp_conv3x3_f4r_filter_f32(const void* w, void* R, int Cout, int Cin, int flip_transpose, void* stream) {
    if (!w || !R || Cout <= 0 || Cin <= 0) return SKP_E_BADARG;
    if ((Cin & 15) || (Cout & 15)) return SKP_E_RANGE;
    const int n = Cout * Cin;
    hipLaunchKernelGGL(skp_wino4r_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                       (float*)R, Cout, Cin, flip_transpose);
    return skp_launch_status();
}

// bytes: the pre-transformed input (36 * Cin * padded tiles floats) followed by the K-split partial outputs
extern "C" int64_t skp_conv3x3_f4r_workspace(int B, int Cin, int Cout, int H, int W) {
    if (!wino4r_layout_ok(B, Cin, Cout, H, W)) return 0;
    const int64_t tiles = (int64_t)B * (H / 4) * (W / 4), vpad = (tiles + 31) / 32 * 32;
    const int S = wino4r_plan(B, Cin, Cout, H, W);
    return 36 * (int64_t)Cin * vpad * 4 + (S > 1 ? (int64_t)S * B * Cout * H * W * 4 : 0);
}

extern "C" int skp_conv3x3_f4r_f32(const void* x, const void* R, const void* bias, const void* residual, void* y, void* workspace,
                                   int B, int Cin, int Cout, int H, int W, void* stream) {
    if (!x || !R || !y || !workspace) return SKP_E_BADARG;
    if (!wino4r_layout_ok(B, Cin, Cout, H, W)) return (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) ? SKP_E_BADARG : SKP_E_RANGE;
    const int S = wino4r_plan(B, Cin, Cout, H, W);
    Wino4Args a;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.tilesX = W / 4;
    a.tilesPerImg = a.tilesX * (H / 4);
    a.nTiles = B * a.tilesPerImg;
    a.vpad = (a.nTiles + 31) / 32 * 32;
    const size_t v_elems = (size_t)36 * Cin * a.vpad;
    float* Vg = (float*)workspace;
    float* part = Vg + v_elems;
    a.x = Vg; a.U = (const float*)R;
    a.x_bytes = (unsigned)(v_elems * 4); a.u_bytes = (unsigned)((size_t)9 * Cin * Cout * 4);
    const size_t out_elems = (size_t)B * Cout * H * W;
    a.y_bytes = (unsigned)(out_elems * 4);
    a.total_steps = Cin / 16;
    a.steps = (a.total_steps + S - 1) / S;
    a.splits = S;
    a.y_split_stride = out_elems;
    a.y = S > 1 ? part : (float*)y;
    a.bias = S > 1 ? nullptr : (const float*)bias;
    a.res = S > 1 ? nullptr : (const float*)residual;
    a.stats = nullptr; a.sblk = 0; a.gncoef = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)2 * W4_STAGE_F4 * sizeof(f32x4);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_wino4r_conv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4r_conv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4r_conv_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)skp_wino4r_conv_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const Wino4Grid g = wino4r_grid(Cout, a.nTiles, S);
    a.ntb = g.ntb; a.ncg = g.ncg; a.tb_per_xcd = g.tb_per_xcd;
    a.gx = (int)g.gx;
    a.vtotal = g.tb_per_xcd ? (int)g.gx * S : (int)g.gx;
    const int nin = a.vpad * Cin;
    hipLaunchKernelGGL(skp_wino4r_input_kernel, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, st, (const float*)x, Vg, B, Cin, H, W,
                       a.tilesX, a.tilesPerImg, a.nTiles, a.vpad);
    const dim3 grid = g.tb_per_xcd ? dim3(g.gx, 1, S) : dim3(g.gx, 1, 1);
    if (a.nTiles <= 16) {                           // one 16-tile block per wave
        if (S > 1) hipLaunchKernelGGL((skp_wino4r_conv_kernel<true, 1>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((skp_wino4r_conv_kernel<false, 1>), grid, dim3(256), lds, st, a);
    } else if (S > 1) hipLaunchKernelGGL(skp_wino4r_conv_kernel<true>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(skp_wino4r_conv_kernel<false>, grid, dim3(256), lds, st, a);
    int rc = skp_launch_status();
    if (rc || S == 1) return rc;
    const size_t n4 = out_elems / 4;
    hipLaunchKernelGGL(skp_wino4_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)part,
                       (const float*)bias, (const float*)residual, (float*)y, n4, out_elems, S, (H * W) / 4, Cout);
    return skp_launch_status();
}

#ifdef W4R_STAMPS
extern "C" int skp_w4r_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w4r_stamps), sizeof(unsigned long long) * 128);
}
#endif
