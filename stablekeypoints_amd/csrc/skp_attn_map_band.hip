// Backward of the fused up-res attention map WITHOUT the dV staging (round 4): row bands, vertical adjoint in registers.
//
// Reference path replaced: autograd through ptp_utils.py:513-538 (bicubic x -> R^2, to_q, einsum, softmax) and
// optimize.py:27-79 (stack / mean), for the gradient the losses produce: non-zero on the K selected token rows only
// (optimize.py:395-414) -- the input of skp_attn_map_bwd_sparse_f32, whose T <= 128 route this is.
//
// The dense route (skp_attn_map.hip) applies the horizontal adjoint of the bicubic per up-res ROW through an LDS transpose +
// gather and stages dV[b,l,h,y,c,t] (210 MB at the step's shape) for a second kernel that applies the vertical adjoint:
// 1.09 GB of HBM-side traffic for 149 MB of algorithmic bytes.  Here the order of the two (separable) adjoints is swapped:
//
//   lane = up-res column x, a thread sweeps 2 groups of K2 = k/2 rows (k = R / s; all rows of a group share floor(src_y), so
//   their four vertical taps land in the same four low-res rows) and accumulates  acc[row slot][token] += wy * f  in REGISTERS
//   (vertical adjoint, no cross-lane traffic);  f_t = p_t (g_t - sum_k p_k g_k),  p_t = exp2(bicubic(S)_t - lse);
//   only then, once per band and low-res row instead of once per up-res row, the horizontal adjoint: lanes that share their
//   four tap columns (aligned groups of k/2 pixels) pre-reduce with DPP, the group sums cross LDS once and a column's
//   <= 9 contributions are added in a fixed order (index lists, no atomics: bit-reproducible).
//
// Tokens go through in chunks of 16 (registers): FIRST the K selected tokens (their logits gathered by token id), which
// gives dot = sum_k p_k g_k per pixel (kept in registers for the thread's rows) and the +p_k g_k part of the gradient; THEN
// the natural 16-token chunks with f_t = -p_t dot (a selected token gets both parts, added by the combine kernel).
// A workgroup (512 threads) = (batch row, layer, head, band of NP = 512 / R low-res rows); rows of neighbouring bands overlap
// in the low-res rows they touch, so a band writes NP + 4 partial rows and skp_map_band_combine_kernel adds them in band
// order, maps the out-of-image tap rows onto the clamped border rows (upsample_bicubic2d clamps tap indices on access) and
// scatters the selected tokens' part.  Partials: 2 x dS bytes at R = 128 (88 MB) instead of the 210 MB dV staging.
//
// STATUS (profiles/r04_map_band.md): parity green, HBM-side traffic down, but SLOWER than the dense route at the step's shape
// (778 vs 541 us: twelve 8-token chunks x (V phase, sweep, eight exchange rounds) are a long chain of barrier-separated
// phases, and a thread's 8 rows only fold into 5 row slots before the exchange) -- opt-in (SKP_MAP_BWD=band), not the default.
//
// Shapes served (skp_attn_map_bwd_band_ok): R in {128, 256}, every layer R = k s with k in {4, 8}, s % (512 / R) == 0,
// T <= 128, K <= 16 -- the SD-1.x path at feature_upsample_res 128 / 256; everything else keeps the other routes.
#include "skp_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BD_TC = 8;             // tokens per chunk (registers: acc[6][8]; 16 spilled hundreds of registers)
constexpr int BD_TS = BD_TC + 4;     // token stride of the V buffer (16-byte aligned quads, conflict-free column gathers)
constexpr int BD_PS = BD_TC + 1;     // token stride of the exchange buffer
constexpr int BD_LCAP = 16;          // list slots per low-res column (<= 9 used)
constexpr int BD_KMAX = 16;
constexpr int BD_LUSE = 10;          // list entries the gather reads per column (<= 9 real ones + padding that points at a zero slot)
constexpr int VBN = BD_TC / 4;      // V phase: the thread's items of a natural chunk, all loads in flight together
#ifndef BD_WAVES
#define BD_WAVES 4                   // waves per SIMD the register budget is held to
#endif

struct BandLayerArgs {
    const float* S;                  // [B,H,s*s,ldt] logits (pre-multiplied by scale*log2 e)
    float* P;                        // partials [B,H,NB,NR,s,NTP]
    int s, l, nb;                    // side, layer index (lse), bands per image
    int blk0;                        // first workgroup (per batch row) of this layer in the launch
};
struct BandArgs {
    BandLayerArgs ly[SKP_MAX_LAYERS];
    int nl, blocks_per_b;
    const int64_t* sel;              // [B,K]
    const float* G;                  // [B,K,R,R]
    const float* lse;                // [B,L*H,R*R] (log2 domain)
    int L, B, H, T, R, K, ldt, NT, NTP;
    float inv_lh;
};

template <int K> __device__ __forceinline__ float bd_quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float bd_pair_swap(float v) {      // value of the other lane of an aligned lane pair
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
// acc += (src of quad lane K) * w as one VOP2-DPP instruction; `src` must come from LDS / VMEM (see skp_attn_map.hip)
template <int K> __device__ __forceinline__ void bd_quad_fmac(float& acc, float src, float w) {
    if (K == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(w));
    if (K == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(w));
    if (K == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(w));
}

// K2 = k/2 rows per group = lanes per tap-sharing group (4: quads, 2: pairs); NP = 512 / R row parts per workgroup: thread =
// (column x, part p), a part = 2 groups = one low-res row of centres, a band = NP low-res rows; 512 threads.
template <int K2, int NP>
__global__ __launch_bounds__(512, BD_WAVES) void skp_map_bwd_band_kernel(BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RPT = 2 * K2;                  // rows per thread (two groups)
    constexpr int BRN = NP * RPT;                // rows per band
    constexpr int NR = NP + 4;                   // low-res rows a band touches (part p: band rows p .. p + 4)
    constexpr int E = (K2 == 2) ? 2 : 1;         // exchange entries per lane
    constexpr int R = 512 / NP, RR = R * R;      // the map side and the layer side are fixed by (K2, NP): every LDS offset
    constexpr int s = R / (2 * K2);              // of the sweep is an immediate (a runtime side made the compiler hoist
    constexpr int XPE = R * E + 1;                // exchange entries per part (+1: the zero slot list padding points at)
    constexpr int VBF = (512 * BD_TS > NP * XPE * BD_PS) ? 512 * BD_TS : NP * XPE * BD_PS;   // hundreds of addresses and spill)
    const int tid = threadIdx.x;
    const int H = a.H;
    const int b = blockIdx.x % a.B;              // batch row fastest: workgroup id % 8 (XCD) == b % 8
    int rest = blockIdx.x / a.B;
    int li = 0;
    while (li + 1 < a.nl && rest >= a.ly[li + 1].blk0) ++li;
    const BandLayerArgs ly = a.ly[li];
    rest -= ly.blk0;
    const int m = rest % ly.nb, h = rest / ly.nb;
    const int xcol = tid & (R - 1), part = tid / R;
    const int y0 = m * BRN;
    const float ratio = (float)s / (float)R;

    float* Vb = smem;                                         // [BRN*s = 512][BD_TS]; the exchange buffer aliases it
    float* tab_wy = smem + VBF;                               // [BRN][4]
    int* tab_cy = (int*)(tab_wy + BRN * 4);                   // [BRN][4]
    int* lst = tab_cy + BRN * 4;                              // [s][BD_LCAP]
    int* cnt = lst + s * BD_LCAP;                             // [s]
    int* selk = cnt + s;                                      // [BD_KMAX]

    // ---- per-pass tables ------------------------------------------------------------------------------------------
    if (tid < BRN) {
        int cy[4]; float wy[4];
        skp_cubic_taps(y0 + tid, ratio, s, cy, wy);
#pragma unroll
        for (int j = 0; j < 4; ++j) { tab_cy[tid * 4 + j] = cy[j]; tab_wy[tid * 4 + j] = wy[j]; }
    }
    if (tid < s) {       // exchange entries whose tap column is `tid`, ascending (fixed summation order)
        int n = 0;
        const int ngrp = R / K2;
        for (int g = 0; g < ngrp; ++g) {
            const int fl = (g - 1) >> 1;                      // floor(src_x) of the group's pixels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int c = fl - 1 + i;
                c = c < 0 ? 0 : (c > s - 1 ? s - 1 : c);
                if (c == tid && n < BD_LUSE) lst[tid * BD_LCAP + n++] = (K2 == 2) ? 2 * (2 * g + (i & 1)) + (i >> 1) : 4 * g + i;
            }
        }
        cnt[tid] = n;
        for (; n < BD_LUSE; ++n) lst[tid * BD_LCAP + n] = R * E;   // padding: the zero slot behind a part's entries
    }
    if (tid < BD_KMAX) selk[tid] = tid < a.K ? (int)a.sel[(size_t)b * a.K + tid] : 0;

    int cx[4]; float wx[4];
    skp_cubic_taps(xcol, ratio, s, cx, wx);
    f32x2 wx2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { cx[i] *= BD_TS; wx2[i] = f32x2{wx[i], wx[i]}; }     // cx: float offset of the tap column in a V row
    // weights of the pre-reduction: what each lane of my group gives to the tap column(s) I keep
    float wq[4];
    if (K2 == 4) {
        const int me = tid & 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c0 = bd_quad_bcast<0>(wx[k]), c1 = bd_quad_bcast<1>(wx[k]);
            const float c2 = bd_quad_bcast<2>(wx[k]), c3 = bd_quad_bcast<3>(wx[k]);
            if (me == k) { wq[0] = c0; wq[1] = c1; wq[2] = c2; wq[3] = c3; }
        }
    } else {             // pairs: I keep taps me and me + 2; wq = {own tap me, partner tap me, own tap me+2, partner tap me+2}
        const int me = tid & 1;
        const float p0 = bd_pair_swap(wx[0]), p1 = bd_pair_swap(wx[1]), p2 = bd_pair_swap(wx[2]), p3 = bd_pair_swap(wx[3]);
        wq[0] = me ? wx[1] : wx[0]; wq[1] = me ? p1 : p0;
        wq[2] = me ? wx[3] : wx[2]; wq[3] = me ? p3 : p2;
    }

    const int lh = ly.l * H + h;
    const float* Sg = ly.S + ((size_t)(b * H + h) * s * s) * a.ldt;
    const float* lse_g = a.lse + ((size_t)b * a.L * H + lh) * RR;
    float lse_r[RPT], dot_r[RPT];
#pragma unroll
    for (int ri = 0; ri < RPT; ++ri) {
        lse_r[ri] = lse_g[(size_t)(y0 + part * RPT + ri) * R + xcol];
        dot_r[ri] = 0.f;
    }
    float* Pg = ly.P + ((size_t)((b * H + h) * ly.nb + m) * NR) * s * a.NTP;
    const i32x4 grs = skp_make_rsrc(a.G + (size_t)b * a.K * RR, (unsigned)a.K * RR * 4u);      // this batch row's K gradient rows
    const int gvo = ((y0 + part * RPT) * R + xcol) * 4;
    const int vbase = (part * RPT * s) * BD_TS;               // my rows in the V buffer
    __syncthreads();
    constexpr int GO = (s * BD_TC + 511) / 512;               // gather outputs per thread
    int gidx[GO][BD_LUSE];                                    // float offsets of my column's exchange entries
#pragma unroll
    for (int oo = 0; oo < GO; ++oo) {
        const int o = tid + oo * 512;
        const int c = o < s * BD_TC ? o / BD_TC : 0;
#pragma unroll
        for (int e = 0; e < BD_LUSE; ++e) gidx[oo][e] = lst[c * BD_LCAP + e] * BD_PS;
    }

    const int nch = a.NT / BD_TC;
    const int nsel = (a.K + BD_TC - 1) / BD_TC;               // selected-token chunks come first: they produce dot
    for (int ch = -nsel; ch < nch; ++ch) {
        const bool issel = ch < 0;
        const int k0 = (ch + nsel) * BD_TC;                   // first selected slot of a selected chunk
        // ---- V phase: Vb[row*s + c][t] = sum_j wy[row][j] * S[cy[row][j]][c][t] for the chunk's tokens ------------------
        if (!issel) {
            for (int it0 = tid; it0 < 512 * (BD_TC / 4); it0 += VBN * 512) {        // loads of VBN items in flight (latency-bound)
                f32x4 raw[VBN][4];
                float wv[VBN][4];
#pragma unroll
                for (int u = 0; u < VBN; ++u) {
                    const int it = it0 + u * 512;
                    const int rs = it / (BD_TC / 4), q4 = it % (BD_TC / 4);
                    const int row = rs / s, c = rs - row * s;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        raw[u][j] = *(const f32x4*)(Sg + ((size_t)(tab_cy[row * 4 + j] * s + c)) * a.ldt + ch * BD_TC + q4 * 4);
                        wv[u][j] = tab_wy[row * 4 + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < VBN; ++u) {
                    const int it = it0 + u * 512;
                    f32x4 acc = wv[u][0] * raw[u][0];
#pragma unroll
                    for (int j = 1; j < 4; ++j) acc += wv[u][j] * raw[u][j];
                    if ((ch + 1) * BD_TC > a.T) {                // pad tokens (t >= T): probability 0
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ch * BD_TC + (it % (BD_TC / 4)) * 4 + e >= a.T) acc[e] = -1.0e30f;   // finite: the horizontal taps have both signs
                    }
                    *(f32x4*)(Vb + (it / (BD_TC / 4)) * BD_TS + (it % (BD_TC / 4)) * 4) = acc;
                }
            }
        } else {
            for (int it0 = tid; it0 < 512 * BD_TC; it0 += 8 * 512) {    // (row*s + c, k): the selected tokens' columns
                float raw[8][4], wv[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = it0 + u * 512;
                    const int rs = it / BD_TC, kk = k0 + it % BD_TC;
                    const int row = rs / s, c = rs - row * s;
                    const int tk = selk[kk];                            // slots k >= K: token 0, weight 0 below
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        raw[u][j] = Sg[((size_t)(tab_cy[row * 4 + j] * s + c)) * a.ldt + tk];
                        wv[u][j] = kk < a.K ? tab_wy[row * 4 + j] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = it0 + u * 512;
                    float v = wv[u][0] * raw[u][0];
#pragma unroll
                    for (int j = 1; j < 4; ++j) v = fmaf(wv[u][j], raw[u][j], v);
                    Vb[(it / BD_TC) * BD_TS + it % BD_TC] = v;
                }
            }
        }
        __syncthreads();

        // ---- sweep: the thread's 2 groups x K2 rows; vertical adjoint into acc[slot][token] -------------------------
        f32x2 acc[5][BD_TC / 2];                               // [slot][token pair]
#pragma unroll
        for (int sl = 0; sl < 5; ++sl)
#pragma unroll
            for (int t = 0; t < BD_TC / 2; ++t) acc[sl][t] = f32x2{0.f, 0.f};
        auto sweep = [&](auto sel_c) {
            constexpr bool SEL = decltype(sel_c)::value;
#pragma unroll
            for (int g = 0; g < 2; ++g) {                      // group g: tap rows in slots g .. g + 3
#pragma unroll
                for (int r = 0; r < K2; ++r) {
                    const int ri = g * K2 + r;
                    // what every probability of this row is multiplied with: selected chunk g_k / (L H) (slots k >= K lie past
                    // the descriptor's range: 0), natural chunks -dot (pad tokens have p = 0: -1e30 logits from the V phase)
                    float gk[BD_TC];
                    if (SEL) {
#pragma unroll
                        for (int kk = 0; kk < BD_TC; ++kk)      // the range check sees the vector offset only
                            gk[kk] = skp_buf_load_f32(grs, gvo + (k0 + kk) * RR * 4, ri * R * 4, 0);
                    }
                    const float nd = SEL ? a.inv_lh : -dot_r[ri];
                    const float lse = lse_r[ri];
                    const f32x4 wyv = *(const f32x4*)(tab_wy + (part * RPT + ri) * 4);
                    const float* vrow = Vb + vbase + ri * s * BD_TS;
                    float d4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < BD_TC / 4; ++q) {
                        // horizontal interpolation of 4 tokens as two register pairs (v_pk_* on the LDS quads; scalars stay
                        // scalars: a float4-minus-scalar makes the compiler keep a 4-register splat per row, hoisted and spilled)
                        const f32x4 t0 = *(const f32x4*)(vrow + cx[0] + 4 * q);
                        const f32x4 t1 = *(const f32x4*)(vrow + cx[1] + 4 * q);
                        const f32x4 t2 = *(const f32x4*)(vrow + cx[2] + 4 * q);
                        const f32x4 t3 = *(const f32x4*)(vrow + cx[3] + 4 * q);
                        f32x2 lo = wx2[0] * f32x2{t0[0], t0[1]}, hi = wx2[0] * f32x2{t0[2], t0[3]};
                        lo = wx2[1] * f32x2{t1[0], t1[1]} + lo; hi = wx2[1] * f32x2{t1[2], t1[3]} + hi;
                        lo = wx2[2] * f32x2{t2[0], t2[1]} + lo; hi = wx2[2] * f32x2{t2[2], t2[3]} + hi;
                        lo = wx2[3] * f32x2{t3[0], t3[1]} + lo; hi = wx2[3] * f32x2{t3[2], t3[3]} + hi;
                        float f[4] = {__builtin_amdgcn_exp2f(lo[0] - lse), __builtin_amdgcn_exp2f(lo[1] - lse),
                                      __builtin_amdgcn_exp2f(hi[0] - lse), __builtin_amdgcn_exp2f(hi[1] - lse)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (SEL) f[e] *= gk[4 * q + e];
                            f[e] *= nd;
                            if (SEL) d4[e] += f[e];
                        }
                        // vertical adjoint: the row's four taps
                        const f32x2 f01 = {f[0], f[1]}, f23 = {f[2], f[3]};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x2 w2 = {wyv[j], wyv[j]};
                            acc[g + j][2 * q] = w2 * f01 + acc[g + j][2 * q];
                            acc[g + j][2 * q + 1] = w2 * f23 + acc[g + j][2 * q + 1];
                        }
                    }
                    if (SEL) dot_r[ri] += (d4[0] + d4[1]) + (d4[2] + d4[3]);
                    __builtin_amdgcn_sched_barrier(0);         // rows one after the other: nothing of the next row hoisted above
                }
            }
        };
        if (issel) sweep(std::true_type{}); else sweep(std::false_type{});
        __syncthreads();                                       // every lane is done reading Vb: it becomes the exchange buffer

        // ---- horizontal adjoint, one band row at a time: pre-reduce in the tap-sharing lane group, exchange, gather ----
        float* xb = Vb;                                        // [NP][R*E + 1][BD_PS]
        if (tid < NP * BD_PS) xb[((tid / BD_PS) * XPE + R * E) * BD_PS + tid % BD_PS] = 0.f;      // the zero slots
        const int tcol = issel ? a.NT + k0 : ch * BD_TC;
#pragma unroll
        for (int rho = 0; rho < NR; ++rho) {
#pragma unroll
            for (int t = 0; t < BD_TC; ++t) {
                float v = 0.f;                                 // my slot for this band row: rho - part in [0, 5), else nothing
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
                    if (rho - pp >= 0 && rho - pp < 5) v = (part == pp) ? acc[(rho - pp >= 0 && rho - pp < 5) ? rho - pp : 0][t >> 1][t & 1] : v;
                if (K2 == 4) {
                    float pq = bd_quad_bcast<0>(v) * wq[0];
                    pq = fmaf(bd_quad_bcast<1>(v), wq[1], pq);
                    pq = fmaf(bd_quad_bcast<2>(v), wq[2], pq);
                    pq = fmaf(bd_quad_bcast<3>(v), wq[3], pq);
                    xb[(part * XPE + xcol) * BD_PS + t] = pq;
                } else {
                    const float vp = bd_pair_swap(v);
                    xb[(part * XPE + 2 * xcol) * BD_PS + t] = fmaf(wq[1], vp, wq[0] * v);
                    xb[(part * XPE + 2 * xcol + 1) * BD_PS + t] = fmaf(wq[3], vp, wq[2] * v);
                }
            }
            __syncthreads();
#pragma unroll
            for (int oo = 0; oo < GO; ++oo) {                  // my output(s) (column c, token t): the lists sit in registers
                const int o = tid + oo * 512;
                if (o < s * BD_TC) {
                    const int t = o % BD_TC;
                    float part_sum[NP];
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) {
                        part_sum[pp] = 0.f;
                        if (rho - pp < 0 || rho - pp >= 5) continue;           // this part has no slot for the row
                        const float* src = xb + pp * XPE * BD_PS + t;
                        float v[BD_LUSE];
#pragma unroll
                        for (int e = 0; e < BD_LUSE; ++e) v[e] = src[gidx[oo][e]];        // independent LDS reads, all in flight
#pragma unroll
                        for (int e = 0; e < BD_LUSE; ++e) part_sum[pp] += v[e];
                    }
                    float sum = part_sum[0];
#pragma unroll
                    for (int pp = 1; pp < NP; ++pp) sum += part_sum[pp];
                    Pg[((size_t)rho * s + o / BD_TC) * a.NTP + tcol + t] = sum;
                }
            }
            __syncthreads();
        }
    }
}

// dS[b,h,cy,c,t] = sum over the band rows that are (or clamp to) low-res row cy, band order; + the selected tokens' part.
struct CombLayerArgs { const float* P; float* dS; int s, nb; long n4; };
struct CombArgs {
    CombLayerArgs ly[SKP_MAX_LAYERS];
    const int64_t* sel;
    int B, H, K, NT, NTP, ldt, NP;
};

__global__ __launch_bounds__(256) void skp_map_band_combine_kernel(CombArgs a) {
    const CombLayerArgs ly = a.ly[blockIdx.y];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ly.n4) return;
    const int Q = a.NT / 4, s = ly.s, NR = a.NP + 4;
    const int q4 = (int)(i % Q);
    long r = i / Q;
    const int c = (int)(r % s); r /= s;
    const int cy = (int)(r % s); r /= s;                       // r = b*H + h
    const int b = (int)(r / a.H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float selacc[BD_KMAX];
#pragma unroll
    for (int k = 0; k < BD_KMAX; ++k) selacc[k] = 0.f;
    const float* Pb = ly.P + (size_t)r * ly.nb * NR * s * a.NTP;
    for (int m = 0; m < ly.nb; ++m) {
        const int r0 = m * a.NP - 2;                       // unclamped low-res row of band row 0
        if (r0 + NR - 1 < cy - 2 || r0 > cy + 2) continue;
        for (int rho = 0; rho < NR; ++rho) {
            int rr = r0 + rho;
            rr = rr < 0 ? 0 : (rr > s - 1 ? s - 1 : rr);
            if (rr != cy) continue;
            const float* p = Pb + ((size_t)(m * NR + rho) * s + c) * a.NTP;
            acc += *(const f32x4*)(p + q4 * 4);
#pragma unroll
            for (int k = 0; k < BD_KMAX; ++k)
                if (k < a.K) selacc[k] += p[a.NT + k];
        }
    }
#pragma unroll
    for (int k = 0; k < BD_KMAX; ++k) {
        if (k < a.K) {
            const int t = (int)a.sel[(size_t)b * a.K + k];
            if ((t >> 2) == q4) acc[t & 3] += selacc[k];
        }
    }
    *(f32x4*)(ly.dS + ((size_t)r * s * s + (size_t)cy * s + c) * a.ldt + q4 * 4) = acc;
}

int band_k2(int R, int s) {                                    // rows per group, or 0 when the layer is not served
    if (s <= 0 || R % s) return 0;
    const int k = R / s;
    return (k == 8 || k == 4) ? k / 2 : 0;
}

}  // namespace

extern "C" int skp_attn_map_bwd_band_ok(const int* s, int L, int H, int T, int R, int K) {
    if (!s || L <= 0 || L > SKP_MAX_LAYERS || H <= 0 || T <= 0 || T > 128 || K <= 0 || K > BD_KMAX) return 0;
    if (R != 128 && R != 256) return 0;
    const int NP = 512 / R;
    for (int l = 0; l < L; ++l) {
        if (!band_k2(R, s[l]) || s[l] > 64 || s[l] % NP) return 0;
    }
    return 1;
}

extern "C" int64_t skp_attn_map_bwd_band_workspace(const int* s, int L, int B, int H, int T, int R, int K) {
    if (!s || B <= 0) return SKP_E_BADARG;
    if (!skp_attn_map_bwd_band_ok(s, L, H, T, R, K)) return SKP_E_RANGE;
    const int NP = 512 / R, NR = NP + 4, ntp = ((T + 15) / 16) * 16 + BD_KMAX;
    int64_t fl = 0;
    for (int l = 0; l < L; ++l) fl += (int64_t)B * H * (s[l] / NP) * NR * s[l] * ntp;
    return fl * (int64_t)sizeof(float) + 64;
}

extern "C" int skp_attn_map_bwd_band_f32(const float* const* S, float* const* dS, const int* s, int L, int B, int H, int T,
                                         int R, const int64_t* sel, const float* G, int K, const float* lse, void* workspace,
                                         int ldt, void* stream) {
    if (!S || !dS || !s || !sel || !G || !lse || !workspace || L <= 0 || B <= 0 || H <= 0 || T <= 0 || R <= 0 || K <= 0)
        return SKP_E_BADARG;
    if (!skp_attn_map_bwd_band_ok(s, L, H, T, R, K)) return SKP_E_RANGE;
    const int nt = ((T + 15) / 16) * 16;
    if (ldt < nt || (ldt & 3)) return SKP_E_BADARG;
    for (int l = 0; l < L; ++l)
        if (!S[l] || !dS[l]) return SKP_E_BADARG;
    const int NP = 512 / R, NR = NP + 4;
    hipStream_t st = (hipStream_t)stream;
    float* pcur = (float*)(((uintptr_t)workspace + 31) & ~(uintptr_t)31);
    CombArgs cb{};
    cb.sel = sel; cb.B = B; cb.H = H; cb.K = K; cb.NT = nt; cb.NTP = nt + BD_KMAX; cb.ldt = ldt; cb.NP = NP;
    float* Pl[SKP_MAX_LAYERS];
    long n4max = 0;
    for (int l = 0; l < L; ++l) {
        Pl[l] = pcur;
        const int nb = s[l] / NP;
        pcur += (size_t)B * H * nb * NR * s[l] * (nt + BD_KMAX);
        cb.ly[l].P = Pl[l]; cb.ly[l].dS = dS[l]; cb.ly[l].s = s[l]; cb.ly[l].nb = nb;
        cb.ly[l].n4 = (long)B * H * s[l] * s[l] * (nt / 4);
        n4max = cb.ly[l].n4 > n4max ? cb.ly[l].n4 : n4max;
    }
    for (int k2 = 4; k2 >= 2; k2 -= 2) {                        // one launch per tap-sharing group size
        BandArgs a{};
        a.sel = sel; a.G = G; a.lse = lse;
        a.L = L; a.B = B; a.H = H; a.T = T; a.R = R; a.K = K; a.ldt = ldt; a.NT = nt; a.NTP = nt + BD_KMAX;
        a.inv_lh = 1.0f / (float)(L * H);
        int blocks = 0, smax = 0;
        for (int l = 0; l < L; ++l) {
            if (band_k2(R, s[l]) != k2) continue;
            BandLayerArgs& y = a.ly[a.nl++];
            y.S = S[l]; y.P = Pl[l]; y.s = s[l]; y.l = l; y.nb = s[l] / NP; y.blk0 = blocks;
            blocks += H * y.nb;
            smax = s[l] > smax ? s[l] : smax;
        }
        if (!a.nl) continue;
        a.blocks_per_b = blocks;
        const int brn = NP * 2 * k2, e = k2 == 2 ? 2 : 1;
        const size_t xf = (size_t)NP * ((size_t)R * e + 1) * BD_PS, vbf = 512 * (size_t)BD_TS > xf ? 512 * (size_t)BD_TS : xf;
        const size_t lds = (vbf + 8 * (size_t)brn + (size_t)smax * (BD_LCAP + 1) + BD_KMAX) * sizeof(float);
        const dim3 grid((unsigned)((long)B * blocks)), block(512);
#define SKP_BAND_LAUNCH(K2V, NPV)                                                                                       \
        {                                                                                                               \
            if (lds > 64 * 1024) {                                                                                      \
                hipError_t e2 = hipFuncSetAttribute((const void*)skp_map_bwd_band_kernel<K2V, NPV>,                     \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
                if (e2 != hipSuccess) return (int)e2;                                                                   \
            }                                                                                                           \
            hipLaunchKernelGGL((skp_map_bwd_band_kernel<K2V, NPV>), grid, block, lds, st, a);                           \
        }
        if (k2 == 4 && NP == 4) SKP_BAND_LAUNCH(4, 4)
        else if (k2 == 4) SKP_BAND_LAUNCH(4, 2)
        else if (NP == 4) SKP_BAND_LAUNCH(2, 4)
        else SKP_BAND_LAUNCH(2, 2)
#undef SKP_BAND_LAUNCH
        int rc = skp_launch_status();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(skp_map_band_combine_kernel, dim3((unsigned)((n4max + 255) / 256), L), dim3(256), 0, st, cb);
    return skp_launch_status();
}
