// Backward of the fused up-res attention map for a SPARSE map gradient, token-major.
//
// Reference path differentiated: ptp_utils.py:513-538 (bicubic x -> R^2, to_q, einsum, softmax) + optimize.py:27-79
// (head / layer mean), with the gradient the losses of optimize.py:157-206 produce: dM is non-zero on the K selected
// tokens of each batch row only (optimize.py:395-414 index the map with `top_embedding_indices`).
//
//   z_up[p,t] = sum_q U[p,q] z[q,t]      (bicubic, separable: rows then columns)
//   p_t       = exp2(z_up[p,t] - lse[p])  (lse = log2-sum-exp over ALL tokens, kept by the forward)
//   dot[p]    = sum_{k<K} p_{sel_k} g_k[p],   g = dM / (L*H)
//   dz[q,t]   = sum_p U[p,q] p_t (g_t[p] - dot[p])
//
// Three kernels, no atomics, fixed summation order (bit-reproducible):
//   1. skp_map_dot_kernel (lane = pixel): the K selected tokens only -- z_up through an LDS V phase like the forward,
//      writes (lse, dot) pairs [B,L,H,R*R].  K/T of the forward's work.
//   2. skp_map_bwd_tok_kernel (lane = TOKEN): a wave = 16 tokens x 4 heads of one (batch row, layer, row band).  With one
//      token per lane the probabilities need no cross-lane step (lse is known), the bicubic taps of a pixel are
//      wave-uniform (scalar registers, fetched from a per-layer tap table by s_buffer_load) and both adjoints are
//      register FMAs: the wave sweeps its band row by row; inside a row a 4-column window of the vertically interpolated
//      logits slides with the source column (new column = 4 coalesced 64-byte buffer loads + 4 FMAs, once per R/s
//      pixels, requested one window position ahead) next to a 4-column window of the horizontal adjoint; a column
//      leaving the window is complete and goes through the vertical adjoint into a 4-row x s window of dz held in
//      VECTOR registers that are indexed with the wave-uniform source column (s_set_gpr_idx); a row leaving THAT window
//      is complete for this band and is stored, 64 B per head.  Tap indices beyond the borders are PyTorch's
//      clamp-on-access.  Pixels go in chunks of four whose taps / (lse, dot) / gradient values are requested while the
//      previous chunk computes; up-res rows that read the same four source rows go through the sweep as a PAIR (shared
//      column loads, window moves and indexed updates; four independent pixel cores per chunk).  No atomics, no LDS.
//      (A form that staged the (lse, dot) / gradient rows through LDS by LDS-DMA measured 8-12 % slower and was removed.)
//   3. skp_map_bwd_bands_kernel: rows touched by several bands (the 3-row halo) are summed in band order.  The sweep is
//      latency-bound per wave, so launches are cut into enough bands to fill and refill every wave slot of the chip.
#include "skp_common.h"
#include <stdlib.h>
#include <type_traits>

#define SKP_TOK_KMAX 32
#define SKP_TOK_SMAX 32

template <int I, int N, class F>
__device__ __forceinline__ void skp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        skp_static_for<I + 1, N>(f);
    }
}

struct __attribute__((aligned(32))) SkpTap { float w[4]; int i0; int pad[3]; };      // weights first: one 16-byte scalar load

// per-layer table: R taps, then xs[s + 2] (first pixel of every window position)
__host__ __device__ static inline size_t skp_tap_table_bytes(int R) {
    return (size_t)R * sizeof(SkpTap) + 128 + (SKP_TOK_SMAX + 4) * sizeof(int);    // + slack read by the last chunk
}
static inline size_t skp_align32(size_t v) { return (v + 31) & ~(size_t)31; }

struct TapArgs { int s[SKP_MAX_LAYERS]; int R; size_t stride; };

__global__ void skp_map_taps_kernel(char* tabs, TapArgs a) {   // grid (blocks over max(R, s + 3), L)
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int R = a.R, s = a.s[blockIdx.y];
    SkpTap* tab = (SkpTap*)(tabs + blockIdx.y * a.stride);
    const float ratio = (float)s / (float)R;
    if (x < R) {
        int idx[4]; float w[4];
        skp_cubic_taps(x, ratio, s, idx, w);                   // the forward's weights, bit for bit
        SkpTap t;
        t.i0 = (int)floorf(ratio * ((float)x + 0.5f) - 0.5f);
        t.w[0] = w[0]; t.w[1] = w[1]; t.w[2] = w[2]; t.w[3] = w[3];
        t.pad[0] = t.pad[1] = t.pad[2] = 0;
        tab[x] = t;
    }
    // xs[P], P = 0 .. s+1: first pixel whose window position (i0 + 1) is >= P; xs[s+1] = xs[s+2] = R
    int* xs = (int*)(tab + R);
    if (x <= s + 2) {
        int first = R;
        for (int q = R - 1; q >= 0; --q) {
            const int i0 = (int)floorf(ratio * ((float)q + 0.5f) - 0.5f);
            if (i0 + 1 >= x) first = q;
        }
        xs[x] = (x == 0) ? 0 : first;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 1. (lse, dot) per (batch row, layer, head, pixel)
// ---------------------------------------------------------------------------------------------------------
struct DotArgs {
    const float* S[SKP_MAX_LAYERS];
    int s[SKP_MAX_LAYERS];
    int L, B, H, T, R, K, ldt;
    int TH, segs, smax;
    float inv_lh;
};

__global__ __launch_bounds__(256) void skp_map_dot_kernel(DotArgs a, const int64_t* __restrict__ sel,
                                                          const float* __restrict__ G, const float* __restrict__ lse,
                                                          f32x2* __restrict__ ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, b = blockIdx.x, l = blockIdx.z;
    const int R = a.R, H = a.H, K = a.K, RR = R * R, s = a.s[l];
    const int KS = K | 1;                                      // odd row stride
    int y0, ry, x, th_eff; bool valid;
    if (R <= 256) {
        y0 = blockIdx.y * a.TH; ry = tid / R; x = tid - ry * R;
        th_eff = (R - y0 < a.TH) ? R - y0 : a.TH; valid = ry < th_eff;
    } else {
        y0 = blockIdx.y / a.segs; const int seg = blockIdx.y - y0 * a.segs;
        ry = 0; x = seg * 256 + tid; th_eff = 1; valid = x < R;
    }
    if (!valid) { ry = 0; x = (R <= 256) ? 0 : R - 1; }
    const int p = (y0 + ry) * R + x;
    float* Vt = smem;                                          // [th_eff * s][KS]
    int* tab_cy = (int*)(smem + a.TH * a.smax * KS);
    float* tab_wy = (float*)(tab_cy + a.TH * 4);
    int* tsel = (int*)(tab_wy + a.TH * 4);

    const float ratio = (float)s / (float)R;
    if (tid < K) {
        int t = (int)sel[(size_t)b * K + tid];
        tsel[tid] = t < 0 ? 0 : (t > a.T - 1 ? a.T - 1 : t);
    }
    if (tid < th_eff) {
        int cy[4]; float wy[4];
        skp_cubic_taps(y0 + tid, ratio, s, cy, wy);
#pragma unroll
        for (int j = 0; j < 4; ++j) { tab_cy[tid * 4 + j] = cy[j]; tab_wy[tid * 4 + j] = wy[j]; }
    }
    int cx[4]; float wx[4];
    skp_cubic_taps(x, ratio, s, cx, wx);
    int base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) base[i] = (ry * s + cx[i]) * KS;
    float g[SKP_TOK_KMAX];
#pragma unroll
    for (int k = 0; k < SKP_TOK_KMAX; ++k)
        g[k] = (k < K && valid) ? G[((size_t)b * K + k) * RR + p] : 0.f;
    const int rc = th_eff * s, items = rc * K;
    const float inv_s = 1.0f / (float)s;
    for (int h = 0; h < H; ++h) {
        const float* Sg = a.S[l] + ((size_t)(b * H + h) * s * s) * a.ldt;
        __syncthreads();                                       // tables ready / previous head's reads done
        for (int it = tid; it < items; it += 256) {
            const int r = it / K, k = it - r * K;
            const int row = (int)(((float)r + 0.5f) * inv_s);
            const int c = r - row * s;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = fmaf(tab_wy[row * 4 + j], Sg[(size_t)(tab_cy[row * 4 + j] * s + c) * a.ldt + tsel[k]], acc);
            Vt[r * KS + k] = acc;
        }
        __syncthreads();
        const size_t li = ((size_t)(b * a.L + l) * H + h) * RR + p;
        const float ls = valid ? lse[li] : 0.f;
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < SKP_TOK_KMAX; ++k) {
            if (k < K) {
                float v = wx[0] * Vt[base[0] + k];
                v = fmaf(wx[1], Vt[base[1] + k], v);
                v = fmaf(wx[2], Vt[base[2] + k], v);
                v = fmaf(wx[3], Vt[base[3] + k], v);
                dot = fmaf(__builtin_amdgcn_exp2f(v - ls), g[k], dot);
            }
        }
        if (valid) ld[li] = f32x2{ls, dot * a.inv_lh};
    }
}

// ---------------------------------------------------------------------------------------------------------
// 2. token-major sweep
// ---------------------------------------------------------------------------------------------------------
struct TokLayer {
    const float* S;                // logits of this layer [B,H,s*s,ldt]
    float* out;                    // dS of this layer (NB == 1) or its band partials
    const SkpTap* tab;
    int l, s;
    long out_bh, out_band;         // float strides between (b,h) blocks / bands of `out`
    int out_row, out_col;          // float strides between source rows / columns
    unsigned s_bytes;
};

struct TokArgs {
    TokLayer ly[SKP_MAX_LAYERS];   // the layers of this launch (same register class)
    int nl;
    const f32x2* ld;               // [B,L,H,R*R] (lse, dot)
    const float* G;                // [B,K,R*R]
    const int64_t* sel;            // [B,K]
    int L, B, H, T, R, K, ldt;
    int NB, BH, NTG, HG, NTGB;     // bands, band height, 16-token groups, head groups, workgroups per token axis
    float inv_lh;
    unsigned tab_bytes;            // buffer size for the tap-table descriptor
};

typedef int i32x8 __attribute__((ext_vector_type(8)));
// scalar (SMEM) buffer loads: the tap tables are wave-uniform and were written by an EARLIER launch.  (Every element of
// a loaded vector is consumed: this compiler narrows a partly used s.buffer.load vector to ONE dword.)
__device__ f32x4 skp_sbuf_load_f32x4(i32x4 rsrc, int offset, int aux) __asm("llvm.amdgcn.s.buffer.load.v4f32");
__device__ int skp_sbuf_load_x1(i32x4 rsrc, int offset, int aux) __asm("llvm.amdgcn.s.buffer.load.i32");

struct TapS { float w0, w1, w2, w3; };
__device__ __forceinline__ TapS skp_tap(i32x4 rs, int idx) {
    const f32x4 v = skp_sbuf_load_f32x4(rs, idx * 32, 0);
    return TapS{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ int skp_tap_i0(i32x4 rs, int idx) { return skp_sbuf_load_x1(rs, idx * 32 + 16, 0); }


template <int N> struct SkpVec;
template <> struct SkpVec<8> { typedef float type __attribute__((ext_vector_type(8))); };
template <> struct SkpVec<16> { typedef float type __attribute__((ext_vector_type(16))); };
template <> struct SkpVec<32> { typedef float type __attribute__((ext_vector_type(32))); };

// SM: register class = width of the dz window rows (vector registers, indexed with the wave-uniform source column).
// A workgroup = up to 8 waves = consecutive 16-token groups of ONE (batch row, layer, head group, band) (no barriers; the
// grouping only keeps the waves that read the same (lse, dot) / gradient rows on one CU).
// Up-res rows go in PAIRS that share their four source rows (R/s even: every even-odd pair does): the column loads, the
// window moves and the indexed updates of the dz window serve both rows, and the two rows x two pixels of a chunk are
// four independent pixel cores for the scheduler.
template <int SM>
__global__ __launch_bounds__(512, (SM <= 16 ? 2 : 1)) void skp_map_bwd_tok_kernel(TokArgs a) {
    typedef typename SkpVec<SM>::type vrow;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // everything derived from it is wave-uniform
    const int WPB = blockDim.x >> 6;
    int task = blockIdx.x;
    const int tgb = task % a.NTGB; task /= a.NTGB;
    const int hg = task % a.HG; task /= a.HG;
    const int li = task % a.nl; task /= a.nl;
    const int m = task % a.NB;
    const int b = task / a.NB;
    const TokLayer& ly = a.ly[li];
    if (tgb * WPB + wave >= a.NTG) return;
    const int tg = tgb * WPB + wave;
    const int hh = lane >> 4, t = tg * 16 + (lane & 15);
    const int h = hg * 4 + hh;
    const bool live = h < a.H && t < a.T;                      // pad tokens / heads beyond H: computed, stored as 0
    const int hA = h < a.H ? h : a.H - 1;
    const int R = a.R, s = ly.s, RR = R * R, K = a.K;
    const i32x4 rs_tab = skp_make_rsrc(ly.tab, a.tab_bytes);
    const i32x4 rs_S = skp_make_rsrc(ly.S, ly.s_bytes);
    const f32x2* __restrict__ ldp = a.ld + ((size_t)(b * a.L + ly.l) * a.H + hA) * RR;
    // which selected row of this batch row is MY token (if any)
    int gk = -1;
    for (int k = 0; k < K; ++k)
        if ((int)a.sel[(size_t)b * K + k] == t) gk = k;
    const bool has_g = gk >= 0 && live;
    const bool any_g = __builtin_amdgcn_ballot_w64(has_g) != 0;          // this WAVE reads gradient rows
    const float* __restrict__ Gp = a.G + ((size_t)b * K + (gk < 0 ? 0 : gk)) * RR;
    const float gscale = has_g ? a.inv_lh : 0.f;

    const int y0 = m * a.BH, y1 = (y0 + a.BH < R) ? y0 + a.BH : R;
    if (y0 >= y1) return;
    float* __restrict__ outp = ly.out + (size_t)(b * a.H + hA) * ly.out_bh + (size_t)m * ly.out_band + t;
    int cur = skp_tap_i0(rs_tab, y0);                          // dz window rows: cur-1 .. cur+2
    const int rlo = cur - 1 < 0 ? 0 : cur - 1;                 // first row this band emits (partials are band-relative)
    const float keep = live ? 1.f : 0.f;
    const int sbase = ((b * a.H + hA) * s * s) * a.ldt * 4 + t * 4;        // byte offset of S[b,h,0,t]
    const int col_bytes = a.ldt * 4;

    vrow win0 = 0.f, win1 = 0.f, win2 = 0.f, win3 = 0.f;       // dz rows cur-1 .. cur+2, indexed by the source column

    auto emit_row = [&](int idx, const vrow& row) {            // idx in [0, s-1]
        if (h < a.H) {
            float* o = outp + (size_t)(a.NB == 1 ? idx : idx - rlo) * ly.out_row;
#pragma unroll
            for (int c = 0; c < SM; ++c)
                if (c < s) o[(size_t)c * ly.out_col] = row[c] * keep;
        }
    };

    for (int y = y0; y < y1;) {
        const TapS tya = skp_tap(rs_tab, y);
        const int ty_i0 = skp_tap_i0(rs_tab, y);
        // the partner row: the next one when it reads the same four source rows, else a copy of this row with zero weight
        const bool pair = y + 1 < y1 && skp_tap_i0(rs_tab, y + 1) == ty_i0;
        const int yb = pair ? y + 1 : y;
        TapS tyb = skp_tap(rs_tab, yb);
        if (!pair) tyb = TapS{0.f, 0.f, 0.f, 0.f};
        while (cur < ty_i0) {                                  // slot 0 is complete for this band
            if (cur - 1 < 0) win1 += win0;                     // clamped row: belongs to the border row
            else emit_row(cur - 1, win0);
            win0 = win1; win1 = win2; win2 = win3; win3 = 0.f;
            ++cur;
        }
        int voff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { int v = cur - 1 + j; v = v < 0 ? 0 : (v > s - 1 ? s - 1 : v); voff[j] = sbase + v * s * col_bytes; }
        auto vcol_load = [&](int c, float (&raw)[4]) {         // source column c (clamped here) = scalar offset
            c = c < 0 ? 0 : (c > s - 1 ? s - 1 : c);
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = skp_buf_load_f32(rs_S, voff[j], c * col_bytes, 0);
        };
        auto vcol = [&](const float (&raw)[4], const TapS& w) {
            float v = w.w0 * raw[0];
            v = fmaf(w.w1, raw[1], v); v = fmaf(w.w2, raw[2], v); v = fmaf(w.w3, raw[3], v);
            return v;
        };
        auto vadj = [&](int c, float va, float vb) {           // vertical adjoint of one finished source column, both rows
            c = c < 0 ? 0 : (c > s - 1 ? s - 1 : c);           // PyTorch clamps tap indices on access
            win0[c] += fmaf(tya.w0, va, tyb.w0 * vb); win1[c] += fmaf(tya.w1, va, tyb.w1 * vb);
            win2[c] += fmaf(tya.w2, va, tyb.w2 * vb); win3[c] += fmaf(tya.w3, va, tyb.w3 * vb);
        };
        // column window: source columns ccur-1 .. ccur+2 (clamped on access); it starts at the first pixel's position
        int ccur = skp_tap_i0(rs_tab, 0);
        float raw[4], raw1[4], raw2[4], vwa[4], vwb[4], acca[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
        vcol_load(ccur - 1, raw); vcol_load(ccur, raw1); vcol_load(ccur + 1, raw2);
        vwa[0] = vcol(raw, tya); vwb[0] = vcol(raw, tyb);
        vwa[1] = vcol(raw1, tya); vwb[1] = vcol(raw1, tyb);
        vwa[2] = vcol(raw2, tya); vwb[2] = vcol(raw2, tyb);
        vcol_load(ccur + 2, raw);
        vwa[3] = vcol(raw, tya); vwb[3] = vcol(raw, tyb);
        vcol_load(ccur + 3, raw);                              // the columns the next two window positions add
        vcol_load(ccur + 4, raw1);
        const f32x2* __restrict__ ldra = ldp + (size_t)y * R;
        const f32x2* __restrict__ ldrb = ldp + (size_t)yb * R;
        const float* __restrict__ gra = Gp + (size_t)y * R;
        const float* __restrict__ grb = Gp + (size_t)yb * R;
        // Pixels go in chunks of two; a chunk's taps and window positions (scalar loads), (lse, dot) and gradient values of
        // both rows are requested while the previous chunk computes.
        TapS tn[2]; int in[2];
        f32x2 lna[2], lnb[2];
        float gna[2] = {0.f, 0.f}, gnb[2] = {0.f, 0.f};
        auto chunk_load = [&](int xc) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                tn[j] = skp_tap(rs_tab, xc + j);               // past the row end: table slack, never used
                in[j] = skp_tap_i0(rs_tab, xc + j);
                const int xi = xc + j < R ? xc + j : R - 1;
                lna[j] = ldra[xi]; lnb[j] = ldrb[xi];
                if (any_g) { gna[j] = has_g ? gra[xi] : 0.f; gnb[j] = has_g ? grb[xi] : 0.f; }
            }
        };
        chunk_load(0);
        for (int x = 0; x < R; x += 2) {
            TapS tc[2]; int ic[2]; f32x2 lca[2], lcb[2]; float gca[2], gcb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { tc[j] = tn[j]; ic[j] = in[j]; lca[j] = lna[j]; lcb[j] = lnb[j]; gca[j] = gna[j]; gcb[j] = gnb[j]; }
            if (x + 2 < R) chunk_load(x + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (x + j < R) {
                    while (ccur < ic[j]) {                     // the window moves on: column ccur-1 is complete for these rows
                        vadj(ccur - 1, acca[0], accb[0]);
                        acca[0] = acca[1]; acca[1] = acca[2]; acca[2] = acca[3]; acca[3] = 0.f;
                        accb[0] = accb[1]; accb[1] = accb[2]; accb[2] = accb[3]; accb[3] = 0.f;
                        vwa[0] = vwa[1]; vwa[1] = vwa[2]; vwa[2] = vwa[3]; vwa[3] = vcol(raw, tya);
                        vwb[0] = vwb[1]; vwb[1] = vwb[2]; vwb[2] = vwb[3]; vwb[3] = vcol(raw, tyb);
                        ++ccur;
#pragma unroll
                        for (int q = 0; q < 4; ++q) raw[q] = raw1[q];
                        vcol_load(ccur + 4, raw1);             // two positions ahead
                    }
                    float za = tc[j].w0 * vwa[0], zb = tc[j].w0 * vwb[0];
                    za = fmaf(tc[j].w1, vwa[1], za); zb = fmaf(tc[j].w1, vwb[1], zb);
                    za = fmaf(tc[j].w2, vwa[2], za); zb = fmaf(tc[j].w2, vwb[2], zb);
                    za = fmaf(tc[j].w3, vwa[3], za); zb = fmaf(tc[j].w3, vwb[3], zb);
                    const float pa = __builtin_amdgcn_exp2f(za - lca[j][0]), pb = __builtin_amdgcn_exp2f(zb - lcb[j][0]);
                    const float da = pa * fmaf(gca[j], gscale, -lca[j][1]), db = pb * fmaf(gcb[j], gscale, -lcb[j][1]);
                    acca[0] = fmaf(tc[j].w0, da, acca[0]); accb[0] = fmaf(tc[j].w0, db, accb[0]);
                    acca[1] = fmaf(tc[j].w1, da, acca[1]); accb[1] = fmaf(tc[j].w1, db, accb[1]);
                    acca[2] = fmaf(tc[j].w2, da, acca[2]); accb[2] = fmaf(tc[j].w2, db, accb[2]);
                    acca[3] = fmaf(tc[j].w3, da, acca[3]); accb[3] = fmaf(tc[j].w3, db, accb[3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) vadj(ccur - 1 + i, acca[i], accb[i]);        // the four columns still in the window
        y = yb + 1;
    }
    // flush: rows beyond the bottom border belong to row s-1, rows above the top border to row 0
    if (cur + 2 > s - 1) win2 += win3;
    if (cur + 1 > s - 1) win1 += win2;
    if (cur > s - 1) win0 += win1;
    if (cur - 1 < 0) win1 += win0;
    if (cur < 0) win2 += win1;
    if (cur + 1 < 0) win3 += win2;
    if (cur - 1 >= 0 && cur - 1 <= s - 1) emit_row(cur - 1, win0);
    if (cur >= 0 && cur <= s - 1) emit_row(cur, win1);
    if (cur + 1 >= 0 && cur + 1 <= s - 1) emit_row(cur + 1, win2);
    if (cur + 2 >= 0 && cur + 2 <= s - 1) emit_row(cur + 2, win3);
}

// ---------------------------------------------------------------------------------------------------------
// 3. sum of the band partials (fixed band order)
// ---------------------------------------------------------------------------------------------------------
struct BandLayer { const float* part; float* dS; const SkpTap* tab; int s, cap, NB, BH; long n4; };
struct BandArgs {
    BandLayer ly[SKP_MAX_LAYERS];
    int B, H, T, R, NT, ldt;
};

__global__ __launch_bounds__(256) void skp_map_bwd_bands_kernel(BandArgs a) {     // grid (blocks, L)
    const BandLayer& ly = a.ly[blockIdx.y];
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    if (it >= ly.n4 || ly.NB <= 1) return;                     // NB == 1: the sweep wrote dS itself
    const int Q = a.NT / 4, s = ly.s;
    const int q4 = (int)(it % Q);
    long r1 = it / Q;
    const int c = (int)(r1 % s); r1 /= s;
    const int r = (int)(r1 % s);
    const long bh = r1 / s;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < ly.NB; ++m) {
        const int y0 = m * ly.BH;
        if (y0 >= a.R) break;
        const int y1 = (y0 + ly.BH < a.R) ? y0 + ly.BH : a.R;
        int lo = ly.tab[y0].i0 - 1, hi = ly.tab[y1 - 1].i0 + 2;
        lo = lo < 0 ? 0 : lo; hi = hi > s - 1 ? s - 1 : hi;
        if (r >= lo && r <= hi)
            acc += *(const f32x4*)(ly.part + (((size_t)bh * ly.NB + m) * ly.cap + (r - lo)) * (size_t)s * a.NT +
                                   (size_t)c * a.NT + q4 * 4);
    }
    *(f32x4*)(ly.dS + ((size_t)bh * s * s + (size_t)r * s + c) * a.ldt + q4 * 4) = acc;
}

// ---------------------------------------------------------------------------------------------------------
static int skp_tok_class(int s) { return s <= 8 ? 0 : (s <= 16 ? 1 : 2); }

// bands of one launch (= the layers of one register class): the sweep is latency-bound per wave, so it wants every wave
// slot of the chip filled and refilled (>= 6 waves per SIMD over the launch; measured), bands of >= 8 rows
static int skp_tok_bands(const int* s, int L, int cls, int B, int H, int T, int R) {
    if (const int nb = skp_tune(SKP_TUNE_MAP_BANDS)) return nb > R ? R : nb;       // tests: force the band count
    int nl = 0;
    for (int l = 0; l < L; ++l) nl += skp_tok_class(s[l]) == cls;
    const long cols = (long)nl * B * ((H + 3) / 4) * ((T + 15) / 16);
    int nb = 1;
    while (cols * nb < 6144 && R / (nb * 2) >= 8) nb *= 2;
    return nb;
}

static int skp_tok_cap(int R, int s, int BH) { return (int)(((long)BH * s + R - 1) / R) + 4; }

extern "C" int64_t skp_attn_map_bwd_sparse_workspace(const int* s, int L, int B, int H, int T, int R, int K) {
    if (!s || L <= 0 || L > SKP_MAX_LAYERS || B <= 0 || H <= 0 || T <= 0 || R <= 0 || K <= 0) return SKP_E_BADARG;
    if (K > SKP_TOK_KMAX || R > 1024) return SKP_E_RANGE;
    const int nt = ((T + 15) / 16) * 16;
    int64_t bytes = (int64_t)L * skp_align32(skp_tap_table_bytes(R));
    bytes += (int64_t)B * L * H * R * R * 8;
    for (int l = 0; l < L; ++l) {
        if (s[l] <= 0) return SKP_E_BADARG;
        if (s[l] > SKP_TOK_SMAX) return SKP_E_RANGE;
    }
    for (int l = 0; l < L; ++l) {
        const int nb = skp_tok_bands(s, L, skp_tok_class(s[l]), B, H, T, R), bh = (R + nb - 1) / nb;
        if (nb > 1) bytes += (int64_t)B * H * nb * skp_tok_cap(R, s[l], bh) * s[l] * nt * 4;
    }
    return bytes + 64;
}

template <int SM>
static int skp_tok_launch(TokArgs& a, int wpb, hipStream_t st) {
    const unsigned nblk = (unsigned)((long)a.B * a.NB * a.nl * a.HG * a.NTGB);
    hipLaunchKernelGGL((skp_map_bwd_tok_kernel<SM>), dim3(nblk), dim3(64 * wpb), 0, st, a);
    return 0;
}

extern "C" int skp_attn_map_bwd_sparse_f32(const float* const* S, float* const* dS, const int* s, int L, int B, int H,
                                           int T, int R, const int64_t* sel, const float* G, int K, const float* lse,
                                           void* workspace, int ldt, void* stream) {
    if (!S || !dS || !s || !sel || !G || !lse || !workspace || L <= 0 || B <= 0 || H <= 0 || T <= 0 || R <= 0 || K <= 0)
        return SKP_E_BADARG;
    if (L > SKP_MAX_LAYERS || K > SKP_TOK_KMAX || R > 1024) return SKP_E_RANGE;
    const int nt = ((T + 15) / 16) * 16;
    if (ldt < nt || (ldt & 3)) return SKP_E_BADARG;
    int smax = 0;
    for (int l = 0; l < L; ++l) {
        if (!S[l] || !dS[l] || s[l] <= 0) return SKP_E_BADARG;
        if (s[l] > SKP_TOK_SMAX) return SKP_E_RANGE;
        smax = s[l] > smax ? s[l] : smax;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)(((uintptr_t)workspace + 31) & ~(uintptr_t)31);
    const size_t tab_bytes = skp_align32(skp_tap_table_bytes(R));
    char* tabs = ws;
    f32x2* ld = (f32x2*)(ws + (size_t)L * tab_bytes);
    float* part = (float*)((char*)ld + (size_t)B * L * H * R * R * 8);
    {
        TapArgs ta{};
        for (int l = 0; l < L; ++l) ta.s[l] = s[l];
        ta.R = R; ta.stride = tab_bytes;
        const int n = R > smax + 3 ? R : smax + 3;
        hipLaunchKernelGGL(skp_map_taps_kernel, dim3((n + 255) / 256, L), dim3(256), 0, st, tabs, ta);
    }
    // 1. (lse, dot)
    DotArgs d{};
    for (int l = 0; l < L; ++l) { d.S[l] = S[l]; d.s[l] = s[l]; }
    d.L = L; d.B = B; d.H = H; d.T = T; d.R = R; d.K = K; d.ldt = ldt; d.smax = smax;
    d.inv_lh = 1.0f / (float)(L * H);
    int ntile;
    if (R <= 256) { d.TH = 256 / R; d.segs = 1; ntile = (R + d.TH - 1) / d.TH; }
    else { d.TH = 1; d.segs = (R + 255) / 256; ntile = R * d.segs; }
    if (ntile > 65535) return SKP_E_RANGE;
    const size_t dlds = ((size_t)d.TH * smax * (K | 1) + 8 * (size_t)d.TH + SKP_TOK_KMAX) * sizeof(float);
    if (dlds > 64 * 1024) return SKP_E_LDS;
    hipLaunchKernelGGL(skp_map_dot_kernel, dim3(B, ntile, L), dim3(256), dlds, st, d, sel, G, lse, ld);
    // 2. token-major sweep: one launch per register class (side <= 8 | <= 16 | <= 32)
    TokArgs a{};
    a.ld = ld; a.G = G; a.sel = sel; a.L = L; a.B = B; a.H = H; a.T = T; a.R = R; a.K = K; a.ldt = ldt;
    a.NTG = nt / 16; a.HG = (H + 3) / 4;
    a.inv_lh = 1.0f / (float)(L * H);
    a.tab_bytes = (unsigned)tab_bytes;
    const int wpb = a.NTG < 8 ? a.NTG : 8;
    a.NTGB = (a.NTG + wpb - 1) / wpb;
    BandArgs r{};
    r.B = B; r.H = H; r.T = T; r.R = R; r.NT = nt; r.ldt = ldt;
    long n4max = 0;
    bool any_bands = false;
    float* pcur = part;
    for (int cls = 0; cls < 3; ++cls) {
        const int nb = skp_tok_bands(s, L, cls, B, H, T, R), bh = (R + nb - 1) / nb;
        a.nl = 0; a.NB = nb; a.BH = bh;
        for (int l = 0; l < L; ++l) {
            if (skp_tok_class(s[l]) != cls) continue;
            TokLayer& y = a.ly[a.nl++];
            y.S = S[l]; y.tab = (const SkpTap*)(tabs + l * tab_bytes); y.l = l; y.s = s[l];
            y.s_bytes = (unsigned)((size_t)B * H * s[l] * s[l] * ldt * 4);
            const int cap = skp_tok_cap(R, s[l], bh);
            BandLayer& q = r.ly[l];
            q.NB = nb; q.BH = bh; q.s = s[l]; q.cap = cap; q.dS = dS[l]; q.tab = y.tab;
            q.n4 = (long)B * H * s[l] * s[l] * (nt / 4);
            if (nb == 1) {
                y.out = dS[l]; y.out_bh = (long)s[l] * s[l] * ldt; y.out_band = 0; y.out_row = s[l] * ldt; y.out_col = ldt;
                q.part = nullptr;
            } else {
                y.out = pcur; y.out_bh = (long)nb * cap * s[l] * nt; y.out_band = (long)cap * s[l] * nt;
                y.out_row = s[l] * nt; y.out_col = nt;
                q.part = pcur;
                n4max = q.n4 > n4max ? q.n4 : n4max;
                any_bands = true;
                pcur += (size_t)B * H * nb * cap * s[l] * nt;
            }
        }
        if (!a.nl) continue;
        int rc = cls == 0 ? skp_tok_launch<8>(a, wpb, st) : cls == 1 ? skp_tok_launch<16>(a, wpb, st) : skp_tok_launch<32>(a, wpb, st);
        if (rc) return rc;
    }
    // 3. band partials -> dS
    if (any_bands)
        hipLaunchKernelGGL(skp_map_bwd_bands_kernel, dim3((unsigned)((n4max + 255) / 256), L), dim3(256), 0, st, r);
    return skp_launch_status();
}
