// Fused up-res cross-attention map: bicubic up-sampling of the low-res logits, softmax over the
// learned tokens in registers, head/layer mean, one coalesced [B,T,R,R] write -- and its backward.
//
// Reference path replaced: ptp_utils.py:513-538 (bicubic x -> R^2, to_q, einsum, softmax, clone,
// per-head store of (B*h, R^2, T)) + optimize.py:27-79 (reshape/permute/stack/mean).
// Algebra: to_q is bias-free and bicubic resize is linear, so
//   scale * to_q(bicubic(x))[p] . k[t]  ==  bicubic( scale * to_q(x) . k[t] )[p]
// i.e. the LOGITS are up-sampled, not the activations: the 174.5 GF projection and the 11.3 GF
// up-res QK^T of the reference collapse to a 0.2 GF low-res QK^T (skp_gemm.hip, fp32 MFMA) plus
// ~9 VALU ops per (layer, head, token, pixel) here.  The kernel is VALU/LDS-issue bound.
//
// Data layout: logits S[l] are [B, H, s*s, NT] fp32, token-contiguous, NT = 16*ceil(T/16)
// (columns t >= T are never read into results).  Pre-multiplied by scale*log2(e): exp2 = v_exp_f32.
//
// Tiling (row-aligned, any R): R <= 256: a workgroup owns TH = 256/R whole rows (lane = pixel,
// x fastest => every M[t, y, x..x+63] store is a full 256-B line); R > 256: 256-pixel segments of a
// row.  ALL tokens of a pixel live in one lane, so the softmax is lane-local (no cross-lane ops).
// Per (layer, head):
//   V phase  Vt[r][c][t] = sum_j wy[r][j] * S[cy[r][j]][c][t]   float4 over t, coalesced -> LDS
//            (token stride NT+1: conflict-free for the per-lane column gathers below)
//   H phase  s_t = sum_i wx[i] * Vt[r(lane)][cx[i]][t]          4 LDS reads with immediate offsets
//            FWD: m = max s; e = exp2(s-m); acc_t += e / sum e;  lse = m + log2(sum e)
//            BWD: p = exp2(s-lse); dS_t = p (g_t - sum p g);  horizontal adjoint by an LDS
//                 transpose + gather (deterministic, no atomics) -> dV[y][seg][c][t] in HBM
//   second backward kernel: vertical adjoint dS[cy][c][t] = sum_y Wy(y,cy) sum_seg dV[y][seg][c][t]
#include "skp_common.h"

#ifndef SKP_MAP_VBATCH
#define SKP_MAP_VBATCH 3
#endif
#define SKP_MAP_XCAP 64           // slots per column in the quad-layer index lists (a column is touched by <= 5k/4 + 4 slots, k = R/s <= 32)

struct MapArgs {
    const float* S[SKP_MAX_LAYERS];
    float* dS[SKP_MAX_LAYERS];
    int s[SKP_MAX_LAYERS];
    long dv_off[SKP_MAX_LAYERS];   // float offset of layer l inside the dV workspace (per batch row 0)
    long dv_per_b;                 // floats of dV per batch row
    int L, B, H, T, R;
    int TH, TW, segs, smax;        // tile rows, tile width, segments per row, max layer side
    int vt_floats;                 // floats of the Vt buffer
    float inv_lh;
    // token-group extension (T > 128): the launch covers tokens [t0, t0+T) of a wider problem
    int ldt;                       // row stride (floats) of S / dS rows (>= NT of this launch)
    long m_bstride;                // floats between batch rows of M / dM
    int mode;                      // 0 self-contained | 1 statistics only | 2 apply external statistics
    int quad[SKP_MAX_LAYERS];      // 1: every aligned group of four pixels shares its four tap columns (R = k*s, k % 8 == 0,
                                   //    full tiles): one LDS read per token quad, the other three taps arrive by DPP
};

// value of quad lane K for all four lanes of an aligned lane quad (the compiler folds it into v_mul_f32_dpp)
template <int K>
__device__ __forceinline__ float skp_quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
// acc += (src of quad lane K) * w as ONE instruction: the DPP combine does not reach v_fmac_f32 (it sees the VOP3 fma),
// so the VOP2 DPP form is written out.  `src` must come from memory (LDS / VMEM return), not from a VALU instruction
// right before (DPP read-after-VALU-write needs wait states the assembler does not insert for inline text).
template <int K>
__device__ __forceinline__ void skp_quad_fmac(float& acc, float src, float w) {
    if (K == 1) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(w));
    if (K == 2) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(w));
    if (K == 3) asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(src), "v"(w));
}

// lane -> pixel of the row-aligned tile
struct Tile { int y0, seg, ry, x, th_eff; bool valid; };

__device__ __forceinline__ Tile skp_tile(const MapArgs& a, int blk, int tid) {
    Tile t;
    if (a.R <= 256) {
        t.y0 = blk * a.TH; t.seg = 0;
        t.ry = tid / a.R; t.x = tid - t.ry * a.R;
        t.th_eff = (a.R - t.y0 < a.TH) ? a.R - t.y0 : a.TH;
        t.valid = t.ry < t.th_eff;
    } else {
        t.y0 = blk / a.segs; t.seg = blk - t.y0 * a.segs;
        t.ry = 0; t.x = t.seg * 256 + tid; t.th_eff = 1;
        t.valid = t.x < a.R;
    }
    if (!t.valid) { t.ry = 0; t.x = (a.R <= 256) ? 0 : a.R - 1; }
    return t;
}

// V phase shared by forward and backward.  tab_cy/tab_wy hold the row taps of the tile's rows.
template <int NT>
__device__ __forceinline__ void skp_v_phase(const float* __restrict__ Sg, float* __restrict__ Vt,
                                            const int* __restrict__ tab_cy, const float* __restrict__ tab_wy,
                                            int s, int rc, int tid, int ldt) {
    constexpr int TS = NT + 4, Q = NT / 4;
    constexpr int VB = SKP_MAP_VBATCH;                         // items whose loads go out before the first is used: the phase is
    const float inv_s = 1.0f / (float)s;                       // latency-bound (L2 round trips), not bandwidth-bound
    const int items = rc * Q;
    for (int it0 = tid; it0 < items; it0 += VB * 256) {
        f32x4 raw[VB][4];
        float wv[VB][4];
        int dst[VB];
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int it = it0 + u * 256;
            const int itc = it < items ? it : tid;             // a thread without a u-th item re-reads its first (discarded)
            const int r = itc / Q, q4 = itc - r * Q;           // r = row*s + c
            const int row = (int)(((float)r + 0.5f) * inv_s);
            const int c = r - row * s;
            dst[u] = it < items ? r * TS + q4 * 4 : -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                raw[u][j] = *(const f32x4*)(Sg + ((size_t)(tab_cy[row * 4 + j] * s + c)) * ldt + q4 * 4);
                wv[u][j] = tab_wy[row * 4 + j];
            }
        }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            if (dst[u] >= 0) {
                f32x4 acc = wv[u][0] * raw[u][0];
#pragma unroll
                for (int j = 1; j < 4; ++j) acc += wv[u][j] * raw[u][j];
                *(f32x4*)(Vt + dst[u]) = acc;
            }
        }
    }
}


template <int NT, int MODE>
__global__ __launch_bounds__(256) void skp_attn_map_fwd_kernel(MapArgs a, float* __restrict__ M,
                                                               float* __restrict__ lse_out,
                                                               const float* __restrict__ lse_in) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TS = NT + 4;                                 // token quads are 16-byte aligned (ds_read_b128)
    const int tid = threadIdx.x, b = blockIdx.x;            // batch row fastest: workgroup id % 8 (XCD) == b % 8
    const int R = a.R, T = a.T, H = a.H, RR = R * R;
    const Tile tl = skp_tile(a, blockIdx.y, tid);
    const int p = (tl.y0 + tl.ry) * R + tl.x;
    float* Vt = smem;
    int* tab_cy = (int*)(smem + a.vt_floats);
    float* tab_wy = (float*)(tab_cy + a.TH * 4);

    constexpr int NP = NT / 2;
    f32x2 acc[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) acc[u] = f32x2{0.f, 0.f};

    int lh = 0;
    for (int l = 0; l < a.L; ++l) {
        const int s = a.s[l];
        const float ratio = (float)s / (float)R;
        int cx[4]; float wx[4];
        skp_cubic_taps(tl.x, ratio, s, cx, wx);
        int base[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) base[i] = (tl.ry * s + cx[i]) * TS;
        const bool quad = a.quad[l] != 0;
        const int base_own = base[tid & 3];
        __syncthreads();                                       // previous layer done with the tables (and raw rows)
        if (tid < tl.th_eff) {
            int cy[4]; float wy[4];
            skp_cubic_taps(tl.y0 + tid, ratio, s, cy, wy);
#pragma unroll
            for (int j = 0; j < 4; ++j) { tab_cy[tid * 4 + j] = cy[j]; tab_wy[tid * 4 + j] = wy[j]; }
        }
        const int rc = tl.th_eff * s;
        for (int h = 0; h < H; ++h, ++lh) {
            const float* Sg = a.S[l] + ((size_t)(b * H + h) * s * s) * a.ldt;
            __syncthreads();                                   // tables ready / previous H phase done
            skp_v_phase<NT>(Sg, Vt, tab_cy, tab_wy, s, rc, tid, a.ldt);
            __syncthreads();
            f32x2 sv[NP];
            float m = -INFINITY;
            // token pairs (v_pk_mul/v_pk_fma, 8-byte LDS reads), tap-major over blocks of UB pairs so that UB
            // independent fma chains are in flight (dependent packed ops otherwise stall a 2-waves/SIMD kernel)
            // token quads: one ds_read_b128 per tap per 4 tokens (256 B/clk LDS form), packed fp32 math
#pragma unroll
            for (int q = 0; q < NT / 4; ++q) {
                f32x4 v;
                if (quad) {
                    // the four lanes of a quad need the same four columns: lane j fetches column cx[j], the taps are
                    // quad broadcasts folded into the fmas (a quarter of the LDS reads, same VALU count)
                    const f32x4 own = *(const f32x4*)(Vt + base_own + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float acc1 = skp_quad_bcast<0>(own[e]) * wx[0];
                        skp_quad_fmac<1>(acc1, own[e], wx[1]);
                        skp_quad_fmac<2>(acc1, own[e], wx[2]);
                        skp_quad_fmac<3>(acc1, own[e], wx[3]);
                        v[e] = acc1;
                    }
                } else {
                    const f32x4 t0 = *(const f32x4*)(Vt + base[0] + 4 * q);
                    const f32x4 t1 = *(const f32x4*)(Vt + base[1] + 4 * q);
                    const f32x4 t2 = *(const f32x4*)(Vt + base[2] + 4 * q);
                    const f32x4 t3 = *(const f32x4*)(Vt + base[3] + 4 * q);
                    v = wx[0] * t0;
                    v = wx[1] * t1 + v;
                    v = wx[2] * t2 + v;
                    v = wx[3] * t3 + v;
                }
                if (4 * q >= NT - 16) {                         // NT = 16*ceil(T/16): only the last 16 can be pads
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * q + e >= T) v[e] = -INFINITY;
                }
                sv[2 * q] = f32x2{v[0], v[1]};
                sv[2 * q + 1] = f32x2{v[2], v[3]};
                m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            }
            const size_t li = ((size_t)b * a.L * H + lh) * RR + p;
            if (MODE == 2) {                                   // probabilities against the GLOBAL log-sum-exp
                const float lse = tl.valid ? lse_in[li] : 0.f;
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const f32x2 e = sv[u] - lse;
                    acc[u] += f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                }
                continue;
            }
            f32x2 sum4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const f32x2 e = sv[u] - m;
                sv[u] = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                sum4[u & 3] += sv[u];
            }
            const f32x2 sum2 = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
            const float sum = sum2[0] + sum2[1];
            if (MODE == 0) {
                const float inv = 1.0f / sum;
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[u] = sv[u] * inv + acc[u];
            }
            if (tl.valid) lse_out[li] = m + __builtin_amdgcn_logf(sum);
        }
    }
    if (tl.valid && MODE != 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < NT - 16 || t < T) M[(size_t)b * a.m_bstride + (size_t)t * RR + p] = acc[t >> 1][t & 1] * a.inv_lh;
    }
}

// Backward, kernel A.  LDS: Vt | dSx[256][TC+1] | Wt[smax][TW] | xlo[smax] xhi[smax] | row tables.
// QUAD selects the layer class a launch covers (a.quad[l]): the two H-phase / adjoint forms do not fit one register budget
template <int NT, int MODE, bool QUAD>
__global__ __launch_bounds__(256) void skp_attn_map_bwd_kernel(MapArgs a, const float* __restrict__ dM,
                                                               const float* __restrict__ lse_in,
                                                               float* __restrict__ dV, float* __restrict__ dot_io) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TS = NT + 4;                                 // token quads are 16-byte aligned (ds_read_b128)
    constexpr int NCH = (NT > 48) ? 2 : 1;                     // the transpose buffer holds NT/NCH tokens at a time
    constexpr int TC = NT / NCH, TSC = TC + 1, TQ = TC / 4;
    const int tid = threadIdx.x, b = blockIdx.x;            // batch row fastest: workgroup id % 8 (XCD) == b % 8
    const int R = a.R, T = a.T, H = a.H, RR = R * R;
    const Tile tl = skp_tile(a, blockIdx.y, tid);
    const int p = (tl.y0 + tl.ry) * R + tl.x;
    const int xl = tl.x - tl.seg * 256;                        // column inside the tile
    float* Vt = smem;
    float* dSx = smem + a.vt_floats;
    float* Wt = dSx + 256 * TSC;
    int* xlo = (int*)(Wt + a.smax * a.TW);
    int* xhi = xlo + a.smax;
    int* tab_cy = xhi + a.smax;
    float* tab_wy = (float*)(tab_cy + a.TH * 4);

    float g[NT];                                               // dM / (L*H), lane-local
#pragma unroll
    for (int t = 0; t < NT; ++t)
        g[t] = (tl.valid && t < T) ? dM[(size_t)b * a.m_bstride + (size_t)t * RR + p] * a.inv_lh : 0.f;

    int lh = 0;
    for (int l = 0; l < a.L; ++l) {
        if ((a.quad[l] != 0) != QUAD) { lh += H; continue; }   // the other launch's layers
        const int s = a.s[l];
        const float ratio = (float)s / (float)R;
        int cx[4]; float wx[4];
        skp_cubic_taps(tl.x, ratio, s, cx, wx);
        int base[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) base[i] = (tl.ry * s + cx[i]) * TS;
        constexpr bool quad = QUAD;
        const int base_own = base[tid & 3];
        __syncthreads();                                       // previous layer done with tables / Wt
        if (tid < tl.th_eff) {
            int cy[4]; float wy[4];
            skp_cubic_taps(tl.y0 + tid, ratio, s, cy, wy);
#pragma unroll
            for (int j = 0; j < 4; ++j) { tab_cy[tid * 4 + j] = cy[j]; tab_wy[tid * 4 + j] = wy[j]; }
        }
        // quad layers: lane i of a pixel quad ends up holding the quad's weighted sum for ITS tap column (below), so the
        // gather of column c is a plain sum over the (pixel) slots whose tap column is c: an index list per column,
        // built in ascending pixel order by one thread per column (deterministic), in the space of the weight table
        int* lst = (int*)Wt;                                   // [s][X_CAP] slots, xhi[c] = count
        float wq[4] = {0.f, 0.f, 0.f, 0.f};                    // weight quad lane j gives to MY tap column
        if (quad) {
            if (tid < s) {
                int n = 0;
                for (int x2 = 0; x2 < a.TW; ++x2) {
                    int c2[4]; float w2[4];
                    skp_cubic_taps(tl.seg * 256 + x2, ratio, s, c2, w2);
                    if (c2[x2 & 3] == tid && n < SKP_MAP_XCAP) lst[tid * SKP_MAP_XCAP + n++] = x2;
                }
                xhi[tid] = n;
            }
            const int me = tid & 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float c0 = skp_quad_bcast<0>(wx[k]), c1 = skp_quad_bcast<1>(wx[k]);
                const float c2 = skp_quad_bcast<2>(wx[k]), c3 = skp_quad_bcast<3>(wx[k]);
                if (me == k) { wq[0] = c0; wq[1] = c1; wq[2] = c2; wq[3] = c3; }
            }
        } else {
            for (int i = tid; i < s * a.TW; i += 256) Wt[i] = 0.f;
            if (tid < s) { xlo[tid] = 0x7fffffff; xhi[tid] = -1; }
            __syncthreads();
            if (tl.valid && tl.ry == 0) {                      // transpose of the horizontal taps: Wt[c][x]
#pragma unroll
                for (int i = 0; i < 4; ++i) {                  // same-lane adds are program-ordered => deterministic
                    atomicAdd(&Wt[cx[i] * a.TW + xl], wx[i]);
                    atomicMin(&xlo[cx[i]], xl);
                    atomicMax(&xhi[cx[i]], xl);
                }
            }
        }
        const int rc = tl.th_eff * s;
        const float inv_s = 1.0f / (float)s;
        for (int h = 0; h < H; ++h, ++lh) {
            const float* Sg = a.S[l] + ((size_t)(b * H + h) * s * s) * a.ldt;
            __syncthreads();                                   // tables/Wt ready; previous gather done
            skp_v_phase<NT>(Sg, Vt, tab_cy, tab_wy, s, rc, tid, a.ldt);
            __syncthreads();
            const size_t li = ((size_t)b * a.L * H + lh) * RR + p;
            const float lse = tl.valid ? lse_in[li] : 0.f;
            float sv[NT];
            f32x4 dot4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NT / 4; ++q) {                  // token quads: ds_read_b128 per tap, packed fp32 math
                f32x4 v;
                if (quad) {                                     // one read per quad lane, taps by DPP (see the forward kernel)
                    const f32x4 own = *(const f32x4*)(Vt + base_own + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float acc1 = skp_quad_bcast<0>(own[e]) * wx[0];
                        skp_quad_fmac<1>(acc1, own[e], wx[1]);
                        skp_quad_fmac<2>(acc1, own[e], wx[2]);
                        skp_quad_fmac<3>(acc1, own[e], wx[3]);
                        v[e] = acc1;
                    }
                } else {
                    const f32x4 t0 = *(const f32x4*)(Vt + base[0] + 4 * q);
                    const f32x4 t1 = *(const f32x4*)(Vt + base[1] + 4 * q);
                    const f32x4 t2 = *(const f32x4*)(Vt + base[2] + 4 * q);
                    const f32x4 t3 = *(const f32x4*)(Vt + base[3] + 4 * q);
                    v = wx[0] * t0;
                    v = wx[1] * t1 + v;
                    v = wx[2] * t2 + v;
                    v = wx[3] * t3 + v;
                }
                v = v - lse;
                f32x4 pr = {__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1]), __builtin_amdgcn_exp2f(v[2]),
                            __builtin_amdgcn_exp2f(v[3])};
                if (4 * q >= NT - 16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * q + e >= T) pr[e] = 0.f;
                }
                const f32x4 g4 = {g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]};
                dot4 += pr * g4;
                sv[4 * q] = pr[0]; sv[4 * q + 1] = pr[1]; sv[4 * q + 2] = pr[2]; sv[4 * q + 3] = pr[3];
            }
            float dot = (dot4[0] + dot4[1]) + (dot4[2] + dot4[3]);
            if (MODE == 1) { if (tl.valid) dot_io[li] = dot; continue; }   // statistics pass of a token group
            if (MODE == 2) dot = tl.valid ? dot_io[li] : 0.f;               // sum over ALL token groups
            float* dVg = dV + (size_t)b * a.dv_per_b + a.dv_off[l] +
                         ((size_t)h * R + tl.y0) * a.segs * s * NT;          // [h][y][seg][c][t]
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch) __syncthreads();                       // previous chunk's gather done
                if (quad) {
                    // lane i of each pixel quad: sum_j w_j[tap i] * dS_j  (four DPP fmas), stored in the lane's own slot
#pragma unroll
                    for (int tt = 0; tt < TC; ++tt) {
                        const int t = ch * TC + tt;
                        const float ds = sv[t] * (g[t] - dot);
                        float pq = skp_quad_bcast<0>(ds) * wq[0];
                        skp_quad_fmac<1>(pq, ds, wq[1]);
                        skp_quad_fmac<2>(pq, ds, wq[2]);
                        skp_quad_fmac<3>(pq, ds, wq[3]);
                        dSx[tid * TSC + tt] = pq;
                    }
                } else {
#pragma unroll
                    for (int tt = 0; tt < TC; ++tt) {
                        const int t = ch * TC + tt;
                        dSx[tid * TSC + tt] = tl.valid ? sv[t] * (g[t] - dot) : 0.f;
                    }
                }
                __syncthreads();
                // gather: dV[row][c][t] = sum_x Wt[c][x] * dS[row][x][t]; each item = (row, c, 4 strided tokens)
                const int items = rc * TQ;
                for (int it = tid; it < items; it += 256) {
                    const int r = it / TQ, tq = it - r * TQ;
                    const int row = (int)(((float)r + 0.5f) * inv_s);
                    const int c = r - row * s;
                    const float* d = dSx + (size_t)(row * a.TW) * TSC + tq;
                    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
                    if (quad) {
                        const int n = xhi[c];
                        const int* li2 = lst + c * SKP_MAP_XCAP;
                        for (int e = 0; e < n; ++e) {
                            const float* dd = d + li2[e] * TSC;
                            o0 += dd[0]; o1 += dd[TQ]; o2 += dd[2 * TQ]; o3 += dd[3 * TQ];
                        }
                    } else {
                        const int lo = xlo[c], hi = xhi[c];
                        const float* w = Wt + c * a.TW;
                        for (int xx = lo; xx <= hi; ++xx) {
                            const float ww = w[xx];
                            const float* dd = d + xx * TSC;
                            o0 = fmaf(ww, dd[0], o0); o1 = fmaf(ww, dd[TQ], o1);
                            o2 = fmaf(ww, dd[2 * TQ], o2); o3 = fmaf(ww, dd[3 * TQ], o3);
                        }
                    }
                    float* out = dVg + ((size_t)(row * a.segs + tl.seg) * s + c) * NT + ch * TC + tq;
                    out[0] = o0; out[TQ] = o1; out[2 * TQ] = o2; out[3 * TQ] = o3;
                }
            }
        }
    }
}

// Backward, kernel B: vertical adjoint.  grid = (sum_l H*s_l, B); one workgroup per (layer, head, low-res row).
struct VAdjArgs {
    float* dS[SKP_MAX_LAYERS];
    int s[SKP_MAX_LAYERS];
    long dv_off[SKP_MAX_LAYERS];
    int blk_off[SKP_MAX_LAYERS + 1];   // first block of layer l
    long dv_per_b;
    int L, H, R, segs, NT;
    int ldt;                           // row stride of dS (floats)
};

__global__ __launch_bounds__(256) void skp_attn_map_vadj_kernel(VAdjArgs a, const float* __restrict__ dV) {
    const int tid = threadIdx.x, b = blockIdx.x;
    int l = 0;
    while (l + 1 < a.L && (int)blockIdx.y >= a.blk_off[l + 1]) ++l;
    const int s = a.s[l], NT = a.NT, Q = NT / 4, R = a.R;
    const int rel = blockIdx.y - a.blk_off[l];
    const int h = rel / s, cy = rel - h * s;
    const float ratio = (float)s / (float)R;
    // rows y whose taps can touch cy: src(y) in (cy-2.5, cy+2.5) (+ everything beyond the clamped borders)
    int ylo = (int)floorf(((float)cy - 2.0f) / ratio) - 2, yhi = (int)ceilf(((float)cy + 3.0f) / ratio) + 2;
    if (cy == 0) ylo = 0;
    if (cy == s - 1) yhi = R - 1;
    ylo = ylo < 0 ? 0 : ylo; yhi = yhi > R - 1 ? R - 1 : yhi;
    const float* dVg = dV + (size_t)b * a.dv_per_b + a.dv_off[l] + (size_t)h * R * a.segs * s * NT;
    float* out = a.dS[l] + (((size_t)b * a.H + h) * s * s + (size_t)cy * s) * a.ldt;
    // each thread owns up to MAXI float4 slots (it = c*Q + q4: contiguous floats of one dV row)
    constexpr int MAXI = 8;                                    // s <= 64, NT <= 128 => s*Q <= 2048 = 8*256
    const int nit = s * Q;
    f32x4 acc[MAXI];
#pragma unroll
    for (int u = 0; u < MAXI; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int y = ylo; y <= yhi; ++y) {
        int ty[4]; float wy[4];
        skp_cubic_taps(y, ratio, s, ty, wy);                   // uniform across the workgroup
        float w = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) w += (ty[j] == cy) ? wy[j] : 0.f;
        if (w == 0.f) continue;
        for (int sg = 0; sg < a.segs; ++sg) {
            const float* row = dVg + ((size_t)(y * a.segs + sg) * s) * NT;
#pragma unroll
            for (int u = 0; u < MAXI; ++u) {
                const int it = tid + u * 256;
                if (it < nit) acc[u] += w * *(const f32x4*)(row + it * 4);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXI; ++u) {
        const int it = tid + u * 256;                           // it = c*Q + q4
        if (it < nit) { const int c = it / Q, q4 = it - c * Q; *(f32x4*)(out + (size_t)c * a.ldt + q4 * 4) = acc[u]; }
    }
}

// ---------------------------------------------------------------------------------------------
static int fill_args(MapArgs& a, const float* const* S, float* const* dS, const int* s, int L, int B, int H,
                     int T, int R) {
    if (!S || !s || L <= 0 || B <= 0 || H <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (L > SKP_MAX_LAYERS || T > 128 || R > 4096) return SKP_E_RANGE;
    const int nt = ((T + 15) / 16) * 16;
    int smax = 0;
    long off = 0;
    a.R = R;
    if (R <= 256) { a.TH = 256 / R; a.TW = R; a.segs = 1; }
    else { a.TH = 1; a.TW = 256; a.segs = (R + 255) / 256; }
    for (int l = 0; l < L; ++l) {
        if (!S[l] || s[l] <= 0) return SKP_E_BADARG;
        if (s[l] > 64) return SKP_E_RANGE;
        a.S[l] = S[l];
        a.dS[l] = dS ? dS[l] : nullptr;
        if (dS && !dS[l]) return SKP_E_BADARG;
        a.s[l] = s[l];
        a.quad[l] = 0;
        if (R % s[l] == 0 && ((R / s[l]) % 8) == 0 && R / s[l] <= 32 && (R <= 256 ? 256 % R == 0 : R % 256 == 0) &&
            (R <= 256 ? R : 256) >= SKP_MAP_XCAP)
            a.quad[l] = 1;
        a.dv_off[l] = off;
        off += (long)H * R * a.segs * s[l] * nt;
        smax = s[l] > smax ? s[l] : smax;
    }
    a.dv_per_b = off;
    a.L = L; a.B = B; a.H = H; a.T = T; a.smax = smax;
    a.vt_floats = a.TH * smax * (nt + 4);
    a.vt_floats = (a.vt_floats + 3) & ~3;
    a.inv_lh = 1.0f / (float)(L * H);
    return 0;
}

static int n_tiles(const MapArgs& a) { return a.R <= 256 ? (a.R + a.TH - 1) / a.TH : a.R * a.segs; }

#define SKP_NT_SWITCH(nt, MACRO)                                                                        \
    switch (nt) {                                                                                       \
        case 16: MACRO(16) break; case 32: MACRO(32) break; case 48: MACRO(48) break;                    \
        case 64: MACRO(64) break; case 80: MACRO(80) break; case 96: MACRO(96) break;                    \
        case 112: MACRO(112) break; case 128: MACRO(128) break;                                         \
        default: return SKP_E_RANGE;                                                                    \
    }

static int n_tiles_ok(const MapArgs& a) { return n_tiles(a) <= 65535; }

// Extended entry points: the launch covers T tokens starting at column offset folded into the S/dS/M/dM
// pointers by the caller; `ldt` = logits row stride, `m_bstride` = floats between batch rows of M/dM.
extern "C" int skp_attn_map_fwd_ex_f32(const float* const* S, const int* s, int L, int B, int H, int T, int R,
                                       float* M, float* lse_out, const float* lse_in, int ldt,
                                       int64_t m_bstride, int mode, void* stream) {
    MapArgs a{};
    int rc = fill_args(a, S, nullptr, s, L, B, H, T, R);
    if (rc) return rc;
    const int nt = ((T + 15) / 16) * 16;
    if (mode < 0 || mode > 2 || ldt < nt || (ldt & 3)) return SKP_E_BADARG;
    if ((mode != 1 && !M) || (mode != 2 && !lse_out) || (mode == 2 && !lse_in)) return SKP_E_BADARG;
    if (!n_tiles_ok(a)) return SKP_E_RANGE;
    a.ldt = ldt; a.m_bstride = m_bstride; a.mode = mode;
    const int ntile = n_tiles(a);
    if (ntile > 65535) return SKP_E_RANGE;
    const size_t lds = ((size_t)a.vt_floats + 8 * (size_t)a.TH) * sizeof(float);
    if (lds > 160 * 1024) return SKP_E_LDS;
    dim3 grid(B, ntile), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SKP_FWD_K(KERNEL, NTV, MD)                                                                       \
    {                                                                                                    \
        if (lds > 64 * 1024) {                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL<NTV, MD>,                             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (e != hipSuccess) return (int)e;                                                          \
        }                                                                                                \
        hipLaunchKernelGGL((KERNEL<NTV, MD>), grid, block, lds, st, a, M, lse_out, lse_in);              \
    }
#define SKP_FWD_M(NTV, MD) SKP_FWD_K(skp_attn_map_fwd_kernel, NTV, MD)
#define SKP_FWD(NTV)                                                                                     \
    if (mode == 0) { SKP_FWD_M(NTV, 0) } else if (mode == 1) { SKP_FWD_M(NTV, 1) } else { SKP_FWD_M(NTV, 2) }
    SKP_NT_SWITCH(nt, SKP_FWD)
#undef SKP_FWD
#undef SKP_FWD_M
#undef SKP_FWD_K
    return skp_launch_status();
}

extern "C" int skp_attn_map_fwd_f32(const float* const* S, const int* s, int L, int B, int H, int T, int R,
                                    float* M, float* lse, void* stream) {
    if (T > 128 || T <= 0 || R <= 0) return T > 128 ? SKP_E_RANGE : SKP_E_BADARG;
    return skp_attn_map_fwd_ex_f32(S, s, L, B, H, T, R, M, lse, nullptr, ((T + 15) / 16) * 16,
                                   (int64_t)T * R * R, 0, stream);
}

extern "C" int64_t skp_attn_map_bwd_workspace(const int* s, int L, int B, int H, int T, int R) {
    if (!s || L <= 0 || L > SKP_MAX_LAYERS || B <= 0 || H <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    const int nt = ((T + 15) / 16) * 16;
    const int segs = R <= 256 ? 1 : (R + 255) / 256;
    int64_t fl = 0;
    for (int l = 0; l < L; ++l) fl += (int64_t)H * R * segs * s[l] * nt;
    return fl * B * (int64_t)sizeof(float);
}

extern "C" int skp_attn_map_bwd_ex_f32(const float* const* S, float* const* dS, const int* s, int L, int B, int H,
                                       int T, int R, const float* dM, const float* lse, float* workspace,
                                       float* dot_io, int ldt, int64_t m_bstride, int mode, void* stream) {
    MapArgs a{};
    if (!dS) return SKP_E_BADARG;
    int rc = fill_args(a, S, dS, s, L, B, H, T, R);
    if (rc) return rc;
    const int nt = ((T + 15) / 16) * 16;
    if (!dM || !lse || !workspace || mode < 0 || mode > 2 || (mode != 0 && !dot_io) || ldt < nt || (ldt & 3))
        return SKP_E_BADARG;
    if (!n_tiles_ok(a)) return SKP_E_RANGE;
    a.ldt = ldt; a.m_bstride = m_bstride; a.mode = mode;
    const int tc = nt > 48 ? nt / 2 : nt;
    const size_t lds = ((size_t)a.vt_floats + 256 * (size_t)(tc + 1) + (size_t)a.smax * a.TW + 2 * (size_t)a.smax +
                        8 * (size_t)a.TH) * sizeof(float);
    if (lds > 160 * 1024) return SKP_E_LDS;
    dim3 grid(B, n_tiles(a)), block(256);
    hipStream_t st = (hipStream_t)stream;
    bool any_quad = false, any_plain = false;                  // one launch per layer class (the map gradient is read by both)
    for (int l = 0; l < L; ++l) { if (a.quad[l]) any_quad = true; else any_plain = true; }
#define SKP_BWD_Q(NTV, MD, QD)                                                                           \
    {                                                                                                    \
        if (lds > 64 * 1024) {                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)skp_attn_map_bwd_kernel<NTV, MD, QD>,        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            if (e != hipSuccess) return (int)e;                                                          \
        }                                                                                                \
        hipLaunchKernelGGL((skp_attn_map_bwd_kernel<NTV, MD, QD>), grid, block, lds, st, a, dM, lse, workspace, dot_io); \
    }
#define SKP_BWD_M(NTV, MD)                                                                               \
    {                                                                                                    \
        if (any_plain) SKP_BWD_Q(NTV, MD, false)                                                         \
        if (any_quad) SKP_BWD_Q(NTV, MD, true)                                                           \
    }
#define SKP_BWD(NTV)                                                                                     \
    if (mode == 0) SKP_BWD_M(NTV, 0) else if (mode == 1) SKP_BWD_M(NTV, 1) else SKP_BWD_M(NTV, 2)
    SKP_NT_SWITCH(nt, SKP_BWD)
#undef SKP_BWD
#undef SKP_BWD_M
#undef SKP_BWD_Q
    rc = skp_launch_status();
    if (rc || mode == 1) return rc;
    VAdjArgs v{};
    int nblk = 0;
    for (int l = 0; l < L; ++l) {
        v.dS[l] = a.dS[l]; v.s[l] = a.s[l]; v.dv_off[l] = a.dv_off[l]; v.blk_off[l] = nblk;
        nblk += H * a.s[l];
    }
    v.blk_off[L] = nblk;
    v.dv_per_b = a.dv_per_b; v.L = L; v.H = H; v.R = R; v.segs = a.segs; v.NT = nt; v.ldt = ldt;
    hipLaunchKernelGGL(skp_attn_map_vadj_kernel, dim3(B, nblk), dim3(256), 0, st, v, (const float*)workspace);
    return skp_launch_status();
}

extern "C" int skp_attn_map_bwd_f32(const float* const* S, float* const* dS, const int* s, int L, int B, int H,
                                    int T, int R, const float* dM, const float* lse, float* workspace,
                                    void* stream) {
    if (T > 128 || T <= 0 || R <= 0) return T > 128 ? SKP_E_RANGE : SKP_E_BADARG;
    return skp_attn_map_bwd_ex_f32(S, dS, s, L, B, H, T, R, dM, lse, workspace, nullptr, ((T + 15) / 16) * 16,
                                   (int64_t)T * R * R, 0, stream);
}
