// Fused up-res attention map, forward, for a WIDE token axis (T > 128; the reference CLI default is --num_tokens 500,
// main.py:77-79) in ONE pass.
//
// Reference path replaced: ptp_utils.py:513-538 + optimize.py:27-79 (see skp_attn_map.hip for the algebra).  The
// T <= 128 kernel keeps all tokens of a pixel in one lane; with more tokens that form needed token groups and TWO passes
// (group statistics, then apply: every logit up-sampled and exponentiated twice).  Here the token axis is cut into
// 32- or 64-token SLICES laid across the lanes instead:
//   workgroup = 32 consecutive pixels of one up-res row x NS = ceil(T/SW) slices; lane = (pixel, slice)
//   per (layer, head):
//     V phase   Vt[c][t] = sum_j wy[j] S[cy[j]][c][t] for the <= 32 s/R + 4 source columns the 32 pixels touch -> LDS
//     H phase   z_t = sum_i wx[i] Vt[cx[i]][t] for the lane's SW tokens (ds_read_b128 per tap and token quad);
//               slice max m, e_t = exp2(z_t - m), slice sum l -> (m, l) to LDS
//     combine   M = max_s m_s, L = sum_s l_s 2^(m_s - M): acc_t += e_t 2^(m - M) / L;  lse = M + log2 L
//   two barriers per (layer, head), the same as the narrow kernel; one coalesced write of M (128 B per token and
//   half-wave) at the end -- optionally only of the rows an index list asks for (inference keeps K of T maps:
//   optimize.py:58-59 `data[:, :, :, indices]`).
#include "skp_common.h"


struct WideArgs {
    const float* S[SKP_MAX_LAYERS];
    int s[SKP_MAX_LAYERS];
    int L, B, H, T, R, ldt, NS;
    int ncmax, vt_stride;            // most source columns of a tile; floats per column of Vt (SW*NS + 4)
    float inv_lh;
    long m_bstride;                  // floats between batch rows of M
};

// SW: slice width = tokens per lane; PX: pixels per workgroup (64 where R allows: one slice per wave, 256-byte stores,
// no bank conflicts between the slices of a wave)
template <int SW, int PX>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(SW == 32 ? 3 : 2)))
void skp_attn_map_fwd_wide_kernel(WideArgs a, float* __restrict__ M,
                                                                    float* __restrict__ lse_out,
                                                                    const int* __restrict__ tokrow) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int px = tid & (PX - 1), slice = tid / PX;
    const int R = a.R, T = a.T, H = a.H, RR = R * R, NS = a.NS;
    const int tiles_x = R / PX;
    const int y = blockIdx.y / tiles_x, x0 = (blockIdx.y - y * tiles_x) * PX;
    const int x = x0 + px;
    const bool act = slice < NS;                               // an odd NS leaves half a wave without a slice
    const int t0 = slice * SW;
    const int TS = a.vt_stride;
    float* Vt = smem;                                          // [ncmax][TS]
    f32x2* red = (f32x2*)(smem + a.ncmax * TS);                // [2][NS][PX]  (max, sum) per (slice, pixel)
    const int nthreads = blockDim.x;
    const int nt16 = ((T + 15) / 16) * 16, Q = nt16 / 4;

    constexpr int NP = SW / 2, NQ = SW / 4;
    f32x2 acc[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) acc[u] = f32x2{0.f, 0.f};

    int lh = 0;
    for (int l = 0; l < a.L; ++l) {
        const int s = a.s[l];
        const float ratio = (float)s / (float)R;
        int cx[4]; float wx[4];
        skp_cubic_taps(x, ratio, s, cx, wx);
        int cy[4]; float wy[4];
        skp_cubic_taps(y, ratio, s, cy, wy);                   // uniform across the workgroup
        int clo, chi;
        {
            int c0[4], c1[4]; float wdummy[4];
            skp_cubic_taps(x0, ratio, s, c0, wdummy);
            skp_cubic_taps(x0 + PX - 1, ratio, s, c1, wdummy);
            clo = c0[0]; chi = c1[3];
        }
        const int nc = chi - clo + 1;
        int base[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) base[i] = (cx[i] - clo) * TS + t0;
        for (int h = 0; h < H; ++h, ++lh) {
            const float* Sg = a.S[l] + ((size_t)(b * H + h) * s * s) * a.ldt;
            // ---- V phase (the previous head's H phase finished before its combine barrier) ----
            // all loads of a thread's items go out before the first is used (the phase is latency-, not bandwidth-bound)
            constexpr int VB = 3;
            for (int it0 = tid; it0 < nc * Q; it0 += VB * nthreads) {
                f32x4 raw[VB][4];
#pragma unroll
                for (int u = 0; u < VB; ++u) {
                    const int it = it0 + u * nthreads;
                    const int itc = it < nc * Q ? it : tid;
                    const int c = itc / Q, q4 = itc - c * Q;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        raw[u][j] = *(const f32x4*)(Sg + (size_t)(cy[j] * s + clo + c) * a.ldt + q4 * 4);
                }
#pragma unroll
                for (int u = 0; u < VB; ++u) {
                    const int it = it0 + u * nthreads;
                    if (it < nc * Q) {
                        const int c = it / Q, q4 = it - c * Q;
                        f32x4 v = wy[0] * raw[u][0];
#pragma unroll
                        for (int j = 1; j < 4; ++j) v += wy[j] * raw[u][j];
                        *(f32x4*)(Vt + c * TS + q4 * 4) = v;
                    }
                }
            }
            __syncthreads();
            // ---- H phase: 64 tokens of this lane ----
            f32x2 sv[NP];
            float m = -INFINITY;
            if (act) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4 v;
                    if (t0 + 4 * q < nt16) {
                        const f32x4 q0 = *(const f32x4*)(Vt + base[0] + 4 * q);
                        const f32x4 q1 = *(const f32x4*)(Vt + base[1] + 4 * q);
                        const f32x4 q2 = *(const f32x4*)(Vt + base[2] + 4 * q);
                        const f32x4 q3 = *(const f32x4*)(Vt + base[3] + 4 * q);
                        v = wx[0] * q0;
                        v = wx[1] * q1 + v;
                        v = wx[2] * q2 + v;
                        v = wx[3] * q3 + v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (t0 + 4 * q + e >= T) v[e] = -INFINITY;
                    } else {
                        v = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                    sv[2 * q] = f32x2{v[0], v[1]};
                    sv[2 * q + 1] = f32x2{v[2], v[3]};
                    m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                }
            }
            // a slice that holds pad tokens only has m = -inf: give it a finite max and a zero sum
            const float ms = (m == -INFINITY) ? -3.0e38f : m;
            f32x2 sum4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            if (act) {
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const f32x2 e = sv[u] - ms;
                    sv[u] = f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                    sum4[u & 3] += sv[u];
                }
            }
            const f32x2 sum2 = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
            const float lsum = sum2[0] + sum2[1];
            f32x2* rb = red + (size_t)(lh & 1) * NS * PX;
            if (act) rb[slice * PX + px] = f32x2{ms, lsum};
            __syncthreads();
            // ---- combine the slices of this pixel ----
            if (act) {
                float Mx = -3.0e38f;
                for (int sl = 0; sl < NS; ++sl) Mx = fmaxf(Mx, rb[sl * PX + px][0]);
                float Lt = 0.f;
                for (int sl = 0; sl < NS; ++sl) {
                    const f32x2 r = rb[sl * PX + px];
                    Lt = fmaf(r[1], __builtin_amdgcn_exp2f(r[0] - Mx), Lt);
                }
                const float scale = __builtin_amdgcn_exp2f(ms - Mx) / Lt;
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[u] = sv[u] * scale + acc[u];
                if (slice == 0) lse_out[((size_t)b * a.L * H + lh) * RR + (size_t)y * R + x] = Mx + __builtin_amdgcn_logf(Lt);
            }
        }
    }
    if (act) {
        float* Mb = M + (size_t)b * a.m_bstride + (size_t)y * R + x;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int t = t0 + 2 * u + e;
                if (t < T) {
                    const int row = tokrow ? tokrow[t] : t;    // uniform per half-wave
                    if (row >= 0) Mb[(size_t)row * RR] = acc[u][e] * a.inv_lh;
                }
            }
        }
    }
}

// Launch plan of the wide kernel; 0 or the SKP_E_* code the entry point would return for these sizes (host logic only).
struct WidePlan { int sw, px, threads; long tiles; size_t lds; };
static int wide_plan(const int* s, int L, int T, int R, WideArgs& a, WidePlan& p) {
    if (L <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (L > SKP_MAX_LAYERS || T > 1024 || R > 4096 || (R % 32)) return SKP_E_RANGE;
    int smax = 0;
    for (int l = 0; l < L; ++l) {
        if (s[l] <= 0) return SKP_E_BADARG;
        if (s[l] > 64) return SKP_E_RANGE;
        a.s[l] = s[l];
        smax = s[l] > smax ? s[l] : smax;
    }
    // 64-token slices measured faster than 32 (1.84 vs 2.58 ms at T = 500); 32 only where 64 would leave most lanes idle
    int sw = T <= 192 ? 32 : 64;
    int px = 32;                                               // (64-pixel tiles measured 4 % slower at T = 500)
    a.NS = (T + sw - 1) / sw;
    if (px * a.NS > 512) px = 32;                              // 512 threads: two waves per SIMD at 256 registers
    a.ncmax = (int)(((long)px * smax + R - 1) / R) + 4;
    if (a.ncmax > smax) a.ncmax = smax;
    a.vt_stride = sw * a.NS + 4;
    p.sw = sw; p.px = px;
    p.threads = ((px * a.NS + 63) / 64) * 64;
    p.tiles = (long)R * (R / px);
    if (p.tiles > 65535 || p.threads > 512) return SKP_E_RANGE;
    p.lds = ((size_t)a.ncmax * a.vt_stride + 2 * (size_t)a.NS * px * 2) * sizeof(float);
    if (p.lds > 160 * 1024) return SKP_E_LDS;
    return 0;
}

// 1 when skp_attn_map_fwd_wide_f32 takes these sizes (callers fall back to the two-pass token-group route otherwise).
extern "C" int skp_attn_map_fwd_wide_ok(const int* s, int L, int T, int R) {
    if (!s) return 0;
    WideArgs a{};
    WidePlan p{};
    return wide_plan(s, L, T, R, a, p) == 0 ? 1 : 0;
}

extern "C" int skp_attn_map_fwd_wide_f32(const float* const* S, const int* s, int L, int B, int H, int T, int R,
                                         float* M, float* lse, const int* tokrow, int n_rows, int ldt, void* stream) {
    if (!S || !s || !M || !lse || L <= 0 || B <= 0 || H <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    const int nt16 = ((T + 15) / 16) * 16;
    if (L <= SKP_MAX_LAYERS && (ldt < nt16 || (ldt & 3) || (tokrow && n_rows <= 0))) return SKP_E_BADARG;
    WideArgs a{};
    WidePlan p{};
    const int rc = wide_plan(s, L, T, R, a, p);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) {
        if (!S[l]) return SKP_E_BADARG;
        a.S[l] = S[l];
    }
    a.L = L; a.B = B; a.H = H; a.T = T; a.R = R; a.ldt = ldt;
    a.inv_lh = 1.0f / (float)(L * H);
    a.m_bstride = (long)(tokrow ? n_rows : T) * R * R;
    const int sw = p.sw, px = p.px, threads = p.threads;
    const long tiles = p.tiles;
    const size_t lds = p.lds;
#define SKP_WIDE_LAUNCH(SWV, PXV)                                                                              \
    {                                                                                                          \
        if (lds > 64 * 1024) {                                                                                 \
            hipError_t e = hipFuncSetAttribute((const void*)skp_attn_map_fwd_wide_kernel<SWV, PXV>,            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            if (e != hipSuccess) return (int)e;                                                                \
        }                                                                                                      \
        hipLaunchKernelGGL((skp_attn_map_fwd_wide_kernel<SWV, PXV>), dim3(B, (unsigned)tiles), dim3(threads), lds, \
                           (hipStream_t)stream, a, M, lse, tokrow);                                            \
    }
    if (sw == 32 && px == 32) SKP_WIDE_LAUNCH(32, 32)
    else if (sw == 32) SKP_WIDE_LAUNCH(32, 64)
    else if (px == 32) SKP_WIDE_LAUNCH(64, 32)
    else SKP_WIDE_LAUNCH(64, 64)
#undef SKP_WIDE_LAUNCH
    return skp_launch_status();
}
