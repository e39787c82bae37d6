// Residual add + LayerNorm of the frozen transformer blocks (diffusers BasicTransformerBlock [third party], reached from the
// hooked UNet forward, ptp_utils.py:213-217):   x = d + h;  n = (x - mean) * rstd * gamma + beta      per token row.
// One pass per direction instead of (add, LayerNorm) / (LayerNorm backward, gradient-accumulation add):
//   forward : reads d, h       writes x, n (+ mean, rstd per row)
//   backward: reads dn, x (+ the gradient that reaches x from its other uses)   writes dx = LN'(dn) + dskip
// A row of C floats is spread over LPR = 16 / 32 / 64 lanes (8 for the small test trees) holding V float4 each, so a wave
// carries 64 / LPR rows: the 320-wide rows of the 64^2 layers are 4 rows per wave with 5 independent 16-byte loads per lane
// and operand in flight -- one wave per row (1.25 float4 per lane) was latency-bound.  Two-pass statistics in registers
// (mean, then sum of squared deviations); row reductions are xor-butterflies inside the LPR-lane group: fixed order,
// bit-reproducible.  gamma / beta are frozen (no gradient).
#include "skp_common.h"

namespace {

template <int LPR>
__device__ __forceinline__ float ln_row_sum(float v) {
#pragma unroll
    for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct LNArgs {
    const float* d;        // may be null (plain LayerNorm of h)
    const float* h;
    const float* gamma;
    const float* beta;
    float* x;              // d + h (unused when d is null)
    float* n;
    float* stat;           // [rows][2] = (mean, rstd)
    int rows, C;
    float eps;
};

template <int LPR, int V>
__global__ __launch_bounds__(256) void skp_add_ln_fwd_kernel(LNArgs a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + wave) * RPW + lane / LPR;
    if (row >= a.rows) return;                       // whole LPR-lane groups leave together
    const long base = row * a.C;
    f32x4 x[V];
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = *(const f32x4*)(a.h + base + (v * LPR + j) * 4);
    if (a.d) {
        f32x4 dd[V];
#pragma unroll
        for (int v = 0; v < V; ++v) dd[v] = *(const f32x4*)(a.d + base + (v * LPR + j) * 4);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            x[v] += dd[v];
            *(f32x4*)(a.x + base + (v * LPR + j) * 4) = x[v];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
    const float inv = 1.0f / (float)a.C;
    const float mean = ln_row_sum<LPR>(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = x[v][e] - mean; q = fmaf(t, t, q); }
    const float rstd = rsqrtf(ln_row_sum<LPR>(q) * inv + a.eps);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const f32x4 g = *(const f32x4*)(a.gamma + (v * LPR + j) * 4), b = *(const f32x4*)(a.beta + (v * LPR + j) * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf((x[v][e] - mean) * rstd, g[e], b[e]);
        *(f32x4*)(a.n + base + (v * LPR + j) * 4) = o;
    }
    if (j == 0) *(f32x2*)(a.stat + row * 2) = f32x2{mean, rstd};
}

struct LNBwdArgs {
    const float* dn;
    const float* dskip;    // may be null
    const float* x;
    const float* stat;
    const float* gamma;
    float* dx;
    int rows, C;
};

template <int LPR, int V>
__global__ __launch_bounds__(256) void skp_add_ln_bwd_kernel(LNBwdArgs a) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + wave) * RPW + lane / LPR;
    if (row >= a.rows) return;
    const long base = row * a.C;
    const f32x2 st = *(const f32x2*)(a.stat + row * 2);
    f32x4 g[V], xh[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        g[v] = *(const f32x4*)(a.dn + base + (v * LPR + j) * 4);
        xh[v] = *(const f32x4*)(a.x + base + (v * LPR + j) * 4);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const f32x4 gm = *(const f32x4*)(a.gamma + (v * LPR + j) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[v][e] *= gm[e];
            xh[v][e] = (xh[v][e] - st[0]) * st[1];
            s1 += g[v][e];
            s2 = fmaf(g[v][e], xh[v][e], s2);
        }
    }
    const float inv = 1.0f / (float)a.C;
    const float m1 = ln_row_sum<LPR>(s1) * inv, m2 = ln_row_sum<LPR>(s2) * inv;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = st[1] * (g[v][e] - m1 - xh[v][e] * m2);
        if (a.dskip) o += *(const f32x4*)(a.dskip + base + (v * LPR + j) * 4);
        *(f32x4*)(a.dx + base + (v * LPR + j) * 4) = o;
    }
}

// lanes per row and float4 per lane for a row width, or false
bool ln_plan(int C, int& lpr, int& v) {
    if (C <= 0 || (C & 3)) return false;
    for (int need = 2; need >= 1; --need)            // the widest lane group that still leaves two float4 per lane, else one
        for (int l : {64, 32, 16, 8}) {
            if (C % (4 * l)) continue;
            const int vv = C / (4 * l);
            if (vv >= need && vv <= 8 && vv != 7) { lpr = l; v = vv; return true; }
        }
    return false;
}

template <int LPR, int V>
int ln_launch_fwd(const LNArgs& a, hipStream_t st) {
    const long rpb = 4 * (64 / LPR);
    hipLaunchKernelGGL((skp_add_ln_fwd_kernel<LPR, V>), dim3((unsigned)((a.rows + rpb - 1) / rpb)), dim3(256), 0, st, a);
    return skp_launch_status();
}
template <int LPR, int V>
int ln_launch_bwd(const LNBwdArgs& a, hipStream_t st) {
    const long rpb = 4 * (64 / LPR);
    hipLaunchKernelGGL((skp_add_ln_bwd_kernel<LPR, V>), dim3((unsigned)((a.rows + rpb - 1) / rpb)), dim3(256), 0, st, a);
    return skp_launch_status();
}

#define LN_DISPATCH(FN, ARGS)                                                                      \
    switch (lpr * 16 + v) {                                                                        \
        case 64 * 16 + 1: return FN<64, 1>(ARGS, st); case 64 * 16 + 2: return FN<64, 2>(ARGS, st); \
        case 64 * 16 + 3: return FN<64, 3>(ARGS, st); case 64 * 16 + 4: return FN<64, 4>(ARGS, st); \
        case 64 * 16 + 5: return FN<64, 5>(ARGS, st); case 64 * 16 + 6: return FN<64, 6>(ARGS, st); \
        case 64 * 16 + 8: return FN<64, 8>(ARGS, st);                                               \
        case 32 * 16 + 1: return FN<32, 1>(ARGS, st); case 32 * 16 + 2: return FN<32, 2>(ARGS, st); \
        case 32 * 16 + 3: return FN<32, 3>(ARGS, st); case 32 * 16 + 4: return FN<32, 4>(ARGS, st); \
        case 32 * 16 + 5: return FN<32, 5>(ARGS, st); case 32 * 16 + 6: return FN<32, 6>(ARGS, st); \
        case 32 * 16 + 8: return FN<32, 8>(ARGS, st);                                               \
        case 16 * 16 + 1: return FN<16, 1>(ARGS, st); case 16 * 16 + 2: return FN<16, 2>(ARGS, st); \
        case 16 * 16 + 3: return FN<16, 3>(ARGS, st); case 16 * 16 + 4: return FN<16, 4>(ARGS, st); \
        case 16 * 16 + 5: return FN<16, 5>(ARGS, st); case 16 * 16 + 6: return FN<16, 6>(ARGS, st); \
        case 16 * 16 + 8: return FN<16, 8>(ARGS, st);                                               \
        case 8 * 16 + 1: return FN<8, 1>(ARGS, st); case 8 * 16 + 2: return FN<8, 2>(ARGS, st);     \
        case 8 * 16 + 3: return FN<8, 3>(ARGS, st); case 8 * 16 + 4: return FN<8, 4>(ARGS, st);     \
        case 8 * 16 + 5: return FN<8, 5>(ARGS, st); case 8 * 16 + 6: return FN<8, 6>(ARGS, st);     \
        case 8 * 16 + 8: return FN<8, 8>(ARGS, st);                                                 \
        default: return SKP_E_RANGE;                                                               \
    }

}  // namespace

extern "C" int skp_add_layer_norm_ok(int C) {
    int lpr, v;
    return ln_plan(C, lpr, v) ? 1 : 0;
}

extern "C" int skp_add_layer_norm_fwd_f32(const float* d, const float* h, const float* gamma, const float* beta, float* x,
                                          float* n, float* stat, int64_t rows, int C, float eps, void* stream) {
    if (!h || !gamma || !beta || !n || !stat || rows <= 0 || (d && !x)) return SKP_E_BADARG;
    int lpr, v;
    if (!ln_plan(C, lpr, v) || rows > 0x7fffffffLL) return SKP_E_RANGE;
    LNArgs a{d, h, gamma, beta, x, n, stat, (int)rows, C, eps};
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(ln_launch_fwd, a)
}

extern "C" int skp_add_layer_norm_bwd_f32(const float* dn, const float* dskip, const float* x, const float* stat,
                                          const float* gamma, float* dx, int64_t rows, int C, void* stream) {
    if (!dn || !x || !stat || !gamma || !dx || rows <= 0) return SKP_E_BADARG;
    int lpr, v;
    if (!ln_plan(C, lpr, v) || rows > 0x7fffffffLL) return SKP_E_RANGE;
    LNBwdArgs a{dn, dskip, x, stat, gamma, dx, (int)rows, C};
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(ln_launch_bwd, a)
}
