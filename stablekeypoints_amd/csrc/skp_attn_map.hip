// Fused up-res cross-attention map: bicubic up-sampling of the low-res logits, softmax over the
// learned tokens in registers, head/layer mean, one coalesced [B,T,R,R] write -- and its backward.
//
// Reference path replaced: ptp_utils.py:513-538 (bicubic x -> R^2, to_q, einsum, softmax, clone,
// per-head store of (B*h, R^2, T)) + optimize.py:27-79 (reshape/permute/stack/mean).
// Algebra: to_q is bias-free and bicubic resize is linear, so
//   scale * to_q(bicubic(x))[p] . k[t]  ==  bicubic( scale * to_q(x) . k[t] )[p]
// i.e. the logits are up-sampled, not the activations: the 174.5 GF projection and the 11.3 GF
// up-res QK^T of the reference collapse to a 0.2 GF low-res QK^T (skp_gemm.hip, fp32 MFMA) plus
// 8 fma per (layer, head, token, pixel) here.  The kernel is VALU/LDS bound, not MFMA bound.
//
// Tiling: one workgroup = 256 consecutive pixels of the row-major R x R grid (lane = pixel, so
// every M[t, y, x..x+63] store is one 256-B line) x ALL tokens (softmax is lane-local).
// Per (layer, head):
//   V phase  all threads: Vt[r][c][t] = sum_j wy[r][j] * S[t][cy[r][j]][c]   (rows r of the tile,
//            every low-res column c) -> LDS, token-contiguous with stride NT+1 (conflict-free)
//   H phase  lane: s_t = sum_i wx[i] * Vt[r(lane)][cx[i]][t]  (4 LDS reads, immediate offsets)
//            m = max_t s_t; e_t = exp2(s_t - m); acc_t += e_t / sum_t e_t
// The logits arrive pre-multiplied by scale*log2(e), so exp2 is the bare v_exp_f32.
#include "skp_common.h"

struct MapArgs {
    const float* S[SKP_MAX_LAYERS];
    float* dS[SKP_MAX_LAYERS];
    int s[SKP_MAX_LAYERS];
    int L, B, H, T, R;
    int th_max;          // max tile rows
    int vt_floats;       // floats of one Vt buffer
    float inv_lh;
};

// Row/column tap tables live after the Vt buffer(s): cy[th_max*4] (int) then wy[th_max*4] (float).
template <int NT, bool BWD>
__global__ __launch_bounds__(256) void skp_attn_map_kernel(MapArgs a, float* __restrict__ M,
                                                           float* __restrict__ lse_out,
                                                           const float* __restrict__ dM,
                                                           const float* __restrict__ lse_in) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TS = NT + 1;                                 // token stride (odd => conflict-free)
    const int tid = threadIdx.x, b = blockIdx.y;
    const int R = a.R, T = a.T, H = a.H, RR = R * R;
    const int p0 = blockIdx.x * 256;
    const int p = p0 + tid;
    const bool valid = p < RR;
    const int pc = valid ? p : RR - 1;
    const int y = pc / R, x = pc - y * R;
    const int y0 = p0 / R;
    const int plast = (p0 + 255 < RR - 1) ? p0 + 255 : RR - 1;
    const int TH = plast / R - y0 + 1;
    const int ry = y - y0;

    float* Vt = smem;
    float* dVt = smem + a.vt_floats;                           // BWD only
    int* tab_cy = (int*)(smem + (BWD ? 2 : 1) * a.vt_floats);
    float* tab_wy = (float*)(tab_cy + a.th_max * 4);

    float acc[NT];                                             // FWD: map accumulator; BWD: g = dM/(L*H)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (BWD) acc[t] = (valid && t < T) ? dM[((size_t)b * T + t) * RR + p] * a.inv_lh : 0.f;
        else acc[t] = 0.f;
    }

    int lh = 0;
    for (int l = 0; l < a.L; ++l) {
        const int s = a.s[l];
        const float ratio = (float)s / (float)R;
        int cx[4]; float wx[4];
        skp_cubic_taps(x, ratio, s, cx, wx);
        int base[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) base[i] = (ry * s + cx[i]) * TS;
        __syncthreads();                                       // previous layer finished with the tables
        if (tid < TH) {
            int cy[4]; float wy[4];
            skp_cubic_taps(y0 + tid, ratio, s, cy, wy);
#pragma unroll
            for (int j = 0; j < 4; ++j) { tab_cy[tid * 4 + j] = cy[j]; tab_wy[tid * 4 + j] = wy[j]; }
        }
        const int rc = TH * s;
        const int nvt = rc * T;
        const float inv_rc = 1.0f / (float)rc, inv_s = 1.0f / (float)s;
        for (int h = 0; h < H; ++h, ++lh) {
            const float* Sg = a.S[l] + ((size_t)(b * H + h) * T) * s * s;
            __syncthreads();                                   // tables ready / previous H phase done
            // ---- V phase -------------------------------------------------------------------
            for (int idx = tid; idx < nvt; idx += 256) {
                const int t = (int)(((float)idx + 0.5f) * inv_rc);
                const int r = idx - t * rc;                    // r = row*s + c
                const int row = (int)(((float)r + 0.5f) * inv_s);
                const int c = r - row * s;
                const float* St = Sg + (size_t)t * s * s + c;
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v = fmaf(tab_wy[row * 4 + j], St[tab_cy[row * 4 + j] * s], v);
                Vt[r * TS + t] = v;
                if (BWD) dVt[r * TS + t] = 0.f;
            }
            __syncthreads();
            // ---- H phase + softmax over tokens (lane-local) ---------------------------------
            float sv[NT];
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v = wx[0] * Vt[base[0] + t];
                v = fmaf(wx[1], Vt[base[1] + t], v);
                v = fmaf(wx[2], Vt[base[2] + t], v);
                v = fmaf(wx[3], Vt[base[3] + t], v);
                sv[t] = (t < T) ? v : -INFINITY;
                m = fmaxf(m, sv[t]);
            }
            if (!BWD) {
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) { sv[t] = __builtin_amdgcn_exp2f(sv[t] - m); sum += sv[t]; }
                const float inv = 1.0f / sum;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = fmaf(sv[t], inv, acc[t]);
                if (valid) lse_out[((size_t)b * a.L * H + lh) * RR + p] = m + __builtin_amdgcn_logf(sum);
            } else {
                const float lse = lse_in[((size_t)b * a.L * H + lh) * RR + pc];
                float dot = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) { sv[t] = __builtin_amdgcn_exp2f(sv[t] - lse); dot = fmaf(sv[t], acc[t], dot); }
                // adjoint of the H phase: scatter w_i * dS_t into dVt (LDS fp32 atomics)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float ds = valid ? sv[t] * (acc[t] - dot) : 0.f;
                    if (t < T) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) atomicAdd(&dVt[base[i] + t], wx[i] * ds);
                    }
                }
                __syncthreads();
                // adjoint of the V phase: dS[t][cy][c] += wy * dVt  (global fp32 atomics; tiles overlap)
                float* dSg = a.dS[l] + ((size_t)(b * H + h) * T) * s * s;
                for (int idx = tid; idx < nvt; idx += 256) {
                    const int t = (int)(((float)idx + 0.5f) * inv_rc);
                    const int r = idx - t * rc;
                    const int row = (int)(((float)r + 0.5f) * inv_s);
                    const int c = r - row * s;
                    const float v = dVt[r * TS + t];
                    float* dSt = dSg + (size_t)t * s * s + c;
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(&dSt[tab_cy[row * 4 + j] * s], tab_wy[row * 4 + j] * v);
                }
            }
        }
    }
    if (!BWD && valid) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < T) M[((size_t)b * T + t) * RR + p] = acc[t] * a.inv_lh;
    }
}

template <bool BWD>
static int launch_map(MapArgs& a, float* M, float* lse_out, const float* dM, const float* lse_in, hipStream_t st) {
    int smax = 0;
    for (int l = 0; l < a.L; ++l) smax = a.s[l] > smax ? a.s[l] : smax;
    const int R = a.R;
    int th = (256 + R - 1) / R + 1;
    if (th > R) th = R;
    a.th_max = th;
    const int nt = ((a.T + 15) / 16) * 16;
    a.vt_floats = th * smax * (nt + 1);
    const size_t lds = ((size_t)(BWD ? 2 : 1) * a.vt_floats + 8 * (size_t)th) * sizeof(float);
    if (lds > 160 * 1024) return SKP_E_LDS;
    dim3 grid((R * R + 255) / 256, a.B), block(256);
#define SKP_MAP_CASE(NTV)                                                                              \
    case NTV:                                                                                          \
        if (lds > 64 * 1024) {                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)skp_attn_map_kernel<NTV, BWD>,            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            if (e != hipSuccess) return (int)e;                                                        \
        }                                                                                              \
        hipLaunchKernelGGL((skp_attn_map_kernel<NTV, BWD>), grid, block, lds, st, a, M, lse_out, dM, lse_in); \
        break;
    switch (nt) {
        SKP_MAP_CASE(16) SKP_MAP_CASE(32) SKP_MAP_CASE(48) SKP_MAP_CASE(64)
        SKP_MAP_CASE(80) SKP_MAP_CASE(96) SKP_MAP_CASE(112) SKP_MAP_CASE(128)
        default: return SKP_E_RANGE;
    }
#undef SKP_MAP_CASE
    return skp_launch_status();
}

static int fill_args(MapArgs& a, const float* const* S, float* const* dS, const int* s, int L, int B, int H,
                     int T, int R) {
    if (!S || !s || L <= 0 || B <= 0 || H <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (L > SKP_MAX_LAYERS || T > 128 || B > 65535) return SKP_E_RANGE;
    for (int l = 0; l < L; ++l) {
        if (!S[l] || s[l] <= 0) return SKP_E_BADARG;
        if (s[l] > 64) return SKP_E_RANGE;
        a.S[l] = S[l];
        a.dS[l] = dS ? dS[l] : nullptr;
        if (dS && !dS[l]) return SKP_E_BADARG;
        a.s[l] = s[l];
    }
    a.L = L; a.B = B; a.H = H; a.T = T; a.R = R;
    a.inv_lh = 1.0f / (float)(L * H);
    return 0;
}

extern "C" int skp_attn_map_fwd_f32(const float* const* S, const int* s, int L, int B, int H, int T, int R,
                                    float* M, float* lse, void* stream) {
    MapArgs a{};
    int rc = fill_args(a, S, nullptr, s, L, B, H, T, R);
    if (rc) return rc;
    if (!M || !lse) return SKP_E_BADARG;
    return launch_map<false>(a, M, lse, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int skp_attn_map_bwd_f32(const float* const* S, float* const* dS, const int* s, int L, int B, int H,
                                    int T, int R, const float* dM, const float* lse, void* stream) {
    MapArgs a{};
    if (!dS) return SKP_E_BADARG;
    int rc = fill_args(a, S, dS, s, L, B, H, T, R);
    if (rc) return rc;
    if (!dM || !lse) return SKP_E_BADARG;
    return launch_map<true>(a, nullptr, nullptr, dM, lse, (hipStream_t)stream);
}
