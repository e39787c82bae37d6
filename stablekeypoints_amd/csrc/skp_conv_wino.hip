// 3x3 / stride 1 / pad 1 convolution of the frozen UNet + VAE blocks as Winograd F(2x2,3x3) on the fp32
// matrix cores.  (The reference runs these through nn.Conv2d of diffusers' ResnetBlock2D / Upsample2D /
// AutoencoderKL encoder [third party]; they are ~64 % of the optimisation step.)
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, summed over input channels
//
// The 16 element-wise products over channels are 16 independent GEMMs  M_p[co][tile] = U_p[co][ci] V_p[ci][tile],
// i.e. 2.25x fewer multiplies than the direct form, all of them on v_mfma_f32_32x32x2_f32.
//
// Work split: a wave owns 32 output channels x 32 tiles x all 16 Winograd positions = 16 accumulator tiles
// (256 accumulator registers: one wave per SIMD, 512-register budget), so the output transform A^T M A is
// purely in-lane and nothing but the final activations goes back to HBM.  The workgroup (4 waves = CB
// channel blocks x TB tile blocks) transforms its input patches once per channel chunk into LDS
// (double buffered, 2 x 64 KB), already in MFMA operand order; the filter side U is transformed once per
// layer (weights are frozen) into the same operand order and streamed from L2 with 512-byte coalesced
// b128 loads.  Tiles sit on the lane axis, so the 2x2 outputs leave as coalesced float2 rows.
#include <type_traits>
#include "skp_common.h"

namespace {

// ---- filter transform: U'[p][ci/8][(ci%8)/4][co][ci%4] = (G g G^T)[p]  ------------------------------------
// flip_t = 0: g = w[co][ci]            (forward)
// flip_t = 1: g = rot180(w[ci][co])    (backward-data: a convolution with Cin and Cout swapped)
__global__ void skp_wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int flip_t) {
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
        const float* p = w + ((size_t)ci * Cout + co) * 9;      // w is [Cin_of_this_conv = original Cout][...]
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = p[(2 - a) * 3 + (2 - b)];
    }
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int c8 = ci >> 3, kh = (ci >> 2) & 1, m = ci & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float u[4];
        u[0] = t[i][0];
        u[1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        u[2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        u[3] = t[i][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = i * 4 + j;
            U[((((size_t)p * (Cin >> 3) + c8) * 2 + kh) * Cout + co) * 4 + m] = u[j];
        }
    }
}

struct WinoArgs {
    const float* x;
    const float* U;
    const float* bias;      // may be null
    const float* res;       // residual added to the output (same shape), may be null
    float* y;
    int B, Cin, Cout, H, W;
    int tilesX, tilesPerImg, nTiles;
    unsigned x_bytes, u_bytes, y_bytes;
    int steps;              // channel stages per workgroup (= Cin / KC / splits)
    size_t y_split_stride;  // elements between the partial outputs of consecutive splits (blockIdx.z)
};

template <int CB, int TB>
struct WinoShape {
    static constexpr int NTILE = 32 * TB;           // tiles per workgroup
    static constexpr int KC = 1024 / NTILE;         // input channels per stage (one patch quad per thread)
    static constexpr int NC8 = KC / 8;
    static constexpr int STAGE_F4 = 16 * NC8 * 2 * NTILE;   // float4 per stage (= 4096 -> 64 KB)
};

// one thread: the 4x4 patches of 4 consecutive channels of one tile -> B^T d B -> 16 float4 into LDS
struct PatchIdx {
    int off[4][4];          // byte offset of (row i, col j) of channel quad 0 of this tile's image, or SKP_OOB
};

template <int CB, int TB, bool VEC>     // VEC: W even -> 2x2 outputs / residuals move as aligned float2 rows
__global__ __launch_bounds__(256, 1) void skp_wino_conv_kernel(WinoArgs a) {
    using S = WinoShape<CB, TB>;
    extern __shared__ f32x4 vst[];                   // [2][16][NC8][2][NTILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int cbi = wave % CB, tbi = wave / CB;
    const int tile0 = blockIdx.x * S::NTILE;
    const int n0 = (blockIdx.y * CB + cbi) * 32;
    const bool wave_active = n0 < a.Cout;
    const int HW = a.H * a.W;

    // ---- this thread's patch (transform role) ----
    const int tl = tid % S::NTILE, qd = tid / S::NTILE;         // tile in WG, channel quad in stage
    PatchIdx pi;
    {
        const int tg = tile0 + tl;
        const bool tv = tg < a.nTiles;
        const int tgc = tv ? tg : 0;
        const int b = tgc / a.tilesPerImg, rem = tgc - b * a.tilesPerImg;
        const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        const int base = (b * a.Cin + 4 * qd) * HW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 2 * ty - 1 + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 2 * tx - 1 + j;
                const bool ok = tv && r >= 0 && r < a.H && c >= 0 && c < a.W;
                pi.off[i][j] = ok ? (base + r * a.W + c) * 4 : SKP_OOB;
            }
        }
    }
    const i32x4 xrs = skp_make_rsrc(a.x, a.x_bytes);
    const i32x4 urs = skp_make_rsrc(a.U, a.u_bytes);
    float d[4][4][4];                                           // [m][row][col]
    auto load_patch = [&](int cin0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int so = (cin0 + m) * HW * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[m][i][j] = skp_buf_load_f32(xrs, pi.off[i][j], so, 0);
        }
    };
    auto transform_store = [&](int buf) {
        f32x4* dst = vst + buf * S::STAGE_F4 + (size_t)qd * S::NTILE + tl;   // (c8*2+kh) == qd
        float v[4][16];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[m][0][j] - d[m][2][j];
                t[1][j] = d[m][1][j] + d[m][2][j];
                t[2][j] = d[m][2][j] - d[m][1][j];
                t[3][j] = d[m][1][j] - d[m][3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[m][i * 4 + 0] = t[i][0] - t[i][2];
                v[m][i * 4 + 1] = t[i][1] + t[i][2];
                v[m][i * 4 + 2] = t[i][2] - t[i][1];
                v[m][i * 4 + 3] = t[i][1] - t[i][3];
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            f32x4 o = {v[0][p], v[1][p], v[2][p], v[3][p]};
            dst[(size_t)p * (S::NC8 * 2 * S::NTILE)] = o;
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int nsteps = a.steps;
    const int cin_begin = blockIdx.z * nsteps * S::KC;           // split-K over input channels
    const int C8 = a.Cin >> 3;
    const int co_l = min(n0 + li, a.Cout - 1);
    // U' as float4: index = ((p*C8 + c8)*2 + kh)*Cout + co ; per-lane part in uvo, the rest is wave-uniform
    const int uvo = (hi * a.Cout + co_l) * 16;
    const int u_c8 = 2 * a.Cout * 16, u_p = C8 * u_c8;

    // ---- output role of this lane (tile = lane, 16 output channels in registers), needed early for the residual ----
    const int tgo = tile0 + tbi * 32 + li;
    const bool out_ok = wave_active && tgo < a.nTiles;
    const int tgoc = out_ok ? tgo : 0;
    const int ob = tgoc / a.tilesPerImg, orem = tgoc - ob * a.tilesPerImg;
    const int oty = orem / a.tilesX, otx = orem - oty * a.tilesX;
    const int oy = 2 * oty, ox = 2 * otx;
    const bool row1 = oy + 1 < a.H, col1 = ox + 1 < a.W;
    const int o_base = (ob * a.Cout * a.H + oy) * a.W + ox;          // + co*H*W
    // residual and bias ride in registers loaded under the last stage's MFMAs; without them every load is out of range -> 0
    const i32x4 rrs = skp_make_rsrc(a.res, a.res ? a.y_bytes : 0u);
    const i32x4 brs = skp_make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);

    load_patch(cin_begin);
    transform_store(0);
    // U operands travel through a ring of UD+1 positions: the position needed UD products from now is requested
    // first, then this position's share of the next stage's patch loads.  VMEM returns in order, so everything
    // queued behind a patch load (HBM latency) must not be needed before UD x 1024 MFMA cycles have passed.
    constexpr int UD = 3;
    f32x4 ua[UD + 1][S::NC8];
#pragma unroll
    for (int q = 0; q < UD; ++q)
#pragma unroll
        for (int c = 0; c < S::NC8; ++c) ua[q][c] = skp_buf_load_f32x4(urs, uvo, (cin_begin >> 3) * u_c8 + q * u_p + c * u_c8, 0);
    __syncthreads();
    // One channel stage.  MODE 0: a further stage follows (its patch loads ride along); 1: last stage; 2: last stage
    // with the residual's 2x2 values riding in the (now free) patch registers.  Three straight-line copies of the
    // position loop instead of wave-uniform branches inside it (and no branch around the accumulators either).
    auto run_stage = [&](int s, auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        // inactive waves (n0 >= Cout) run the same products on clamped rows: no divergent region around the accumulators
        const f32x4* vb = vst + (s & 1) * S::STAGE_F4 + (size_t)hi * S::NTILE + tbi * 32 + li;
        const int ub = ((cin_begin >> 3) + s * S::NC8) * u_c8;
        f32x4 va[2][S::NC8];
#pragma unroll
        for (int c = 0; c < S::NC8; ++c) va[0][c] = vb[(size_t)c * (2 * S::NTILE)];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            if (MODE == 0 || p + UD < 16) {   // U for position p+UD (wrapping into the next stage)
                const int q = p + UD;
                const int uo = q < 16 ? ub + q * u_p : ub + S::NC8 * u_c8 + (q - 16) * u_p;
#pragma unroll
                for (int c = 0; c < S::NC8; ++c) ua[q % (UD + 1)][c] = skp_buf_load_f32x4(urs, uvo, uo + c * u_c8, 0);
            }
            if (MODE == 0) {                  // this position's quarter-channel of the next stage's patches
                const int m = p >> 2, i = p & 3;
                const int so = (cin_begin + (s + 1) * S::KC + m) * HW * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) d[m][i][j] = skp_buf_load_f32(xrs, pi.off[i][j], so, 0);
            } else if (MODE == 2) {           // the 2x2 residual values of output-channel register p
                const int co = n0 + (p & 3) + 8 * (p >> 2) + 4 * hi;
                const bool ok = out_ok && co < a.Cout;
                const int vo = (o_base + co * HW) * 4;
                if (VEC) {
                    const f32x2 r0 = skp_buf_load_f32x2(rrs, ok ? vo : SKP_OOB, 0, 0);
                    const f32x2 r1 = skp_buf_load_f32x2(rrs, ok && row1 ? vo + a.W * 4 : SKP_OOB, 0, 0);
                    d[p >> 2][p & 3][0] = r0[0]; d[p >> 2][p & 3][1] = r0[1];
                    d[p >> 2][p & 3][2] = r1[0]; d[p >> 2][p & 3][3] = r1[1];
                } else {
                    d[p >> 2][p & 3][0] = skp_buf_load_f32(rrs, ok ? vo : SKP_OOB, 0, 0);
                    d[p >> 2][p & 3][1] = skp_buf_load_f32(rrs, ok && col1 ? vo + 4 : SKP_OOB, 0, 0);
                    d[p >> 2][p & 3][2] = skp_buf_load_f32(rrs, ok && row1 ? vo + a.W * 4 : SKP_OOB, 0, 0);
                    d[p >> 2][p & 3][3] = skp_buf_load_f32(rrs, ok && row1 && col1 ? vo + a.W * 4 + 4 : SKP_OOB, 0, 0);
                }
            }
            if (p + 1 < 16) {
#pragma unroll
                for (int c = 0; c < S::NC8; ++c) va[(p + 1) & 1][c] = vb[(size_t)((p + 1) * S::NC8 + c) * (2 * S::NTILE)];
            }
#pragma unroll
            for (int c = 0; c < S::NC8; ++c)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[p % (UD + 1)][c][m], va[p & 1][c][m], acc[p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s = 0; s + 1 < nsteps; ++s) {
        run_stage(s, std::integral_constant<int, 0>{});
        transform_store((s + 1) & 1);
        __syncthreads();
    }
    float bvs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = n0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        bvs[r] = skp_buf_load_f32(brs, co < a.Cout ? co * 4 : SKP_OOB, 0, 0);
    }
    run_stage(nsteps - 1, std::integral_constant<int, 2>{});       // without a residual its loads are out of range -> 0

    // ---- output transform (in-lane) + store: lane = tile, register r = output channel.  No branches: invalid
    //      lanes / channels / rows store to an out-of-range buffer offset, which the hardware drops. ----
    const i32x4 yrs = skp_make_rsrc(a.y + blockIdx.z * a.y_split_stride, a.y_bytes);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = n0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = out_ok && co < a.Cout;
        float sr[4], dr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sr[i] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
            dr[i] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
        }
        const float bv = bvs[r];
        const float y00 = sr[0] + sr[1] + sr[2] + bv + d[r >> 2][r & 3][0], y01 = dr[0] + dr[1] + dr[2] + bv + d[r >> 2][r & 3][1];
        const float y10 = sr[1] - sr[2] - sr[3] + bv + d[r >> 2][r & 3][2], y11 = dr[1] - dr[2] - dr[3] + bv + d[r >> 2][r & 3][3];
        const int vo = (o_base + co * HW) * 4;
        if (VEC) {
            skp_buf_store_f32x2(f32x2{y00, y01}, yrs, ok ? vo : SKP_OOB, 0, 0);
            skp_buf_store_f32x2(f32x2{y10, y11}, yrs, ok && row1 ? vo + a.W * 4 : SKP_OOB, 0, 0);
        } else {
            skp_buf_store_f32(y00, yrs, ok ? vo : SKP_OOB, 0, 0);
            skp_buf_store_f32(y01, yrs, ok && col1 ? vo + 4 : SKP_OOB, 0, 0);
            skp_buf_store_f32(y10, yrs, ok && row1 ? vo + a.W * 4 : SKP_OOB, 0, 0);
            skp_buf_store_f32(y11, yrs, ok && row1 && col1 ? vo + a.W * 4 + 4 : SKP_OOB, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);       // one output channel at a time: keeps the accumulator reads from piling up in VGPRs
    }
}

// y[i] = sum_z part[z][i] (+ bias[channel]) in fixed order: the deterministic tail of the split-K launches
__global__ void skp_wino_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                       const float* __restrict__ res, float* __restrict__ y, size_t n4, size_t stride,
                                       int splits, int HW4, int Cout) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = ((const f32x4*)part)[i];
    for (int z = 1; z < splits; ++z) acc += ((const f32x4*)(part + z * stride))[i];
    if (bias) {
        const float bv = bias[(i / HW4) % Cout];
        acc += f32x4{bv, bv, bv, bv};
    }
    if (res) acc += ((const f32x4*)res)[i];
    ((f32x4*)y)[i] = acc;
}

template <int CB, int TB, bool VEC>
int launch_wino_v(const WinoArgs& a, int splits, hipStream_t st) {
    using S = WinoShape<CB, TB>;
    const size_t lds = (size_t)2 * S::STAGE_F4 * sizeof(f32x4);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_wino_conv_kernel<CB, TB, VEC>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((a.nTiles + S::NTILE - 1) / S::NTILE, (a.Cout + 32 * CB - 1) / (32 * CB), splits);
    hipLaunchKernelGGL((skp_wino_conv_kernel<CB, TB, VEC>), grid, dim3(256), lds, st, a);
    return skp_launch_status();
}

}  // namespace

extern "C" int skp_conv3x3_filter_f32(const void* w, void* U, int Cout, int Cin, int flip_transpose, void* stream) {
    if (!w || !U || Cout <= 0 || Cin <= 0) return SKP_E_BADARG;
    if ((Cin & 7) != 0) return SKP_E_RANGE;
    const int n = Cout * Cin;
    hipLaunchKernelGGL(skp_wino_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float*)w, (float*)U, Cout, Cin, flip_transpose);
    return skp_launch_status();
}

template <int CB, int TB>
int launch_wino(const WinoArgs& a, int splits, hipStream_t st) {
    return (a.W & 1) ? launch_wino_v<CB, TB, false>(a, splits, st) : launch_wino_v<CB, TB, true>(a, splits, st);
}

// Split-K choice: fill the 256 CUs (one workgroup each: 512 registers per lane, 128 KB LDS).  Cost model in
// microseconds: rounds x (stages + ~1 stage of prologue/epilogue) x stage time, plus the reduce pass.
static int wino_pick_split(int wgs, int nsteps, double stage_us, double out_bytes) {
    int best = 1;
    double best_cost = 1e30;
    for (int S = 1; S <= 16; ++S) {
        if (nsteps % S) continue;
        const int rounds = (wgs * S + 255) / 256;
        double cost = rounds * (nsteps / S + 1.0) * stage_us;
        if (S > 1) cost += 6.0 + (S + 1) * out_bytes / 4.0e6;
        if (cost < best_cost * (S > 1 ? 0.92 : 1.0)) { best_cost = cost; best = S; }
    }
    return best;
}

static int wino_plan(int B, int Cin, int Cout, int H, int W, int variant, int* v_out) {
    if (variant == 0) variant = (Cin % 32 == 0) ? 1 : 2;   // 128-channel workgroups; a partial last group idles whole waves
    const int tiles = B * ((W + 1) / 2) * ((H + 1) / 2);
    const int ntile = variant == 2 ? 64 : 32, cg = variant == 2 ? 64 : 128, kc = variant == 2 ? 16 : 32;
    *v_out = variant;
    if (Cin % kc) return 0;
    const int wgs = ((tiles + ntile - 1) / ntile) * ((Cout + cg - 1) / cg);
    if ((H * W) % 4) return 1;                              // the reduce pass works on float4 within a channel plane
    return wino_pick_split(wgs, Cin / kc, variant == 2 ? 3.4 : 6.8, (double)B * Cout * H * W * 4);
}

extern "C" int64_t skp_conv3x3_workspace(int B, int Cin, int Cout, int H, int W, int variant) {
    int v;
    const int S = wino_plan(B, Cin, Cout, H, W, variant, &v);
    return S > 1 ? (int64_t)S * B * Cout * H * W * (int64_t)sizeof(float) : 0;
}

extern "C" int skp_conv3x3_f32(const void* x, const void* U, const void* bias, const void* residual, void* y, void* workspace,
                               int B, int Cin, int Cout, int H, int W, int variant, void* stream) {
    if (!x || !U || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return SKP_E_BADARG;
    if ((Cout & 31) != 0) return SKP_E_RANGE;
    int v;
    int S = wino_plan(B, Cin, Cout, H, W, variant, &v);
    if (S == 0) return SKP_E_RANGE;
    if (!workspace) S = 1;
    WinoArgs a;
    a.x = (const float*)x; a.U = (const float*)U;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.tilesX = (W + 1) / 2;
    a.tilesPerImg = a.tilesX * ((H + 1) / 2);
    a.nTiles = B * a.tilesPerImg;
    const unsigned long long xb = (unsigned long long)B * Cin * H * W * 4, ub = (unsigned long long)16 * Cin * Cout * 4;
    if (xb >= 0x80000000ull || ub >= 0x80000000ull) return SKP_E_RANGE;     // 32-bit buffer offsets
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub;
    const unsigned long long yb = (unsigned long long)B * Cout * H * W * 4;
    if (yb >= 0x80000000ull) return SKP_E_RANGE;
    a.y_bytes = (unsigned)yb;
    const size_t out_elems = (size_t)B * Cout * H * W;
    a.steps = Cin / (v == 2 ? 16 : 32) / S;
    a.y_split_stride = out_elems;
    a.y = S > 1 ? (float*)workspace : (float*)y;
    a.bias = S > 1 ? nullptr : (const float*)bias;
    a.res = S > 1 ? nullptr : (const float*)residual;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (v == 1) rc = launch_wino<4, 1>(a, S, st);
    else rc = launch_wino<2, 2>(a, S, st);
    if (rc || S == 1) return rc;
    if ((H * W) % 4) return SKP_E_RANGE;                    // (wino_plan only splits when the reduce pass applies)
    const size_t n4 = out_elems / 4;
    hipLaunchKernelGGL(skp_wino_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)workspace,
                       (const float*)bias, (const float*)residual, (float*)y, n4, out_elems, S, (H * W) / 4, Cout);
    return skp_launch_status();
}
