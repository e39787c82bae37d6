// 3x3 / stride 2 convolution (forward) as an implicit GEMM on the fp32 matrix cores, NCHW in and out, zero padding and
// bias folded in: the down-sampling convolutions of the frozen VAE encoder (diffusers Downsample2D with padding 0:
// F.pad(x, (0,1,0,1)) then conv(stride 2) [third party], reached from ptp_utils.py:289-304 `image2latent`) and the UNet's
// Downsample2D (padding 1).  On the library path these three VAE layers cost the step 3.7 ms of implicit-GEMM kernels at
// ~0.5 of the fp32 peak plus 1.5 ms of NCHW<->NHWC transposes plus 1 ms of F.pad copies; Winograd does not apply to
// stride 2, but a direct form needs no transform arithmetic at all, so nearly every issued instruction is an MFMA.
//
//   y[b,co,oy,ox] = bias[co] + sum_{ci,a,c} w[co,ci,a,c] * x[b,ci, 2 oy + a - p, 2 ox + c - p]      (out of range = 0)
//   p = 0: the asymmetric (0,1,0,1) padding of the VAE;  p = 1: symmetric padding 1.   OH = H/2, OW = W/2.
//
// GEMM view: M = Cout, N = output pixels, K = (tap, ci).  v_mfma_f32_16x16x4_f32; k-slot kq = lane >> 4 and MFMA step m
// contract input channel 4 kq + m of the current 16-channel stage (the operand order of skp_conv_wino4.hip):
//   A  filter, pre-arranged once per frozen weight as U[tap][ci/16][kq][co][m]  -> one 16-byte load per (tap, 16 channels)
//   B  input patch in LDS, channel-interleaved [kq][row][col][m]                -> one ds_read_b128 per (tap, 16 pixels)
// Workgroup = 8 x 16 output pixels x 128 output channels; wave = 32 channels (2 M tiles) x 8 rows of 16 pixels = 16
// accumulator tiles (64 registers) -> two workgroups per CU.  Per 16-channel stage a wave issues 576 MFMAs for 72 LDS
// reads and 18 filter loads; the next stage's 17 x 33 x 16 input patch is fetched with 48 buffer_load_dword per thread
// (position fixed per thread, channel = scalar offset: no address arithmetic in the loop; out-of-range offsets return 0 =
// the zero padding) while the current stage computes, and goes to the other LDS buffer afterwards (one barrier per stage).
#include "skp_common.h"
#include <stdlib.h>

namespace {

constexpr int S2_TOH = 8, S2_TOW = 16;                    // output tile
constexpr int S2_ROWS = 2 * S2_TOH + 1, S2_COLS = 2 * S2_TOW + 1;   // 17 x 33 input patch
constexpr int S2_POS = S2_ROWS * S2_COLS;                 // 561 positions per channel
constexpr int S2_SLOTS = (S2_POS + 255) / 256;            // 3 positions per thread
constexpr int S2_STAGE = 4 * S2_POS * 4;                  // floats per LDS stage: [kq][pos][m]

__global__ void skp_conv_s2_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    const int c16 = ci >> 4, kq = (ci >> 2) & 3, m = ci & 3, C16 = Cin >> 4;
    const float* p = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) U[((((size_t)tap * C16 + c16) * 4 + kq) * Cout + co) * 4 + m] = p[tap];
}

struct S2Args {
    const float* x; const float* U; const float* bias; float* y;
    int B, Cin, Cout, H, W, OH, OW, pad;
    int tilesX, tilesPerImg;
    unsigned x_bytes, u_bytes, y_bytes;
    float* stats;            // optional [B][Cout][tilesPerImg][2] = {mean, sum (y - mean)^2} per 8x16-pixel tile (next GroupNorm), or null
    // K split (small grids: the UNet's down-sampling layers are 80-192 workgroups at 8 rows): blockIdx.z = split, each takes
    // `per` 16-channel stages and writes its partial sums (no bias) to part + z * y_elems; skp_conv_s2_reduce_kernel adds them
    int nsplit, per;
    float* part;
    unsigned y_elems;
};

__global__ __launch_bounds__(256, 2) void skp_conv_s2_kernel(S2Args a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [2][4 kq][561 pos][4 m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x, cg = blockIdx.y;
    const int b = tile / a.tilesPerImg, rem = tile - b * a.tilesPerImg;
    const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
    const int oy0 = ty * S2_TOH, ox0 = tx * S2_TOW;
    const int HW = a.H * a.W;
    const int co0 = cg * 128 + wave * 32;
    const int C16 = a.Cin >> 4;
    const int s_begin = a.nsplit > 1 ? (int)blockIdx.z * a.per : 0;
    const int s_end = a.nsplit > 1 ? min(C16, s_begin + a.per) : C16;

    // ---- staging role: thread -> up to 3 fixed positions of the 17 x 33 patch; the channel is a scalar offset ----
    int goff[S2_SLOTS], loff[S2_SLOTS];
#pragma unroll
    for (int u = 0; u < S2_SLOTS; ++u) {
        const int pos = tid + 256 * u;
        const int row = pos / S2_COLS, col = pos - row * S2_COLS;
        const int iy = 2 * oy0 + row - a.pad, ix = 2 * ox0 + col - a.pad;
        const bool ok = pos < S2_POS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        goff[u] = ok ? ((b * a.Cin) * HW + iy * a.W + ix) * 4 : SKP_OOB;
        loff[u] = pos < S2_POS ? pos * 4 : -1;
    }
    const i32x4 xrs = skp_make_rsrc(a.x, a.x_bytes);
    const i32x4 urs = skp_make_rsrc(a.U, a.u_bytes);
    float pre[S2_SLOTS][16];
    auto fetch = [&](int s) {
#pragma unroll
        for (int u = 0; u < S2_SLOTS; ++u)
#pragma unroll
            for (int c = 0; c < 16; ++c) pre[u][c] = skp_buf_load_f32(xrs, goff[u], (s * 16 + c) * HW * 4, 0);
    };
    auto put = [&](int buf) {
        float* dst = xs + buf * S2_STAGE;
#pragma unroll
        for (int u = 0; u < S2_SLOTS; ++u)
            if (loff[u] >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4*)(dst + q * (S2_POS * 4) + loff[u]) = f32x4{pre[u][4 * q], pre[u][4 * q + 1], pre[u][4 * q + 2], pre[u][4 * q + 3]};
            }
    };

    f32x4 acc[2][S2_TOH];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < S2_TOH; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // filter operand offsets: U[tap][c16][kq][co][m]
    int uvo[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) uvo[mt] = (kq * a.Cout + min(co0 + 16 * mt + i16, a.Cout - 1)) * 16;
    const int u_c16 = 4 * a.Cout * 16, u_tap = C16 * u_c16;

    fetch(s_begin);
    put(0);
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        if (s + 1 < s_end) fetch(s + 1);
        const float* xb = xs + ((s - s_begin) & 1) * S2_STAGE + kq * (S2_POS * 4);
        f32x4 ua[3][2];                                           // filter ring, two taps ahead
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) ua[t][mt] = skp_buf_load_f32x4(urs, uvo[mt], t * u_tap + s * u_c16, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 2 < 9) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) ua[(tap + 2) % 3][mt] = skp_buf_load_f32x4(urs, uvo[mt], (tap + 2) * u_tap + s * u_c16, 0);
            }
            const int ta = tap / 3, tc = tap - 3 * ta;
            f32x4 bv[S2_TOH];
#pragma unroll
            for (int nt = 0; nt < S2_TOH; ++nt) bv[nt] = *(const f32x4*)(xb + ((2 * nt + ta) * S2_COLS + 2 * i16 + tc) * 4);
#pragma unroll
            for (int m = 0; m < 4; ++m)                         // an accumulator is revisited after 15 other MFMAs
#pragma unroll                                                   // (dependent-issue latency of 16x16x4 is 40 cycles > its 32-cycle issue)
                for (int nt = 0; nt < S2_TOH; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[tap % 3][mt][m], bv[nt][m], acc[mt][nt], 0, 0, 0);
        }
        if (s + 1 < s_end) put((s + 1 - s_begin) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane = pixel (16 consecutive ox of row oy0 + nt), registers = 4 output channels ----
    const bool split = a.nsplit > 1;
    const i32x4 yrs = skp_make_rsrc(split ? a.part + (size_t)blockIdx.z * a.y_elems : a.y, a.y_bytes);
    const i32x4 brs = skp_make_rsrc(a.bias, (a.bias && !split) ? (unsigned)a.Cout * 4u : 0u);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * mt + 4 * kq + r;
            const bool cok = co < a.Cout;
            const float bv = skp_buf_load_f32(brs, cok ? co * 4 : SKP_OOB, 0, 0);
            const int base = ((b * a.Cout + co) * a.OH + oy0) * a.OW + ox0 + i16;
            float vals[S2_TOH];
#pragma unroll
            for (int nt = 0; nt < S2_TOH; ++nt) {
                const float v = acc[mt][nt][r] + bv;
                skp_buf_store_f32(v, yrs, cok ? (base + nt * a.OW) * 4 : SKP_OOB, 0, 0);
                vals[nt] = v;
            }
            if (a.stats) {                                       // uniform branch: {mean, sum (y - mean)^2} of the 8x16-pixel tile
                // lane: about its first value; the 16 pixel lanes merge pairwise with equal counts (Chan et al.)
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int nt = 1; nt < S2_TOH; ++nt) { const float d = vals[nt] - vals[0]; s1 += d; s2 += d * d; }
                const float dm = s1 * (1.0f / S2_TOH);
                float mean = vals[0] + dm, m2 = s2 - s1 * dm;
                float cnt = (float)S2_TOH;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const float om = __shfl_xor(mean, o, 64), o2 = __shfl_xor(m2, o, 64);
                    const float d = om - mean;
                    m2 = m2 + o2 + d * d * (cnt * 0.5f);
                    mean = 0.5f * (mean + om);
                    cnt *= 2.0f;
                }
                if (i16 == 0 && cok) {
                    float* dst = a.stats + (((size_t)b * a.Cout + co) * a.tilesPerImg + rem) * 2;
                    dst[0] = mean; dst[1] = m2;
                }
            }
        }
}

// y = bias[c] + sum over the splits of part[z]   (fixed order; float4 per thread, OH * OW % 4 == 0)
__global__ __launch_bounds__(256) void skp_conv_s2_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                 float* __restrict__ y, unsigned n4, unsigned stride, int nsplit,
                                                                 int hw4, int Cout) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = ((const f32x4*)part)[i];
    for (int z = 1; z < nsplit; ++z) acc += ((const f32x4*)(part + (size_t)z * stride))[i];
    if (bias) acc += bias[(i / hw4) % Cout];
    ((f32x4*)y)[i] = acc;
}

// K splits of a launch: enough workgroups for two per CU, at least four 16-channel stages each, at most eight
static int s2_splits(int B, int Cin, int Cout, int H, int W) {
    const long wgs = (long)B * ((H / 2) / S2_TOH) * ((W / 2) / S2_TOW) * ((Cout + 127) / 128);
    if (wgs >= 256) return 1;
    long s = (512 + wgs - 1) / wgs;
    const long cap = (Cin >> 4) / 4;
    if (s > cap) s = cap;
    if (s > 8) s = 8;
    return s < 2 ? 1 : (int)s;
}

}  // namespace

extern "C" int skp_conv3x3_s2_filter_f32(const void* w, void* U, int Cout, int Cin, void* stream) {
    if (!w || !U || Cout <= 0 || Cin <= 0) return SKP_E_BADARG;
    if (Cin & 15) return SKP_E_RANGE;
    const int n = Cout * Cin;
    hipLaunchKernelGGL(skp_conv_s2_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                       (float*)U, Cout, Cin);
    return skp_launch_status();
}

static int s2_run(const void* x, const void* U, const void* bias, void* y, float* stats, float* workspace, int B, int Cin, int Cout,
                  int H, int W, int pad, void* stream);

extern "C" int skp_conv3x3_s2_f32(const void* x, const void* U, const void* bias, void* y, int B, int Cin, int Cout, int H,
                                  int W, int pad, void* stream) {
    return s2_run(x, U, bias, y, nullptr, nullptr, B, Cin, Cout, H, W, pad, stream);
}

// Bytes of scratch skp_conv3x3_s2_ws_f32 wants for this shape: 0 = the launch fills the chip on its own; else the partial sums
// of a K split (small grids: 640 -> 640 @32^2, 8 rows is 80 workgroups of 128 channels x 128 pixels on 256 CUs).
extern "C" int64_t skp_conv3x3_s2_workspace(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (Cin & 15) || (H & 1) || (W & 1)) return 0;
    if (((H / 2) % S2_TOH) || ((W / 2) % S2_TOW)) return 0;
    const int s = s2_splits(B, Cin, Cout, H, W);
    return s > 1 ? (int64_t)s * B * Cout * (H / 2) * (W / 2) * 4 : 0;
}

// skp_conv3x3_s2_f32 with that scratch (NULL where skp_conv3x3_s2_workspace is 0): same result up to the summation order
extern "C" int skp_conv3x3_s2_ws_f32(const void* x, const void* U, const void* bias, void* y, void* workspace, int B, int Cin,
                                     int Cout, int H, int W, int pad, void* stream) {
    return s2_run(x, U, bias, y, nullptr, (float*)workspace, B, Cin, Cout, H, W, pad, stream);
}

// as above + stats [B][Cout][(H/16)*(W/32)][2]: {mean, sum of squared deviations} of y over each 8x16-pixel output tile
extern "C" int skp_conv3x3_s2_stats_f32(const void* x, const void* U, const void* bias, void* y, float* stats, int B, int Cin,
                                        int Cout, int H, int W, int pad, void* stream) {
    if (!stats) return SKP_E_BADARG;
    return s2_run(x, U, bias, y, stats, nullptr, B, Cin, Cout, H, W, pad, stream);
}

static int s2_run(const void* x, const void* U, const void* bias, void* y, float* stats, float* workspace, int B, int Cin, int Cout,
                  int H, int W, int pad, void* stream) {
    if (!x || !U || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (pad != 0 && pad != 1)) return SKP_E_BADARG;
    if ((Cin & 15) || (Cout & 31) || (H & 1) || (W & 1)) return SKP_E_RANGE;
    const int OH = H / 2, OW = W / 2;
    if ((OH % S2_TOH) || (OW % S2_TOW)) return SKP_E_RANGE;
    const unsigned long long xb = (unsigned long long)B * Cin * H * W * 4, ub = (unsigned long long)9 * Cin * Cout * 4,
                             yb = (unsigned long long)B * Cout * OH * OW * 4;
    if (xb >= 0x80000000ull || ub >= 0x80000000ull || yb >= 0x80000000ull) return SKP_E_RANGE;
    S2Args a;
    a.x = (const float*)x; a.U = (const float*)U; a.bias = (const float*)bias; a.y = (float*)y;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.pad = pad;
    a.tilesX = OW / S2_TOW;
    a.tilesPerImg = a.tilesX * (OH / S2_TOH);
    a.x_bytes = (unsigned)xb; a.u_bytes = (unsigned)ub; a.y_bytes = (unsigned)yb;
    a.stats = stats;
    a.nsplit = (workspace && !stats) ? s2_splits(B, Cin, Cout, H, W) : 1;
    a.per = ((Cin >> 4) + a.nsplit - 1) / a.nsplit;
    a.part = workspace;
    a.y_elems = (unsigned)(yb / 4);
    const long tiles = (long)B * a.tilesPerImg;
    if (tiles > 0x7fffffffL) return SKP_E_RANGE;
    const size_t lds = (size_t)2 * S2_STAGE * sizeof(float);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)skp_conv_s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(skp_conv_s2_kernel, dim3((unsigned)tiles, (Cout + 127) / 128, a.nsplit), dim3(256), lds, (hipStream_t)stream, a);
    int rc = skp_launch_status();
    if (rc || a.nsplit == 1) return rc;
    const unsigned n4 = a.y_elems / 4;
    hipLaunchKernelGGL(skp_conv_s2_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                       (const float*)bias, (float*)y, n4, a.y_elems, a.nsplit, OH * OW / 4, Cout);
    return skp_launch_status();
}
