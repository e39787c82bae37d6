// 3x3 / stride 1 / padding 1 convolution for layers with at most four input channels: the `conv_in` of the frozen VAE
// encoder (3 -> 128 at the image resolution) and of the UNet (4 -> 320 on the latents), reached from
// ptp_utils.py:289-304 (`image2latent`) and ptp_utils.py:227 (the UNet call).  With K = 9 * Cin <= 36 there is nothing for
// the matrix cores to amortise: the layer is bound by writing its output (1.07 GB at 512^2 x 128 channels x 8 rows), so
// this is a plain VALU kernel shaped for that write:
//   thread = two horizontally adjacent output pixels, 3 x 4 x Cin input patch in registers (zero padding by predication);
//   loop over output channels: the 9 * Cin weights of a channel are wave-uniform (scalar loads, one SGPR operand per
//   packed fma), two accumulators per lane, one 8-byte store per channel -> every wave writes 512 contiguous bytes.
// NCHW in and out, bias folded in.
#include "skp_common.h"
#include <algorithm>

namespace {

constexpr int CIN_STATS_MAX_CO = 256;     // output channels a workgroup can keep block statistics for (LDS: 4 waves x 8 bytes each)

// sum over the 64 lanes of a wave, result in lane 63 (DPP row shifts + row broadcasts: no LDS, no cross-lane permutes)
__device__ __forceinline__ float cin_wave_sum(float v) {
#define SKP_DPP_ADD(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))
    SKP_DPP_ADD(0x111, 0xf);   // row_shr:1
    SKP_DPP_ADD(0x112, 0xf);   // row_shr:2
    SKP_DPP_ADD(0x114, 0xf);   // row_shr:4  (lanes 7 / 15 of a row now hold 8-lane sums ... after the next step lane 15 holds the row)
    SKP_DPP_ADD(0x118, 0xf);   // row_shr:8
    SKP_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    SKP_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef SKP_DPP_ADD
    return v;
}

// STATS: the kernel also leaves {mean, sum (y - mean)^2} of every (image, channel, workgroup's 512 pixels) block behind for the
// GroupNorm that follows (the VAE's first norm otherwise re-reads the 1 GB activation for its statistics: 0.2 ms per step).
// Sums are taken about the channel's bias (the output IS bias + a zero-mean-ish filter response: no cancellation).
template <int CI, bool STATS = false>
__global__ __launch_bounds__(256) void skp_conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int Co, int H,
                                                          int W, int co_per, float* __restrict__ stats = nullptr) {
    __shared__ f32x2 part[STATS ? CIN_STATS_MAX_CO * 4 : 1];
    const int W2 = W >> 1;
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (!STATS && pi >= H * W2) return;                         // (STATS launches cover the image exactly: H * W % 512 == 0)
    const int b = blockIdx.y;
    const int yy = pi / W2, x0 = 2 * (pi - yy * W2);
    const size_t plane = (size_t)H * W;
    const float* xb = x + (size_t)b * CI * plane;
    f32x2 p[CI][3][3];                                          // {col j, col j + 1} pairs for the three taps of a row
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = yy + r - 1;
            const bool rin = iy >= 0 && iy < H;
            const float* row = xb + (size_t)ci * plane + (size_t)(rin ? iy : 0) * W;
            const f32x2 mid = rin ? *(const f32x2*)(row + x0) : f32x2{0.f, 0.f};
            const float lf = (rin && x0 > 0) ? row[x0 - 1] : 0.f;
            const float rt = (rin && x0 + 2 < W) ? row[x0 + 2] : 0.f;
            p[ci][r][0] = f32x2{lf, mid[0]};
            p[ci][r][1] = mid;
            p[ci][r][2] = f32x2{mid[1], rt};
        }
    float* yb = y + (size_t)b * Co * plane + (size_t)yy * W + x0;
    // blockIdx.z = slice of `co_per` output channels (small images: the pixel grid alone leaves most CUs idle and a thread would
    // walk all Cout channels serially -- the UNet's conv_in at 64^2 was 64 workgroups x 320 channels: 262 us for a 42 MB write)
    const int co_begin = blockIdx.z * co_per, co_end = min(Co, co_begin + co_per);
#pragma unroll 2
    for (int co = co_begin; co < co_end; ++co) {
        const float* wc = w + (size_t)co * (CI * 9);            // uniform: scalar loads
        const float bv = bias ? bias[co] : 0.f;
        f32x2 acc = {bv, bv};
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float wv = wc[(ci * 3 + r) * 3 + c];
                    acc = f32x2{wv, wv} * p[ci][r][c] + acc;
                }
        *(f32x2*)(yb + (size_t)co * plane) = acc;
        if (STATS) {
            const float d0 = acc[0] - bv, d1 = acc[1] - bv;
            const float s1 = cin_wave_sum(d0 + d1), s2 = cin_wave_sum(d0 * d0 + d1 * d1);
            if ((threadIdx.x & 63) == 63) part[co * 4 + (threadIdx.x >> 6)] = f32x2{s1, s2};
        }
    }
    if (STATS) {
        __syncthreads();
        const int co = threadIdx.x;
        if (co < Co) {
            const f32x2 a0 = part[co * 4], a1 = part[co * 4 + 1], a2 = part[co * 4 + 2], a3 = part[co * 4 + 3];
            const float s1 = (a0[0] + a1[0]) + (a2[0] + a3[0]), s2 = (a0[1] + a1[1]) + (a2[1] + a3[1]);
            const float bv = bias ? bias[co] : 0.f, dm = s1 * (1.0f / 512.0f);
            *(f32x2*)(stats + (((size_t)b * Co + co) * gridDim.x + blockIdx.x) * 2) = f32x2{bv + dm, s2 - s1 * dm};
        }
    }
}

}  // namespace

// 512-pixel blocks per image when skp_conv3x3_small_stats_f32 serves the launch, else 0
extern "C" int skp_conv3x3_small_stats_blocks(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin != 3 || Cout <= 0 || Cout > CIN_STATS_MAX_CO || H <= 0 || W <= 0 || (W & 1)) return 0;
    const long hw = (long)H * W;
    if (hw % 512 || (hw / 512) * B < 1024) return 0;            // whole workgroups only, and a grid that needs no channel slices
    return (int)(hw / 512);
}

// y as skp_conv3x3_small_f32, plus stats [B][Cout][blocks][2] = {mean, sum (y - mean)^2} per 512-pixel block (row-major pixel order)
extern "C" int skp_conv3x3_small_stats_f32(const void* x, const void* w, const void* bias, void* y, float* stats, int B, int Cin,
                                           int Cout, int H, int W, void* stream) {
    if (!x || !w || !y || !stats) return SKP_E_BADARG;
    const int nblk = skp_conv3x3_small_stats_blocks(B, Cin, Cout, H, W);
    if (!nblk || B > 65535) return SKP_E_RANGE;
    hipLaunchKernelGGL((skp_conv_in_kernel<3, true>), dim3((unsigned)nblk, B, 1), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)w, (const float*)bias, (float*)y, Cout, H, W, Cout, stats);
    return skp_launch_status();
}

extern "C" int skp_conv3x3_small_f32(const void* x, const void* w, const void* bias, void* y, int B, int Cin, int Cout, int H,
                                     int W, void* stream) {
    if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return SKP_E_BADARG;
    if (Cin > 4 || (W & 1) || B > 65535 || (long)H * W / 2 > (1L << 30)) return SKP_E_RANGE;
    const long wgs = (((long)H * (W / 2) + 255) / 256) * B;
    int nz = 1;                                               // channel slices: >= ~1024 workgroups, >= 16 channels each
    if (wgs < 1024) nz = (int)std::min<long>((Cout + 15) / 16, (1024 + wgs - 1) / wgs);
    const int co_per = (Cout + nz - 1) / nz;
    nz = (Cout + co_per - 1) / co_per;
    dim3 grid((unsigned)(((long)H * (W / 2) + 255) / 256), B, nz), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SKP_CIN(CI) \
    hipLaunchKernelGGL(skp_conv_in_kernel<CI>, grid, block, 0, st, (const float*)x, (const float*)w, (const float*)bias, (float*)y, Cout, H, W, co_per, (float*)nullptr)
    switch (Cin) {
        case 1: SKP_CIN(1); break;
        case 2: SKP_CIN(2); break;
        case 3: SKP_CIN(3); break;
        default: SKP_CIN(4); break;
    }
#undef SKP_CIN
    return skp_launch_status();
}
