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

template <int CI>
__global__ __launch_bounds__(256) void skp_conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int Co, int H,
                                                          int W, int co_per) {
    const int W2 = W >> 1;
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= H * W2) return;
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
    }
}

}  // namespace

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
    hipLaunchKernelGGL(skp_conv_in_kernel<CI>, grid, block, 0, st, (const float*)x, (const float*)w, (const float*)bias, (float*)y, Cout, H, W, co_per)
    switch (Cin) {
        case 1: SKP_CIN(1); break;
        case 2: SKP_CIN(2); break;
        case 3: SKP_CIN(3); break;
        default: SKP_CIN(4); break;
    }
#undef SKP_CIN
    return skp_launch_status();
}
