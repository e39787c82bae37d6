// Pieces shared by the Winograd F(4x4,3x3) convolution kernels (skp_conv_wino4.hip: fp32 matrix instructions;
// skp_conv_wino4s.hip: bf16 matrix instructions on three-term operand splits): launch arguments, the workgroup-id ->
// work-unit order, the 1-D transforms, the epilogue statistics and the K-split reduction.
#pragma once
#include <type_traits>
#include <utility>
#include "skp_common.h"

namespace {

struct Wino4Args {
    const float* x;
    const float* U;
    const float* bias;      // may be null
    const float* res;       // may be null
    float* y;
    int B, Cin, Cout, H, W;
    int tilesX, tilesPerImg, nTiles;
    unsigned x_bytes, u_bytes, y_bytes;
    int steps;              // 16-channel stages per workgroup (the last K split may hold fewer: total_steps - z * steps)
    int total_steps;        // Cin / 16
    int ntb, ncg, tb_per_xcd;   // tile blocks, channel groups, tile blocks per XCD band (0: unit-grouped order)
    int splits;                 // K splits
    int gx, vtotal;             // workgroup ids of one K split (band order) and in total: the persistent form walks id, id + grid, ...
    size_t y_split_stride;
    float* stats;               // optional [B][Cout][tilesPerImg/16][2] = {mean, sum (y - mean)^2} per 16-tile block (next GroupNorm), or null
    int sblk;                   // 16-tile blocks per image
    const float* gncoef;        // GNF kernels: [B][Cin][2] = (scale, shift) of the GroupNorm(+offset) in front of this convolution
    int vpad;                   // raw-filter form: tiles per row of the pre-transformed input (tile blocks x 32)
};

// Block statistics without register pressure: every lane parks {mean, M2 = sum (y - mean)^2} of its 4x4 outputs -- taken
// about the lane's first value, so a channel whose mean dwarfs its spread loses nothing to cancellation -- in the (idle) LDS
// stage buffers, [slot][64 lanes] float2; after the epilogue 128 threads merge the 16 tile lanes of one (channel, block) each
// in fixed order (equal counts: mean = average of means, M2 = sum M2_i + 16 sum (mean_i - mean)^2) and store {mean, M2}.
__device__ __forceinline__ void w4_park_stats(f32x2* sbuf, int slot, int lane, const f32x4 (&o)[4], bool ok) {
    const float K = o[0][0];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int oy = 0; oy < 4; ++oy)
#pragma unroll
        for (int ox = 0; ox < 4; ++ox) { const float d = o[oy][ox] - K; s1 += d; s2 += d * d; }
    const float dm = s1 * (1.0f / 16.0f);
    sbuf[slot * 64 + lane] = ok ? f32x2{K + dm, s2 - s1 * dm} : f32x2{0.f, 0.f};
}
__device__ __forceinline__ void w4_store_stats(const Wino4Args& a, const f32x2* sbuf, int slot, int kq, int tile_first, int co) {
    float msum = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const f32x2 t = sbuf[slot * 64 + kq * 16 + i]; msum += t[0]; m2 += t[1]; }
    const float mean = msum * (1.0f / 16.0f);
    float dev = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = sbuf[slot * 64 + kq * 16 + i][0] - mean; dev += d * d; }
    if (co < a.Cout && tile_first < a.nTiles) {
        const int b = tile_first / a.tilesPerImg, blk = (tile_first - b * a.tilesPerImg) >> 4;
        *(f32x2*)(a.stats + (((size_t)b * a.Cout + co) * a.sblk + blk) * 2) = f32x2{mean, m2 + 16.0f * dev};
    }
}

// Workgroup id -> (tile block, channel group, K split).  Ids are dealt round-robin to the 8 XCDs (each with its own
// L2), so the order decides what the co-resident workgroups of an XCD share:
//  * many tile blocks (tb_per_xcd > 0): XCD k = id % 8 takes a contiguous band of tile blocks and walks it with the
//    channel group fastest -> input tiles (all channel groups of a block, vertical halos of neighbours) are shared;
//  * few tile blocks (small-spatial UNet layers, operand-traffic bound): all tile blocks of one (split, channel group)
//    unit go to the same XCD, unit u -> XCD u % 8, so the unit's filter slice is fetched into that L2 once.
__device__ __forceinline__ bool w4_work(const Wino4Args& a, int vid, int& tblock, int& cg, int& z) {
    if (a.tb_per_xcd > 0) {
        z = vid / a.gx;
        const int x = vid - z * a.gx;
        const int xcd = x & 7, seq = x >> 3;
        const int tb_local = seq / a.ncg;
        cg = seq - tb_local * a.ncg;
        tblock = xcd * a.tb_per_xcd + tb_local;
        return tblock < a.ntb;                                   // ragged band (whole workgroup)
    }
    const int xcd = vid & 7, q = vid >> 3;
    const int ul = q / a.ntb;
    tblock = q - ul * a.ntb;
    const int u = ul * 8 + xcd;
    z = u / a.ncg;
    cg = u - z * a.ncg;
    return u < a.ncg * a.splits;
}

constexpr int W4_STAGE_F4 = 36 * 4 * 32;          // f32x4 per stage: [36 positions][4 k-quads][32 tiles]
constexpr int W4_RING = 12;                       // filter ring slots (divides 36); prefetch distance RING-1 positions

// B^T applied to a 6-vector (T = float, or float2 for two columns at once on the packed-fp32 VALU ops)
template <class T>
__device__ __forceinline__ void w4_in1d(const T (&d)[6], T (&t)[6]) {
    const T a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1];
    const T c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = a + b;
    t[2] = a - b;
    t[3] = c + e;
    t[4] = c - e;
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A^T applied to a 6-vector
__device__ __forceinline__ void w4_out1d(const float (&m)[6], float (&y)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// y = sum_z part[z] (+ bias[channel]) (+ res), fixed order
__global__ void skp_wino4_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
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

struct Wino4Grid { int ntb, ncg, tb_per_xcd; unsigned gx; int rounds; };

// ---- accumulators ---------------------------------------------------------------------------------------------------
// 36 positions x 2 tile blocks x 4 registers = 288 accumulator registers; a wave owns 256 AGPRs + 256 VGPRs.  Left to the
// register allocator, the 32 that do not fit the AGPR file get shuttled through temporaries around every MFMA (and one
// unlucky schedule turned ALL of them into VGPR <-> AGPR copies).  So: positions 0-31 live in a[0:255] by NAME -- their MFMAs
// are inline assembly (w4s_mfma_named / w4c_mfma_named), the compiler never sees those registers (every statement lists the whole AGPR file as clobbered, so
// it keeps nothing of its own there; audit: no compiler-generated v_accvgpr_* may appear in the .s) -- and positions 32-35 are
// ordinary variables, which the allocator then has to keep in VGPRs (v_mfma takes a VGPR accumulator as well).
#define W4_AGPR_CLOBBERS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
template <int R>
__device__ __forceinline__ float w4_acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
    return v;
}
template <class F, int... I>
__device__ __forceinline__ void w4_unroll(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }



}  // namespace
