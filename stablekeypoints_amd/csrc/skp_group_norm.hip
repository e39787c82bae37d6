// Fused GroupNorm (+ per-(sample,channel) offset) (+ SiLU), forward and backward, NCHW fp32.
//
// Inside the hooked UNet / VAE forward the reference (diffusers) runs  x -> [+ conv bias / time embedding]
// -> GroupNorm -> SiLU  as 4-5 separate full passes over the activation (ATen: RowwiseMoments, the GN apply
// kernel, silu, broadcast adds); at B=8, 512^2 the VAE activations are 1 GB each and ATen's statistics kernel
// gets one workgroup per (sample, group) = 256 workgroups => ~1.2 TB/s (profiles/).  Here:
//   stats  : every (sample, group) row is split over `nsplit` workgroups (float4 loads, shifted sums so that
//            E[(x-K)^2] - E[x-K]^2 does not cancel), partials combined in fp64 by the consumer
//   apply  : y = act( ((x + off[n,c]) - mean) * rstd * gamma[c] + beta[c] ),  act = SiLU or identity
//   bwd    : dz = dy * act'(z);  dx = rstd * (g - mean_grp(g) - xhat * mean_grp(g * xhat)),  g = dz * gamma[c]
// HBM-bound: 1 read for stats + 1 read + 1 write for apply (3 passes instead of 5+).
#include "skp_common.h"
#include <stdlib.h>

struct GNArgs {
    const float* x; const float* off; const float* gamma; const float* beta;
    int N, C, G, HW, nsplit, silu;
    float eps;
    long L;              // elements per (sample, group) row = (C/G) * HW
};

// partial[(row * nsplit + split) * 3 + {0,1,2}] = {K, sum(v-K), sum((v-K)^2)} over this split's elements
__global__ __launch_bounds__(256) void skp_gn_stats_kernel(GNArgs a, float* __restrict__ partial) {
    __shared__ float red[4];
    const int row = blockIdx.y, split = blockIdx.x, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    const float* xr = a.x + (size_t)row * a.L;
    const float* offr = a.off ? a.off + (size_t)n * a.C + g * Cg : nullptr;
    const long q4 = a.L / 4;                                   // HW % 4 == 0 is required by the launcher
    const long per = (q4 + a.nsplit - 1) / a.nsplit;
    const long lo = split * per, hi = (lo + per < q4) ? lo + per : q4;
    const float K = xr[0] + (offr ? offr[0] : 0.f);
    float s1 = 0.f, s2 = 0.f;
    const int hw4 = a.HW / 4;
    for (long i = lo + tid; i < hi; i += 256) {
        f32x4 v = *(const f32x4*)(xr + i * 4);
        if (offr) v += offr[i / hw4];
        v -= K;
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s1 = skp_block_sum_256(s1, red);
    s2 = skp_block_sum_256(s2, red);
    if (tid == 0) {
        float* p = partial + ((size_t)row * a.nsplit + split) * 3;
        p[0] = K; p[1] = s1; p[2] = s2;
    }
}

__device__ __forceinline__ void skp_gn_finalize(const float* partial, int row, int nsplit, long L, float eps,
                                                float& mean, float& rstd) {
    double s1 = 0.0, s2 = 0.0;
    const float K = partial[(size_t)row * nsplit * 3];
    for (int i = 0; i < nsplit; ++i) {
        const float* p = partial + ((size_t)row * nsplit + i) * 3;
        s1 += (double)p[1]; s2 += (double)p[2];
    }
    const double m = s1 / (double)L;
    double var = s2 / (double)L - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)((double)K + m);
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// Statistics from the block moments a producing convolution left behind (skp_conv3x3_f4_stats_f32 / skp_conv3x3_s2_stats_f32):
// bs[n][c][blk] = {mean, M2 = sum (y - mean)^2} over `pix` pixels each.  One workgroup per (sample, group); the blocks are
// merged in fp64 (group mean first, then M2 = sum [M2_b + pix (mean_b + o_c - mean)^2]: no difference of large sums
// anywhere); the per-(sample, channel) offset only shifts a block's mean.
__global__ __launch_bounds__(256) void skp_gn_from_blocks_kernel(GNArgs a, const float* __restrict__ bs, int nblk, int pix,
                                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double red[4];
    __shared__ double gmean;
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    auto block_sum = [&](double v) {                           // fixed order: wave butterflies, then the four waves
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    double s1 = 0.0;
    for (int e = tid; e < Cg * nblk; e += 256) {
        const int cl = e / nblk, blk = e - cl * nblk, c = g * Cg + cl;
        const double o = a.off ? (double)a.off[(size_t)n * a.C + c] : 0.0;
        s1 += (double)bs[(((size_t)n * a.C + c) * nblk + blk) * 2] + o;
    }
    const double m = block_sum(s1) / (double)(Cg * nblk);      // equal block sizes: the mean of the block means
    if (tid == 0) gmean = m;
    __syncthreads();
    double m2 = 0.0;
    for (int e = tid; e < Cg * nblk; e += 256) {
        const int cl = e / nblk, blk = e - cl * nblk, c = g * Cg + cl;
        const float* p = bs + (((size_t)n * a.C + c) * nblk + blk) * 2;
        const double o = a.off ? (double)a.off[(size_t)n * a.C + c] : 0.0;
        const double d = (double)p[0] + o - gmean;
        m2 += (double)p[1] + (double)pix * d * d;
    }
    const double t2 = block_sum(m2);
    if (tid == 0) {
        double var = t2 / (double)a.L;
        var = var < 0.0 ? 0.0 : var;
        mean_out[row] = (float)gmean;
        rstd_out[row] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

// partial == nullptr: mean / rstd were produced by skp_gn_from_blocks_kernel (read, not written)
__global__ __launch_bounds__(256) void skp_gn_apply_kernel(GNArgs a, const float* __restrict__ partial,
                                                           float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out) {
    const int row = blockIdx.y, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    float mean, rstd;
    if (partial) {
        skp_gn_finalize(partial, row, a.nsplit, a.L, a.eps, mean, rstd);
        if (blockIdx.x == 0 && tid == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    } else {
        mean = mean_out[row]; rstd = rstd_out[row];
    }
    const float* xr = a.x + (size_t)row * a.L;
    float* yr = y + (size_t)row * a.L;
    const int hw4 = a.HW / 4;
    const long q4 = a.L / 4;
    for (long i = (long)blockIdx.x * 256 + tid; i < q4; i += (long)gridDim.x * 256) {
        const int cl = (int)(i / hw4), c = g * Cg + cl;
        const float sc = rstd * a.gamma[c];
        const float sh = a.beta[c] + ((a.off ? a.off[(size_t)n * a.C + c] : 0.f) - mean) * sc;
        f32x4 v = *(const f32x4*)(xr + i * 4);
        v = v * sc + sh;
        if (a.silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + __expf(-v[e]));
        }
        *(f32x4*)(yr + i * 4) = v;
    }
}

// backward pass 1: partial[(row*nsplit+split)*2 + {0,1}] = {sum g, sum g*xhat},  g = dy * act'(z) * gamma
__global__ __launch_bounds__(256) void skp_gn_bwd_stats_kernel(GNArgs a, const float* __restrict__ dy,
                                                               const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in,
                                                               float* __restrict__ partial) {
    __shared__ float red[4];
    const int row = blockIdx.y, split = blockIdx.x, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const float* xr = a.x + (size_t)row * a.L;
    const float* dr = dy + (size_t)row * a.L;
    const int hw4 = a.HW / 4;
    const long q4 = a.L / 4;
    const long per = (q4 + a.nsplit - 1) / a.nsplit;
    const long lo = split * per, hi = (lo + per < q4) ? lo + per : q4;
    float s1 = 0.f, s2 = 0.f;
    for (long i = lo + tid; i < hi; i += 256) {
        const int c = g * Cg + (int)(i / hw4);
        const float gam = a.gamma[c], bet = a.beta[c];
        const float o = a.off ? a.off[(size_t)n * a.C + c] : 0.f;
        const f32x4 xv = *(const f32x4*)(xr + i * 4);
        const f32x4 dv = *(const f32x4*)(dr + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] + o - mean) * rstd;
            float gz = dv[e];
            if (a.silu) {
                const float z = xh * gam + bet;
                const float sg = 1.0f / (1.0f + __expf(-z));
                gz *= sg * (1.0f + z * (1.0f - sg));
            }
            gz *= gam;
            s1 += gz; s2 += gz * xh;
        }
    }
    s1 = skp_block_sum_256(s1, red);
    s2 = skp_block_sum_256(s2, red);
    if (tid == 0) {
        float* p = partial + ((size_t)row * a.nsplit + split) * 2;
        p[0] = s1; p[1] = s2;
    }
}

__global__ __launch_bounds__(256) void skp_gn_bwd_apply_kernel(GNArgs a, const float* __restrict__ dy,
                                                               const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in,
                                                               const float* __restrict__ partial,
                                                               float* __restrict__ dx, const float* __restrict__ dadd) {
    const int row = blockIdx.y, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    const float mean = mean_in[row], rstd = rstd_in[row];
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < a.nsplit; ++i) {
        const float* p = partial + ((size_t)row * a.nsplit + i) * 2;
        t1 += (double)p[0]; t2 += (double)p[1];
    }
    const float m1 = (float)(t1 / (double)a.L), m2 = (float)(t2 / (double)a.L);
    const float* xr = a.x + (size_t)row * a.L;
    const float* dr = dy + (size_t)row * a.L;
    float* dxr = dx + (size_t)row * a.L;
    const int hw4 = a.HW / 4;
    const long q4 = a.L / 4;
    for (long i = (long)blockIdx.x * 256 + tid; i < q4; i += (long)gridDim.x * 256) {
        const int c = g * Cg + (int)(i / hw4);
        const float gam = a.gamma[c], bet = a.beta[c];
        const float o = a.off ? a.off[(size_t)n * a.C + c] : 0.f;
        const f32x4 xv = *(const f32x4*)(xr + i * 4);
        const f32x4 dv = *(const f32x4*)(dr + i * 4);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] + o - mean) * rstd;
            float gz = dv[e];
            if (a.silu) {
                const float z = xh * gam + bet;
                const float sg = 1.0f / (1.0f + __expf(-z));
                gz *= sg * (1.0f + z * (1.0f - sg));
            }
            gz *= gam;
            r[e] = rstd * (gz - m1 - xh * m2);
        }
        if (dadd) r += *(const f32x4*)(dadd + (size_t)row * a.L + i * 4);     // gradient of x's other use (fork entry)
        *(f32x4*)(dxr + i * 4) = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One-pass forms for rows that fit the registers of one workgroup (every GroupNorm of the UNet: (C/G) * HW <= 64 K elements).
// The multi-kernel forms above cost a UNet GroupNorm 2-3 launches of ~8-12 us each on a few hundred KB (statistics or block
// merge, [coefficients,] apply) and two more backward -- launch ramp and tail, not bandwidth.  Here ONE workgroup of 1024
// threads owns a (sample, group) row: the row is read once into registers (VPT float4 per thread), mean and variance are two
// in-register passes (exact two-pass form, no shifted sums needed), the normalised (+SiLU) row is written from the registers;
// backward: x and dy in registers, the two group sums, dx.  Fixed-order reductions (wave butterflies, then the 16 waves): the
// result is bit-reproducible.  HBM traffic: 1 read + 1 write forward (was 2 + 1), 2 reads + 1 write backward (was 4 + 1).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float skp_block_sum_1024(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                                           // `red` may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
}

template <int VPT>
__global__ __launch_bounds__(1024) void skp_gn_onepass_fwd_kernel(GNArgs a, float* __restrict__ y, float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    const float* xr = a.x + (size_t)row * a.L;
    float* yr = y + (size_t)row * a.L;
    const int hw4 = a.HW / 4, q4 = (int)(a.L / 4);
    f32x4 v[VPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + 1024 * j;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < q4) {
            v[j] = *(const f32x4*)(xr + (size_t)i * 4);
            if (a.off) v[j] += a.off[(size_t)n * a.C + g * Cg + i / hw4];
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = skp_block_sum_1024(s, red) / (float)a.L;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
        if (tid + 1024 * j < q4) {
            const f32x4 d = v[j] - mean;
            s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    const float var = skp_block_sum_1024(s2, red) / (float)a.L;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (tid == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + 1024 * j;
        if (i < q4) {
            const int c = g * Cg + i / hw4;
            const float sc = rstd * a.gamma[c];
            const float sh = a.beta[c] - mean * sc;             // the offset is already inside v
            f32x4 r = v[j] * sc + sh;
            if (a.silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = r[e] / (1.0f + __expf(-r[e]));
            }
            *(f32x4*)(yr + (size_t)i * 4) = r;
        }
    }
}

template <int VPT>
__global__ __launch_bounds__(1024) void skp_gn_onepass_bwd_kernel(GNArgs a, const float* __restrict__ dy, const float* __restrict__ mean_in,
                                                                  const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                                  const float* __restrict__ dadd) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = row / a.G, g = row - n * a.G, Cg = a.C / a.G;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const float* xr = a.x + (size_t)row * a.L;
    const float* dr = dy + (size_t)row * a.L;
    float* dxr = dx + (size_t)row * a.L;
    const int hw4 = a.HW / 4, q4 = (int)(a.L / 4);
    f32x4 xh[VPT], gz[VPT];                                    // xhat and g = dy * act'(z) * gamma of this thread's elements
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + 1024 * j;
        xh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        gz[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < q4) {
            const int c = g * Cg + i / hw4;
            const float gam = a.gamma[c], bet = a.beta[c];
            const float o = a.off ? a.off[(size_t)n * a.C + c] : 0.f;
            const f32x4 xv = *(const f32x4*)(xr + (size_t)i * 4);
            const f32x4 dv = *(const f32x4*)(dr + (size_t)i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (xv[e] + o - mean) * rstd;
                float t = dv[e];
                if (a.silu) {
                    const float z = h * gam + bet;
                    const float sg = 1.0f / (1.0f + __expf(-z));
                    t *= sg * (1.0f + z * (1.0f - sg));
                }
                t *= gam;
                xh[j][e] = h; gz[j][e] = t;
                s1 += t; s2 += t * h;
            }
        }
    }
    const float m1 = skp_block_sum_1024(s1, red) / (float)a.L;
    const float m2 = skp_block_sum_1024(s2, red) / (float)a.L;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = tid + 1024 * j;
        if (i < q4) {
            f32x4 r = (gz[j] - m1 - xh[j] * m2) * rstd;
            if (dadd) r += *(const f32x4*)(dadd + (size_t)row * a.L + (size_t)i * 4);
            *(f32x4*)(dxr + (size_t)i * 4) = r;
        }
    }
}

// float4 per thread the one-pass forms would need for this row length; 0 = not served
static int gn_onepass_vpt(const GNArgs& a, bool backward) {
    const long q4 = a.L / 4, need = (q4 + 1023) / 1024;
    // (the forward at 8 float4 per thread is left out: this compiler spills it -- 128 registers + 408 bytes of scratch -- while
    // 10 and 16 compile clean at 86 / 122; the backward holds two arrays and stops at 10 = 112 registers)
    const int fsizes[] = {1, 2, 4, 10, 16}, bsizes[] = {1, 2, 4, 8, 10};
    for (int k = 0; k < 5; ++k) {
        const int v = backward ? bsizes[k] : fsizes[k];
        if (need <= v) return v;
    }
    return 0;
}

static int gn_onepass_fwd(const GNArgs& a, float* y, float* mean, float* rstd, hipStream_t st) {
    const dim3 grid(a.N * a.G), block(1024);
    switch (gn_onepass_vpt(a, false)) {
        case 1: hipLaunchKernelGGL(skp_gn_onepass_fwd_kernel<1>, grid, block, 0, st, a, y, mean, rstd); break;
        case 2: hipLaunchKernelGGL(skp_gn_onepass_fwd_kernel<2>, grid, block, 0, st, a, y, mean, rstd); break;
        case 4: hipLaunchKernelGGL(skp_gn_onepass_fwd_kernel<4>, grid, block, 0, st, a, y, mean, rstd); break;
        case 10: hipLaunchKernelGGL(skp_gn_onepass_fwd_kernel<10>, grid, block, 0, st, a, y, mean, rstd); break;
        case 16: hipLaunchKernelGGL(skp_gn_onepass_fwd_kernel<16>, grid, block, 0, st, a, y, mean, rstd); break;
        default: return -100;
    }
    return skp_launch_status();
}

static int gn_onepass_bwd(const GNArgs& a, const float* dy, const float* mean, const float* rstd, float* dx, const float* dadd,
                          hipStream_t st) {
    const dim3 grid(a.N * a.G), block(1024);
    switch (gn_onepass_vpt(a, true)) {
        case 1: hipLaunchKernelGGL(skp_gn_onepass_bwd_kernel<1>, grid, block, 0, st, a, dy, mean, rstd, dx, dadd); break;
        case 2: hipLaunchKernelGGL(skp_gn_onepass_bwd_kernel<2>, grid, block, 0, st, a, dy, mean, rstd, dx, dadd); break;
        case 4: hipLaunchKernelGGL(skp_gn_onepass_bwd_kernel<4>, grid, block, 0, st, a, dy, mean, rstd, dx, dadd); break;
        case 8: hipLaunchKernelGGL(skp_gn_onepass_bwd_kernel<8>, grid, block, 0, st, a, dy, mean, rstd, dx, dadd); break;
        case 10: hipLaunchKernelGGL(skp_gn_onepass_bwd_kernel<10>, grid, block, 0, st, a, dy, mean, rstd, dx, dadd); break;
        default: return -100;
    }
    return skp_launch_status();
}

static int gn_fill(GNArgs& a, const float* x, const float* off, const float* gamma, const float* beta, int N, int C,
                   int G, int HW, float eps, int silu) {
    if (!x || !gamma || !beta || N <= 0 || C <= 0 || G <= 0 || HW <= 0) return SKP_E_BADARG;
    if (C % G || HW % 4 || (long)N * G > 65535) return SKP_E_RANGE;
    a.x = x; a.off = off; a.gamma = gamma; a.beta = beta;
    a.N = N; a.C = C; a.G = G; a.HW = HW; a.eps = eps; a.silu = silu;
    a.L = (long)(C / G) * HW;
    // enough workgroups to fill the chip: ~2048 total, each split >= 4096 elements
    long rows = (long)N * G, want = (2048 + rows - 1) / rows, cap = a.L / 4096;
    if (cap < 1) cap = 1;
    a.nsplit = (int)(want < cap ? want : cap);
    if (a.nsplit < 1) a.nsplit = 1;
    if (a.nsplit > 64) a.nsplit = 64;
    return 0;
}

extern "C" int skp_group_norm_nsplit(int N, int C, int G, int HW) {
    GNArgs a{};
    static const float dummy = 0.f;
    int rc = gn_fill(a, &dummy, nullptr, &dummy, &dummy, N, C, G, HW, 1e-5f, 0);
    return rc ? rc : a.nsplit;
}

// 1 when the forward of this shape holds its (sample, group) rows in registers (one launch, exact statistics from the loaded row):
// callers use it to decide whether a producing convolution should leave block statistics behind at all.
extern "C" int skp_group_norm_onepass_ok(int N, int C, int G, int HW) {
    if (N <= 0 || C <= 0 || G <= 0 || HW <= 0 || C % G || (HW & 3)) return 0;
    GNArgs a{};
    a.N = N; a.C = C; a.G = G; a.HW = HW; a.L = (long)(C / G) * HW;
    return gn_onepass_vpt(a, false) ? 1 : 0;
}

extern "C" int skp_group_norm_fwd_f32(const float* x, const float* off, const float* gamma, const float* beta,
                                      float* y, float* mean, float* rstd, float* workspace, int N, int C, int G,
                                      int HW, float eps, int silu, void* stream) {
    GNArgs a{};
    int rc = gn_fill(a, x, off, gamma, beta, N, C, G, HW, eps, silu);
    if (rc) return rc;
    if (!y || !mean || !rstd || !workspace) return SKP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (gn_onepass_vpt(a, false)) return gn_onepass_fwd(a, y, mean, rstd, st);
    hipLaunchKernelGGL(skp_gn_stats_kernel, dim3(a.nsplit, N * G), dim3(256), 0, st, a, workspace);
    rc = skp_launch_status();
    if (rc) return rc;
    long blocks = (a.L / 4 + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(skp_gn_apply_kernel, dim3((unsigned)blocks, N * G), dim3(256), 0, st, a,
                       (const float*)workspace, y, mean, rstd);
    return skp_launch_status();
}

// Forward with the statistics taken from a producer's block sums instead of a pass over x:
// bs [N][C][nblk][2], every block covering `pix` pixels (nblk * pix == HW).
extern "C" int skp_group_norm_fwd_blocks_f32(const float* x, const float* off, const float* gamma, const float* beta,
                                             float* y, float* mean, float* rstd, const float* bs, int nblk, int pix, int N, int C,
                                             int G, int HW, float eps, int silu, void* stream) {
    GNArgs a{};
    int rc = gn_fill(a, x, off, gamma, beta, N, C, G, HW, eps, silu);
    if (rc) return rc;
    if (!y || !mean || !rstd || !bs || nblk <= 0 || pix <= 0) return SKP_E_BADARG;
    if ((long)nblk * pix != HW) return SKP_E_RANGE;
    hipStream_t st = (hipStream_t)stream;
    if (gn_onepass_vpt(a, false)) return gn_onepass_fwd(a, y, mean, rstd, st);   // the row is in registers anyway: exact statistics
    hipLaunchKernelGGL(skp_gn_from_blocks_kernel, dim3(N * G), dim3(256), 0, st, a, bs, nblk, pix, mean, rstd);
    rc = skp_launch_status();
    if (rc) return rc;
    long blocks = (a.L / 4 + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(skp_gn_apply_kernel, dim3((unsigned)blocks, N * G), dim3(256), 0, st, a, (const float*)nullptr, y, mean,
                       rstd);
    return skp_launch_status();
}

// (scale, shift) per (sample, channel) of y = (x + off - mean) * rstd * gamma + beta = x * scale + shift, for consumers that
// apply the normalisation themselves (skp_conv3x3_f4_gn_f32).  partial == nullptr: mean / rstd are read, else finalised here.
__global__ __launch_bounds__(256) void skp_gn_coef_kernel(GNArgs a, const float* __restrict__ partial, float* __restrict__ mean_io,
                                                          float* __restrict__ rstd_io, float* __restrict__ coef) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N * a.C) return;
    const int n = i / a.C, c = i - n * a.C, Cg = a.C / a.G, row = n * a.G + c / Cg;
    float mean, rstd;
    if (partial) {
        skp_gn_finalize(partial, row, a.nsplit, a.L, a.eps, mean, rstd);
        if (c % Cg == 0) { mean_io[row] = mean; rstd_io[row] = rstd; }
    } else {
        mean = mean_io[row]; rstd = rstd_io[row];
    }
    const float sc = rstd * a.gamma[c];
    coef[2 * i] = sc;
    coef[2 * i + 1] = a.beta[c] + ((a.off ? a.off[i] : 0.f) - mean) * sc;
}

/* Statistics of GroupNorm(x + off) and the per-(sample, channel) (scale, shift) pairs, without the apply pass.
 * bs != NULL: statistics from a producing convolution's block sums (as skp_group_norm_fwd_blocks_f32), x is not read;
 * else a statistics pass over x (workspace: N*G*64*3 floats).  coef: [N,C,2] written; mean, rstd: [N,G] written. */
extern "C" int skp_group_norm_coef_f32(const float* x, const float* off, const float* gamma, const float* beta, float* mean,
                                       float* rstd, float* coef, const float* bs, int nblk, int pix, float* workspace, int N,
                                       int C, int G, int HW, float eps, void* stream) {
    GNArgs a;
    int rc = gn_fill(a, x ? x : gamma, off, gamma, beta, N, C, G, HW, eps, 0);
    if (rc) return rc;
    if (!mean || !rstd || !coef) return SKP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = (unsigned)((N * C + 255) / 256);
    if (bs) {
        if (nblk <= 0 || pix <= 0) return SKP_E_BADARG;
        if ((long)nblk * pix != HW) return SKP_E_RANGE;
        hipLaunchKernelGGL(skp_gn_from_blocks_kernel, dim3(N * G), dim3(256), 0, st, a, bs, nblk, pix, mean, rstd);
        hipLaunchKernelGGL(skp_gn_coef_kernel, dim3(nb), dim3(256), 0, st, a, (const float*)nullptr, mean, rstd, coef);
    } else {
        if (!x || !workspace) return SKP_E_BADARG;
        hipLaunchKernelGGL(skp_gn_stats_kernel, dim3(a.nsplit, N * G), dim3(256), 0, st, a, workspace);
        hipLaunchKernelGGL(skp_gn_coef_kernel, dim3(nb), dim3(256), 0, st, a, (const float*)workspace, mean, rstd, coef);
    }
    return skp_launch_status();
}

static int gn_bwd_run(const float* x, const float* off, const float* gamma, const float* beta, const float* dy, const float* mean,
                      const float* rstd, float* dx, const float* dadd, float* workspace, int N, int C, int G, int HW, float eps,
                      int silu, void* stream);

extern "C" int skp_group_norm_bwd_f32(const float* x, const float* off, const float* gamma, const float* beta,
                                      const float* dy, const float* mean, const float* rstd, float* dx,
                                      float* workspace, int N, int C, int G, int HW, float eps, int silu,
                                      void* stream) {
    return gn_bwd_run(x, off, gamma, beta, dy, mean, rstd, dx, nullptr, workspace, N, C, G, HW, eps, silu, stream);
}

// dx = (input gradient of the norm) + dadd: x feeds the norm AND something else (the residual path of its block); the
// gradient of that other use rides in this kernel instead of an add pass of its own.  dadd [N,C,HW] (may alias nothing).
extern "C" int skp_group_norm_bwd_add_f32(const float* x, const float* off, const float* gamma, const float* beta,
                                          const float* dy, const float* mean, const float* rstd, float* dx, const float* dadd,
                                          float* workspace, int N, int C, int G, int HW, float eps, int silu, void* stream) {
    if (!dadd) return SKP_E_BADARG;
    return gn_bwd_run(x, off, gamma, beta, dy, mean, rstd, dx, dadd, workspace, N, C, G, HW, eps, silu, stream);
}

static int gn_bwd_run(const float* x, const float* off, const float* gamma, const float* beta, const float* dy, const float* mean,
                      const float* rstd, float* dx, const float* dadd, float* workspace, int N, int C, int G, int HW, float eps,
                      int silu, void* stream) {
    GNArgs a{};
    int rc = gn_fill(a, x, off, gamma, beta, N, C, G, HW, eps, silu);
    if (rc) return rc;
    if (!dy || !mean || !rstd || !dx || !workspace) return SKP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (gn_onepass_vpt(a, true)) return gn_onepass_bwd(a, dy, mean, rstd, dx, dadd, st);
    hipLaunchKernelGGL(skp_gn_bwd_stats_kernel, dim3(a.nsplit, N * G), dim3(256), 0, st, a, dy, mean, rstd, workspace);
    rc = skp_launch_status();
    if (rc) return rc;
    long blocks = (a.L / 4 + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(skp_gn_bwd_apply_kernel, dim3((unsigned)blocks, N * G), dim3(256), 0, st, a, dy, mean, rstd,
                       (const float*)workspace, dx, dadd);
    return skp_launch_status();
}

// out[n,c,p] = a[n,c,p] + b[n,c,p] + bias[c]   (ResnetBlock2D tail: shortcut + conv2(no bias) + conv2.bias in one pass
// instead of a broadcast bias-add pass followed by a residual-add pass)
__global__ __launch_bounds__(256) void skp_add_bias_residual_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    int C, int hw4, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c = (int)((i / hw4) % C);
        const f32x4 va = *(const f32x4*)(a + i * 4), vb = *(const f32x4*)(b + i * 4);
        *(f32x4*)(out + i * 4) = va + vb + bias[c];
    }
}

extern "C" int skp_add_bias_residual_f32(const float* a, const float* b, const float* bias, float* out, int N, int C,
                                         int HW, void* stream) {
    if (!a || !b || !bias || !out || N <= 0 || C <= 0 || HW <= 0) return SKP_E_BADARG;
    if (HW % 4) return SKP_E_RANGE;
    const long total4 = (long)N * C * HW / 4;
    long blocks = (total4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(skp_add_bias_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, bias,
                       out, C, HW / 4, total4);
    return skp_launch_status();
}

// GEGLU of the transformer feed-forward (diffusers attention.GEGLU [third party]: h, gate = proj(x).chunk(2, -1);
// h * gelu(gate), exact erf form) on the projection p [rows, 2*inner] in one pass per direction, instead of
// strided gelu / mul kernels forward and gelu_backward / two muls / two slice copies backward.
__device__ __forceinline__ float skp_gelu(float g) { return 0.5f * g * (1.f + erff(g * 0.70710678118654752f)); }
__device__ __forceinline__ float skp_gelu_grad(float g) {
    return 0.5f * (1.f + erff(g * 0.70710678118654752f)) + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}

__global__ __launch_bounds__(256) void skp_geglu_fwd_kernel(const float* __restrict__ p, float* __restrict__ y, int inner4,
                                                            long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long row = i / inner4;
        const int c4 = (int)(i - row * inner4);
        const f32x4* pr = (const f32x4*)p + row * (2 * inner4);
        const f32x4 h = pr[c4], g = pr[inner4 + c4];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = h[j] * skp_gelu(g[j]);
        ((f32x4*)y)[i] = o;
    }
}

__global__ __launch_bounds__(256) void skp_geglu_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dy,
                                                            float* __restrict__ dp, int inner4, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long row = i / inner4;
        const int c4 = (int)(i - row * inner4);
        const f32x4* pr = (const f32x4*)p + row * (2 * inner4);
        f32x4* dr = (f32x4*)dp + row * (2 * inner4);
        const f32x4 h = pr[c4], g = pr[inner4 + c4], d = ((const f32x4*)dy)[i];
        f32x4 dh, dg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dh[j] = d[j] * skp_gelu(g[j]);
            dg[j] = d[j] * h[j] * skp_gelu_grad(g[j]);
        }
        dr[c4] = dh;
        dr[inner4 + c4] = dg;
    }
}

extern "C" int skp_geglu_fwd_f32(const float* p, float* y, int64_t rows, int inner, void* stream) {
    if (!p || !y || rows <= 0 || inner <= 0) return SKP_E_BADARG;
    if (inner % 4) return SKP_E_RANGE;
    const long total4 = rows * (inner / 4);
    long blocks = (total4 + 256 * 2 - 1) / (256 * 2);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(skp_geglu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, y, inner / 4, total4);
    return skp_launch_status();
}

extern "C" int skp_geglu_bwd_f32(const float* p, const float* dy, float* dp, int64_t rows, int inner, void* stream) {
    if (!p || !dy || !dp || rows <= 0 || inner <= 0) return SKP_E_BADARG;
    if (inner % 4) return SKP_E_RANGE;
    const long total4 = rows * (inner / 4);
    long blocks = (total4 + 256 * 2 - 1) / (256 * 2);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(skp_geglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, dy, dp, inner / 4,
                       total4);
    return skp_launch_status();
}

// Layout changes around the transformer blocks (Transformer2DModel [third party]: NCHW <-> tokens permutes), as LDS-tiled
// transposes with coalesced float4 traffic on both sides; the way back also adds the block's residual in the same pass.
//   to_tokens: y[b, p, c] = x[b, c, p]
//   to_nchw  : y[b, c, p] = t[b, p, c] (+ res[b, c, p])
template <bool TO_TOKENS>
__global__ __launch_bounds__(256) void skp_layout_kernel(const float* __restrict__ src, const float* __restrict__ res,
                                                         float* __restrict__ dst, int C, int HW) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tid = threadIdx.x, q = tid & 15, r = tid >> 4;       // float4 column, row (16 rows per pass)
    const size_t base = (size_t)b * C * HW;
    if (TO_TOKENS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // read x[c][p..p+3]
            const int c = c0 + r + 16 * k, p = p0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < C && p < HW) v = *(const f32x4*)(src + base + (size_t)c * HW + p);
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[r + 16 * k][4 * q + j] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // write y[p][c..c+3]
            const int p = p0 + r + 16 * k, c = c0 + 4 * q;
            if (p < HW && c < C) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tile[4 * q + j][r + 16 * k];
                *(f32x4*)(dst + base + (size_t)p * C + c) = v;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // read t[p][c..c+3]
            const int p = p0 + r + 16 * k, c = c0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < HW && c < C) v = *(const f32x4*)(src + base + (size_t)p * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[4 * q + j][r + 16 * k] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // write y[c][p..p+3] (+ res)
            const int c = c0 + r + 16 * k, p = p0 + 4 * q;
            if (c < C && p < HW) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tile[r + 16 * k][4 * q + j];
                const size_t o = base + (size_t)c * HW + p;
                if (res) v += *(const f32x4*)(res + o);
                *(f32x4*)(dst + o) = v;
            }
        }
    }
}

extern "C" int skp_nchw_to_tokens_f32(const float* x, float* y, int B, int C, int HW, void* stream) {
    if (!x || !y || B <= 0 || C <= 0 || HW <= 0) return SKP_E_BADARG;
    if ((C % 4) || (HW % 4) || B > 65535) return SKP_E_RANGE;
    hipLaunchKernelGGL(skp_layout_kernel<true>, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, y, C, HW);
    return skp_launch_status();
}

extern "C" int skp_tokens_to_nchw_f32(const float* t, const float* residual, float* y, int B, int C, int HW, void* stream) {
    if (!t || !y || B <= 0 || C <= 0 || HW <= 0) return SKP_E_BADARG;
    if ((C % 4) || (HW % 4) || B > 65535) return SKP_E_RANGE;
    hipLaunchKernelGGL(skp_layout_kernel<false>, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, t,
                       residual, y, C, HW);
    return skp_launch_status();
}
