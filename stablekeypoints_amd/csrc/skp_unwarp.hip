// Tail of the augmented inference (eval.py:239-353), one kernel per batch of views:
//   per view v and selected token k:  U = bilinear_{R->S}(M[v,k])                       optimize.py:63-70 (collect_maps, upsample_res)
//                                      sum[k] += grid_sample(U, affine_grid(theta_inv_v))  eval.py:268-272 (invertible_transform.inverse)
//                                      num    += grid_sample(1, affine_grid(theta_inv_v))  (coverage of the un-warped view)
// The [n,K,S,S] up-sampled maps, the ones tensor and the two un-warped [n,K,S,S] tensors of the reference are never
// materialised: a thread owns one output pixel, walks the views, and for each view samples the R x R map through the
// composed bilinear o bilinear footprint (4 neighbours x 4 taps) with PyTorch's formulas term by term
// (upsample_bilinear2d align_corners=False: src = max(scale (dst + 0.5) - 0.5, 0); grid_sample bilinear / zeros /
// align_corners=False: ix = ((x + 1) W - 1) / 2, corner order nw, ne, sw, se).  The coverage is the same for every token.
#include "skp_common.h"

#define SKP_UNWARP_KMAX 32

struct UpTap { int i0, i1; float l0, l1; };

__device__ __forceinline__ UpTap skp_up_tap(int dst, float scale, int n) {
    // at::native upsample_bilinear2d: area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    UpTap t;
    t.i0 = (int)src;
    t.i1 = t.i0 + ((t.i0 < n - 1) ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}

__global__ __launch_bounds__(256) void skp_unwarp_accumulate_kernel(const float* __restrict__ M, const float* __restrict__ theta_inv,
                                                                    int n, int K, int R, int S, float* __restrict__ tot,
                                                                    float* __restrict__ num, int finish) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= S * S) return;
    const int i = p / S, j = p - i * S;
    // affine_grid base coordinates (align_corners=False)
    const float xn = (2.0f * (float)j + 1.0f) / (float)S - 1.0f;
    const float yn = (2.0f * (float)i + 1.0f) / (float)S - 1.0f;
    const float scale = (float)R / (float)S;
    float acc[SKP_UNWARP_KMAX];
#pragma unroll
    for (int k = 0; k < SKP_UNWARP_KMAX; ++k) acc[k] = 0.f;
    float cnt = 0.f;
    const size_t RR = (size_t)R * R;
    for (int v = 0; v < n; ++v) {
        const float* th = theta_inv + v * 6;
        const float xs = th[0] * xn + th[1] * yn + th[2];
        const float ys = th[3] * xn + th[4] * yn + th[5];
        const float ix = ((xs + 1.0f) * (float)S - 1.0f) * 0.5f;
        const float iy = ((ys + 1.0f) * (float)S - 1.0f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix;      // ix_se - ix  with ix_se = ix_nw + 1
        const float wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const float wq[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};      // nw, ne, sw, se
        const int qx[4] = {x0, x0 + 1, x0, x0 + 1}, qy[4] = {y0, y0, y0 + 1, y0 + 1};
        bool in[4];
        UpTap tx[4], ty[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            in[q] = qx[q] >= 0 && qx[q] < S && qy[q] >= 0 && qy[q] < S;
            tx[q] = skp_up_tap(in[q] ? qx[q] : 0, scale, R);
            ty[q] = skp_up_tap(in[q] ? qy[q] : 0, scale, R);
            if (in[q]) cnt += wq[q];
        }
        const float* Mv = M + (size_t)v * K * RR;
#pragma unroll
        for (int k = 0; k < SKP_UNWARP_KMAX; ++k) {
            if (k < K) {
                const float* m = Mv + (size_t)k * RR;
                float a = acc[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (in[q]) {
                        const float* r0 = m + (size_t)ty[q].i0 * R;
                        const float* r1 = m + (size_t)ty[q].i1 * R;
                        const float u = ty[q].l0 * (tx[q].l0 * r0[tx[q].i0] + tx[q].l1 * r0[tx[q].i1]) +
                                        ty[q].l1 * (tx[q].l0 * r1[tx[q].i0] + tx[q].l1 * r1[tx[q].i1]);
                        a += u * wq[q];
                    }
                }
                acc[k] = a;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < SKP_UNWARP_KMAX; ++k) {
        if (k < K) {
            float o = acc[k];
            if (finish) { o = o / cnt; o = (o != o) ? 0.f : o; }          // eval.py:343-346: sum / count, NaN -> 0
            tot[(size_t)k * S * S + p] = o;
        }
    }
    if (num) num[p] = cnt;
}

extern "C" int skp_unwarp_accumulate_f32(const float* M, const float* theta_inv, int n, int K, int R, int S, float* tot,
                                         float* num, int finish, void* stream) {
    if (!M || !theta_inv || !tot || n <= 0 || K <= 0 || R <= 0 || S <= 0) return SKP_E_BADARG;
    if (K > SKP_UNWARP_KMAX || S > 8192 || R > 8192) return SKP_E_RANGE;
    const long np = (long)S * S;
    hipLaunchKernelGGL(skp_unwarp_accumulate_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       M, theta_inv, n, K, R, S, tot, num, finish);
    return skp_launch_status();
}
