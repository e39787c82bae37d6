// Token statistics, on-device token selection, and the sharpening / equivariance losses.
// HBM-bound reductions over the reduced map M [T,R,R] (5 MB at T=77, R=128) + a latency-bound
// single-workgroup selection that replaces ~500 tiny launches and .item() syncs of the reference
// (ptp_utils.py:115-159).
#include "skp_common.h"
#include <string.h>

// (value desc, index asc) ordering == torch.argmax "first maximal index".
__device__ __forceinline__ bool skp_better(float v, int i, float bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}

struct StatsArgs { int T, R, S; float sigma, eps; };

// One workgroup per token.  eval.py:39-111 (find_max_pixel / find_k_max_pixels / mask_radius),
// ptp_utils.py:95-108 (KL against the normalised gaussian), optimize_token.py:203-241.
__global__ __launch_bounds__(256) void skp_token_stats_kernel(const float* __restrict__ M, StatsArgs a,
                                                              int32_t* __restrict__ argmax_out,
                                                              float* __restrict__ kl_out, float* __restrict__ ent_out) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ float red[4];
    __shared__ float s_cy[SKP_MAX_SUBJECTS], s_cx[SKP_MAX_SUBJECTS];   // centres incl. the +0.5 offset
    const int t = blockIdx.x, tid = threadIdx.x, R = a.R, RR = R * R;
    const float* m = M + (size_t)t * RR;
    const float rad = 0.05f * (float)R;                         // eval.py:79
    const float rad2 = rad * rad;
    float maxval0 = 0.f;
    for (int j = 0; j < a.S; ++j) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int p = tid; p < RR; p += 256) {
            const int row = p / R, col = p - row * R;
            float v = m[p];
            for (int q = 0; q < j; ++q) {                       // cumulative masks (eval.py:81,100-109)
                const float dx = (float)col - s_cx[q], dy = (float)row - s_cy[q];
                const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                v = v * ((d2 > rad2) ? 1.0f : 0.0f);
            }
            if (skp_better(v, p, bv, bi)) { bv = v; bi = p; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (skp_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
        __syncthreads();
        bv = red_v[0]; bi = red_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (skp_better(red_v[w], red_i[w], bv, bi)) { bv = red_v[w]; bi = red_i[w]; }
        if (tid == 0) {
            argmax_out[j * a.T + t] = bi;
            s_cy[j] = (float)(bi / R) + 0.5f;
            s_cx[j] = (float)(bi % R) + 0.5f;
        }
        if (j == 0) maxval0 = bv;
        __syncthreads();
    }
    if (ent_out) {
        // ptp_utils.py:165-187 entropy_sort: p = softmax_{R*R}(M[t]) (no epsilon), then torch's Categorical(probs=p):
        // probs = p / sum(p), logits = log(clamp(probs, eps, 1-eps)) with eps = FLT_EPSILON, H = -sum probs*logits.
        float se0 = 0.f;
        for (int p = tid; p < RR; p += 256) se0 += expf(m[p] - maxval0);
        se0 = skp_block_sum_256(se0, red);
        float sp = 0.f;
        for (int p = tid; p < RR; p += 256) sp += expf(m[p] - maxval0) / se0;
        sp = skp_block_sum_256(sp, red);
        float h = 0.f;
        for (int p = tid; p < RR; p += 256) {
            const float pr = (expf(m[p] - maxval0) / se0) / sp;
            const float cl = fminf(fmaxf(pr, 1.1920928955078125e-07f), 1.0f - 1.1920928955078125e-07f);
            h += pr * logf(cl);
        }
        h = skp_block_sum_256(h, red);
        if (tid == 0) ent_out[t] = -h;
    }
    if (!kl_out) return;
    // gaussian centres in pixels: (loc / R) * R, fp32 like the reference (ptp_utils.py:97, optimize_token.py:210)
    float pcy[SKP_MAX_SUBJECTS], pcx[SKP_MAX_SUBJECTS];
    for (int j = 0; j < a.S; ++j) {
        pcy[j] = __fmul_rn(__fdiv_rn(s_cy[j], (float)R), (float)R);
        pcx[j] = __fmul_rn(__fdiv_rn(s_cx[j], (float)R), (float)R);
    }
    const float two_sig2 = 2.0f * a.sigma * a.sigma;
    const float inv_S = 1.0f / (float)a.S;
    const float mx = maxval0 + a.eps;
    // pass B: softmax denominator and gaussian normaliser
    float se = 0.f, zg = 0.f;
    for (int p = tid; p < RR; p += 256) {
        const int row = p / R, col = p - row * R;
        se += expf(m[p] + a.eps - mx);
        float g = 0.f;
        for (int j = 0; j < a.S; ++j) {
            const float dx = (float)col + 0.5f - pcx[j], dy = (float)row + 0.5f - pcy[j];
            g += expf(-(dx * dx + dy * dy) / two_sig2);
        }
        zg += g * inv_S + a.eps;
    }
    se = skp_block_sum_256(se, red);
    zg = skp_block_sum_256(zg, red);
    // pass C: KL(target || softmax)
    float kl = 0.f;
    for (int p = tid; p < RR; p += 256) {
        const int row = p / R, col = p - row * R;
        const float sm = expf(m[p] + a.eps - mx) / se;
        float g = 0.f;
        for (int j = 0; j < a.S; ++j) {
            const float dx = (float)col + 0.5f - pcx[j], dy = (float)row + 0.5f - pcy[j];
            g += expf(-(dx * dx + dy * dy) / two_sig2);
        }
        const float tn = (g * inv_S + a.eps) / zg;
        kl += tn * (logf(tn) - logf(sm));
    }
    kl = skp_block_sum_256(kl, red);
    if (tid == 0) kl_out[t] = kl;
}

extern "C" int skp_token_stats_f32(const float* M, int T, int R, int num_subjects, float sigma, float eps,
                                   int32_t* argmax, float* kl, float* entropy, void* stream) {
    if (!M || !argmax || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (num_subjects < 1 || num_subjects > SKP_MAX_SUBJECTS || R > 4096) return SKP_E_RANGE;
    StatsArgs a{T, R, num_subjects, sigma, eps};
    hipLaunchKernelGGL(skp_token_stats_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, M, a, argmax, kl, entropy);
    return skp_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Selection: ascending-KL candidates + furthest point sampling.  ptp_utils.py:110-112,115-159.
// ---------------------------------------------------------------------------------------------
#define SKP_SEL_MAXT 1024
#define SKP_SEL_MAXC 64

__device__ __forceinline__ float skp_dist(float ay, float ax, float by, float bx) {
    const float dy = __fsub_rn(ay, by), dx = __fsub_rn(ax, bx);     // no fma contraction: keep ties exact
    return __fsqrt_rn(__fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx)));
}

// blockIdx.x = image of a batch (skp_select_tokens_batched): every pointer advances by its per-image stride
__global__ __launch_bounds__(256) void skp_select_kernel(const float* __restrict__ kl,
                                                         const int32_t* __restrict__ argmax_t, int T, int R,
                                                         int n_cand, int top_k, int64_t* __restrict__ cand_out,
                                                         int64_t* __restrict__ sel_out) {
    kl += (size_t)blockIdx.x * T;
    argmax_t += (size_t)blockIdx.x * T;
    cand_out += (size_t)blockIdx.x * n_cand;
    sel_out += (size_t)blockIdx.x * top_k;
    __shared__ float s_kl[SKP_SEL_MAXT];
    __shared__ int s_cand[SKP_SEL_MAXC];
    __shared__ float s_ly[SKP_SEL_MAXC], s_lx[SKP_SEL_MAXC];
    const int tid = threadIdx.x;
    for (int i = tid; i < T; i += 256) s_kl[i] = kl[i];
    __syncthreads();
    // rank by (value asc, index asc) with NaN ordered after every number (torch.argsort puts NaN last): the ranks
    // form a permutation of 0..T-1 for ANY input, so every candidate slot is written exactly once
    for (int i = tid; i < T; i += 256) {
        const float v = s_kl[i];
        const bool vn = v != v;
        int rank = 0;
        for (int j = 0; j < T; ++j) {
            const float u = s_kl[j];
            const bool un = u != u;
            const bool before = (un != vn) ? vn : (un ? (j < i) : (u < v || (u == v && j < i)));
            rank += before ? 1 : 0;
        }
        if (rank < n_cand) s_cand[rank] = i;
    }
    __syncthreads();
    if (tid < n_cand) {
        const int tok = s_cand[tid];
        const int flat = argmax_t[tok];
        // find_max_pixel(...)/image_h : (row+0.5)/R, (col+0.5)/R     ptp_utils.py:127
        s_ly[tid] = __fdiv_rn((float)(flat / R) + 0.5f, (float)R);
        s_lx[tid] = __fdiv_rn((float)(flat % R) + 0.5f, (float)R);
        cand_out[tid] = tok;
    }
    __syncthreads();
    if (tid >= 64) return;                                      // one wave finishes the selection
    // lane c (< n_cand) owns candidate slot c.  Tie rules of the reference's python loops (strict '>' in
    // ascending scan order => the FIRST maximum wins) are kept by (value desc, index asc) wave reductions.
    const int lane = tid;
    const bool act = lane < n_cand;
    const float my_y = act ? s_ly[lane] : 0.f, my_x = act ? s_lx[lane] : 0.f;
    float best = -1.0f; int ba = 0, bb = 1;
    for (int i = 0; i + 1 < n_cand; ++i) {                      // ptp_utils.py:132-137
        float d = (act && lane > i) ? skp_dist(s_ly[i], s_lx[i], my_y, my_x) : -INFINITY;
        int j = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(d, o, 64); const int oj = __shfl_xor(j, o, 64);
            if (skp_better(od, oj, d, j)) { d = od; j = oj; }
        }
        if (d > best) { best = d; ba = i; bb = j; }
    }
    // greedy max-min: keep each candidate's distance to the chosen set incrementally (min is exact, so this
    // equals the reference's recomputation over all selected points)
    int nch = 2;
    int mine = (lane == 0) ? ba : bb;                           // lane q remembers the q-th chosen slot
    bool taken = act && (lane == ba || lane == bb);
    float dmin = fminf(skp_dist(my_y, my_x, __shfl(my_y, ba, 64), __shfl(my_x, ba, 64)),
                       skp_dist(my_y, my_x, __shfl(my_y, bb, 64), __shfl(my_x, bb, 64)));
    for (int it = 0; it < top_k - 2; ++it) {                    // ptp_utils.py:142-157
        float d = (act && !taken) ? dmin : -INFINITY;
        int j = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(d, o, 64); const int oj = __shfl_xor(j, o, 64);
            if (skp_better(od, oj, d, j)) { d = od; j = oj; }
        }
        if (d > -1.0f) {                                        // `this_min_dist > max_min_dist` starting from -1
            if (lane == nch) mine = j;
            if (lane == j) taken = true;
            dmin = fminf(dmin, skp_dist(my_y, my_x, __shfl(my_y, j, 64), __shfl(my_x, j, 64)));
            ++nch;
        }
    }
    const int last = __shfl(mine, nch - 1, 64);
    if (lane < top_k) sel_out[lane] = (int64_t)s_cand[lane < nch ? mine : last];
}

extern "C" int skp_select_tokens(const float* kl, const int32_t* argmax_t, int T, int R, int n_cand, int top_k,
                                 int64_t* cand, int64_t* sel, void* stream) {
    if (!kl || !argmax_t || !cand || !sel || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (T > SKP_SEL_MAXT || n_cand > SKP_SEL_MAXC || n_cand > T || top_k < 2 || top_k > n_cand) return SKP_E_RANGE;
    hipLaunchKernelGGL(skp_select_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kl, argmax_t, T, R, n_cand,
                       top_k, cand, sel);
    return skp_launch_status();
}

// n images at once: kl, argmax_t [n,T]; cand [n,n_cand], sel [n,top_k] -- one workgroup per image, the same code path
extern "C" int skp_select_tokens_batched(const float* kl, const int32_t* argmax_t, int n, int T, int R, int n_cand, int top_k,
                                         int64_t* cand, int64_t* sel, void* stream) {
    if (!kl || !argmax_t || !cand || !sel || n <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (T > SKP_SEL_MAXT || n_cand > SKP_SEL_MAXC || n_cand > T || top_k < 2 || top_k > n_cand) return SKP_E_RANGE;
    hipLaunchKernelGGL(skp_select_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, kl, argmax_t, T, R, n_cand, top_k, cand, sel);
    return skp_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Losses (optimize.py:157-206) with their unit gradients, fused: one pass over the K selected rows.
// grid = (ceil(R*R/1024), K); each thread handles 4 strided pixels.
// ---------------------------------------------------------------------------------------------
struct LossArgs {
    int K, T, R, S, nchunk;
    float sigma;
    float th[6];                                                // inverse affine, row-major 2x3
    const float* th_dev;                                        // ... or, if not null, read from device memory (captured steps)
};
// the argument block with the inverse affine in place (a captured hipGraph replays the same launch for a new affine every step:
// the six numbers then come from a device buffer the host refreshes before the replay)
__device__ __forceinline__ LossArgs skp_loss_args(const LossArgs& in) {
    LossArgs a = in;
    if (in.th_dev) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.th[i] = in.th_dev[i];
    }
    return a;
}

// Bilinear sample position of output pixel (col,row) in the transformed map: affine_grid + grid_sample with
// align_corners=False (invertable_transform.py:72-92).  Explicitly rounded ops: the loss kernel and the gradient gather
// below must agree bit for bit on the footprint of every pixel.
struct SkpBilin { int x0, y0; float wx0, wx1, wy0, wy1; };
__device__ __forceinline__ SkpBilin skp_bilin(const LossArgs& a, int col, int row) {
    const float Rf = (float)a.R;
    const float xs = __fsub_rn(__fdiv_rn(__fadd_rn(__fmul_rn(2.0f, (float)col), 1.0f), Rf), 1.0f);
    const float ys = __fsub_rn(__fdiv_rn(__fadd_rn(__fmul_rn(2.0f, (float)row), 1.0f), Rf), 1.0f);
    const float gx = __fadd_rn(__fadd_rn(__fmul_rn(a.th[0], xs), __fmul_rn(a.th[1], ys)), a.th[2]);
    const float gy = __fadd_rn(__fadd_rn(__fmul_rn(a.th[3], xs), __fmul_rn(a.th[4], ys)), a.th[5]);
    const float fx = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), Rf), 1.0f), 0.5f);
    const float fy = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), Rf), 1.0f), 0.5f);
    const float x0f = floorf(fx), y0f = floorf(fy);
    SkpBilin b;
    // clamp before the int conversion so a diverged affine cannot overflow it (every clamped value is out of range)
    b.x0 = (int)fminf(fmaxf(x0f, -2.0f), Rf + 1.0f);
    b.y0 = (int)fminf(fmaxf(y0f, -2.0f), Rf + 1.0f);
    b.wx1 = __fsub_rn(fx, x0f); b.wx0 = __fsub_rn(1.0f, b.wx1);
    b.wy1 = __fsub_rn(fy, y0f); b.wy0 = __fsub_rn(1.0f, b.wy1);
    return b;
}

__global__ __launch_bounds__(256) void skp_losses_kernel(const float* __restrict__ M, const float* __restrict__ Mt,
                                                         const int64_t* __restrict__ sel,
                                                         const int32_t* __restrict__ argmax, LossArgs a_in,
                                                         float* __restrict__ partial, float* __restrict__ g_sharp,
                                                         float* __restrict__ g_eq_a) {
    __shared__ float red[4];
    const LossArgs a = skp_loss_args(a_in);
    const int k = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int R = a.R, RR = R * R;
    const int tok = (int)sel[k];
    const float* m = M + (size_t)tok * RR;
    const float* mt = Mt + (size_t)tok * RR;
    float pcy[SKP_MAX_SUBJECTS], pcx[SKP_MAX_SUBJECTS];
    for (int j = 0; j < a.S; ++j) {                             // optimize.py:168 + optimize_token.py:210
        const int flat = argmax[j * a.T + tok];
        pcy[j] = __fmul_rn(__fdiv_rn((float)(flat / R) + 0.5f, (float)R), (float)R);
        pcx[j] = __fmul_rn(__fdiv_rn((float)(flat % R) + 0.5f, (float)R), (float)R);
    }
    const float two_sig2 = 2.0f * a.sigma * a.sigma;
    const float inv_S = 1.0f / (float)a.S;
    const float gscale = 2.0f / ((float)a.K * (float)RR);       // d mean((a-b)^2) / d a = 2 (a-b) / n
    float ss = 0.f, se = 0.f;
    for (int u = 0; u < 4; ++u) {
        const int p = chunk * 1024 + u * 256 + tid;
        if (p >= RR) break;
        const int row = p / R, col = p - row * R;
        const float v = m[p];
        // --- sharpening: gaussian target at the token's own arg-maxima
        float g = 0.f;
        for (int j = 0; j < a.S; ++j) {
            const float dx = (float)col + 0.5f - pcx[j], dy = (float)row + 0.5f - pcy[j];
            g += expf(-(dx * dx + dy * dy) / two_sig2);
        }
        const float ds = v - g * inv_S;
        ss += ds * ds;
        g_sharp[(size_t)k * RR + p] = gscale * ds;
        // --- equivariance: bilinear sample of the transformed map through the inverse affine
        const SkpBilin bl = skp_bilin(a, col, row);
        const int x0 = bl.x0, y0 = bl.y0, x1 = x0 + 1, y1 = y0 + 1;
        const bool vx0 = x0 >= 0 && x0 < R, vx1 = x1 >= 0 && x1 < R, vy0 = y0 >= 0 && y0 < R, vy1 = y1 >= 0 && y1 < R;
        float sval = 0.f;
        if (vy0 && vx0) sval += mt[y0 * R + x0] * __fmul_rn(bl.wy0, bl.wx0);
        if (vy0 && vx1) sval += mt[y0 * R + x1] * __fmul_rn(bl.wy0, bl.wx1);
        if (vy1 && vx0) sval += mt[y1 * R + x0] * __fmul_rn(bl.wy1, bl.wx0);
        if (vy1 && vx1) sval += mt[y1 * R + x1] * __fmul_rn(bl.wy1, bl.wx1);
        const float de = v - sval;
        se += de * de;
        g_eq_a[(size_t)k * RR + p] = gscale * de;
    }
    ss = skp_block_sum_256(ss, red);
    se = skp_block_sum_256(se, red);
    if (tid == 0) {
        partial[(size_t)(0 * a.K + k) * a.nchunk + chunk] = ss;
        partial[(size_t)(1 * a.K + k) * a.nchunk + chunk] = se;
    }
}

// d equiv / d Mt[sel[k]] as a GATHER (no atomics => bit-reproducible): source pixel (sx,sy) of the transformed map
// collects -g_eq_a[p] * w(p -> s) from every output pixel p whose bilinear footprint contains it.  Those p lie in the
// pre-image of the box (sx-1,sx+1) x (sy-1,sy+1) under the affine pixel map; its bounding box (+1 pixel of slack for
// rounding) is scanned in row-major order and each candidate's footprint is recomputed with skp_bilin.
__global__ __launch_bounds__(256) void skp_equiv_grad_kernel(LossArgs a_in, const float* __restrict__ g_eq_a,
                                                            float* __restrict__ g_eq_b) {
    const LossArgs a = skp_loss_args(a_in);
    const int k = blockIdx.y, R = a.R, RR = R * R;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= RR) return;
    const int sy = s / R, sx = s - sy * R;
    // pixel map: f = A (c + 0.5) + o,  A = [[th0,th1],[th3,th4]],  o = R/2 (t + 1 - rowsum(A)) - 0.5
    const float a00 = a.th[0], a01 = a.th[1], a10 = a.th[3], a11 = a.th[4];
    const float ox = 0.5f * (float)R * (a.th[2] + 1.0f - a00 - a01) - 0.5f;
    const float oy = 0.5f * (float)R * (a.th[5] + 1.0f - a10 - a11) - 0.5f;
    const float det = a00 * a11 - a01 * a10;
    int c0 = 0, c1 = R - 1, r0 = 0, r1 = R - 1;
    if (fabsf(det) > 1e-12f) {
        const float i00 = a11 / det, i01 = -a01 / det, i10 = -a10 / det, i11 = a00 / det;
        float cmin = INFINITY, cmax = -INFINITY, rmin = INFINITY, rmax = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float fx = (float)sx + ((q & 1) ? 1.0f : -1.0f) - ox, fy = (float)sy + ((q & 2) ? 1.0f : -1.0f) - oy;
            const float c = i00 * fx + i01 * fy - 0.5f, r = i10 * fx + i11 * fy - 0.5f;
            cmin = fminf(cmin, c); cmax = fmaxf(cmax, c); rmin = fminf(rmin, r); rmax = fmaxf(rmax, r);
        }
        if (cmin == cmin && cmax == cmax && rmin == rmin && rmax == rmax) {     // NaN => scan everything
            c0 = (int)fmaxf(floorf(cmin) - 1.0f, 0.0f); c1 = (int)fminf(ceilf(cmax) + 1.0f, (float)(R - 1));
            r0 = (int)fmaxf(floorf(rmin) - 1.0f, 0.0f); r1 = (int)fminf(ceilf(rmax) + 1.0f, (float)(R - 1));
        }
    }
    const float* ga = g_eq_a + (size_t)k * RR;
    float acc = 0.f;
    for (int row = r0; row <= r1; ++row)
        for (int col = c0; col <= c1; ++col) {
            const SkpBilin bl = skp_bilin(a, col, row);
            const float wx = (bl.x0 == sx) ? bl.wx0 : ((bl.x0 + 1 == sx) ? bl.wx1 : 0.f);
            const float wy = (bl.y0 == sy) ? bl.wy0 : ((bl.y0 + 1 == sy) ? bl.wy1 : 0.f);
            const bool hit = (bl.x0 == sx || bl.x0 + 1 == sx) && (bl.y0 == sy || bl.y0 + 1 == sy);
            if (hit) acc -= ga[row * R + col] * __fmul_rn(wy, wx);
        }
    g_eq_b[(size_t)k * RR + s] = acc;
}

static int losses_run(const float* M, const float* Mt, const int64_t* sel, int K, int T, int R, const int32_t* argmax, int num_subjects,
                      float sigma, const float* theta_inv_host, const float* theta_inv_dev, float* partial, float* g_sharp,
                      float* g_eq_a, float* g_eq_b, void* stream) {
    if (!M || !Mt || !sel || !argmax || (!theta_inv_host && !theta_inv_dev) || !partial || !g_sharp || !g_eq_a || !g_eq_b) return SKP_E_BADARG;
    if (K <= 0 || T <= 0 || R <= 0) return SKP_E_BADARG;
    if (num_subjects < 1 || num_subjects > SKP_MAX_SUBJECTS || K > 65535) return SKP_E_RANGE;
    LossArgs a{};
    a.K = K; a.T = T; a.R = R; a.S = num_subjects; a.sigma = sigma;
    a.nchunk = (R * R + 1023) / 1024;
    a.th_dev = theta_inv_dev;
    if (theta_inv_host)
        for (int i = 0; i < 6; ++i) a.th[i] = theta_inv_host[i];
    hipLaunchKernelGGL(skp_losses_kernel, dim3(a.nchunk, K), dim3(256), 0, (hipStream_t)stream, M, Mt, sel, argmax, a,
                       partial, g_sharp, g_eq_a);
    int rc = skp_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(skp_equiv_grad_kernel, dim3((R * R + 255) / 256, K), dim3(256), 0, (hipStream_t)stream, a, g_eq_a,
                       g_eq_b);
    return skp_launch_status();
}

extern "C" int skp_losses_fwd_f32(const float* M, const float* Mt, const int64_t* sel, int K, int T, int R,
                                  const int32_t* argmax, int num_subjects, float sigma, const float* theta_inv,
                                  float* partial, float* g_sharp, float* g_eq_a, float* g_eq_b, void* stream) {
    return losses_run(M, Mt, sel, K, T, R, argmax, num_subjects, sigma, theta_inv, nullptr, partial, g_sharp, g_eq_a, g_eq_b, stream);
}

// the same with the inverse affine (6 floats) in DEVICE memory, read by the kernels at run time: the launch can be captured in a
// hipGraph and replayed for a new augmentation every step
extern "C" int skp_losses_fwd_dev_f32(const float* M, const float* Mt, const int64_t* sel, int K, int T, int R,
                                      const int32_t* argmax, int num_subjects, float sigma, const float* theta_inv_dev,
                                      float* partial, float* g_sharp, float* g_eq_a, float* g_eq_b, void* stream) {
    if (!theta_inv_dev) return SKP_E_BADARG;
    return losses_run(M, Mt, sel, K, T, R, argmax, num_subjects, sigma, nullptr, theta_inv_dev, partial, g_sharp, g_eq_a, g_eq_b, stream);
}

// dst[sel[k], :] += a*x[k,:] + b*y[k,:]
__global__ __launch_bounds__(256) void skp_rows_axpy_kernel(float* __restrict__ dst, const int64_t* __restrict__ sel,
                                                            int64_t n, const float* __restrict__ x,
                                                            const float* __restrict__ a, const float* __restrict__ y,
                                                            const float* __restrict__ b) {
    const int k = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = a[0] * x[(size_t)k * n + i];
    if (y) v += b[0] * y[(size_t)k * n + i];
    dst[(size_t)sel[k] * n + i] += v;
}

extern "C" int skp_rows_axpy_f32(float* dst, const int64_t* sel, int K, int64_t n, const float* x, const float* a,
                                 const float* y, const float* b, void* stream) {
    if (!dst || !sel || !x || !a || K <= 0 || n <= 0 || (y && !b)) return SKP_E_BADARG;
    if (K > 65535) return SKP_E_RANGE;
    hipLaunchKernelGGL(skp_rows_axpy_kernel, dim3((unsigned)((n + 255) / 256), K), dim3(256), 0, (hipStream_t)stream,
                       dst, sel, n, x, a, y, b);
    return skp_launch_status();
}

extern "C" int skp_abi_version(void) { return 39; }

// ---- developer overrides (include/skp.h: skp_tune_set) ----
static int g_tune[SKP_TUNE_COUNT] = {0};
static const char* const g_tune_names[SKP_TUNE_COUNT] = {"wino_split", "wino_raw_max_tiles", "map_bands", "fa2_two_kernel_bwd", "gn_fold_max_cout", "cross_attn_ts"};
int skp_tune(int key) { return (key >= 0 && key < SKP_TUNE_COUNT) ? g_tune[key] : 0; }
extern "C" int skp_tune_set(const char* key, int value) {
    if (!key || value < 0) return SKP_E_BADARG;
    for (int i = 0; i < SKP_TUNE_COUNT; ++i)
        if (!strcmp(key, g_tune_names[i])) { g_tune[i] = value; return 0; }
    return SKP_E_RANGE;
}
extern "C" int skp_tune_get(const char* key) {
    if (!key) return SKP_E_BADARG;
    for (int i = 0; i < SKP_TUNE_COUNT; ++i)
        if (!strcmp(key, g_tune_names[i])) return g_tune[i];
    return SKP_E_RANGE;
}
