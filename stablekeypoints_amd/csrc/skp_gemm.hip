// Batched NT GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): exact fp32 fma chain,
// 157 TF/s peak on MI355X.  Replaces the q.k^T contractions of ptp_utils.py:493,534 and their
// backward products.  One wave owns one 32x32 output tile; operands are read straight from
// global/L2 (the matrices of this path are <= a few MB and L2 resident).
#include "skp_common.h"

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K, Z1;
    int Nload;                 // rows of B that exist (n >= Nload reads as 0; still stored when n < N)
    int ksplit, kchunk;        // split-K: blockIdx.z = (z0*Z1+z1)*ksplit + ks handles k in [ks*kchunk, ...)
    int64_t part_stride;       // floats between the ksplit partial copies of C (0 when ksplit == 1)
    int64_t sa0, sa1, sam, sak;
    int64_t sb0, sb1, sbn, sbk;
    int64_t sc0, sc1, scm;
    float alpha;
};

// K is walked 8 at a time; lane l supplies k = k0 + 4*(l>>5) + m for MFMA m = 0..3, so an operand whose k
// axis is contiguous (sXk == 1) is fetched with ONE float4 load per lane per 4 MFMAs.
template <bool AV, bool BV>
__global__ __launch_bounds__(256) void skp_gemm_nt_kernel(GemmArgs g) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, hi = lane >> 5;
    const int n0 = (blockIdx.x * 2 + (wave & 1)) * 32;
    const int m0 = (blockIdx.y * 2 + (wave >> 1)) * 32;
    if (m0 >= g.M || n0 >= g.N) return;                       // wave-uniform
    const int zz = blockIdx.z / g.ksplit, ks = blockIdx.z - zz * g.ksplit;
    const int z0 = zz / g.Z1, z1 = zz - z0 * g.Z1;
    const bool mv = (m0 + i) < g.M, nv = (n0 + i) < g.Nload, ns = (n0 + i) < g.N;
    const float* Ap = g.A + z0 * g.sa0 + z1 * g.sa1 + (int64_t)(mv ? m0 + i : 0) * g.sam;
    const float* Bp = g.B + z0 * g.sb0 + z1 * g.sb1 + (int64_t)(nv ? n0 + i : 0) * g.sbn;
    f32x16 acc = {0};
    const int kbeg = ks * g.kchunk;
    const int K = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;     // kchunk is a multiple of 8
    int k = kbeg;
    auto load8 = [&](int k8, f32x4& a, f32x4& b) {
        const int kk = k8 + 4 * hi;
        if (AV) a = *(const f32x4*)(Ap + kk);
        else { a[0] = Ap[(int64_t)kk * g.sak]; a[1] = Ap[(int64_t)(kk + 1) * g.sak]; a[2] = Ap[(int64_t)(kk + 2) * g.sak]; a[3] = Ap[(int64_t)(kk + 3) * g.sak]; }
        if (BV) b = *(const f32x4*)(Bp + kk);
        else { b[0] = Bp[(int64_t)kk * g.sbk]; b[1] = Bp[(int64_t)(kk + 1) * g.sbk]; b[2] = Bp[(int64_t)(kk + 2) * g.sbk]; b[3] = Bp[(int64_t)(kk + 3) * g.sbk]; }
    };
    // 32 k per step, the NEXT step's operands requested before this step's 16 MFMAs: the contraction is a chain of L2 / HBM round
    // trips otherwise (one per 8 k: the dK | dV reductions of the cross-attention backward, K = 256 per split, were 21-82 us launches
    // for 10 us of traffic)
    constexpr int U = 4;
    if (k + 8 * U <= K) {
        f32x4 a[U], b[U], an[U], bn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) load8(k + 8 * u, a[u], b[u]);
        for (; k + 8 * U <= K; k += 8 * U) {
            const bool more = k + 16 * U <= K;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) load8(k + 8 * U + 8 * u, an[u], bn[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mv ? a[u][m] : 0.f, nv ? b[u][m] : 0.f, acc, 0, 0, 0);
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) { a[u] = an[u]; b[u] = bn[u]; }
            }
        }
    }
    for (; k + 8 <= K; k += 8) {
        f32x4 a, b;
        load8(k, a, b);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mv ? a[u] : 0.f, nv ? b[u] : 0.f, acc, 0, 0, 0);
    }
    for (; k < K; k += 2) {                                    // tail: k-slot = k + hi
        const bool kv = (k + hi) < K;
        const float a = (kv && mv) ? Ap[(int64_t)(k + hi) * g.sak] : 0.f;
        const float b = (kv && nv) ? Bp[(int64_t)(k + hi) * g.sbk] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // C/D layout: col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
    float* Cp = g.C + (int64_t)ks * g.part_stride + z0 * g.sc0 + z1 * g.sc1;
    if (ns) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < g.M) Cp[(int64_t)m * g.scm + n0 + i] = g.alpha * acc[r];
        }
    }
}

static void gemm_dispatch(const GemmArgs& g, dim3 grid, hipStream_t st) {
    // float4 path needs a unit k stride and 16-byte aligned rows (base pointer, row and batch strides multiples of 4)
    auto vec_ok = [](const float* p, int64_t s0, int64_t s1, int64_t sr, int64_t sk) {
        return sk == 1 && ((uintptr_t)p & 15) == 0 && (s0 & 3) == 0 && (s1 & 3) == 0 && (sr & 3) == 0;
    };
    const bool av = vec_ok(g.A, g.sa0, g.sa1, g.sam, g.sak) && (g.kchunk % 8 == 0);
    const bool bv = vec_ok(g.B, g.sb0, g.sb1, g.sbn, g.sbk) && (g.kchunk % 8 == 0);
    if (av && bv) hipLaunchKernelGGL((skp_gemm_nt_kernel<true, true>), grid, dim3(256), 0, st, g);
    else if (av) hipLaunchKernelGGL((skp_gemm_nt_kernel<true, false>), grid, dim3(256), 0, st, g);
    else if (bv) hipLaunchKernelGGL((skp_gemm_nt_kernel<false, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((skp_gemm_nt_kernel<false, false>), grid, dim3(256), 0, st, g);
}

static int gemm_launch(const float* A, const float* B, float* C, int M, int N, int Nload, int K, int Z0, int Z1,
                       int64_t sa0, int64_t sa1, int64_t sam, int64_t sak,
                       int64_t sb0, int64_t sb1, int64_t sbn, int64_t sbk,
                       int64_t sc0, int64_t sc1, int64_t scm, float alpha, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || Z0 <= 0 || Z1 <= 0) return SKP_E_BADARG;
    if ((int64_t)Z0 * Z1 > 65535) return SKP_E_RANGE;
    GemmArgs g{A, B, C, M, N, K, Z1, Nload, 1, (K + 7) & ~7, 0, sa0, sa1, sam, sak, sb0, sb1, sbn, sbk, sc0, sc1, scm, alpha};
    dim3 grid((N + 63) / 64, (M + 63) / 64, Z0 * Z1);
    gemm_dispatch(g, grid, (hipStream_t)stream);
    return skp_launch_status();
}

// out[i] = sum_{ks} part[ks*stride + i]   (fixed order => deterministic)
__global__ __launch_bounds__(256) void skp_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                int64_t n, int ksplit, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += part[k * stride + i];
    out[i] = s;
}

// Split-K form for long contractions over few output tiles (the dK / dV reductions over up to 4096 queries):
// `ksplit` partial products go to `partial` (ksplit * c_elems floats), then a fixed-order reduction writes the dense
// output C (c_elems floats, laid out by sc0/sc1/scm).  Deterministic, no atomics.
int skp_gemm_nt_splitk(const float* A, const float* B, float* C, float* partial, int64_t c_elems, int ksplit,
                       int M, int N, int K, int Z0, int Z1,
                       int64_t sa0, int64_t sa1, int64_t sam, int64_t sak,
                       int64_t sb0, int64_t sb1, int64_t sbn, int64_t sbk,
                       int64_t sc0, int64_t sc1, int64_t scm, float alpha, void* stream) {
    if (ksplit <= 1 || !partial)
        return gemm_launch(A, B, C, M, N, N, K, Z0, Z1, sa0, sa1, sam, sak, sb0, sb1, sbn, sbk, sc0, sc1, scm, alpha, stream);
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || Z0 <= 0 || Z1 <= 0) return SKP_E_BADARG;
    if ((int64_t)Z0 * Z1 * ksplit > 65535) return SKP_E_RANGE;
    const int kchunk = (((K + ksplit - 1) / ksplit) + 7) & ~7;
    GemmArgs g{A, B, partial, M, N, K, Z1, N, ksplit, kchunk, c_elems, sa0, sa1, sam, sak, sb0, sb1, sbn, sbk, sc0, sc1, scm, alpha};
    dim3 grid((N + 63) / 64, (M + 63) / 64, Z0 * Z1 * ksplit);
    gemm_dispatch(g, grid, (hipStream_t)stream);
    int rc = skp_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(skp_splitk_reduce_kernel, dim3((unsigned)((c_elems + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)partial, C, c_elems, ksplit, c_elems);
    return skp_launch_status();
}

extern "C" int skp_gemm_nt_f32(const float* A, const float* B, float* C, int M, int N, int K, int Z0, int Z1,
                               int64_t sa0, int64_t sa1, int64_t sam, int64_t sak,
                               int64_t sb0, int64_t sb1, int64_t sbn, int64_t sbk,
                               int64_t sc0, int64_t sc1, int64_t scm, float alpha, void* stream) {
    return gemm_launch(A, B, C, M, N, N, K, Z0, Z1, sa0, sa1, sam, sak, sb0, sb1, sbn, sbk, sc0, sc1, scm, alpha, stream);
}

extern "C" int skp_qk_logits_f32(const float* q, const float* k, float* S, int B, int Bk, int H, int T, int s2,
                                 int d, float scale, void* stream) {
    if (!q || !k || !S || B <= 0 || H <= 0 || T <= 0 || s2 <= 0 || d <= 0) return SKP_E_BADARG;
    if (Bk != 1 && Bk != B) return SKP_E_BADARG;
    const int64_t C = (int64_t)H * d;
    const int NT = ((T + 15) / 16) * 16;
    // S[b,h][p,t] = alpha * sum_c q[b,p,h*d+c] * k[bk,t,h*d+c]; token-contiguous rows of NT floats, pad columns = 0
    return gemm_launch(q, k, S, s2, NT, T, d, B, H,
                       (int64_t)s2 * C, d, C, 1,
                       Bk == 1 ? 0 : (int64_t)T * C, d, C, 1,
                       (int64_t)H * s2 * NT, (int64_t)s2 * NT, NT, scale * SKP_LOG2E, stream);
}
