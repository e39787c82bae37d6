// Backward of the fused up-res attention map, "column sweep" form (round 4): NO dV staging, NO band partials.
//
// Reference path replaced: autograd through ptp_utils.py:513-538 (bicubic x -> R^2, to_q, einsum, softmax) and
// optimize.py:27-79 (stack / mean), for the gradient the losses produce: non-zero on the K selected token rows only
// (optimize.py:395-414) -- the input of skp_attn_map_bwd_sparse_f32, whose T <= 128 route this is.
//
// The two adjoints of the separable bicubic are applied in the order that shrinks the data first:
//   * lane = up-res column x; a workgroup (R threads) owns one (batch row, layer, head, 8-token chunk) and sweeps ALL R rows
//     top to bottom.  f_t = p_t (g_t - sum_k p_k g_k) with p_t = exp2(bicubic(S)_t - lse) is formed per row and its VERTICAL
//     adjoint accumulated in registers: acc[row & 3][token] += wy * f (a row's four taps land in four consecutive low-res
//     rows; rows come in groups of K2 = k/2 that share their taps, k = R / s).
//   * a low-res row is COMPLETE once the sweep has passed its last contributing group: only then the HORIZONTAL adjoint is
//     applied -- s times per image instead of R times: lanes that share their four tap columns (aligned groups of k/2
//     pixels) pre-reduce with DPP, the group sums cross LDS once, a column's <= 9 contributions are added in list order (no
//     atomics: bit-reproducible) and the finished dS row goes straight to its place.  Out-of-image tap rows are folded onto
//     the clamped border rows in the register window (upsample_bicubic2d clamps tap indices on access).
// The vertical interpolation the forward needs (V phase) is done per group for the group's K2 rows only (3 KB of LDS,
// double-buffered, the next group's loads in flight under the current group's rows).
//
// Launches (round 5): dot = sum_k p_k g_k / (L H) per (layer, head, pixel) by a row-parallel kernel (skp_map_bwd_dot_kernel: the
// K selected tokens only, no sweep), then ONE sweep launch over the natural 8-token chunks: f_t = p_t (g_t / (L H) - dot) for a
// selected token -- its gradient row is loaded through a descriptor offset that lies past the buffer for every other token, so
// the row code is branch-free --, f_t = -p_t dot otherwise; chunks without a selected token run the plain row code (wave-uniform
// choice per group).  dS is written once, nothing is staged.  The two-sweep form it replaces (selected tokens first: dot + their
// part in a side buffer, then the natural chunks) stays behind SKP_MAP_COL_DOT=0.
//
// Shapes served (skp_attn_map_bwd_col_ok): R in {128, 256}, every layer R = k s with k in {4, 8} and s >= 8, T <= 1024, K <= 16
// -- the SD-1.x path at feature_upsample_res 128 / 256; everything else keeps the other routes.
#include "skp_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int CL_TC = 8;             // tokens per chunk
constexpr int CL_TS = CL_TC + 4;     // token stride of the V rows (16-byte aligned quads, conflict-free column gathers)
constexpr int CL_PS = CL_TC + 1;     // token stride of the exchange buffer
constexpr int CL_LUSE = 10;          // list entries read per low-res column (<= 9 real + padding pointing at a zero slot)
constexpr int CL_KMAX = 16;
// Prefetch of the next group's lse / dot and V-phase loads under the current group's rows: costs 8 + 16 registers.  The natural
// pass cannot afford them (measured: 601 us with the lse prefetch against 537 without at 168 registers) and the selected-token
// pass (512 two-wave workgroups, one wave per SIMD) does not gain from them (394 us either way on one box): both off.
#ifndef CL_PF_SEL
#define CL_PF_SEL 0
#endif
#ifndef CL_PF_NAT
#define CL_PF_NAT 0
#endif
// Waves per SIMD the register budgets are held to.  Measured (B = 8, T = 77, R = 128; us for the whole backward): natural pass at
// 4 / 3 / 2 waves per SIMD 942 / 535 / 381 -- below 240 registers the allocator spills the accumulator window inside the row
// loop (300-450 bytes of scratch per lane, 0.8 GB of spill traffic per launch) and every reload waits on the memory pipeline.
#ifndef CL_WAVES_NAT
#define CL_WAVES_NAT 2
#endif
#ifndef CL_WAVES_SEL
#define CL_WAVES_SEL 3
#endif

struct ColLayer {
    const float* S;                  // [B,H,s*s,ldt] logits (pre-multiplied by scale*log2 e)
    float* dS;                       // [B,H,s*s,ldt]
    float* Psel;                     // [B,H,s*s,CL_KMAX] the selected tokens' +p g part
    int l;                           // layer index (lse / dot)
    int s;                           // the layer's side (dot kernel)
};
struct ColArgs {
    ColLayer ly[SKP_MAX_LAYERS];
    int nl, nl4;                     // layers in the launch; the first nl4 have k = R / s = 8 (quads), the rest k = 4 (pairs)
    const int64_t* sel;              // [B,K]
    const float* G;                  // [B,K,R,R]
    const float* lse;                // [B,L*H,R*R] (log2 domain)
    float* dotp;                     // [nsel][B,L*H,R*R]
    int L, B, H, T, K, ldt, NT, nch, nsel;
    float inv_lh;
};

template <int K> __device__ __forceinline__ float cl_quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float cl_pair_swap(float v) {      // value of the other lane of an aligned lane pair
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

// K2 = k/2 rows per group = lanes per tap-sharing group (4: quads, 2: pairs); R = map side = threads; SEL: selected-token launch.
// INL (natural launch only): the selected tokens of a chunk are handled INSIDE the natural chunk -- dot comes complete from
// skp_map_bwd_dot_kernel, a selected token's f is p (g / (L H) - dot) with its gradient row loaded under a wave-uniform branch --
// so there is no selected-token sweep, no side buffer and no second dot plane.
template <int K2, int R, bool SEL, bool INL = false>
__device__ __forceinline__ void col_body(const ColArgs& a, float* smem) {
    static_assert(!(SEL && INL), "INL is a form of the natural launch");
    constexpr int RR = R * R, s = R / (2 * K2), E = (K2 == 2) ? 2 : 1;
    constexpr int VG = K2 * s * CL_TS;           // floats of one V buffer (K2 rows)
    constexpr int XB = (R * E + 1) * CL_PS;      // floats of one exchange buffer (+1: the zero slot list padding points at)
    constexpr int GO = (s * CL_TC + R - 1) / R;  // gather outputs per thread
    constexpr int VI = SEL ? (K2 * s * CL_TC) / R : 1;   // V-phase items per thread: natural K2*s*2 quads == R items
    static_assert(K2 * s * 2 == R, "one natural V-phase item per thread");
    const int tid = threadIdx.x, H = a.H;
    const int b = blockIdx.x % a.B;              // batch row fastest: workgroup id % 8 (XCD) == b % 8
    int rest = blockIdx.x / a.B;
    const int ch = rest % a.nch;
    rest /= a.nch;
    const int h = rest % H, li = rest / H;
    const ColLayer ly = a.ly[li];
    const float ratio = (float)s / (float)R;
    const int k0 = ch * CL_TC;                   // SEL: first selected slot; natural: first token

    float* Vg = smem;                            // [2][K2*s][CL_TS]
    float* tabw = smem + 2 * VG;                 // [R][4] vertical tap weights of every up-res row, read as f32x4: it sits
    static_assert((2 * VG) % 4 == 0, "tabw must start on a 16-byte boundary (ds_read_b128)");   // BEFORE the odd-sized xb
    float* xb = tabw + R * 4;                    // [2][R*E + 1][CL_PS]
    int* lst = (int*)(xb + 2 * XB);              // [s][CL_LUSE]
    int* selk = lst + s * CL_LUSE;               // [CL_KMAX]

    {   // tables
        int cy[4]; float wy[4];
        skp_cubic_taps(tid, ratio, s, cy, wy);
#pragma unroll
        for (int j = 0; j < 4; ++j) tabw[tid * 4 + j] = wy[j];
    }
    if (tid < s) {       // exchange entries whose tap column is `tid`, ascending (fixed summation order)
        int n = 0;
        for (int g = 0; g < R / K2; ++g) {
            const int fl = (g - 1) >> 1;         // floor(src_x) of the group's pixels
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int c = fl - 1 + i;
                c = c < 0 ? 0 : (c > s - 1 ? s - 1 : c);
                if (c == tid && n < CL_LUSE) lst[tid * CL_LUSE + n++] = (K2 == 2) ? 2 * (2 * g + (i & 1)) + (i >> 1) : 4 * g + i;
            }
        }
        for (; n < CL_LUSE; ++n) lst[tid * CL_LUSE + n] = R * E;
    }
    if (tid < CL_KMAX) selk[tid] = tid < a.K ? (int)a.sel[(size_t)b * a.K + tid] : 0;
    if (tid < 2 * CL_PS) xb[(tid / CL_PS) * XB + R * E * CL_PS + tid % CL_PS] = 0.f;       // the zero slots

    int cx[4]; float wx[4];
    skp_cubic_taps(tid, ratio, s, cx, wx);
    f32x2 wx2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { cx[i] *= CL_TS; wx2[i] = f32x2{wx[i], wx[i]}; }
    float wq[4];         // pre-reduction weights: what each lane of my group gives to the tap column(s) I keep
    if (K2 == 4) {
        const int me = tid & 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c0 = cl_quad_bcast<0>(wx[k]), c1 = cl_quad_bcast<1>(wx[k]);
            const float c2 = cl_quad_bcast<2>(wx[k]), c3 = cl_quad_bcast<3>(wx[k]);
            if (me == k) { wq[0] = c0; wq[1] = c1; wq[2] = c2; wq[3] = c3; }
        }
    } else {             // pairs: I keep taps me and me + 2; wq = {own tap me, partner tap me, own tap me+2, partner tap me+2}
        const int me = tid & 1;
        const float p0 = cl_pair_swap(wx[0]), p1 = cl_pair_swap(wx[1]), p2 = cl_pair_swap(wx[2]), p3 = cl_pair_swap(wx[3]);
        wq[0] = me ? wx[1] : wx[0]; wq[1] = me ? p1 : p0;
        wq[2] = me ? wx[3] : wx[2]; wq[3] = me ? p3 : p2;
    }

    const int lh = ly.l * H + h;
    // every read goes through a buffer descriptor (wave-uniform base in SGPRs, 32-bit lane offset, scalar row offset): with
    // 64-bit pointers the per-row addresses of the fourteen inlined groups were kept in VGPR pairs and spilled
    const size_t plane = ((size_t)b * a.L * H + lh) * RR, dot_plane = (size_t)a.B * a.L * H * RR;
    const i32x4 srs = skp_make_rsrc(ly.S + ((size_t)(b * H + h) * s * s) * a.ldt, (unsigned)(s * s) * a.ldt * 4u);
    const i32x4 lrs = skp_make_rsrc(a.lse + plane, RR * 4u);
    const i32x4 drs0 = skp_make_rsrc(a.dotp + plane + (SEL ? (size_t)ch * dot_plane : 0), RR * 4u);
    const i32x4 drs1 = skp_make_rsrc(a.dotp + plane + dot_plane, RR * 4u);
    const i32x4 grs = skp_make_rsrc(a.G + (size_t)b * a.K * RR, (unsigned)a.K * RR * 4u);      // this batch row's K gradient rows
    const int dvo1 = (!INL && a.nsel > 1) ? tid * 4 : SKP_OOB;
    float* out_g = SEL ? ly.Psel + ((size_t)(b * H + h) * s * s) * CL_KMAX + k0
                       : ly.dS + ((size_t)(b * H + h) * s * s) * a.ldt + k0;
    const int out_ld = SEL ? CL_KMAX : a.ldt;
    __syncthreads();
    // natural launch: the selected slot (or -1) of each token this thread writes -- its +p g part is added on the way out
    int kslot[CL_TC];                            // INL: selected slot (or -1) of token k0 + t, wave-uniform
#pragma unroll
    for (int t = 0; t < CL_TC; ++t) {
        kslot[t] = -1;
        if (INL) {
            int ks = -1;
            for (int k = 0; k < a.K; ++k) ks = (selk[k] == k0 + t) ? k : ks;
            // byte offset of the token's gradient plane, or one past every plane (it goes into the VECTOR offset: the range check sees that one)
            kslot[t] = __builtin_amdgcn_readfirstlane(ks >= 0 ? ks * RR * 4 : CL_KMAX * RR * 4);
        }
    }
    bool has_sel = false;                        // wave-uniform
#pragma unroll
    for (int t = 0; t < CL_TC; ++t) has_sel = has_sel || (INL && kslot[t] != CL_KMAX * RR * 4);
    int ksel[GO];
#pragma unroll
    for (int oo = 0; oo < GO; ++oo) {
        ksel[oo] = -1;
        if (!SEL && !INL) {
            const int t = k0 + (tid + oo * R) % CL_TC;
            for (int k = 0; k < a.K; ++k) ksel[oo] = (selk[k] == t) ? k : ksel[oo];
        }
    }

    // ---- V phase of one group: issue the loads, combine, store ----
    float vraw[VI][4][SEL ? 1 : 4];
    auto v_issue = [&](int gg) {
        int fl = (gg - 1) >> 1;
#pragma unroll
        for (int u = 0; u < VI; ++u) {
            const int idx = tid + u * R;
            const int rc = SEL ? idx / CL_TC : idx >> 1;
            const int c = rc % s;
            const int tcol = SEL ? selk[k0 + idx % CL_TC] : k0 + 4 * (idx & 1);
            const int vo = (c * a.ldt + tcol) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int cy = fl - 1 + j;
                cy = cy < 0 ? 0 : (cy > s - 1 ? s - 1 : cy);
                const int so = cy * s * a.ldt * 4;                              // uniform
                if (SEL) vraw[u][j][0] = skp_buf_load_f32(srs, vo, so, 0);
                else {
                    const f32x4 v4 = skp_buf_load_f32x4(srs, vo, so, 0);
#pragma unroll
                    for (int e = 0; e < (SEL ? 1 : 4); ++e) vraw[u][j][e] = v4[e];
                }
            }
        }
    };
    auto v_store = [&](int gg, int buf) {
#pragma unroll
        for (int u = 0; u < VI; ++u) {
            const int idx = tid + u * R;
            const int rc = SEL ? idx / CL_TC : idx >> 1;
            const int r = rc / s;
            const f32x4 w = *(const f32x4*)(tabw + (gg * K2 + r) * 4);
            if (SEL) {
                const int kk = idx % CL_TC;
                float v = w[0] * vraw[u][0][0];
                v = fmaf(w[1], vraw[u][1][0], v); v = fmaf(w[2], vraw[u][2][0], v); v = fmaf(w[3], vraw[u][3][0], v);
                Vg[buf * VG + rc * CL_TS + kk] = (k0 + kk < a.K) ? v : 0.f;     // slots k >= K: finite filler, their g is 0
            } else {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < (SEL ? 1 : 4); ++e) {
                    float x = w[0] * vraw[u][0][e];
                    x = fmaf(w[1], vraw[u][1][e], x); x = fmaf(w[2], vraw[u][2][e], x); x = fmaf(w[3], vraw[u][3][e], x);
                    if (k0 + 4 * (idx & 1) + e >= a.T) x = -1.0e30f;            // pad tokens: probability 0 (finite: taps of both signs)
                    v[e] = x;
                }
                *(f32x4*)(Vg + buf * VG + rc * CL_TS + 4 * (idx & 1)) = v;
            }
        }
    };

    f32x2 acc[4][CL_TC / 2];                     // [low-res row & 3][token pair]: the register window of the vertical adjoint
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int t = 0; t < CL_TC / 2; ++t) acc[sl][t] = f32x2{0.f, 0.f};

    // the K2 rows of group gg (V rows in Vg[buf]); tap j accumulates into window slot j (rows floor(src_y) - 1 .. + 2)
    float lse_r[K2], nd_r[K2];                   // lse and -dot (or 1 / (L H)) of the current group's rows
    float lse_n[K2], d0_n[K2], d1_n[K2];         // ... of the next group, in flight while the current rows run (PF)
    auto ln_issue = [&](int gg) {
#pragma unroll
        for (int r = 0; r < K2; ++r) {
            const int so = (gg * K2 + r) * R * 4;
            lse_n[r] = skp_buf_load_f32(lrs, tid * 4, so, 0);
            if (!SEL) {   // second part only when K > 8: otherwise an out-of-range offset (returns 0), no branch in the sweep
                d0_n[r] = skp_buf_load_f32(drs0, tid * 4, so, 0);
                d1_n[r] = INL ? 0.f : skp_buf_load_f32(drs1, dvo1, so, 0);
            }
        }
    };
    auto ln_take = [&]() {
#pragma unroll
        for (int r = 0; r < K2; ++r) { lse_r[r] = lse_n[r]; nd_r[r] = SEL ? a.inv_lh : -(d0_n[r] + d1_n[r]); }
    };
    auto rows = [&](int gg, int buf, auto withg_c) {
        constexpr bool WITHG = decltype(withg_c)::value;      // INL: this chunk holds a selected token (else the plain row code)
        float gk[2][CL_TC];
        auto g_load = [&](int y, int slot) {             // slots k >= K lie past the descriptor: 0 (the range check sees the vector offset)
#pragma unroll
            for (int kk = 0; kk < CL_TC; ++kk) gk[slot][kk] = skp_buf_load_f32(grs, (tid + (k0 + kk) * RR) * 4, y * R * 4, 0);
        };
        if (SEL) g_load(gg * K2, 0);
#pragma unroll
        for (int r = 0; r < K2; ++r) {
            const int y = gg * K2 + r;
            if (SEL && r + 1 < K2) g_load(y + 1, (r + 1) & 1);       // the next row's gradient values fly under this row
            const f32x4 wyv = *(const f32x4*)(tabw + y * 4);
            const float* vrow = Vg + buf * VG + r * s * CL_TS;
            const float lse = lse_r[r], nd = nd_r[r];
            float dsum = 0.f;
            // INL: a token's multiplier of p is -dot, plus g / (L H) when it is a selected one (wave-uniform test; formed where
            // it is used: eight more live registers in this loop spill the accumulator window)
            // (branch-free: a token that is not selected reads past the descriptor, which returns 0 without a memory access)
            auto mult = [&](int t) -> float {
                return fmaf(skp_buf_load_f32(grs, tid * 4 + kslot[t], y * R * 4, 0), a.inv_lh, nd);
            };
#pragma unroll
            for (int q = 0; q < CL_TC / 4; ++q) {
                const f32x4 t0 = *(const f32x4*)(vrow + cx[0] + 4 * q);
                const f32x4 t1 = *(const f32x4*)(vrow + cx[1] + 4 * q);
                const f32x4 t2 = *(const f32x4*)(vrow + cx[2] + 4 * q);
                const f32x4 t3 = *(const f32x4*)(vrow + cx[3] + 4 * q);
                f32x2 lo = wx2[0] * f32x2{t0[0], t0[1]}, hi = wx2[0] * f32x2{t0[2], t0[3]};
                lo = wx2[1] * f32x2{t1[0], t1[1]} + lo; hi = wx2[1] * f32x2{t1[2], t1[3]} + hi;
                lo = wx2[2] * f32x2{t2[0], t2[1]} + lo; hi = wx2[2] * f32x2{t2[2], t2[3]} + hi;
                lo = wx2[3] * f32x2{t3[0], t3[1]} + lo; hi = wx2[3] * f32x2{t3[2], t3[3]} + hi;
                float f[4] = {__builtin_amdgcn_exp2f(lo[0] - lse), __builtin_amdgcn_exp2f(lo[1] - lse),
                              __builtin_amdgcn_exp2f(hi[0] - lse), __builtin_amdgcn_exp2f(hi[1] - lse)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (SEL) f[e] *= gk[r & 1][4 * q + e];
                    f[e] *= (INL && WITHG) ? mult(4 * q + e) : nd;
                    if (SEL) dsum += f[e];
                }
                const f32x2 f01 = {f[0], f[1]}, f23 = {f[2], f[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 w2 = {wyv[j], wyv[j]};
                    acc[j][2 * q] = w2 * f01 + acc[j][2 * q];
                    acc[j][2 * q + 1] = w2 * f23 + acc[j][2 * q + 1];
                }
            }
            if (SEL) skp_buf_store_f32(dsum, drs0, tid * 4, y * R * 4, 0);
            asm volatile("" ::: "memory");               // rows one after the other: without it the LDS reads of all K2 rows are
            __builtin_amdgcn_sched_barrier(0);           // hoisted to the top of the group (56 registers)
        }
    };

    int fpar = 0;                                // exchange buffer parity
    // low-res row `cy` is complete in window slot 0: horizontal adjoint, write the finished row
    auto flush = [&](int cy) {
        float* xw = xb + fpar * XB;
#pragma unroll
        for (int t = 0; t < CL_TC; ++t) {
            const float v = acc[0][t >> 1][t & 1];
            if (K2 == 4) {
                float pq = cl_quad_bcast<0>(v) * wq[0];
                pq = fmaf(cl_quad_bcast<1>(v), wq[1], pq);
                pq = fmaf(cl_quad_bcast<2>(v), wq[2], pq);
                pq = fmaf(cl_quad_bcast<3>(v), wq[3], pq);
                xw[tid * CL_PS + t] = pq;
            } else {
                const float vp = cl_pair_swap(v);
                xw[(2 * tid) * CL_PS + t] = fmaf(wq[1], vp, wq[0] * v);
                xw[(2 * tid + 1) * CL_PS + t] = fmaf(wq[3], vp, wq[2] * v);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int oo = 0; oo < GO; ++oo) {          // one output at a time: unrolled, two outputs' lists and values cost 40 registers
            const int o = tid + oo * R;
            if (o < s * CL_TC) {
                const int t = o % CL_TC, c = o / CL_TC;
                int gi[CL_LUSE];
#pragma unroll
                for (int e = 0; e < CL_LUSE; ++e) gi[e] = lst[c * CL_LUSE + e];      // the column's list: LDS, read together
                float v[CL_LUSE];
#pragma unroll
                for (int e = 0; e < CL_LUSE; ++e) v[e] = xw[gi[e] * CL_PS + t];      // independent LDS reads, all in flight
                float sum = v[0];
#pragma unroll
                for (int e = 1; e < CL_LUSE; ++e) sum += v[e];
                const size_t row = (size_t)cy * s + c;
                const int ks = GO == 1 ? ksel[0] : (oo ? ksel[GO - 1] : ksel[0]);
                if (!SEL && !INL && ks >= 0) sum += ly.Psel[((size_t)(b * H + h) * s * s + row) * CL_KMAX + ks];
                out_g[row * out_ld + t] = sum;
            }
        }
        fpar ^= 1;
    };
    auto rotate = [&]() {                         // the window moves down one low-res row: slot j <- slot j + 1, new slot 3 = 0
#pragma unroll
        for (int t = 0; t < CL_TC / 2; ++t) {
            acc[0][t] = acc[1][t]; acc[1][t] = acc[2][t]; acc[2][t] = acc[3][t]; acc[3][t] = f32x2{0.f, 0.f};
        }
    };
    auto fold = [&](int dst) {                    // slot 0 holds an out-of-image row above the map: it IS low-res row 0 (clamped taps)
#pragma unroll
        for (int t = 0; t < CL_TC / 2; ++t) {
            if (dst == 1) acc[1][t] += acc[0][t];
            else acc[2][t] += acc[0][t];
        }
    };

    // one group: sweep its rows, then build the next group's V rows (loads + combine back to back: holding the loads in
    // registers across the rows made the allocator spill them, with a wait for every load; the ~8 resident workgroups of a CU
    // cover each other's latency instead), barrier
    constexpr bool PF = SEL ? (CL_PF_SEL != 0) : (CL_PF_NAT != 0);
    auto group = [&](int gg) {
        const bool more = gg + 1 < 2 * s;
        if (PF) { if (more) { ln_issue(gg + 1); v_issue(gg + 1); } }
        else { ln_issue(gg); ln_take(); }
        if (INL && has_sel) rows(gg, gg & 1, std::true_type{});
        else rows(gg, gg & 1, std::false_type{});
        if (!PF && more) v_issue(gg + 1);
        if (more) v_store(gg + 1, (gg + 1) & 1);
        if (PF && more) ln_take();
        __syncthreads();
    };

    // The window: slot j = low-res row fl - 1 + j, fl = floor(src_y) of the rows being swept.  Group 0 has fl = -1, groups
    // 2 fl + 1 and 2 fl + 2 have fl = 0 .. s-1 (the last fl has one group).  After the groups of fl, row fl - 1 has received its
    // last contribution: it leaves through the horizontal adjoint and the window rotates (a rotation instead of a statically
    // unrolled slot index keeps ONE instance of the row code: fourteen instances were 64 KB of instructions).
    v_issue(0);
    v_store(0, 0);
    if (PF) { ln_issue(0); ln_take(); }
    __syncthreads();
    group(0);
    fold(2);                                      // row -2 -> row 0 (slot 2 at fl = -1)
    rotate();
    for (int fl = 0; fl < s; ++fl) {
        group(2 * fl + 1);
        if (fl < s - 1) group(2 * fl + 2);
        if (fl == s - 1) break;
        if (fl == 0) fold(1);                     // row -1 -> row 0 (slot 1 at fl = 0)
        else flush(fl - 1);
        rotate();
    }
    // fl = s - 1: slots = rows s-2, s-1, s, s+1: the out-of-image rows below the map are row s - 1
#pragma unroll
    for (int t = 0; t < CL_TC / 2; ++t) acc[1][t] += acc[2][t] + acc[3][t];
    flush(s - 2);
    rotate();
    flush(s - 1);
}

// ONE launch per pass for all layers: the layers of the k = 8 class come first (nl4 of them), then the k = 4 class; a
// launch per class left the smaller class (640 workgroups of two waves at the step's shape) alone on the chip.
template <int R, bool SEL, bool INL = false>
__global__ __launch_bounds__(R, SEL ? CL_WAVES_SEL : CL_WAVES_NAT) void skp_map_bwd_col_kernel(ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int li = (int)(blockIdx.x / a.B) / a.nch / a.H;
    if (li < a.nl4) col_body<4, R, SEL, INL>(a, smem);
    else col_body<2, R, SEL, INL>(a, smem);
}

// dot[b, l, h, y, x] = (1 / (L H)) sum_{k < K} p_k g_k   with p_k = exp2(bicubic(S_l)[sel_k] - lse): the softmax backward's row
// term, needed before any token's gradient can be formed.  Row-parallel (no sweep): a workgroup of R threads (lane = column)
// takes DR up-res rows of one (batch row, layer, head): vertical taps of the K selected tokens into LDS (DR x s x K values), then
// the horizontal taps, exp2, the product with the gradient rows.  B * L * H * R / DR workgroups, independent.
#ifndef CL_DR
#define CL_DR 4
#endif
template <int R>
__global__ __launch_bounds__(R) void skp_map_bwd_dot_kernel(ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [CL_DR][s][CL_KMAX]
    constexpr int RR = R * R;
    const int tid = threadIdx.x, H = a.H;
    const int b = blockIdx.x % a.B;
    int rest = blockIdx.x / a.B;
    const int yb = rest % (R / CL_DR);
    rest /= (R / CL_DR);
    const int h = rest % H, li = rest / H;
    const ColLayer ly = a.ly[li];
    const int s = ly.s;
    const float ratio = (float)s / (float)R;
    const float* Sb = ly.S + ((size_t)(b * H + h) * s * s) * a.ldt;
    const int items = CL_DR * s * a.K;
    for (int idx = tid; idx < items; idx += R) {
        const int k = idx % a.K, c = (idx / a.K) % s, r = idx / (a.K * s);
        int cy[4]; float wy[4];
        skp_cubic_taps(yb * CL_DR + r, ratio, s, cy, wy);
        const int tok = (int)a.sel[(size_t)b * a.K + k];
        float v = wy[0] * Sb[((size_t)cy[0] * s + c) * a.ldt + tok];
        v = fmaf(wy[1], Sb[((size_t)cy[1] * s + c) * a.ldt + tok], v);
        v = fmaf(wy[2], Sb[((size_t)cy[2] * s + c) * a.ldt + tok], v);
        v = fmaf(wy[3], Sb[((size_t)cy[3] * s + c) * a.ldt + tok], v);
        smem[(r * s + c) * CL_KMAX + k] = v;
    }
    __syncthreads();
    int cx[4]; float wx[4];
    skp_cubic_taps(tid, ratio, s, cx, wx);
    const size_t plane = ((size_t)b * a.L * H + ly.l * H + h) * RR;
    const float* gb = a.G + (size_t)b * a.K * RR;
#pragma unroll
    for (int r = 0; r < CL_DR; ++r) {
        const int y = yb * CL_DR + r;
        const float lse = a.lse[plane + (size_t)y * R + tid];
        const float* v0 = smem + (r * s + cx[0]) * CL_KMAX;
        const float* v1 = smem + (r * s + cx[1]) * CL_KMAX;
        const float* v2 = smem + (r * s + cx[2]) * CL_KMAX;
        const float* v3 = smem + (r * s + cx[3]) * CL_KMAX;
        float acc = 0.f;
        for (int k = 0; k < a.K; ++k) {
            float val = wx[0] * v0[k];
            val = fmaf(wx[1], v1[k], val); val = fmaf(wx[2], v2[k], val); val = fmaf(wx[3], v3[k], val);
            acc = fmaf(__builtin_amdgcn_exp2f(val - lse), gb[(size_t)k * RR + (size_t)y * R + tid], acc);
        }
        a.dotp[plane + (size_t)y * R + tid] = acc * a.inv_lh;
    }
}

int col_k2(int R, int s) {
    if (s < 8 || R % s) return 0;
    const int k = R / s;
    return (k == 8 || k == 4) ? k / 2 : 0;
}

}  // namespace

extern "C" int skp_attn_map_bwd_col_ok(const int* s, int L, int H, int T, int R, int K) {
    if (!s || L <= 0 || L > SKP_MAX_LAYERS || H <= 0 || T <= 0 || T > 1024 || K <= 0 || K > CL_KMAX) return 0;
    if (R != 128 && R != 256) return 0;
    for (int l = 0; l < L; ++l)
        if (!col_k2(R, s[l]) || s[l] > 64 || (s[l] & 3)) return 0;
    return 1;
}

extern "C" int64_t skp_attn_map_bwd_col_workspace(const int* s, int L, int B, int H, int T, int R, int K) {
    if (!s || B <= 0) return SKP_E_BADARG;
    if (!skp_attn_map_bwd_col_ok(s, L, H, T, R, K)) return SKP_E_RANGE;
    int64_t fl = 2 * (int64_t)B * L * H * R * R;               // dot parts
    for (int l = 0; l < L; ++l) fl += (int64_t)B * H * s[l] * s[l] * CL_KMAX;
    return fl * (int64_t)sizeof(float) + 64;
}

extern "C" int skp_attn_map_bwd_col_f32(const float* const* S, float* const* dS, const int* s, int L, int B, int H, int T,
                                        int R, const int64_t* sel, const float* G, int K, const float* lse, void* workspace,
                                        int ldt, void* stream) {
    if (!S || !dS || !s || !sel || !G || !lse || !workspace || L <= 0 || B <= 0 || H <= 0 || T <= 0 || R <= 0 || K <= 0)
        return SKP_E_BADARG;
    if (!skp_attn_map_bwd_col_ok(s, L, H, T, R, K)) return SKP_E_RANGE;
    const int nt = ((T + 15) / 16) * 16;
    if (ldt < nt || (ldt & 3)) return SKP_E_BADARG;
    for (int l = 0; l < L; ++l)
        if (!S[l] || !dS[l]) return SKP_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)(((uintptr_t)workspace + 31) & ~(uintptr_t)31);
    float* dotp = ws;
    float* pcur = ws + 2 * (size_t)B * L * H * R * R;
    float* Psel[SKP_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
        Psel[l] = pcur;
        pcur += (size_t)B * H * s[l] * s[l] * CL_KMAX;
    }
    const int nsel = (K + CL_TC - 1) / CL_TC;
    ColArgs a{};
    a.sel = sel; a.G = G; a.lse = lse; a.dotp = dotp;
    a.L = L; a.B = B; a.H = H; a.T = T; a.K = K; a.ldt = ldt; a.NT = nt; a.nsel = nsel;
    a.inv_lh = 1.0f / (float)(L * H);
    size_t lds = 0;
    for (int k2 = 4; k2 >= 2; k2 -= 2) {
        for (int l = 0; l < L; ++l) {
            if (col_k2(R, s[l]) != k2) continue;
            ColLayer& y = a.ly[a.nl++];
            y.S = S[l]; y.dS = dS[l]; y.Psel = Psel[l]; y.l = l; y.s = s[l];
            const int e = k2 == 2 ? 2 : 1;
            const size_t need = (2 * (size_t)k2 * s[l] * CL_TS + 2 * ((size_t)R * e + 1) * CL_PS + 4 * (size_t)R +
                                 (size_t)s[l] * CL_LUSE + CL_KMAX) * sizeof(float);
            lds = need > lds ? need : lds;
        }
        if (k2 == 4) a.nl4 = a.nl;
    }
    // Route: the inline form everywhere it was measured (B = 8, R = 128, 16^2 x 3 + 32^2 layers: T = 77 344 -> 268 us, T = 128
    // 428 -> 394, T = 500 1 379 -> 1 217; 2 rows 213 -> 128; R = 256 746 -> 624).  With the row code instantiated twice (chunks
    // with / without a selected token) the sweep kernel needs 156 registers and no scratch (the two-sweep natural kernel: 240 + 12
    // bytes).
    {              // dot by its own row-parallel kernel, then ONE sweep launch: natural chunks with their selected tokens inline
        int smax = 0;
        for (int l = 0; l < L; ++l) smax = s[l] > smax ? s[l] : smax;
        const dim3 dgrid((unsigned)((long)B * (R / CL_DR) * H * a.nl)), block(R);
        const size_t dlds = (size_t)CL_DR * smax * CL_KMAX * sizeof(float);
        if (R == 128) hipLaunchKernelGGL((skp_map_bwd_dot_kernel<128>), dgrid, block, dlds, st, a);
        else hipLaunchKernelGGL((skp_map_bwd_dot_kernel<256>), dgrid, block, dlds, st, a);
        int rc = skp_launch_status();
        if (rc) return rc;
        a.nch = nt / CL_TC;
        const dim3 grid((unsigned)((long)B * a.nch * H * a.nl));
        if (R == 128) hipLaunchKernelGGL((skp_map_bwd_col_kernel<128, false, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((skp_map_bwd_col_kernel<256, false, true>), grid, block, lds, st, a);
        return skp_launch_status();
    }
}
