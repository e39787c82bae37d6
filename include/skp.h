/* skp.h -- C ABI of libskp_hip.so: the MI355X (gfx950) implementation of the StableKeypoints
 * token-optimisation hot path.  Plain pointers + sizes only; no torch types.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller unless marked [host];
 *     kernels never allocate, never synchronise, and launch on `stream` (a hipStream_t
 *     passed as void*; NULL = the default stream);
 *   - tensors are dense, row-major, fp32 ("f32") unless stated; indices are int64 ("i64")
 *     or int32 ("i32") as named;
 *   - return value: 0 on success; a positive value is the hipError_t of the failed launch;
 *     a negative value is an argument error (SKP_E_*).  Nothing is written on error.
 *
 * Reference interfaces replaced (paths relative to the reference's unsupervised_keypoints/):
 *   skp_qk_logits_f32 / skp_gemm_nt_f32     ptp_utils.py:483-493,531-534 (q.k^T contractions)
 *   skp_attn_map_fwd_f32 / _bwd_f32         ptp_utils.py:513-538 + optimize.py:27-79
 *                                           (bicubic up-res softmax map, per-head store, layer/head mean)
 *   skp_cross_attn_fwd_f32 / _bwd_f32       ptp_utils.py:493-506,540 (ordinary softmax(QK^T)V, cross layers)
 *   skp_self_attn_fwd_f32 / _bwd_f32        ptp_utils.py:493-506,540 (self-attention layers, flash-style)
 *   skp_flash_attn_fwd_f32 / _bwd_f32       ptp_utils.py:493-506,540 (any key count: cross layers with T > 128 tokens)
 *   skp_conv3x3[_f4]_f32 / skp_conv3x3_s2_f32 / skp_conv3x3_small_f32  3x3 convolutions of the frozen UNet / VAE blocks (ptp_utils.py:227-229, 289-304)
 *   skp_group_norm_fwd_f32 / _bwd_f32       GroupNorm+SiLU of the hooked UNet / VAE forward (ptp_utils.py:227-229, 289-304)
 *   skp_token_stats_f32                     eval.py:39-111 + ptp_utils.py:95-108
 *   skp_select_tokens                       ptp_utils.py:110-112,115-159
 *   skp_losses_fwd_f32                      optimize.py:157-206, optimize_token.py:203-241,
 *                                           invertable_transform.py:72-92
 *   skp_rows_axpy_f32                       autograd scatter of the loss gradients into [T,R,R]
 */
#ifndef SKP_H
#define SKP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKP_MAX_LAYERS 8
#define SKP_MAX_SUBJECTS 4

#define SKP_E_BADARG (-1)   /* null pointer / non-positive size */
#define SKP_E_RANGE  (-2)   /* size outside what the kernels are built for (see each function) */
#define SKP_E_LDS    (-3)   /* tile does not fit the 160 KiB LDS of a gfx950 CU */

/* ABI version; bumped on any signature change. */
int skp_abi_version(void);

/* Developer overrides of launch plans, for tests/ and tools/ only (the product path never calls them; the library reads no
 * environment variables).  Keys: "wino_split" (force the K split of the F(4x4,3x3) launches), "wino_raw_max_tiles" (widen the
 * raw-filter form's gate), "map_bands" (band count of the token-major map backward), "fa2_two_kernel_bwd" (1: the two-kernel
 * flash backward at the fused form's shapes), "gn_fold_max_cout", "cross_attn_ts" (1: the 128-query
 * cross-attention kernels where the token-split form would run).  value 0 = the library's own choice.  Process-global, not
 * thread-safe.  Returns 0 / the value, SKP_E_RANGE for an unknown key. */
int skp_tune_set(const char* key, int value);
int skp_tune_get(const char* key);

/* Batched NT GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32; exact fp32 fma chain):
 *   C[z0,z1][m,n] = alpha * sum_k A[z0,z1][m,k] * B[z0,z1][n,k]          z0<Z0, z1<Z1
 * with arbitrary element strides (sXm,sXk / sXn,sXk) and two batch strides per operand
 * (a batch stride of 0 broadcasts).  C is written densely with strides (sc0,sc1,scm,1).
 * Used for S_low = scale*K.Q^T and for its two backward products. */
int skp_gemm_nt_f32(const float* A, const float* B, float* C, int M, int N, int K, int Z0, int Z1,
                    int64_t sa0, int64_t sa1, int64_t sam, int64_t sak,
                    int64_t sb0, int64_t sb1, int64_t sbn, int64_t sbk,
                    int64_t sc0, int64_t sc1, int64_t scm, float alpha, void* stream);

/* Low-resolution cross-attention logits of one hooked layer, log2 domain:
 *   S[b,h,p,t] = (scale*log2(e)) * sum_c q[b,p,h*d+c] * k[bk,t,h*d+c]
 * q: [B, s*s, H*d] (= to_q(x), ptp_utils.py:483), k: [Bk, T, H*d] with Bk in {1,B}
 * (= to_k(context), :487; Bk==1 is the reference's context.repeat(B,1,1), ptp_utils.py:229),
 * S: [B, H, s*s, NT] token-contiguous, NT = 16*ceil(T/16); columns t >= T are written as 0. */
int skp_qk_logits_f32(const float* q, const float* k, float* S, int B, int Bk, int H, int T, int s2,
                      int d, float scale, void* stream);

/* Fused up-res attention map (forward).  For every layer l<L (side s[l], logits S[l] from
 * skp_qk_logits_f32), head h<H and output pixel p of the R x R grid:
 *   P[l,h,t,p] = softmax_t( bicubic_{s[l]->R}(S[l][b,h,t,:,:])[p] )     (align_corners=False, A=-0.75)
 *   M[b,t,p]   = 1/(L*H) * sum_{l,h} P[l,h,t,p]
 * which equals collect_maps(upsample_res=-1) over the reference's stored tensors because
 * to_q is bias-free and bicubic resize is linear.
 * M:   [B, T, R, R]            (written)
 * lse: [B, L*H, R*R]           (written; log2-sum-exp per (layer, head, pixel), needed by _bwd)
 * S[l]: [B, H, s[l]*s[l], NT] (layout of skp_qk_logits_f32).
 * Limits: 1 <= T <= 128 (more tokens: skp_attn_map_fwd_ex_f32), L <= SKP_MAX_LAYERS, s[l] <= 64, R <= 2048.  */
int skp_attn_map_fwd_f32(const float* const* S /*[host] L device ptrs*/, const int* s /*[host] L*/,
                         int L, int B, int H, int T, int R, float* M, float* lse, void* stream);

/* Token-group form of the forward for T > 128 learned tokens (the reference CLI default is 500, main.py:77-79).
 * The launch covers T (<= 128) tokens whose column offset the caller has folded into the S[l] and M pointers;
 * ldt = row stride of the logits (floats, multiple of 4), m_bstride = floats between batch rows of M.
 *   mode 0: self-contained (== skp_attn_map_fwd_f32)
 *   mode 1: statistics only -- writes this group's log2-sum-exp to lse_out, M untouched
 *   mode 2: apply -- P = exp2(S - lse_in) against the log2-sum-exp over ALL groups (caller combines the
 *           per-group values: lse = log2 sum_g 2^lse_g), writes this group's rows of M */
int skp_attn_map_fwd_ex_f32(const float* const* S /*[host]*/, const int* s /*[host]*/, int L, int B, int H,
                            int T, int R, float* M, float* lse_out, const float* lse_in, int ldt,
                            int64_t m_bstride, int mode, void* stream);

/* Bytes of scratch skp_attn_map_bwd_f32 needs (staging of the horizontally-reduced gradient between its
 * two kernels); negative on bad arguments. */
int64_t skp_attn_map_bwd_workspace(const int* s /*[host]*/, int L, int B, int H, int T, int R);

/* Backward of skp_attn_map_fwd_f32: dS[l] (natural-log domain, i.e. d loss / d (scale*q.k), shape and
 * layout of S[l]) is WRITTEN (every element, pad columns = 0).  Deterministic: no atomics.
 * dM: [B,T,R,R]; workspace: skp_attn_map_bwd_workspace() bytes, contents irrelevant. */
int skp_attn_map_bwd_f32(const float* const* S /*[host]*/, float* const* dS /*[host]*/,
                         const int* s /*[host]*/, int L, int B, int H, int T, int R,
                         const float* dM, const float* lse, float* workspace, void* stream);

/* One-pass forward for a WIDE token axis (128 < T <= 1024; the reference CLI default is --num_tokens 500, main.py:77-79):
 * same result as skp_attn_map_fwd_f32, the tokens of a pixel spread over ceil(T/64) lanes (csrc/skp_attn_map_wide.hip)
 * instead of token groups x two passes.  S[l] rows have stride ldt >= 16*ceil(T/16).  lse: [B,L*H,R*R] written.
 * tokrow (device, [T] int32, or NULL): output row of token t or -1 -- only those rows of M are written and M is
 * [B,n_rows,R,R] (optimize.py:58-59 keeps `indices` of the T maps); NULL: M is [B,T,R,R].  Requires R % 32 == 0. */
int skp_attn_map_fwd_wide_f32(const float* const* S /*[host]*/, const int* s /*[host]*/, int L, int B, int H, int T,
                              int R, float* M, float* lse, const int* tokrow, int n_rows, int ldt, void* stream);
/* 1 when the wide kernel takes these sizes (side <= 64, R % 32 == 0, R*(R/32) <= 65535 tiles, the LDS plan fits), else 0:
 * the host-side gate callers use to choose between it and the two-pass token-group route (host logic, no launch). */
int skp_attn_map_fwd_wide_ok(const int* s /*[host]*/, int L, int T, int R);

/* "Column sweep" form of the T <= 128 sparse-gradient backward (csrc/skp_attn_map_col.hip, round 4): a workgroup sweeps all R rows
 * of one (batch row, layer, head, 8-token chunk), the vertical adjoint lives in a four-row register window and the horizontal
 * adjoint is applied to COMPLETE low-res rows only -- no dV staging, no band partials, dS written once.  Same arguments and
 * result as skp_attn_map_bwd_sparse_f32.  skp_attn_map_bwd_col_ok: 1 when served (R in {128, 256}, every layer R = k*s with
 * k in {4, 8}, 8 <= s <= 64, T <= 128, K <= 16), else 0 (host logic).  workspace: skp_attn_map_bwd_col_workspace() bytes
 * (dot = sum_k p_k g_k per (layer, head, pixel) + the selected tokens' part). */
int skp_attn_map_bwd_col_ok(const int* s /*[host]*/, int L, int H, int T, int R, int K);
int64_t skp_attn_map_bwd_col_workspace(const int* s /*[host]*/, int L, int B, int H, int T, int R, int K);
int skp_attn_map_bwd_col_f32(const float* const* S /*[host]*/, float* const* dS /*[host]*/, const int* s /*[host]*/, int L,
                             int B, int H, int T, int R, const int64_t* sel, const float* G, int K, const float* lse,
                             void* workspace, int ldt, void* stream);

/* Backward of the fused map for a SPARSE map gradient: the losses of optimize.py:157-206 index the map with the K
 * selected tokens (optimize.py:395-414), so dM is non-zero on K rows per batch row only.
 *   sel: [B,K] int64 token ids (distinct per row), G: [B,K,R,R] = those rows of dM (NOT divided by L*H),
 *   lse: [B,L*H,R*R] from the forward.  dS[l] (layout of S[l], row stride ldt >= 16*ceil(T/16); natural-log domain) is
 *   WRITTEN: every row, columns t < 16*ceil(T/16), pad columns = 0.  Any T (no token groups).
 * Token-major sweep (csrc/skp_attn_map_tok.hip): no atomics, fixed summation order => bit-reproducible.
 * Limits: s[l] <= 32, K <= 32, R <= 1024 (SKP_E_RANGE otherwise: use skp_attn_map_bwd_f32 with the dense dM).
 * workspace: skp_attn_map_bwd_sparse_workspace() bytes. */
int64_t skp_attn_map_bwd_sparse_workspace(const int* s /*[host]*/, int L, int B, int H, int T, int R, int K);
int skp_attn_map_bwd_sparse_f32(const float* const* S /*[host]*/, float* const* dS /*[host]*/, const int* s /*[host]*/,
                                int L, int B, int H, int T, int R, const int64_t* sel, const float* G, int K,
                                const float* lse, void* workspace, int ldt, void* stream);

/* Tail of the augmented inference (eval.py:239-353) for n views of one image and K selected tokens, one kernel:
 *   tot[k] = sum_v grid_sample(bilinear_{R->S}(M[v,k]), affine_grid(theta_inv[v]))     (bilinear, zeros, align_corners=False)
 *   num    = sum_v grid_sample(ones,                    affine_grid(theta_inv[v]))     (coverage; the same for every k)
 * M: [n,K,R,R]; theta_inv: [n,6] (device) = the INVERSE 2x3 affine of every view (invertable_transform.py:77-84);
 * tot: [K,S,S] written; num: [S,S] written (may be NULL).  finish != 0: tot = tot / num with 0/0 -> 0 (eval.py:343-346).
 * Limits: K <= 32. */
int skp_unwarp_accumulate_f32(const float* M, const float* theta_inv, int n, int K, int R, int S, float* tot, float* num,
                              int finish, void* stream);

/* GroupNorm folded into the consuming convolution (forward only; the VAE encoder of ptp_utils.py:289-304 runs without
 * autograd): skp_group_norm_coef_f32 leaves (scale, shift) per (sample, channel) -- statistics from the producing
 * convolution's block sums (bs != NULL) or from one pass over x -- and skp_conv3x3_f4_gn_f32 computes
 * conv3x3(silu(x * scale + shift)) with the normalisation applied in its patch load: the GroupNorm apply pass (one read
 * and one write of the activation) disappears.  skp_conv3x3_f4_gn_ok: 1 where the folded form serves the launch: 128-channel
 * workgroup form, unsplit, at most four output-channel groups (Cout <= 512; every group redoes the SiLU of its patches, which
 * is cheaper than the apply pass up to there). */
int skp_group_norm_coef_f32(const float* x, const float* off, const float* gamma, const float* beta, float* mean, float* rstd,
                            float* coef, const float* bs, int nblk, int pix, float* workspace, int N, int C, int G, int HW,
                            float eps, void* stream);
int skp_conv3x3_f4_gn_ok(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_f4_gn_f32(const void* x, const void* U, const void* bias, const void* residual, void* y, float* stats,
                          const float* coef, int B, int Cin, int Cout, int H, int W, void* stream);

/* The same convolution (diffusers ResnetBlock2D.conv1 / conv2 of the UNet's 8^2 .. 16^2 levels, reached from
 * ptp_utils.py:213-217) for SMALL spatial sizes with many channels, where the F(4x4,3x3) kernels above are bound by their
 * filter stream (36 transformed values per channel pair for a handful of tiles): the filter stays as its 9 taps
 * (skp_conv3x3_f4r_filter_f32: R, 9 * Cin * Cout floats, in MFMA operand order; flip_transpose as above), G g G^T is applied
 * by the lanes on the way into the matrix cores, and the input transform runs once per launch into the workspace.
 * Shapes the kernel RUNS: Cin % 16 == 0, Cout % 64 == 0, H, W % 4 == 0 (else SKP_E_RANGE).  skp_conv3x3_f4r_ok: 1 where
 * it also PAYS against the kernels above (measured: >= 1280 channels on both sides up to 1536 tiles, <= 512 tiles with
 * >= 1920 on one; SKP_WINO_RAW=0 turns it off) -- the rule the Python layer routes by.  workspace:
 * skp_conv3x3_f4r_workspace() bytes, REQUIRED (pre-transformed input + K-split partials).  fp32, fixed K-split order
 * (bit-reproducible). */
int skp_conv3x3_f4r_ok(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_f4r_filter_f32(const void* w, void* R, int Cout, int Cin, int flip_transpose, void* stream);
int64_t skp_conv3x3_f4r_workspace(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_f4r_f32(const void* x, const void* R, const void* bias, const void* residual, void* y, void* workspace,
                        int B, int Cin, int Cout, int H, int W, void* stream);

/* Ordinary cross-attention core (ptp_utils.py:493-506,540) for a short key axis, fp32 MFMA, K/V staged in
 * LDS, softmax over the tokens in registers:
 *   out[b,n,h*d+c] = sum_t softmax_t(scale * q[b,n,h,:].k[bk,t,h,:]) * v[bk,t,h*d+c]
 * q, out: [B,N,H*d]; k, v: [Bk,T,H*d] with Bk in {1,B}; lse: [B,H,N] natural-log sum-exp (for _bwd).
 * Limits: T <= 128, d in {8,16,32,40,64,80,160}. */
int skp_cross_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                           int B, int Bk, int H, int N, int T, int d, float scale, void* stream);
/* Scratch bytes of skp_cross_attn_bwd_f32 (token-major staging of P and dS); negative on bad arguments. */
int64_t skp_cross_attn_bwd_workspace(int B, int H, int N, int T, int d);
/* Backward: dq [B,N,H*d] and dk, dv [B,T,H*d] are WRITTEN (per batch row; the caller sums dk/dv over b
 * when Bk == 1).  Deterministic (no atomics). */
int skp_cross_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out,
                           const float* dout, const float* lse, float* dq, float* dk, float* dv,
                           float* workspace, int B, int Bk, int H, int N, int T, int d, float scale,
                           void* stream);

/* Token-group form of the backward.  mode 1: writes dot_io[b,l*H+h,p] = sum_{t in group} P*dM/(L*H) only;
 * mode 2: reads dot_io = that sum over ALL groups and writes this group's columns of dS[l]; mode 0 as above.
 * lse is always the log2-sum-exp over all tokens. */
int skp_attn_map_bwd_ex_f32(const float* const* S /*[host]*/, float* const* dS /*[host]*/, const int* s /*[host]*/,
                            int L, int B, int H, int T, int R, const float* dM, const float* lse,
                            float* workspace, float* dot_io, int ldt, int64_t m_bstride, int mode, void* stream);

/* Flash-style attention for any key count (ptp_utils.py:493-506,540): the general form behind skp_self_attn_* and the
 * cross-attention layers whose key axis exceeds skp_cross_attn_*'s 128 tokens (the reference CLI default is
 * --num_tokens 500, main.py:77-79).  fp32 MFMA, 64-key tiles through LDS, online softmax (running max / sum).
 *   out[b,n,h*d+c] = sum_t softmax_t(scale * q[b,n,h,:].k[bk,t,h,:]) * v[bk,t,h*d+c]
 * q, out: [B,N,H*d]; k, v: [Bk,Nk,H*d] with Bk in {1,B} (Bk == 1: one k/v shared by all rows, ptp_utils.py:229);
 * lse: [B,H,N] natural-log sum-exp.  d in {8,16,32,40,64,80,160}.
 * _bwd: dq [B,N,H*d], dk, dv [B,Nk,H*d] WRITTEN per batch row (the caller sums dk/dv over b when Bk == 1);
 * workspace: skp_flash_attn_bwd_workspace() bytes (>= B*H*N floats; the fused single-pass backward of the big 40- / 64- / 80-wide
 * self-attention layers adds per-key-block dQ partials).  Deterministic (no atomics).  skp_self_attn_bwd_f32 keeps its
 * B*H*N-float workspace contract and therefore always runs the two-kernel form. */
int64_t skp_flash_attn_bwd_workspace(int B, int Bk, int H, int N, int Nk, int d);
int skp_flash_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                           int B, int Bk, int H, int N, int Nk, int d, float scale, void* stream);
int skp_flash_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                           const float* lse, float* dq, float* dk, float* dv, float* workspace,
                           int B, int Bk, int H, int N, int Nk, int d, float scale, void* stream);
/* _bwd_ld: dq, dk, dv are column bands of wider row-major buffers, row stride `ldg` floats (>= H*d, a multiple of 4): the three
 * gradients of a self-attention block written side by side into one [B*N, 3*H*d] buffer make the input gradient of its frozen
 * q / k / v projections (ptp_utils.py:513-520) a single GEMM.  d in {40,64,80,160}; SKP_E_RANGE for the other head sizes. */
int skp_flash_attn_bwd_ld_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                              const float* lse, float* dq, float* dk, float* dv, float* workspace,
                              int B, int Bk, int H, int N, int Nk, int d, float scale, int ldg, void* stream);

/* The flash-attention FORWARD on the BF16 matrix cores with three-term operand splits (skp_flash_attn_s.hip; same role and
 * contract as skp_flash_attn_fwd_f32: out, natural-log lse for the fp32 backward kernels): fp32 in / out, fp32 softmax and
 * accumulation, every operand of S^T = K.Q^T and O^T += V^T.P^T the exact sum of three bf16 terms, six products per fp32 product
 * on v_mfma_f32_16x16x32_bf16.  K / V are split once per call into tile images in `workspace`
 * (skp_flash_attn_fwd_split_workspace() bytes, REQUIRED).  OPT-IN experiment; d in {40, 80} (skp_flash_attn_fwd_split_ok). */
int skp_flash_attn_fwd_split_ok(int B, int Bk, int H, int N, int Nk, int d);
int64_t skp_flash_attn_fwd_split_workspace(int B, int Bk, int H, int N, int Nk, int d);
int skp_flash_attn_fwd_split_f32(const float* q, const float* k, const float* v, float* out, float* lse, void* workspace,
                                 int B, int Bk, int H, int N, int Nk, int d, float scale, void* stream);
/* Its backward for the self-attention shapes (Bk == B, Nk == N, d == 40: the 64^2 layers): dq, dk, dv as skp_flash_attn_bwd_f32
 * from (q, k, v, out, dout, lse); two kernels (dQ: lane = query; dK / dV: lane = key), seven tile products on the split
 * tuples, D = rowsum(dout * out) computed on the way.  workspace: skp_flash_attn_bwd_split_workspace() bytes, REQUIRED. */
int skp_flash_attn_bwd_split_ok(int B, int Bk, int H, int N, int Nk, int d);
int64_t skp_flash_attn_bwd_split_workspace(int B, int Bk, int H, int N, int Nk, int d);
int skp_flash_attn_bwd_split_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                 const float* lse, float* dq, float* dk, float* dv, void* workspace, int B, int Bk, int H, int N,
                                 int Nk, int d, float scale, void* stream);
/* ... with dq, dk, dv as column bands of wider buffers (row stride ldg floats), as skp_flash_attn_bwd_ld_f32 */
int skp_flash_attn_bwd_split_ld_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                    const float* lse, float* dq, float* dk, float* dv, void* workspace, int B, int Bk, int H,
                                    int N, int Nk, int d, float scale, int ldg, void* stream);

/* Flash-style self-attention (ptp_utils.py:493-506 with context = x) for the long image-token sequences: fp32 MFMA,
 * 64-key tiles in LDS, online softmax; the [B*h,N,N] scores are never materialised.
 * q, k, v, out: [B,N,H*d]; lse: [B,H,N] (natural log).  Limits: d in {8,16,32,40,64,80,160}. */
int skp_self_attn_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                          int B, int H, int N, int d, float scale, void* stream);
/* Backward: dq, dk, dv [B,N,H*d] written; workspace: B*H*N floats.  Deterministic (no atomics). */
int skp_self_attn_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                          const float* lse, float* dq, float* dk, float* dv, float* workspace,
                          int B, int H, int N, int d, float scale, void* stream);

/* Fused GroupNorm (+ per-(sample,channel) offset) (+ SiLU) of the frozen UNet/VAE blocks (diffusers ResnetBlock2D:
 * conv -> [+bias, + time embedding] -> GroupNorm -> SiLU), NCHW:
 *   y[n,c,p] = act( ((x[n,c,p] + off[n,c]) - mean[n,g]) * rstd[n,g] * gamma[c] + beta[c] ),  act = SiLU if silu else id
 * x, y: [N,C,HW] (HW % 4 == 0, C % G == 0, N*G <= 65535); off: [N,C] or NULL; mean, rstd: [N,G] written (for _bwd);
 * workspace: N*G*64*3 floats. */
int skp_group_norm_fwd_f32(const float* x, const float* off, const float* gamma, const float* beta, float* y,
                           float* mean, float* rstd, float* workspace, int N, int C, int G, int HW, float eps,
                           int silu, void* stream);
/* 1 when skp_group_norm_fwd_f32 serves this shape with its one-pass form (the (sample, group) row held in registers, exact
 * statistics, one launch): a producer then need not leave block statistics behind (skp_conv3x3_f4_stats_f32). */
int skp_group_norm_onepass_ok(int N, int C, int G, int HW);
/* dx [N,C,HW] = d loss / d x given dy (gamma/beta/off are frozen on this path). workspace as above. */
int skp_group_norm_bwd_f32(const float* x, const float* off, const float* gamma, const float* beta,
                           const float* dy, const float* mean, const float* rstd, float* dx, float* workspace,
                           int N, int C, int G, int HW, float eps, int silu, void* stream);
/* The same plus `dadd` [N,C,HW]: dx = (input gradient of the norm) + dadd -- x feeds the norm AND its block's residual path
 * (ResnetBlock2D / Transformer2DModel); the residual path's gradient is added here instead of by an add pass of its own. */
int skp_group_norm_bwd_add_f32(const float* x, const float* off, const float* gamma, const float* beta,
                               const float* dy, const float* mean, const float* rstd, float* dx, const float* dadd,
                               float* workspace, int N, int C, int G, int HW, float eps, int silu, void* stream);

/* out[n,c,p] = a[n,c,p] + b[n,c,p] + bias[c]  (ResnetBlock2D tail: shortcut + conv2 + conv2.bias in one pass). HW % 4 == 0. */
int skp_add_bias_residual_f32(const float* a, const float* b, const float* bias, float* out, int N, int C, int HW,
                              void* stream);

/* Residual add + LayerNorm of the transformer blocks (diffusers BasicTransformerBlock [third party]: `attn(norm(h)) + h`
 * chains, reached from ptp_utils.py:213-217) in one pass per direction.  Rows of C floats, C a multiple of 32 with
 * C / 32 .. C / 256 in {1..6, 8} float4 per lane (skp_add_layer_norm_ok(C) == 1: 320, 640, 1280, 1024, 768, 512, 64, 32 ...).
 * fwd: x = d + h (d may be NULL: x is then not written and n = LN(h)), n = (x - mean) * rstd * gamma + beta,
 *      stat [rows][2] = (mean, rstd); two-pass statistics.
 * bwd: dx = LN'(dn; x, stat, gamma) (+ dskip, the gradient reaching x from its other uses; may be NULL).  gamma / beta are
 *      frozen: no parameter gradients.  Deterministic. */
int skp_add_layer_norm_ok(int C);
int skp_add_layer_norm_fwd_f32(const float* d, const float* h, const float* gamma, const float* beta, float* x, float* n,
                               float* stat, int64_t rows, int C, float eps, void* stream);
int skp_add_layer_norm_bwd_f32(const float* dn, const float* dskip, const float* x, const float* stat, const float* gamma,
                               float* dx, int64_t rows, int C, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of the frozen blocks (diffusers ResnetBlock2D.conv1/conv2, Upsample2D.conv,
 * AutoencoderKL encoder resnets [third party], reached from ptp_utils.py:213-217 and :287) as Winograd F(2x2,3x3)
 * on the fp32 matrix cores.
 * skp_conv3x3_filter_f32: one-off transform of a frozen weight w [Cout,Cin,3,3] into the kernel's operand order
 *   U [16][Cin/8][2][Cout][4] (16*Cin*Cout floats; Cin % 8 == 0).  flip_transpose = 1 builds the filter of the
 *   backward-data convolution instead: pass the SAME w [Cw_out,Cw_in,3,3] with Cout = Cw_in, Cin = Cw_out.
 * skp_conv3x3_f32: y [B,Cout,H,W] = conv(x [B,Cin,H,W]) (+ bias[Cout]) (+ residual [B,Cout,H,W]), both optional
 *   (NULL); the residual is the ResnetBlock2D shortcut, added in the epilogue instead of a separate pass.  Cout % 32 == 0;
 *   variant 0 = auto, 1 = workgroup of 128 channels x 32 tiles (Cin % 32 == 0), 2 = 64 channels x 64 tiles
 *   (Cin % 16 == 0).  x and U must each be < 2 GiB (32-bit buffer offsets), else SKP_E_RANGE.
 *   Layers with too few workgroups for the 256 CUs are split over input channels (deterministic: partial outputs in
 *   `workspace`, summed in fixed order by a second kernel); skp_conv3x3_workspace gives the bytes needed (0 = no
 *   split).  workspace NULL forces the unsplit launch. */
int skp_conv3x3_filter_f32(const void* w, void* U, int Cout, int Cin, int flip_transpose, void* stream);
int64_t skp_conv3x3_workspace(int B, int Cin, int Cout, int H, int W, int variant);
int skp_conv3x3_f32(const void* x, const void* U, const void* bias, const void* residual, void* y, void* workspace,
                    int B, int Cin, int Cout, int H, int W, int variant, void* stream);

/* The same convolution as Winograd F(4x4,3x3) (4x fewer multiplies than direct; skp_conv_wino4.hip) for H % 4 == 0,
 * W % 4 == 0, Cin % 16 == 0, Cout % 16 == 0.  U [36][Cin/16][4][Cout][4] = 36*Cin*Cout floats from
 * skp_conv3x3_f4_filter_f32 (same flip_transpose convention as skp_conv3x3_filter_f32).  fp32 throughout; the larger
 * transform costs ~1 decimal digit (relative error ~3e-6 of the output maximum instead of ~3e-7). */
int skp_conv3x3_f4_filter_f32(const void* w, void* U, int Cout, int Cin, int flip_transpose, void* stream);
int64_t skp_conv3x3_f4_workspace(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_f4_f32(const void* x, const void* U, const void* bias, const void* residual, void* y, void* workspace,
                       int B, int Cin, int Cout, int H, int W, void* stream);

/* Output statistics for the GroupNorm that follows a convolution (diffusers ResnetBlock2D: conv1 -> norm2, conv2 + shortcut
 * -> the next block's norm1 [third party]): the convolution epilogues leave per-(image, channel, pixel-block) sums of their
 * OUTPUT (after bias / shortcut) behind, and the norm takes its mean / variance from them instead of re-reading the
 * activation (one full HBM pass per norm, 1 GB at the VAE's 512^2 level).
 *   skp_conv3x3_f4_stats_blocks  -> blocks per image (16 tiles = 256 pixels each) for this launch shape, 0 = not available
 *                                   (split-K launch, or blocks would straddle images)
 *   skp_conv3x3_f4_stats_f32     = skp_conv3x3_f4_f32 (unsplit) + stats [B][Cout][blocks][2] = {sum y, sum y^2}
 *   skp_conv3x3_s2_stats_f32     = skp_conv3x3_s2_f32 + stats [B][Cout][(H/16)*(W/32)][2] over 8x16-pixel output tiles
 *   skp_group_norm_fwd_blocks_f32 = skp_group_norm_fwd_f32 with mean / rstd computed from bs [N][C][nblk][2] (each block =
 *                                   `pix` pixels, nblk * pix == HW; fp64 combine; the per-(sample, channel) offset is folded in
 *                                   analytically).  mean, rstd [N,G] are written as usual. */
int skp_conv3x3_f4_stats_blocks(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_f4_stats_f32(const void* x, const void* U, const void* bias, const void* residual, void* y, float* stats,
                             int B, int Cin, int Cout, int H, int W, void* stream);
int skp_conv3x3_s2_stats_f32(const void* x, const void* U, const void* bias, void* y, float* stats, int B, int Cin, int Cout,
                             int H, int W, int pad, void* stream);
int skp_group_norm_fwd_blocks_f32(const float* x, const float* off, const float* gamma, const float* beta, float* y,
                                  float* mean, float* rstd, const float* bs, int nblk, int pix, int N, int C, int G, int HW,
                                  float eps, int silu, void* stream);

/* 3x3 / STRIDE 2 convolution, forward (diffusers Downsample2D [third party]: the three down-sampling convolutions of the
 * frozen VAE encoder, reached from ptp_utils.py:289-304 `image2latent`, run under no_grad), direct implicit GEMM on the fp32
 * matrix cores, NCHW in and out, zero padding and bias folded in (no F.pad copy, no NCHW<->NHWC transposes):
 *   y[b,co,oy,ox] = bias[co] + sum_{ci,a,c} w[co,ci,a,c] * x[b,ci,2*oy+a-pad,2*ox+c-pad]   (out of range = 0), OH = H/2, OW = W/2
 *   pad = 0: the asymmetric (0,1,0,1) padding of the VAE's Downsample2D(padding=0); pad = 1: symmetric padding 1.
 * skp_conv3x3_s2_filter_f32: one-off re-ordering of a frozen weight w [Cout,Cin,3,3] into U [9][Cin/16][4][Cout][4].
 * Limits: Cin % 16 == 0, Cout % 32 == 0, H/2 % 8 == 0, W/2 % 16 == 0, x, U, y < 2 GiB each; bias may be NULL. */
int skp_conv3x3_s2_filter_f32(const void* w, void* U, int Cout, int Cin, void* stream);
int skp_conv3x3_s2_f32(const void* x, const void* U, const void* bias, void* y, int B, int Cin, int Cout, int H, int W,
                       int pad, void* stream);
/* The same with scratch for a K split over the input channels where the launch alone would not fill the chip (the UNet's
 * down-sampling layers at a few rows: 80-192 workgroups): skp_conv3x3_s2_workspace() bytes (0 = none needed, pass NULL);
 * partial sums per split, added in fixed order with the bias by a second kernel.  Deterministic. */
int64_t skp_conv3x3_s2_workspace(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_s2_ws_f32(const void* x, const void* U, const void* bias, void* y, void* workspace, int B, int Cin, int Cout,
                          int H, int W, int pad, void* stream);

/* 3x3 / stride 1 / padding 1 convolution with AT MOST FOUR input channels (the `conv_in` layers: VAE encoder 3 -> 128 on the
 * image, ptp_utils.py:289-304; UNet 4 -> 320 on the latents, ptp_utils.py:227), forward, NCHW, bias folded in (may be NULL).
 * Output-bandwidth bound VALU kernel (two pixels per thread, wave-uniform weights).  w: the module's own [Cout,Cin,3,3].
 * Limits: Cin <= 4, W even, B <= 65535, else SKP_E_RANGE. */
int skp_conv3x3_small_f32(const void* x, const void* w, const void* bias, void* y, int B, int Cin, int Cout, int H, int W,
                          void* stream);
/* ... that also leaves the block statistics of its output behind for the GroupNorm that follows (the VAE's conv_in: Cin == 3,
 * Cout <= 256, H * W % 512 == 0): stats [B][Cout][blocks][2] = {mean, sum (y - mean)^2} per block of 512 consecutive pixels,
 * blocks = skp_conv3x3_small_stats_blocks() (0: not served). */
int skp_conv3x3_small_stats_blocks(int B, int Cin, int Cout, int H, int W);
int skp_conv3x3_small_stats_f32(const void* x, const void* w, const void* bias, void* y, float* stats, int B, int Cin, int Cout,
                                int H, int W, void* stream);

/* GEGLU of the transformer feed-forward (diffusers attention.GEGLU [third party], inside the hooked UNet forward):
 *   y[r, c] = p[r, c] * gelu(p[r, inner + c])    p: [rows, 2*inner], y: [rows, inner], exact (erf) gelu, inner % 4 == 0
 * _bwd writes dp [rows, 2*inner] = d loss / d p given dy [rows, inner]. */
int skp_geglu_fwd_f32(const float* p, float* y, int64_t rows, int inner, void* stream);
int skp_geglu_bwd_f32(const float* p, const float* dy, float* dp, int64_t rows, int inner, void* stream);

/* Layout changes around the transformer blocks (Transformer2DModel [third party]): y[b,p,c] = x[b,c,p] and back,
 * the way back adding the block's residual [B,C,HW] (may be NULL) in the same pass.  C % 4 == 0, HW % 4 == 0. */
int skp_nchw_to_tokens_f32(const float* x, float* y, int B, int C, int HW, void* stream);
int skp_tokens_to_nchw_f32(const float* t, const float* residual, float* y, int B, int C, int HW, void* stream);

/* Per-token statistics of a reduced map M [T,R,R] (eval.py:39-111, ptp_utils.py:95-108):
 *   argmax[j*T+t] (i32) = flat index (row*R+col) of the j-th masked maximum, j<num_subjects
 *                         (first index wins ties; radius 0.05*R masking between maxima)
 *   kl[t] = KL( normalised gaussian(sigma) at those maxima  ||  softmax_{R*R}(M[t]+eps) )
 *   entropy[t] = entropy of Categorical(probs = softmax_{R*R}(M[t]))   (ptp_utils.py:165-187 entropy_sort: probs
 *                renormalised, log of the probs clamped to [FLT_EPSILON, 1-FLT_EPSILON] as torch.distributions does)
 * kl and entropy may each be NULL (not computed). */
int skp_token_stats_f32(const float* M, int T, int R, int num_subjects, float sigma, float eps,
                        int32_t* argmax, float* kl, float* entropy, void* stream);

/* Token selection (ptp_utils.py:110-112 + 115-159), entirely on device:
 *   cand = first n_cand tokens of argsort(kl, ascending)  (`kl` is any per-token score: KL, entropy, or a fixed
 *          order; ties by index; NaN scores rank after every number, as torch.argsort does)
 *   sel  = furthest_point_sampling over cand using the arg-max locations of the
 *          TRANSFORMED map (argmax_t, i32 flat indices, grid side R)
 * cand: i64[n_cand], sel: i64[top_k].  Limits: T <= 1024, n_cand <= 64, 2 <= top_k <= n_cand. */
int skp_select_tokens(const float* kl, const int32_t* argmax_t, int T, int R, int n_cand, int top_k,
                      int64_t* cand, int64_t* sel, void* stream);
/* n images per launch: kl, argmax_t [n,T] -> cand [n,n_cand], sel [n,top_k] (one workgroup per image, same selection). */
int skp_select_tokens_batched(const float* kl, const int32_t* argmax_t, int n, int T, int R, int n_cand, int top_k,
                              int64_t* cand, int64_t* sel, void* stream);

/* Sharpening + equivariance losses and their unit gradients for the K selected tokens
 * (optimize.py:157-206):
 *   sharp = mean_{k,p} (M[sel[k],p] - G[k,p])^2,  G = mean_j gaussian at argmax[j*T+sel[k]]
 *   equiv = mean_{k,p} (M[sel[k],p] - unwarp(Mt[sel[k]], theta_inv)[p])^2
 * theta_inv [host] 6 floats = the 2x3 INVERSE affine (invertable_transform.py:77-84); the
 * warp is affine_grid + bilinear grid_sample, zeros padding, align_corners=False.
 * partial: [2,K,nchunk] sums of squares, nchunk = ceil(R*R/1024) (caller sums, divides by K*R*R);
 * g_sharp, g_eq_a: [K,R,R] written = d sharp / d M[sel], d equiv / d M[sel];
 * g_eq_b: [K,R,R] written = d equiv / d Mt[sel], computed as a gather over the bilinear footprints (a second
 *         kernel on the same stream; no atomics, bit-reproducible). */
int skp_losses_fwd_f32(const float* M, const float* Mt, const int64_t* sel, int K, int T, int R,
                       const int32_t* argmax, int num_subjects, float sigma,
                       const float* theta_inv /*[host] 6*/, float* partial, float* g_sharp,
                       float* g_eq_a, float* g_eq_b, void* stream);
/* ... with theta_inv [6] in DEVICE memory (read by the kernels, not at launch): the call can be captured in a hipGraph and replayed
 * with a new augmentation per step (stablekeypoints_amd/optimize.py: GraphedStep). */
int skp_losses_fwd_dev_f32(const float* M, const float* Mt, const int64_t* sel, int K, int T, int R,
                           const int32_t* argmax, int num_subjects, float sigma, const float* theta_inv_dev,
                           float* partial, float* g_sharp, float* g_eq_a, float* g_eq_b, void* stream);

/* dst[sel[k], :] += a * x[k, :] + b * y[k, :]   (y may be NULL); rows of length n; sel i64[K] distinct. */
int skp_rows_axpy_f32(float* dst, const int64_t* sel, int K, int64_t n, const float* x, const float* a,
                      const float* y, const float* b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SKP_H */
