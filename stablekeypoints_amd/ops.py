"""Torch-facing wrappers of the HIP kernels (libskp_hip.so via ctypes).

PyTorch is used here only for device memory, streams and autograd bookkeeping; every op below
launches hand-written gfx950 kernels through the C ABI of include/skp.h on torch's CURRENT
stream.  No op has an eager/CPU fallback: a non-CUDA tensor raises.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import _native as N


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the StableKeypoints hot path runs on the HIP kernels only "
                           f"(got a {t.device} tensor; there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# low-res logits (fp32 MFMA)                                            ptp_utils.py:483-493
# ---------------------------------------------------------------------------------------------
def qk_logits(q: torch.Tensor, k: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """S[b,h,p,t] = scale*log2(e) * <q[b,p,h,:], k[bk,t,h,:]>;  q [B,s2,C], k [Bk,T,C] -> [B,H,s2,NT]
    (token-contiguous, NT = 16*ceil(T/16), pad columns zero)."""
    q, k = _dev(q, "q"), _dev(k, "k")
    B, s2, C = q.shape
    Bk, T, Ck = k.shape
    if Ck != C or C % heads:
        raise RuntimeError("qk_logits: channel mismatch")
    S = torch.empty(B, heads, s2, (T + 15) // 16 * 16, device=q.device, dtype=torch.float32)
    N.check(N.lib().skp_qk_logits_f32(q.data_ptr(), k.data_ptr(), S.data_ptr(), B, Bk, heads, T, s2,
                                      C // heads, float(scale), _stream()), "skp_qk_logits_f32")
    return S


def _gemm_nt(A, B, Cout, M, Nn, K, Z0, Z1, sa, sb, sc, alpha):
    N.check(N.lib().skp_gemm_nt_f32(A.data_ptr(), B.data_ptr(), Cout.data_ptr(), M, Nn, K, Z0, Z1,
                                    *sa, *sb, *sc, float(alpha), _stream()), "skp_gemm_nt_f32")


TOKEN_GROUP = 128          # most tokens one fused-map launch can hold (one softmax row per lane, in registers)
GROUP_STEP = 96            # group width when T > TOKEN_GROUP (NT=96 instantiation: no register spills)


def _groups(T: int):
    return [(t0, min(T, t0 + GROUP_STEP)) for t0 in range(0, T, GROUP_STEP)]


def _log2sumexp2(xs: List[torch.Tensor]) -> torch.Tensor:
    st = torch.stack(xs, 0)
    m = st.max(dim=0).values
    return m + torch.log2(torch.exp2(st - m).sum(dim=0))


MAP_WIDE = True          # False (tests): token groups x two passes for T > 128
MAP_WIDE_MAX_T = 1024


def map_wide_supported(T: int, R: int, sides: Sequence[int] = (16,)) -> bool:
    """The one-pass wide kernel serves T in (128, 1024] where its launch plan fits (the library's own gate: layer side
    <= 64, R % 32 == 0, tile count, LDS); everything else takes the two-pass token-group route."""
    if not (MAP_WIDE and TOKEN_GROUP < T <= MAP_WIDE_MAX_T and R % 32 == 0):
        return False
    si, _keep = N.int_array([int(v) for v in sides])
    return bool(N.lib().skp_attn_map_fwd_wide_ok(si, len(sides), int(T), int(R)))


def _map_fwd(S: Sequence[torch.Tensor], sides: Sequence[int], B: int, H: int, T: int, R: int, tokrow=None, n_rows: int = 0):
    """-> (M [B,T,R,R], lse [B,L*H,R*R] log2-sum-exp over all tokens).  T <= 128: all tokens of a pixel in one lane;
    more tokens: ONE pass with the token axis in 64-token slices across lanes (csrc/skp_attn_map_wide.hip); shapes that
    kernel does not take run token groups in two passes (group statistics -> combine -> apply).
    `tokrow` (int32 [T] on the device: output row of a token or -1) restricts the written maps to `n_rows` rows -- wide
    kernel only."""
    dev = S[0].device
    L, ldt = len(S), S[0].shape[-1]
    si, _k2 = N.int_array(sides)
    lib, st = N.lib(), _stream()
    if map_wide_supported(T, R, sides):
        M = torch.empty(B, n_rows if tokrow is not None else T, R, R, device=dev, dtype=torch.float32)
        lse = torch.empty(B, L * H, R * R, device=dev, dtype=torch.float32)
        sp, _k1 = N.ptr_array([t.data_ptr() for t in S])
        N.check(lib.skp_attn_map_fwd_wide_f32(sp, si, L, B, H, T, R, M.data_ptr(), lse.data_ptr(),
                                              tokrow.data_ptr() if tokrow is not None else None, int(n_rows), ldt, st),
                "skp_attn_map_fwd_wide_f32")
        return M, lse
    if tokrow is not None:
        raise RuntimeError("_map_fwd: row selection is served by the wide-token kernel only")
    M = torch.empty(B, T, R, R, device=dev, dtype=torch.float32)

    def launch(t0, t1, mode, lse_out, lse_in):
        sp, _k1 = N.ptr_array([t.data_ptr() + 4 * t0 for t in S])
        N.check(lib.skp_attn_map_fwd_ex_f32(sp, si, L, B, H, t1 - t0, R, M.data_ptr() + 4 * t0 * R * R,
                                            lse_out.data_ptr() if lse_out is not None else None,
                                            lse_in.data_ptr() if lse_in is not None else None,
                                            ldt, T * R * R, mode, st), "skp_attn_map_fwd_ex_f32")

    if T <= TOKEN_GROUP:
        lse = torch.empty(B, L * H, R * R, device=dev, dtype=torch.float32)
        launch(0, T, 0, lse, None)
        return M, lse
    parts = []
    for t0, t1 in _groups(T):
        parts.append(torch.empty(B, L * H, R * R, device=dev, dtype=torch.float32))
        launch(t0, t1, 1, parts[-1], None)
    lse = _log2sumexp2(parts)
    for t0, t1 in _groups(T):
        launch(t0, t1, 2, None, lse)
    return M, lse


def _map_bwd(S, dS, sides, B, H, T, R, dM, lse):
    dev = dM.device
    L, ldt = len(S), S[0].shape[-1]
    si, _k3 = N.int_array(sides)
    lib, st = N.lib(), _stream()
    nbytes = lib.skp_attn_map_bwd_workspace(si, L, B, H, T if T <= TOKEN_GROUP else GROUP_STEP, R)
    if nbytes < 0:
        N.check(int(nbytes), "skp_attn_map_bwd_workspace")
    ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)

    def launch(t0, t1, mode, dot):
        sp, _k1 = N.ptr_array([t.data_ptr() + 4 * t0 for t in S])
        dp, _k2 = N.ptr_array([t.data_ptr() + 4 * t0 for t in dS])
        N.check(lib.skp_attn_map_bwd_ex_f32(sp, dp, si, L, B, H, t1 - t0, R, dM.data_ptr() + 4 * t0 * R * R,
                                            lse.data_ptr(), ws.data_ptr(), dot.data_ptr() if dot is not None else None,
                                            ldt, T * R * R, mode, st), "skp_attn_map_bwd_ex_f32")

    if T <= TOKEN_GROUP:
        launch(0, T, 0, None)
        return
    dots = []
    for t0, t1 in _groups(T):
        dots.append(torch.empty(B, L * H, R * R, device=dev, dtype=torch.float32))
        launch(t0, t1, 1, dots[-1])
    dot = torch.stack(dots, 0).sum(dim=0)
    for t0, t1 in _groups(T):
        launch(t0, t1, 2, dot)


# Which map backward the fused step takes.  The losses give the map gradient as K selected rows per batch row; "auto" hands
# it over in that form (one map+losses node, ops.MapLossesFn) wherever a sparse kernel is the faster route:
#   the column sweep (csrc/skp_attn_map_col.hip, round 4: vertical adjoint in a register window, horizontal adjoint on complete
#   low-res rows, no dV staging) wherever it serves the shapes (R in {128, 256}, R / s in {4, 8}, K <= 16, any T <= 1024):
#   377 us against 541 us for the dense kernels at T = 77, 1.56 ms against 2.47 ms for the token-major sweep at T = 500;
#   other shapes: T <= 128 the dense-gradient kernels (the token-major sweep is slower there: 636 us), T > 128 the
#   token-major sweep (2.4 vs 7.9 ms at T = 500, B = 8, for token groups x two passes of the dense kernels).
# "col" / "sweep" force one sparse kernel where it serves the shapes, "sparse" = either, "dense" forces the dense route.
MAP_BWD_MODE = "auto"
MAP_SPARSE_MAX_SIDE, MAP_SPARSE_MAX_K, MAP_SPARSE_MAX_R = 32, 32, 1024     # limits of csrc/skp_attn_map_tok.hip
COL_MAX_T = 1024                                                             # token counts the column sweep is taken for


def map_bwd_col_supported(sides, K: int, R: int, T: int, heads: int = 8) -> bool:
    si, _keep = N.int_array([int(v) for v in sides])
    return bool(N.lib().skp_attn_map_bwd_col_ok(si, len(sides), int(heads), int(T), int(R), int(K)))


def map_bwd_sparse_supported(sides, K: int, R: int, T: int = 10 ** 9, heads: int = 8) -> bool:
    if MAP_BWD_MODE == "dense":
        return False
    col = MAP_BWD_MODE != "sweep" and T <= COL_MAX_T and map_bwd_col_supported(sides, K, R, T, heads)
    sweep = (MAP_BWD_MODE != "col" and max(sides) <= MAP_SPARSE_MAX_SIDE and 1 <= K <= MAP_SPARSE_MAX_K
             and R <= MAP_SPARSE_MAX_R)
    if MAP_BWD_MODE == "auto":
        return col or (sweep and T > TOKEN_GROUP)     # at T <= 128 the token-major sweep is slower than the dense route
    return col or sweep


def _map_bwd_sparse(S, sides, B, H, T, R, sel, G, lse):
    """dS[l] for a map gradient that is non-zero on the rows sel[b,:] of batch row b only (G [B,K,R,R] = those rows):
    token-major sweep, any T, no token groups, no dV staging (csrc/skp_attn_map_tok.hip)."""
    L, ldt, K = len(S), S[0].shape[-1], sel.shape[1]
    sel = sel.to(torch.int64).contiguous()
    G = _dev(G, "G")
    dS = [torch.empty_like(s_) for s_ in S]
    si, _k0 = N.int_array(sides)
    lib = N.lib()
    if MAP_BWD_MODE != "sweep" and T <= COL_MAX_T and map_bwd_col_supported(sides, K, R, T, H):
        nbytes = lib.skp_attn_map_bwd_col_workspace(si, L, B, H, T, R, K)
        if nbytes < 0:
            N.check(int(nbytes), "skp_attn_map_bwd_col_workspace")
        ws = torch.empty((nbytes + 3) // 4, device=G.device, dtype=torch.float32)
        sp, _k1 = N.ptr_array([t.data_ptr() for t in S])
        dp, _k2 = N.ptr_array([t.data_ptr() for t in dS])
        N.check(lib.skp_attn_map_bwd_col_f32(sp, dp, si, L, B, H, T, R, sel.data_ptr(), G.data_ptr(), K, lse.data_ptr(),
                                             ws.data_ptr(), ldt, _stream()), "skp_attn_map_bwd_col_f32")
        if ldt > (T + 15) // 16 * 16:
            for d_ in dS:
                d_[..., (T + 15) // 16 * 16:] = 0
        return dS
    nbytes = lib.skp_attn_map_bwd_sparse_workspace(si, L, B, H, T, R, K)
    if nbytes < 0:
        N.check(int(nbytes), "skp_attn_map_bwd_sparse_workspace")
    ws = torch.empty((nbytes + 3) // 4, device=G.device, dtype=torch.float32)
    sp, _k1 = N.ptr_array([t.data_ptr() for t in S])
    dp, _k2 = N.ptr_array([t.data_ptr() for t in dS])
    N.check(lib.skp_attn_map_bwd_sparse_f32(sp, dp, si, L, B, H, T, R, sel.data_ptr(), G.data_ptr(), K, lse.data_ptr(),
                                            ws.data_ptr(), ldt, _stream()), "skp_attn_map_bwd_sparse_f32")
    if ldt > (T + 15) // 16 * 16:                              # gap columns of a wider logits buffer: never read, keep finite
        for d_ in dS:
            d_[..., (T + 15) // 16 * 16:] = 0
    return dS


class AttnMapFn(torch.autograd.Function):
    """M[b,t,:,:] = mean over (layer, head) of softmax_t(bicubic_R(scale * q_l k_l^T)).

    Replaces ptp_utils.py:513-538 + optimize.py:27-79 (see csrc/skp_attn_map.hip).
    apply(R, heads, scales(tuple), q_0, k_0, q_1, k_1, ...) with q_l [B,s_l^2,C_l], k_l [Bk,T,C_l].
    """

    @staticmethod
    def forward(ctx, R: int, heads: int, scales: Tuple[float, ...], *qk: torch.Tensor):
        L = len(qk) // 2
        qs = [_dev(qk[2 * i], "q") for i in range(L)]
        ks = [_dev(qk[2 * i + 1], "k") for i in range(L)]
        B, T = qs[0].shape[0], ks[0].shape[1]
        sides = []
        for q in qs:
            s = int(round(q.shape[1] ** 0.5))
            if s * s != q.shape[1]:
                raise RuntimeError("AttnMapFn: query length is not a square")
            sides.append(s)
        S = [qk_logits(q, k, heads, sc) for q, k, sc in zip(qs, ks, scales)]
        M, lse = _map_fwd(S, sides, B, heads, T, R)
        ctx.save_for_backward(lse, *qs, *ks, *S)
        ctx.meta = (R, heads, tuple(scales), tuple(sides), B, T, L)
        return M

    @staticmethod
    def backward(ctx, dM: torch.Tensor):
        R, H, scales, sides, B, T, L = ctx.meta
        saved = ctx.saved_tensors
        lse, qs, ks, S = saved[0], saved[1:1 + L], saved[1 + L:1 + 2 * L], saved[1 + 2 * L:]
        dM = _dev(dM, "dM")
        # pad columns (t >= T) of a LAST partial group are written as 0 by the kernels; zero-fill covers the
        # (never read) gap columns when T > 128 is not a multiple of 16
        dS = [torch.empty_like(s_) if T <= TOKEN_GROUP else torch.zeros_like(s_) for s_ in S]
        _map_bwd(S, dS, sides, B, H, T, R, dM, lse)
        return (None, None, None, *_dqk_from_dS(qs, ks, dS, scales, H, T, ctx.needs_input_grad[3:]))


def _dqk_from_dS(qs, ks, dS, scales, H: int, T: int, needs) -> List[torch.Tensor]:
    """dq_l = scale * dS_l k_l, dk_l = scale * dS_l^T q_l on the fp32-MFMA GEMM (deterministic; a shared k sums over the
    batch rows).  `needs[2l]`, `needs[2l+1]`: which gradients autograd asked for."""
    grads: List[torch.Tensor] = []
    for l in range(len(qs)):
        q, k, ds, sc = qs[l], ks[l], dS[l], scales[l]
        B, s2, C = q.shape
        d = C // H
        Bk = k.shape[0]
        NT = ds.shape[-1]
        dq = dk = None
        if needs[2 * l]:
            dq = torch.empty_like(q)       # dq[b,p,h*d+c] = sc * sum_t dS[b,h,p,t] k[bk,t,h*d+c]
            _gemm_nt(ds, k, dq, s2, d, T, B, H,
                     (H * s2 * NT, s2 * NT, NT, 1), (0 if Bk == 1 else T * C, d, 1, C),
                     (s2 * C, d, C), sc)
        if needs[2 * l + 1]:
            dkb = torch.empty(B, T, C, device=q.device, dtype=torch.float32)
            _gemm_nt(ds, q, dkb, T, d, s2, B, H,    # dk[b,t,h*d+c] = sc * sum_p dS[b,h,p,t] q[b,p,h*d+c]
                     (H * s2 * NT, s2 * NT, 1, NT), (s2 * C, d, 1, C), (T * C, d, C), sc)
            dk = dkb.sum(dim=0, keepdim=True) if (Bk == 1 and B > 1) else dkb
        grads += [dq, dk]
    return grads


def attn_map(qs: Sequence[torch.Tensor], ks: Sequence[torch.Tensor], heads: int, scales: Sequence[float],
             R: int) -> torch.Tensor:
    flat = []
    for q, k in zip(qs, ks):
        flat += [q, k]
    return AttnMapFn.apply(int(R), int(heads), tuple(float(s) for s in scales), *flat)


@torch.no_grad()
def attn_map_rows(qs: Sequence[torch.Tensor], ks: Sequence[torch.Tensor], heads: int, scales: Sequence[float], R: int,
                  indices: torch.Tensor) -> torch.Tensor:
    """[B,len(indices),R,R]: the maps of the tokens `indices` only (optimize.py:58-59 `data[:, :, :, indices]`; inference
    keeps K of the T maps).  With a wide token axis (T > 128) the kernel writes just those rows -- the softmax still runs
    over all T tokens; otherwise all T maps are written and gathered.  No autograd (inference path)."""
    T = ks[0].shape[1]
    idx = torch.as_tensor(indices, device=qs[0].device).long().reshape(-1)
    sides = [int(round(q.shape[1] ** 0.5)) for q in qs]
    if any(s_ * s_ != q.shape[1] or q.shape[0] != qs[0].shape[0] or k.shape[1] != T for s_, q, k in zip(sides, qs, ks)):
        raise RuntimeError("attn_map_rows: hooked layers disagree on batch rows / token count, or a query length is not a square")
    S = [qk_logits(q.detach(), k.detach(), heads, sc) for q, k, sc in zip(qs, ks, scales)]
    B = qs[0].shape[0]
    if map_wide_supported(T, R, sides):
        uniq, inv = torch.unique(idx, return_inverse=True)      # a token asked for twice is written once
        tokrow = torch.full((T,), -1, device=idx.device, dtype=torch.int32)
        tokrow[uniq] = torch.arange(uniq.numel(), device=idx.device, dtype=torch.int32)
        M, _ = _map_fwd(S, sides, B, heads, T, R, tokrow=tokrow, n_rows=int(uniq.numel()))
        return M if uniq.numel() == idx.numel() and bool((inv == torch.arange(idx.numel(), device=idx.device)).all()) else M[:, inv]
    M, _ = _map_fwd(S, sides, B, heads, T, R)
    return M[:, idx]


def materialize_probs(q: torch.Tensor, k: torch.Tensor, heads: int, scale: float, R: int) -> torch.Tensor:
    """Reference-layout tensor (B*h, R*R, T) of one hooked layer (ptp_utils.py:535-538), produced by
    the SAME fused kernel run one head at a time (L=1, H=1).  Compatibility/testing path only."""
    B, T = q.shape[0], k.shape[1]
    s = int(round(q.shape[1] ** 0.5))
    S = qk_logits(q, k, heads, scale)
    outs = []
    for h in range(heads):
        Sh = S[:, h:h + 1].contiguous()
        Mh, _ = _map_fwd([Sh], [s], B, 1, T, R)               # [B,T,R,R]
        outs.append(Mh.reshape(B, T, R * R).permute(0, 2, 1))
    return torch.stack(outs, dim=1).reshape(B * heads, R * R, T)


# ---------------------------------------------------------------------------------------------
# token statistics / selection                      eval.py:39-111, ptp_utils.py:86-159
# ---------------------------------------------------------------------------------------------
def token_stats(M: torch.Tensor, num_subjects: int = 1, sigma: float = 2.0, eps: float = 1e-5,
                want_kl: bool = True, want_entropy: bool = False):
    """-> (argmax int32 [num_subjects, T] flat indices, kl float32 [T] or None[, entropy float32 [T]])."""
    M = _dev(M.detach(), "M")
    T, R, R2 = M.shape
    if R != R2:
        raise RuntimeError("token_stats: map must be square")
    am = torch.empty(num_subjects, T, device=M.device, dtype=torch.int32)
    kl = torch.empty(T, device=M.device, dtype=torch.float32) if want_kl else None
    ent = torch.empty(T, device=M.device, dtype=torch.float32) if want_entropy else None
    N.check(N.lib().skp_token_stats_f32(M.data_ptr(), T, R, num_subjects, float(sigma), float(eps),
                                        am.data_ptr(), kl.data_ptr() if want_kl else None,
                                        ent.data_ptr() if want_entropy else None, _stream()),
            "skp_token_stats_f32")
    return (am, kl, ent) if want_entropy else (am, kl)


SELECT_MAX_CANDIDATES = 64     # SKP_SEL_MAXC / SKP_SEL_MAXT of csrc/skp_select_loss.hip
SELECT_MAX_TOKENS = 1024


def select_tokens(kl: torch.Tensor, argmax_t: torch.Tensor, R: int, n_cand: int, top_k: int):
    """-> (cand int64 [n_cand], sel int64 [top_k]); argmax_t = first-subject arg-max of the TRANSFORMED map."""
    T = kl.shape[0]
    cand = torch.empty(n_cand, device=kl.device, dtype=torch.int64)
    sel = torch.empty(top_k, device=kl.device, dtype=torch.int64)
    am = argmax_t.reshape(-1)[:T].contiguous()
    N.check(N.lib().skp_select_tokens(kl.data_ptr(), am.data_ptr(), T, R, n_cand, top_k, cand.data_ptr(),
                                      sel.data_ptr(), _stream()), "skp_select_tokens")
    return cand, sel


# ---------------------------------------------------------------------------------------------
# losses                                             optimize.py:157-206
# ---------------------------------------------------------------------------------------------
def invert_affine(theta) -> List[float]:
    """Inverse of the 2x3 affine [[a,b,tx],[c,d,ty]] (invertable_transform.py:77-84), closed form in fp64."""
    a, b, tx, c, d, ty = [float(v) for v in theta]
    det = a * d - b * c
    ia, ib, ic, id_ = d / det, -b / det, -c / det, a / det
    return [ia, ib, -(ia * tx + ib * ty), ic, id_, -(ic * tx + id_ * ty)]


class LossesFn(torch.autograd.Function):
    """(sharpening_loss, equivariance_loss) of optimize.py:157-206 for the selected tokens, fused with
    their gradients.  apply(M, Mt, sel, argmax, theta_inv(6 floats), sigma, num_subjects)."""

    @staticmethod
    def forward(ctx, M, Mt, sel, argmax, theta_inv, sigma, num_subjects):
        M, Mt = _dev(M, "M"), _dev(Mt, "Mt")
        T, R, _ = M.shape
        K = sel.shape[0]
        nchunk = (R * R + 1023) // 1024
        partial = torch.empty(2, K, nchunk, device=M.device, dtype=torch.float32)
        g_sharp = torch.empty(K, R, R, device=M.device, dtype=torch.float32)
        g_eq_a = torch.empty_like(g_sharp)
        g_eq_b = torch.empty_like(g_sharp)
        th, _keep = N.float_array(theta_inv)
        N.check(N.lib().skp_losses_fwd_f32(M.data_ptr(), Mt.data_ptr(), sel.data_ptr(), K, T, R, argmax.data_ptr(),
                                           int(num_subjects), float(sigma), th, partial.data_ptr(),
                                           g_sharp.data_ptr(), g_eq_a.data_ptr(), g_eq_b.data_ptr(), _stream()),
                "skp_losses_fwd_f32")
        sums = partial.sum(dim=(1, 2)) / float(K * R * R)
        ctx.save_for_backward(sel, g_sharp, g_eq_a, g_eq_b)
        ctx.shape = (T, R)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, go_sharp, go_equiv):
        sel, g_sharp, g_eq_a, g_eq_b = ctx.saved_tensors
        T, R = ctx.shape
        dev = g_sharp.device
        z = torch.zeros((), device=dev, dtype=torch.float32)
        gs = z if go_sharp is None else go_sharp.reshape(()).float().contiguous()
        ge = z if go_equiv is None else go_equiv.reshape(()).float().contiguous()
        dM = torch.zeros(T, R, R, device=dev, dtype=torch.float32)
        dMt = torch.zeros(T, R, R, device=dev, dtype=torch.float32)
        K, n = sel.shape[0], R * R
        N.check(N.lib().skp_rows_axpy_f32(dM.data_ptr(), sel.data_ptr(), K, n, g_sharp.data_ptr(), gs.data_ptr(),
                                          g_eq_a.data_ptr(), ge.data_ptr(), _stream()), "skp_rows_axpy_f32")
        N.check(N.lib().skp_rows_axpy_f32(dMt.data_ptr(), sel.data_ptr(), K, n, g_eq_b.data_ptr(), ge.data_ptr(),
                                          None, None, _stream()), "skp_rows_axpy_f32")
        return dM, dMt, None, None, None, None, None


def fused_losses(M, Mt, sel, argmax, theta, sigma: float, num_subjects: int = 1):
    """theta: the FORWARD 2x3 affine of this image (6 floats, row-major). -> (sharp, equiv)."""
    return LossesFn.apply(M, Mt, sel, argmax, invert_affine(theta), float(sigma), int(num_subjects))


MAP_LOSSES_BATCHED = True     # False (tests): statistics / selection per image, through meta["score_fn"]


class MapLossesFn(torch.autograd.Function):
    """One node for `maps -> selection -> losses` of a group of images (optimize.py:347-414 for n images x 2 views), so that
    the map backward sees the gradient as it is -- K selected rows per batch row -- instead of a dense [B,T,R,R] tensor:
        forward : logits -> fused maps (+ lse) -> per image: token scores, selection, both losses and their unit gradients
        backward: G = go_sharp * g_sharp + go_equiv * g_eq (K rows per batch row) -> sparse token-major map backward
                  (csrc/skp_attn_map_tok.hip) -> dq_l, dk_l on the fp32-MFMA GEMM.
    apply(meta, q_0, k_0, q_1, k_1, ...) with rows 0..n-1 = the images, n..2n-1 = their affine copies;
    meta = dict(R, heads, scales, thetas [n][6] (host) or theta_inv_dev [n,6] (device), sigma, num_subjects, strategy, n_cand,
    top_k, score_fn).
    Returns (sum_i sharp_i, sum_i equiv_i, sel [n,K] int64)."""

    @staticmethod
    def forward(ctx, meta, *qk: torch.Tensor):
        L = len(qk) // 2
        qs = [_dev(qk[2 * i], "q") for i in range(L)]
        ks = [_dev(qk[2 * i + 1], "k") for i in range(L)]
        R, H, scales = int(meta["R"]), int(meta["heads"]), tuple(float(v) for v in meta["scales"])
        B, T = qs[0].shape[0], ks[0].shape[1]
        n = B // 2
        th_dev = meta.get("theta_inv_dev")          # [n, 6] float32 on the device, instead of meta["thetas"] (host numbers)
        if th_dev is not None and (not th_dev.is_cuda or th_dev.dtype != torch.float32 or tuple(th_dev.shape) != (n, 6) or not th_dev.is_contiguous()):
            raise RuntimeError("MapLossesFn: theta_inv_dev must be a contiguous float32 [n, 6] device tensor")
        if B != 2 * n or len(scales) != L or (th_dev is None and len(meta["thetas"]) != n):
            raise RuntimeError("MapLossesFn: rows must be n images followed by their n affine copies, one scale per layer")
        sides = []
        for q, k in zip(qs, ks):
            s_ = int(round(q.shape[1] ** 0.5))
            if s_ * s_ != q.shape[1] or q.shape[0] != B or k.shape[1] != T or q.shape[2] % H or k.shape[2] != q.shape[2]:
                raise RuntimeError("MapLossesFn: hooked layers disagree on batch rows / token count / head split")
            sides.append(s_)
        S = [qk_logits(q, k, H, sc) for q, k, sc in zip(qs, ks, scales)]
        M, lse = _map_fwd(S, sides, B, H, T, R)
        sigma, ns = float(meta["sigma"]), int(meta["num_subjects"])
        n_cand = min(int(meta["n_cand"]), T)
        K = min(int(meta["top_k"]), n_cand)
        dev = M.device
        sel_all = torch.empty(n, K, device=dev, dtype=torch.int64)
        g_sharp = torch.empty(n, K, R, R, device=dev, dtype=torch.float32)
        g_eq_a, g_eq_b = torch.empty_like(g_sharp), torch.empty_like(g_sharp)
        nchunk = (R * R + 1023) // 1024
        partial = torch.empty(n, 2, K, nchunk, device=dev, dtype=torch.float32)
        lib, st = N.lib(), _stream()
        # token statistics and the selection of ALL images in three launches (one workgroup per (image, token) / per image: the
        # per-image launches were 77 workgroups of 40 us each, twelve of them per step); per-token results are what the
        # per-image calls give, bit for bit
        strategy = meta["strategy"]
        # (a caller-supplied score function is honoured: only the stock order of optimize.token_order is batched)
        batched = (strategy in ("gaussian", "entropy", "consistent") and T <= SELECT_MAX_TOKENS and n_cand <= SELECT_MAX_CANDIDATES
                   and K >= 2 and MAP_LOSSES_BATCHED and getattr(meta.get("score_fn"), "_skp_stock_order", False))
        if batched:
            flat = M.reshape(B * T, R, R)
            st_all = token_stats(flat[:n * T], num_subjects=ns, sigma=sigma, want_kl=strategy == "gaussian",
                                 want_entropy=strategy == "entropy")
            am_all = st_all[0].reshape(ns, n, T).permute(1, 0, 2).contiguous()                  # [n, ns, T]
            if strategy == "gaussian":
                score_all = st_all[1].reshape(n, T)
            elif strategy == "entropy":
                score_all = st_all[2].reshape(n, T)
            else:
                score_all = torch.arange(T, device=dev, dtype=torch.float32).repeat(n, 1)
            amt_all, _ = token_stats(flat[n * T:], num_subjects=1, sigma=sigma, want_kl=False)  # [1, n*T]
            cand_all = torch.empty(n, n_cand, device=dev, dtype=torch.int64)
            N.check(lib.skp_select_tokens_batched(score_all.contiguous().data_ptr(), amt_all.data_ptr(), n, T, R, n_cand, K,
                                                  cand_all.data_ptr(), sel_all.data_ptr(), st), "skp_select_tokens_batched")
        for i in range(n):
            if batched:
                am = am_all[i]
            else:
                am, score = meta["score_fn"](M[i], strategy, ns, sigma)
                am_t, _ = token_stats(M[n + i], num_subjects=1, sigma=sigma, want_kl=False)
                _, sel_i = select_tokens(score, am_t[0], R, n_cand, K)
                sel_all[i].copy_(sel_i)
            if th_dev is not None:                   # inverse affines in device memory (a captured step: new numbers per replay)
                N.check(lib.skp_losses_fwd_dev_f32(M[i].data_ptr(), M[n + i].data_ptr(), sel_all[i].data_ptr(), K, T, R,
                                                   am.data_ptr(), ns, sigma, th_dev[i].data_ptr(), partial[i].data_ptr(),
                                                   g_sharp[i].data_ptr(), g_eq_a[i].data_ptr(), g_eq_b[i].data_ptr(), st),
                        "skp_losses_fwd_dev_f32")
                continue
            th, _keep = N.float_array(invert_affine(meta["thetas"][i]))
            N.check(lib.skp_losses_fwd_f32(M[i].data_ptr(), M[n + i].data_ptr(), sel_all[i].data_ptr(), K, T, R,
                                           am.data_ptr(), ns, sigma, th, partial[i].data_ptr(), g_sharp[i].data_ptr(),
                                           g_eq_a[i].data_ptr(), g_eq_b[i].data_ptr(), st), "skp_losses_fwd_f32")
        sums = partial.sum(dim=(0, 2, 3)) / float(K * R * R)
        ctx.save_for_backward(lse, sel_all, g_sharp, g_eq_a, g_eq_b, *qs, *ks, *S)
        ctx.meta = (R, H, scales, tuple(sides), B, T, L)
        ctx.mark_non_differentiable(sel_all)
        return sums[0], sums[1], sel_all

    @staticmethod
    def backward(ctx, go_sharp, go_equiv, _go_sel):
        R, H, scales, sides, B, T, L = ctx.meta
        saved = ctx.saved_tensors
        lse, sel_all, g_sharp, g_eq_a, g_eq_b = saved[:5]
        qs, ks, S = saved[5:5 + L], saved[5 + L:5 + 2 * L], saved[5 + 2 * L:]
        z = torch.zeros((), device=lse.device)
        gs = z if go_sharp is None else go_sharp.reshape(()).float()
        ge = z if go_equiv is None else go_equiv.reshape(()).float()
        G = torch.cat([gs * g_sharp + ge * g_eq_a, ge * g_eq_b], dim=0)            # [B,K,R,R]
        dS = _map_bwd_sparse(S, sides, B, H, T, R, torch.cat([sel_all, sel_all], dim=0), G, lse)
        return (None, *_dqk_from_dS(qs, ks, dS, scales, H, T, ctx.needs_input_grad[1:]))


UNWARP_MAX_K = 32


@torch.no_grad()
def unwarp_accumulate(maps: torch.Tensor, theta_inv: torch.Tensor, size: int, finish: bool = True):
    """Tail of the augmented inference (eval.py:239-353): maps [n,K,R,R] of n views, theta_inv [n,2,3] their inverse
    affines -> (sum_v unwarp(bilinear_{R->size}(maps[v])) [K,size,size], coverage count [size,size]); with `finish` the
    first is already sum / count with 0/0 -> 0.  One kernel (csrc/skp_unwarp.hip); K > 32 runs in chunks of 32 tokens."""
    maps = _dev(maps, "maps")
    n, K, R, _ = maps.shape
    th = theta_inv.to(device=maps.device, dtype=torch.float32).reshape(n, 6).contiguous()
    tot = torch.empty(K, size, size, device=maps.device, dtype=torch.float32)
    num = torch.empty(size, size, device=maps.device, dtype=torch.float32)
    for k0 in range(0, K, UNWARP_MAX_K):
        k1 = min(K, k0 + UNWARP_MAX_K)
        chunk = maps if (k0 == 0 and k1 == K) else maps[:, k0:k1].contiguous()
        N.check(N.lib().skp_unwarp_accumulate_f32(chunk.data_ptr(), th.data_ptr(), n, k1 - k0, R, size, tot[k0:k1].data_ptr(),
                                                  num.data_ptr(), 1 if finish else 0, _stream()), "skp_unwarp_accumulate_f32")
    return tot, num


# ---------------------------------------------------------------------------------------------
# ordinary cross-attention core (short key axis)                 ptp_utils.py:493-506,540
# ---------------------------------------------------------------------------------------------
CROSS_ATTN_HEAD_DIMS = (8, 16, 32, 40, 64, 80, 160)
CROSS_ATTN_MAX_T = 128


def cross_attn_supported(C: int, heads: int, T: int) -> bool:
    """Every key count is served: T <= 128 by the in-register softmax kernel, more tokens (the reference default is
    --num_tokens 500, main.py:77-79) by the key-tiled online-softmax (flash) kernels."""
    return C % heads == 0 and (C // heads) in CROSS_ATTN_HEAD_DIMS


class CrossAttnFn(torch.autograd.Function):
    """out[b,n,:] = merge_heads(softmax(scale q k^T) v); q [B,N,C], k,v [Bk,T,C] (Bk in {1,B}) -> [B,N,C]."""

    @staticmethod
    def forward(ctx, q, k, v, heads: int, scale: float):
        q, k, v = _dev(q, "q"), _dev(k, "k"), _dev(v, "v")
        B, Nq, C = q.shape
        Bk, T, _ = k.shape
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, Nq, device=q.device, dtype=torch.float32)
        N.check(N.lib().skp_cross_attn_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                               B, Bk, heads, Nq, T, C // heads, float(scale), _stream()),
                "skp_cross_attn_fwd_f32")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta = (heads, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        dout = _dev(dout, "dout")
        B, Nq, C = q.shape
        Bk, T, _ = k.shape
        dq = torch.empty_like(q)
        dk = torch.empty(B, T, C, device=q.device, dtype=torch.float32)
        dv = torch.empty_like(dk)
        nbytes = N.lib().skp_cross_attn_bwd_workspace(B, heads, Nq, T, C // heads)
        ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
        N.check(N.lib().skp_cross_attn_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                               lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(),
                                               B, Bk, heads, Nq, T, C // heads, scale, _stream()),
                "skp_cross_attn_bwd_f32")
        if Bk == 1 and B > 1:
            dk, dv = dk.sum(dim=0, keepdim=True), dv.sum(dim=0, keepdim=True)
        return dq, dk, dv, None, None


def cross_attention(q, k, v, heads: int, scale: float):
    if k.shape[1] > CROSS_ATTN_MAX_T:
        return FlashAttnFn.apply(q, k, v, int(heads), float(scale))
    return CrossAttnFn.apply(q, k, v, int(heads), float(scale))


# ---------------------------------------------------------------------------------------------
# fused GroupNorm (+offset) (+SiLU) of the frozen network's blocks
# ---------------------------------------------------------------------------------------------
def group_norm_supported(x: torch.Tensor, groups: int) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % groups == 0
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] * groups <= 65535)


class GroupNormSiLUFn(torch.autograd.Function):
    """y = act(GroupNorm(x + off[:, :, None, None])) with act = SiLU or identity; gamma/beta/off frozen."""

    @staticmethod
    def forward(ctx, x, off, gamma, beta, groups: int, eps: float, silu: bool, blocks=None):
        x = _dev(x, "x")
        Nn, C, Hh, Ww = x.shape
        off_c = _dev(off.reshape(Nn, C), "off") if off is not None else None
        y = torch.empty_like(x)
        mean = torch.empty(Nn, groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        if blocks is not None:                       # statistics from the producing convolution's epilogue
            bs, nblk, pix = blocks
            N.check(N.lib().skp_group_norm_fwd_blocks_f32(x.data_ptr(), off_c.data_ptr() if off_c is not None else None,
                                                          gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                          rstd.data_ptr(), bs.data_ptr(), int(nblk), int(pix), Nn, C, groups,
                                                          Hh * Ww, float(eps), int(silu), _stream()),
                    "skp_group_norm_fwd_blocks_f32")
        else:
            ws = torch.empty(Nn * groups * 64 * 3, device=x.device, dtype=torch.float32)
            N.check(N.lib().skp_group_norm_fwd_f32(x.data_ptr(), off_c.data_ptr() if off_c is not None else None,
                                                   gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr(), ws.data_ptr(), Nn, C, groups, Hh * Ww, float(eps),
                                                   int(silu), _stream()), "skp_group_norm_fwd_f32")
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x, off_c if off_c is not None else x.new_empty(0), gamma, beta, mean, rstd)
        ctx.meta = (groups, float(eps), bool(silu), off_c is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, off_c, gamma, beta, mean, rstd = ctx.saved_tensors
        groups, eps, silu, has_off = ctx.meta
        dy = _dev(dy, "dy")
        Nn, C, Hh, Ww = x.shape
        dx = torch.empty_like(x)
        ws = torch.empty(Nn * groups * 64 * 3, device=x.device, dtype=torch.float32)
        N.check(N.lib().skp_group_norm_bwd_f32(x.data_ptr(), off_c.data_ptr() if has_off else None, gamma.data_ptr(),
                                               beta.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               dx.data_ptr(), ws.data_ptr(), Nn, C, groups, Hh * Ww, eps, int(silu),
                                               _stream()), "skp_group_norm_bwd_f32")
        return dx, None, None, None, None, None, None, None


class GroupNormSiLUForkFn(torch.autograd.Function):
    """(y, x') = (act(GroupNorm(x + off)), x): the norm of a block whose input ALSO feeds the block's residual path
    (ResnetBlock2D: conv2(...) + x; Transformer2DModel: proj_out(...) + x).  x' is x itself; routing the residual path through
    it hands BOTH gradients of x to this backward, where the residual path's gradient is added inside the norm's
    input-gradient kernel (skp_group_norm_bwd_add_f32) instead of by an accumulation pass of autograd's."""

    @staticmethod
    def forward(ctx, x, off, gamma, beta, groups: int, eps: float, silu: bool, blocks=None):
        y = GroupNormSiLUFn.forward(ctx, x, off, gamma, beta, groups, eps, silu, blocks)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dxp):
        if dxp is None:
            return GroupNormSiLUFn.backward(ctx, dy)
        x, off_c, gamma, beta, mean, rstd = ctx.saved_tensors
        groups, eps, silu, has_off = ctx.meta
        Nn, C, Hh, Ww = x.shape
        if dy is None:
            return dxp, None, None, None, None, None, None, None
        dy, dxp = _dev(dy, "dy"), _dev(dxp, "dx'")
        dx = torch.empty_like(x)
        ws = torch.empty(Nn * groups * 64 * 3, device=x.device, dtype=torch.float32)
        N.check(N.lib().skp_group_norm_bwd_add_f32(x.data_ptr(), off_c.data_ptr() if has_off else None, gamma.data_ptr(),
                                                   beta.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   dx.data_ptr(), dxp.data_ptr(), ws.data_ptr(), Nn, C, groups, Hh * Ww, eps,
                                                   int(silu), _stream()), "skp_group_norm_bwd_add_f32")
        return dx, None, None, None, None, None, None, None


GN_FUSED_STATS = True
GN_FORK = True


def group_norm_silu(x, norm: torch.nn.GroupNorm, off=None, silu: bool = True):
    """GroupNorm(+offset)(+SiLU).  If `x` is the output of one of this library's convolutions that left block sums
    behind (`x._skp_blocks`, set by conv3x3_auto / conv3x3_s2), the statistics pass over `x` is skipped."""
    blocks = getattr(x, "_skp_blocks", None) if GN_FUSED_STATS else None
    if blocks is not None and (blocks[0].shape[0] != x.shape[0] or blocks[0].shape[1] != x.shape[1]
                               or blocks[1] * blocks[2] != x.shape[2] * x.shape[3]):
        blocks = None
    return GroupNormSiLUFn.apply(x, off, norm.weight, norm.bias, norm.num_groups, norm.eps, silu, blocks)


def group_norm_silu_fork(x, norm: torch.nn.GroupNorm, off=None, silu: bool = True):
    """(group_norm_silu(x), x') with x' = x for the block's residual path: see GroupNormSiLUForkFn.  Without a gradient to
    carry (or with the switch off) x' is x itself and nothing changes."""
    if not (GN_FORK and torch.is_grad_enabled() and x.requires_grad):
        return group_norm_silu(x, norm, off=off, silu=silu), x
    blocks = getattr(x, "_skp_blocks", None) if GN_FUSED_STATS else None
    if blocks is not None and (blocks[0].shape[0] != x.shape[0] or blocks[0].shape[1] != x.shape[1]
                               or blocks[1] * blocks[2] != x.shape[2] * x.shape[3]):
        blocks = None
    return GroupNormSiLUForkFn.apply(x, off, norm.weight, norm.bias, norm.num_groups, norm.eps, silu, blocks)


# ---------------------------------------------------------------------------------------------
# flash-style self-attention (long image-token sequences)          ptp_utils.py:493-506,540
# ---------------------------------------------------------------------------------------------
def self_attn_supported(C: int, heads: int) -> bool:
    return C % heads == 0 and (C // heads) in CROSS_ATTN_HEAD_DIMS


# Flash attention of the big self-attention layers (>= 1024 keys, d = 40 / 80: the 64^2 and 32^2 levels) runs on the BF16 matrix
# cores with THREE-TERM OPERAND SPLITS (csrc/skp_flash_attn_s.hip): fp32 in / out, fp32 softmax and accumulation, every operand of the
# tile products the exact sum of three bf16 terms, six products per fp32 product.  Error against fp64 0.16-0.87x the
# fp32-instruction kernels' on the shapes it serves (tests/test_round5_gpu.py), forward 1.43-1.87x their speed, backward (d = 40)
# 1.10x (profiles/r05_flash_split.md).  Round 6 made it the route of record; FLASH_SPLIT = False restores the fp32-instruction
# kernels everywhere (bench.py times that step beside the line as `f32_instr`).
FLASH_SPLIT = True
FLASH_SPLIT_MIN_KEYS = 1024
def flash_split_ok(B, Bk, heads, Nq, Nk, d) -> bool:
    return Nk >= FLASH_SPLIT_MIN_KEYS and bool(N.lib().skp_flash_attn_fwd_split_ok(B, Bk, heads, Nq, Nk, d))


def _flash_fwd_split(q, k, v, out, lse, heads, scale):
    B, Nq, C = q.shape
    Bk, Nk, _ = k.shape
    nbytes = N.lib().skp_flash_attn_fwd_split_workspace(B, Bk, heads, Nq, Nk, C // heads)
    ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
    N.check(N.lib().skp_flash_attn_fwd_split_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), ws.data_ptr(),
                                                 B, Bk, heads, Nq, Nk, C // heads, scale, _stream()), "skp_flash_attn_fwd_split_f32")


def _flash_bwd_split(q, k, v, out, dout, lse, dq, dk, dv, heads, scale):
    B, Nq, C = q.shape
    nbytes = N.lib().skp_flash_attn_bwd_split_workspace(B, B, heads, Nq, Nq, C // heads)
    ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
    N.check(N.lib().skp_flash_attn_bwd_split_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                                 dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, B, heads, Nq, Nq, C // heads,
                                                 float(scale), _stream()), "skp_flash_attn_bwd_split_f32")


def flash_attn_bwd_split(q, k, v, out, dout, lse, heads: int, scale: float):
    """Direct entry (tests / tools): (dq, dk, dv) of the split backward (self-attention shapes)."""
    q, k, v, out, dout, lse = (_dev(t_, "t") for t_ in (q, k, v, out, dout, lse))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _flash_bwd_split(q, k, v, out, dout, lse, dq, dk, dv, heads, scale)
    return dq, dk, dv


def flash_attn_fwd_split(q, k, v, heads: int, scale: float):
    """Direct entry (tests / tools): (out, lse) of the split forward."""
    q, k, v = _dev(q, "q"), _dev(k, "k"), _dev(v, "v")
    out = torch.empty_like(q)
    lse = torch.empty(q.shape[0], heads, q.shape[1], device=q.device, dtype=torch.float32)
    _flash_fwd_split(q, k, v, out, lse, heads, float(scale))
    return out, lse


class FlashAttnFn(torch.autograd.Function):
    """out = merge_heads(softmax(scale q k^T) v); q [B,N,C], k, v [Bk,Nk,C] with Bk in {1,B}; key-tiled online
    softmax, scores never materialised.  Self-attention is the Bk == B, Nk == N case."""

    @staticmethod
    def forward(ctx, q, k, v, heads: int, scale: float):
        q, k, v = _dev(q, "q"), _dev(k, "k"), _dev(v, "v")
        B, Nq, C = q.shape
        Bk, Nk, _ = k.shape
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, Nq, device=q.device, dtype=torch.float32)
        if FLASH_SPLIT and flash_split_ok(B, Bk, heads, Nq, Nk, C // heads):
            _flash_fwd_split(q, k, v, out, lse, heads, float(scale))
        else:
            N.check(N.lib().skp_flash_attn_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                   B, Bk, heads, Nq, Nk, C // heads, float(scale), _stream()),
                    "skp_flash_attn_fwd_f32")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta = (heads, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        dout = _dev(dout, "dout")
        B, Nq, C = q.shape
        Bk, Nk, _ = k.shape
        dq = torch.empty_like(q)
        dk = torch.empty(B, Nk, C, device=q.device, dtype=torch.float32)
        dv = torch.empty_like(dk)
        if FLASH_SPLIT and Nk >= FLASH_SPLIT_MIN_KEYS and N.lib().skp_flash_attn_bwd_split_ok(B, Bk, heads, Nq, Nk, C // heads):
            _flash_bwd_split(q, k, v, out, dout, lse, dq, dk, dv, heads, scale)
            return dq, dk, dv, None, None
        nbytes = N.lib().skp_flash_attn_bwd_workspace(B, Bk, heads, Nq, Nk, C // heads)
        if nbytes < 0:
            N.check(int(nbytes), "skp_flash_attn_bwd_workspace")
        ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
        N.check(N.lib().skp_flash_attn_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                               lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(),
                                               B, Bk, heads, Nq, Nk, C // heads, scale, _stream()), "skp_flash_attn_bwd_f32")
        if Bk == 1 and B > 1:
            dk, dv = dk.sum(dim=0, keepdim=True), dv.sum(dim=0, keepdim=True)
        return dq, dk, dv, None, None


SelfAttnFn = FlashAttnFn


def self_attention(q, k, v, heads: int, scale: float):
    return FlashAttnFn.apply(q, k, v, int(heads), float(scale))


class AddBiasResidualFn(torch.autograd.Function):
    """out = a + b + bias[None,:,None,None] in one pass (bias frozen); gradients pass straight through."""

    @staticmethod
    def forward(ctx, a, b, bias):
        a, b = _dev(a, "a"), _dev(b, "b")
        Nn, C, Hh, Ww = a.shape
        out = torch.empty_like(a)
        N.check(N.lib().skp_add_bias_residual_f32(a.data_ptr(), b.data_ptr(), bias.data_ptr(), out.data_ptr(), Nn, C,
                                                  Hh * Ww, _stream()), "skp_add_bias_residual_f32")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy, None


def add_bias_residual(a, b, bias):
    return AddBiasResidualFn.apply(a, b, bias)


# ---------------------------------------------------------------------------------------------------------
# residual add + LayerNorm of the transformer blocks, one pass per direction (csrc/skp_layer_norm.hip)
# ---------------------------------------------------------------------------------------------------------
ADD_LN = True


def add_layer_norm_supported(h, norm: torch.nn.LayerNorm) -> bool:
    return bool(ADD_LN and h.is_cuda and h.dtype == torch.float32 and len(norm.normalized_shape) == 1
                and norm.elementwise_affine and norm.bias is not None and h.shape[-1] == norm.normalized_shape[0]
                and not (norm.weight.requires_grad or norm.bias.requires_grad)
                and N.lib().skp_add_layer_norm_ok(int(h.shape[-1])))


class AddLayerNormFn(torch.autograd.Function):
    """(x, n) = (d + h, LayerNorm(d + h)); with d None: (h, LayerNorm(h)).  The first output carries the residual stream on,
    so the gradient that reaches it from its later uses is added inside the norm's input-gradient kernel instead of a
    separate accumulation pass; d and h both receive that sum."""

    @staticmethod
    def forward(ctx, d, h, weight, bias, eps):
        h = _dev(h, "h")
        C = h.shape[-1]
        rows = h.numel() // C
        n = torch.empty_like(h)
        stat = torch.empty(rows, 2, device=h.device, dtype=torch.float32)
        if d is not None:
            d = _dev(d, "d")
            x = torch.empty_like(h)
        else:
            x = h
        N.check(N.lib().skp_add_layer_norm_fwd_f32(d.data_ptr() if d is not None else None, h.data_ptr(), weight.data_ptr(),
                                                   bias.data_ptr(), x.data_ptr() if d is not None else None, n.data_ptr(),
                                                   stat.data_ptr(), rows, C, float(eps), _stream()), "skp_add_layer_norm_fwd_f32")
        ctx.save_for_backward(x, stat, weight)
        ctx.has_d = d is not None
        return x, n                                  # d None: x is the input itself (autograd hands back an alias)

    @staticmethod
    def backward(ctx, gx, gn):
        x, stat, weight = ctx.saved_tensors
        C = x.shape[-1]
        if gn is None:
            dx = gx
        else:
            gn = _dev(gn, "gn")
            gx = _dev(gx, "gx") if gx is not None else None
            dx = torch.empty_like(x)
            N.check(N.lib().skp_add_layer_norm_bwd_f32(gn.data_ptr(), gx.data_ptr() if gx is not None else None, x.data_ptr(),
                                                       stat.data_ptr(), weight.data_ptr(), dx.data_ptr(), x.numel() // C, C,
                                                       _stream()), "skp_add_layer_norm_bwd_f32")
        return (dx if ctx.has_d else None), dx, None, None, None


def add_layer_norm(d, h, norm: torch.nn.LayerNorm):
    """Returns (d + h, norm(d + h)); d may be None -> (h, norm(h))."""
    return AddLayerNormFn.apply(d, h, norm.weight, norm.bias, norm.eps)


# ---------------------------------------------------------------------------------------------------------
# 3x3 / stride 1 / pad 1 convolutions of the frozen blocks: Winograd F(2x2,3x3) on the fp32 matrix cores.
# ---------------------------------------------------------------------------------------------------------
def conv3x3_supported(x_shape, w_shape, need_grad=True):
    """Shapes the kernel takes (everything else stays on the library convolution)."""
    if len(w_shape) != 4 or tuple(w_shape[2:]) != (3, 3) or len(x_shape) != 4:
        return False
    co, ci = int(w_shape[0]), int(w_shape[1])
    if co % 32 or ci % 32:
        return False
    _, _, h, w = (int(v) for v in x_shape)
    # the kernels address each tensor with 32-bit byte offsets (< 2 GiB per launch); larger batches are launched in
    # batch chunks (_conv3x3_run), so the limit applies to one image
    return max(ci * h * w, co * h * w, 16 * ci * co) * 4 < 2 ** 31


def _wino_filters(weight, backward):
    """Transformed filter of a frozen weight, built once per (weight storage, version) and kept resident."""
    key = "_skp_wino_bwd" if backward else "_skp_wino_fwd"
    hit = getattr(weight, key, None)
    tag = (weight._version, weight.data_ptr())       # data_ptr: `weight.data = ...` / load_state_dict(assign=True)
    if hit is not None and hit[0] == tag and hit[1].device == weight.device:
        return hit[1]
    w = _dev(weight.detach(), "weight")
    co, ci = w.shape[:2]
    U = torch.empty(16 * co * ci, device=w.device, dtype=torch.float32)
    if backward:
        N.check(N.lib().skp_conv3x3_filter_f32(w.data_ptr(), U.data_ptr(), ci, co, 1, _stream()), "skp_conv3x3_filter_f32")
    else:
        N.check(N.lib().skp_conv3x3_filter_f32(w.data_ptr(), U.data_ptr(), co, ci, 0, _stream()), "skp_conv3x3_filter_f32")
    setattr(weight, key, (tag, U))
    return U


def _conv3x3_raw(x, U, bias, cout, variant=0, split=True, residual=None, out=None):
    B, ci, H, W = x.shape
    y = out if out is not None else torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    nbytes = N.lib().skp_conv3x3_workspace(B, ci, cout, H, W, int(variant)) if split else 0
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes else None
    N.check(N.lib().skp_conv3x3_f32(x.data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                    ws.data_ptr() if ws is not None else None, B, ci, cout, H, W, int(variant), _stream()),
            "skp_conv3x3_f32")
    return y


def _wino4_filters(weight, backward):
    """F(4x4,3x3) transformed filter of a frozen weight (36*Cin*Cout floats), built once and kept resident."""
    key = "_skp_wino4_bwd" if backward else "_skp_wino4_fwd"
    hit = getattr(weight, key, None)
    tag = (weight._version, weight.data_ptr())
    if hit is not None and hit[0] == tag and hit[1].device == weight.device:
        return hit[1]
    w = _dev(weight.detach(), "weight")
    co, ci = w.shape[:2]
    U = torch.empty(36 * co * ci, device=w.device, dtype=torch.float32)
    if backward:
        N.check(N.lib().skp_conv3x3_f4_filter_f32(w.data_ptr(), U.data_ptr(), ci, co, 1, _stream()), "skp_conv3x3_f4_filter_f32")
    else:
        N.check(N.lib().skp_conv3x3_f4_filter_f32(w.data_ptr(), U.data_ptr(), co, ci, 0, _stream()), "skp_conv3x3_f4_filter_f32")
    setattr(weight, key, (tag, U))
    return U


def _conv3x3_f4_raw(x, U, bias, cout, split=True, residual=None, out=None, stats=None):
    B, ci, H, W = x.shape
    y = out if out is not None else torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    if stats is not None:                            # unsplit launch that also leaves the output's block sums behind
        N.check(N.lib().skp_conv3x3_f4_stats_f32(x.data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                                 stats.data_ptr(), B, ci, cout, H, W, _stream()), "skp_conv3x3_f4_stats_f32")
        return y
    nbytes = N.lib().skp_conv3x3_f4_workspace(B, ci, cout, H, W) if split else 0
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes else None
    N.check(N.lib().skp_conv3x3_f4_f32(x.data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                       ws.data_ptr() if ws is not None else None, B, ci, cout, H, W, _stream()),
            "skp_conv3x3_f4_f32")
    return y


def _wino4r_filters(weight, backward):
    """The 9 taps of a frozen weight in MFMA operand order for the raw-filter form (9*Cin*Cout floats), built once."""
    key = "_skp_wino4r_bwd" if backward else "_skp_wino4r_fwd"
    hit = getattr(weight, key, None)
    tag = (weight._version, weight.data_ptr())
    if hit is not None and hit[0] == tag and hit[1].device == weight.device:
        return hit[1]
    w = _dev(weight.detach(), "weight")
    co, ci = w.shape[:2]
    R = torch.empty(9 * co * ci, device=w.device, dtype=torch.float32)
    if backward:
        N.check(N.lib().skp_conv3x3_f4r_filter_f32(w.data_ptr(), R.data_ptr(), ci, co, 1, _stream()), "skp_conv3x3_f4r_filter_f32")
    else:
        N.check(N.lib().skp_conv3x3_f4r_filter_f32(w.data_ptr(), R.data_ptr(), co, ci, 0, _stream()), "skp_conv3x3_f4r_filter_f32")
    setattr(weight, key, (tag, R))
    return R


def _conv3x3_f4r_raw(x, R, bias, cout, residual=None, out=None):
    """Small-spatial form (csrc/skp_conv_wino4.hip, skp_wino4r_*): raw taps + in-lane filter transform, input transform in the workspace."""
    B, ci, H, W = x.shape
    y = out if out is not None else torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    nbytes = N.lib().skp_conv3x3_f4r_workspace(B, ci, cout, H, W)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    N.check(N.lib().skp_conv3x3_f4r_f32(x.data_ptr(), R.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        residual.data_ptr() if residual is not None else None, y.data_ptr(), ws.data_ptr(),
                                        B, ci, cout, H, W, _stream()), "skp_conv3x3_f4r_f32")
    return y


def conv3x3_f4r_ok(x_shape, cout) -> bool:
    b, ci, h, w = (int(v) for v in x_shape)
    return CONV3X3_MODE == "f4" and bool(N.lib().skp_conv3x3_f4r_ok(b, ci, int(cout), h, w))


# "f4": Winograd F(4x4,3x3) where the shape allows (H, W % 4 == 0), F(2x2,3x3) otherwise; "f2": F(2x2,3x3) only;
# "lib": library convolution everywhere (A/B runs and the consistency test).  Default f4.
CONV3X3_MODE = "f4"


def conv3x3_f4_ok(x_shape, w_shape):
    """F(4x4,3x3) serves the launch: >= 32 tiles (one tile block of the transformed-filter kernels), or fewer on the
    raw-filter form where it is routed (>= 1280 channels on both sides: the 8^2 layers of a 1- or 2-row step -- config 3's
    per-rank shape -- ran on the library at 120 us per call incl. its layout transposes; the raw-filter kernel takes the
    same 55 us as at 8 rows, its idle tile lanes cost nothing extra)."""
    b, _, h, w = (int(v) for v in x_shape)
    if not (CONV3X3_MODE == "f4" and h % 4 == 0 and w % 4 == 0 and int(w_shape[0]) % 16 == 0 and int(w_shape[1]) % 16 == 0
            and 36 * int(w_shape[0]) * int(w_shape[1]) * 4 < 2 ** 31):
        return False
    return b * (h // 4) * (w // 4) >= 32 or conv3x3_f4r_ok(x_shape, int(w_shape[0]))


def conv3x3_wanted(x_shape, w_shape):
    """Supported AND enough tiles to fill the MFMA column blocks (layers with few workgroups are split over input
    channels inside the library call, so the channel counts do not matter here)."""
    if CONV3X3_MODE == "lib" or not conv3x3_supported(x_shape, w_shape):
        return False
    b, _, h, w = (int(v) for v in x_shape)
    return conv3x3_f4_ok(x_shape, w_shape) or b * ((h + 1) // 2) * ((w + 1) // 2) >= 128


def conv3x3_stats_blocks(x_shape, w_shape) -> int:
    """256-pixel blocks per image if the forward convolution of this shape can leave output statistics behind, else 0."""
    if not (GN_FUSED_STATS and conv3x3_f4_ok(x_shape, w_shape)):
        return 0
    b, ci, h, w = (int(v) for v in x_shape)
    co = int(w_shape[0])
    if max(ci, co) * h * w * 4 * b >= 2 ** 31:           # chunked launches: keep the plain path
        return 0
    return int(N.lib().skp_conv3x3_f4_stats_blocks(b, ci, co, h, w))


def _conv3x3_run(x, weight, backward, bias, residual, cout, stats=None):
    w_shape = (weight.shape[1], weight.shape[0], 3, 3) if backward else weight.shape
    f4 = conv3x3_f4_ok(x.shape, w_shape)
    B, ci, H, W = x.shape
    if f4 and stats is None and conv3x3_f4r_ok(x.shape, cout):          # small spatial size, many channels: raw-filter form
        return _conv3x3_f4r_raw(x, _wino4r_filters(weight, backward), bias, cout, residual=residual)
    U = _wino4_filters(weight, backward) if f4 else _wino_filters(weight, backward)
    run = _conv3x3_f4_raw if f4 else _conv3x3_raw
    per_image = max(ci, cout) * H * W * 4
    chunk = max(1, (2 ** 31 - 1) // per_image)           # rows per launch under the kernels' 2 GiB addressing limit
    if stats is not None:
        return _conv3x3_f4_raw(x, U, bias, cout, residual=residual, stats=stats)
    if B <= chunk:
        return run(x, U, bias, cout, residual=residual)
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        run(x[b0:b1], U, bias, cout, residual=None if residual is None else residual[b0:b1], out=y[b0:b1])
    return y


class Conv3x3Fn(torch.autograd.Function):
    """y = conv2d(x, weight, bias, stride 1, padding 1) (+ residual) with frozen weight/bias; dx is the same kernel
    run with the rotated, transposed filter; the residual's gradient is dy."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, stats=None):
        x = _dev(x, "x")
        ctx.weight = weight
        if residual is not None:
            residual = _dev(residual, "residual")
        return _conv3x3_run(x, weight, False, bias, residual, weight.shape[0], stats=stats)

    @staticmethod
    def backward(ctx, dy):
        dx = None
        if ctx.needs_input_grad[0]:
            w = ctx.weight
            co, ci = w.shape[:2]
            if conv3x3_wanted(dy.shape, (ci, co, 3, 3)):
                dx = _conv3x3_run(_dev(dy, "dy"), w, True, None, None, ci)
            else:       # too few tiles for these kernels: library backward-data
                dx = torch.nn.grad.conv2d_input((dy.shape[0], ci, dy.shape[2], dy.shape[3]), w, dy, padding=1)
        return dx, None, None, (dy if ctx.needs_input_grad[3] else None), None


def conv3x3(x, weight, bias=None, residual=None):
    if not conv3x3_supported(x.shape, weight.shape):
        raise ValueError(f"conv3x3: unsupported shape x {tuple(x.shape)} w {tuple(weight.shape)}")
    if weight.requires_grad or (bias is not None and bias.requires_grad):
        raise ValueError("conv3x3: the Winograd kernels serve FROZEN weights only (no weight/bias gradient)")
    return Conv3x3Fn.apply(x, weight, bias, residual)


GN_ONEPASS = True        # False (tests): producers leave block statistics behind even where the norm would not want them


def _stats_useful(cout: int, h: int, w: int, groups: int = 32, rows: int = 1) -> bool:
    """Block statistics in a convolution's epilogue only pay where the GroupNorm that follows does not hold its rows in
    registers anyway (the one-pass form computes exact statistics from the row it has loaded; the library says which)."""
    return not (GN_ONEPASS and cout % groups == 0 and N.lib().skp_group_norm_onepass_ok(int(rows), int(cout), int(groups), int(h * w)))


def conv3x3_auto(x, weight, bias=None, residual=None, want_stats=False):
    """The frozen blocks' 3x3 convolution (+ bias + residual): Winograd kernel where it is wanted, library
    convolution (and the fused bias+residual pass) otherwise."""
    frozen = not (weight.requires_grad or (bias is not None and bias.requires_grad))     # the kernels give no dW / db
    if frozen and x.is_cuda and x.dtype == torch.float32 and conv3x3_wanted(x.shape, weight.shape):
        nblk = conv3x3_stats_blocks(x.shape, weight.shape) if (want_stats and _stats_useful(weight.shape[0], x.shape[2], x.shape[3])) else 0
        if nblk:
            stats = torch.empty(x.shape[0], weight.shape[0], nblk, 2, device=x.device, dtype=torch.float32)
            y = Conv3x3Fn.apply(x, weight, bias, residual, stats)
            y._skp_blocks = (stats, nblk, 256)           # consumed by group_norm_silu (same tensor object only)
            return y
        return Conv3x3Fn.apply(x, weight, bias, residual)
    if (frozen and residual is None and x.is_cuda and x.dtype == torch.float32 and CONV3X3_MODE != "lib"
            and weight.shape[1] <= 4 and x.shape[3] % 2 == 0 and not (torch.is_grad_enabled() and x.requires_grad)):
        return conv3x3_small(x, weight, bias, want_stats=want_stats)   # conv_in layers: output-bandwidth bound, own VALU kernel
    if residual is None:
        return torch.nn.functional.conv2d(x, weight, bias, padding=1)
    y = torch.nn.functional.conv2d(x, weight, None, padding=1)
    if bias is not None and y.is_cuda and y.dtype == torch.float32 and (y.shape[2] * y.shape[3]) % 4 == 0:
        return add_bias_residual(residual, y, bias)
    return (y + bias[None, :, None, None] if bias is not None else y) + residual


# GroupNorm(+offset)+SiLU folded into the consuming Winograd convolution (forward only, single output-channel group):
# saves the norm's apply pass (one read + one write of the activation); SKP_GN_FOLD=0 for A/B runs.
GN_FOLD = True


def conv3x3_gn_fold_ok(x, norm: torch.nn.GroupNorm, weight, *others) -> bool:
    """`others`: every further tensor the folded call will read (offset, bias, residual).  The folded kernel is forward
    only: with autograd on, ANY operand that requires a gradient (input, norm affine, weight, offset, bias, residual) sends
    the block down the differentiable route."""
    if not (GN_FOLD and x.is_cuda and x.dtype == torch.float32):
        return False
    if torch.is_grad_enabled() and any(t_ is not None and t_.requires_grad
                                       for t_ in (x, weight, norm.weight, norm.bias, *others)):
        return False
    if weight.requires_grad or not conv3x3_f4_ok(x.shape, weight.shape) or not group_norm_supported(x, norm.num_groups):
        return False
    B, ci, H, W = x.shape
    co = int(weight.shape[0])
    chunk = max(1, (2 ** 31 - 1) // (max(ci, co) * H * W * 4))            # the launch shape the batch chunks will have
    rows = min(B, chunk)
    return bool(N.lib().skp_conv3x3_f4_gn_ok(rows, ci, co, H, W)) and (B % rows == 0 or bool(
        N.lib().skp_conv3x3_f4_gn_ok(B % rows, ci, co, H, W)))


@torch.no_grad()
def conv3x3_gn_silu(x, norm: torch.nn.GroupNorm, weight, off=None, bias=None, residual=None, want_stats=False):
    """conv3x3(silu(GroupNorm(x + off))) (+ bias) (+ residual) with the normalisation applied in the convolution's patch load
    (csrc/skp_conv_wino4.hip, GNF kernels).  Statistics come from x's producer when it left block sums behind."""
    x = _dev(x, "x")
    B, C, H, W = x.shape
    G, cout = norm.num_groups, int(weight.shape[0])
    off_c = _dev(off.reshape(B, C), "off") if off is not None else None
    bias = _dev(bias, "bias") if bias is not None else None
    residual = _dev(residual, "residual") if residual is not None else None
    if residual is not None and tuple(residual.shape) != (B, cout, H, W):
        raise RuntimeError("conv3x3_gn_silu: residual must have the output's shape")
    mean = torch.empty(B, G, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    coef = torch.empty(B, C, 2, device=x.device, dtype=torch.float32)
    blocks = getattr(x, "_skp_blocks", None) if GN_FUSED_STATS else None
    if blocks is not None and (blocks[0].shape[0] != B or blocks[0].shape[1] != C or blocks[1] * blocks[2] != H * W):
        blocks = None
    lib, st = N.lib(), _stream()
    if blocks is not None:
        bs, nblk, pix = blocks
        N.check(lib.skp_group_norm_coef_f32(None, off_c.data_ptr() if off_c is not None else None, norm.weight.data_ptr(),
                                            norm.bias.data_ptr(), mean.data_ptr(), rstd.data_ptr(), coef.data_ptr(),
                                            bs.data_ptr(), int(nblk), int(pix), None, B, C, G, H * W, float(norm.eps), st),
                "skp_group_norm_coef_f32")
    else:
        ws = torch.empty(B * G * 64 * 3, device=x.device, dtype=torch.float32)
        N.check(lib.skp_group_norm_coef_f32(x.data_ptr(), off_c.data_ptr() if off_c is not None else None,
                                            norm.weight.data_ptr(), norm.bias.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            coef.data_ptr(), None, 0, 0, ws.data_ptr(), B, C, G, H * W, float(norm.eps), st),
                "skp_group_norm_coef_f32")
    U = _wino4_filters(weight, False)
    y = torch.empty(B, cout, H, W, device=x.device, dtype=torch.float32)
    nblk = conv3x3_stats_blocks(x.shape, weight.shape) if (want_stats and _stats_useful(cout, H, W)) else 0
    stats = torch.empty(B, cout, nblk, 2, device=x.device, dtype=torch.float32) if nblk else None
    chunk = max(1, (2 ** 31 - 1) // (max(C, cout) * H * W * 4))          # rows per launch under the kernels' 2 GiB addressing limit
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        N.check(lib.skp_conv3x3_f4_gn_f32(x[b0:b1].data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          residual[b0:b1].data_ptr() if residual is not None else None, y[b0:b1].data_ptr(),
                                          stats[b0:b1].data_ptr() if stats is not None else None, coef[b0:b1].data_ptr(),
                                          b1 - b0, C, cout, H, W, st), "skp_conv3x3_f4_gn_f32")
    if stats is not None:
        y._skp_blocks = (stats, nblk, 256)
    return y


def conv3x3_small(x, weight, bias=None, want_stats=False):
    """3x3 / stride 1 / padding 1 convolution with <= 4 input channels (the conv_in layers), forward only.  `want_stats`: where the
    kernel serves it (the VAE's 3 -> 128 at image resolution) the output carries its block statistics for the GroupNorm that
    follows (`y._skp_blocks`, consumed by group_norm_silu / conv3x3_gn_silu): no separate pass over the 1 GB activation."""
    x, w = _dev(x.detach(), "x"), _dev(weight.detach(), "weight")
    B, ci, H, W = x.shape
    y = torch.empty(B, w.shape[0], H, W, device=x.device, dtype=torch.float32)
    bb = _dev(bias.detach(), "bias") if bias is not None else None
    nblk = int(N.lib().skp_conv3x3_small_stats_blocks(B, ci, w.shape[0], H, W)) if (want_stats and GN_FUSED_STATS) else 0
    if nblk:
        stats = torch.empty(B, w.shape[0], nblk, 2, device=x.device, dtype=torch.float32)
        N.check(N.lib().skp_conv3x3_small_stats_f32(x.data_ptr(), w.data_ptr(), bb.data_ptr() if bb is not None else None, y.data_ptr(),
                                                    stats.data_ptr(), B, ci, w.shape[0], H, W, _stream()), "skp_conv3x3_small_stats_f32")
        y._skp_blocks = (stats, nblk, 512)
        return y
    N.check(N.lib().skp_conv3x3_small_f32(x.data_ptr(), w.data_ptr(), bb.data_ptr() if bb is not None else None, y.data_ptr(),
                                          B, ci, w.shape[0], H, W, _stream()), "skp_conv3x3_small_f32")
    return y


def mfma_issue_rate(waves_per_simd: int = 1, iters: int = 20000, device=None) -> float:
    """Measurement aid: TFLOP/s of back-to-back independent fp32 MFMAs with `waves_per_simd` waves on every SIMD."""
    import ctypes
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    scratch = torch.empty(256 * 4 * 256, device=dev, dtype=torch.float32)
    out = ctypes.c_float(0.0)
    with torch.cuda.device(dev):
        N.check(N.lab().skp_probe_mfma_f32(int(waves_per_simd), int(iters), scratch.data_ptr(), ctypes.addressof(out), _stream()),
                "skp_probe_mfma_f32")
    return float(out.value)


_QKV_CACHE = {}


def _qkv_stack(wq, wk, wv):
    """[3, C, C] stack of the three frozen projection weights of a self-attention block (made once per block)."""
    return weight_stack([wq, wk, wv])


def weight_stack(ws):
    """[len(ws), N, K] stack of frozen [N, K] weights (~200 MB over SD-1.5's attention projections), one entry per list of
    parameters: keyed by the identity of the first one, re-made -- in place of the old entry, which is thereby released -- whenever
    a parameter's storage or version has moved (`.to()`, dtype change, `load_state_dict`), dropped with its model."""
    import weakref
    key = (id(ws[0]), len(ws))
    tag = tuple((w._version, w.data_ptr(), w.device, w.dtype) for w in ws)
    hit = _QKV_CACHE.get(key)
    if hit is not None and hit[0] == tag and hit[2]() is ws[0]:
        return hit[1]
    w3 = torch.stack([w.detach() for w in ws]).contiguous()
    for dead in [k_ for k_, v_ in _QKV_CACHE.items() if v_[2]() is None]:     # stacks of models that are gone
        del _QKV_CACHE[dead]
    _QKV_CACHE[key] = (tag, w3, weakref.ref(ws[0]))
    return w3


QKV_STACKED = True   # one batched GEMM with a broadcast A operand


def _qkv_forward(x, wq, wk, wv):
    """q, k, v as ONE strided-batched GEMM: A = x with batch stride 0, B_i = W_i^T, C = [3, M, C] -- three times the
    workgroups of a single projection in one launch and three contiguous outputs (no strided views for the attention
    kernels).  Bit-equal to three F.linear calls per library kernel choice; 8 rows: 220 -> 180 us at 64^2, 186 -> 165 us
    at 16^2, 2 rows: 79 -> 49 us (tools/qkv_probe.py)."""
    F = torch.nn.functional
    if not (QKV_STACKED and x.is_cuda and wq.shape == wk.shape == wv.shape):
        return F.linear(x, wq), F.linear(x, wk), F.linear(x, wv)
    c_in, c_out = wq.shape[1], wq.shape[0]
    x2 = x.reshape(-1, c_in)
    m = x2.shape[0]
    out = torch.bmm(x2.unsqueeze(0).expand(3, m, c_in), _qkv_stack(wq, wk, wv).transpose(1, 2))
    shp = (*x.shape[:-1], c_out)
    return out[0].view(shp), out[1].view(shp), out[2].view(shp)


class QKVProjFn(torch.autograd.Function):
    """q, k, v = x.Wq^T, x.Wk^T, x.Wv^T of a frozen self-attention block (bias-free projections, ptp_utils.py:513-520).
    Forward: one batched GEMM (_qkv_forward).  The input gradient accumulates inside the GEMMs (dx = dq.Wq, then two
    beta = 1 GEMMs) instead of three GEMMs and two add passes over [B, N, C]."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        ctx.save_for_backward(wq, wk, wv)
        return _qkv_forward(x, wq, wk, wv)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        wq, wk, wv = ctx.saved_tensors
        shp = dq.shape
        dx = torch.mm(dq.reshape(-1, shp[-1]), wq)
        dx.addmm_(dk.reshape(-1, shp[-1]), wk)
        dx.addmm_(dv.reshape(-1, shp[-1]), wv)
        return dx.reshape(*shp[:-1], wq.shape[1]), None, None, None


QKV_ACCUM = True


def qkv_proj(x, wq, wk, wv):
    """Self-attention projections; frozen bias-free weights take the accumulate-in-GEMM backward."""
    frozen = not (wq.requires_grad or wk.requires_grad or wv.requires_grad)
    if QKV_ACCUM and x.is_cuda and x.requires_grad and torch.is_grad_enabled() and frozen:
        return QKVProjFn.apply(x, wq, wk, wv)
    if frozen and not (x.requires_grad and torch.is_grad_enabled()):
        return _qkv_forward(x, wq, wk, wv)
    F = torch.nn.functional
    return F.linear(x, wq), F.linear(x, wk), F.linear(x, wv)


class SelfAttnQKVFn(torch.autograd.Function):
    """One self-attention block core with frozen bias-free projections (ptp_utils.py:513-520, 493-506, 540):
    out = merge_heads(softmax(scale q k^T) v), q | k | v = x.W^T as one batched GEMM.  Backward: the flash backward writes
    dq, dk, dv as column bands of ONE [B*N, 3C] buffer (skp_flash_attn_bwd_ld_f32), so dx is a single GEMM against the
    stacked weights [3C, C] instead of three accumulating ones (8 rows: 224 -> 174 us at 64^2, tools/qkv_probe.py)."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, heads, scale):
        q, k, v = _qkv_forward(x, wq, wk, wv)
        B, Nq, C = q.shape
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, Nq, device=q.device, dtype=torch.float32)
        if FLASH_SPLIT and flash_split_ok(B, B, heads, Nq, Nq, C // heads):
            _flash_fwd_split(q, k, v, out, lse, heads, float(scale))
        else:
            N.check(N.lib().skp_flash_attn_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                                   B, B, heads, Nq, Nq, C // heads, float(scale), _stream()), "skp_flash_attn_fwd_f32")
        ctx.save_for_backward(q, k, v, out, lse, wq, wk, wv)
        ctx.meta = (int(heads), float(scale), tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, wq, wk, wv = ctx.saved_tensors
        heads, scale, xshape = ctx.meta
        dout = _dev(dout, "dout")
        B, Nq, C = q.shape
        d3 = torch.empty(B * Nq, 3 * C, device=q.device, dtype=torch.float32)
        p = d3.data_ptr()
        if FLASH_SPLIT and Nq >= FLASH_SPLIT_MIN_KEYS and N.lib().skp_flash_attn_bwd_split_ok(B, B, heads, Nq, Nq, C // heads):
            nbytes = N.lib().skp_flash_attn_bwd_split_workspace(B, B, heads, Nq, Nq, C // heads)
            ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
            N.check(N.lib().skp_flash_attn_bwd_split_ld_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                                            lse.data_ptr(), p, p + 4 * C, p + 8 * C, ws.data_ptr(), B, B, heads, Nq, Nq,
                                                            C // heads, scale, 3 * C, _stream()), "skp_flash_attn_bwd_split_ld_f32")
            dx = torch.mm(d3, _qkv_stack(wq, wk, wv).view(3 * C, wq.shape[1]))
            return dx.view(xshape), None, None, None, None, None
        nbytes = N.lib().skp_flash_attn_bwd_workspace(B, B, heads, Nq, Nq, C // heads)
        if nbytes < 0:
            N.check(int(nbytes), "skp_flash_attn_bwd_workspace")
        ws = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
        N.check(N.lib().skp_flash_attn_bwd_ld_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(),
                                                  lse.data_ptr(), p, p + 4 * C, p + 8 * C, ws.data_ptr(), B, B, heads, Nq, Nq,
                                                  C // heads, scale, 3 * C, _stream()), "skp_flash_attn_bwd_ld_f32")
        dx = torch.mm(d3, _qkv_stack(wq, wk, wv).view(3 * C, wq.shape[1]))
        return dx.view(xshape), None, None, None, None, None


FA2_HEAD_DIMS = (40, 64, 80, 160)          # head sizes of the second-generation flash kernels (the *_ld entry serves these)


def self_attention_block(x, wq, wk, wv, heads: int, scale: float):
    """Self-attention core of a block whose q / k / v projections are frozen and bias-free: projections + attention, with
    the fused input gradient where the kernels serve the head size; the composition of `qkv_proj` and `self_attention`
    otherwise (no gradient wanted, other head sizes)."""
    frozen = not (wq.requires_grad or wk.requires_grad or wv.requires_grad)
    if (QKV_STACKED and QKV_ACCUM and x.is_cuda and x.dim() == 3 and frozen and x.requires_grad
            and torch.is_grad_enabled() and wq.shape == wk.shape == wv.shape and wq.shape[0] % heads == 0
            and (wq.shape[0] // heads) in FA2_HEAD_DIMS):
        return SelfAttnQKVFn.apply(_dev(x, "x"), wq, wk, wv, int(heads), float(scale))
    q, k, v = qkv_proj(x, wq, wk, wv)
    return self_attention(q, k, v, heads, scale)


def conv1x1_nobias(x, weight):
    """1x1 convolution without its bias as one batched GEMM over the NCHW planes: y[b] = W [Co,Ci] . x[b] [Ci, H*W]
    (the library convolution wraps the same product in NCHW<->NHWC transposes).  Autograd: dx[b] = W^T . dy[b]."""
    b, ci, h, w = x.shape
    w2 = weight.reshape(weight.shape[0], ci)
    return torch.matmul(w2, x.reshape(b, ci, h * w)).reshape(b, weight.shape[0], h, w)


# ---------------------------------------------------------------------------------------------------------
# 3x3 / stride 2 down-sampling convolutions of the frozen VAE encoder (forward only: they run under no_grad).
# ---------------------------------------------------------------------------------------------------------
def conv3x3_s2_supported(x, weight) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    b, ci, h, w = (int(v) for v in x.shape)
    co = int(weight.shape[0])
    return (CONV3X3_MODE != "lib" and int(weight.shape[1]) == ci and ci % 16 == 0 and co % 32 == 0 and h % 16 == 0
            and w % 32 == 0 and max(b * ci * h * w, b * co * (h // 2) * (w // 2), 9 * ci * co) * 4 < 2 ** 31)


S2_BWD_OWN = True


class ConvS2Fn(torch.autograd.Function):
    """Stride-2 3x3 convolution of a frozen UNet Downsample2D whose INPUT needs a gradient: forward on the direct MFMA
    kernel.  Input gradient: the transposed stride-2 convolution is the stride-1 convolution of the ZERO-STUFFED output
    gradient (dy_up[2i, 2j] = dy[i, j]) with the rotated, transposed filter -- i.e. the Winograd backward-data kernel of
    the stride-1 layers on a dy_up of the input's size (three quarters of its taps multiply zeros, which costs what the
    direct form's 4x multiplies would; no library call, no NCHW <-> NHWC transposes).  pad = 1 (UNet): out[i] reads
    x[2i - 1 + k] => dx[y] = sum_k w[k] dy_up[y + 1 - k]; pad = 0 (asymmetric (0,1,0,1) extension): x[2i + k] =>
    dy_up is shifted by one: dy_up[2i + 1, 2j + 1] = dy[i, j]."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        ctx.weight = weight
        ctx.xshape, ctx.pad = tuple(x.shape), int(pad)
        return _conv3x3_s2_raw(x, weight, bias, pad, False)

    @staticmethod
    def backward(ctx, dy):
        w = ctx.weight
        b, c, h, wd = ctx.xshape
        co = w.shape[0]
        if S2_BWD_OWN and h % 2 == 0 and wd % 2 == 0 and conv3x3_wanted((b, co, h, wd), (c, co, 3, 3)):
            dy = _dev(dy, "dy")
            up = torch.zeros(b, co, h, wd, device=dy.device, dtype=torch.float32)
            o = 0 if ctx.pad == 1 else 1
            up[:, :, o::2, o::2] = dy
            return _conv3x3_run(up, w, True, None, None, c), None, None, None
        if ctx.pad == 1:
            return torch.nn.grad.conv2d_input(ctx.xshape, w, dy.contiguous(), stride=2, padding=1), None, None, None
        dxp = torch.nn.grad.conv2d_input((b, c, h + 1, wd + 1), w, dy.contiguous(), stride=2, padding=0)   # pad 0 = (0,1,0,1) extension
        return dxp[:, :, :h, :wd], None, None, None


def conv3x3_s2(x, weight, bias=None, pad: int = 0, want_stats: bool = False):
    """y = conv2d(zero-extended x, weight, bias, stride 2): pad = 0 is F.pad(x, (0,1,0,1)) + padding 0 (the VAE's
    Downsample2D), pad = 1 is padding 1.  Frozen weights; an input that needs a gradient goes through ConvS2Fn."""
    if weight.requires_grad or (bias is not None and bias.requires_grad):
        raise RuntimeError("conv3x3_s2: frozen weights only")
    if torch.is_grad_enabled() and x.requires_grad:
        return ConvS2Fn.apply(x, weight, bias, int(pad))
    return _conv3x3_s2_raw(x, weight, bias, pad, want_stats)


def _conv3x3_s2_raw(x, weight, bias, pad, want_stats):
    x = _dev(x.detach(), "x")
    key = "_skp_s2"
    hit = getattr(weight, key, None)
    tag = (weight._version, weight.data_ptr())
    if hit is not None and hit[0] == tag and hit[1].device == weight.device:
        U = hit[1]
    else:
        w = _dev(weight.detach(), "weight")
        U = torch.empty(9 * w.shape[0] * w.shape[1], device=w.device, dtype=torch.float32)
        N.check(N.lib().skp_conv3x3_s2_filter_f32(w.data_ptr(), U.data_ptr(), w.shape[0], w.shape[1], _stream()),
                "skp_conv3x3_s2_filter_f32")
        setattr(weight, key, (tag, U))
    B, ci, H, W = x.shape
    co = weight.shape[0]
    y = torch.empty(B, co, H // 2, W // 2, device=x.device, dtype=torch.float32)
    if want_stats and GN_FUSED_STATS:
        nblk = (H // 16) * (W // 32)
        stats = torch.empty(B, co, nblk, 2, device=x.device, dtype=torch.float32)
        N.check(N.lib().skp_conv3x3_s2_stats_f32(x.data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 y.data_ptr(), stats.data_ptr(), B, ci, co, H, W, int(pad), _stream()),
                "skp_conv3x3_s2_stats_f32")
        y._skp_blocks = (stats, nblk, 128)
        return y
    nbytes = N.lib().skp_conv3x3_s2_workspace(B, ci, co, H, W)    # > 0: small grid, K split over the input channels
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes > 0 else None
    N.check(N.lib().skp_conv3x3_s2_ws_f32(x.data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                          ws.data_ptr() if ws is not None else None, B, ci, co, H, W, int(pad), _stream()),
            "skp_conv3x3_s2_ws_f32")
    return y


class GEGLUFn(torch.autograd.Function):
    """h * gelu(gate) on the feed-forward projection [.., 2*inner] (h = first half, gate = second half)."""

    @staticmethod
    def forward(ctx, proj):
        proj = _dev(proj, "proj")
        inner = proj.shape[-1] // 2
        rows = proj.numel() // (2 * inner)
        y = torch.empty(*proj.shape[:-1], inner, device=proj.device, dtype=torch.float32)
        N.check(N.lib().skp_geglu_fwd_f32(proj.data_ptr(), y.data_ptr(), rows, inner, _stream()), "skp_geglu_fwd_f32")
        ctx.save_for_backward(proj)
        return y

    @staticmethod
    def backward(ctx, dy):
        (proj,) = ctx.saved_tensors
        dy = _dev(dy, "dy")
        inner = proj.shape[-1] // 2
        dp = torch.empty_like(proj)
        N.check(N.lib().skp_geglu_bwd_f32(proj.data_ptr(), dy.data_ptr(), dp.data_ptr(), proj.numel() // (2 * inner), inner,
                                          _stream()), "skp_geglu_bwd_f32")
        return dp


def geglu(proj):
    return GEGLUFn.apply(proj)


def _to_tokens_raw(x):
    B, C, H, W = x.shape
    y = torch.empty(B, H * W, C, device=x.device, dtype=torch.float32)
    N.check(N.lib().skp_nchw_to_tokens_f32(x.data_ptr(), y.data_ptr(), B, C, H * W, _stream()), "skp_nchw_to_tokens_f32")
    return y


def _to_nchw_raw(t, res, H, W):
    B, HW, C = t.shape
    y = torch.empty(B, C, H, W, device=t.device, dtype=torch.float32)
    N.check(N.lib().skp_tokens_to_nchw_f32(t.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(), B, C,
                                           HW, _stream()), "skp_tokens_to_nchw_f32")
    return y


class NchwToTokensFn(torch.autograd.Function):
    """[B,C,H,W] -> [B,H*W,C] (the permute(0,2,3,1).reshape of Transformer2DModel) as one tiled transpose."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[2], x.shape[3])
        return _to_tokens_raw(_dev(x, "x"))

    @staticmethod
    def backward(ctx, dy):
        return _to_nchw_raw(_dev(dy, "dy"), None, *ctx.hw)


class TokensToNchwAddFn(torch.autograd.Function):
    """[B,H*W,C] -> [B,C,H,W] plus the block residual in the same pass."""

    @staticmethod
    def forward(ctx, t, res):
        return _to_nchw_raw(_dev(t, "t"), _dev(res, "res"), res.shape[2], res.shape[3])

    @staticmethod
    def backward(ctx, dy):
        dy = _dev(dy, "dy")
        return (_to_tokens_raw(dy) if ctx.needs_input_grad[0] else None), (dy if ctx.needs_input_grad[1] else None)


def layout_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0


def nchw_to_tokens(x):
    return NchwToTokensFn.apply(x)


def tokens_to_nchw_add(t, res):
    return TokensToNchwAddFn.apply(t, res)
