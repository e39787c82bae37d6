"""Map reduction: `collect_maps` of optimize.py:27-79 on the fused HIP kernel."""
from __future__ import annotations

from typing import List

import torch
import torch.nn.functional as F

from . import ops


class FusedAttn:
    """Handle stored by the hooked cross-attention instead of the (B*h, R^2, T) probability tensor of
    ptp_utils.py:535-538.  Holds q = to_q(x) [B,s^2,C] and k = to_k(context) [B,T,C] (with autograd
    history) -- 8 MB per forward at SD-1.5/512^2 instead of the reference's 161 MB.

    Tensor duck type: code written against the reference's store -- the reference's OWN `optimize.collect_maps`
    (optimize.py:52-75: `data.reshape(...)`, `data[:, :, :, indices]`, `.permute`, `F.interpolate`, `torch.stack`),
    reached from its `keypoint_regressor.find_best_indices` / `eval.run_image_with_context_augmented` through its
    `ptp_utils.run_and_find_attn` -- sees the reference tensor: any tensor attribute, index or torch function applied to
    a handle materialises `(B*h, R^2, T)` once (cached, no autograd) and forwards to it.  The fused `collect_maps` of
    this package never does that."""

    __slots__ = ("q", "k", "heads", "scale", "R", "_mat")

    def __init__(self, q, k, heads, scale, R):
        self.q, self.k, self.heads, self.scale, self.R = q, k, int(heads), float(scale), int(R)
        self._mat = None

    @property
    def shape(self):
        return torch.Size((self.q.shape[0] * self.heads, self.R * self.R, self.k.shape[1]))

    @property
    def device(self):
        return self.q.device

    @property
    def dtype(self):
        return self.q.dtype

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 3

    def materialize(self) -> torch.Tensor:
        """The reference's stored tensor, (B*h, R^2, T) (no autograd; compat/testing).  GPU tensors: the fused map kernel
        run per head (ops.materialize_probs); host tensors (the module tree on CPU, which is what the build container and
        oracle/ can drive): the same quantity in torch ops -- softmax_t(bicubic_R(scale q k^T))."""
        if self._mat is None:
            if torch.is_grad_enabled() and (self.q.requires_grad or self.k.requires_grad):
                raise RuntimeError(
                    "FusedAttn: a stored cross-attention handle was turned into the reference's (B*h, R^2, T) tensor while "
                    "autograd is recording and its q / k carry gradient history.  The materialised tensor is a compatibility "
                    "view WITHOUT autograd, so a loss built on it cannot reach the embedding (`loss.backward()` would fail "
                    "with 'does not require grad').  This happens when only `optimize_token.load_ldm` was re-bound and the "
                    "reference's own `optimize.optimize_embedding` / `optimize.collect_maps` still run the training loop: "
                    "re-bind `optimize.optimize_embedding` (or at least `optimize.collect_maps`) to stablekeypoints_amd.optimize "
                    "as well (INTEGRATION.md section A, stage 1), or wrap inference-only callers in torch.no_grad().")
            q, k = self.q.detach(), self.k.detach()
            if q.is_cuda:
                self._mat = ops.materialize_probs(q, k, self.heads, self.scale, self.R)
            else:
                self._mat = _materialize_host(q, k, self.heads, self.scale, self.R)
        return self._mat

    # ---- tensor duck typing (see the class docstring) ----
    # what reference-order code calls on a stored entry (optimize.py:52-75, eval.py, visualize.py) plus the usual
    # conversions; anything else (a typo, a hasattr() probe) must NOT silently build the 161 MB tensor
    _FORWARDED = frozenset((
        "reshape", "view", "permute", "transpose", "flatten", "unsqueeze", "squeeze", "contiguous", "clone", "detach",
        "mean", "sum", "max", "min", "argmax", "softmax", "float", "double", "half", "to", "cpu", "cuda", "numpy", "tolist",
        "type", "chunk", "split", "unbind", "index_select", "T", "mT", "abs", "mul", "add", "div", "sub", "expand",
        "repeat", "narrow", "select", "amax", "amin", "topk", "sort", "isfinite", "isnan", "all", "any", "item"))

    @property
    def ndim(self):
        return 3

    @property
    def is_cuda(self):
        return self.q.is_cuda

    @property
    def requires_grad(self):
        return False                                             # of the materialised view (see materialize())

    def numel(self):
        return self.shape.numel()

    def __getattr__(self, name):
        if name in FusedAttn._FORWARDED:
            return getattr(self.materialize(), name)
        raise AttributeError(f"FusedAttn has no attribute '{name}' (tensor methods forwarded to the materialised "
                             f"(B*h, R^2, T) view: {', '.join(sorted(FusedAttn._FORWARDED))})")

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __len__(self):
        return self.shape[0]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def conv(a):
            if isinstance(a, FusedAttn):
                return a.materialize()
            if isinstance(a, (list, tuple)):                     # rebuilt as plain list / tuple (a namedtuple's constructor
                return (list if isinstance(a, list) else tuple)(conv(v) for v in a)   # does not take a generator)
            return a
        return func(*conv(args), **{k_: conv(v) for k_, v in (kwargs or {}).items()})


def _materialize_host(q, k, heads, scale, R):
    """(B*h, R^2, T) on host tensors: logits at the layer's resolution, bicubic to R (align_corners=False, the resize of
    ptp_utils.py:520-524 moved behind the bias-free, linear `to_q`), softmax over the tokens."""
    B, s2, C = q.shape
    Bk, T, _ = k.shape
    s, d = int(round(s2 ** 0.5)), C // heads
    qh = q.reshape(B, s2, heads, d).permute(0, 2, 1, 3)
    kh = k.expand(B, T, C).reshape(B, T, heads, d).permute(0, 2, 3, 1)
    S = torch.matmul(qh, kh) * scale                                             # [B,h,s2,T]
    S = S.permute(0, 1, 3, 2).reshape(B * heads, T, s, s)
    if s != R:
        S = F.interpolate(S, size=(R, R), mode="bicubic", align_corners=False)
    return S.reshape(B * heads, T, R * R).permute(0, 2, 1).softmax(dim=-1).contiguous()


def fused_maps(records: List[FusedAttn]) -> torch.Tensor:
    """[B,T,R,R] = mean over (layer, head) of the up-res softmax maps (one kernel launch)."""
    R, heads = records[0].R, records[0].heads
    for r in records:
        if r.R != R or r.heads != heads:
            raise RuntimeError("hooked layers disagree on feature_upsample_res / head count")
    return ops.attn_map([r.q for r in records], [r.k for r in records], heads, [r.scale for r in records], R)


def collect_maps(controller, from_where=["up_cross"], upsample_res=512, layers=[0, 1, 2, 3], indices=None):
    """optimize.py:27-79: mean over the selected layers and over the (batch*heads) axis -> [T',R',R'];
    optional token gather and bilinear resize; RESETS the controller.  The resize guard reproduces the
    reference's `sqrt(T') != upsample_res` comparison (optimize.py:63)."""
    store = controller.step_store["attn"]
    chosen = [rec for i, rec in enumerate(store) if i in layers]
    if not chosen:
        raise RuntimeError("collect_maps: no stored attention layer matches `layers`")
    if all(isinstance(r, FusedAttn) for r in chosen):
        if indices is not None and not torch.is_grad_enabled():  # inference: only the asked-for rows are computed / written
            for r in chosen:
                if r.R != chosen[0].R or r.heads != chosen[0].heads:
                    raise RuntimeError("hooked layers disagree on feature_upsample_res / head count")
            m = ops.attn_map_rows([r.q for r in chosen], [r.k for r in chosen], chosen[0].heads,
                                  [r.scale for r in chosen], chosen[0].R, indices).mean(dim=0)
        else:
            m = fused_maps(chosen).mean(dim=0)                    # mean over the batch rows (B*h axis)
            if indices is not None:
                m = m[torch.as_tensor(indices, device=m.device)]
        if upsample_res != -1 and m.shape[0] ** 0.5 != upsample_res:
            # bilinear resize is linear, so it commutes with the layer/head mean
            m = F.interpolate(m[None], size=(upsample_res, upsample_res), mode="bilinear", align_corners=False)[0]
        out = m
    else:
        # materialised entries (AttentionStore(materialize=True) or a foreign controller): the reference's
        # op sequence on device tensors
        per_layer = []
        for data in chosen:
            if isinstance(data, FusedAttn):
                data = data.materialize()
            side = int(data.shape[1] ** 0.5)
            data = data.reshape(data.shape[0], side, side, data.shape[2])
            if indices is not None:
                data = data[:, :, :, indices]
            data = data.permute(0, 3, 1, 2)
            if upsample_res != -1 and data.shape[1] ** 0.5 != upsample_res:
                data = F.interpolate(data, size=(upsample_res, upsample_res), mode="bilinear", align_corners=False)
            per_layer.append(data)
        out = torch.stack(per_layer, dim=0).mean(dim=(0, 1))
    controller.reset()
    return out


def collect_maps_batched(controller, layers=(0, 1, 2, 3), indices=None) -> torch.Tensor:
    """[B,T,R,R]: one reduced map per batch row (the batched engine's variant); resets the controller.
    `indices` (inference, no autograd): only those tokens' maps, [B,len(indices),R,R]."""
    chosen = [rec for i, rec in enumerate(controller.step_store["attn"]) if i in layers]
    if not chosen:
        raise RuntimeError("collect_maps_batched: no stored attention layer matches `layers`")
    if indices is None:
        out = fused_maps(chosen)
    elif torch.is_grad_enabled() and any(r.q.requires_grad or r.k.requires_grad for r in chosen):
        # the row-selecting kernel path is inference only (it detaches q / k): with autograd on, gather differentiably
        out = fused_maps(chosen)[:, torch.as_tensor(indices, device=chosen[0].q.device).long()]
    else:
        R, heads = chosen[0].R, chosen[0].heads
        for r in chosen:
            if r.R != R or r.heads != heads:
                raise RuntimeError("hooked layers disagree on feature_upsample_res / head count")
        out = ops.attn_map_rows([r.q for r in chosen], [r.k for r in chosen], heads, [r.scale for r in chosen], R, indices)
    controller.reset()
    return out
