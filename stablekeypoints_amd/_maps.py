"""Map reduction: `collect_maps` of optimize.py:27-79 on the fused HIP kernel."""
from __future__ import annotations

from typing import List

import torch
import torch.nn.functional as F

from . import ops


class FusedAttn:
    """Handle stored by the hooked cross-attention instead of the (B*h, R^2, T) probability tensor of
    ptp_utils.py:535-538.  Holds q = to_q(x) [B,s^2,C] and k = to_k(context) [B,T,C] (with autograd
    history) -- 8 MB per forward at SD-1.5/512^2 instead of the reference's 161 MB."""

    __slots__ = ("q", "k", "heads", "scale", "R")

    def __init__(self, q, k, heads, scale, R):
        self.q, self.k, self.heads, self.scale, self.R = q, k, int(heads), float(scale), int(R)

    @property
    def shape(self):
        return (self.q.shape[0] * self.heads, self.R * self.R, self.k.shape[1])

    def materialize(self) -> torch.Tensor:
        """The reference's stored tensor, (B*h, R^2, T) (no autograd; compat/testing)."""
        return ops.materialize_probs(self.q.detach(), self.k.detach(), self.heads, self.scale, self.R)


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
    if indices is None:
        out = fused_maps(chosen)
    else:
        R, heads = chosen[0].R, chosen[0].heads
        out = ops.attn_map_rows([r.q for r in chosen], [r.k for r in chosen], heads, [r.scale for r in chosen], R, indices)
    controller.reset()
    return out
