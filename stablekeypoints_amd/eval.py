"""Arg-max helpers with the reference's names/semantics (eval.py:39-111), running on the HIP
token-statistics kernel (csrc/skp_select_loss.hip).  The rest of the reference's eval.py (dataset
metrics, plotting) is out of scope."""
from __future__ import annotations

import torch

from . import ops


def _points(flat: torch.Tensor, w: int) -> torch.Tensor:
    return torch.stack([flat // w, flat % w], dim=-1).to(torch.float32) + 0.5


def find_max_pixel(map):
    """[B,h,w] -> [B,2] = (row+0.5, col+0.5) of the first maximal element (eval.py:39-60)."""
    am, _ = ops.token_stats(map, num_subjects=1, want_kl=False)
    return _points(am[0].long(), map.shape[-1])


def find_k_max_pixels(map, num=3):
    """[B,h,w] -> [num,B,2]: repeated arg-max with 0.05*h radius masking (eval.py:62-81)."""
    am, _ = ops.token_stats(map, num_subjects=num, want_kl=False)
    return _points(am.long(), map.shape[-1])


def mask_radius(map, max_coords, radius):
    """eval.py:83-111 (elementwise helper, kept for API parity; the fused kernels mask internally)."""
    b, h, w = map.shape
    xs = torch.arange(w, device=map.device).view(1, 1, w)
    ys = torch.arange(h, device=map.device).view(1, h, 1)
    d2 = (xs - max_coords[:, 1].view(b, 1, 1)) ** 2 + (ys - max_coords[:, 0].view(b, 1, 1)) ** 2
    return map * (d2 > radius ** 2).float()
