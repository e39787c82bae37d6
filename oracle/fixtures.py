"""Seeded synthetic inputs shared by the golden generator and the tests (TEST INFRASTRUCTURE).

Inputs are regenerated from seeds (CPU torch.Generator => identical wherever the same torch
build runs) so the committed fixtures only need to hold the reference's OUTPUTS.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def seeded(shape, seed: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randn(*shape, generator=g, dtype=torch.float32).to(dtype)


def attention_weights(C: int, ctx_dim: int, seed: int, gain: float = 1.5):
    """(wq, wk, wv, wo, bo) in nn.Linear layout (out, in)."""
    wq = seeded((C, C), seed * 7 + 1) * (gain / C ** 0.5)
    wk = seeded((C, ctx_dim), seed * 7 + 2) * (gain / ctx_dim ** 0.5)
    wv = seeded((C, ctx_dim), seed * 7 + 3) / ctx_dim ** 0.5
    wo = seeded((C, C), seed * 7 + 4) / C ** 0.5
    bo = seeded((C,), seed * 7 + 5) * 0.1
    return wq, wk, wv, wo, bo


def load_weights_into(mod, seed: int):
    """Fill a module with the 0.8.0 CrossAttention attribute layout (to_q/to_k/to_v/to_out[0])."""
    C = mod.to_q.weight.shape[0]
    ctx_dim = mod.to_k.weight.shape[1]
    wq, wk, wv, wo, bo = attention_weights(C, ctx_dim, seed)
    with torch.no_grad():
        mod.to_q.weight.copy_(wq)
        mod.to_k.weight.copy_(wk)
        mod.to_v.weight.copy_(wv)
        out = mod.to_out[0] if isinstance(mod.to_out, torch.nn.ModuleList) else mod.to_out
        out.weight.copy_(wo)
        out.bias.copy_(bo)


# G1: one hooked module.  `full` cases store every output element; the SD-shape cases a strided sample.
HOOK_CASES = {
    "tiny_int": dict(B=1, s=8, C=64, heads=2, T=13, ctx_dim=48, R=16, seed=11, full=True),
    "tiny_frac": dict(B=2, s=4, C=32, heads=4, T=5, ctx_dim=24, R=10, seed=12, full=True),
    "tiny_down": dict(B=1, s=8, C=32, heads=2, T=9, ctx_dim=16, R=6, seed=13, full=True),
    "sd_s16": dict(B=1, s=16, C=1280, heads=8, T=77, ctx_dim=768, R=32, seed=21, full=False, stride=997),
    "sd_s32": dict(B=1, s=32, C=640, heads=8, T=77, ctx_dim=768, R=64, seed=22, full=False, stride=997),
    "sd_s8": dict(B=1, s=8, C=1280, heads=8, T=100, ctx_dim=768, R=32, seed=23, full=False, stride=997),
}

# G2: a 4-layer store, SD-like structure (3 coarse layers + 1 finer), reduced dims.
STACK_CASE = dict(layers=[(4, 64), (4, 64), (4, 64), (8, 32)], heads=4, T=12, ctx_dim=40, R=16,
                  seed=31, indices=[7, 0, 3, 3, 11])

# G3/G4: selection + losses on fixed maps.
SEL_CASE = dict(T=40, R=32, n_cand=12, top_k=5, sigma=2.0, sel=[3, 17, 5, 22, 9], seed=41)

# G5: hook subgraph end to end (two views, selection, both losses, backward to the context).
E2E_CASE = dict(layers=[(4, 64), (4, 64), (4, 64), (8, 32)], heads=4, T=24, ctx_dim=40, R=32,
                n_cand=10, top_k=4, sigma=2.0, seed=51)


# G10: BASELINE config 2's real launch shape (SD-1.5 hooked layers, R = 128, T = 77): the reference hook + collect_maps
# at full size; the fixture keeps per-token arg-max / sums, a strided sample and a checksum.
FULL_CASE = dict(layers=[(16, 1280), (16, 1280), (16, 1280), (32, 640)], heads=8, T=77, ctx_dim=768, R=128, seed=71,
                 stride=997)


# G8/G9: the reference's driver functions on the reduced-width SD-topology model ("tiny", seed 0).
TINY_CASE = dict(size=128, T=16, R=32, n_cand=8, top_k=4, sigma=2.0, seed=61, aug_iters=3, upscale=64)


# G11/G12: the reference's optimize_embedding (3 optimizer steps x 2 accumulated images) and find_best_indices (24 images)
# on the same reduced-width model; 6 seeded images behind a tensor-dataset stub.  `ctx_gain` scales the start embedding so
# that the token softmax is clearly non-uniform and the KL ranking is not decided by rounding.
LOOP_CASE = dict(size=128, T=16, R=32, n_cand=8, top_k=4, sigma=2.0, seed=81, n_images=6, steps=3, accum=2,
                 num_indices=24, ctx_gain=1.0)


def selection_maps():
    """Two [T,R,R] non-negative maps with sum_t == 1 per pixel, some peaky tokens, and an exact
    two-way tie (pins the first-index argmax rule)."""
    c = SEL_CASE
    out = []
    for v in (0, 1):
        low = seeded((1, c["T"], 8, 8), c["seed"] + v) * 3.0
        up = F.interpolate(low, size=(c["R"], c["R"]), mode="bicubic", align_corners=False)[0]
        out.append(torch.softmax(up, dim=0).contiguous())
    maps, maps_t = out
    maps[5, 3, 4] = 1.5
    maps[5, 10, 20] = 1.5          # equal maxima: argmax must report (3,4)
    maps_t[6, 31, 31] = 1.25
    return maps, maps_t


def sharp_entropy_maps(maps):
    """The second map family of G3b (oracle/gen_golden.py): per-token entropies spread far apart."""
    n = maps.shape[0]
    return torch.softmax((maps * 40.0).view(n, -1) * torch.linspace(0.2, 3.0, n)[:, None], dim=-1).view_as(maps).contiguous()


# G13: the reference's keypoint_regressor.precompute_all_keypoints (:111-198) over a 5-image keypoint dataset stub on the
# same reduced-width model (the embedding after G11's last step, G12's voted indices): per image `aug_iters` affine
# views -> maps of the voted tokens at the reference's hard-wired 512 x 512 -> arg-max / weighted-average locations.
KPTS_CASE = dict(size=128, n_images=5, aug_iters=3, n_kpts=3, seed=91)
