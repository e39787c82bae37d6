"""CPU restatement of the reference hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function names the reference lines it restates (paths relative to
/root/reference/unsupervised_keypoints/).  The op ORDER of the reference is kept
(materialised softmax, bicubic upsample of the layer *input*, second to_q, per-head stores,
stack+mean reduction) because this file doubles as the timed "reference CPU path".

All tensors fp32 unless `dtype=torch.float64` is passed to get a high-precision checker.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# a1  AttentionControl / AttentionStore                      ptp_utils.py:32-83
# ----------------------------------------------------------------------------------------
class OracleStore:
    """Container semantics of ptp_utils.AttentionStore (ptp_utils.py:63-83)."""

    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0
        self.step_store = {"attn": []}

    def __call__(self, d, is_cross, place_in_unet):           # ptp_utils.py:47-51,70-75
        self.step_store["attn"].append(d["attn"])
        return d["attn"]

    def reset(self):                                            # ptp_utils.py:53-55,77-79
        self.cur_step = 0
        self.cur_att_layer = 0
        self.step_store = {"attn": []}


# ----------------------------------------------------------------------------------------
# a2  patched CrossAttention.forward                          ptp_utils.py:480-541
# ----------------------------------------------------------------------------------------
def split_heads(t: torch.Tensor, heads: int) -> torch.Tensor:
    """diffusers 0.8.0 CrossAttention.reshape_heads_to_batch_dim [3P]: (B,N,C)->(B*h,N,C/h)."""
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, n, c // heads)


def merge_heads(t: torch.Tensor, heads: int) -> torch.Tensor:
    """diffusers 0.8.0 CrossAttention.reshape_batch_dim_to_heads [3P]: (B*h,N,d)->(B,N,h*d)."""
    bh, n, d = t.shape
    b = bh // heads
    return t.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)


def hooked_attention(x, context, wq, wk, wv, wo, bo, heads, store: Optional[OracleStore],
                     feature_upsample_res: int, max_stored: int = 4, max_seq: int = 32 ** 2):
    """ptp_utils.py:480-541.  Weights are nn.Linear-style (out,in); to_q/k/v have no bias.

    Returns the layer output; appends the up-res probability tensor (B*h, R*R, T) to `store`
    when the gate of ptp_utils.py:508-512 passes.
    """
    bsz, seq, dim = x.shape
    d_head = wq.shape[0] // heads
    scale = d_head ** -0.5
    q = F.linear(x, wq)                                         # :483
    is_cross = context is not None                              # :484
    ctx = context if is_cross else x                            # :486
    k = F.linear(ctx, wk)                                       # :487
    v = F.linear(ctx, wv)                                       # :488
    q, k, v = split_heads(q, heads), split_heads(k, heads), split_heads(v, heads)
    sim = torch.einsum("bid,bjd->bij", q, k) * scale            # :493
    attn = sim.softmax(dim=-1).clone()                          # :503-504
    out = torch.matmul(attn, v)                                 # :506
    if (is_cross and seq <= max_seq and store is not None
            and len(store.step_store["attn"]) < max_stored):   # :508-512
        side = int(seq ** 0.5)
        xr = x.reshape(bsz, side, side, dim).permute(0, 3, 1, 2)            # :513-518
        xr = F.interpolate(xr, size=(feature_upsample_res, feature_upsample_res),
                           mode="bicubic", align_corners=False)              # :520-526
        xr = xr.permute(0, 2, 3, 1).reshape(bsz, -1, dim)                    # :527-528
        q_up = split_heads(F.linear(xr, wq), heads)                          # :531-532
        sim_up = torch.einsum("bid,bjd->bij", q_up, k) * scale               # :534
        p_up = sim_up.softmax(dim=-1).clone()                                # :535-536
        store({"attn": p_up}, is_cross, "up")                                # :538
    out = merge_heads(out, heads)                                             # :540
    return F.linear(out, wo, bo)                                              # :541


# ----------------------------------------------------------------------------------------
# a5  collect_maps                                             optimize.py:27-79
# ----------------------------------------------------------------------------------------
def collect_maps(store: OracleStore, upsample_res=512, layers=(0, 1, 2, 3), indices=None):
    """optimize.py:27-79 incl. the sqrt(T) resize-guard quirk of :63 and the reset of :77."""
    per_layer = []
    for li, data in enumerate(store.step_store["attn"]):        # :44-48
        if li not in layers:
            continue
        side = int(data.shape[1] ** 0.5)
        data = data.reshape(data.shape[0], side, side, data.shape[2])        # :52-54
        if indices is not None:
            data = data[:, :, :, indices]                                    # :58-59
        data = data.permute(0, 3, 1, 2)                                      # :61
        if upsample_res != -1 and data.shape[1] ** 0.5 != upsample_res:      # :63 (sic)
            data = F.interpolate(data, size=(upsample_res, upsample_res),
                                 mode="bilinear", align_corners=False)       # :65-70
        per_layer.append(data)
    out = torch.stack(per_layer, dim=0).mean(dim=(0, 1))                      # :75
    store.reset()                                                             # :77
    return out


# ----------------------------------------------------------------------------------------
# a7  find_max_pixel / find_k_max_pixels / mask_radius         eval.py:39-111
# ----------------------------------------------------------------------------------------
def find_max_pixel(m: torch.Tensor) -> torch.Tensor:
    """eval.py:39-60: per-map argmax -> (row+0.5, col+0.5); first index wins ties."""
    n, h, w = m.shape
    flat = torch.argmax(m.reshape(n, -1), dim=-1)
    rc = torch.stack([flat // w, flat % w], dim=-1)
    return rc + 0.5


def mask_radius(m: torch.Tensor, centre: torch.Tensor, radius: float) -> torch.Tensor:
    """eval.py:83-111: multiply by [dist^2 > radius^2]; x grid is cols, y grid is rows."""
    n, h, w = m.shape
    xs = torch.arange(w, device=m.device).view(1, 1, w)
    ys = torch.arange(h, device=m.device).view(1, h, 1)
    d2 = (xs - centre[:, 1].view(n, 1, 1)) ** 2 + (ys - centre[:, 0].view(n, 1, 1)) ** 2
    return m * (d2 > radius ** 2).float()


def find_k_max_pixels(m: torch.Tensor, num: int = 3) -> torch.Tensor:
    """eval.py:62-81: repeated argmax with a 0.05*h masking radius; -> [num, n, 2]."""
    n, h, w = m.shape
    pts = []
    for _ in range(num):
        p = find_max_pixel(m)
        pts.append(p)
        m = mask_radius(m, p, 0.05 * h)
    return torch.stack(pts)


# ----------------------------------------------------------------------------------------
# a10 gaussian_circle(s)                                       optimize_token.py:203-241
# ----------------------------------------------------------------------------------------
def gaussian_circle(pos: torch.Tensor, size: int, sigma: float) -> torch.Tensor:
    """optimize_token.py:203-225.  pos[:,0] is the row, pos[:,1] the column, both in [0,1].

    meshgrid(ij) + stack(-1) makes grid[...,0]=row index, grid[...,1]=col index; the
    reference pairs grid[...,1] with pos[...,1] and grid[...,0] with pos[...,0].
    """
    p = (pos * size).view(-1, 1, 1, 2)
    ar = torch.arange(size, device=pos.device)
    rows = ar.view(1, size, 1) + 0.5
    cols = ar.view(1, 1, size) + 0.5
    d2 = (cols - p[..., 1]) ** 2 + (rows - p[..., 0]) ** 2
    return torch.exp(-1 * d2 / (2.0 * sigma ** 2.0))


def gaussian_circles(pos: torch.Tensor, size: int, sigma: float) -> torch.Tensor:
    """optimize_token.py:227-241: pos [num_points, batch, 2] -> mean over points."""
    return torch.stack([gaussian_circle(pos[i], size, sigma) for i in range(pos.shape[0])]).mean(dim=0)


# ----------------------------------------------------------------------------------------
# a8  find_top_k_gaussian                                      ptp_utils.py:86-112
# ----------------------------------------------------------------------------------------
def gaussian_kl(maps: torch.Tensor, sigma: float, epsilon: float = 1e-5, num_subjects: int = 1):
    """ptp_utils.py:95-108: KL(normalised gaussian || softmax_spatial(map+eps)) per token."""
    n, h, w = maps.shape
    loc = find_k_max_pixels(maps, num=num_subjects) / h
    sm = torch.softmax(maps.reshape(n, h * w) + epsilon, dim=-1)
    tgt = gaussian_circles(loc, size=h, sigma=sigma).reshape(n, h * w) + epsilon
    tgt = tgt / tgt.sum(dim=-1, keepdim=True)
    return torch.sum(tgt * (torch.log(tgt) - torch.log(sm)), dim=-1)


def find_top_k_gaussian(maps, top_k, sigma=3, epsilon=1e-5, num_subjects=1):
    """ptp_utils.py:86-112: ascending argsort of the KL, first top_k."""
    kl = gaussian_kl(maps, sigma, epsilon, num_subjects)
    return torch.argsort(kl, dim=-1, descending=False)[:top_k]


def token_entropy(maps: torch.Tensor) -> torch.Tensor:
    """ptp_utils.py:177-181: entropy of torch.distributions.Categorical(probs=softmax_spatial(map)) per token.
    Categorical renormalises the probs and takes log(clamp(probs, eps, 1-eps)) [torch.distributions.utils
    probs_to_logits]; entropy = -sum(probs * logits)."""
    n, h, w = maps.shape
    sm = torch.softmax(maps.reshape(n, h * w), dim=-1)
    probs = sm / sm.sum(dim=-1, keepdim=True)
    eps = torch.finfo(probs.dtype).eps
    logits = torch.log(probs.clamp(min=eps, max=1 - eps))
    return -(logits * probs).sum(dim=-1)


def entropy_sort(maps, top_k, min_dist=0.05):
    """ptp_utils.py:165-187: ascending argsort of the per-token entropy, first top_k."""
    return torch.argsort(token_entropy(maps), dim=-1, descending=False)[:top_k]


def pixel_from_weighted_avg(heatmaps: torch.Tensor, distance=5) -> torch.Tensor:
    """eval.py:113-155: zero everything farther than `distance` px from the arg-max (IN PLACE, like the reference),
    then the intensity-weighted mean (row, col) + 0.5."""
    b, m, n = heatmaps.shape
    x = torch.arange(0, m).float().view(1, m, 1)
    y = torch.arange(0, n).float().view(1, 1, n)
    if distance != -1:
        mx = find_max_pixel(heatmaps)
        x_max, y_max = mx[:, 0].long(), mx[:, 1].long()
        d = torch.sqrt((x - x_max.view(b, 1, 1)) ** 2 + (y - y_max.view(b, 1, 1)) ** 2)
        heatmaps[d > distance] = 0.0
    norm = heatmaps / (heatmaps.sum(dim=[1, 2], keepdim=True) + 1e-6)
    return torch.stack([(x * norm).sum(dim=[1, 2]), (y * norm).sum(dim=[1, 2])], dim=-1) + 0.5


# ----------------------------------------------------------------------------------------
# a9  furthest_point_sampling                                  ptp_utils.py:115-159
# ----------------------------------------------------------------------------------------
def furthest_point_sampling(maps: torch.Tensor, top_k: int, candidates: torch.Tensor):
    """ptp_utils.py:115-159.  Strict '>' comparisons, i<j scan order => first maximum wins."""
    n, h, w = maps.shape
    loc = find_max_pixel(maps) / h                                            # :127
    cand = [int(c) for c in candidates]
    best, pair = -1.0, None
    for a in range(len(cand)):                                                # :132-137
        for b in range(a + 1, len(cand)):
            dist = torch.sqrt(torch.sum((loc[cand[a]] - loc[cand[b]]) ** 2))
            if dist > best:
                best, pair = dist, (cand[a], cand[b])
    chosen = [pair[0], pair[1]]                                               # :140
    for _ in range(top_k - 2):                                                # :142-157
        far, far_id = -1.0, None
        for c in cand:
            if c in chosen:
                continue
            dmin = torch.min(torch.sqrt(torch.sum((loc[c] - loc[chosen]) ** 2, dim=-1)))
            if dmin > far:
                far, far_id = dmin, c
        if far_id is not None:
            chosen.append(far_id)
    return torch.tensor(chosen, device=maps.device)


# ----------------------------------------------------------------------------------------
# a10 sharpening loss                                          optimize.py:166-206
# ----------------------------------------------------------------------------------------
def sharpening_loss(maps: torch.Tensor, sigma: float = 1.0, num_subjects: int = 1):
    """optimize.py:166-206: MSE(map, gaussian centred at the map's own (masked) arg-maxima)."""
    pos = find_k_max_pixels(maps, num=num_subjects) / maps.shape[-1]
    target = gaussian_circles(pos, size=maps.shape[1], sigma=sigma)
    return F.mse_loss(maps, target)


# ----------------------------------------------------------------------------------------
# a11 RandomAffineWithInverse / equivariance loss              invertable_transform.py:6-92
# ----------------------------------------------------------------------------------------
def affine_matrix(angle_deg: float, scale: float, translate) -> torch.Tensor:
    """invertable_transform.py:22-36: [[s cos, s sin, tx], [-s sin, s cos, ty]] (1,2,3)."""
    a = math.radians(angle_deg)
    th = torch.tensor([[math.cos(a), math.sin(a), translate[0]],
                       [-math.sin(a), math.cos(a), translate[1]]], dtype=torch.float)
    th[:, :2] = th[:, :2] * scale
    return th.unsqueeze(0)


def draw_affine_params(gen_rand, degrees, scale, translate):
    """invertable_transform.py:42-51: four uniforms per image in this exact order."""
    angle = gen_rand() * (2 * degrees) - degrees
    sc = gen_rand() * (scale[1] - scale[0]) + scale[0]
    tx = gen_rand() * (2 * translate[0]) - translate[0]
    ty = gen_rand() * (2 * translate[1]) - translate[1]
    return angle, sc, (tx, ty)


def affine_warp(img: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """invertable_transform.py:65-68: affine_grid + grid_sample(bilinear, zeros, ac=False)."""
    grid = F.affine_grid(theta.to(img.dtype), img.size(), align_corners=False)
    return F.grid_sample(img, grid, align_corners=False)


def invert_theta(theta: torch.Tensor) -> torch.Tensor:
    """invertable_transform.py:77-84: append [0,0,1], 3x3 inverse, keep the top 2 rows."""
    last = torch.tensor([[0.0, 0.0, 1.0]], dtype=theta.dtype).expand(theta.shape[0], -1, -1)
    return torch.inverse(torch.cat([theta, last], dim=1))[:, :2, :]


def affine_unwarp(img: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """invertable_transform.py:72-92 (`inverse`)."""
    return affine_warp(img, invert_theta(theta))


def equivariance_loss(maps: torch.Tensor, maps_t: torch.Tensor, theta: torch.Tensor, index: int):
    """optimize.py:157-163: MSE(map[idx], unwarp(map_T[idx] repeated G times)[index])."""
    g = theta.shape[0]
    back = affine_unwarp(maps_t[None].repeat(g, 1, 1, 1), theta)[index]      # optimize.py:400,159
    return F.mse_loss(maps, back)


# ----------------------------------------------------------------------------------------
# a4  scheduler restatement [3P]                               optimize_token.py:25-34, ptp_utils.py:221-223
# ----------------------------------------------------------------------------------------
def ddim_alphas_cumprod(beta_start=0.00085, beta_end=0.012, n=1000):
    """diffusers DDIMScheduler 'scaled_linear' [3P]: betas = linspace(sqrt b0, sqrt b1, n)^2."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddim_timesteps(num_inference_steps=50, n=1000):
    """[3P] 0.8.0 set_timesteps: (arange(0,steps) * (n//steps))[::-1]; timesteps[-1] == 0."""
    return (torch.arange(0, num_inference_steps) * (n // num_inference_steps)).flip(0)


def add_noise(latent, noise, t: int):
    """[3P] sqrt(acp[t]) * latent + sqrt(1-acp[t]) * noise."""
    acp = ddim_alphas_cumprod()[int(t)]
    return acp.sqrt() * latent + (1 - acp).sqrt() * noise


# ----------------------------------------------------------------------------------------
# a12 one image of the optimisation loop, given the two reduced maps   optimize.py:380-420
# ----------------------------------------------------------------------------------------
def select_tokens(attn_map, attn_map_t, furthest_point_num_samples, top_k, sigma, num_subjects=1,
                  top_k_strategy="gaussian"):
    """optimize.py:382-395 (all three strategies)."""
    if top_k_strategy == "gaussian":
        cand = find_top_k_gaussian(attn_map, furthest_point_num_samples, sigma=sigma, num_subjects=num_subjects)
    elif top_k_strategy == "entropy":
        cand = entropy_sort(attn_map, furthest_point_num_samples)
    elif top_k_strategy == "consistent":
        cand = torch.arange(furthest_point_num_samples)
    else:
        raise NotImplementedError(top_k_strategy)
    return furthest_point_sampling(attn_map_t, top_k, cand)


def image_loss(attn_map, attn_map_t, theta, index, *, furthest_point_num_samples=25, top_k=10,
               sigma=2.0, num_subjects=1, sharpening_loss_weight=100.0,
               equivariance_attn_loss_weight=1000.0, top_k_strategy="gaussian"):
    """optimize.py:380-414 for one image: returns (loss, sharp, equiv, selected indices)."""
    idx = select_tokens(attn_map.detach(), attn_map_t.detach(), furthest_point_num_samples,
                        top_k, sigma, num_subjects, top_k_strategy)
    sharp = sharpening_loss(attn_map[idx], sigma=sigma, num_subjects=num_subjects)
    equiv = equivariance_loss(attn_map[idx], attn_map_t[idx], theta, index)
    loss = equiv * equivariance_attn_loss_weight + sharp * sharpening_loss_weight
    return loss, sharp, equiv, idx
