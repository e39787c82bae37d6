"""Attention plumbing with the reference's API (ptp_utils.py) on the MI355X kernels.

Kept names/signatures: `AttentionControl`, `AttentionStore`,
`register_attention_control`, `find_pred_noise`, `run_and_find_attn`, `image2latent`,
`find_top_k_gaussian`, `furthest_point_sampling`, `init_random_noise`.

What changes under the hood
  * the hooked cross-attention does NOT up-sample the layer input, re-project it and materialise a
    (B*h, R^2, T) tensor per layer (ptp_utils.py:513-538).  It records a light `FusedAttn` handle
    (q = to_q(x) which the ordinary attention needs anyway, k = to_k(context)); `collect_maps`
    turns all handles into the reduced [T,R,R] map with ONE fused HIP kernel
    (csrc/skp_attn_map.hip).  `AttentionStore.step_store["attn"]` still is a list whose length is
    the gate the patcher reads (ptp_utils.py:511); `FusedAttn.materialize()` yields the reference's
    tensor on demand (compat / tests);
  * token selection never leaves the device (no `.item()`), see csrc/skp_select_loss.hip.
"""
from __future__ import annotations

import abc
from typing import Optional

import numpy as np

import torch

from . import ops
from .ldm.unet import StopForward
from ._maps import FusedAttn, collect_maps


# ---------------------------------------------------------------------------------------------
# controllers                                                       reference ptp_utils.py:32-83
# ---------------------------------------------------------------------------------------------
class AttentionControl(abc.ABC):
    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return 0

    @abc.abstractmethod
    def forward(self, dict, is_cross: bool, place_in_unet: str):
        raise NotImplementedError

    def __call__(self, dict, is_cross: bool, place_in_unet: str):
        dict = self.forward(dict, is_cross, place_in_unet)
        return dict["attn"]

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0

    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0


class AttentionStore(AttentionControl):
    """`step_store["attn"]` holds one entry per hooked layer: a `FusedAttn` handle (default) or, with
    `materialize=True`, the reference-layout tensor (B*h, R^2, T)."""

    @staticmethod
    def get_empty_store():
        return {"attn": []}

    def forward(self, dict, is_cross: bool, place_in_unet: str):
        self.step_store["attn"].append(dict["attn"])
        return dict

    def reset(self):
        super(AttentionStore, self).reset()
        self.step_store = self.get_empty_store()

    def __init__(self, materialize: bool = False):
        super(AttentionStore, self).__init__()
        self.step_store = self.get_empty_store()
        self.materialize = materialize
        self.stop_after: Optional[int] = None      # early exit once this many maps are stored


MAX_STORED_LAYERS = 4          # ptp_utils.py:511
MAX_STORED_SEQ = 32 ** 2       # ptp_utils.py:510


# ---------------------------------------------------------------------------------------------
# context projections of one forward                          reference ptp_utils.py:513-520
# ---------------------------------------------------------------------------------------------
CTX_KV_BATCHED = True        # False (tests): every layer projects the context for itself
_CTX_KV = {}                 # id(CrossAttention module) -> (k, v) [1,T,C] of the forward in flight


def _ctx_layer_ok(m, width):
    return (m.to_k.bias is None and m.to_v.bias is None and not m.to_k.weight.requires_grad and not m.to_v.weight.requires_grad
            and m.to_k.weight.shape[1] == width and "forward" not in m.to_k.__dict__ and "forward" not in m.to_v.__dict__)


def _context_kv_begin(unet, context, early_exit=False):
    """k = to_k(context), v = to_v(context) of the cross-attention layers for the ONE shared context row, before the UNet
    runs: the reference expands the learned embedding to the batch and projects it inside every layer (B x 77 rows through
    2 x 16 small GEMMs, as many again backward plus their accumulation adds).  The rows are identical, the weights frozen
    and bias-free: one batched GEMM per layer width (stacked weights, broadcast A operand) gives every layer its [1,T,C]
    pair, the attention kernels take a shared k / v (Bk = 1) and autograd sums the layers' gradients in one stack + one
    batched GEMM per width.  Layers that did not consume their pair in an earlier forward (beyond the early exit) are left
    out from the second forward on; a layer without a pair projects for itself, as before."""
    _CTX_KV.clear()
    if not (CTX_KV_BATCHED and context.is_cuda and context.dim() == 3 and context.shape[0] == 1):
        return
    plan = getattr(unet, "_skp_ctx_plan", None)
    # the plan holds only layers whose projections are frozen, bias-free and unpatched -- re-checked on every forward (32
    # modules): a layer that was unfrozen or re-patched since must project for itself again, or its weights would silently
    # get no gradient
    if plan is None or not all(_ctx_layer_ok(m, context.shape[-1]) for m in plan["mods"]):
        mods = [m.attn2 for m in unet.modules() if m.__class__.__name__ == "BasicTransformerBlock" and hasattr(m, "attn2")]
        mods = [m for m in mods if _ctx_layer_ok(m, context.shape[-1])]
        plan = {"mods": mods, "used": {}, "ids": {id(m) for m in mods}}
        unet._skp_ctx_plan = plan
    # which layers consume their pair depends on where the forward ends: one record per mode (early exit / full forward)
    used = plan["used"].get(bool(early_exit))
    mods = plan["mods"] if used is None else [m for m in plan["mods"] if id(m) in used]
    if used is None:
        used = plan["used"][bool(early_exit)] = set()
    _CTX_KV["plan"] = {"ids": plan["ids"], "used": used}
    groups = {}
    for m in mods:
        groups.setdefault(m.to_k.weight.shape[0], []).append(m)
    ctx2 = context[0]
    for c, ms in groups.items():
        w = ops.weight_stack([t for m in ms for t in (m.to_k.weight, m.to_v.weight)])        # [2n, C, Dctx], made once
        kv = torch.bmm(ctx2.unsqueeze(0).expand(w.shape[0], -1, -1), w.transpose(1, 2)).unbind(0)
        for i, m in enumerate(ms):
            _CTX_KV[id(m)] = (kv[2 * i].unsqueeze(0), kv[2 * i + 1].unsqueeze(0), used)


def _context_kv_end():
    _CTX_KV.clear()


def _context_kv(module):
    """(k, v) of `module` for the forward in flight, or None (no shared-context forward, or the layer was left out: it
    projects for itself this time and is part of the batch from the next forward on)."""
    hit = _CTX_KV.get(id(module))
    if hit is None:
        plan = _CTX_KV.get("plan")
        if plan is not None and id(module) in plan["ids"]:
            plan["used"].add(id(module))
        return None
    hit[2].add(id(module))
    return hit[0], hit[1]


def _attention_core(module, q, k, v, is_cross=False):
    """softmax(scale q k^T) v per head (ptp_utils.py:493-506) on [B,N,C]/[B,T,C] tensors.  Cross layers with
    a short key axis run on the fused fp32-MFMA kernel (csrc/skp_cross_attn.hip); self-attention runs on the
    flash-style kernels (csrc/skp_flash_attn.hip, csrc/skp_self_attn.hip).  On the GPU there is NO other route: a head
    size without a kernel raises.  Host tensors (the module tree on CPU is what oracle/ drives) take the plain
    baddbmm / softmax / bmm formulation of the reference."""
    if q.is_cuda:
        if is_cross and ops.cross_attn_supported(q.shape[-1], module.heads, k.shape[1]):
            return ops.cross_attention(q, k, v, module.heads, module.scale)
        if ops.self_attn_supported(q.shape[-1], module.heads):
            return ops.self_attention(q, k, v, module.heads, module.scale)
        raise RuntimeError(f"attention with {module.heads} heads of {q.shape[-1] // module.heads} channels has no HIP kernel "
                           f"(built head sizes: {ops.CROSS_ATTN_HEAD_DIMS}); there is no eager fallback on the GPU")
    qh = module.reshape_heads_to_batch_dim(q)
    kh = module.reshape_heads_to_batch_dim(k)
    vh = module.reshape_heads_to_batch_dim(v)
    attn = torch.baddbmm(torch.empty(qh.shape[0], qh.shape[1], kh.shape[1], dtype=q.dtype, device=q.device),
                         qh, kh.transpose(1, 2), beta=0, alpha=module.scale).softmax(dim=-1)
    return module.reshape_batch_dim_to_heads(torch.bmm(attn, vh))


def register_attention_control(model, controller, feature_upsample_res=256):
    """Same contract as ptp_utils.py:472-573: patches `.forward` of every module whose class is named
    `CrossAttention` under top-level children whose name contains "up", sets
    `controller.num_att_layers`, asserts that at least one layer was found."""

    def ca_forward(self, place_in_unet):
        to_out = self.to_out[0] if isinstance(self.to_out, torch.nn.ModuleList) else self.to_out

        def forward(x, context=None, mask=None):
            if mask is not None:
                raise NotImplementedError("attention masks are never used on this path (ptp_utils.py:496)")
            batch_size, sequence_length, dim = x.shape
            is_cross = context is not None
            ctx = context if is_cross else x
            stop = getattr(controller, "stop_after", None)
            if (is_cross and stop is not None and sequence_length <= MAX_STORED_SEQ
                    and len(controller.step_store["attn"]) + 1 >= min(stop, MAX_STORED_LAYERS)
                    and len(controller.step_store["attn"]) < MAX_STORED_LAYERS):
                # early exit: this is the LAST map the gate will store and the forward ends here (the caller discards the
                # prediction, ptp_utils.py:246) -- the layer's value projection and attention output have no consumer
                pre = _context_kv(self)
                rec = FusedAttn(self.to_q(x), pre[0] if pre is not None else self.to_k(ctx), self.heads, self.scale,
                                feature_upsample_res)
                if getattr(controller, "materialize", False):
                    rec = rec.materialize()
                controller({"attn": rec}, is_cross, place_in_unet)
                raise StopForward()
            if (not is_cross) and self.to_q.bias is None and self.to_k.bias is None and self.to_v.bias is None \
                    and "forward" not in self.to_q.__dict__:
                if x.is_cuda and ops.self_attn_supported(dim, self.heads):
                    return to_out(ops.self_attention_block(x, self.to_q.weight, self.to_k.weight, self.to_v.weight,
                                                           self.heads, self.scale))
                q, k, v = ops.qkv_proj(x, self.to_q.weight, self.to_k.weight, self.to_v.weight)
            else:
                q = self.to_q(x)
                pre = _context_kv(self) if is_cross else None
                k, v = pre if pre is not None else (self.to_k(ctx), self.to_v(ctx))
            out = _attention_core(self, q, k, v, is_cross)
            if (is_cross and sequence_length <= MAX_STORED_SEQ
                    and len(controller.step_store["attn"]) < MAX_STORED_LAYERS):
                rec = FusedAttn(q, k, self.heads, self.scale, feature_upsample_res)
                if getattr(controller, "materialize", False):
                    rec = rec.materialize()
                controller({"attn": rec}, is_cross, place_in_unet)
                stop = getattr(controller, "stop_after", None)
                if stop is not None and len(controller.step_store["attn"]) >= stop:
                    raise StopForward()
            elif (is_cross and sequence_length > MAX_STORED_SEQ and controller.step_store["attn"]
                  and getattr(controller, "stop_after", None) is not None):
                # only `up` blocks are hooked and their resolution never decreases: once a cross layer is past the
                # 32^2 gate no later layer can store (SD-2.x at 768^2 stores 3 layers, not 4) => nothing left to record
                raise StopForward()
            return to_out(out)

        return forward

    class DummyController:
        def __call__(self, *args):
            return args[0]

        def __init__(self):
            self.num_att_layers = 0
            self.step_store = {"attn": []}

    if controller is None:
        controller = DummyController()

    def register_recr(net_, count, place_in_unet):
        if net_.__class__.__name__ == "CrossAttention":
            net_.forward = ca_forward(net_, place_in_unet)
            return count + 1
        elif hasattr(net_, "children"):
            for net__ in net_.children():
                count = register_recr(net__, count, place_in_unet)
        return count

    cross_att_count = 0
    for name, child in model.named_children():
        if "up" in name:
            cross_att_count += register_recr(child, 0, "up")
    controller.num_att_layers = cross_att_count
    assert cross_att_count != 0, ("No cross attention layers found in the model. The module tree must use the "
                                  "diffusers==0.8.0 `CrossAttention` layout.")


def accelerate_cross_attention(net):
    """Route the UNPATCHED (down/mid) cross-attention layers through the same fused kernel; they never store
    maps (the reference only hooks `up` blocks, ptp_utils.py:565-568) so only their `forward` core changes."""
    for mod in net.modules():
        if mod.__class__.__name__ == "CrossAttention" and "forward" not in mod.__dict__:
            def make(m):
                to_out = m.to_out[0] if isinstance(m.to_out, torch.nn.ModuleList) else m.to_out

                def forward(x, context=None, mask=None):
                    is_cross = context is not None
                    ctx = context if is_cross else x
                    if (not is_cross) and m.to_q.bias is None and m.to_k.bias is None and m.to_v.bias is None:
                        if x.is_cuda and ops.self_attn_supported(x.shape[-1], m.heads):
                            return to_out(ops.self_attention_block(x, m.to_q.weight, m.to_k.weight, m.to_v.weight, m.heads,
                                                                   m.scale))
                        q, k, v = ops.qkv_proj(x, m.to_q.weight, m.to_k.weight, m.to_v.weight)
                        return to_out(_attention_core(m, q, k, v, False))
                    pre = _context_kv(m) if is_cross else None
                    k, v = pre if pre is not None else (m.to_k(ctx), m.to_v(ctx))
                    return to_out(_attention_core(m, m.to_q(x), k, v, is_cross))
                return forward
            mod.forward = make(mod)


# ---------------------------------------------------------------------------------------------
# noised forward driver                                    reference ptp_utils.py:205-304
# ---------------------------------------------------------------------------------------------
def image2latent(model, image, device):
    """ptp_utils.py:289-304: [0,1] image -> *2-1 -> VAE posterior mean * 0.18215.  Accepts the reference's
    numpy NHWC array or a [B,3,H,W] tensor (no CPU round trip for tensors)."""
    with torch.no_grad():
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(image).float().permute(0, 3, 1, 2)
        image = image.to(device=device, dtype=torch.float32) * 2 - 1
        vae = model.vae.module if isinstance(model.vae, torch.nn.DataParallel) else model.vae
        latents = vae.encode(image)["latent_dist"].mean
        return latents * 0.18215


def find_pred_noise(ldm, image, context, noise_level=-1, device="cuda", noise=None, early_exit=False,
                    controllers=None, latents=None):
    """ptp_utils.py:205-231.  `noise` lets a caller inject the gaussian (the reference draws it from the
    device RNG, :219).  `early_exit` stops the UNet after the last stored map -- result-identical for
    `run_and_find_attn`, which discards the prediction (ptp_utils.py:246).  `latents` [B,4,h,w]: already encoded
    rows (image2latent's output, e.g. kept from an earlier epoch) -- `image` is then ignored."""
    if latents is not None:
        latent = latents.to(device=device, dtype=torch.float32)
    else:
        if isinstance(image, torch.Tensor) and image.dim() == 3:
            image = image[None]
        latent = image2latent(ldm, image, device)
    if noise is None:
        noise = torch.randn_like(latent)
    t = ldm.scheduler.timesteps[noise_level]
    noisy_image = ldm.scheduler.add_noise(latent, noise, t)
    b = noisy_image.shape[0]
    if early_exit and controllers is not None:
        for c in controllers.values():
            c.stop_after = MAX_STORED_LAYERS
    try:
        _context_kv_begin(ldm.unet, context, early_exit=bool(early_exit and controllers is not None))
        pred_noise = ldm.unet(noisy_image, t.repeat(b), context.expand(b, -1, -1) if context.shape[0] == 1 else context)["sample"]
    except StopForward:
        pred_noise = None
    finally:
        _context_kv_end()
        if early_exit and controllers is not None:
            for c in controllers.values():
                c.stop_after = None
    return noise, pred_noise


def run_and_find_attn(ldm, image, context, noise_level=-1, device="cuda",
                      from_where=["down_cross", "mid_cross", "up_cross"], layers=[0, 1, 2, 3, 4, 5],
                      upsample_res=32, indices=None, controllers=None, noise=None, early_exit=True):
    """ptp_utils.py:234-272: one noised UNet forward, then `collect_maps` per controller."""
    find_pred_noise(ldm, image, context, noise_level=noise_level, device=device, noise=noise,
                    early_exit=early_exit, controllers=controllers)
    attention_maps = []
    for controller in controllers:
        attention_maps.append(collect_maps(controllers[controller], from_where=from_where,
                                           upsample_res=upsample_res, layers=layers, indices=indices))
        controllers[controller].reset()
    return attention_maps


# ---------------------------------------------------------------------------------------------
# token selection                                              reference ptp_utils.py:86-159
# ---------------------------------------------------------------------------------------------
def _ranked(score, argmax, R, top_k):
    """First `top_k` token ids by ascending score (ties by index, NaN last -- torch.argsort's order): the selection
    kernel's ranking stage; candidate lists longer than the kernel's 64 slots are a stable device sort."""
    n = score.shape[0]
    top_k = min(int(top_k), n)
    if 2 <= top_k <= ops.SELECT_MAX_CANDIDATES and n <= ops.SELECT_MAX_TOKENS:
        cand, _ = ops.select_tokens(score, argmax, R, top_k, 2)
        return cand
    return torch.sort(score, stable=True).indices[:top_k]


def find_top_k_gaussian(attention_maps, top_k, sigma=3, epsilon=1e-5, num_subjects=1):
    """ptp_utils.py:86-112 -> int64[top_k] (device)."""
    am, kl = ops.token_stats(attention_maps, num_subjects=num_subjects, sigma=sigma, eps=epsilon)
    return _ranked(kl, am[0], attention_maps.shape[-1], top_k)


def furthest_point_sampling(attention_maps, top_k, top_initial_candidates):
    """ptp_utils.py:115-159 -> int64[min(top_k, candidates)] (device); greedy max-min over the candidates' arg-max
    pixels (the reference's loop stops adding once every candidate is chosen, :142-157)."""
    am, _ = ops.token_stats(attention_maps, num_subjects=1, want_kl=False)
    n = attention_maps.shape[0]
    cand = torch.as_tensor(top_initial_candidates, device=attention_maps.device).long()
    order = torch.full((n,), float("inf"), device=attention_maps.device)
    order[cand] = torch.arange(cand.numel(), device=attention_maps.device, dtype=torch.float32)
    _, sel = ops.select_tokens(order, am[0], attention_maps.shape[-1], int(cand.numel()),
                               min(int(top_k), int(cand.numel())))
    return sel


def entropy_sort(attention_maps, top_k, min_dist=0.05):
    """ptp_utils.py:165-187 -> int64[top_k] (device): tokens by ascending entropy of softmax_{R*R}(map)."""
    am, _, ent = ops.token_stats(attention_maps, num_subjects=1, want_kl=False, want_entropy=True)
    return _ranked(ent, am[0], attention_maps.shape[-1], top_k)


def init_random_noise(device, num_words=77, dim=768):
    """ptp_utils.py:649-650 (`dim` = the UNet's cross_attention_dim: 768 SD-1.x, 1024 SD-2.x, 2048 SDXL)."""
    return torch.randn(1, num_words, dim).to(device)
