"""Fused execution of the frozen UNet/VAE blocks on the HIP kernels (GroupNorm+offset+SiLU).

`fuse_norms(module)` swaps the `forward` of every `ResnetBlock2D` / `Transformer2DModel` /
`AttentionBlock` / final-norm site for a version that is mathematically the module's own forward
(same weights, same order of operations per element) but runs
    conv1(no bias) -> [bias + time embedding folded into the norm's per-(n,c) offset] -> GN+SiLU
as ONE fused kernel pair instead of bias-add, temb-add, GroupNorm statistics, GroupNorm apply and SiLU
passes.  CUDA fp32 only; anything else falls through to the module's original forward.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import ops
from .attention import AttentionBlock, Transformer2DModel
from .unet import Downsample2D, ResnetBlock2D


def _resnet_forward(m: ResnetBlock2D):
    orig = m.forward

    def forward(x, temb=None):
        if not ops.group_norm_supported(x, m.norm1.num_groups):
            return orig(x, temb)
        # where the convolution runs with one output-channel group and no autograd (the VAE's first level: 128 -> 128 at the
        # image resolution, 1 GB activations) the norm is applied inside the convolution's patch load: no apply pass
        if ops.conv3x3_gn_fold_ok(x, m.norm1, m.conv1.weight):
            h = ops.conv3x3_gn_silu(x, m.norm1, m.conv1.weight, want_stats=True)
        else:
            h = ops.group_norm_silu(x, m.norm1)                          # (statistics from x's producer when it left them)
            h = ops.conv3x3_auto(h, m.conv1.weight, want_stats=True)     # bias folded into norm2's offset
        off = m.conv1.bias[None, :].expand(x.shape[0], -1)
        if m.time_emb_proj is not None and temb is not None:
            off = off + m.time_emb_proj(F.silu(temb))
        bias = m.conv2.bias
        res = x
        if m.conv_shortcut is not None:
            # 1x1 shortcut = one batched GEMM on the NCHW planes (no layout transposes); its bias joins conv2's
            res, bias = ops.conv1x1_nobias(x, m.conv_shortcut.weight), _summed_bias(m)
        if ops.conv3x3_gn_fold_ok(h, m.norm2, m.conv2.weight):
            return ops.conv3x3_gn_silu(h, m.norm2, m.conv2.weight, off=off.contiguous(), bias=bias, residual=res, want_stats=True)
        h = ops.group_norm_silu(h, m.norm2, off=off.contiguous())
        # bias + shortcut in the conv epilogue; its block sums serve the next block's first norm
        return ops.conv3x3_auto(h, m.conv2.weight, bias, residual=res, want_stats=True)
    return forward


def _summed_bias(m: ResnetBlock2D):
    """conv2.bias + conv_shortcut.bias of a frozen block, cached against the parameters' versions."""
    b2, bs = m.conv2.bias, m.conv_shortcut.bias
    if bs is None:
        return b2
    if b2.requires_grad or bs.requires_grad:
        return b2 + bs
    tag = (b2._version, bs._version, b2.data_ptr(), bs.data_ptr())
    hit = m.__dict__.get("_skp_bias_sum")
    if hit is None or hit[0] != tag:
        hit = (tag, (b2.detach() + bs.detach()))
        m.__dict__["_skp_bias_sum"] = hit
    return hit[1]


def _transformer_forward(m: Transformer2DModel):
    orig = m.forward

    def forward(x, context=None):
        if not ops.group_norm_supported(x, m.norm.num_groups):
            return orig(x, context)
        if not ops.layout_supported(x):
            return orig(x, context)                  # the module's own forward knows both projection layouts
        h = ops.group_norm_silu(x, m.norm, silu=False)
        # token layout throughout: the 1x1 convolutions are nn.Linear over tokens (bias fused in the GEMM), the two
        # permutes are tiled transposes and the residual add rides on the way back
        t = ops.nchw_to_tokens(h)
        t = ops.linear_auto(t, m.proj_in.weight.flatten(1), m.proj_in.bias)
        for blk in m.transformer_blocks:
            t = blk(t, context=context)
        t = ops.linear_auto(t, m.proj_out.weight.flatten(1), m.proj_out.bias)
        return ops.tokens_to_nchw_add(t, x)
    return forward


def _vae_attention_forward(m: AttentionBlock):
    """VAE mid-block attention: GroupNorm on the fused kernel, tiled NCHW -> token transpose instead of a strided copy, the
    way back fused with the residual add.  The d = 512 single-head core stays on the library GEMMs + softmax."""
    orig = m.forward

    def forward(x):
        if not (ops.group_norm_supported(x, m.group_norm.num_groups) and ops.layout_supported(x)):
            return orig(x)
        b, c, hh, ww = x.shape
        t = ops.nchw_to_tokens(ops.group_norm_silu(x, m.group_norm, silu=False))        # [b, hh*ww, c]
        q, k, v = m.query(t), m.key(t), m.value(t)
        nh = m.num_heads
        d = c // nh

        def split(u):
            return u.reshape(b, -1, nh, d).permute(0, 2, 1, 3).reshape(b * nh, -1, d)
        q, k, v = split(q), split(k), split(v)
        attn = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                             q, k.transpose(1, 2), beta=0, alpha=1.0 / (d ** 0.5)).softmax(dim=-1)
        o = torch.bmm(attn, v).reshape(b, nh, -1, d).permute(0, 2, 1, 3).reshape(b, -1, c)
        return ops.tokens_to_nchw_add(m.proj_attn(o), x)
    return forward


def _conv_forward(m: torch.nn.Conv2d):
    def forward(x):
        return ops.conv3x3_auto(x, m.weight, m.bias)
    return forward


def _downsample_forward(m):
    """Downsample2D: the stride-2 convolution, its zero padding and its bias are one kernel (csrc/skp_conv_s2.hip); where the
    input needs a gradient (UNet) only the input gradient stays on the library.  Unsupported shapes: the module's own forward."""
    orig = m.forward

    def forward(x):
        frozen = not (m.conv.weight.requires_grad or (m.conv.bias is not None and m.conv.bias.requires_grad))
        if frozen and m.conv.stride == (2, 2) and ops.conv3x3_s2_supported(x, m.conv.weight):
            return ops.conv3x3_s2(x, m.conv.weight, m.conv.bias, pad=0 if m.padding == 0 else 1, want_stats=True)
        return orig(x)
    return forward


def _linear_forward(m: torch.nn.Linear):
    def forward(x):
        return ops.linear_auto(x, m.weight, m.bias)
    return forward


def fuse_norms(module: torch.nn.Module) -> int:
    n = 0
    for mod in module.modules():
        if ops.EMULATED_F32 and isinstance(mod, torch.nn.Linear) and "forward" not in mod.__dict__:
            mod.forward = _linear_forward(mod)      # experiment: frozen Linear layers on the split-bf16 GEMM
            continue
        if isinstance(mod, Downsample2D) and "forward" not in mod.__dict__:
            mod.forward = _downsample_forward(mod); n += 1
            continue
        if (isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (3, 3) and mod.stride == (1, 1)
                and mod.padding == (1, 1) and mod.dilation == (1, 1) and mod.groups == 1
                and mod.padding_mode == "zeros" and "forward" not in mod.__dict__):
            mod.forward = _conv_forward(mod)        # Upsample2D.conv and friends (resnet convs are called below)
            continue
        if isinstance(mod, AttentionBlock) and "forward" not in mod.__dict__:
            mod.forward = _vae_attention_forward(mod); n += 1
            continue
        if isinstance(mod, ResnetBlock2D) and "forward" not in mod.__dict__:
            mod.forward = _resnet_forward(mod); n += 1
        elif isinstance(mod, Transformer2DModel) and "forward" not in mod.__dict__:
            mod.forward = _transformer_forward(mod); n += 1
    return n
