#!/usr/bin/env python
"""TF/s of F.conv2d (fp32, MIOpen) at the VAE/UNet shapes of the step, forward and backward-data."""
import torch, torch.nn.functional as F
dev = "cuda"
shapes = [  # (name, N, Cin, Cout, H, k, stride)
    ("vae 512^2 3->128", 8, 3, 128, 512, 3, 1), ("vae 512^2 128->128", 8, 128, 128, 512, 3, 1),
    ("vae down 512->256 128", 8, 128, 128, 513, 3, 2), ("vae 256^2 128->256", 8, 128, 256, 256, 3, 1),
    ("vae 256^2 256->256", 8, 256, 256, 256, 3, 1), ("vae 128^2 256->512", 8, 256, 512, 128, 3, 1),
    ("vae 128^2 512->512", 8, 512, 512, 128, 3, 1), ("vae 64^2 512->512", 8, 512, 512, 64, 3, 1),
    ("unet 64^2 320->320", 8, 320, 320, 64, 3, 1), ("unet 32^2 640->640", 8, 640, 640, 32, 3, 1),
    ("unet 16^2 1280->1280", 8, 1280, 1280, 16, 3, 1), ("unet 8^2 1280->1280", 8, 1280, 1280, 8, 3, 1),
    ("unet 16^2 2560->1280", 8, 2560, 1280, 16, 3, 1), ("unet 32^2 1920->640", 8, 1920, 640, 32, 3, 1),
    ("unet 64^2 320->320 1x1", 8, 320, 320, 64, 1, 1),
]
def t(fn, it=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for name, N, ci, co, H, k, st in shapes:
    x = torch.randn(N, ci, H, H, device=dev, requires_grad=True); w = torch.randn(co, ci, k, k, device=dev)
    pad = 0 if st == 2 else k // 2
    y = F.conv2d(x, w, None, st, pad)
    fl = 2.0 * N * co * ci * k * k * y.shape[-1] * y.shape[-2]
    tf = t(lambda: F.conv2d(x, w, None, st, pad))
    g = torch.randn_like(y)
    tb = t(lambda: torch.autograd.grad(F.conv2d(x, w, None, st, pad), x, g)) - tf
    print(f"{name:26s} fwd {tf*1e3:7.2f} ms {fl/tf/1e12:6.1f} TF/s | bwd-data {tb*1e3:7.2f} ms {fl/max(tb,1e-9)/1e12:6.1f} TF/s")
