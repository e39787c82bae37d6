#!/usr/bin/env python
"""fp32 TF/s of the library GEMMs at the nn.Linear shapes of the SD-1.5 512^2 step (B = 8 rows), forward (x W^T + b) and
backward-data (dy W), with the committed TunableOp selection enabled."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from stablekeypoints_amd import tuning  # noqa: E402


def timed(fn, it=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def main():
    tuning.enable()
    dev = torch.device("cuda", 0)
    shapes = []
    for (tok, C) in ((32768, 320), (8192, 640), (2048, 1280)):
        shapes += [(tok, C, C, True, f"{C}->{C} (to_q/to_out/proj) M={tok}"), (tok, C, 8 * C, True, f"{C}->{8 * C} (ff GEGLU proj) M={tok}"),
                   (tok, 4 * C, C, True, f"{4 * C}->{C} (ff out) M={tok}")]
    shapes += [(32768, 512, 512, True, "512->512 (VAE mid attention) M=32768"), (616, 768, 320, False, "768->320 (to_k/to_v of the context) M=616"),
               (616, 768, 1280, False, "768->1280 (to_k/to_v) M=616")]
    tot_f = tot_b = 0.0
    for (M, K, N, bias, name) in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev) if bias else None
        dy = torch.randn(M, N, device=dev)
        tf = timed(lambda: F.linear(x, w, b))
        tb = timed(lambda: dy @ w)
        fl = 2.0 * M * K * N
        print(f"{name:48s} fwd {tf * 1e6:8.1f} us {fl / tf / 1e12:6.1f} TF/s ({fl / tf / 1e12 / 157.3:.2f}) | bwd-data {tb * 1e6:8.1f} us {fl / tb / 1e12:6.1f} TF/s ({fl / tb / 1e12 / 157.3:.2f})")


if __name__ == "__main__":
    main()
