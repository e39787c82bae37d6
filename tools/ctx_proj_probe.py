"""Context projections of the cross-attention layers (k = to_k(ctx), v = to_v(ctx), ctx [1, T, 768], ptp_utils.py:513-520): per
layer as the modules run them vs ONE GEMM against the concatenated weights of every layer of the step."""
import sys, torch
sys.path.insert(0, "/root/repo")
import stablekeypoints_amd.ops as ops
from stablekeypoints_amd import tuning
tuning.enable()
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
F = torch.nn.functional
for T in (77, 500):
    ctx = torch.randn(1, T, 768, device="cuda")
    layers = [320, 320, 640, 640, 1280, 1280, 1280, 1280, 1280, 1280, 640]      # SD-1.5 cross-attention layers up to the early exit
    ws = [torch.randn(c, 768, device="cuda") * 768 ** -0.5 for c in layers for _ in range(2)]
    for c in (320, 640, 1280):
        w = torch.randn(c, 768, device="cuda")
        print(f"T {T}: F.linear [1,{T},768] -> {c}: {timeit(lambda: F.linear(ctx, w)):6.1f} us", flush=True)
    t_sep = timeit(lambda: [F.linear(ctx, w) for w in ws])
    wcat = torch.cat(ws, 0).contiguous()
    t_cat = timeit(lambda: F.linear(ctx, wcat))
    d = torch.randn(T, wcat.shape[0], device="cuda")
    ds = [torch.randn(1, T, w.shape[0], device="cuda") for w in ws]
    def bsep():
        acc = None
        for g, w in zip(ds, ws):
            t = torch.matmul(g, w)
            acc = t if acc is None else acc + t
        return acc
    t_bsep = timeit(bsep)
    t_bcat = timeit(lambda: torch.mm(d, wcat))
    print(f"T {T}: {len(ws)} projections one by one {t_sep:7.1f} us, one GEMM [{T} x 768] x [768 x {wcat.shape[0]}] {t_cat:6.1f} us | "
          f"input gradient one by one (+ adds) {t_bsep:7.1f} us, one GEMM {t_bcat:6.1f} us", flush=True)
