"""d = 160 flash attention at the 16^2 layers' shape (N = 256, 8 heads, 8 / 2 rows): forward and backward time of the build's variants."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from stablekeypoints_amd import ops
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for B in (8, 2):
    H, N, d = 8, 256, 160
    g = torch.Generator().manual_seed(0)
    q, k, v, go = (torch.randn(B, N, H * d, generator=g).cuda() for _ in range(4))
    qg = q.clone().requires_grad_(True); kg = k.clone().requires_grad_(True); vg = v.clone().requires_grad_(True)
    tf = timeit(lambda: ops.self_attention(q, k, v, H, d ** -0.5))
    out = ops.self_attention(qg, kg, vg, H, d ** -0.5)
    tb = timeit(lambda: torch.autograd.grad(out, (qg, kg, vg), go, retain_graph=True))
    print(f"variant {os.environ.get('SKP_FA2_VARIANT', '0')} rows {B}: fwd {tf:6.1f} us  bwd {tb:6.1f} us", flush=True)
