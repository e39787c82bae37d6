import torch, time, sys
sys.path.insert(0, "/root/repo")
import stablekeypoints_amd.ops as ops   # loads the TunableOp file as the step does
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
for rows in (8, 2):
  for (npx, C) in ((4096, 320), (1024, 640), (256, 1280)):
    M = rows * npx
    x = torch.randn(rows, npx, C, device="cuda"); ws = [torch.randn(C, C, device="cuda") * C ** -0.5 for _ in range(3)]
    wf = torch.cat(ws, 0).contiguous()
    F = torch.nn.functional
    t3 = timeit(lambda: [F.linear(x, w) for w in ws])
    t1 = timeit(lambda: F.linear(x, wf))
    d = [torch.randn(M, C, device="cuda") for _ in range(3)]; df = torch.randn(M, 3 * C, device="cuda")
    def b3():
        dx = torch.mm(d[0], ws[0]); dx.addmm_(d[1], ws[1]); dx.addmm_(d[2], ws[2]); return dx
    tb3 = timeit(b3); tb1 = timeit(lambda: torch.mm(df, wf))
    print(f"rows {rows} M {M:6d} C {C:5d}: fwd 3 GEMMs {t3:7.1f} us, fused {t1:7.1f} us | bwd 3 (accumulating) {tb3:7.1f} us, fused {tb1:7.1f} us", flush=True)

print("bmm with a broadcast A (batch stride 0) -> [3, M, C] contiguous q | k | v")
F = torch.nn.functional
for rows in (8, 2):
  for (npx, C) in ((4096, 320), (1024, 640), (256, 1280)):
    M = rows * npx
    x = torch.randn(M, C, device="cuda"); w3 = torch.randn(3, C, C, device="cuda") * C ** -0.5
    w3t = w3.transpose(1, 2)                                   # [3, C(in), C(out)] view: y_i = x . w_i^T
    out = torch.empty(3, M, C, device="cuda")
    f = lambda: torch.bmm(x.unsqueeze(0).expand(3, M, C), w3t, out=out)
    t = timeit(f)
    ref = F.linear(x, w3[1])
    err = (out[1] - ref).abs().max().item()
    w3c = w3t.contiguous()
    t2 = timeit(lambda: torch.bmm(x.unsqueeze(0).expand(3, M, C), w3c, out=out))
    print(f"rows {rows} M {M:6d} C {C:5d}: bmm(expand, W^T view) {t:7.1f} us, bmm(expand, W^T contiguous) {t2:7.1f} us, err {err:.1e}", flush=True)
