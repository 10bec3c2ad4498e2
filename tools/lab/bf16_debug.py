import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nast_amd import ops
M, N, K = 4100, 1024, 256
g = torch.Generator().manual_seed(M + N + K)
x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
for relu in (False, True):
    ref = torch.nn.functional.linear(x, W, b)
    if relu: ref = torch.relu(ref)
    with ops.gemm_precision("bf16"):
        y = ops.linear(x.cuda(), W.cuda(), b.cuda(), None, relu).cpu()
    err = (y - ref).abs()
    print("relu", relu, "max err", err.max().item(), "at", divmod(int(err.argmax()), N), "count > 0.1:", int((err > 0.1).sum()))
    bad = (err > 0.1).nonzero()
    if len(bad):
        print("rows", bad[:, 0].min().item(), bad[:, 0].max().item(), "cols", bad[:, 1].min().item(), bad[:, 1].max().item())
        print(bad[:10].tolist())
R = torch.randn(M, N, generator=g)
gx, gW, gb = (t.cuda().requires_grad_(True) for t in (x, W, b))
with ops.gemm_precision("bf16"):
    y = ops.linear(gx, gW, gb, None, True)
    (y * R.cuda()).sum().backward()
mask = (y.detach().cpu() > 0).float()
dy = R * mask
for name, got, ref in (("gx", gx.grad.cpu(), dy @ W), ("gW", gW.grad.cpu(), dy.t() @ x), ("gb", gb.grad.cpu(), dy.sum(0))):
    err = (got - ref).abs()
    print(name, "max err", err.max().item(), "scale", ref.abs().max().item(), "at", divmod(int(err.argmax()), ref.shape[-1] if ref.dim() > 1 else 1),
          "bad", int((err > 0.02 * ref.abs().max()).sum()))
    if name == "gx":
        bad = (err > 0.02 * ref.abs().max()).nonzero()
        if len(bad): print(" rows", bad[:, 0].min().item(), bad[:, 0].max().item(), "cols", bad[:, 1].min().item(), bad[:, 1].max().item(), bad[:8].tolist())
