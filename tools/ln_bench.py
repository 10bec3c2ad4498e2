#!/usr/bin/env python3
"""Kernel-only timing of LayerNorm forward / backward at the training step's row count (run on the GPU box).
usage: ln_bench.py [rows] [d] [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nast_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 9920
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.manual_seed(0)
xs = [torch.randn(rows, d, device="cuda").requires_grad_(True) for _ in range(8)]     # rotate operands: ~80 MB, past the L2s
pg = os.environ.get("LN_FROZEN") != "1"     # LN_FROZEN=1: no parameter gradients, i.e. the kernel without its atomic tail
w = torch.ones(d, device="cuda", requires_grad=pg); b = torch.zeros(d, device="cuda", requires_grad=pg)
R = torch.randn(rows, d, device="cuda")
ys = [ops.layer_norm(x, w, b) for x in xs]
for name in ("fwd", "bwd"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters):
        if name == "fwd":
            with torch.no_grad():
                ops.layer_norm(xs[i % 8], w, b)
        else:
            ys[i % 8].backward(R, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"layer_norm {name}: rows={rows} d={d}  {us:7.2f} us per call (host loop; kernel-only figures: rocprofv3 --kernel-trace --stats)")
