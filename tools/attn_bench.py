#!/usr/bin/env python3
"""Kernel-only timing of the fused attention forward / backward at the training step's shape (run on the GPU box).
usage: attn_bench.py [B] [n] [d] [h] [iters] [fwd|bwd|both] [padded|packed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nast_amd import ops  # noqa: E402


def main():
    a = sys.argv[1:]
    B = int(a[0]) if len(a) > 0 else 512
    n = int(a[1]) if len(a) > 1 else 33
    d = int(a[2]) if len(a) > 2 else 256
    h = int(a[3]) if len(a) > 3 else 4
    iters = int(a[4]) if len(a) > 4 else 50
    which = a[5] if len(a) > 5 else "both"
    dev = "cuda"
    torch.manual_seed(0)
    # q/k/v as the step has them: column blocks of one [B, n, 3d] projection output are NOT used (separate tensors)
    q, k, v, do = (torch.randn(B, n, d, device=dev) for _ in range(4))
    lens = torch.randint(4, n + 1, (B,), device=dev)
    lens[0] = n
    mask = (torch.arange(n, device=dev)[None, :] >= lens[:, None]).view(B, 1, 1, n)
    if os.environ.get("AB_RAGGED_DO", "1") != "0":       # padded target positions carry no loss in the training step
        do = do * (torch.arange(n, device=dev)[None, :] < lens[:, None]).unsqueeze(-1)
    packed = len(a) > 6 and a[6] == "packed"
    if packed:        # what the training step runs at 512 sentences: packed real rows, no mask
        ps = ops.PackedSeq(lens, n, int(lens.sum()))
        pair = ops.PackedPair(ps, ps)
        q, k, v, do = (torch.randn(ps.rows, d, device=dev) for _ in range(4))
        out, lse = ops.mha_fwd_packed(q, k, v, pair, h)
        fns = (("fwd", lambda: ops.mha_fwd_packed(q, k, v, pair, h)), ("bwd", lambda: ops.mha_bwd_packed(q, k, v, pair, h, do, lse, out)))
    else:
        out, lse = ops.mha_fwd(q, k, v, mask, h)
        fns = (("fwd", lambda: ops.mha_fwd(q, k, v, mask, h)), ("bwd", lambda: ops.mha_bwd(q, k, v, mask, h, out, do, lse)))
    fns[1][1]()
    torch.cuda.synchronize()
    for name, fn in fns:
        if which not in (name, "both"):
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        byts = (4 if name == "fwd" else 8) * (q.shape[0] if packed else B * n) * d * 4
        print(f"attention {name}{' packed' if packed else ''}: B={B} n={n} d={d} h={h}  {us:8.2f} us  {byts / us / 1e6:7.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
