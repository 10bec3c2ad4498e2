"""Where do the HIP path and the CPU oracle differ?  For several seeds: per-tensor gradient error, and for the worst tensors how
the error is distributed (a flipped ReLU gate -- a pre-activation within rounding distance of zero -- shows up as one row of an
FFN weight gradient / one bias element, not as a dense error)."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 4]
for seed in seeds:
    rep = T._oracle_vs_hip_step(T.BENCH_CFG, B=B, seed=seed, name=f"diag_b{B}_s{seed}")
    model, tr = T._oracle_vs_hip_step.last
    named = dict(model.net.named_parameters())
    gref = {k: v for k, v in tr.last_grads.items() if v is not None}
    gmax = max(float(v.abs().max()) for v in gref.values())
    rows = []
    for k, gr in gref.items():
        gg = named[k].grad.detach().cpu()
        d = (gg - gr).abs()
        ts = max(float(gr.abs().max()), 1e-3 * gmax)
        rows.append((float(d.max()) / ts, k, d, ts))
    rows.sort(key=lambda r: -r[0])
    print(f"== seed {seed}: worst tensors")
    for e, k, d, ts in rows[:6]:
        over = d > 1e-4 * ts
        info = f"   {e:.2e} {k} shape {tuple(d.shape)} elems>1e-4: {int(over.sum())}/{d.numel()}"
        if d.dim() == 2:
            per_row = (d / ts).max(1).values
            top = torch.topk(per_row, min(3, per_row.numel()))
            info += f"  worst rows {top.indices.tolist()} -> {[f'{v:.1e}' for v in top.values.tolist()]}; median row {float(per_row.median()):.1e}"
        else:
            top = torch.topk(d / ts, min(3, d.numel()))
            info += f"  worst elems {top.indices.tolist()} -> {[f'{v:.1e}' for v in top.values.tolist()]}; median {float((d / ts).median()):.1e}"
        print(info)
    del model, tr
    torch.cuda.empty_cache()
