#!/usr/bin/env python3
"""Where do the full-size elementwise adds / fills of one f_step come from?  (torch profiler, shapes + Python stacks)"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
    torch.cuda.set_device(0)
    model = bench.build_model(cfg, "cuda")
    model.overlap_passes = False
    batches = [bench.synth_incoming(512, 32, cfg["V"], 1000 + i, "cuda") for i in range(2)]
    for i in range(3):
        model.f_step(batches[i % 2], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        model.f_step(batches[0], 1.0, use_inference_z=True)
        torch.cuda.synchronize()
    want = sys.argv[1:] or ["aten::add", "aten::add_", "aten::zeros", "aten::zeros_like", "aten::zero_", "aten::fill_"]
    for name in want:
        c = Counter()
        for e in prof.events():
            if e.name != name:
                continue
            shp = tuple(tuple(s) if isinstance(s, (list, tuple)) else s for s in (e.input_shapes or []))[:2]
            st = [f for f in (e.stack or []) if "nast_amd" in f or "autograd" in f][:3]
            c[(str(shp), " <- ".join(s.split("/")[-1] for s in st))] += 1
        print(f"===== {name}: {sum(c.values())} calls")
        for (shp, st), n in c.most_common(14):
            print(f"  {n:4d}  {shp:40s} {st}")


if __name__ == "__main__":
    main()
