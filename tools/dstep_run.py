#!/usr/bin/env python3
"""Runs only the discriminator step (for rocprofv3 --kernel-trace --stats): dstep_run.py [steps] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
    torch.cuda.set_device(0)
    model = bench.build_model(cfg, "cuda")
    if os.environ.get("NO_OVERLAP") == "1":
        model.overlap_passes = False
    batches = [bench.synth_incoming(B, 32, cfg["V"], 1000 + i, "cuda") for i in range(4)]
    import time
    for i in range(3):
        model.d_step(batches[i % 4], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        model.d_step(batches[i % 4], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    print(f"d_step: {(time.perf_counter() - t) / steps * 1e3:.2f} ms/step at {B} sentences")


if __name__ == "__main__":
    main()
