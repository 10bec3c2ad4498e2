#!/usr/bin/env python3
"""Host-side profile (cProfile) of f_step at a small batch, where the step is bound by the host launch rate.
usage: cpu_profile_step.py [B] [steps]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
    model = bench.build_model(cfg, "cuda")
    batches = [bench.synth_incoming(B, 32, 9600, i, "cuda") for i in range(4)]
    np.random.seed(0)
    for i in range(3):
        model.f_step(batches[i % 4], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(steps):
        model.f_step(batches[i % 4], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(60)


if __name__ == "__main__":
    main()
