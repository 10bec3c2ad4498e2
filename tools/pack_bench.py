#!/usr/bin/env python3
"""200 PackedSeq constructions at the training step's shape (run under rocprofv3 --kernel-trace --stats on the GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nast_amd import ops  # noqa: E402
lens = torch.randint(6, 35, (512,), device="cuda")
tot = int(lens.sum())
for _ in range(200):
    ops.PackedSeq(lens, 34, tot)
torch.cuda.synchronize()
