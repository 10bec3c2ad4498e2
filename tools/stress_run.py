"""f_step at BASELINE configs[4]'s shape (V=32000, d=1024, h=16, 8 layers, len<=128, learnable) -- timing aid:
   python tools/stress_run.py [sentences] [steps] [fp32|bf16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from nast_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
cfg = dict(V=32000, d=1024, h=16, N=8, L=128, learnable=True)
model = bench.build_model(cfg, "cuda")
ops.set_gemm_precision(prec)
data = [bench.synth_incoming(B, 128, 32000, 9000 + i, "cuda") for i in range(2)]
for i in range(2):
    model.f_step(data[i % 2], 1.0, use_inference_z=True)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(steps):
    model.f_step(data[i % 2], 1.0, use_inference_z=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f"configs[4] shape, {B} sentences, {prec}: {dt * 1e3:.1f} ms/step, {B / dt:.1f} sentences/s, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
