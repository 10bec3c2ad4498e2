"""Run the 64-sentence f_step (eager or replayed from a HIP graph) a few times: the workload for
`rocprofv3 --kernel-trace --stats -- python3 tools/small_batch_run.py [graph|eager] [steps]`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
model = bench.build_model(cfg, "cuda")
data = [bench.synth_incoming(64, 32, 9600, 5000 + i, "cuda") for i in range(4)]
model.graph_steps = mode == "graph"
if os.environ.get("NAST_NO_OVERLAP") == "1":
    model.graph_overlap = False
    model.overlap_passes = False
for i in range(4):
    model.f_step(data[i % 4], 1.0, use_inference_z=True)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(steps):
    model.f_step(data[i % 4], 1.0, use_inference_z=True)
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t) / steps * 1e3:.3f} ms/step over {steps} steps", flush=True)
