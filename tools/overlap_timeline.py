"""Where the two streams of f_step are in time, without a profiler (rocprofv3's kernel trace serialises the two queues):
HIP events around the decode (side stream) and around the self-reconstruction pass issued from the decode hook (main
stream), relative to the start of the step.   python tools/overlap_timeline.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from nast_amd import ops

cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
model = bench.build_model(cfg, "cuda")
data = [bench.synth_incoming(512, 32, 9600, 1000 + i, "cuda") for i in range(4)]
lp = model.net.model_F.length_predictor
ev = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()                      # on the current stream
    ev[name] = e


orig_decode = lp._native_decode
def decode(*a, **k):
    mark("decode_start")
    r = orig_decode(*a, **k)
    mark("decode_end")
    return r
lp._native_decode = decode

orig_def = ops.run_deferred
def run_deferred():
    fn = ops._deferred
    if fn is not None:
        def wrapped():
            with torch.cuda.stream(torch.cuda.default_stream()):
                mark("pass1_start")
            fn()
            with torch.cuda.stream(torch.cuda.default_stream()):
                mark("pass1_end")
        ops._deferred = wrapped
    orig_def()
ops.run_deferred = run_deferred

for i in range(5):
    model.f_step(data[i % 4], 1.0, use_inference_z=True)
for i in range(3):
    torch.cuda.synchronize()
    ev.clear()
    mark("step_start")
    model.f_step(data[i % 4], 1.0, use_inference_z=True)
    mark("step_end")
    torch.cuda.synchronize()
    t0 = ev["step_start"]
    print({k: round(t0.elapsed_time(v), 2) for k, v in ev.items() if k != "step_start"}, flush=True)
