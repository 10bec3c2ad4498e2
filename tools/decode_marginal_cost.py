"""What does the skip decode cost the overlapped f_step?  The step is timed as it is, and again with the decode's kernels
removed (its result -- the alignment positions of the same batch from a previous real decode -- is copied in instead), with
the two-stream overlap on and off.   python tools/decode_marginal_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

cfg = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)
model = bench.build_model(cfg, "cuda")
data = [bench.synth_incoming(512, 32, 9600, 1000, "cuda")]          # one batch: its decode result can be replayed
lp = model.net.model_F.length_predictor
real = lp._native_decode
store = {}


def recording(memory, origin_length, eos, unk, allpos_buf, **k):
    r = real(memory, origin_length, eos, unk, allpos_buf, **k)
    store["allpos"], store["steps"] = allpos_buf.clone(), r
    return r


def replaying(memory, origin_length, eos, unk, allpos_buf, **k):
    allpos_buf.copy_(store["allpos"])
    return store["steps"]


def timed(n=20):
    for _ in range(3):
        model.f_step(data[0], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        model.f_step(data[0], 1.0, use_inference_z=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for overlap in (True, False):
    model.overlap_passes = overlap
    lp._native_decode = recording
    a = timed()
    lp._native_decode = replaying          # (weights move a little between steps; the positions are only stand-ins of the right shape)
    b = timed()
    print(f"overlap={overlap}: step {a:.2f} ms, without the decode's kernels {b:.2f} ms -> the decode costs {a - b:.2f} ms", flush=True)
