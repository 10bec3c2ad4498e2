import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import make_param
from nast_amd import ops, transformer as tf
torch.manual_seed(0)
F = tf.StyleTransformer(make_param(V=9600, d=256, h=4, N=4, L=32, learnable=True)).cuda()
lp = F.length_predictor
B, S = 512, 32
memory = torch.randn(B, S + 1, 256).cuda(); lengths = torch.randint(4, S + 1, (B,)).cuda()
eos, unk = torch.randn(256).cuda(), torch.randn(256).cuda()
buf = torch.zeros(B, 33, dtype=torch.long, device="cuda")
with torch.no_grad(), ops.async_decode(None):
    for _ in range(3):
        lp._native_decode(memory, lengths, eos, unk, buf)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); lp._native_decode(memory, lengths, eos, unk, buf); ts.append(time.perf_counter() - t); torch.cuda.synchronize()
print("host enqueue time of one decode (ms):", [round(x * 1e3, 2) for x in ts])
# the same decode captured once into a HIP graph: host time of a replay, and GPU time of both forms
g = torch.cuda.CUDAGraph()
with torch.no_grad(), ops.async_decode(None):
    with torch.cuda.graph(g):
        lp._native_decode(memory, lengths, eos, unk, buf)
    torch.cuda.synchronize()
    ref = buf.clone()
    ts = []
    for _ in range(5):
        buf.zero_(); torch.cuda.synchronize()
        t = time.perf_counter(); g.replay(); ts.append(time.perf_counter() - t); torch.cuda.synchronize()
    assert torch.equal(buf, ref)
    print("host time of one graph replay (ms):", [round(x * 1e3, 2) for x in ts])
    for name, fn in (("eager", lambda: lp._native_decode(memory, lengths, eos, unk, buf)), ("graph", g.replay)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(name, "GPU ms per decode:", round(e0.elapsed_time(e1) / 5, 3))
