"""GPU time of the skip decode alone (nast_lp_decode_f32, all L steps, no read-backs): the cluster kernel of decode_cluster.hip
(and, with DECODE_BENCH_FUSED=1, the one-workgroup-per-block kernel of decode_fused.hip) against the step-by-step path, at 64 and 512 sentences of the bench shape.   python tools/decode_bench.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from helpers import make_param
from nast_amd import ops, transformer as tf

Bs = [int(x) for x in sys.argv[1:]] or [64, 512]
L = S = 32
d = 256
torch.manual_seed(0)
F = tf.StyleTransformer(make_param(V=9600, d=d, h=4, N=4, L=L, learnable=True)).cuda()
lp = F.length_predictor
for B in Bs:
    memory = torch.randn(B, S + 1, d).cuda()
    lengths = torch.randint(4, S + 1, (B,)).cuda()
    eos, unk = torch.randn(d).cuda(), torch.randn(d).cuda()
    buf = torch.zeros(B, L + 1, dtype=torch.long, device="cuda")
    for mode in ("step", "fused", "cluster"):
        if mode == "fused" and os.environ.get("DECODE_BENCH_FUSED", "0") == "0":
            continue
        os.environ["NAST_DECODE_FUSED"] = "1" if mode == "fused" else "0"
        os.environ["NAST_DECODE_CLUSTER"] = "1" if mode == "cluster" else "0"
        with torch.no_grad(), ops.async_decode(None):
            for _ in range(2):
                lp._native_decode(memory, lengths, eos, unk, buf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lp._native_decode(memory, lengths, eos, unk, buf)
            e1.record()
            torch.cuda.synchronize()
        print(f"B={B} {mode}: {e0.elapsed_time(e1) / 5:.3f} ms per decode", flush=True)
