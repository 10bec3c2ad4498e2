"""CU-masked streams (hipExtStreamCreateWithCUMask): can the latency-bound decode get its own CUs so that it runs BESIDE the
big GEMMs of the other pass instead of queueing behind their workgroups?  Decode alone (all CUs / N CUs), a GEMM-heavy
loop alone (all CUs / the complement), then both together, unmasked and partitioned.   python tools/cu_mask_lab.py [ncu]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from helpers import make_param
from nast_amd import ops, transformer as tf

NCU = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.zeros(1, device="cuda")
def _loaded_hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return line.split()[-1]
    return "libamdhip64.so"


hip = C.CDLL(_loaded_hip())           # the runtime instance torch itself uses
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


# mask bit i = CU (i // 8) of XCD (i % 8); every XCD must keep at least one CU or the runtime refuses the mask (tools/mask_probe.py)
small = list(range(NCU))
big = list(range(NCU, 256))
s_small, s_big = masked_stream(small), masked_stream(big)
s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()

torch.manual_seed(0)
F = tf.StyleTransformer(make_param(V=9600, d=256, h=4, N=4, L=32, learnable=True)).cuda()
lp = F.length_predictor
B, S = 512, 32
memory = torch.randn(B, S + 1, 256).cuda(); lengths = torch.randint(4, S + 1, (B,)).cuda()
eos, unk = torch.randn(256).cuda(), torch.randn(256).cuda()
buf = torch.zeros(B, 33, dtype=torch.long, device="cuda")
x = torch.randn(9728, 256, device="cuda"); W = torch.randn(1024, 256, device="cuda"); y = torch.empty(9728, 1024, device="cuda")


def decode(stream):
    with torch.cuda.stream(stream), torch.no_grad(), ops.async_decode(None):
        lp._native_decode(memory, lengths, eos, unk, buf)


def gemms(stream, n=250):          # ~13 ms of the step's FFN-1 GEMM
    with torch.cuda.stream(stream):
        for _ in range(n):
            ops.gemm([ops.d_fwd(x, W, y)])


def run(label, fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for f in fns:
        f()
    torch.cuda.synchronize()
    print(f"{label:58s} {(time.perf_counter() - t) * 1e3:7.2f} ms", flush=True)


run("decode alone, all CUs", [lambda: decode(s_a)])
run(f"decode alone, {NCU} CUs", [lambda: decode(s_small)])
run("GEMM loop alone, all CUs", [lambda: gemms(s_b)])
run(f"GEMM loop alone, {256 - NCU} CUs", [lambda: gemms(s_big)])
run("decode + GEMM loop, two plain streams", [lambda: decode(s_a), lambda: gemms(s_b)])
run(f"decode ({NCU} CUs) + GEMM loop ({256 - NCU} CUs)", [lambda: decode(s_small), lambda: gemms(s_big)])
run(f"decode ({NCU} CUs) + GEMM loop (all CUs)", [lambda: decode(s_small), lambda: gemms(s_b)])
