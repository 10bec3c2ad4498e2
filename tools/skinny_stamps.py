"""Reader for the in-kernel phase stamps of the small-M GEMM (round-2 study).  Needs a STAMPED build of
gemm_f32.hip (clock64() stamps accumulated into a __device__ array by workgroup 0 and an exported nast_skinny_dbg(); the
patch is not part of the product kernel): on the shipped library this script says so and exits."""
import ctypes as C, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from helpers import make_param
from nast_amd import ops, transformer as tf, _lib
lib = C.CDLL(os.path.join("nast_amd", "lib", "libnast_hip.so"))
if not hasattr(lib, "nast_skinny_dbg"):
    sys.exit("this libnast_hip.so carries no stamps (see the docstring)")
torch.manual_seed(0)
F = tf.StyleTransformer(make_param(V=9600, d=256, h=4, N=4, L=32, learnable=True)).cuda()
lp = F.length_predictor
for B in (512, 64):
    memory = torch.randn(B, 33, 256).cuda(); lengths = torch.randint(4, 33, (B,)).cuda()
    eos, unk = torch.randn(256).cuda(), torch.randn(256).cuda(); buf = torch.zeros(B, 33, dtype=torch.long, device="cuda")
    out = (C.c_ulonglong * 8)()
    with torch.no_grad(), ops.async_decode(None):
        lp._native_decode(memory, lengths, eos, unk, buf); lib.nast_skinny_dbg(out)
        lp._native_decode(memory, lengths, eos, unk, buf); lib.nast_skinny_dbg(out)
    n = out[7]
    names = ["setup", "loads", "LN", "MFMA", "reduce+barrier", "epilogue"]
    print(f"B={B}: {n} launches; cycles per launch (workgroup 0, thread 0):", {k: round(out[i] / n) for i, k in enumerate(names)}, "total", round(sum(out[:6]) / n))
