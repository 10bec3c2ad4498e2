"""Per-workgroup segment times of the tiled GEMM kernel on one of the step's problems (round-2 study).  Needs a STAMPED build:
    rm nast_amd/csrc/gemm_f32.o && make -C nast_amd/csrc EXTRA=-DGEMM_STAMPS        (then rebuild plainly afterwards)
On the shipped library this script says so and exits.   usage: gemm_stamps.py [M] [N] [K] [fwd|fwdres|dgrad]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nast_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nast_amd", "lib", "libnast_hip.so"))
if not hasattr(lib, "nast_gemm_stamps_read"):
    sys.exit("this libnast_hip.so carries no stamps (see the docstring)")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 9920
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
kind = sys.argv[4] if len(sys.argv) > 4 else "fwdres"
torch.manual_seed(0)
R = 8                                                     # rotate operands (~80+ MB): nothing of the next call is in an L2
xs = [torch.randn(M, K, device="cuda") for _ in range(R)]
rs = [torch.randn(M, N, device="cuda") for _ in range(R)]
ys = [torch.empty(M, N, device="cuda") for _ in range(R)]
W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
def call(i):
    if kind == "fwd":
        ops.gemm([ops.d_fwd(xs[i % R], W, ys[i % R], bias=b)])
    elif kind == "fwdres":
        ops.gemm([ops.d_fwd(xs[i % R], W, ys[i % R], bias=b, residual=rs[i % R])])
    else:
        ops.gemm([ops.d_dgrad([rs[i % R]], [W], xs[i % R])])
for i in range(20):
    call(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(100):
    call(i)
e1.record(); torch.cuda.synchronize()
print(f"{kind} M={M} N={N} K={K}: {e0.elapsed_time(e1) * 10:.2f} us per call (host loop)")
W_ = min(((M + 63) // 64) * ((N + 63) // 64), 8192)
buf = (C.c_ulonglong * (W_ * 8))()
assert lib.nast_gemm_stamps_read(buf, W_ * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(W_, 8).astype(np.int64)
names = ["start -> tile found, epilogue operands requested", "first K-tile: loads + LDS stores", "barrier", "K loop", "epilogue"]
seg = np.diff(t[:, :6], axis=1)
print(f"{W_} workgroups; s_memtime cycles / 2400 -> us (thread 0 of each workgroup)")
for i, nm in enumerate(names):
    print(f"  {nm:52s} mean {seg[:, i].mean() / 2400:6.2f}   p10 {np.percentile(seg[:, i], 10) / 2400:6.2f}   p90 {np.percentile(seg[:, i], 90) / 2400:6.2f}")
life = (t[:, 5] - t[:, 0]) / 2400
print(f"  workgroup lifetime mean {life.mean():.2f} us, p90 {np.percentile(life, 90):.2f}")
