"""Small-M GEMM (decode of a wide model) under rocprofv3: python tools/lab/skinny_lab.py M N K nweights
   launches the same problem over `nweights` rotating weight matrices (1 = hot in L2; 64 x 4 MB = past the Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nast_amd import ops
M, N, K, NWT = (int(a) for a in sys.argv[1:5])
x = torch.randn(M, K, device="cuda")
Ws = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(NWT)]
b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 0
    for it in range(max(1, 256 // NWT)):
        for W in Ws:
            ops.gemm([ops.d_fwd(x, W, y, bias=b)])
            n += 1
    e1.record(); torch.cuda.synchronize()
print(f"M={M} N={N} K={K} weights={NWT} wide={os.environ.get('NAST_GEMM_SKINNY_WIDE','1')}: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch (back to back)")
