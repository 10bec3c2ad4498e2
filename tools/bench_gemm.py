"""Micro-benchmark of nast_gemm_f32 on the shapes the training step uses (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nast_amd import ops

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us

def main():
    dev = "cuda"
    rows = []
    for (M, N, K) in [(64, 256, 256), (64, 1024, 256), (64, 256, 1024), (512, 256, 256), (2112, 256, 256), (2112, 1024, 256),
                      (2112, 256, 1024), (2048, 9600, 256), (16896, 256, 256), (16896, 1024, 256), (16896, 256, 1024), (16384, 9600, 256)]:
        x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
        dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        fl = 2.0 * M * N * K
        t_f = timeit(lambda: ops.gemm([ops.d_fwd(x, W, y, bias=b)]))
        t_d = timeit(lambda: ops.gemm([ops.d_dgrad([dy], [W], dx)]))
        t_w = timeit(lambda: ops.gemm([ops.d_wgrad(dy, x, dW, db)]))
        t_t = timeit(lambda: torch.nn.functional.linear(x, W, b))
        print(f"M={M:6d} N={N:5d} K={K:5d}  fwd {t_f:8.1f}us {fl/t_f/1e6:7.1f}TF | dgrad {t_d:8.1f}us {fl/t_d/1e6:7.1f}TF | wgrad {t_w:8.1f}us {fl/t_w/1e6:7.1f}TF | torch(hipblaslt) fwd {t_t:8.1f}us {fl/t_t/1e6:7.1f}TF", flush=True)
    # grouped: QKV
    M, d = 2112, 256
    z = torch.randn(M, d, device=dev); Ws = [torch.randn(d, d, device=dev) for _ in range(3)]; ys = [torch.empty(M, d, device=dev) for _ in range(3)]
    t = timeit(lambda: ops.gemm([ops.d_fwd(z, Ws[i], ys[i]) for i in range(3)]))
    print(f"grouped QKV M={M}: {t:.1f}us {3*2.0*M*d*d/t/1e6:.1f}TF")
    # soft embed NN: [M,V]x[V,d]
    M, V = 2048, 9600
    s = torch.randn(M, V, device=dev); E = torch.randn(V, d, device=dev); o = torch.empty(M, d, device=dev)
    t = timeit(lambda: ops.gemm([ops.desc(s, E, o, M, d, V, V, d, d, 0, 0)]))
    print(f"soft-embed NN M={M} K={V} N={d}: {t:.1f}us {2.0*M*V*d/t/1e6:.1f}TF")
    dE = torch.zeros(V, d, device=dev); do = torch.randn(M, d, device=dev)
    t = timeit(lambda: ops.gemm([ops.desc(s, do, dE, V, d, M, V, d, d, 1, 0, flags=2)]))
    print(f"soft-embed wgrad TN M={V} N={d} K={M}: {t:.1f}us {2.0*M*V*d/t/1e6:.1f}TF")

if __name__ == "__main__":
    main()


def cold():
    """decode-like: many distinct weight matrices in rotation (L2-cold), M = 64"""
    dev = "cuda"
    for (N, K) in [(256, 256), (1024, 256), (256, 1024)]:
        Ws = [torch.randn(N, K, device=dev) for _ in range(64)]
        xs = [torch.randn(64, K, device=dev) for _ in range(8)]
        y = torch.empty(64, N, device=dev)
        ds = [ops.d_fwd(xs[i % 8], Ws[i], y) for i in range(64)]
        arrs = [(ops.GemmDesc * 1)(d) for d in ds]
        st = ops._stream()
        def run():
            for a in arrs:
                ops.call("nast_gemm_f32", a, 1, st)
        t = timeit(run, iters=10) / 64
        print(f"cold rotation M=64 N={N} K={K}: {t:.1f} us per launch (pre-built descs)")

if __name__ == "__main__":
    cold()
