"""Per-workgroup segment times of the attention backward kernel (round-2 study).  Needs a STAMPED build of the library:
    rm nast_amd/csrc/attention_mfma.o && make -C nast_amd/csrc EXTRA=-DATTN_STAMPS     (then rebuild plainly afterwards)
On the shipped library this script says so and exits.   usage: attn_stamps.py [B] [n] [packed|padded]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nast_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nast_amd", "lib", "libnast_hip.so"))
if not hasattr(lib, "nast_attn_stamps_read"):
    sys.exit("this libnast_hip.so carries no stamps (see the docstring)")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 34
d, h = 256, 4
torch.manual_seed(0)
q, k, v, do = (torch.randn(B, n, d, device="cuda") for _ in range(4))
lens = torch.randint(4, n + 1, (B,), device="cuda"); lens[0] = n
mask = (torch.arange(n, device="cuda")[None, :] >= lens[:, None]).view(B, 1, 1, n)
do = do * (torch.arange(n, device="cuda")[None, :] < lens[:, None]).unsqueeze(-1)
if len(sys.argv) > 3 and sys.argv[3] == "packed":     # what the training step runs at 512 sentences: packed real rows, no mask
    ps = ops.PackedSeq(lens, n, int(lens.sum()))
    pair = ops.PackedPair(ps, ps)
    q, k, v, do = (torch.randn(ps.rows, d, device="cuda") for _ in range(4))
    out, lse = ops.mha_fwd_packed(q, k, v, pair, h)
    for _ in range(3):
        ops.mha_bwd_packed(q, k, v, pair, h, do, lse, out)
else:
    out, lse = ops.mha_fwd(q, k, v, mask, h)
    for _ in range(3):
        ops.mha_bwd(q, k, v, mask, h, out, do, lse)
torch.cuda.synchronize()
W = min(B * h, 8192)
buf = (C.c_ulonglong * (W * 8))()
assert lib.nast_attn_stamps_read(buf, W * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(W, 8).astype(np.int64)
names = ["start -> mask loads issued", "K/V loads + LDS stores", "lse, flags, barrier", "phase 1 + barrier", "Q/dO restage + barrier", "phase 2"]
seg = np.diff(t[:, :7], axis=1)   # s_memtime counts shader cycles; the division below assumes 2.4 GHz
print(f"B={B} n={n}: {W} workgroups; cycles / 2400 -> us")
for i, nm in enumerate(names):
    print(f"  {nm:28s} mean {seg[:, i].mean() / 2400:7.2f} us   p10 {np.percentile(seg[:, i], 10) / 2400:6.2f}   p90 {np.percentile(seg[:, i], 90) / 2400:6.2f}")
life = (t[:, 6] - t[:, 0]) / 2400
print(f"  workgroup lifetime mean {life.mean():.2f} us; (stamps of different XCDs share no time base)")
ln = lens.repeat_interleave(h).cpu().numpy()[:W]
for lo, hi in ((4, 16), (17, 32), (33, 64)):
    m = (ln >= lo) & (ln <= hi)
    if m.any():
        print(f"  len {lo}-{hi}: n={m.sum()} lifetime {life[m].mean():.2f} us; phase1 {seg[m, 3].mean() / 2400:.2f} phase2 {seg[m, 5].mean() / 2400:.2f} stage {seg[m, 1].mean() / 2400:.2f}")
