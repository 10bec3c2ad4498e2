
import os, sys
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from helpers import make_param, synth_batch
import nast_amd.utils as U
import nast_amd.ddp as ddp_mod
from nast_amd.seq2seq import Seq2seq, reserve_streams
from nast_amd.utils import Storage
import faulthandler
faulthandler.dump_traceback_later(100, exit=True)
torch.cuda.set_device(0)
reserve_streams()                                  # before RCCL takes its hardware queues
ddp_mod.FORCE_SINGLE = True                        # reduce even in a group of one rank: RCCL's real launch path
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = dict(V=48, d=32, h=4, N=2, L=8, learnable=True, gen_approx_mode="soft", dis_approx_mode="soft")
U.cuda_init(0, "cuda")
Seq2seq.overlap_min_tokens = 0                     # both streams, as production batches
Seq2seq.pack_min_tokens = 0


def build():
    torch.manual_seed(0); np.random.seed(0)
    m = Seq2seq(make_param(device="cuda", **cfg))
    m.net.model_F.train(); m.net.model_D.train()
    m.net.noiseLayer.forward = lambda w, l, rev_domain: (w, l)
    return m


dp = build()
assert dp.reducer_F.enabled and dp.reducer_F.arena is not None and dp.reducer_D.arena is not None
ddp_mod.ENABLED = False
ref = build()
ddp_mod.ENABLED = True
assert not ref.reducer_F.enabled
rng = np.random.default_rng(3)
for step in range(3):
    sent, slen = synth_batch(rng, 8, cfg["L"], cfg["V"])
    inc = Storage(); inc.data = Storage()
    inc.data.sent, inc.data.sent_length, inc.data.domain = sent.cuda(), slen.cuda(), torch.tensor([0, 0, 1, 1] * 2).cuda()
    fn = "f_step" if step < 2 else "d_step"
    wd = max((p - q).abs().max().item() for p, q in zip(dp.net.parameters(), ref.net.parameters()))
    nd = sum(int((p != q).sum()) for p, q in zip(dp.net.parameters(), ref.net.parameters()))
    print("step", step, "max weight diff", wd, "differing elements", nd, flush=True)
    if os.environ.get("SYNC_WEIGHTS") == "1":
        with torch.no_grad():
            for p, q in zip(dp.net.parameters(), ref.net.parameters()): q.copy_(p)
    a = getattr(dp, fn)(inc, 1.0, use_inference_z=True)
    b = getattr(ref, fn)(inc, 1.0, use_inference_z=True)
    print('losses equal', np.allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=1e-5, atol=1e-6), a[1], b[1], flush=True)
    group = "model_F." if step < 2 else "model_D."
    pairs = [(n, p.grad, q.grad) for (n, p), (_, q) in zip(dp.net.named_parameters(), ref.net.named_parameters())
             if n.startswith(group) and q.grad is not None]
    assert len(pairs) > 20
    scale = max(q.abs().max().item() for _, _, q in pairs)
    worst = max((g - q).abs().max().item() for _, g, q in pairs) / scale
    print('worst grad diff', worst, flush=True)
    # every gradient of the group is a slice of the flat buffer that went through the all-reduce
    lo, hi = dp.reducer_F.arena.data_ptr(), dp.reducer_F.arena.data_ptr() + dp.reducer_F.arena.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in dp.reducer_F.params)
# RCCL: buckets leave during the armed backward (first step of a variant learns the plan), nothing arrives late
assert dp.reducer_F.overlap and dp.reducer_F.stats == {"learnt": 1, "overlapped": 1, "after_backward": 0, "late": 0}, dp.reducer_F.stats
assert dp.reducer_D.stats == {"learnt": 1, "overlapped": 0, "after_backward": 0, "late": 0}, dp.reducer_D.stats
dp.reducer_F._verify_previous(block=True); dp.reducer_D._verify_previous(block=True)
# sos_token never receives a gradient: it is not stepped although its .grad is a (zero) slice of the flat buffer
sos = dp.net.model_F.sos_token
assert sos.grad is not None and len(dp.optimizer_F.state[sos]) == 0
dist.destroy_process_group()
print("ok")
