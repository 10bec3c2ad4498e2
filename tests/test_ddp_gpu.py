"""GPU: data-parallel f_step / d_step == single-process step on the concatenated batch.
Two ranks share the one GPU of the test box and exchange gradients over gloo (RCCL needs one GPU per rank; the
reducer code path -- arming, in-place gradient notifications, bucket launches, write-back -- is the same)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from helpers import make_param, synth_batch
import nast_amd.utils as U
from nast_amd.seq2seq import Seq2seq
from nast_amd.utils import Storage

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import faulthandler
faulthandler.dump_traceback_later(60, exit=True)       # a stuck collective names its line instead of hanging the suite
learnable = sys.argv[2].startswith("learnable")
if sys.argv[2].endswith("2streams"):          # the second-stream overlap that production batches (>= 4096 tokens) use
    Seq2seq.overlap_min_tokens = 0
    Seq2seq.pack_min_tokens = 0
if "multibucket" in sys.argv[2]:             # several buckets in flight while backward still runs (64 KiB each: ~8 buckets for F)
    Seq2seq.overlap_min_tokens = 0
    Seq2seq.pack_min_tokens = 0
    import nast_amd.ddp as _d
    _d.GradReducer.BUCKET_BYTES = 64 << 10
    _d.GradReducer.OVERLAP = True            # (the default under RCCL; gloo launches after backward unless asked)
if sys.argv[2].endswith("2streams"):
    import nast_amd.ddp as _d
    _d.GradReducer.OVERLAP = True
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = dict(V=48, d=32, h=4, N=2, L=8, learnable=learnable, gen_approx_mode="soft", dis_approx_mode="soft")
prez = "prez" in sys.argv[2]
if prez:
    # Steps with use_inference_z=False (`--pre_z_batch`, run.py:61, seq2seq.py:241-247; usable in the reference only with
    # cyc_rec_enable=False: its generate branch returns a 2-tuple there, transformer.py:78, :521-522): the length predictor
    # sits out -- before it ever had a gradient (step 0) and after it has Adam state (step 2).  With weight decay on, a parameter that
    # sits out must neither decay nor move on its momentum nor advance its step count, under data parallelism as alone.
    cfg["L2"] = 1e-2
use_z = [False, True, False, True] if prez else [True, True, True]
n_f = len(use_z) - 1
U.cuda_init(0, "cuda")


def build():
    torch.manual_seed(0); np.random.seed(0)
    m = Seq2seq(make_param(device="cuda", **cfg))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in m.net.named_parameters():
            if n.endswith("embed.weight"):
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.3).cuda())
    m.net.model_F.train(); m.net.model_D.train()
    return m


def incoming(sent, slen, dom):
    inc = Storage(); inc.data = Storage()
    inc.data.sent, inc.data.sent_length, inc.data.domain = sent.cuda(), slen.cuda(), dom.cuda()
    return inc


dp = build()                      # reducers are live: the process group exists
rng = np.random.default_rng(3)
B = 4 * world
for step in range(len(use_z)):
    sent, slen = synth_batch(rng, B, cfg["L"], cfg["V"])
    dom = torch.tensor(([0, 0, 1, 1]) * world)
    sl = slice(4 * rank, 4 * rank + 4)
    # no input noise (identity NoiseLayer) so that the shards see exactly the rows of the full batch
    ident = lambda w, l, rev_domain: (w, l)
    dp.net.noiseLayer.forward = ident
    # approx mode "soft" (seq2seq.py:409): no Gumbel draw, so shards and the full batch see the same soft tokens
    if step < n_f:
        out = dp.f_step(incoming(sent[sl], slen[sl], dom[sl]), 1.0, cyc_rec_enable=use_z[step], use_inference_z=use_z[step])
    else:
        out = dp.d_step(incoming(sent[sl], slen[sl], dom[sl]), 1.0, use_inference_z=use_z[step])
    if rank == 0:
        # reference: same weights before the step, whole batch, no process group involvement
        if step == 0:
            import nast_amd.ddp as ddp_mod
            ddp_mod.ENABLED = False
            ref = build()
            ddp_mod.ENABLED = True
            ref.net.noiseLayer.forward = ident
        # (ref was built from the same seed, so it tracks dp step by step)
        if step < n_f:
            ref.f_step(incoming(sent, slen, dom), 1.0, cyc_rec_enable=use_z[step], use_inference_z=use_z[step])
        else:
            ref.d_step(incoming(sent, slen, dom), 1.0, use_inference_z=use_z[step])
        # compare the (averaged, clipped) gradients the optimiser consumed.  Weights are not compared directly: Adam turns
        # the rounding noise of analytically-zero gradients (key biases: softmax is shift invariant) into +-lr updates.
        group = "model_F." if step < n_f else "model_D."
        pairs = [(n, p.grad, q.grad) for (n, p), (_, q) in zip(dp.net.named_parameters(), ref.net.named_parameters())
                 if n.startswith(group) and q.grad is not None and getattr(q, "_nast_touched", True)]
        assert len(pairs) > 20 and all(g is not None for _, g, _ in pairs)
        # the same parameters were stepped, equally often, on both sides (sos_token never; the length predictor only in
        # use_inference_z=True steps)
        opt_dp, opt_ref = (dp.optimizer_F, ref.optimizer_F) if step < n_f else (dp.optimizer_D, ref.optimizer_D)
        for (n, p), (_, q) in zip(dp.net.named_parameters(), ref.net.named_parameters()):
            if n.startswith(group):
                a, b = opt_dp.state.get(p, {}), opt_ref.state.get(q, {})
                assert float(a.get("step", 0)) == float(b.get("step", 0)), (step, n, a.get("step"), b.get("step"))
        if prez and step < n_f:
            lp_steps = {float(opt_dp.state[p].get("step", 0)) for n, p in dp.net.named_parameters()
                        if n.startswith("model_F.length_predictor.") and p in opt_dp.state and len(opt_dp.state[p])}
            assert lp_steps == ({float(sum(use_z[:step + 1]))} if any(use_z[:step + 1]) else set()), (step, lp_steps)
            assert len(opt_dp.state.get(dp.net.model_F.sos_token, {})) == 0
        scale = max(q.abs().max().item() for _, _, q in pairs)
        worst = max((g - q).abs().max().item() for _, g, q in pairs) / scale
        assert worst < 1e-4, (step, worst)
        big = [(n, p, q) for (n, p), (_, q) in zip(dp.net.named_parameters(), ref.net.named_parameters()) if "weight" in n and p.dim() == 2]
        for n, p, q in big:
            assert (p - q).abs().max().item() <= 2.5e-4 * (q.abs().max().item() + 1e-3), (step, n)
        print("step", step, "max grad diff / max grad", worst, flush=True)
    dist.barrier()
dist.destroy_process_group()
if "multibucket" in sys.argv[2]:
    assert dp.reducer_F.n_buckets >= 2 and dp.reducer_F.stats["overlapped"] >= 1 and dp.reducer_F.stats["late"] == 0, dp.reducer_F.stats
    print("buckets", dp.reducer_F.n_buckets, dp.reducer_F.stats, flush=True)
print("ok", rank)
'''


@pytest.mark.parametrize("mode", ["simple", "learnable", "learnable-2streams", "learnable-multibucket", "learnable-prez"])
def test_two_rank_step_equals_single_process(tmp_path, mode):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = {"simple": "29621", "learnable": "29622", "learnable-2streams": "29623", "learnable-multibucket": "29624", "learnable-prez": "29625"}[mode]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2", OMP_NUM_THREADS="2", NAST_DDP_TIMEOUT_S="40")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=150)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


RCCL_WORKER = r'''
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
reserve_streams()                                  # before RCCL takes its hardware queues (INTEGRATION.md section 1)
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
    # The weights of the two trainers agree to the last bits only: their weight gradients are summed by fp32 atomics in whatever
    # order the hardware takes them.  That is enough to flip a near-tie of the greedy skip decode in the next step (observed: the
    # step-1 cycle loss 10.5463 or 10.5880, both trainers together or one each), so every step starts from identical bits: the
    # update of the previous step is compared here, then the plain trainer takes over the group trainer's weights and Adam state.
    wdiff = max((p - q).abs().max().item() for p, q in zip(dp.net.parameters(), ref.net.parameters()))
    assert wdiff < 5e-6, (step, wdiff)
    with torch.no_grad():
        for p, q in zip(dp.net.parameters(), ref.net.parameters()):
            q.copy_(p)
        for od, orf in ((dp.optimizer_F, ref.optimizer_F), (dp.optimizer_D, ref.optimizer_D)):
            for gd, gr in zip(od.param_groups, orf.param_groups):
                for p, q in zip(gd["params"], gr["params"]):
                    assert set(od.state[p].keys()) == set(orf.state[q].keys()), (step, od.state[p].keys(), orf.state[q].keys())
                    for k, v in od.state[p].items():
                        if torch.is_tensor(v):
                            orf.state[q][k].copy_(v)
                        else:
                            orf.state[q][k] = v
    a = getattr(dp, fn)(inc, 1.0, use_inference_z=True)
    b = getattr(ref, fn)(inc, 1.0, use_inference_z=True)
    assert np.allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=1e-5, atol=1e-6), (step, a, b)
    group = "model_F." if step < 2 else "model_D."
    pairs = [(n, p.grad, q.grad) for (n, p), (_, q) in zip(dp.net.named_parameters(), ref.net.named_parameters())
             if n.startswith(group) and q.grad is not None]
    assert len(pairs) > 20
    scale = max(q.abs().max().item() for _, _, q in pairs)
    worst = max((g - q).abs().max().item() for _, g, q in pairs) / scale
    assert worst < 1e-5, (step, worst)
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
'''


def test_one_rank_rccl_group_step_equals_plain_step(tmp_path):
    """The default exchange (gradients in one flat buffer, buckets all-reduced during the last backward) through a real RCCL process group --
    of one rank, which is what a one-GPU box can hold: f_step, f_step, d_step give the losses and gradients of a trainer that
    takes no part in the group."""
    script = tmp_path / "w.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", NAST_DDP_TIMEOUT_S="40", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=170)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
