"""GPU: edge batches through the training steps -- one sentence, sentences that are a lone <eos>, every sentence at max_length --
run and give the same losses in the packed-row execution and the padded one (two independent code paths over the same
kernels), learnable and simple alignment."""
import numpy as np
import pytest
import torch

from helpers import make_param

pytestmark = pytest.mark.gpu

CFG = dict(V=60, d=64, h=4, N=2, L=10)


def _trainer(learnable, pack):
    import nast_amd.utils as U
    from nast_amd.seq2seq import Seq2seq
    U.cuda_init(0, "cuda")
    torch.manual_seed(4)
    tr = Seq2seq(make_param(device="cuda", learnable=learnable, gen_approx_mode="soft", dis_approx_mode="soft", **CFG))
    tr.net.noiseLayer.forward = lambda w, l, rev_domain: (w, l)          # identity: both trainers see the same input
    tr.pack_min_tokens = 0 if pack else 1 << 60
    tr.overlap_min_tokens = 0
    return tr


def _inc(lens, seed=0):
    from nast_amd.utils import Storage
    rng = np.random.default_rng(seed)
    B, W = len(lens), max(lens)
    sent = np.zeros((B, W), dtype=np.int64)
    for b, n in enumerate(lens):
        sent[b, :n - 1] = rng.integers(4, CFG["V"], size=n - 1)
        sent[b, n - 1] = 2
    inc = Storage(); inc.data = Storage()
    inc.data.sent = torch.from_numpy(sent).cuda()
    inc.data.sent_length = torch.tensor(lens, dtype=torch.long).cuda()
    inc.data.domain = torch.tensor([i % 2 for i in range(B)]).cuda()
    return inc


@pytest.mark.parametrize("learnable", [True, False])
@pytest.mark.parametrize("lens", [[5], [1], [1, 1, 1], [10, 10, 10, 10], [1, 10, 2, 7, 1]])
def test_edge_batches_packed_equals_padded(learnable, lens):
    a, b = _trainer(learnable, True), _trainer(learnable, False)
    for step, fn in enumerate(("f_step", "f_step", "d_step")):
        inc = _inc(lens, seed=step)
        ra = getattr(a, fn)(inc, 1.0, use_inference_z=True)
        rb = getattr(b, fn)(inc, 1.0, use_inference_z=True)
        ra, rb = np.atleast_1d(np.asarray(ra, dtype=float)), np.atleast_1d(np.asarray(rb, dtype=float))
        assert np.isfinite(ra).all() and np.isfinite(rb).all(), (fn, ra, rb)
        assert np.allclose(ra, rb, rtol=2e-4, atol=1e-5), (fn, step, ra, rb)
    for (k, x), (_, y) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert torch.isfinite(x).all(), k
