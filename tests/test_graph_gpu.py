"""HIP-graph replay of f_step (nast_amd/graph.py) against the eager step: the fixed-width execution the graph records gives
the eager step's numbers, and a replayed step gives the fixed-width eager step's numbers -- same seeds in, same losses and
parameters out."""
import copy

import numpy as np
import pytest
import torch

from helpers import make_param, synth_batch

pytestmark = pytest.mark.gpu

CFG = dict(V=300, d=64, h=4, N=2, L=12)


def _trainer(learnable, **over):
    import nast_amd.utils as U
    from nast_amd.seq2seq import Seq2seq
    U.cuda_init(0, "cuda")
    torch.manual_seed(11)
    tr = Seq2seq(make_param(device="cuda", learnable=learnable, **CFG, **over))
    tr.net.noiseLayer.device_side = "always"      # the twin runs must take their NoiseLayer draws from the same generator
    return tr


def _batch(seed, B=8, full_first=True):
    from nast_amd.utils import Storage
    rng = np.random.default_rng(seed)
    sent, slen = synth_batch(rng, B, CFG["L"], CFG["V"], full_first=full_first)
    if not full_first:
        sent = sent[:, :int(slen.max())]
    inc = Storage(); inc.data = Storage()
    inc.data.sent, inc.data.sent_length = sent.cuda(), slen.cuda()
    inc.data.domain = torch.tensor([i % 2 for i in range(B)]).cuda()
    return inc


def _same_state(a, b, atol, mean_atol=None):
    """Parameters of two trainers.  Gradients are summed with atomics (run-to-run rounding noise ~1e-7 of their size), and
    Adam's update lr * m / (sqrt(v) + eps) of an element whose gradient is itself of the size of eps = 1e-8 turns that noise
    into a sizeable part of one lr = 1e-4 step: with `mean_atol` the bound on single elements is loose (a few lr) and the
    mean difference per tensor -- which any systematic error, e.g. a wrong bias correction, moves by ~1e-5 -- is tight."""
    grads = {k: p.grad for k, p in b.net.named_parameters()}
    for (k, x), (_, y) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        d = (x - y).abs().float()
        # (key biases -- of attention and of the skip logits -- have a mathematically zero gradient: softmax ignores a shift of
        # all scores.  Theirs is rounding noise only, which Adam normalises to full-size steps: no bound for such tensors)
        noise_only = mean_atol is not None and (grads.get(k) is None or grads[k].abs().max().item() <= 1e-6)
        if not noise_only:
            assert d.max().item() <= atol, (k, d.max().item())
            if mean_atol is not None and d.numel():
                assert d.mean().item() <= mean_atol, (k, d.mean().item())


def _close(a, b, tol=2e-5):
    for i, (x, y) in enumerate(zip(a, b)):
        assert abs(x - y) <= tol * max(1.0, abs(y)), (i, a, b)


@pytest.mark.parametrize("learnable", [False, True])
def test_fixed_width_step_equals_eager_step(learnable):
    """ops.fixed_width changes shapes only: with the same draws (no Gumbel noise: approx mode "soft"; NoiseLayer's draws have
    the batch's shape in both runs because the batch is max_length wide) an eager step and a fixed-width step give the same
    losses and the same updated parameters."""
    from nast_amd import ops
    a = _trainer(learnable, gen_approx_mode="soft")
    b = _trainer(learnable, gen_approx_mode="soft")
    _same_state(a, b, 0.0)
    for step in range(2):
        inc = _batch(step)
        torch.manual_seed(100 + step)
        ra = a.f_step(inc, 1.0, True, use_inference_z=True)
        torch.manual_seed(100 + step)
        with ops.packed_rows(False), ops.fixed_width(True, graph_noise=False):
            rb = tuple(0 if x is None else x.item() for x in b._f_step(inc, 1.0, True, use_inference_z=True))
        _close(rb, ra)
        _same_state(a, b, 5e-6)


@pytest.mark.parametrize("learnable,cyc", [(True, True), (False, True), (True, False)])
def test_graph_replay_equals_fixed_width_eager(learnable, cyc):
    """Four steps with graph_steps on (two eager warm-up steps, then capture + replay, then replay) against four eager
    fixed-width steps of a twin trainer under the same seeds: losses per step, parameters and Adam state at the end.  The
    Gumbel noise and NoiseLayer draw through torch's generator in both, so the replays must refresh them per step."""
    from nast_amd import ops
    a = _trainer(learnable)
    b = _trainer(learnable)
    a.graph_steps = True
    outs = []
    for step in range(4):
        inc = _batch(step, full_first=step != 2)        # step 2 (the captured one) arrives narrower than max_length
        torch.manual_seed(200 + step)
        ra = a.f_step(inc, 1.0, cyc, use_inference_z=True)
        torch.manual_seed(200 + step)
        incb = copy.copy(inc); incb.data = copy.copy(inc.data)
        w = inc.data.sent.shape[1]
        full = torch.zeros(inc.data.sent.shape[0], CFG["L"], dtype=torch.int64, device="cuda")
        full[:, :w] = inc.data.sent
        incb.data.sent = full
        with ops.packed_rows(False), ops.fixed_width():
            rb = tuple(0 if x is None else x.item() for x in b._f_step(incb, 1.0, cyc, use_inference_z=True))
        _close(ra, rb)
        outs.append(ra)
    assert a._graphs.stats == {"eager": 2, "captured": 1, "replayed": 2}, a._graphs.stats
    assert outs[2] != outs[3]
    _same_state(a, b, 2.5e-4, mean_atol=5e-7)
    sa, sb = a.optimizer_F.state_dict()["state"], b.optimizer_F.state_dict()["state"]
    assert len(sa) == len(sb) > 0
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 4.0
        assert (sa[k]["exp_avg"] - sb[k]["exp_avg"]).abs().max().item() <= 1e-5 * (1 + sb[k]["exp_avg"].abs().max().item())
    assert float(a.optimizer_F.device_step_counter().item()) == 4.0
    # same input, no reseeding: a replay draws fresh noise
    inc = _batch(9)
    r1 = a.f_step(inc, 1.0, cyc, use_inference_z=True)
    # an eager step in between keeps the device step counter in line
    a.graph_steps = False
    a.f_step(inc, 1.0, cyc, use_inference_z=True)
    a.graph_steps = True
    a.f_step(inc, 1.0, cyc, use_inference_z=True)
    assert float(a.optimizer_F.device_step_counter().item()) == 7.0
    assert all(float(v["step"]) == 7.0 for v in a.optimizer_F.state_dict()["state"].values())
    assert r1[0] > 0


@pytest.mark.parametrize("learnable", [True, False])
def test_d_step_graph_replay_equals_fixed_width_eager(learnable):
    """The discriminator step under graph_steps: two eager fixed-width steps, capture + replay, replay -- against a twin
    trainer's eager fixed-width d_steps under the same seeds (loss per step, parameters of D, Adam state), and the
    fixed-width eager step against the ordinary eager step where no Gumbel noise is drawn (approx mode "soft")."""
    from nast_amd import ops
    a = _trainer(learnable)
    b = _trainer(learnable)
    a.graph_steps = True
    for step in range(4):
        inc = _batch(20 + step, full_first=step != 2)
        torch.manual_seed(300 + step)
        ra = a.d_step(inc, 1.0, use_inference_z=True)
        torch.manual_seed(300 + step)
        incb = copy.copy(inc); incb.data = copy.copy(inc.data)
        w = inc.data.sent.shape[1]
        full = torch.zeros(inc.data.sent.shape[0], CFG["L"], dtype=torch.int64, device="cuda")
        full[:, :w] = inc.data.sent
        incb.data.sent = full
        with ops.packed_rows(False), ops.fixed_width():
            rb = b._d_step(incb, 1.0, use_inference_z=True).item()
        _close((ra,), (rb,))
    assert a._graphs.stats == {"eager": 2, "captured": 1, "replayed": 2}, a._graphs.stats
    _same_state(a, b, 2.5e-4, mean_atol=5e-7)
    assert all(float(v["step"]) == 4.0 for v in a.optimizer_D.state_dict()["state"].values())
    assert float(a.optimizer_D.device_step_counter().item()) == 4.0
    # fixed width == ordinary eager where the draws do not depend on widths
    c = _trainer(learnable, dis_approx_mode="soft")
    d = _trainer(learnable, dis_approx_mode="soft")
    inc = _batch(31)
    torch.manual_seed(5)
    rc = c.d_step(inc, 1.0, use_inference_z=True)
    torch.manual_seed(5)
    with ops.packed_rows(False), ops.fixed_width(True, graph_noise=False):
        rd = d._d_step(inc, 1.0, use_inference_z=True).item()
    _close((rd,), (rc,))
    _same_state(c, d, 5e-6)
