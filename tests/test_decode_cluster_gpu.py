"""GPU: the cluster skip decode (csrc/decode_cluster.hip: one persistent launch, 8 clusters x 32 workgroups, cluster barriers
between the operations) against the step-by-step path (decode.hip; NAST_DECODE_CLUSTER=0) through nast_lp_decode_f32: the same
alignment positions and step count, both early-exit modes, greedy and sampled, batch sizes that do not fill the clusters'
32-row tiles, 1 and 3 decoder layers."""
import numpy as np
import pytest
import torch

from helpers import make_param

pytestmark = pytest.mark.gpu


def _decode(lp, memory, lengths, eos, unk, sync_mode, sample_u=None):
    from nast_amd import ops
    B = memory.shape[0]
    L = lp.args.max_length
    buf = torch.full((B, L + 1), -7, dtype=torch.long, device="cuda")
    with torch.no_grad():
        if sync_mode:
            steps = lp._native_decode(memory, lengths, eos, unk, buf, sample_u=sample_u)
        else:
            with ops.async_decode(None):
                steps = lp._native_decode(memory, lengths, eos, unk, buf, sample_u=sample_u)
    torch.cuda.synchronize()
    return steps, buf.cpu().numpy()


def _status():
    import ctypes as C
    from nast_amd import _lib
    st = C.c_int(-1)
    assert _lib.load().nast_lp_decode_cluster_status(C.byref(st)) == 0
    return st.value


@pytest.mark.parametrize("L,S,Bs", [(32, 32, (37, 5, 64, 512, 700)), (32, 9, (16, 300)), (12, 12, (1, 130)), (35, 20, (1100, 257))])
def test_cluster_decode_equals_stepwise(L, S, Bs, monkeypatch):
    from nast_amd import transformer as tf
    monkeypatch.setenv("NAST_DECODE_FUSED", "0")
    torch.manual_seed(3)
    d = 256
    F = tf.StyleTransformer(make_param(V=50, d=d, h=4, N=1, L=L, learnable=True)).cuda()
    lp = F.length_predictor
    with torch.no_grad():
        for p in lp.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator().manual_seed(L * 100 + S)
    for B in Bs:
        memory = torch.randn(B, S + 1, d, generator=g).cuda()
        lengths = torch.randint(2, S + 1, (B,), generator=g).cuda()
        lengths[0] = S
        eos, unk = torch.randn(d, generator=g).cuda(), torch.randn(d, generator=g).cuda()
        u = torch.rand(L, B, generator=g).cuda()
        for sync_mode in (True, False):
            for su in (None, u):
                monkeypatch.setenv("NAST_DECODE_CLUSTER", "0")
                s_ref, a_ref = _decode(lp, memory, lengths, eos, unk, sync_mode, su)
                monkeypatch.setenv("NAST_DECODE_CLUSTER", "1")
                s_got, a_got = _decode(lp, memory, lengths, eos, unk, sync_mode, su)
                assert _status() == 0
                assert s_got == s_ref and s_ref >= 1, (B, sync_mode, s_got, s_ref)
                same = a_got[:, :s_ref + 1] == a_ref[:, :s_ref + 1]
                assert same.all(), (B, sync_mode, su is not None, float(same.mean()), np.argwhere(~same)[:5],
                                    a_got[~same.all(1)][:2], a_ref[~same.all(1)][:2])
                assert (a_ref[:, 1:s_ref + 1] >= 0).all()


def test_cluster_decode_beside_a_loaded_stream(monkeypatch):
    """The hand-offs of the cluster kernel (sc1 stores -> drain -> counter -> sc1 loads) under UNEVEN load, which is where a
    stale or early read would show (MI355X_MICROARCH.md, visibility: idle chips hide such failures): the decode runs on a side
    stream while the main stream keeps every CU busy with large GEMMs and row kernels, several times in a row on one
    workspace, and must reproduce the positions the step-by-step path gives on an idle GPU."""
    from nast_amd import ops, transformer as tf
    monkeypatch.setenv("NAST_DECODE_FUSED", "0")
    torch.manual_seed(5)
    d, L, S, B = 256, 32, 32, 512
    F = tf.StyleTransformer(make_param(V=50, d=d, h=4, N=1, L=L, learnable=True)).cuda()
    lp = F.length_predictor
    with torch.no_grad():
        for p in lp.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator().manual_seed(77)
    memory = torch.randn(B, S + 1, d, generator=g).cuda()
    lengths = torch.randint(2, S + 1, (B,), generator=g).cuda()
    lengths[0] = S
    eos, unk = torch.randn(d, generator=g).cuda(), torch.randn(d, generator=g).cuda()
    monkeypatch.setenv("NAST_DECODE_CLUSTER", "0")
    _, a_ref = _decode(lp, memory, lengths, eos, unk, False)
    monkeypatch.setenv("NAST_DECODE_CLUSTER", "1")
    x = torch.randn(16384, 1024, device="cuda")
    W = torch.randn(1024, 1024, device="cuda") / 32
    side = torch.cuda.Stream()
    bufs = [torch.full((B, L + 1), -7, dtype=torch.long, device="cuda") for _ in range(6)]
    torch.cuda.synchronize()
    with torch.no_grad():
        for i, buf in enumerate(bufs):
            with torch.cuda.stream(side), ops.async_decode(None):
                lp._native_decode(memory, lengths, eos, unk, buf)
            for _ in range(3 + 2 * (i % 3)):                      # a different amount of competing work each time
                y = ops.linear(x, W, None, None, True)
                y = torch.softmax(y, -1)
    torch.cuda.synchronize()
    assert _status() == 0
    for buf in bufs:
        assert (buf.cpu().numpy() == a_ref).all()
