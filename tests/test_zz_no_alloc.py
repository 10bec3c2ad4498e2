"""GPU: the allocation-free entry point nast_lp_decode_static_f32 (greedy skip decoding with the library's built-in
workspace) against nast_lp_decode_f32 (caller workspace) through the C-ABI: same alignment positions and step count, on two
calls in a row, the second with a smaller input (the built-in buffer keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import make_param

pytestmark = pytest.mark.gpu


def _decode(lp, memory, lengths, eos, unk, static, sync_mode):
    from nast_amd import ops
    B = memory.shape[0]
    L = lp.args.max_length
    buf = torch.full((B, L + 1), -7, dtype=torch.long, device="cuda")
    with torch.no_grad():
        if sync_mode:
            steps = lp._native_decode(memory, lengths, eos, unk, buf, static=static)
        else:
            with ops.async_decode(None):             # *steps_out = -1: all L steps, no read-back
                steps = lp._native_decode(memory, lengths, eos, unk, buf, static=static)
    torch.cuda.synchronize()
    return steps, buf.cpu().numpy()


@pytest.mark.parametrize("d,h,L", [(64, 4, 12), (256, 4, 32)])
def test_static_workspace_decode_equals_caller_workspace(d, h, L):
    from nast_amd import transformer as tf
    torch.manual_seed(7)
    V = 50
    F = tf.StyleTransformer(make_param(V=V, d=d, h=h, N=1, L=L, learnable=True)).cuda()
    lp = F.length_predictor
    with torch.no_grad():
        for p in lp.parameters():                    # biases and embeddings away from their zero init
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator().manual_seed(d)
    for B, S in ((37, L), (5, max(3, L // 3))):      # the second call is smaller: stale words of the first stay in the buffer
        memory = torch.randn(B, S + 1, d, generator=g).cuda()
        lengths = torch.randint(2, S + 1, (B,), generator=g).cuda()
        lengths[0] = S
        eos, unk = torch.randn(d, generator=g).cuda(), torch.randn(d, generator=g).cuda()
        for sync_mode in (True, False):
            s_ref, a_ref = _decode(lp, memory, lengths, eos, unk, False, sync_mode)
            s_got, a_got = _decode(lp, memory, lengths, eos, unk, True, sync_mode)
            assert s_got == s_ref and s_ref >= 1
            assert np.array_equal(a_got[:, :s_ref + 1], a_ref[:, :s_ref + 1]), (B, sync_mode)
            assert (a_ref[:, 1:s_ref + 1] >= 0).all()


def test_static_workspace_rejects_oversize():
    from nast_amd import transformer as tf, _lib
    F = tf.StyleTransformer(make_param(V=50, d=256, h=4, N=1, L=32, learnable=True)).cuda()
    lp = F.length_predictor
    B = 600                                          # needs ~290 MB: beyond the built-in 64 MiB
    memory = torch.zeros(B, 33, 256, device="cuda")
    lengths = torch.full((B,), 32, dtype=torch.long, device="cuda")
    z = torch.zeros(256, device="cuda")
    buf = torch.zeros(B, 33, dtype=torch.long, device="cuda")
    with pytest.raises(_lib.NastHipError, match="built-in buffer"):
        lp._native_decode(memory, lengths, z, z, buf, static=True)
    lp._native_decode(memory, lengths, z, z, buf, static=False)      # the caller-workspace entry takes it
    torch.cuda.synchronize()
