"""GPU: device-side NoiseLayer (nast_noise_i64 through ops.noise_tokens) against the host stages of nast_amd/noise.py
(`blank_apply` / `add_apply` / `dropout_apply`, which restate the reference's per-sentence loops, noise.py:76-194, and are
pinned on CPU by tests/test_host_cpu.py) on the SAME uniform draws.  Integer work: bit-exact."""
import numpy as np
import pytest
import torch

from helpers import FakeDM, synth_batch

pytestmark = pytest.mark.gpu


def host_chain(nl, x, l, d):
    """The host path of NoiseLayer.forward on given draws (stage order of noise.py:60-74)."""
    if "u_blank" in d:
        x, l = nl.blank_apply(x, l, d["u_blank"])
    if "u_add" in d:
        x, l = nl.add_apply(x, l, d["u_add"], d["new_words"])
    x0, l0 = x, l
    picks = iter([])
    if "u_keep" in d:
        # pick(n) is called for the sentences whose words were all dropped, in batch order: floor(u * n) like the kernel
        M = nl.max_length
        keep = d["u_keep"] >= nl.dropout_prob
        n = np.where(x0[l0 - 1, np.arange(x0.shape[1])] == nl.eos_index, l0 - 1, l0)
        km = keep[:x0.shape[0]] & (np.arange(x0.shape[0])[:, None] < n[None, :])
        empty = np.nonzero(km.sum(0) == 0)[0]
        picks = iter([int(d["u_pick"][i] * n[i]) for i in empty])
        x, l = nl.dropout_apply(x0, l0, d["u_keep"], d["u_rep"], lambda n_: next(picks))
    return x, l


@pytest.mark.parametrize("B,L,S,probs", [
    (1, 6, 6, (0.5, 0.5, 0.5, 0.0)),
    (130, 16, 16, (0.1, 0.05, 1 / 0.95 - 1, 0.0)),       # run.py:90-95 settings, ragged over three wavefronts
    (64, 32, 32, (0.1, 0.05, 1 / 0.95 - 1, 0.0)),
    (97, 12, 9, (0.6, 0.9, 0.7, 0.3)),                    # input narrower than max_length; most words dropped; repeats
    (33, 8, 8, (0.0, 0.3, 0.0, 0.0)),                     # dropout stage alone
    (33, 8, 8, (0.3, 0.2, 0.0, 0.5)),                     # no add stage
])
def test_device_noise_equals_host_stages(B, L, S, probs):
    from nast_amd import ops
    from nast_amd.noise import NoiseLayer
    blank, drop, add, rep = probs
    V = 50
    nl = NoiseLayer(blank, drop, 0, rep, add, keep_go=False, max_length=L, dm=FakeDM(V, L))
    rng = np.random.default_rng(B * 1000 + L)
    for trial in range(6):
        sent, sl = synth_batch(rng, B, S, V, full_first=trial % 2 == 0)
        if trial == 3:
            sl[:] = 2                      # one word + <eos>
            sent[:, 1] = 2
            sent[:, 2:] = 0
        x, l = sent.t().contiguous().numpy(), sl.numpy()
        d = {}
        if blank:
            d["u_blank"] = rng.random((S, B))
        if add:
            d["u_add"] = rng.random((L - 1, B))
            d["new_words"] = rng.integers(4, V, size=(L, B))
        d["u_keep"], d["u_rep"], d["u_pick"] = rng.random((L, B)), rng.random((L - 1, B)), rng.random(B)
        want_x, want_l = host_chain(nl, x, l, d)
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
        out, out_len = ops.noise_tokens(torch.from_numpy(x).cuda(), sl.cuda(), L, blank_prob=blank, add_prob=add, dropout_prob=drop,
                                        repeat_prob=rep, pad=nl.pad_index, unk=nl.blank_index, eos=nl.eos_index, **dev)
        got_l = out_len.cpu().numpy()
        assert np.array_equal(got_l, want_l), (trial, np.nonzero(got_l != want_l)[0][:5])
        got = out.cpu().numpy()
        assert np.array_equal(got[:want_x.shape[0]], want_x), trial
        assert (got[want_x.shape[0]:] == nl.pad_index).all()


def test_device_noise_matches_reference_captures():
    """nast_noise_i64 against the per-stage captures of the REFERENCE's NoiseLayer (tests/golden/g11_noise_stages.npz,
    written by oracle/gen_golden.py::case_noise from noise.py:76-194 with its rand / multinomial / randint values
    recorded): each stage alone on the reference's stage input, and the three stages chained in one launch.  Bit-exact."""
    from conftest import load_golden, sub
    from nast_amd import ops
    g = load_golden("g11_noise_stages")
    C = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for name in g["names"].tolist():
        c = sub(g, name + "/")
        L, B = int(c["cfg/L"]), int(c["cfg/B"])
        probs = dict(blank_prob=float(c["cfg/blank"]), add_prob=float(c["cfg/add"]), dropout_prob=float(c["cfg/drop"]),
                     repeat_prob=float(c["cfg/rep"]))
        ids = dict(pad=0, unk=3, eos=2)
        # the reference picks a survivor with np.random.randint(0, n) for the sentences that lost every word (noise.py:179), in
        # batch order; the kernel takes floor(u * n) from a per-sentence uniform: u = (r + 0.5) / n reproduces pick r
        xa, la = c["add/x"], c["add/l"].astype(np.int64)
        nwords = np.where(xa[la - 1, np.arange(B)] == 2, la - 1, la)
        keep = c["drop/u_keep"] >= probs["dropout_prob"]
        km = keep[:xa.shape[0]] & (np.arange(xa.shape[0])[:, None] < nwords[None, :])
        empty = np.nonzero(km.sum(0) == 0)[0]
        picks = c["drop/picks"].reshape(-1, 3)
        assert len(empty) == len(picks) and np.array_equal(picks[:, 1], nwords[empty])
        u_pick = np.full(B, 0.5)
        u_pick[empty] = (picks[:, 2] + 0.5) / picks[:, 1]

        def run(x, l, **kw):
            out, out_len = ops.noise_tokens(C(x), C(l.astype(np.int64)), L, **ids, **kw)
            return out.cpu().numpy(), out_len.cpu().numpy()

        def same(got, want_x, want_l, what):
            gx, gl = got
            assert np.array_equal(gl, want_l), (name, what)
            w = want_x.shape[0]
            # before the dropout stage rows past a sentence's new length are not defined by the kernel's contract
            live = np.arange(w)[:, None] < want_l[None, :]
            assert np.array_equal(np.where(live, gx[:w], 0), np.where(live, want_x, 0)), (name, what)

        same(run(c["in/x"], c["in/l"], u_blank=C(c["blank/u"]), blank_prob=probs["blank_prob"]), c["blank/x"], c["blank/l"], "blank")
        same(run(c["blank/x"], c["blank/l"], u_add=C(c["add/u"]), new_words=C(c["add/new_words"]), add_prob=probs["add_prob"]),
             c["add/x"], c["add/l"], "add")
        drop = dict(u_keep=C(c["drop/u_keep"]), u_rep=C(c["drop/u_rep"]), u_pick=C(u_pick), dropout_prob=probs["dropout_prob"],
                    repeat_prob=probs["repeat_prob"])
        gx, gl = run(xa, la, **drop)
        assert np.array_equal(gl, c["drop/l"]), (name, "dropout")
        w = c["drop/x"].shape[0]
        assert np.array_equal(gx[:w], c["drop/x"]) and (gx[w:] == 0).all(), (name, "dropout")
        # the whole chain in one launch, as NoiseLayer.forward issues it
        gx, gl = run(c["in/x"], c["in/l"], u_blank=C(c["blank/u"]), blank_prob=probs["blank_prob"], u_add=C(c["add/u"]),
                     new_words=C(c["add/new_words"]), add_prob=probs["add_prob"], **drop)
        assert np.array_equal(gl, c["drop/l"]), (name, "chain")
        assert np.array_equal(gx[:w], c["drop/x"]) and (gx[w:] == 0).all(), (name, "chain")


def test_noise_layer_device_path_invariants():
    """NoiseLayer.forward on CUDA tensors takes the device path; outputs keep the reference's invariants."""
    from nast_amd.noise import NoiseLayer
    L, V, B = 16, 60, 200
    nl = NoiseLayer(0.1, 0.05, 0, 0, 1 / 0.95 - 1, keep_go=False, max_length=L, dm=FakeDM(V, L)).cuda()
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    changed = 0
    for _ in range(5):
        sent, sl = synth_batch(rng, B, L, V)
        w, l = nl(sent.t().cuda(), sl.cuda(), torch.randint(0, 2, (B,)).cuda())
        assert w.is_cuda and l.nast_total == int(l.sum())
        w, l = w.t().cpu().numpy(), l.cpu().numpy()
        assert w.shape[1] == l.max() <= L and l.min() >= 1
        for b in range(B):
            assert (w[b, l[b]:] == 0).all()
            assert w[b, l[b] - 1] == 2 or l[b] == L
            assert (w[b, :l[b]] != 0).all() and (w[b, :l[b]] != 1).all()
        changed += int((l != sl.numpy()).sum())
    assert changed > 0          # the noise does something


def test_noise_rejects_oversize():
    from nast_amd import ops, _lib
    x = torch.zeros(4, 2, dtype=torch.long, device="cuda")
    with pytest.raises(_lib.NastHipError):
        ops.noise_tokens(x, torch.ones(2, dtype=torch.long, device="cuda"), 400, pad=0, unk=3, eos=2)
