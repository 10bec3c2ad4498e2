"""CPU: host-side logic of the product (no kernel launches): C-ABI exports, state_dict contract,
parameter groups, noise layer, batch preprocessing, and the data-parallel reducer on gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, sub
from helpers import make_param, cfg_of, load_weights, FakeDM


def test_cabi_exports_every_declared_symbol():
    from nast_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nast_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nast_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)        # loads without a GPU; no compute call is made here
    for name in declared:
        assert hasattr(lib, name), name
    for name in declared:                   # argument counts of the binding follow the header
        m = re.search(name + r"\s*\(([^;]*?)\)\s*;", hdr, re.S)
        args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(args) == len(_lib.SIGNATURES[name]), name
    assert _lib.load().nast_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    from nast_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnast_hip.so")
    with pytest.raises(_lib.NastHipError):
        _lib.load()


def test_cpu_tensor_is_rejected_not_emulated():
    from nast_amd import ops, _lib
    with pytest.raises(_lib.NastHipError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3))


@pytest.mark.parametrize("tag", ["simple", "learnable"])
def test_state_dict_contract(tag):
    """Key names + shapes equal the reference's (captured in the golden files), incl. noiseLayer.cnt."""
    from nast_amd.transformer import Network
    g = load_golden(f"g9_{tag}_f_step")
    net = Network(make_param(**cfg_of(g)))
    gold = sub(g, "w0/")
    own = net.state_dict()
    assert set(own) == set(gold), set(own) ^ set(gold)
    for k, v in gold.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    f_names = {n for n, _ in net.named_parameters() if n.startswith("model_F.")}
    d_names = {n for n, _ in net.named_parameters() if n.startswith("model_D.")}
    name_of = {id(p): n for n, p in net.named_parameters()}
    assert {name_of[id(p)] for p in net.get_parameters_by_name("F", silent=True)} == f_names
    assert {name_of[id(p)] for p in net.get_parameters_by_name("D", silent=True)} == d_names


def test_embedding_init_quirk():
    """All embeddings start at exactly 0 except length_predictor.pos_embed (only row 0 zero): SURVEY section 0."""
    from nast_amd.transformer import Network
    net = Network(make_param(V=40, d=32, h=4, N=1, L=8, learnable=True))
    F = net.model_F
    assert F.embed.token_embed.weight.abs().max() == 0 and F.style_embed.weight.abs().max() == 0
    assert F.length_predictor.pos_embed.weight[0].abs().max() == 0
    assert F.length_predictor.pos_embed.weight[1:].abs().max() > 0


def test_noise_layer_matches_reference_draws():
    """With the numpy seed the golden run used, word_blank reproduces the reference's noisy batch exactly."""
    from nast_amd.noise import NoiseLayer
    g = load_golden("g9_simple_f_step")
    nl = NoiseLayer(0.2, 0, 0, 0, 0, keep_go=False, max_length=8, dm=FakeDM(40, 8))
    np.testing.assert_allclose(nl.cnt.numpy(), g["w0/noiseLayer.cnt"], rtol=1e-6)
    np.random.seed(19)
    for s in ("s0/", "s1/"):
        sent, sl = torch.from_numpy(g[s + "in/sent"]), torch.from_numpy(g[s + "in/sent_length"])
        w, l = nl(sent.t(), sl, 1 - torch.from_numpy(g[s + "in/domain"]))
        assert np.array_equal(w.t().numpy(), g[s + "in/noise_tok"]) and np.array_equal(l.numpy(), g[s + "in/noise_len"])


def test_noise_layer_learnable_invariants():
    from nast_amd.noise import NoiseLayer
    L, V, B = 16, 60, 64
    nl = NoiseLayer(0.1, 0.05, 0, 0, 1 / 0.95 - 1, keep_go=False, max_length=L, dm=FakeDM(V, L))
    rng = np.random.default_rng(1)
    from helpers import synth_batch
    np.random.seed(3); torch.manual_seed(3)
    for _ in range(20):
        sent, sl = synth_batch(rng, B, L, V)
        w, l = nl(sent.t(), sl, torch.randint(0, 2, (B,)))
        w, l = w.t().numpy(), l.numpy()
        assert w.shape[1] == l.max() <= L and l.min() >= 1
        for b in range(B):
            assert (w[b, l[b]:] == 0).all()                      # padded with <pad>
            assert w[b, l[b] - 1] == 2 or l[b] == L              # ends with <eos> unless truncated at max_length
            assert (w[b, :l[b]] != 0).all() and (w[b, :l[b]] != 1).all()


def test_noise_stages_known_answers():
    from nast_amd.noise import NoiseLayer
    nl = NoiseLayer(0.5, 0.5, 0, 0, 0.5, keep_go=False, max_length=6, dm=FakeDM(40, 6))
    x = np.array([[10, 11, 12, 2, 0, 0]]).T
    l = np.array([4])
    u = np.array([[0.1, 0.9, 0.1, 0.1, 0.1, 0.1]]).T
    assert nl.blank_apply(x, l, u)[0][:, 0].tolist() == [3, 11, 3, 2, 0, 0]       # eos and padding never blanked
    # add: slots 1 and 3 hold inserted words
    ua = np.array([[0.1, 0.9, 0.1, 0.9, 0.9]]).T
    new = np.arange(30, 36)[:, None]
    out, ln = nl.add_apply(x, l, ua, new)
    assert out[:, 0].tolist() == [10, 31, 11, 33, 12, 2] and ln.tolist() == [6]
    # dropout: drop the middle word
    uk = np.array([[0.9, 0.1, 0.9, 0.9, 0.9, 0.9]]).T
    out, ln = nl.dropout_apply(x, l, uk, np.ones((5, 1)), lambda n: 0)
    assert out[:, 0].tolist() == [10, 12, 2] and ln.tolist() == [3]
    # everything dropped -> one picked word survives
    out, ln = nl.dropout_apply(x, l, np.zeros((6, 1)), np.ones((5, 1)), lambda n: 1)
    assert out[:, 0].tolist() == [11, 2]


def _noise_cases():
    g = load_golden("g11_noise_stages")
    for name in g["names"].tolist():
        c = sub(g, name + "/")
        yield name, c


def test_noise_stages_match_reference_captures():
    """blank_apply / add_apply / dropout_apply against per-stage captures of the reference's word_blank / word_add /
    word_dropout (noise.py:103-135, :76-101, :137-194; oracle/gen_golden.py::case_noise recorded the np.random.rand /
    torch.multinomial / np.random.randint values each stage consumed): same draws in, same ids and lengths out."""
    from nast_amd.noise import NoiseLayer
    n = 0
    for name, c in _noise_cases():
        L, V = int(c["cfg/L"]), int(c["cfg/V"])
        nl = NoiseLayer(float(c["cfg/blank"]), float(c["cfg/drop"]), 0, float(c["cfg/rep"]), float(c["cfg/add"]), keep_go=False,
                        max_length=L, dm=FakeDM(V, L))
        x, l = c["in/x"], c["in/l"].astype(np.int64)
        x1, l1 = nl.blank_apply(x, l, c["blank/u"])
        assert np.array_equal(x1, c["blank/x"]) and np.array_equal(l1, c["blank/l"]), name
        x2, l2 = nl.add_apply(c["blank/x"], c["blank/l"].astype(np.int64), c["add/u"], c["add/new_words"])
        assert np.array_equal(l2, c["add/l"]), name
        assert np.array_equal(x2, c["add/x"]), name
        picks = iter(c["drop/picks"].reshape(-1, 3).tolist())

        def pick(nwords):
            lo, hi, r = next(picks)
            assert (lo, hi) == (0, nwords)            # the reference asked for the same range
            return r

        x3, l3 = nl.dropout_apply(c["add/x"], c["add/l"].astype(np.int64), c["drop/u_keep"], c["drop/u_rep"], pick)
        assert np.array_equal(l3, c["drop/l"]), name
        assert np.array_equal(x3, c["drop/x"]), name
        assert next(picks, None) is None
        n += 1
    assert n == 12


def test_reference_import_names_resolve_through_compat():
    """the modules `main.py:9` / `seq2seq.py:14,16` import, under the reference's own names (compat/)."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from seq2seq import Seq2seq\n"
            "from transformer import Network, StyleTransformer, Discriminator, LengthPredictor, scaled_attention\n"
            "from utils import Storage, cuda, BaseModel, LongTensor, gumbel_max, generateMask, cuda_init, debug, try_cache\n"
            "from utils.gumbel import gumbel_softmax, gumbel_max_with_mask, straight_max, onehot\n"
            "import nast_amd.seq2seq, nast_amd.transformer\n"
            "assert issubclass(Seq2seq, nast_amd.seq2seq.Seq2seq) and Network is nast_amd.transformer.Network\n"
            "print('ok')\n") % (os.path.join(ROOT, "compat"), ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_preprocess_batch_and_temperature():
    from nast_amd.seq2seq import Seq2seq
    s = Seq2seq.__new__(Seq2seq)
    d0 = {"sent": np.array([[1, 5, 6, 2], [1, 7, 2, 0]]), "sent_length": [4, 3]}
    d1 = {"sent": np.array([[1, 8, 9, 10, 2]]), "sent_length": [5]}
    from nast_amd.utils import cuda_init
    cuda_init(0, "cpu")
    inc = s._preprocess_batch(d0, d1)
    assert inc.data.sent.tolist() == [[5, 6, 2, 0], [7, 2, 0, 0], [8, 9, 10, 2]]
    assert inc.data.sent_length.tolist() == [3, 2, 4] and inc.data.domain.tolist() == [0, 0, 1]
    assert s.calc_temperature([(1, 0)], 7) == 1
    assert abs(s.calc_temperature([(1.0, 0), (0.5, 10)], 5) - 0.75) < 1e-12


def test_utils_surface():
    import nast_amd.utils as U
    g = load_golden("g10_misc")
    U.cuda_init(0, "cpu")
    np.testing.assert_array_equal(U.generateMask(5, g["generateMask/len"], type=float).numpy(), g["generateMask/out"])
    st = U.Storage(a=1)
    st.b = U.Storage(c=2)
    assert st.a == 1 and st.b.c == 2 and "a" in st
    with pytest.raises(AttributeError):
        st.zzz
    assert U.onehot(torch.tensor([[1, 0]]), 3).tolist() == [[[0, 1, 0], [1, 0, 0]]]
    for name in ("gumbel_max", "gumbel_max_with_mask", "gumbel_softmax", "straight_max", "onehot", "cdist",
                 "recenter_gradient", "RebarGradient", "generateMask", "Storage", "BaseNetwork", "cuda_init"):
        assert hasattr(U, name)


DDP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from nast_amd.ddp import GradReducer
from nast_amd import ops
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
mk = lambda: torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 4))
net, ref = mk(), mk()
ref.load_state_dict(net.state_dict())
unused = torch.nn.Parameter(torch.zeros(3))                      # like sos_token: never gets a gradient
w2 = torch.nn.Parameter(torch.rand(4) + 0.5)                     # gradient written IN PLACE, twice per backward
w2_ref = torch.nn.Parameter(w2.detach().clone())
late = torch.nn.Parameter(torch.rand(4) + 0.5)                   # first takes part in the last step (like the length predictor)
late_ref = torch.nn.Parameter(late.detach().clone())
params = list(net.parameters()) + [unused, w2, late]
red = GradReducer(params, bucket_bytes=4096)
assert red.arena is not None and red.n_buckets > 1
assert all(p.grad is not None and p.grad.data_ptr() == s.data_ptr() for p, s in zip(red.params, red._slices))
# layout: reverse registration order, buckets are contiguous ranges of the flat buffer
assert red.bucket_of[len(params) - 1] == 0 and red.ranges[0][0] == 0 and red.ranges[-1][1] == red.arena.numel()


class InPlaceScale(torch.autograd.Function):                    # the pattern of the HIP backward kernels
    @staticmethod
    def forward(ctx, x, w):
        ctx.w = w
        ctx.save_for_backward(x)
        return x * w.detach()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        w = ctx.w
        (buf,), _ = ops.grad_buffers([w])       # the parameter's .grad, or a side buffer when its bucket has left
        buf += (g * x).sum(0)
        ops.grads_written([w])
        return g * w.detach(), None


def fwd(m, X, w, inplace):
    y = m(X)
    if inplace:
        return InPlaceScale.apply(InPlaceScale.apply(y, w).tanh(), w)
    return (y * w).tanh() * w


def check(pairs, what):
    for p, q in pairs:
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6), (what, (p.grad - q.grad).abs().max())


pairs = list(zip(list(net.parameters()) + [w2], list(ref.parameters()) + [w2_ref]))
g = torch.Generator().manual_seed(5)
for step in range(4):
    X = torch.randn(2 * world * 4, 16, generator=g); Y = torch.randn(2 * world * 4, 4, generator=g)
    Xa, Ya = X[: world * 4], Y[: world * 4]; Xb, Yb = X[world * 4:], Y[world * 4:]
    sl = slice(rank * 4, rank * 4 + 4)
    red.zero_grads()
    # two backward passes accumulating into the same grads; only the second is reduced (f_step pattern)
    ((fwd(net, Xa[sl], w2, True) - Ya[sl]) ** 2).sum().div(4).backward()
    red.arm("two-pass")
    ((fwd(net, Xb[sl], w2, True) - Yb[sl]) ** 2).sum().div(4).backward()
    red.finish()
    for p in list(ref.parameters()) + [w2_ref]: p.grad = None
    (((fwd(ref, Xa, w2_ref, False) - Ya) ** 2).sum() / (4 * world) + ((fwd(ref, Xb, w2_ref, False) - Yb) ** 2).sum() / (4 * world)).backward()
    check(pairs, step)
    plan = red.plans["two-pass"]
    iu, il = red.index[id(unused)], red.index[id(late)]
    assert iu not in plan.part and il not in plan.part and float(unused.grad.abs().sum()) == 0
    assert plan.counts[red.index[id(w2)]] >= 2
assert red.stats["learnt"] == 1 and red.stats["overlapped" if red.overlap else "after_backward"] == 3, red.stats
# a step whose graph has MORE in-place contributions than the learnt plan, and a parameter that takes part for the first
# time, must still be exact (side buffers, reduced on their own)
red.zero_grads()
X = torch.randn(world * 4, 16, generator=g); Y = torch.randn(world * 4, 4, generator=g)
sl = slice(rank * 4, rank * 4 + 4)
red.arm("two-pass")
((InPlaceScale.apply(InPlaceScale.apply(fwd(net, X[sl], w2, True), w2), late) - Y[sl]) ** 2).sum().div(4).backward()
red.finish()
for p in list(ref.parameters()) + [w2_ref, late_ref]: p.grad = None
(((fwd(ref, X, w2_ref, False) * w2_ref * late_ref - Y) ** 2).sum() / (4 * world)).backward()
check(pairs + [(late, late_ref)], "changed graph")
assert red.stats["late"] == 1 and red.index[id(late)] in red.plans["two-pass"].part, red.stats
# ... and the re-learnt plan covers that graph from now on
red.zero_grads()
red.arm("two-pass")
((InPlaceScale.apply(InPlaceScale.apply(fwd(net, X[sl], w2, True), w2), late) - Y[sl]) ** 2).sum().div(4).backward()
red.finish()
check(pairs + [(late, late_ref)], "re-learnt graph")
assert red.stats["late"] == 1, red.stats
red._verify_previous(block=True)
dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("overlap,world", [("0", 2), ("1", 2), ("0", 4), ("1", 4)])
def test_grad_reducer_gloo_world2(tmp_path, overlap, world):
    """N-rank averaged gradients == single-process gradients on the concatenated batch, with the buckets
    launched after backward (the default) and from the gradient hooks during backward (NAST_DDP_OVERLAP=1)."""
    script = tmp_path / "w.py"
    script.write_text(DDP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29610 + int(overlap) + 2 * world), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1", NAST_DDP_OVERLAP=overlap)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_packed_seq_index_maps():
    """ops.PackedSeq: packed <-> padded row maps, the multiple-of-64 padding, tail() (positions >= 1) and host totals."""
    import torch
    from nast_amd import ops
    lengths = torch.tensor([3, 1, 5, 2])
    n, d = 5, 4
    ps = ops.PackedSeq(ops.set_total(lengths.clone(), 11), n)
    assert ps.total == 11 and ps.rows == 64 and ps.idx.tolist() == [0, 1, 2, 5, 10, 11, 12, 13, 14, 15, 16]
    assert ps.off32.tolist() == [0, 3, 4, 9] and ps.len32.tolist() == [3, 1, 5, 2]
    assert ps.idx_p.shape[0] == 64 and ps.idx_p[11:].abs().sum() == 0           # padding rows copy row 0
    x = torch.arange(4 * n * d, dtype=torch.float32).view(4, n, d).requires_grad_(True)
    xp = ops.pack_rows(x, ps)
    assert xp.shape == (64, d) and torch.equal(xp[:11], x.view(-1, d)[ps.idx])
    back = ops.unpack_rows(xp, ps)
    live = (torch.arange(n)[None] < lengths[:, None]).unsqueeze(-1)
    assert torch.equal(back, x.detach() * live)
    back.sum().backward()                                                        # only real rows receive gradient
    assert torch.equal(x.grad, live.expand_as(x).float())
    rows, sub = ps.tail()
    assert sub.total == 7 and sub.width == n - 1 and sub.len32.tolist() == [2, 0, 4, 1]
    assert rows[:7].tolist() == [1, 2, 5, 6, 7, 8, 10] and sub.idx.tolist() == [0, 1, 8, 9, 10, 11, 12]
    # without a host total the constructor reads it from the device tensor
    assert ops.PackedSeq(lengths, n).total == 11
    # PackedSoft answers the shape queries of a [B, T, V] tensor
    soft = ops.PackedSoft(torch.zeros(ps.rows, 7), ps)
    assert soft.dim() == 3 and soft.size(0) == 4 and soft.size(1) == n and soft.shape == (4, n, 7)
    assert soft.column(3).shape == (4, n)


def test_oracle_sample_skip_follows_softmax():
    """oracle.sample_skip (the restatement of torch.multinomial(logits.softmax(-1), 1), transformer.py:560-561, with the
    uniforms as an input): frequencies over many uniforms match softmax(logits); masked entries (-1e9) are never drawn."""
    from oracle import nast_oracle as O
    logits = torch.tensor([[-1e9, 0.3, -0.7, 1.1], [0.0, 0.0, 0.0, 0.0], [2.0, -1e9, -1e9, -1e9]])
    n = 200000
    u = torch.rand(n, generator=torch.Generator().manual_seed(0))
    for row in logits:
        d = O.sample_skip(row.expand(n, 4), u)
        freq = torch.bincount(d, minlength=4).double() / n
        p = torch.softmax(row.double(), -1)
        assert (freq - p).abs().max() < 5e-3, (freq, p)
        assert freq[p == 0].sum() == 0
    assert O.sample_skip(logits[:1], torch.tensor([0.999999999])).item() == 3


def test_restore_from_checkpoint_file(tmp_path):
    """args.restore (utils/model_helper.py:53-88): a checkpoint file written by save_checkpoint restores counters, weights
    and optimizer state into a fresh trainer; a manager name / URL raises instead of being ignored."""
    from helpers import make_param
    from nast_amd.seq2seq import Seq2seq
    kw = dict(V=40, d=16, h=2, N=1, L=6, learnable=True)
    a = Seq2seq(make_param(**kw))
    a.now_batch, a.now_epoch = 7, 2
    with torch.no_grad():
        for p in a.net.parameters():
            p.add_(0.01)
    p0 = next(iter(a.net.model_F.parameters()))
    a.optimizer_F.state[p0] = {"step": torch.tensor(3.0), "exp_avg": torch.full_like(p0, 0.5), "exp_avg_sq": torch.full_like(p0, 0.25)}
    path = str(tmp_path / "ck.model")
    a.save_checkpoint(path)
    pb = make_param(**kw)
    pb.args.restore = path
    pb.args.lr_F = 3e-4                                         # the new run's hyper-parameter must survive the restore
    b = Seq2seq(pb)
    assert (b.now_batch, b.now_epoch) == (7, 2)
    for (k, x), (_, y) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert torch.equal(x, y), k
    q0 = next(iter(b.net.model_F.parameters()))
    st = b.optimizer_F.state[q0]
    assert float(st["step"]) == 3.0 and torch.equal(st["exp_avg"], torch.full_like(q0, 0.5))
    assert b.optimizer_F.param_groups[0]["lr"] == 3e-4
    pc = make_param(**kw)
    pc.args.restore = "last"
    with pytest.raises(NotImplementedError):
        Seq2seq(pc)


def test_fused_adam_participation_host_path():
    """The same bookkeeping on the host path (CPU parameters take torch's own Adam step): see
    test_kernels_gpu.py::test_fused_adam_skips_parameters_that_sat_out for the device kernels."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_tk", os.path.join(ROOT, "tests", "test_kernels_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod._participation_case("cpu")
