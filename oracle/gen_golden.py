#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE, runs only in the build container).

Imports the *unmodified* reference (thu-coai/NAST at /root/reference) on CPU, with in-process
stubs for the packages it imports at module top but never uses on the numeric path
(cotk, tensorboardX, nltk), runs the hot-path functions on tiny seeded inputs and writes
inputs + outputs (+ gradients) as .npz fixtures under tests/golden/.

The fixtures are DATA ONLY (weights, inputs, expected outputs). No reference source is copied.
The GPU box never has /root/reference; it only sees the committed .npz files.

Usage:  python oracle/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


# ----------------------------------------------------------------------------------------------
# reference import (stubs for absent, non-numeric dependencies)
# ----------------------------------------------------------------------------------------------
class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("cotk")
    _stub("cotk.file_utils", load_file_from_url=lambda *a, **k: None)
    _stub("cotk.metric", BleuCorpusMetric=_Anything, MetricChain=_Anything,
          LanguageGenerationRecorder=_Anything, MetricBase=_Anything)
    _stub("cotk.dataloader", LanguageProcessing=_Anything)
    _stub("tensorboardX", SummaryWriter=_Anything)
    _stub("nltk")
    _stub("nltk.tokenize", WordPunctTokenizer=_Anything)
    sys.path[:0] = [REF, os.path.join(REF, "styletransformer")]
    import utils as ref_utils
    ref_utils.cuda_init(0, "cpu")
    import transformer as ref_tf
    import seq2seq as ref_s2s
    return ref_utils, ref_tf, ref_s2s


# ----------------------------------------------------------------------------------------------
# fake data manager + args
# ----------------------------------------------------------------------------------------------
class FakeDM:
    pad_id, go_id, eos_id, unk_id = 0, 1, 2, 3

    def __init__(self, vocab, rng, max_length):
        self.frequent_vocab_size = vocab
        self._sents = {}
        for dom in (0, 1):
            sents = []
            for _ in range(50):
                n = int(rng.integers(2, max_length))
                sents.append([1] + [int(x) for x in rng.integers(4, vocab, size=n)] + [2])
            self._sents[dom] = sents

    def get_all_batch(self, key):
        return {"sent": self._sents[int(key[-1])]}


def make_param(ref_utils, *, V, d, h, N, L, learnable, disc="Cond", seed=0, **over):
    S = ref_utils.Storage
    a = S()
    a.num_styles, a.num_layers, a.d_model, a.max_length, a.h, a.dropout = 2, N, d, L, h, 0
    a.learned_pos_embed = True
    a.use_learnable = learnable
    a.decoder_input = "encoded"
    a.z_decode_mode = "max"
    a.discriminator_method = disc
    a.num_classes = 3 if disc == "Multi" else 2
    if learnable:
        a.word_blank, a.word_dropout, a.word_repeat, a.word_add, a.word_shuffle = 0.1, 0.05, 0, 1 / 0.95 - 1, 0
    else:
        a.word_blank, a.word_dropout, a.word_repeat, a.word_add, a.word_shuffle = 0.2, 0, 0, 0, 0
    a.gen_approx_mode = a.dis_approx_mode = "gumbel"
    a.slf_factor = a.cyc_factor = 0.5
    a.length_factor = 1.0
    a.fw_adv_factor, a.bw_adv_factor = 1.5, 1.0
    a.lr_F = a.lr_D = 1e-4
    a.L2 = 0
    a.radam = False
    a.pre_z_batch = 0
    a.F_pretrain_iter = 0
    a.temperature_config = [(1, 0)]
    a.batch_size = 2
    a.iter_D, a.iter_F = 10, 5
    a.name, a.model_dir, a.log_dir, a.out_dir = "golden", "/tmp/golden_model", "/tmp/golden_tb", "/tmp/golden_out"
    a.checkpoint_steps, a.checkpoint_max_to_keep = 5, 20
    a.show_sample = [0]
    a.restore = None
    a.cuda = "cpu"
    a.restore_optimizer = True
    for k, v in over.items():
        a[k] = v
    p = S()
    p.args = a
    p.volatile = S()
    p.volatile.dm = FakeDM(V, np.random.default_rng(seed + 17), L)
    p.volatile.load_exclude_set = []
    p.volatile.restoreCallback = None
    return p


def redraw_embeddings(module, gen):
    """Reference quirk: all embeddings initialise to exactly 0. Re-draw them so that
    alignment costs are not all ties (the all-zero case is covered by a dedicated KAT)."""
    for name, p in module.named_parameters():
        if name.endswith("embed.weight") or name.endswith("_embed.weight"):
            bound = float(np.sqrt(6.0 / (p.shape[0] + p.shape[1])))
            with torch.no_grad():
                p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1) * bound)
        if name.endswith(".bias") and "layer_norm" not in name and "norm." not in name:
            with torch.no_grad():
                p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1) * 0.05)
        if name.endswith("norm.weight") or name.endswith("norm.bias"):
            with torch.no_grad():
                p.add_((torch.rand(p.shape, generator=gen) * 2 - 1) * 0.1)


def synth_batch(rng, B, L, V, full_first=True):
    """[B,L] token ids: len-1 random tokens, eos, then pad; lengths include eos."""
    lens = rng.integers(3, L + 1, size=B)
    if full_first:
        lens[0] = L
    sent = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = int(lens[b])
        sent[b, :n - 1] = rng.integers(4, V, size=n - 1)
        sent[b, n - 1] = 2
    return torch.tensor(sent), torch.tensor(lens, dtype=torch.long)


def np_(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()   # copy: Adam updates parameters in place
    return np.asarray(t)


def put_state(dst, prefix, module):
    for k, v in module.state_dict().items():
        dst[f"{prefix}{k}"] = np_(v)


def put_grads(dst, prefix, module):
    for k, p in module.named_parameters():
        if p.grad is not None:
            dst[f"{prefix}{k}"] = np_(p.grad)


def save(out_dir, name, d):
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in d.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB, {len(d)} arrays)")


def gumbel_from_exp(E):
    return torch.min(-E.log(), torch.ones(1) * 1e10)


class ExpRecorder:
    """Records every tensor produced by Tensor.exponential_() (the reference's Gumbel source)."""

    def __init__(self):
        self.draws = []
        self._orig = torch.Tensor.exponential_

    def __enter__(self):
        rec = self

        def patched(t, *a, **k):
            r = rec._orig(t, *a, **k)
            rec.draws.append(r.detach().clone())
            return r

        torch.Tensor.exponential_ = patched
        return self

    def __exit__(self, *exc):
        torch.Tensor.exponential_ = self._orig


# ----------------------------------------------------------------------------------------------
# cases
# ----------------------------------------------------------------------------------------------
def case_layers(ref_tf, out_dir):
    """G1: MultiHeadAttention / EncoderLayer / DecoderLayer fwd+bwd with padding + causal masks."""
    torch.manual_seed(101)
    gen = torch.Generator().manual_seed(101)
    B, n, m, d, h = 3, 7, 9, 32, 4
    g = {}
    x = torch.randn(B, n, d, requires_grad=True)
    mem = torch.randn(B, m, d, requires_grad=True)
    klen_self = torch.tensor([7, 4, 1])
    klen_mem = torch.tensor([9, 5, 2])
    pad_self = (torch.arange(n)[None] >= klen_self[:, None]).view(B, 1, 1, n)
    pad_mem = (torch.arange(m)[None] >= klen_mem[:, None]).view(B, 1, 1, m)
    causal = (torch.ones(n, n).tril() == 0).view(1, 1, n, n)
    g["in/x"], g["in/mem"] = np_(x), np_(mem)
    g["in/pad_self"], g["in/pad_mem"], g["in/causal"] = np_(pad_self), np_(pad_mem), np_(causal)

    mha = ref_tf.MultiHeadAttention(d, h, 0)
    redraw_embeddings(mha, gen)
    put_state(g, "w/mha/", mha)
    for tag, q, kv, mask in (("self_pad", x, x, pad_self), ("cross_pad", x, mem, pad_mem), ("self_causal", x, x, causal)):
        R = torch.randn(B, n, d, generator=gen)
        y = mha(q, kv, kv, mask)
        mha.zero_grad(); x.grad = None; mem.grad = None
        (y * R).sum().backward()
        g[f"in/mha/{tag}/R"] = np_(R)
        g[f"out/mha/{tag}/y"] = np_(y)
        g[f"g/mha/{tag}/x"] = np_(x.grad)
        if kv is mem:
            g[f"g/mha/{tag}/mem"] = np_(mem.grad)
        put_grads(g, f"g/mha/{tag}/w/", mha)

    enc = ref_tf.EncoderLayer(d, h, 0)
    redraw_embeddings(enc, gen)
    put_state(g, "w/enc/", enc)
    R = torch.randn(B, n, d, generator=gen)
    y = enc(x, pad_self)
    enc.zero_grad(); x.grad = None
    (y * R).sum().backward()
    g["in/enc/R"], g["out/enc/y"], g["g/enc/x"] = np_(R), np_(y), np_(x.grad)
    put_grads(g, "g/enc/w/", enc)

    dec = ref_tf.DecoderLayer(d, h, 0)
    redraw_embeddings(dec, gen)
    put_state(g, "w/dec/", dec)
    for tag, tmask in (("pad", pad_self), ("causal", causal)):
        R = torch.randn(B, n, d, generator=gen)
        y = dec(x, mem, pad_mem, tmask)
        dec.zero_grad(); x.grad = None; mem.grad = None
        (y * R).sum().backward()
        g[f"in/dec/{tag}/R"], g[f"out/dec/{tag}/y"] = np_(R), np_(y)
        g[f"g/dec/{tag}/x"], g[f"g/dec/{tag}/mem"] = np_(x.grad), np_(mem.grad)
        put_grads(g, f"g/dec/{tag}/w/", dec)
    save(out_dir, "g1_layers", g)


def _cfg_arrays(g, **cfg):
    for k, v in cfg.items():
        g[f"cfg/{k}"] = np.asarray(v)


TINY = dict(V=40, d=32, h=4, N=2, L=8)


def case_style_transformer(ref_utils, ref_tf, out_dir):
    """G2/G3 (simple, hard + soft tokens) and G4/G5 (learnable teacher-forced + generate)."""
    for learnable in (False, True):
        tag = "learnable" if learnable else "simple"
        torch.manual_seed(7 if learnable else 5)
        gen = torch.Generator().manual_seed(11)
        rng = np.random.default_rng(3)
        B = 4
        param = make_param(ref_utils, learnable=learnable, **TINY)
        V, d, L = TINY["V"], TINY["d"], TINY["L"]
        F = ref_tf.StyleTransformer(param)
        redraw_embeddings(F, gen)
        g = {}
        _cfg_arrays(g, B=B, learnable=int(learnable), **TINY)
        put_state(g, "w/", F)
        gold, gold_len = synth_batch(rng, B, L, V)
        # noisy source of different length (learnable) / same length (simple)
        if learnable:
            src, src_len = synth_batch(rng, B, L - 1, V)
            # share some tokens with gold so that the alignment is non-trivial
            for b in range(B):
                n = int(min(src_len[b], gold_len[b])) - 1
                src[b, :n] = gold[b, :n]
                if n > 2:
                    src[b, 1] = 3  # an <unk> (blanked word)
        else:
            src, src_len = gold.clone(), gold_len.clone()
            src[0, 2] = 3
        style = torch.tensor([0, 0, 1, 1])
        g["in/src"], g["in/src_len"], g["in/gold"], g["in/gold_len"], g["in/style"] = \
            np_(src), np_(src_len), np_(gold), np_(gold_len), np_(style)

        # --- teacher-forced, hard source
        lp, tl, ll, ge, _ = F(src, src_len, gold, gold_len, style, generate=False, temperature=1.0, use_inference_z=True)
        R = torch.randn(lp.shape, generator=gen)
        loss = (lp * R).sum()
        if learnable:
            W = torch.rand(ll.shape, generator=gen)
            loss = loss + (ll * W).sum()
            g["in/teacher_hard/W"] = np_(W)
            g["out/teacher_hard/length_loss"], g["out/teacher_hard/gen_err"] = np_(ll), np_(ge)
            g["out/teacher_hard/tgt_lengths"] = np_(tl)
            with torch.no_grad():
                g["out/teacher_hard/allpos"] = np_(F.length_predictor.match(src, src_len, gold, gold_len, F.embed))
        F.zero_grad()
        loss.backward()
        g["in/teacher_hard/R"], g["out/teacher_hard/log_probs"] = np_(R), np_(lp)
        put_grads(g, "g/teacher_hard/", F)

        # --- teacher-forced with temperature != 1
        lp2 = F(src, src_len, gold, gold_len, style, generate=False, temperature=0.7, use_inference_z=True)[0]
        g["out/teacher_hard_t07/log_probs"] = np_(lp2)

        # --- teacher-forced, soft source [B,S,V]
        soft = torch.softmax(torch.randn(B, src.shape[1], V, generator=gen) * 2.0, -1)
        hard1h = torch.zeros(B, src.shape[1], V).scatter_(-1, src.unsqueeze(-1), 1.0)
        soft = (0.7 * hard1h + 0.3 * soft).requires_grad_(True)
        lp, tl, ll, ge, _ = F(soft, src_len, gold, gold_len, style, generate=False, temperature=1.0, use_inference_z=True)
        R = torch.randn(lp.shape, generator=gen)
        loss = (lp * R).sum()
        if learnable:
            W = torch.rand(ll.shape, generator=gen)
            loss = loss + (ll * W).sum()
            g["in/teacher_soft/W"] = np_(W)
            g["out/teacher_soft/length_loss"], g["out/teacher_soft/gen_err"] = np_(ll), np_(ge)
            with torch.no_grad():
                g["out/teacher_soft/allpos"] = np_(F.length_predictor.match(soft.detach(), src_len, gold, gold_len, F.embed))
        F.zero_grad()
        loss.backward()
        g["in/teacher_soft/soft"], g["in/teacher_soft/R"] = np_(soft), np_(R)
        g["out/teacher_soft/log_probs"] = np_(lp)
        g["gx/teacher_soft/soft"] = np_(soft.grad)
        put_grads(g, "g/teacher_soft/", F)

        # --- generate (gold=None), hard source, other style
        lp, tl, _, _, allpos = F(gold, gold_len, None, None, 1 - style, generate=True, temperature=1.0, use_inference_z=True)
        R = torch.randn(lp.shape, generator=gen)
        F.zero_grad()
        (lp * R).sum().backward()
        g["in/generate/R"], g["out/generate/log_probs"] = np_(R), np_(lp)
        if learnable:
            g["out/generate/tgt_lengths"], g["out/generate/allpos"] = np_(tl), np_(allpos)
        put_grads(g, "g/generate/", F)

        if learnable:
            # use_inference_z=False branches (pre_z_batch regime): identity alignment
            lp = F(gold, gold_len, gold, gold_len, style, generate=False, temperature=1.0, use_inference_z=False)
            g["out/teacher_noz/log_probs"], g["out/teacher_noz/length_loss"] = np_(lp[0]), np_(lp[2])
            # NB: generate=True with use_inference_z=False raises in the reference itself
            # (transformer.py:78 unpacks 3 values, :522 returns 2) -> no golden for that branch.
            # G11: HiddenDecoder forward (no final LN) vs incremental_forward (final LN)
            hd = F.length_predictor.decoder
            n = 6
            x = torch.randn(B, n, d, generator=gen)
            mem = torch.randn(B, 5, d, generator=gen)
            smask = (torch.arange(5)[None] >= torch.tensor([5, 3, 4, 2])[:, None]).view(B, 1, 1, 5)
            causal = (torch.ones(n, n).tril() == 0).view(1, 1, n, n)
            with torch.no_grad():
                yf = hd(x, mem, smask, causal)
                st, ys = None, []
                for k in range(n):
                    yk, st = hd.incremental_forward(x[:, k:k + 1], mem, smask, causal[:, :, k:k + 1, :k + 1], st)
                    ys.append(yk)
            g["in/hd/x"], g["in/hd/mem"], g["in/hd/smask"] = np_(x), np_(mem), np_(smask)
            g["out/hd/forward"], g["out/hd/incremental"] = np_(yf), np_(torch.cat(ys, 1))
        save(out_dir, f"g2_style_transformer_{tag}", g)


def case_match(ref_utils, ref_tf, out_dir):
    """G6: alignment-DP known-answer tests with the cost tensors captured."""
    g = {}
    gen = torch.Generator().manual_seed(23)
    cases = []

    def run(name, V, d, L, src, src_len, tgt, tgt_len, zero_embed=False, soft=False):
        param = make_param(ref_utils, learnable=True, V=V, d=d, h=4, N=1, L=L)
        torch.manual_seed(1)
        F = ref_tf.StyleTransformer(param)
        if not zero_embed:
            redraw_embeddings(F, gen)
        lp = F.length_predictor
        src_t, tgt_t = torch.tensor(src), torch.tensor(tgt)
        sl, tl = torch.tensor(src_len), torch.tensor(tgt_len)
        with torch.no_grad():
            if soft:
                oh = torch.zeros(src_t.shape + (V,)).scatter_(-1, src_t.unsqueeze(-1), 1.0)
                rnd = torch.softmax(torch.randn(src_t.shape + (V,), generator=gen), -1)
                src_in = 0.8 * oh + 0.2 * rnd
            else:
                src_in = src_t
            allpos = lp.match(src_in, sl, tgt_t, tl, F.embed)
            # recompute the cost tensors exactly as transformer.py:396-412 does
            if soft:
                o = src_in.clone()
                o[:, :, 0] += o[:, :, 3]
                o[:, :, 3] *= 0
                oe = torch.matmul(o, F.embed.token_embed.weight)
            else:
                o = src_t.clone()
                o[o == 3] = 0
                oe = F.embed.token_embed(o)
            te = F.embed.token_embed(tgt_t)
            cost0 = (te ** 2).sum(-1) + 1e-6
            cost = torch.cdist(te, oe) ** 2
        g[f"{name}/embed"] = np_(F.embed.token_embed.weight)
        g[f"{name}/src"], g[f"{name}/src_len"], g[f"{name}/tgt"], g[f"{name}/tgt_len"] = np_(src_t), np_(sl), np_(tgt_t), np_(tl)
        if soft:
            g[f"{name}/src_soft"] = np_(src_in)
        g[f"{name}/cost0"], g[f"{name}/cost"], g[f"{name}/allpos"] = np_(cost0), np_(cost), np_(allpos)
        g[f"{name}/origin_emb"], g[f"{name}/target_emb"] = np_(oe), np_(te)
        cases.append(name)
        return allpos

    # deletion/insertion example from SURVEY §8c (ids: 2=eos, 3=unk)
    a = run("insdel", 20, 16, 8, [[10, 12, 3, 14, 15, 2, 0]], [6], [[10, 11, 12, 13, 14, 15, 2]], [7])
    print("   insdel allpos:", a.tolist())
    a = run("identity", 20, 16, 8, [[10, 11, 12, 13, 14, 15, 2]], [7], [[10, 11, 12, 13, 14, 15, 2]], [7])
    print("   identity allpos:", a.tolist())
    # all-zero embeddings: every cost ties, tie-breaking rules observable
    a = run("allzero", 20, 16, 8, [[10, 12, 3, 14, 15, 2, 0], [5, 2, 0, 0, 0, 0, 0]], [6, 2],
            [[10, 11, 12, 13, 14, 15, 2], [7, 8, 9, 2, 0, 0, 0]], [7, 4], zero_embed=True)
    print("   allzero allpos:", a.tolist())
    # short sources (origin_length < 3), target_length 1
    run("short", 20, 16, 8, [[9, 2, 0, 0], [2, 0, 0, 0], [4, 5, 6, 2]], [2, 1, 4],
        [[9, 8, 7, 2, 0], [2, 0, 0, 0, 0], [2, 0, 0, 0, 0]], [4, 1, 1])
    # random ragged batch, hard and soft source
    rng = np.random.default_rng(5)
    src, sl = synth_batch(rng, 6, 12, 30)
    tgt, tl = synth_batch(rng, 6, 14, 30)
    for b in range(6):
        n = int(min(sl[b], tl[b])) - 1
        tgt[b, :n] = src[b, :n]
    src[1, 2] = 3
    run("ragged_hard", 30, 24, 16, src.tolist(), sl.tolist(), tgt.tolist(), tl.tolist())
    run("ragged_soft", 30, 24, 16, src.tolist(), sl.tolist(), tgt.tolist(), tl.tolist(), soft=True)
    # S = T = 32 (torch.cdist switches to the matmul formulation above 25 rows)
    src, sl = synth_batch(rng, 5, 32, 60)
    tgt, tl = synth_batch(rng, 5, 32, 60)
    for b in range(5):
        n = int(min(sl[b], tl[b])) - 1
        tgt[b, 2:n] = src[b, 1:n - 1]
    run("len32", 60, 32, 32, src.tolist(), sl.tolist(), tgt.tolist(), tl.tolist())
    g["cases"] = np.array(cases)
    save(out_dir, "g6_match", g)


def case_gumbel(ref_utils, out_dir):
    """G7: gumbel family with the Exp(1) draws captured; G10: get_lengths KATs are in case_misc."""
    g = {}
    gen = torch.Generator().manual_seed(31)
    B, L, V = 3, 5, 40
    logits = torch.log_softmax(torch.randn(B, L, V, generator=gen) * 1.5, -1)
    g["in/logits"] = np_(logits)
    for name, tau in (("gumbel_max", 1), ("gumbel_max_tau", 0.5)):
        x = logits.clone().requires_grad_(True)
        with ExpRecorder() as rec:
            torch.manual_seed(77)
            ret, y_soft = ref_utils.gumbel_max(x, tau=tau)
        R1, R2 = torch.randn(ret.shape, generator=gen), torch.randn(ret.shape, generator=gen)
        (ret * R1).sum().backward(retain_graph=True)
        g_ret = x.grad.clone(); x.grad = None
        (y_soft * R2).sum().backward()
        g[f"in/{name}/E"], g[f"in/{name}/tau"] = np_(rec.draws[0]), np.asarray(tau, dtype=np.float32)
        g[f"in/{name}/R_ret"], g[f"in/{name}/R_soft"] = np_(R1), np_(R2)
        g[f"out/{name}/ret"], g[f"out/{name}/y_soft"] = np_(ret), np_(y_soft)
        g[f"g/{name}/from_ret"], g[f"g/{name}/from_soft"] = np_(g_ret), np_(x.grad)

    x = logits.clone().requires_grad_(True)
    with ExpRecorder() as rec:
        torch.manual_seed(78)
        y = ref_utils.gumbel_softmax(x, tau=0.8)
    R = torch.randn(y.shape, generator=gen)
    (y * R).sum().backward()
    g["in/gumbel_softmax/E"], g["in/gumbel_softmax/R"] = np_(rec.draws[0]), np_(R)
    g["out/gumbel_softmax/y"], g["g/gumbel_softmax/x"] = np_(y), np_(x.grad)

    mask = (torch.rand(B, L, V, generator=gen) > 0.3).long()
    mask[..., 0] = 1
    x = logits.clone().requires_grad_(True)
    with ExpRecorder() as rec:
        torch.manual_seed(79)
        y = ref_utils.gumbel_max_with_mask(x, mask, tau=1)
    R = torch.randn(y.shape, generator=gen)
    (y * R).sum().backward()
    g["in/gumbel_max_with_mask/E"], g["in/gumbel_max_with_mask/R"], g["in/gumbel_max_with_mask/mask"] = np_(rec.draws[0]), np_(R), np_(mask)
    g["out/gumbel_max_with_mask/y"], g["g/gumbel_max_with_mask/x"] = np_(y), np_(x.grad)

    for allow in (False, True):
        x = logits.clone().requires_grad_(True)
        y_hard, y_soft = ref_utils.straight_max(x, allow_gradient=allow)
        R = torch.randn(y_hard.shape, generator=gen)
        if allow:
            (y_hard * R).sum().backward()
            g["g/straight_max_grad/x"] = np_(x.grad)
        g[f"in/straight_max_{int(allow)}/R"] = np_(R)
        g[f"out/straight_max_{int(allow)}/y_hard"], g[f"out/straight_max_{int(allow)}/y_soft"] = np_(y_hard), np_(y_soft)
    sent = torch.randint(0, V, (B, L), generator=gen)
    g["in/onehot/sent"], g["out/onehot/y"] = np_(sent), np_(ref_utils.onehot(sent, V))
    save(out_dir, "g7_gumbel", g)


def case_discriminator(ref_utils, ref_tf, out_dir):
    """G8: Discriminator Cond (style given) and Multi (style=None), hard + soft tokens."""
    for disc in ("Cond", "Multi"):
        torch.manual_seed(13)
        gen = torch.Generator().manual_seed(41)
        rng = np.random.default_rng(9)
        B = 4
        V, d, L = TINY["V"], TINY["d"], TINY["L"]
        param = make_param(ref_utils, learnable=False, disc=disc, **TINY)
        D = ref_tf.Discriminator(param)
        redraw_embeddings(D, gen)
        g = {}
        _cfg_arrays(g, B=B, disc=disc, **TINY)
        put_state(g, "w/", D)
        tok, tl = synth_batch(rng, B, L, V)
        style = torch.tensor([0, 1, 1, 0]) if disc == "Cond" else None
        g["in/tok"], g["in/len"] = np_(tok), np_(tl)
        if style is not None:
            g["in/style"] = np_(style)
        lp = D(tok, tl, style)
        R = torch.randn(lp.shape, generator=gen)
        D.zero_grad()
        (lp * R).sum().backward()
        g["in/hard/R"], g["out/hard/log_probs"] = np_(R), np_(lp)
        put_grads(g, "g/hard/", D)
        soft = torch.softmax(torch.randn(B, L, V, generator=gen) * 2, -1).requires_grad_(True)
        lp = D(soft, tl, style)
        R = torch.randn(lp.shape, generator=gen)
        D.zero_grad()
        (lp * R).sum().backward()
        g["in/soft/soft"], g["in/soft/R"], g["out/soft/log_probs"] = np_(soft), np_(R), np_(lp)
        g["gx/soft/soft"] = np_(soft.grad)
        put_grads(g, "g/soft/", D)
        save(out_dir, f"g8_discriminator_{disc.lower()}", g)


def case_steps(ref_utils, ref_tf, ref_s2s, out_dir):
    """G9: two consecutive f_step / d_step calls (simple + learnable, + pretrain variant).
    NoiseLayer outputs and Exp(1) draws are captured as inputs."""
    import torch.nn.utils as tnu
    for learnable in (False, True):
        for mode in ("f_step", "d_step", "f_step_pretrain"):
            tag = f"{'learnable' if learnable else 'simple'}_{mode}"
            torch.manual_seed(19)
            np.random.seed(19)
            gen = torch.Generator().manual_seed(43)
            rng = np.random.default_rng(21)
            B = 4
            V, d, L = TINY["V"], TINY["d"], TINY["L"]
            param = make_param(ref_utils, learnable=learnable, **TINY)
            model = ref_s2s.Seq2seq(param)
            redraw_embeddings(model.net, gen)
            g = {}
            _cfg_arrays(g, B=B, learnable=int(learnable), **TINY)
            put_state(g, "w0/", model.net)
            model.net.model_F.train(); model.net.model_D.train()

            norms = []
            orig_clip = ref_s2s.clip_grad_norm_

            def rec_clip(params, max_norm, *a, **k):
                r = orig_clip(params, max_norm, *a, **k)
                norms.append(float(r))
                return r

            ref_s2s.clip_grad_norm_ = rec_clip
            noise_out = []
            orig_noise = model.net.noiseLayer.forward

            def rec_noise(words, lengths, rev_domain):
                w, l = orig_noise(words, lengths, rev_domain)
                noise_out.append((w.clone(), l.clone()))
                return w, l

            model.net.noiseLayer.forward = rec_noise
            try:
                for step in range(2):
                    sent, slen = synth_batch(rng, B, L, V, full_first=True)
                    inc = ref_utils.Storage()
                    inc.data = ref_utils.Storage()
                    inc.data.sent, inc.data.sent_length = sent, slen
                    inc.data.domain = torch.tensor([0, 0, 1, 1])
                    inc.data.batch_size = B
                    g[f"s{step}/in/sent"], g[f"s{step}/in/sent_length"], g[f"s{step}/in/domain"] = np_(sent), np_(slen), np_(inc.data.domain)
                    noise_out.clear()
                    with ExpRecorder() as rec:
                        if mode == "f_step":
                            ret = model.f_step(inc, 1.0, use_inference_z=True)
                        elif mode == "f_step_pretrain":
                            ret = model.f_step(inc, 1.0, cyc_rec_enable=False, use_inference_z=True)
                        else:
                            ret = (model.d_step(inc, 1.0, use_inference_z=True),)
                    for i, E in enumerate(rec.draws):
                        g[f"s{step}/in/E{i}"] = np_(E)
                    g[f"s{step}/in/nE"] = np.asarray(len(rec.draws))
                    if noise_out:
                        g[f"s{step}/in/noise_tok"] = np_(noise_out[0][0].transpose(0, 1))
                        g[f"s{step}/in/noise_len"] = np_(noise_out[0][1])
                    g[f"s{step}/out/ret"] = np.asarray([float(r) for r in ret], dtype=np.float64)
                    g[f"s{step}/out/grad_norm"] = np.asarray(norms[-1] if norms else -1.0)
                    which = model.net.model_D if mode == "d_step" else model.net.model_F
                    put_grads(g, f"s{step}/g/", which)
                    put_state(g, f"s{step}/w/", model.net)
            finally:
                ref_s2s.clip_grad_norm_ = orig_clip
            save(out_dir, f"g9_{tag}", g)


def case_misc(ref_utils, ref_s2s, out_dir):
    """G10: get_lengths KATs and generateMask."""
    g = {}
    gl = ref_s2s.Seq2seq.get_lengths
    toks = torch.tensor([[5, 6, 7, 8], [2, 5, 6, 7], [5, 6, 7, 2], [5, 2, 6, 2], [2, 2, 2, 2]])
    g["get_lengths/in"], g["get_lengths/out"] = np_(toks), np_(gl(None, toks, 2))
    lens = np.array([1, 3, 5, 0])
    g["generateMask/len"] = lens
    g["generateMask/out"] = np_(ref_utils.generateMask(5, lens, type=float))
    save(out_dir, "g10_misc", g)


class DrawRecorder:
    """Records what np.random.rand / np.random.randint / torch.multinomial hand to the reference's NoiseLayer."""

    def __init__(self):
        self.rand, self.randint, self.multinomial = [], [], []

    def __enter__(self):
        self._o = (np.random.rand, np.random.randint, torch.multinomial)
        rec = self

        def rand(*shape):
            r = rec._o[0](*shape)
            rec.rand.append(np.array(r, copy=True))
            return r

        def randint(lo, hi=None, *a, **k):
            r = rec._o[1](lo, hi, *a, **k)
            rec.randint.append((int(lo), int(hi), int(r)))
            return r

        def multinomial(*a, **k):
            r = rec._o[2](*a, **k)
            rec.multinomial.append(r.detach().clone())
            return r

        np.random.rand, np.random.randint, torch.multinomial = rand, randint, multinomial
        return self

    def __exit__(self, *exc):
        np.random.rand, np.random.randint, torch.multinomial = self._o


def case_noise(ref_utils, out_dir):
    """G11: the three stages of the reference's NoiseLayer.forward that the run configuration uses (noise.py:60-74:
    word_blank :103-135, word_add :76-101, word_dropout :137-194), each with the uniform draws / multinomial candidates /
    randint picks it consumed captured next to its input and output."""
    import noise as ref_noise                     # /root/reference/styletransformer/noise.py
    g = {}
    cases = [("run", dict(blank=0.1, drop=0.05, rep=0.0, add=1 / 0.95 - 1, L=16, V=60, B=37)),      # run.py:90-95
             ("heavy", dict(blank=0.5, drop=0.6, rep=0.3, add=0.5, L=12, V=40, B=70)),
             ("alldrop", dict(blank=0.3, drop=0.93, rep=0.0, add=0.3, L=8, V=30, B=33)),           # most sentences lose every word
             ("narrow", dict(blank=0.2, drop=0.3, rep=0.4, add=0.6, L=14, V=30, B=20, S=9))]        # input narrower than max_length
    names = []
    for ci, (tag, c) in enumerate(cases):
        rng = np.random.default_rng(300 + ci)
        np.random.seed(300 + ci)
        torch.manual_seed(300 + ci)
        L, V, B = c["L"], c["V"], c["B"]
        S = c.get("S", L)
        nl = ref_noise.NoiseLayer(c["blank"], c["drop"], 0, c["rep"], c["add"], keep_go=False, max_length=L, dm=FakeDM(V, rng, L))
        for trial in range(3):
            name = f"{tag}{trial}"
            names.append(name)
            sent, slen = synth_batch(rng, B, S, V, full_first=(trial != 1))
            if trial == 2:                        # a few one-word sentences
                sent[:5, 1] = 2
                sent[:5, 2:] = 0
                slen[:5] = 2
            words = sent.t().contiguous()
            rev = torch.tensor(rng.integers(0, 2, size=B))
            l0 = slen.numpy()
            p = f"{name}/"
            for k in ("blank", "drop", "rep", "add", "L", "V", "B"):
                g[p + "cfg/" + k] = np.asarray(c[k])
            g[p + "in/x"], g[p + "in/l"], g[p + "in/rev_domain"] = np_(words), np.asarray(l0), np_(rev)
            with DrawRecorder() as r:
                x1, l1 = nl.word_blank(words, l0)
            g[p + "blank/u"] = r.rand[0]
            g[p + "blank/x"], g[p + "blank/l"] = np_(x1), np.asarray(l1)
            with DrawRecorder() as r:
                x2, l2 = nl.word_add(x1, l1, rev)
            g[p + "add/u"] = r.rand[0]
            g[p + "add/new_words"] = np_(r.multinomial[0].transpose(0, 1))            # [M, B] as the reference uses them
            g[p + "add/x"], g[p + "add/l"] = np_(x2), np.asarray(l2)
            with DrawRecorder() as r:
                x3, l3 = nl.word_dropout(x2, l2)
            g[p + "drop/u_keep"], g[p + "drop/u_rep"] = r.rand[0], r.rand[1]
            g[p + "drop/picks"] = np.asarray(r.randint, dtype=np.int64).reshape(-1, 3)   # (lo, hi, result) per all-dropped sentence, batch order
            g[p + "drop/x"], g[p + "drop/l"] = np_(x3), np.asarray(l3, dtype=np.int64)
    g["names"] = np.asarray(names)
    g["cnt_check/p"] = np_(nl.cnt)
    save(out_dir, "g11_noise_stages", g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    ap.add_argument("--only", default=None, help="generate one case only (noise)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(1)  # deterministic summation order
    ref_utils, ref_tf, ref_s2s = import_reference()
    print("reference imported; generating goldens ...")
    if a.only == "noise":
        case_noise(ref_utils, a.out)
        return
    case_layers(ref_tf, a.out)
    case_style_transformer(ref_utils, ref_tf, a.out)
    case_match(ref_utils, ref_tf, a.out)
    case_gumbel(ref_utils, a.out)
    case_discriminator(ref_utils, ref_tf, a.out)
    case_steps(ref_utils, ref_tf, ref_s2s, a.out)
    case_misc(ref_utils, ref_s2s, a.out)
    case_noise(ref_utils, a.out)
    print("done")


if __name__ == "__main__":
    main()
