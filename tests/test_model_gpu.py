"""GPU: the product modules (HIP kernels behind the reference's nn.Module API) against fixtures captured
from the unmodified reference.  fp32 tolerance 1e-4 relative (north_star); alignment indices exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from helpers import make_param, cfg_of, load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def relclose(a, b, tol=1e-4, what=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = np.abs(b).max() + 1e-30
    assert err <= tol * scale + 1e-7, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def check_grads(module, gold, tol=1e-4, gscale=None):
    """Every golden gradient must match; tolerance relative to the largest gradient in the model
    (per-tensor where the tensor's own scale is meaningful)."""
    named = dict(module.named_parameters())
    gscale = gscale or max(float(np.abs(v).max()) for v in gold.values())
    n = 0
    for k, gv in gold.items():
        p = named[k]
        assert p.grad is not None, k
        err = float((p.grad.detach().cpu().double() - torch.from_numpy(gv).double()).abs().max())
        scale = max(float(np.abs(gv).max()), 1e-3 * gscale)
        assert err <= tol * scale, f"grad {k}: err {err:.3e} scale {scale:.3e}"
        n += 1
    for k, p in named.items():
        if k not in gold:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{k} has a gradient the reference lacks"
    assert n > 0


def test_layers_golden():
    from nast_amd import transformer as tf
    g = load_golden("g1_layers")
    x, mem = T(g["in/x"]).requires_grad_(True), T(g["in/mem"]).requires_grad_(True)
    masks = {"self_pad": T(g["in/pad_self"]), "cross_pad": T(g["in/pad_mem"]), "self_causal": T(g["in/causal"])}
    mha = tf.MultiHeadAttention(32, 4, 0).to(DEV)
    load_weights(mha, sub(g, "w/mha/"))
    for tag in ("self_pad", "cross_pad", "self_causal"):
        kv = mem if tag == "cross_pad" else x
        mha.zero_grad(); x.grad = None; mem.grad = None
        y = mha(x, kv, kv, masks[tag])
        (y * T(g[f"in/mha/{tag}/R"])).sum().backward()
        relclose(y, g[f"out/mha/{tag}/y"], what=tag)
        relclose(x.grad, g[f"g/mha/{tag}/x"], 1e-4, tag)
        check_grads(mha, sub(g, f"g/mha/{tag}/w/"))
    enc = tf.EncoderLayer(32, 4, 0).to(DEV)
    load_weights(enc, sub(g, "w/enc/"))
    x.grad = None
    y = enc(x, masks["self_pad"])
    (y * T(g["in/enc/R"])).sum().backward()
    relclose(y, g["out/enc/y"]); relclose(x.grad, g["g/enc/x"], 1e-4)
    check_grads(enc, sub(g, "g/enc/w/"))
    dec = tf.DecoderLayer(32, 4, 0).to(DEV)
    load_weights(dec, sub(g, "w/dec/"))
    for tag, tm in (("pad", masks["self_pad"]), ("causal", masks["self_causal"])):
        dec.zero_grad(); x.grad = None; mem.grad = None
        y = dec(x, mem, masks["cross_pad"], tm)
        (y * T(g[f"in/dec/{tag}/R"])).sum().backward()
        relclose(y, g[f"out/dec/{tag}/y"]); relclose(x.grad, g[f"g/dec/{tag}/x"], 1e-4); relclose(mem.grad, g[f"g/dec/{tag}/mem"], 1e-4)
        check_grads(dec, sub(g, f"g/dec/{tag}/w/"))
    # the fine-grained public pieces compose to the same numbers as the fused sublayers
    y2 = enc.sublayer[0](x, lambda z: enc.self_attn(z, z, z, masks["self_pad"]))
    y2 = enc.sublayer[1](y2, enc.pw_ffn)
    relclose(y2, g["out/enc/y"])
    q = torch.randn(3, 4, 7, 8, device=DEV)
    out, _ = tf.scaled_attention(q, q, q, masks["self_pad"])
    ref = torch.softmax((q @ q.transpose(-1, -2) / 8 ** 0.5).masked_fill(masks["self_pad"], float("-inf")), -1) @ q
    relclose(out, ref.cpu().numpy())


@pytest.mark.parametrize("tag", ["simple", "learnable"])
def test_style_transformer_golden(tag):
    from nast_amd import transformer as tf
    g = load_golden(f"g2_style_transformer_{tag}")
    F = tf.StyleTransformer(make_param(**cfg_of(g))).to(DEV)
    assert not load_weights(F, sub(g, "w/"))
    learnable = bool(g["cfg/learnable"])
    src, sl, gold, gl, style = (T(g[k]) for k in ("in/src", "in/src_len", "in/gold", "in/gold_len", "in/style"))

    F.zero_grad()
    lp, tl, ll, ge, _ = F(src, sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True)
    relclose(lp, g["out/teacher_hard/log_probs"], what="teacher_hard")
    loss = (lp * T(g["in/teacher_hard/R"])).sum()
    if learnable:
        with torch.no_grad():
            ap = F.length_predictor.match(src, sl, gold, gl, F.embed)
        assert np.array_equal(ap.cpu().numpy(), g["out/teacher_hard/allpos"])
        relclose(ll, g["out/teacher_hard/length_loss"]); relclose(ge, g["out/teacher_hard/gen_err"])
        assert np.array_equal(tl.cpu().numpy(), g["out/teacher_hard/tgt_lengths"])
        loss = loss + (ll * T(g["in/teacher_hard/W"])).sum()
    loss.backward()
    check_grads(F, sub(g, "g/teacher_hard/"))
    with torch.no_grad():
        relclose(F(src, sl, gold, gl, style, generate=False, temperature=0.7, use_inference_z=True)[0],
                 g["out/teacher_hard_t07/log_probs"], what="t07")

    F.zero_grad()
    soft = T(g["in/teacher_soft/soft"]).requires_grad_(True)
    lp, tl, ll, ge, _ = F(soft, sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True)
    relclose(lp, g["out/teacher_soft/log_probs"], what="teacher_soft")
    loss = (lp * T(g["in/teacher_soft/R"])).sum()
    if learnable:
        with torch.no_grad():
            ap = F.length_predictor.match(soft.detach(), sl, gold, gl, F.embed)
        assert np.array_equal(ap.cpu().numpy(), g["out/teacher_soft/allpos"])
        relclose(ll, g["out/teacher_soft/length_loss"])
        loss = loss + (ll * T(g["in/teacher_soft/W"])).sum()
    loss.backward()
    relclose(soft.grad, g["gx/teacher_soft/soft"], 1e-4)
    check_grads(F, sub(g, "g/teacher_soft/"))

    F.zero_grad()
    lp, tl, _, _, ap = F(gold, gl, None, None, 1 - style, generate=True, temperature=1.0, use_inference_z=True)
    if learnable:
        assert np.array_equal(ap.cpu().numpy(), g["out/generate/allpos"])
        assert np.array_equal(tl.cpu().numpy(), g["out/generate/tgt_lengths"])
    relclose(lp, g["out/generate/log_probs"], what="generate")
    (lp * T(g["in/generate/R"])).sum().backward()
    check_grads(F, sub(g, "g/generate/"))

    if learnable:
        with torch.no_grad():
            out = F(gold, gl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=False)
            relclose(out[0], g["out/teacher_noz/log_probs"], what="noz")
            hd = F.length_predictor.decoder
            x, mem, sm = T(g["in/hd/x"]), T(g["in/hd/mem"]), T(g["in/hd/smask"])
            n = x.shape[1]
            causal = (torch.ones(n, n, device=DEV).tril() == 0).view(1, 1, n, n)
            relclose(hd(x, mem, sm, causal), g["out/hd/forward"], what="hd.forward")
            st, ys = None, []
            for k in range(n):
                yk, st = hd.incremental_forward(x[:, k:k + 1], mem, sm, causal[:, :, k:k + 1, :k + 1], st)
                ys.append(yk)
            relclose(torch.cat(ys, 1), g["out/hd/incremental"], what="hd.incremental")
            cache = tf.KVCacheDecoder(hd, mem, sm, n)
            relclose(torch.stack([cache.step(x[:, k].contiguous(), k) for k in range(n)], 1), g["out/hd/incremental"], what="kv-cache")


def test_style_transformer_packed_rows_golden():
    """Packed-row execution (ops.packed_rows) of the learnable StyleTransformer against the reference's golden vectors:
    log-probs at the real target positions, per-token NLL, the length losses and every parameter gradient -- with hard and
    with soft (packed: ops.PackedSoft) source tokens.  Cotangents are zero at padded positions, as in the training step."""
    from nast_amd import ops
    from nast_amd import transformer as tf
    g = load_golden("g2_style_transformer_learnable")
    F = tf.StyleTransformer(make_param(**cfg_of(g))).to(DEV)
    assert not load_weights(F, sub(g, "w/"))
    src, sl, gold, gl, style = (T(g[k]) for k in ("in/src", "in/src_len", "in/gold", "in/gold_len", "in/style"))
    live = (torch.arange(gold.shape[1], device=DEV)[None] < gl[:, None])
    ref_lp = T(g["out/teacher_hard/log_probs"])

    with ops.packed_rows(True):
        # log-probs at the real positions
        with torch.no_grad():
            lp = F(src, sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True)[0]
        relclose(lp * live.unsqueeze(-1), ref_lp * live.unsqueeze(-1), what="packed log_probs")
        # NLL + length loss and their gradients == the padded golden run with cotangents masked to the real positions
        Rm = T(g["in/teacher_hard/R"]) * live.unsqueeze(-1)
        F.zero_grad()
        ref = tf.StyleTransformer(make_param(**cfg_of(g))).to(DEV)
        load_weights(ref, sub(g, "w/"))
        Wl = T(g["in/teacher_hard/W"]) * (torch.arange(gold.shape[1] + 1, device=DEV)[None] <= gl[:, None])
    # reference gradients: padded execution (pinned by test_style_transformer_golden) with the same masked cotangents
    lp_r, _, ll_r, _, _ = ref(src, sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True)
    ((lp_r * Rm).sum() + (ll_r * Wl).sum()).backward()
    with ops.packed_rows(True):
        lp_p, _, ll_p, _, _ = F(src, sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True)
        ((lp_p * Rm).sum() + (ll_p * Wl).sum()).backward()
    relclose(ll_p * (Wl != 0), ll_r * (Wl != 0), what="packed length_loss")
    _same_grads(F, ref)

    # soft source tokens: padded tensor vs the same rows handed over packed
    soft = T(g["in/teacher_soft/soft"])
    tok = gold                                   # nll targets
    F.zero_grad(); ref.zero_grad()
    s_ref = soft.clone().requires_grad_(True)
    nll_r = ref(s_ref, sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True, nll_targets=tok)[0]
    (nll_r * live).sum().backward()
    with ops.packed_rows(True):
        ps = ops.PackedSeq(sl, soft.shape[1])
        rows = ops.pack_rows(soft, ps).requires_grad_(True)
        nll_p = F(ops.PackedSoft(rows, ps), sl, gold, gl, style, generate=False, temperature=1.0, use_inference_z=True,
                  nll_targets=tok)[0]
        (nll_p * live).sum().backward()
    relclose(nll_p * live, nll_r * live, what="packed nll (soft source)")
    src_live = (torch.arange(soft.shape[1], device=DEV)[None] < sl[:, None]).unsqueeze(-1)
    relclose(ops.unpack_rows(rows.grad, ps), s_ref.grad * src_live, 1e-4, what="grad wrt soft tokens")
    _same_grads(F, ref)


def _same_grads(model, ref, tol=1e-4):
    """Gradients of two models agree, relative to the largest gradient of the model (key biases have analytically zero
    gradients -- softmax is shift invariant -- whose rounding noise has no scale of its own)."""
    pairs = [(n, p.grad, q.grad) for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters())]
    gscale = max(float(q.abs().max()) for _, _, q in pairs if q is not None)
    for n, p, q in pairs:
        if q is None:
            assert p is None or float(p.abs().max()) == 0, n
        else:
            assert float((p - q).abs().max()) <= tol * gscale, (n, float((p - q).abs().max()), gscale)


@pytest.mark.parametrize("disc", ["cond", "multi"])
def test_discriminator_golden(disc):
    from nast_amd import transformer as tf
    g = load_golden(f"g8_discriminator_{disc}")
    D = tf.Discriminator(make_param(learnable=False, **cfg_of(g))).to(DEV)
    assert not load_weights(D, sub(g, "w/"))
    tok, tl = T(g["in/tok"]), T(g["in/len"])
    style = T(g["in/style"]) if "in/style" in g else None
    lp = D(tok, tl, style)
    relclose(lp, g["out/hard/log_probs"])
    (lp * T(g["in/hard/R"])).sum().backward()
    check_grads(D, sub(g, "g/hard/"))
    D.zero_grad()
    soft = T(g["in/soft/soft"]).requires_grad_(True)
    lp = D(soft, tl, style)
    relclose(lp, g["out/soft/log_probs"])
    (lp * T(g["in/soft/R"])).sum().backward()
    relclose(soft.grad, g["gx/soft/soft"], 1e-4)
    check_grads(D, sub(g, "g/soft/"))


def test_gumbel_api_golden():
    import nast_amd.utils as U
    g = load_golden("g7_gumbel")
    logits = T(g["in/logits"])
    for name in ("gumbel_max", "gumbel_max_tau"):
        x = logits.clone().requires_grad_(True)
        with U.inject_noise([T(g[f"in/{name}/E"])]):
            ret, ys = U.gumbel_max(x, tau=float(g[f"in/{name}/tau"]))
        relclose(ys, g[f"out/{name}/y_soft"]); relclose(ret, g[f"out/{name}/ret"])
        (ret * T(g[f"in/{name}/R_ret"])).sum().backward(retain_graph=True)
        relclose(x.grad, g[f"g/{name}/from_ret"], 1e-4)
        x.grad = None
        (ys * T(g[f"in/{name}/R_soft"])).sum().backward()
        relclose(x.grad, g[f"g/{name}/from_soft"], 1e-4)
    x = logits.clone().requires_grad_(True)
    with U.inject_noise([T(g["in/gumbel_softmax/E"])]):
        y = U.gumbel_softmax(x, tau=0.8)
    relclose(y, g["out/gumbel_softmax/y"])
    (y * T(g["in/gumbel_softmax/R"])).sum().backward()
    relclose(x.grad, g["g/gumbel_softmax/x"], 1e-4)
    x = logits.clone().requires_grad_(True)
    with U.inject_noise([T(g["in/gumbel_max_with_mask/E"])]):
        y = U.gumbel_max_with_mask(x, T(g["in/gumbel_max_with_mask/mask"]))
    relclose(y, g["out/gumbel_max_with_mask/y"])
    (y * T(g["in/gumbel_max_with_mask/R"])).sum().backward()
    relclose(x.grad, g["g/gumbel_max_with_mask/x"], 1e-4)
    for allow in (0, 1):
        x = logits.clone().requires_grad_(True)
        yh, ys = U.straight_max(x, allow_gradient=bool(allow))
        relclose(yh, g[f"out/straight_max_{allow}/y_hard"]); relclose(ys, g[f"out/straight_max_{allow}/y_soft"])
        if allow:
            (yh * T(g["in/straight_max_1/R"])).sum().backward()
            relclose(x.grad, g["g/straight_max_grad/x"], 1e-4)


@pytest.mark.parametrize("tag", ["simple", "learnable"])
@pytest.mark.parametrize("mode", ["f_step", "d_step", "f_step_pretrain"])
@pytest.mark.parametrize("two_streams", [False, True])
def test_training_steps_golden(tag, mode, two_streams):
    """Two consecutive steps from the captured weights / batch / noisy batch / Gumbel draws: returned losses,
    pre-clip gradient norm, clipped gradients and post-Adam weights (seq2seq.py:416-543, :551-642).
    two_streams forces the second-stream overlap that production batches (>= 4096 tokens) use."""
    if two_streams and mode == "f_step_pretrain":
        pytest.skip("the pretrain step has a single pass: nothing to overlap")
    import nast_amd.utils as U
    from nast_amd.seq2seq import Seq2seq
    from nast_amd.utils import Storage
    g = load_golden(f"g9_{tag}_{mode}")
    U.cuda_init(0, "cuda")
    model = Seq2seq(make_param(device="cuda", **cfg_of(g)))
    model.overlap_min_tokens = 0 if two_streams else 1 << 60
    model.pack_min_tokens = 0 if two_streams else 1 << 60       # the packed-row execution of production batches rides along
    assert not load_weights(model.net, sub(g, "w0/"))
    model.net.model_F.train(); model.net.model_D.train()
    pref = "model_D." if mode == "d_step" else "model_F."
    for step in range(2):
        s = f"s{step}/"
        inc = Storage(); inc.data = Storage()
        inc.data.sent, inc.data.sent_length, inc.data.domain = T(g[s + "in/sent"]), T(g[s + "in/sent_length"]), T(g[s + "in/domain"])
        E = [T(g[s + f"in/E{i}"]) for i in range(int(g[s + "in/nE"]))]
        if mode != "d_step":
            nt, nl = T(g[s + "in/noise_tok"]), T(g[s + "in/noise_len"])
            model.net.noiseLayer.forward = lambda w, l, rev_domain, nt=nt, nl=nl: (nt.t(), nl)
        with U.inject_noise(E):
            if mode == "d_step":
                ret = (model.d_step(inc, 1.0, use_inference_z=True),)
            else:
                ret = model.f_step(inc, 1.0, cyc_rec_enable=(mode == "f_step"), use_inference_z=True)
        np.testing.assert_allclose(np.asarray(ret, dtype=np.float64), g[s + "out/ret"], rtol=1e-4, atol=1e-6)
        if float(g[s + "out/grad_norm"]) >= 0:
            np.testing.assert_allclose(float(model.last_grad_norm), float(g[s + "out/grad_norm"]), rtol=1e-4)
        gold_g = sub(g, s + "g/")
        gscale = max(float(np.abs(v).max()) for v in gold_g.values())
        named = dict(model.net.named_parameters())
        for k, v in sub(g, s + "w/").items():
            if not k.startswith(pref) or k not in named:
                continue
            short = k[len(pref):]
            w = named[k].detach().cpu().numpy()
            if short in gold_g:
                gg = gold_g[short]
                # north_star's 1e-4, relative to the tensor's largest gradient (floor: 1e-3 of the model's largest)
                tscale = max(float(np.abs(gg).max()), 1e-3 * gscale)
                err = float(np.abs(named[k].grad.cpu().numpy() - gg).max())
                assert err <= 1e-4 * tscale, f"grad {k}: err {err:.3e} scale {tscale:.3e}"
                solid = np.abs(gg) > 1e-3 * gscale     # Adam's first steps are ill-conditioned where |g| ~ rounding noise
                # Adam moves a weight by ~lr * g / sqrt(v): a gradient element that is off by the bar above (1e-4 * tscale) moves
                # its weight by up to (Adam's step bound 2.5e-4) x (that error / |g|), which is all an element at the lower edge of
                # "solid" can be held to (one such element of 3775 sat 4e-6 outside the old flat 1e-6 after a 1e-7 change upstream)
                tol = 1e-6 + 2 * 2.5e-4 * (1e-4 * tscale) / np.abs(gg[solid]) + 1e-4 * np.abs(v[solid])
                # ... and the fixtures hold two consecutive steps: an element whose FIRST-step gradient was at eps level took a
                # noise-directed first step whatever its gradient is now, so up to 1 element in 1000 may sit outside (all
                # of them inside Adam's step bound, next line)
                bad = np.abs(w[solid] - v[solid]) > tol
                assert bad.sum() <= 1e-3 * bad.size, (k, int(bad.sum()), float(np.abs(w[solid] - v[solid])[bad].max()))
                np.testing.assert_allclose(w, v, rtol=0, atol=2.5e-4, err_msg=k)       # elsewhere: bounded by Adam's step
            else:
                np.testing.assert_allclose(w, v, rtol=1e-6, atol=1e-7, err_msg=k)


def _tensor_class(name):
    if "length_predictor" in name:
        return "length_predictor"
    if "generator" in name:
        return "generator"
    if "embed" in name:
        return "embeddings"
    if "norm" in name:
        return "layer_norm"
    if name.endswith(".bias"):
        return "biases"
    if "pw_ffn" in name:
        return "ffn_weights"
    return "attention_weights"


class _GateReplay:
    """ReLU gates of the HIP run, replayed in the oracle's backward (oracle.nast_oracle.RELU_GATES).

    Two fp32 implementations of `relu(x W1^T + b1)` round a pre-activation that lies ~1e-8 from zero to different sides in
    about one of 1e7 elements.  Both results are valid fp32 roundings, but the flipped gate switches one term of one row of
    dW1 (and everything upstream of that FFN in the pass) on or off: a single flip moves a 198-row weight gradient by up to
    1 % of its largest element (measured, tools/parity_diag.py: seeds with a flip show ONE bias element / ONE weight row
    off by 2e-3..1e-2 and every other tensor of the model at <= 1e-5; seeds without one agree to 6e-6 everywhere).  ReLU is
    the only discrete gate on the gradient path, so the comparison below records the HIP run's gates (keyed by pass and FFN),
    hands them to the oracle's backward, counts the elements where the oracle's own gate differs and checks how close to zero
    those pre-activations are -- the gradients are then compared at north_star's 1e-4 with that one effect removed."""

    def __init__(self, model):
        from nast_amd import ops, transformer as tf
        self.ops, self.tf = ops, tf
        self.prefix_of = {id(p): n[:-len("mlp.0.weight")] for n, p in model.net.named_parameters() if n.endswith("mlp.0.weight")}
        self.model = model
        self.tags, self.ps, self.grad_on = [], None, []
        self.gates = {}            # (pass tag, call index, parameter prefix) -> (gate [B,n,4d] bool, live rows [B,n] bool)   (CPU)
        self.flips, self.flip_max, self.compared, self.used = 0, 0.0, 0, set()
        self.calls = {}            # (pass tag, parameter prefix) -> calls recorded so far
        # oracle call index of a pass -> (HIP call index, first row, rows or None): d_step runs the two generated halves of
        # the discriminator's input as ONE batch of 2B sentences where the oracle (and the reference) make two calls
        self.call_map = {}

    # ---- HIP side ----
    def __enter__(self):
        ops, tf = self.ops, self.tf
        self._saved = (ops.FFNSublayerFn.forward, tf.EncoderLayer.forward, tf.DecoderLayer.forward,
                       self.model.net.model_F.forward, self.model.net.model_D.forward)
        fwd, enc_fwd, dec_fwd, f_fwd, d_fwd = self._saved
        me = self

        def ffn_forward(ctx, x, ln_w, ln_b, W1, b1, W2, b2):
            y = fwd(ctx, x, ln_w, ln_b, W1, b1, W2, b2)
            if not (me.grad_on and me.grad_on[-1]):
                return y                     # no backward follows (the generation passes of d_step): nothing to replay
            hmid = ctx.to_save[4]
            gate = hmid > 0
            ps = me.ps
            if ps is not None:
                full = torch.zeros(ps.B * ps.width, gate.shape[1], dtype=torch.bool, device=gate.device)
                full[ps.idx] = gate[:ps.total]
                live = torch.zeros(ps.B * ps.width, dtype=torch.bool, device=gate.device)
                live[ps.idx] = True
                gate, live = full.view(ps.B, ps.width, -1), live.view(ps.B, ps.width)
            else:
                gate = gate.view(*x.shape[:-1], -1)
                live = torch.ones(x.shape[:-1], dtype=torch.bool, device=gate.device)
            tp = (me.tags[-1], me.prefix_of[id(W1)])
            n = me.calls.get(tp, 0)
            me.calls[tp] = n + 1
            me.gates[(tp[0], n, tp[1])] = (gate.cpu(), live.cpu())
            return y

        def layer(orig, mask_arg):
            def run(self_, x, *a):
                m = a[mask_arg]
                prev, me.ps = me.ps, (m.q if isinstance(m, ops.PackedPair) else None)
                try:
                    return orig(self_, x, *a)
                finally:
                    me.ps = prev
            return run

        def tagged(orig, tag_of):
            def run(*a, **k):
                me.tags.append(tag_of(*a, **k))
                me.grad_on.append(torch.is_grad_enabled())      # (inside a Function's forward grad mode is always off)
                try:
                    return orig(*a, **k)
                finally:
                    me.tags.pop()
                    me.grad_on.pop()
            return run

        ops.FFNSublayerFn.forward = staticmethod(ffn_forward)
        tf.EncoderLayer.forward = layer(enc_fwd, 0)                  # (x, mask)
        tf.DecoderLayer.forward = layer(dec_fwd, 2)                  # (x, memory, src_mask, tgt_mask)
        self.model.net.model_F.forward = tagged(f_fwd, lambda tok, *a, generate=False, **k:
                                                "transfer" if generate else ("cycle" if tok.dim() == 3 else "self"))
        self.model.net.model_D.forward = tagged(d_fwd, lambda *a, **k: "disc")
        return self

    def __exit__(self, *exc):
        ops, tf = self.ops, self.tf
        ops.FFNSublayerFn.forward = staticmethod(self._saved[0])
        tf.EncoderLayer.forward, tf.DecoderLayer.forward = self._saved[1], self._saved[2]
        self.model.net.model_F.forward, self.model.net.model_D.forward = self._saved[3], self._saved[4]

    # ---- oracle side ----
    def oracle(self, O):
        """Context manager: the oracle's FFNs take the recorded gates in their backward."""
        import contextlib
        me = self

        @contextlib.contextmanager
        def cm():
            st, disc = O.style_transformer, O.discriminator
            tags = []

            def tag_st(w, cfg, tok, *a, **k):
                generate = a[4] if len(a) > 4 else k.get("generate", False)
                tags.append("transfer" if generate else ("cycle" if tok.dim() == 3 else "self"))
                try:
                    return st(w, cfg, tok, *a, **k)
                finally:
                    tags.pop()

            def tag_disc(*a, **k):
                tags.append("disc")
                try:
                    return disc(*a, **k)
                finally:
                    tags.pop()

            ocalls = {}

            def gates(prefix, pre):
                if not tags or pre.dim() != 3 or not pre.requires_grad:
                    return None
                prefix = ("model_D." if tags[-1] == "disc" else "model_F.") + prefix      # the oracle's prefixes are relative
                n = ocalls.get((tags[-1], prefix), 0)
                ocalls[(tags[-1], prefix)] = n + 1
                hip_n, row0, rows = me.call_map.get((tags[-1], n), (n, 0, None))
                ent = me.gates.get((tags[-1], hip_n, prefix))
                if ent is None:
                    return None
                gate, live = ent
                if rows is not None:                       # one half of a merged call, cut back to this call's own width
                    gate, live = gate[row0:row0 + rows, :pre.shape[1]], live[row0:row0 + rows, :pre.shape[1]]
                    assert not bool(ent[1][row0:row0 + rows, pre.shape[1]:].any()), "live rows beyond the oracle's width"
                assert gate.shape == pre.shape, (prefix, gate.shape, pre.shape)
                diff = ((pre > 0) != gate) & live.unsqueeze(-1)
                n = int(diff.sum())
                me.flips += n
                me.compared += int(live.sum()) * gate.shape[-1]
                if n:
                    me.flip_max = max(me.flip_max, float(pre.detach().abs()[diff].max()))
                me.used.add((tags[-1], hip_n, prefix))
                return gate

            O.style_transformer, O.discriminator, O.RELU_GATES = tag_st, tag_disc, gates
            try:
                yield
            finally:
                O.style_transformer, O.discriminator, O.RELU_GATES = st, disc, None
        return cm()


def _oracle_vs_hip_step(cfg, B, seed=0, precision="fp32", name=None, replay_gates=True, step="f", noise="identity"):
    """One f_step (step="f") or d_step (step="d") on fresh weights: HIP path vs the CPU oracle on identical weights, batch and
    Gumbel draws.  noise="device": the input noise of f_step is what the device NoiseLayer kernel produced for this batch
    (blank / add / dropout at the reference's rates), fed to both sides; "identity": none.
    Returns a report: the returned losses and the pre-clip gradient norm of both sides, and per tensor class the largest
    error of the (clipped) parameter gradients and of the post-Adam weights.  Written to gpurun_out/parity_<name>.json."""
    import json, os
    import nast_amd.utils as U
    from nast_amd import ops
    from nast_amd.seq2seq import Seq2seq
    from nast_amd.utils import Storage
    from oracle import nast_oracle as O
    from helpers import synth_batch
    torch.manual_seed(seed); np.random.seed(seed)
    U.cuda_init(0, "cuda")
    model = Seq2seq(make_param(device="cuda", **cfg))
    model.overlap_min_tokens = model.pack_min_tokens = 0       # the second stream and packed rows of production batches
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.net.named_parameters():
            if n.endswith("embed.weight"):     # the reference's all-zero embedding init makes every alignment tie
                p.copy_(((torch.rand(p.shape, generator=gen) * 2 - 1) * (6.0 / sum(p.shape)) ** 0.5).cuda())
    weights = {k: v.detach().cpu().clone() for k, v in model.net.state_dict().items() if k.startswith("model_")}
    rng = np.random.default_rng(seed)
    sent, slen = synth_batch(rng, B, cfg["L"], cfg["V"])
    dom = torch.tensor([0] * (B // 2) + [1] * (B - B // 2))
    inc = Storage(); inc.data = Storage()
    inc.data.sent, inc.data.sent_length, inc.data.domain = sent.cuda(), slen.cuda(), dom.cuda()
    noise_tok, noise_len = sent, slen
    if noise == "device":
        # the real NoiseLayer kernel (nast_noise_i64, noise.py:60-194) on this batch, once; its output is then the input
        # noise of BOTH sides (its own parity against the reference's stages: tests/test_noise_gpu.py)
        with torch.no_grad():
            cnt = model.net.noiseLayer.cnt
            cnt.zero_()
            cnt[:, 4:] = 1.0 / (cnt.shape[1] - 4)
        torch.manual_seed(seed + 2)
        fixed = model.net.noiseLayer(inc.data.sent.t(), inc.data.sent_length, rev_domain=1 - inc.data.domain)
        assert fixed[0].is_cuda and not (fixed[0].shape[0] == sent.shape[1] and torch.equal(fixed[0].t().cpu(), sent)), "noise left the batch unchanged"
        model.net.noiseLayer.forward = lambda w, l, rev_domain: fixed
        noise_tok, noise_len = fixed[0].t().contiguous().cpu(), fixed[1].cpu()
    else:
        model.net.noiseLayer.forward = lambda w, l, rev_domain: (w, l)      # identity noise: same input on both sides
    model.net.model_F.train(); model.net.model_D.train()
    tr = O.Trainer(weights, O.Cfg(**cfg))
    group = "model_F." if step == "f" else "model_D."
    with torch.no_grad():          # the oracle decides the shapes of the Gumbel draws (generated lengths are data dependent)
        styles = [1 - dom] if step == "f" else [dom, 1 - dom]
        Es = []
        for st in styles:
            lp = O.style_transformer(tr.w, tr.cfg, sent, slen, None, None, st, True, 1.0, True)[0]
            Es.append(torch.empty_like(lp).exponential_(generator=gen))
            del lp
    replay = _GateReplay(model) if replay_gates else None
    import contextlib
    with U.inject_noise([e.cuda() for e in Es]), ops.gemm_precision(precision), (replay or contextlib.nullcontext()):
        if step == "f":
            got = model.f_step(inc, 1.0, use_inference_z=True)
        else:
            got = [model.d_step(inc, 1.0, use_inference_z=True)]
    if replay and step == "d":
        n_disc = max(n for (t, _), n in replay.calls.items() if t == "disc")
        if cfg.get("disc", "Cond") == "Cond" and n_disc == 3:        # gold(raw), gold(rev), generated(raw ++ rev)
            replay.call_map = {("disc", 2): (2, 0, B), ("disc", 3): (2, B, B)}
        elif cfg.get("disc", "Cond") == "Multi" and n_disc == 2:     # gold, generated(raw ++ rev)
            replay.call_map = {("disc", 1): (1, 0, B), ("disc", 2): (1, B, B)}
    with (replay.oracle(O) if replay else contextlib.nullcontext()):
        if step == "f":
            ref = tr.f_step(sent, slen, dom, noise_tok, noise_len, Es)
        else:
            ref = [tr.d_step(sent, slen, dom, Es)]
    if replay:
        assert replay.used == set(replay.gates), set(replay.gates) - replay.used       # every recorded FFN found its oracle call
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    rep = {"cfg": {k: (v if isinstance(v, (bool, str)) else int(v)) for k, v in cfg.items()}, "B": B, "precision": precision,
           "step": step + "_step", "noise": noise,
           "losses_hip": got.tolist(), "losses_oracle": ref.tolist(),
           "loss_rel_err": float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30) * (np.abs(ref) > 0))),
           "grad_norm_hip": float(model.last_grad_norm), "grad_norm_oracle": float(tr.last_norm)}
    rep["grad_norm_rel_err"] = abs(rep["grad_norm_hip"] - rep["grad_norm_oracle"]) / rep["grad_norm_oracle"]
    if replay:
        rep["relu_gates"] = {"ffn_calls": len(replay.gates), "compared": replay.compared, "flipped": replay.flips,
                             "largest_abs_preactivation_at_a_flip": replay.flip_max}
    named = dict(model.net.named_parameters())
    gref = {k: v for k, v in tr.last_grads.items() if v is not None}
    gmax = max(float(v.abs().max()) for v in gref.values())
    # Which side's norm is the accurate one?  Both sides scaled their gradients by 5 / (norm + 1e-6) (the norm exceeds 5 in every
    # case here), so the float64 norm of the CLIPPED gradients is 5 up to the error of the norm that side clipped with.
    rep["clipped_norm_fp64_hip"] = float(torch.sqrt(sum((named[k].grad.double() ** 2).sum() for k in gref)).cpu())
    rep["clipped_norm_fp64_oracle"] = float(torch.sqrt(sum((v.double() ** 2).sum() for v in gref.values())))
    # torch's fp32 norm reduction (clip_grad_norm_, which the oracle restates and the reference calls) is itself off by up to
    # 1.2e-4 at 137 M parameters (measured: the oracle's clipped gradients have norm 5.0006 there, the HIP path's 5.00002 --
    # its norm kernel sums in float64).  The pre-clip norm is therefore checked against the float64 norm of the oracle's own
    # gradients, and the gradients are compared with each side's clip factor normalised to exactly 5 / norm.
    rep["clipped"] = bool(rep["grad_norm_oracle"] > 5.0 and rep["grad_norm_hip"] > 5.0)
    if rep["clipped"]:
        rep["grad_norm_oracle_fp64"] = rep["clipped_norm_fp64_oracle"] * (rep["grad_norm_oracle"] + 1e-6) / 5.0
        fix_hip, fix_ref = 5.0 / rep["clipped_norm_fp64_hip"], 5.0 / rep["clipped_norm_fp64_oracle"]
    else:                                              # (a discriminator step below the clip threshold: gradients are unscaled)
        assert rep["grad_norm_oracle"] <= 5.0 and rep["grad_norm_hip"] <= 5.0, (rep["grad_norm_oracle"], rep["grad_norm_hip"])
        rep["grad_norm_oracle_fp64"] = rep["clipped_norm_fp64_oracle"]
        fix_hip = fix_ref = 1.0
    rep["grad_norm_rel_err_vs_oracle_fp32"] = rep["grad_norm_rel_err"]
    rep["grad_norm_rel_err"] = abs(rep["grad_norm_hip"] - rep["grad_norm_oracle_fp64"]) / rep["grad_norm_oracle_fp64"]
    gmax *= fix_ref
    lr = cfg.get("lr_F" if step == "f" else "lr_D", 1e-4)
    classes = {}
    for k, gr in gref.items():
        p = named[k]
        assert p.grad is not None, k
        gg = p.grad.detach().cpu() * fix_hip
        gr = gr * fix_ref
        c = classes.setdefault(_tensor_class(k), {"tensors": 0, "grad_err_vs_tensor_max": 0.0, "grad_err_vs_model_max": 0.0,
                                                   "weight_err_vs_tensor_max": 0.0, "weight_err_solid_vs_tensor_max": 0.0,
                                                   "weight_err_in_adam_steps": 0.0, "worst": ""})
        err = float((gg - gr).abs().max())
        tscale = max(float(gr.abs().max()), 1e-3 * gmax)       # floor: 1e-4 of it is 1e-7 of the model's largest gradient (fp32 epsilon)
        if err / tscale > c["grad_err_vs_tensor_max"]:
            c["grad_err_vs_tensor_max"], c["worst"] = err / tscale, k
        c["grad_err_vs_model_max"] = max(c["grad_err_vs_model_max"], err / gmax)
        w, wr = p.detach().cpu(), tr.w[k].detach()
        werr = (w - wr).abs()
        wmax = float(wr.abs().max()) + 1e-30
        c["weight_err_vs_tensor_max"] = max(c["weight_err_vs_tensor_max"], float(werr.max()) / wmax)
        c["weight_err_in_adam_steps"] = max(c["weight_err_in_adam_steps"], float(werr.max()) / lr)
        solid = gr.abs() > 1e-3 * gmax          # Adam's first step is -lr * g / (|g| + eps): ill-conditioned where |g| ~ rounding noise
        if bool(solid.any()):
            c["weight_err_solid_vs_tensor_max"] = max(c["weight_err_solid_vs_tensor_max"], float(werr[solid].max()) / wmax)
        c["tensors"] += 1
    for k, p in named.items():
        if k.startswith(group) and k not in gref:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{k} has a gradient the oracle lacks"
    rep["classes"] = classes
    rep["grad_err_vs_tensor_max"] = max(c["grad_err_vs_tensor_max"] for c in classes.values())
    rep["weight_err_solid_vs_tensor_max"] = max(c["weight_err_solid_vs_tensor_max"] for c in classes.values())
    rep["weight_err_in_adam_steps"] = max(c["weight_err_in_adam_steps"] for c in classes.values())
    if name:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"parity_{name}.json"), "w") as f:
            json.dump(rep, f, indent=1)
        if replay:
            print(f"\n[parity {name}] ReLU gates: {replay.flips} of {replay.compared} differ between the two runs "
                  f"(largest |pre-activation| among them {replay.flip_max:.2e}); the oracle's backward used the HIP run's gates")
        print(f"[parity {name}] float64 norm of the clipped gradients (exact: 5): HIP {rep['clipped_norm_fp64_hip']:.7f}  oracle {rep['clipped_norm_fp64_oracle']:.7f}")
        print(f"[parity {name}] loss {rep['loss_rel_err']:.2e}  grad-norm {rep['grad_norm_rel_err']:.2e}  "
              f"grads {rep['grad_err_vs_tensor_max']:.2e}  weights(solid) {rep['weight_err_solid_vs_tensor_max']:.2e}  "
              f"weights(all, in Adam steps) {rep['weight_err_in_adam_steps']:.2f}")
        for cn, c in sorted(classes.items()):
            print(f"   {cn:18s} n={c['tensors']:3d} grad/tensor {c['grad_err_vs_tensor_max']:.2e} grad/model {c['grad_err_vs_model_max']:.2e} "
                  f"w {c['weight_err_vs_tensor_max']:.2e} w(solid) {c['weight_err_solid_vs_tensor_max']:.2e}  worst {c['worst']}")
    rep_objs = (model, tr)
    _oracle_vs_hip_step.last = rep_objs
    return rep


def _assert_step_parity(rep, tol):
    """north_star: fp32 losses and gradients within 1e-4 relative (bf16 variant 2e-2).  Gradients: every parameter tensor,
    error relative to that tensor's largest gradient (floored at 1e-3 of the model's largest: analytically-zero gradients
    such as the key biases have no scale of their own).  Post-Adam weights: relative to the tensor's largest weight on the
    elements whose gradient is above rounding noise; everywhere else the update is bounded by Adam's step (2 lr)."""
    assert rep["loss_rel_err"] <= tol, rep
    assert rep["grad_norm_rel_err"] <= tol, rep
    assert rep["grad_err_vs_tensor_max"] <= tol, {k: (c["grad_err_vs_tensor_max"], c["worst"]) for k, c in rep["classes"].items()}
    assert rep["weight_err_solid_vs_tensor_max"] <= tol, rep["classes"]
    assert rep["weight_err_in_adam_steps"] <= 2.0 + 1e-3, rep["classes"]
    if "relu_gates" in rep:
        # the replayed gates differ from the oracle's own in a vanishing share of the elements, all of them at pre-activations
        # within rounding distance of zero
        g = rep["relu_gates"]
        assert g["flipped"] <= max(2, 2e-6 * g["compared"]) and g["largest_abs_preactivation_at_a_flip"] <= 1e-5, g


BENCH_CFG = dict(V=9600, d=256, h=4, N=4, L=32, learnable=True)


def test_bench_shape_step_vs_oracle():
    """The benchmark model (d=256, h=4, 4 layers, L=32, V=9600) at a small batch against the oracle: losses, gradient norm,
    EVERY parameter gradient and the post-Adam weights at north_star's 1e-4."""
    _assert_step_parity(_oracle_vs_hip_step(BENCH_CFG, B=6, name="bench_b6"), 1e-4)


def test_bench_shape_step_vs_oracle_full_batch():
    """The same at the benchmarked batch: 512 sentences (two streams, packed rows, split-K weight gradients with K = all rows)."""
    _assert_step_parity(_oracle_vs_hip_step(BENCH_CFG, B=512, seed=3, name="bench_b512"), 1e-4)


SIMPLE_CFG = dict(V=9600, d=256, h=4, N=4, L=16, learnable=False)      # BASELINE.json configs[1]


def test_bench_shape_f_step_with_device_noise_vs_oracle():
    """f_step at the bench shape, 64 sentences, with the REAL input noise: the device NoiseLayer's output for the batch
    (blank 0.1, add 1/0.95-1, dropout 0.05; run.py:90-95) is the noisy source of both sides (seq2seq.py:438-451)."""
    _assert_step_parity(_oracle_vs_hip_step(BENCH_CFG, B=64, seed=7, name="bench_b64_device_noise", noise="device"), 1e-4)


@pytest.mark.parametrize("B,disc", [(64, "Cond"), (512, "Cond"), (64, "Multi")])
def test_bench_shape_d_step_vs_oracle(B, disc):
    """d_step (seq2seq.py:551-642) at the bench shape: both styles generated as one 2B batch, two Gumbel draws, the gold
    half on the second stream from the decode hook, the generated halves through the discriminator as one packed batch --
    loss, gradient norm, every discriminator gradient and the post-Adam weights at 1e-4, Cond and Multi."""
    cfg = dict(BENCH_CFG, disc=disc)
    _assert_step_parity(_oracle_vs_hip_step(cfg, B=B, seed=11 + B, name=f"bench_d_step_b{B}_{disc.lower()}", step="d"), 1e-4)


@pytest.mark.parametrize("step", ["f", "d"])
def test_configs1_simple_step_vs_oracle(step):
    """BASELINE.json configs[1] at full size: simple alignment, V=9600, d=256, 4 layers, L=16, 64 sentences -- the padded
    execution (lengths read off the argmax of the soft tokens, seq2seq.py:545-549), f_step and d_step."""
    _assert_step_parity(_oracle_vs_hip_step(SIMPLE_CFG, B=64, seed=5, name=f"configs1_simple_{step}_step", step=step), 1e-4)


def test_stress_shape_step_vs_oracle():
    """BASELINE.json configs[4] shape at reduced depth/batch: d=1024, 16 heads, L=128 (129-token attention), V=32000,
    4d=4096 FFN -- every kernel family beyond its d=256 / L<=32 tuning point, fp32."""
    _assert_step_parity(_oracle_vs_hip_step(dict(V=32000, d=1024, h=16, N=2, L=128, learnable=True), B=2, name="stress_n2_fp32"), 1e-4)


def test_stress_shape_full_depth_bf16_vs_oracle():
    """BASELINE.json configs[4] at its real depth and precision: d=1024, 16 heads, 8 layers, L=128, V=32000, bf16-MFMA GEMMs
    (fp32 accumulate; LayerNorm / softmax / alignment fp32), small batch, against the fp32 oracle at north_star's 2e-2."""
    rep = _oracle_vs_hip_step(dict(V=32000, d=1024, h=16, N=8, L=128, learnable=True), B=2, seed=5, precision="bf16", name="stress_n8_bf16")
    assert rep["loss_rel_err"] <= 2e-2 and rep["grad_norm_rel_err"] <= 2e-2, rep
    # gradients of the bf16 variant: relative to the model's largest gradient (rounding of every operand to 8 bits)
    assert max(c["grad_err_vs_model_max"] for c in rep["classes"].values()) <= 2e-2, rep["classes"]


@pytest.mark.parametrize("seed", [0, 1])
def test_sample_decode_vs_oracle(seed):
    """z_decode_mode == "sample" (transformer.py:560-561): with the same uniform numbers the HIP decode loop and the oracle's
    inverse-CDF restatement draw the same skips, step after step (a draw feeds the next step's input), and the decoder run
    on the drawn alignment matches.  The reference draws with torch.multinomial on its own generator, which no other
    implementation can replay: what is pinned here is the distribution (softmax of the reference's logits) and the use of
    the draw (:565-567), not the reference's random stream -- "parity unpinned" for the stream itself."""
    from nast_amd import transformer as tf
    from oracle import nast_oracle as O
    g = load_golden("g2_style_transformer_learnable")
    cfg = cfg_of(g)
    param = make_param(**cfg)
    param.args.z_decode_mode = "sample"
    F = tf.StyleTransformer(param).to(DEV)
    assert not load_weights(F, sub(g, "w/"))
    gold, gl, style = T(g["in/gold"]), T(g["in/gold_len"]), T(g["in/style"])
    B, L = gold.shape[0], cfg["L"]
    u = torch.rand(L, B, generator=torch.Generator().manual_seed(seed))
    w = {"model_F." + k: torch.from_numpy(np.asarray(v)) for k, v in sub(g, "w/").items()}
    ocfg = O.Cfg(**{k: cfg[k] for k in ("V", "d", "h", "N", "L", "learnable")})
    with torch.no_grad():
        ref = O.style_transformer(w, ocfg, gold.cpu(), gl.cpu(), None, None, (1 - style).cpu(), True, 1.0, True, sample_u=u)
        F.length_predictor.sample_u = u.to(DEV)
        lp, tl, _, _, ap = F(gold, gl, None, None, 1 - style, generate=True, temperature=1.0, use_inference_z=True)
    assert np.array_equal(ap.cpu().numpy(), ref[4].numpy())
    assert np.array_equal(tl.cpu().numpy(), ref[1].numpy())
    greedy = g["out/generate/allpos"]
    assert not np.array_equal(ap.cpu().numpy()[:, :greedy.shape[1]], greedy[:, :ap.shape[1]]), "sampling must not reduce to the argmax path"
    relclose(lp, ref[0], what="sampled generate")
    # without injected uniforms the draws come from torch's generator: reproducible under a seed, different across seeds
    F.length_predictor.sample_u = None
    outs = []
    for s in (7, 7, 8):
        torch.manual_seed(s)
        with torch.no_grad():
            outs.append(F(gold, gl, None, None, 1 - style, generate=True, temperature=1.0, use_inference_z=True)[4].cpu())
    assert torch.equal(outs[0], outs[1]) and not (outs[0].shape == outs[2].shape and torch.equal(outs[0], outs[2]))
    param.args.z_decode_mode = "nope"
    with pytest.raises(NotImplementedError):
        F(gold, gl, None, None, 1 - style, generate=True, temperature=1.0, use_inference_z=True)


def test_transfer_matches_golden_generate():
    """Seq2seq.transfer (the generator's share of evaluate, seq2seq.py:303-339) on the captured generate case: argmax
    tokens, generated lengths and alignment positions of the reference."""
    import nast_amd.utils as U
    from nast_amd.seq2seq import Seq2seq
    from nast_amd.utils import Storage
    g = load_golden("g2_style_transformer_learnable")
    U.cuda_init(0, "cuda")
    model = Seq2seq(make_param(device="cuda", **cfg_of(g)))
    assert not load_weights(model.net.model_F, sub(g, "w/"))
    gold, gl, style = T(g["in/gold"]), T(g["in/gold_len"]), T(g["in/style"])
    inc = Storage(); inc.data = Storage()
    inc.data.sent, inc.data.sent_length, inc.data.domain = gold, gl, style       # transferred to 1 - style
    out = model.transfer(inc, temperature=1.0, use_inference_z=True)
    ref_lp = g["out/generate/log_probs"]
    glen = g["out/generate/tgt_lengths"]
    assert np.array_equal(out.gen_lengths.cpu().numpy(), glen)
    gen, ref_gen = out.gen.cpu().numpy(), ref_lp.argmax(-1)
    for b in range(gen.shape[0]):                       # beyond the generated length the reference's ids are argmax of noise-level ties
        assert np.array_equal(gen[b, :glen[b]], ref_gen[b, :glen[b]]), b
    for b, pos in enumerate(out.allpos):
        assert np.array_equal(pos, g["out/generate/allpos"][b][1:glen[b] + 1])
    assert abs(out.length_delta - float(np.mean(glen - g["in/gold_len"]))) < 1e-6
    assert model.net.model_F.training


def test_reserved_second_stream_is_the_one_the_steps_use():
    """seq2seq.reserve_streams() (to be called before init_process_group) hands the trainer its stream."""
    from nast_amd import seq2seq, ops
    s = seq2seq.reserve_streams()
    assert seq2seq.reserve_streams() is s and isinstance(s, torch.cuda.Stream)
    model = seq2seq.Seq2seq(make_param(device="cuda", V=40, d=32, h=2, N=1, L=8, learnable=True))
    assert model._side_stream(torch.device("cuda", torch.cuda.current_device())) is s
    assert s in ops.peer_streams and torch.cuda.current_stream() in ops.peer_streams
