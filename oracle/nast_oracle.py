"""CPU oracle for the NAST-StyTrans generator training step  --  TEST INFRASTRUCTURE ONLY.

A functional, plain-PyTorch (CPU, fp32) restatement of the reference algorithm for the hot path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product (nast_amd/) never does -- it runs hand-written HIP kernels and fails loudly without them.

Pinned against the reference: every function here is checked in tests/test_oracle_golden.py
against fixtures produced by importing the unmodified reference (oracle/gen_golden.py ->
tests/golden/*.npz).

Weights are a flat dict {state_dict key: tensor} using the reference's state_dict names
(SURVEY.md section 8b), so a reference checkpoint feeds this oracle directly.

Reference citations are `file:line` under /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

MAX_SKIP = 4          # transformer.py:388
BIG = 1e9             # sentinel used by the alignment DP, transformer.py:414-434


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
class Cfg:
    """The `param.args` / `param.volatile.dm` keys the hot path reads."""

    def __init__(self, **kw):
        self.V = kw["V"]; self.d = kw["d"]; self.h = kw["h"]; self.N = kw["N"]; self.L = kw["L"]
        self.learnable = bool(kw.get("learnable", False))
        self.disc = kw.get("disc", "Cond")
        self.decoder_input = kw.get("decoder_input", "encoded")
        self.pad_id, self.eos_id, self.unk_id = 0, 2, 3
        self.slf_factor = kw.get("slf_factor", 0.5); self.cyc_factor = kw.get("cyc_factor", 0.5)
        self.length_factor = kw.get("length_factor", 1.0)
        self.fw_adv_factor = kw.get("fw_adv_factor", 1.5); self.bw_adv_factor = kw.get("bw_adv_factor", 1.0)
        self.lr_F = kw.get("lr_F", 1e-4); self.lr_D = kw.get("lr_D", 1e-4)


# ----------------------------------------------------------------------------------------------
# layers  (transformer.py:233-350)
# ----------------------------------------------------------------------------------------------
def _ln(w, p, x):
    return F.layer_norm(x, (x.shape[-1],), w[p + "weight"], w[p + "bias"], 1e-6)   # :348-350


def _lin(w, p, x):
    return F.linear(x, w[p + "weight"], w[p + "bias"])


def attention(w, p, q_in, kv_in, mask, h):
    """MultiHeadAttention.forward + scaled_attention (transformer.py:284-303)."""
    B, nq, d = q_in.shape
    dk = d // h
    q = _lin(w, p + "head_projs.0.", q_in).view(B, -1, h, dk).transpose(1, 2)
    k = _lin(w, p + "head_projs.1.", kv_in).view(B, -1, h, dk).transpose(1, 2)
    v = _lin(w, p + "head_projs.2.", kv_in).view(B, -1, h, dk).transpose(1, 2)
    s = q.matmul(k.transpose(-2, -1)) / math.sqrt(dk)
    s = s.masked_fill(mask, float("-inf"))
    a = torch.softmax(s, -1).matmul(v)
    a = a.transpose(1, 2).contiguous().view(B, nq, d)
    return _lin(w, p + "fc.", a)


# Test hook (tests/test_model_gpu.py, gate replay): None, or a callable (parameter prefix, pre-activation [B,n,4d]) -> bool
# mask [B,n,4d] or None.  With a mask, the ReLU's BACKWARD uses it as the gate instead of (pre-activation > 0); the forward
# is the plain ReLU.  Two fp32 implementations round a pre-activation that is ~1e-8 from zero to different sides in about one
# of 1e7 elements; such a flipped gate changes one row of one weight gradient by percents although both results are valid
# roundings.  Replaying the other side's gates removes that one discrete effect from a gradient comparison.
RELU_GATES = None


class _GateReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        ctx.save_for_backward(gate)
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        return g * gate.to(g.dtype), None


def ffn(w, p, x):
    """PositionwiseFeedForward (transformer.py:305-316): Linear, ReLU, Dropout(0), Linear."""
    pre = _lin(w, p + "mlp.0.", x)
    gate = RELU_GATES(p, pre) if (RELU_GATES is not None and torch.is_grad_enabled()) else None
    act = torch.relu(pre) if gate is None else _GateReLU.apply(pre, gate)
    return _lin(w, p + "mlp.3.", act)


def encoder_layer(w, p, x, mask, h):
    """EncoderLayer.forward (transformer.py:240-242), pre-LN residual blocks (:324-326)."""
    x = x + attention(w, p + "self_attn.", _ln(w, p + "sublayer.0.layer_norm.", x), _ln(w, p + "sublayer.0.layer_norm.", x), mask, h)
    return x + ffn(w, p + "pw_ffn.", _ln(w, p + "sublayer.1.layer_norm.", x))


def decoder_layer(w, p, x, mem, src_mask, tgt_mask, h):
    """DecoderLayer.forward (transformer.py:253-257); self-attention mask is whatever the caller passes."""
    z = _ln(w, p + "sublayer.0.layer_norm.", x)
    x = x + attention(w, p + "self_attn.", z, z, tgt_mask, h)
    x = x + attention(w, p + "src_attn.", _ln(w, p + "sublayer.1.layer_norm.", x), mem, src_mask, h)
    return x + ffn(w, p + "pw_ffn.", _ln(w, p + "sublayer.2.layer_norm.", x))


def encoder(w, p, x, mask, h, N):
    """Encoder.forward (transformer.py:163-171)."""
    for i in range(N):
        x = encoder_layer(w, f"{p}layers.{i}.", x, mask, h)
    return _ln(w, p + "norm.", x)


def decoder_stack(w, p, x, mem, src_mask, tgt_mask, h, N):
    for i in range(N):
        x = decoder_layer(w, f"{p}layers.{i}.", x, mem, src_mask, tgt_mask, h)
    return x


def generator(w, p, x, temperature):
    """Generator.forward (transformer.py:213-214)."""
    return torch.log_softmax(_lin(w, p + "proj.", x) / temperature, -1)


def embed(w, p, x, pos):
    """EmbeddingLayer.forward (transformer.py:225-231): id gather or soft-token matmul, + positions."""
    pe = w[p + "pos_embed.weight"][pos]
    if x.dim() == 2:
        return w[p + "token_embed.weight"][x] + pe
    return torch.matmul(x, w[p + "token_embed.weight"]) + pe


def _pad_mask(lengths, width, extra):
    """[B,1,1,extra+width] bool, True = masked; `extra` leading columns are never masked
    (transformer.py:56-61, :135-139, :459-461)."""
    B = lengths.shape[0]
    m = torch.arange(width, device=lengths.device)[None, :] >= lengths[:, None]
    m = torch.cat([torch.zeros(B, extra, dtype=torch.bool, device=lengths.device), m], 1)
    return m.view(B, 1, 1, extra + width)


# ----------------------------------------------------------------------------------------------
# alignment DP  (LengthPredictor.match, transformer.py:394-449)
# ----------------------------------------------------------------------------------------------
def match_costs(w, embed_prefix, origin_token, target_token, unk_id):
    """cost0 [B,T], cost [B,T,S] as transformer.py:396-412 builds them."""
    E = w[embed_prefix + "token_embed.weight"]
    if origin_token.dim() == 2:
        o = origin_token.clone()
        o[o == unk_id] = 0
        oe = E[o]
    else:
        o = origin_token.clone()
        o[:, :, 0] += o[:, :, unk_id]
        o[:, :, unk_id] *= 0
        oe = torch.matmul(o, E)
    te = E[target_token]
    cost0 = (te ** 2).sum(-1) + 1e-6
    cost = torch.cdist(te, oe) ** 2
    return cost0, cost


def viterbi_align(cost0, cost, origin_length, target_length):
    """Min-plus DP with skips {0,1,2,3} + backtracking (transformer.py:413-448), numpy restatement.

    f_0[s] = cost[0,s] for s<3 else 1e9;  f_i[s] = min_j ( f_{i-1}[s-j] + c_j ),
    c_0 = cost0[i], c_{1..3} = cost[i,s]; first minimal j wins; rows i >= target_length are frozen;
    final state restricted to s >= origin_length-3 (first minimal s wins); positions are shifted
    by +1 and a leading 0 is prepended for the style token."""
    cost0 = np.asarray(cost0, dtype=np.float32)
    cost = np.asarray(cost, dtype=np.float32).copy()
    olen = np.asarray(origin_length, dtype=np.int64)
    tlen = np.asarray(target_length, dtype=np.int64)
    B, T, S = cost.shape
    big = np.float32(BIG)
    s_idx = np.arange(S)[None, :]
    cost[np.broadcast_to((s_idx >= olen[:, None])[:, None, :], cost.shape)] = big      # :413-414
    f = np.full((B, S), big, dtype=np.float32)
    f[:, :MAX_SKIP - 1] = cost[:, 0, :MAX_SKIP - 1]                                      # :415
    back = np.zeros((B, max(T - 1, 0), S), dtype=np.int64)
    for i in range(1, T):
        fp = np.concatenate([np.full((B, MAX_SKIP), big, dtype=np.float32), f], 1)       # :424
        cand = np.stack([fp[:, MAX_SKIP - j: MAX_SKIP - j + S] for j in range(MAX_SKIP)], -1)
        cnow = np.concatenate([np.broadcast_to(cost0[:, i, None, None], (B, S, 1)),
                               np.broadcast_to(cost[:, i, :, None], (B, S, MAX_SKIP - 1))], -1)
        tot = (cand + cnow).astype(np.float32)
        j = tot.argmin(-1)                                                                # first min (:427)
        newf = np.take_along_axis(tot, j[..., None], -1)[..., 0]
        frozen = (i >= tlen)[:, None]
        f = np.where(frozen, f, newf).astype(np.float32)                                  # :429
        back[:, i - 1] = np.where(frozen, 0, j)                                           # :430
    f = np.where(s_idx >= (olen[:, None] - MAX_SKIP + 1), f, big)                         # :433-434
    last = f.argmin(-1)                                                                   # :435
    track = [last]
    for i in range(T - 1, 0, -1):                                                         # :439-442
        last = last - back[np.arange(B), i - 1, last]
        track.append(last)
    allpos = np.stack(track[::-1], -1)                                                    # [B,T]
    t_idx = np.arange(T)[None, :]
    allpos = np.where(t_idx >= tlen[:, None], olen[:, None], allpos)                      # :445
    return np.concatenate([np.zeros((B, 1), dtype=np.int64), allpos + 1], 1)              # :448


def match(w, embed_prefix, origin_token, origin_length, target_token, target_length, unk_id=3):
    with torch.no_grad():
        cost0, cost = match_costs(w, embed_prefix, origin_token, target_token, unk_id)
        ap = viterbi_align(cost0.numpy(), cost.numpy(), origin_length.numpy(), target_length.numpy())
    return torch.from_numpy(ap)


# ----------------------------------------------------------------------------------------------
# LengthPredictor  (transformer.py:452-582)
# ----------------------------------------------------------------------------------------------
def _lp_candidates(w, cfg, lp, ep, origin_input, origin_length):
    B, n, d = origin_input.shape
    S = n - 1
    padded = _pad_mask(origin_length, S, 1)[:, 0, 0, :, None].float()               # :466
    eos = w[ep + "token_embed.weight"][cfg.eos_id]
    unk = w[ep + "token_embed.weight"][cfg.unk_id]
    cand_in = origin_input * (1 - padded) + eos * padded                            # :471
    cand_in = torch.cat([cand_in, eos.view(1, 1, d).expand(B, MAX_SKIP + 1, d)], 1)  # :472-473  [B,S+6,d]
    cand_out = torch.stack([unk.view(1, 1, d).expand(B, S + 2, d)] +
                           [cand_in[:, i:S + 2 + i] for i in range(1, MAX_SKIP)], -2)  # :476-477
    return cand_in, cand_out, eos, unk


def _gather_dec_input(cand_in, unk, allpos):
    prev = torch.cat([torch.full_like(allpos[:, :1], -1), allpos[:, :-1]], 1)
    stay = (allpos == prev).float().unsqueeze(-1)                                    # :488 / :576
    d = cand_in.shape[-1]
    picked = torch.gather(cand_in, 1, allpos.unsqueeze(-1).expand(-1, -1, d))
    return picked * (1 - stay) + unk.view(1, 1, d) * stay                             # :490-491


def length_predictor_teacher(w, cfg, origin_input, origin_length, origin_token, target_token, target_length,
                             use_inference_z=True, lp="length_predictor.", ep="embed."):
    """Teacher-forced branch (transformer.py:479-519). Returns dec_input, target_length, length_loss, gen_err, allpos."""
    B, n, d = origin_input.shape
    S = n - 1
    if not use_inference_z:                                                           # :480-484
        z = torch.zeros((), dtype=torch.float)
        return origin_input, target_length, z, z, None
    src_mask = _pad_mask(origin_length, S, 1)
    cand_in, cand_out, eos, unk = _lp_candidates(w, cfg, lp, ep, origin_input, origin_length)
    allpos = match(w, ep, origin_token.detach(), origin_length, target_token, target_length, cfg.unk_id)
    T = target_token.shape[1]
    dec_input = _gather_dec_input(cand_in, unk, allpos)
    # pos_embed has padding_idx=0 (transformer.py:385): row 0 is used in forward but gets no gradient
    x = dec_input + F.embedding(torch.arange(T + 1), w[lp + "pos_embed.weight"], padding_idx=0)[None]   # :493
    causal = (torch.ones(T + 1, T + 1).tril() == 0).view(1, 1, T + 1, T + 1)          # :463-464
    hidden = decoder_stack(w, lp + "decoder.", x, origin_input, src_mask, causal, cfg.h, 3)   # no final LN (:359-364)
    sel = torch.gather(cand_out, 1, allpos.view(B, T + 1, 1, 1).expand(-1, -1, MAX_SKIP, d))   # :504-505
    logits = torch.einsum("bld,bltd->blt", _lin(w, lp + "wq.", hidden), _lin(w, lp + "wk.", sel)) / np.sqrt(d)
    logits = logits + _lin(w, lp + "linear.", hidden)                                 # :506-507
    target = torch.cat([allpos[:, 1:], origin_length.unsqueeze(-1) + 1], 1) - allpos  # :509
    gen_err = (target >= 4).sum().float() / B                                         # :511
    target = target.clamp(max=3)                                                      # :512
    loss = F.cross_entropy(logits.transpose(1, 2), target, reduction="none")          # :517
    return dec_input, target_length, loss, gen_err, allpos


def hidden_decoder_incremental(w, p, x_steps, mem, src_mask, h, nl=3):
    """HiddenDecoder.incremental_forward unrolled over all steps (transformer.py:366-377, :259-272):
    position k attends to layer *inputs* 0..k (equivalent to a K/V cache) and the output gets the
    final LayerNorm, unlike HiddenDecoder.forward."""
    cache, outs = _IncState(nl), []
    for k in range(x_steps.shape[1]):
        x = x_steps[:, k:k + 1]
        for li in range(nl):
            x = _dec_layer_step(w, f"{p}layers.{li}.", x, cache, li, mem, src_mask, h)
        outs.append(_ln(w, p + "norm.", x))
    return torch.cat(outs, 1)


class _IncState:
    """Per-layer caches of LN'ed layer inputs (what the reference recomputes from prev_states)."""

    def __init__(self, nl):
        self.self_in = [None] * nl


def _dec_layer_step(w, p, x_new, cache, li, mem, src_mask, h):
    z_new = _ln(w, p + "sublayer.0.layer_norm.", x_new)
    cache.self_in[li] = z_new if cache.self_in[li] is None else torch.cat([cache.self_in[li], z_new], 1)
    zs = cache.self_in[li]
    nomask = torch.zeros(1, 1, 1, zs.shape[1], dtype=torch.bool)
    x = x_new + attention(w, p + "self_attn.", z_new, zs, nomask, h)                 # :265
    x = x + attention(w, p + "src_attn.", _ln(w, p + "sublayer.1.layer_norm.", x), mem, src_mask, h)   # :268
    return x + ffn(w, p + "pw_ffn.", _ln(w, p + "sublayer.2.layer_norm.", x))        # :271


def sample_skip(logits, u):
    """z_decode_mode == "sample" (transformer.py:560-561: torch.multinomial(logits.softmax(-1), 1)) with the randomness
    made an input: u [B] uniform in [0,1); the draw is the inverse CDF, delta = first j with p_0+..+p_j > u (a j with
    p_j == 0 is never drawn; the last positive j if rounding leaves none)."""
    p = torch.softmax(logits.double(), -1)
    c = torch.cumsum(p, -1)
    ok = (c > u.double().unsqueeze(-1)) & (p > 0)
    first = torch.where(ok.any(-1), ok.float().argmax(-1), torch.full_like(u, -1, dtype=torch.long))
    last = (p > 0).float().cumsum(-1).argmax(-1)
    return torch.where(first >= 0, first, last)


def length_predictor_generate(w, cfg, origin_input, origin_length, use_inference_z=True,
                              lp="length_predictor.", ep="embed.", sample_u=None):
    """Skip decoding (transformer.py:520-582): z_decode_mode == "max" (sample_u None), or "sample" with the uniforms
    sample_u [L,B] that `sample_skip` turns into draws.  Returns dec_input[:, :maxT+1], target_length, allpos."""
    if not use_inference_z:
        raise ValueError("reference raises here as well (transformer.py:78 vs :522)")
    B, n, d = origin_input.shape
    S = n - 1
    src_mask = _pad_mask(origin_length, S, 1)
    padded = src_mask[:, 0, 0, :].float()
    cand_in, cand_out, eos, unk = _lp_candidates(w, cfg, lp, ep, origin_input, origin_length)
    with torch.no_grad():
        now = torch.zeros(B, dtype=torch.long)
        stay = torch.zeros(B, 1, 1)
        cols = [now]
        cache = _IncState(3)
        for k in range(cfg.L):
            nxt = torch.gather(cand_in, 1, now.view(B, 1, 1).expand(-1, -1, d))
            nxt = nxt * (1 - stay) + unk.view(1, 1, d) * stay                          # :535-537
            x = nxt + w[lp + "pos_embed.weight"][k].view(1, 1, d)
            for li in range(3):
                x = _dec_layer_step(w, f"{lp}decoder.layers.{li}.", x, cache, li, origin_input, src_mask, cfg.h)
            hid = _ln(w, lp + "decoder.norm.", x)[:, 0]                                # :376 (final LN)
            cand = torch.gather(cand_out, 1, now.view(B, 1, 1, 1).expand(-1, 1, MAX_SKIP, d))[:, 0]
            logits = torch.einsum("bd,bjd->bj", _lin(w, lp + "wq.", hid), _lin(w, lp + "wk.", cand)) / np.sqrt(d)
            logits = logits + _lin(w, lp + "linear.", hid)                             # :551-552
            if k == 0:
                logits[:, 0] = -1e9
                logits = logits - padded[:, :4] * 1e9                                  # :553-555
            delta = logits.max(-1)[1] if sample_u is None else sample_skip(logits, sample_u[k])   # :559 first max / :561
            now = torch.clamp(now + delta, max=S + 1)                                  # :565
            stay = (delta == 0).view(B, 1, 1).float()
            cols.append(now)
            if k % 5 == 0 and int((now >= origin_length + 1).sum()) == B:              # :569
                break
    allpos = torch.stack(cols, 1)
    dec_input = _gather_dec_input(cand_in, unk, allpos)                                # :576-578
    tlen = (allpos < (origin_length + 1).unsqueeze(-1)).sum(1) - 1                     # :580
    return dec_input[:, :int(tlen.max()) + 1], tlen, allpos


# ----------------------------------------------------------------------------------------------
# StyleTransformer / Discriminator  (transformer.py:48-153)
# ----------------------------------------------------------------------------------------------
def style_transformer(w, cfg, inp_tokens, inp_lengths, gold_tokens, gold_lengths, style,
                      generate=False, temperature=1.0, use_inference_z=True, p="model_F.", sample_u=None):
    """Returns (log_probs[:,1:], tgt_lengths, length_loss, gen_error, allpos) like transformer.py:103."""
    w = _sub(w, p)
    B, S = inp_tokens.shape[0], inp_tokens.shape[1]
    assert int(inp_lengths.max()) <= S <= cfg.L
    pos = torch.arange(S)[None].expand(B, -1)
    src_mask = _pad_mask(inp_lengths, S, 1)
    enc_in = torch.cat([w["style_embed.weight"][style][:, None], embed(w, "embed.", inp_tokens, pos)], 1)   # :63-65
    memory = encoder(w, "encoder.", enc_in, src_mask, cfg.h, cfg.N)
    dec_in = memory if cfg.decoder_input == "encoded" else enc_in                    # :69-74
    tgt_lengths = length_loss = gen_err = allpos = None
    if cfg.learnable:
        if generate:
            dec_in, tgt_lengths, allpos = length_predictor_generate(w, cfg, dec_in, inp_lengths, use_inference_z,
                                                                    sample_u=sample_u)
        else:
            dec_in, tgt_lengths, length_loss, gen_err, _ = length_predictor_teacher(
                w, cfg, dec_in, inp_lengths, inp_tokens, gold_tokens, gold_lengths, use_inference_z)
        tgt_mask = _pad_mask(tgt_lengths, dec_in.shape[1] - 1, 1)                      # :86-89
    else:
        tgt_mask = src_mask
    y = decoder_stack(w, "decoder.", dec_in, memory, src_mask, tgt_mask, cfg.h, cfg.N)
    logp = generator(w, "decoder.generator.", _ln(w, "decoder.norm.", y), temperature)
    return logp[:, 1:], tgt_lengths, length_loss, gen_err, allpos


def discriminator(w, cfg, inp_tokens, inp_lengths, style=None, p="model_D."):
    """Discriminator.forward (transformer.py:130-153)."""
    w = _sub(w, p)
    B, S = inp_tokens.shape[0], inp_tokens.shape[1]
    extra = 1 if style is None else 2
    mask = _pad_mask(inp_lengths, S, extra)
    parts = [w["cls_token"].view(1, 1, -1).expand(B, -1, -1)]
    if style is not None:
        parts.append(w["style_embed.weight"][style][:, None])
    parts.append(embed(w, "embed.", inp_tokens, torch.arange(S)[None].expand(B, -1)))
    feats = encoder(w, "encoder.", torch.cat(parts, 1), mask, cfg.h, cfg.N)
    return torch.log_softmax(_lin(w, "classifier.", feats[:, 0]), -1)


class _sub(dict):
    """View of a flat weight dict under a prefix."""

    def __init__(self, w, prefix):
        super().__init__()
        self._w, self._p = w, prefix

    def __getitem__(self, k):
        return self._w[self._p + k]


# ----------------------------------------------------------------------------------------------
# gumbel family  (utils/gumbel.py:9-69) -- noise is injected as the Exp(1) draw E
# ----------------------------------------------------------------------------------------------
def gumbel_noise(E):
    return torch.min(-E.log(), torch.ones(1) * 1e10)                                   # gumbel.py:36


def gumbel_softmax(logits, E, tau=1.0):
    return ((logits + gumbel_noise(E)) / tau).softmax(-1)                              # gumbel.py:28-33


def gumbel_max(logits, E, tau=1.0):
    y_soft = gumbel_softmax(logits, E, tau)
    idx = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, idx, 1.0)
    return y_hard - y_soft.detach() + y_soft, y_soft                                   # gumbel.py:41-44


def gumbel_max_with_mask(logits, mask, E, tau=1.0):
    g = ((logits + gumbel_noise(E)) / tau).masked_fill(mask == 0, -1e9)               # gumbel.py:60-62
    y_soft = g.softmax(-1)
    idx = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, idx, 1.0)
    return y_hard - y_soft.detach() + y_soft


def straight_max(logits, allow_gradient=False):
    y_soft = logits.softmax(-1)
    idx = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, idx, 1.0)
    if allow_gradient:
        y_hard = y_hard - y_soft.detach() + y_soft
    return y_hard, y_soft


def onehot(sent, V):
    return torch.zeros(*sent.shape, V).scatter_(-1, sent.unsqueeze(-1), 1.0)


def get_lengths(tokens, eos_id):
    """Seq2seq.get_lengths (seq2seq.py:545-549): index of first eos + 1, capped at the width."""
    seen = torch.cumsum(tokens == eos_id, 1)
    n = (seen == 0).long().sum(-1)
    return torch.clamp(n + 1, max=tokens.shape[1])


def generate_mask(seqlen, length):
    """utils/operator_helper.py:99-101 -> [seqlen, B] float mask."""
    return torch.tensor((np.arange(seqlen)[:, None] < np.asarray(length)[None, :]).astype(np.float32))


# ----------------------------------------------------------------------------------------------
# training steps  (seq2seq.py:416-543 f_step, :551-642 d_step) with a restated Adam + clip
# ----------------------------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam defaults used by the reference (seq2seq.py:33-36): betas (0.9,0.999), eps 1e-8."""

    def __init__(self, names, lr, weight_decay=0.0):
        self.names, self.lr, self.wd = list(names), lr, weight_decay
        self.m, self.v, self.t = {}, {}, {}

    def step(self, w, grads):
        b1, b2, eps = 0.9, 0.999, 1e-8
        with torch.no_grad():
            for k in self.names:
                g = grads.get(k)
                if g is None:
                    continue
                if self.wd:
                    g = g + self.wd * w[k]
                if k not in self.m:
                    self.m[k], self.v[k], self.t[k] = torch.zeros_like(g), torch.zeros_like(g), 0
                self.t[k] += 1
                t = self.t[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
                w[k].addcdiv_(self.m[k], denom, value=-self.lr / (1 - b1 ** t))


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): returns the pre-clip total norm, scales grads in place."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in gs:
        g.mul_(coef)
    return float(total)


class Trainer:
    """Holds weights + the two Adam states; f_step / d_step follow seq2seq.py line by line in effect."""

    def __init__(self, weights, cfg):
        self.cfg = cfg
        self.w = {k: v.clone().float() for k, v in weights.items()}
        self.f_names = [k for k in self.w if k.startswith("model_F.")]
        self.d_names = [k for k in self.w if k.startswith("model_D.")]
        self.opt_F = Adam(self.f_names, cfg.lr_F)
        self.opt_D = Adam(self.d_names, cfg.lr_D)
        self.last_grads, self.last_norm = {}, None

    def _leaf(self, names):
        for k in names:
            self.w[k].requires_grad_(True)
            self.w[k].grad = None

    def _collect(self, names):
        g = {k: (self.w[k].grad.clone() if self.w[k].grad is not None else None) for k in names}
        for k in self.w:
            self.w[k].requires_grad_(False)
            self.w[k].grad = None
        return g

    def f_step(self, sent, sent_length, domain, noise_tok, noise_len, E_list, temperature=1.0,
               cyc_rec_enable=True, use_inference_z=True):
        cfg, w = self.cfg, self.w
        B = sent.shape[0]
        raw, rev = domain, 1 - domain
        tok_mask = (sent != cfg.pad_id).float()
        # D's parameters also receive (discarded) gradients in the reference; only F's are kept here
        self._leaf(self.f_names)
        slf_lp, _, slf_ll, slf_ge, _ = style_transformer(w, cfg, noise_tok, noise_len, sent, sent_length, raw,
                                                         False, temperature, use_inference_z)
        slf = (F.nll_loss(slf_lp.transpose(1, 2), sent, reduction="none") * tok_mask).sum() / B * cfg.slf_factor
        slf_len = None
        if cfg.learnable:
            if slf_ll.dim() <= 1:
                slf_len = slf_ll
            else:
                lm = generate_mask(slf_ll.shape[1], sent_length.numpy() + 1).t()
                slf_len = (slf_ll * lm).sum() / B * cfg.length_factor
            (slf + slf_len).backward()
        else:
            slf.backward()
        if not cyc_rec_enable:                                                           # seq2seq.py:471-474 (no clip)
            grads = self._collect(self.f_names)
            self.last_grads, self.last_norm = grads, None
            self.opt_F.step(w, grads)
            return (float(slf.detach()), 0, 0, _item(slf_len), 0, _item(slf_ge), 0)
        gen_lp, gen_len, _, _, _ = style_transformer(w, cfg, sent, sent_length, None, None, rev,
                                                     True, temperature, use_inference_z)
        soft = gumbel_max(gen_lp, E_list[0])[1]                                          # seq2seq.py:407-414
        if not cfg.learnable:
            gen_len = get_lengths(soft.argmax(-1), cfg.eos_id)
        cyc_lp, _, cyc_ll, cyc_ge, _ = style_transformer(w, cfg, soft, gen_len, sent, sent_length, raw,
                                                         False, temperature, use_inference_z)
        cyc = (F.nll_loss(cyc_lp.transpose(1, 2), sent, reduction="none") * tok_mask).sum() / B * cfg.cyc_factor
        cyc_len = None
        if cfg.learnable:
            if cyc_ll.dim() <= 1:
                cyc_len = cyc_ll
            else:
                lm = generate_mask(cyc_ll.shape[1], sent_length.numpy() + 1).t()
                cyc_len = (cyc_ll * lm).sum() / B * cfg.length_factor
        adv_lp = discriminator(w, cfg, soft, gen_len, rev if cfg.disc == "Cond" else rev)
        labels = rev + 1 if cfg.disc == "Multi" else torch.ones_like(rev)
        adv = F.nll_loss(adv_lp, labels, reduction="none")
        fac = (raw == 0) * cfg.fw_adv_factor + (raw == 1) * cfg.bw_adv_factor
        adv = (adv * fac).sum() / B
        (cyc + adv + cyc_len if cfg.learnable else cyc + adv).backward()
        grads = self._collect(self.f_names)
        self.last_norm = clip_grad_norm({k: g for k, g in grads.items()}, 5)
        self.last_grads = grads
        self.opt_F.step(w, grads)
        return (float(slf.detach()), float(cyc.detach()), float(adv.detach()), _item(slf_len), _item(cyc_len), _item(slf_ge), _item(cyc_ge))

    def d_step(self, sent, sent_length, domain, E_list, temperature=1.0, use_inference_z=True):
        cfg, w = self.cfg, self.w
        B = sent.shape[0]
        raw, rev = domain, 1 - domain
        with torch.no_grad():
            raw_lp, raw_len, _, _, _ = style_transformer(w, cfg, sent, sent_length, None, None, raw, True, temperature, use_inference_z)
            rev_lp, rev_len, _, _, _ = style_transformer(w, cfg, sent, sent_length, None, None, rev, True, temperature, use_inference_z)
        raw_soft = gumbel_max(raw_lp, E_list[0])[1]
        if not cfg.learnable:
            raw_len = get_lengths(raw_soft.argmax(-1), cfg.eos_id)
        rev_soft = gumbel_max(rev_lp, E_list[1])[1]
        if not cfg.learnable:
            rev_len = get_lengths(rev_soft.argmax(-1), cfg.eos_id)
        self._leaf(self.d_names)
        if cfg.disc == "Multi":
            lp = torch.cat([discriminator(w, cfg, sent, sent_length), discriminator(w, cfg, raw_soft, raw_len),
                            discriminator(w, cfg, rev_soft, rev_len)], 0)
            labels = torch.cat([raw + 1, raw + 1, torch.zeros_like(rev)], 0)
        else:
            lp = torch.cat([discriminator(w, cfg, sent, sent_length, raw), discriminator(w, cfg, sent, sent_length, rev),
                            discriminator(w, cfg, raw_soft, raw_len, raw), discriminator(w, cfg, rev_soft, rev_len, rev)], 0)
            labels = torch.cat([torch.ones_like(raw), torch.zeros_like(rev), torch.ones_like(raw), torch.zeros_like(rev)], 0)
        adv = F.nll_loss(lp, labels, reduction="none").sum() / B
        adv.backward()
        grads = self._collect(self.d_names)
        self.last_norm = clip_grad_norm(grads, 5)
        self.last_grads = grads
        self.opt_D.step(w, grads)
        return float(adv.detach())


def _item(x):
    if x is None:
        return 0
    return float(x.detach()) if isinstance(x, torch.Tensor) else float(x)
