"""NAST-StyTrans generator / discriminator on hand-written gfx950 kernels.

Same classes, constructor arguments, forward signatures and state_dict keys as the reference's
styletransformer/transformer.py (cited below as transformer.py:line), so checkpoints and callers are
interchangeable.  The arithmetic is done by libnast_hip.so through nast_amd.ops:

  * every sublayer `x + f(LN(x))` is ONE autograd node (LayerNorm -> grouped QKV GEMM on MFMA -> fused
    masked attention -> output GEMM with bias+residual epilogue), instead of ~25 ATen calls;
  * head split / merge are strides inside the attention kernel, no transpose/contiguous copies;
  * style/cls rows, token gather (or soft-token GEMM) and positions are assembled by one kernel;
  * the decoder's final LayerNorm + generator run on rows 1.. only (row 0 is discarded at transformer.py:103);
  * the alignment DP runs one wavefront per sentence; greedy skip decoding keeps a real K/V cache.

Remaining PyTorch glue (index arithmetic on [B, T] integer tensors and the 4-way skip-logit einsum /
cross-entropy of LengthPredictor's teacher-forced branch) is listed in DESIGN.md.
"""
import math

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .noise import NoiseLayer
from .utils.network_helper import BaseNetwork


class Network(BaseNetwork):
    """transformer.py:9-16 -- attribute names model_F / model_D drive the optimiser split."""

    def __init__(self, param):
        super().__init__(param)
        args = self.param.args
        self.model_F = StyleTransformer(param)
        self.model_D = Discriminator(param)
        self.noiseLayer = NoiseLayer(args.word_blank, args.word_dropout, args.word_shuffle, args.word_repeat,
                                     args.word_add, keep_go=False, max_length=args.max_length, dm=param.volatile.dm)


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's initialisation (transformer.py:332-350)
# ------------------------------------------------------------------------------------------------
class HipLinear(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias)


def Linear(in_features, out_features, bias=True, uniform=True):
    m = HipLinear(in_features, out_features, bias)
    if uniform:
        nn.init.xavier_uniform_(m.weight)
    else:
        nn.init.xavier_normal_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.)
    return m


def Embedding(num_embeddings, embedding_dim, padding_idx=None):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.xavier_uniform_(m.weight)
    # Reference quirk kept for init parity (transformer.py:345): weight[None] is the whole
    # matrix, so with padding_idx=None every embedding starts at exactly 0.
    nn.init.constant_(m.weight[padding_idx], 0)
    return m


def LayerNorm(embedding_dim, eps=1e-6):
    assert eps == 1e-6
    return HipLayerNorm(embedding_dim, eps)


def _pad_mask(lengths, width, extra):
    """[B,1,1,extra+width] bool, True = masked (transformer.py:56-61, :135-139, :459-461)."""
    B = lengths.shape[0]
    if lengths.is_cuda and lengths.dtype == torch.int64:
        m = torch.empty(B, extra + width, dtype=torch.bool, device=lengths.device)      # one launch instead of four
        ops.call("nast_pad_mask_u8", lengths.contiguous().data_ptr(), m.data_ptr(), B, int(width), int(extra), ops._stream())
    else:
        m = torch.arange(width, device=lengths.device).unsqueeze(0) >= lengths.unsqueeze(-1)
        if extra:
            m = torch.cat((torch.zeros(B, extra, dtype=torch.bool, device=lengths.device), m), 1)
    m = m.view(B, 1, 1, extra + width)
    m.nast_lengths = (lengths, extra)          # lets Encoder / Decoder run on packed rows (ops.packed_of)
    return m


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def scaled_attention(query, key, value, mask):
    """transformer.py:296-303 on [B,h,n,dk] tensors -> (attn_feature, None).
    The attention matrix is never materialised by the fused kernel, so the second return value is None."""
    B, h, nq, dk = query.shape
    q, k, v = (t.transpose(1, 2).reshape(B, -1, h * dk) for t in (query, key, value))
    out = ops.attention(q, k, v, mask, h)
    return out.view(B, nq, h, dk).transpose(1, 2), None


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model, h, dropout):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.head_projs = nn.ModuleList([HipLinear(d_model, d_model) for _ in range(3)])
        self.fc = HipLinear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, query, key, value, mask):
        """transformer.py:284-294"""
        q, k, v = [l(x) for x, l in zip((query, key, value), self.head_projs)]
        return self.fc(ops.attention(q, k, v, mask, self.h))


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, dropout):
        super().__init__()
        self.mlp = nn.Sequential(Linear(d_model, 4 * d_model), nn.ReLU(inplace=True), nn.Dropout(dropout),
                                 Linear(4 * d_model, d_model))
        _require_no_dropout(dropout)

    def forward(self, x):
        h = ops.linear(x, self.mlp[0].weight, self.mlp[0].bias, relu=True)
        return self.mlp[3](h)


def _require_no_dropout(p):
    if p != 0:
        raise NotImplementedError("the HIP path implements the reference configuration dropout=0 (run.py:139)")


class SublayerConnection(nn.Module):
    def __init__(self, d_model, dropout):
        super().__init__()
        self.layer_norm = LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        _require_no_dropout(dropout)

    def forward(self, x, sublayer):
        return x + sublayer(self.layer_norm(x))                      # transformer.py:324-326

    def incremental_forward(self, x, sublayer):
        return x[:, -1:] + sublayer(self.layer_norm(x))              # transformer.py:328-330


class EncoderLayer(nn.Module):
    def __init__(self, d_model, h, dropout):
        super().__init__()
        self.self_attn = MultiHeadAttention(d_model, h, dropout)
        self.pw_ffn = PositionwiseFeedForward(d_model, dropout)
        self.sublayer = nn.ModuleList([SublayerConnection(d_model, dropout) for _ in range(2)])

    def forward(self, x, mask):
        x = ops.attn_sublayer(x, None, mask, self.self_attn.h, self.sublayer[0].layer_norm, self.self_attn)
        return ops.ffn_sublayer(x, self.sublayer[1].layer_norm, self.pw_ffn)


class DecoderLayer(nn.Module):
    def __init__(self, d_model, h, dropout):
        super().__init__()
        self.self_attn = MultiHeadAttention(d_model, h, dropout)
        self.src_attn = MultiHeadAttention(d_model, h, dropout)
        self.pw_ffn = PositionwiseFeedForward(d_model, dropout)
        self.sublayer = nn.ModuleList([SublayerConnection(d_model, dropout) for _ in range(3)])

    def forward(self, x, memory, src_mask, tgt_mask):
        h = self.self_attn.h
        x = ops.attn_sublayer(x, None, tgt_mask, h, self.sublayer[0].layer_norm, self.self_attn)
        x = ops.attn_sublayer(x, memory, src_mask, h, self.sublayer[1].layer_norm, self.src_attn)
        return ops.ffn_sublayer(x, self.sublayer[2].layer_norm, self.pw_ffn)

    def incremental_forward(self, x, memory, src_mask, tgt_mask, prev_states=None):
        """transformer.py:259-272 with the same state convention (lists of layer inputs)."""
        new_states = []
        m = memory
        x = torch.cat((prev_states[0], x), 1) if prev_states else x
        new_states.append(x)
        x = self.sublayer[0].incremental_forward(x, lambda x: self.self_attn(x[:, -1:], x, x, tgt_mask))
        x = torch.cat((prev_states[1], x), 1) if prev_states else x
        new_states.append(x)
        x = self.sublayer[1].incremental_forward(x, lambda x: self.src_attn(x[:, -1:], m, m, src_mask))
        x = torch.cat((prev_states[2], x), 1) if prev_states else x
        new_states.append(x)
        x = self.sublayer[2].incremental_forward(x, lambda x: self.pw_ffn(x[:, -1:]))
        return x, new_states


class Encoder(nn.Module):
    def __init__(self, num_layers, d_model, vocab_size, h, dropout):
        super().__init__()
        self.layers = nn.ModuleList([EncoderLayer(d_model, h, dropout) for _ in range(num_layers)])
        self.norm = LayerNorm(d_model)

    def forward(self, x, mask, rows=None):
        """transformer.py:163-171.  rows=(grp, grp_stride, grp_off, rows_out): normalise only those rows."""
        assert x.size(1) == mask.size(-1)
        ps = ops.packed_of(mask, x, x.shape[-1] // self.layers[0].self_attn.h)
        if ps is not None and (rows is None or (rows[0] == 1 and rows[2] == 0)):
            # packed rows: only the real tokens of every sentence go through the layers
            y = ops.pack_rows(x, ps)
            pair = ops.PackedPair(ps, ps)
            for layer in self.layers:
                y = layer(y, pair)
            if rows is not None:                               # first row of every sentence (the discriminator's cls row)
                return self.norm(y.index_select(0, ps.off))
            return ops.unpack_rows(self.norm(y), ps)
        y = x
        for layer in self.layers:
            y = layer(y, mask)
        if rows is not None:
            return ops.layer_norm_rows(y, self.norm.weight, self.norm.bias, *rows)
        return self.norm(y)


class Generator(nn.Module):
    def __init__(self, d_model, vocab_size):
        super().__init__()
        self.proj = nn.Linear(d_model, vocab_size)

    def forward(self, x, temperature, normalise=True):
        return ops.generator(x, self.proj.weight, self.proj.bias, temperature, normalise)   # transformer.py:213-214

    def nll(self, x, temperature, tokens):
        """-forward(x, temperature).gather(-1, tokens): the NLLLoss(reduction='none') of seq2seq.py:453 fused in."""
        return ops.generator_nll(x, self.proj.weight, self.proj.bias, temperature, tokens)


class Decoder(nn.Module):
    lazy_log_softmax = True     # packed log-probs of the transfer pass leave the generator as logits (see _forward_packed)

    def __init__(self, num_layers, d_model, vocab_size, h, dropout):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayer(d_model, h, dropout) for _ in range(num_layers)])
        self.norm = LayerNorm(d_model)
        self.generator = Generator(d_model, vocab_size)

    def forward(self, x, memory, src_mask, tgt_mask, temperature, skip_first=False, nll_targets=None, packed_out=False):
        """transformer.py:180-188.  skip_first=True returns log-probs for positions 1.. only
        (what StyleTransformer keeps, transformer.py:103) without computing row 0; with nll_targets [B, n-1]
        the per-token negative log-likelihood of those tokens is returned instead of the log-probs."""
        assert x.size(1) == tgt_mask.size(-1)
        dk = x.shape[-1] // self.layers[0].self_attn.h
        pt, pm = ops.packed_of(tgt_mask, x, dk), ops.packed_of(src_mask, memory, dk)
        if pt is not None and pm is not None and skip_first:
            return self._forward_packed(x, memory, pt, pm, temperature, nll_targets, packed_out)
        y = x
        memory = ops.shared_kv(memory)      # one gradient buffer for all layers' cross-attention (ops.SharedKVFn)
        for layer in self.layers:
            y = layer(y, memory, src_mask, tgt_mask)
        B, n, d = y.shape
        if skip_first:
            z = ops.layer_norm_rows(y, self.norm.weight, self.norm.bias, n - 1, n, 1, B * (n - 1))
            if nll_targets is not None:
                assert nll_targets.shape == (B, n - 1)
                return self.generator.nll(z, temperature, nll_targets).view(B, n - 1)
            return self.generator(z, temperature).view(B, n - 1, -1)
        return self.generator(self.norm(y), temperature)

    def _forward_packed(self, x, memory, pt, pm, temperature, nll_targets, packed_out=False):
        """The same stack on the real rows only (ops.PackedSeq); positions 1.. of every sentence are kept."""
        B, n, d = x.shape
        y, mem = ops.pack_rows(x, pt), ops.pack_rows(memory, pm)
        self_pair, cross_pair = ops.PackedPair(pt, pt), ops.PackedPair(pt, pm)
        mem = ops.shared_kv(mem)
        for layer in self.layers:
            y = layer(y, mem, cross_pair, self_pair)
        rows, sub = pt.tail()                                   # packed rows of positions >= 1, their [B, n-1] layout
        z = self.norm(y.index_select(0, rows))
        if nll_targets is not None:
            # generator + NLL on the real target positions only (padded ones are multiplied by a zero mask downstream)
            assert nll_targets.shape == (B, n - 1)
            tok = nll_targets.reshape(-1).index_select(0, sub.idx_p)
            nll = self.generator.nll(z, temperature, tok).view(-1)[:sub.total]
            return torch.zeros(B * (n - 1), device=x.device, dtype=nll.dtype).index_copy(0, sub.idx, nll).view(B, n - 1)
        if packed_out:                                          # log-probs of the real positions only: [sub.rows, V]
            # (at temperature 1 the log-sum-exp pass is left to the consumer: Seq2seq._approx_packed's Gumbel-softmax does
            # not need it, ops.GeneratorFn)
            raw = float(temperature) == 1.0 and self.lazy_log_softmax
            return ops.PackedSoft(self.generator(z, temperature, normalise=not raw), sub, unnormalised=raw)
        return self.generator(ops.unpack_rows(z, sub), temperature)


class EmbeddingLayer(nn.Module):
    def __init__(self, dm, d_model, max_length, pad_idx, learned_pos_embed, wordvec=None):
        super().__init__()
        self.token_embed = Embedding(dm.frequent_vocab_size, d_model)
        self.pos_embed = Embedding(max_length, d_model)
        self.vocab_size = dm.frequent_vocab_size
        if wordvec is not None:
            self.token_embed = nn.Embedding.from_pretrained(torch.tensor(wordvec, dtype=torch.float32))

    def soft_rows(self, x):
        """[B,L,V] soft tokens -> [B,L,d] = x @ token_embed.weight (transformer.py:229), MFMA GEMM."""
        if isinstance(x, ops.PackedSoft):
            return x.soft_embed(self.token_embed.weight)
        return ops.matmul_nn(x, self.token_embed.weight)

    def forward(self, x, pos):
        """transformer.py:225-231.  `pos` must be the leading positions 0..L-1 (the only use in the reference)."""
        tok = x if x.dim() == 2 else self.soft_rows(x)
        return ops.embed_assemble(tok, self.token_embed.weight if x.dim() == 2 else None, self.pos_embed.weight)

    def assemble(self, x, extra, soft_rows=None):
        """[extra rows | embed(x, 0..L-1)] in one kernel (replaces torch.cat((style_emb, embed(..)), 1))."""
        if x.dim() == 2:
            return ops.embed_assemble(x, self.token_embed.weight, self.pos_embed.weight, extra)
        rows = soft_rows if soft_rows is not None else self.soft_rows(x)
        return ops.embed_assemble(rows, None, self.pos_embed.weight, extra)


# ------------------------------------------------------------------------------------------------
# StyleTransformer / Discriminator
# ------------------------------------------------------------------------------------------------
class StyleTransformer(nn.Module):
    check_lengths = True   # the reference asserts max(inp_lengths) <= width (a host sync); set False to skip

    def __init__(self, param):
        super().__init__()
        self.param = param
        config = args = self.args = param.args
        dm = param.volatile.dm
        self.max_length = config.max_length
        self.eos_idx = dm.eos_id
        self.pad_idx = dm.pad_id
        self.style_embed = Embedding(config.num_styles, config.d_model)
        self.embed = EmbeddingLayer(dm, config.d_model, config.max_length, self.pad_idx, config.learned_pos_embed)
        self.sos_token = nn.Parameter(torch.randn(config.d_model))   # dead parameter kept for checkpoints (:41)
        self.encoder = Encoder(config.num_layers, config.d_model, dm.frequent_vocab_size, config.h, config.dropout)
        self.decoder = Decoder(config.num_layers, config.d_model, dm.frequent_vocab_size, config.h, config.dropout)
        if args.use_learnable:
            self.length_predictor = LengthPredictor(param)

    def forward(self, inp_tokens, inp_lengths, gold_tokens, gold_lengths, style,
                generate=False, differentiable_decode=False, temperature=1.0, *, use_inference_z, nll_targets=None,
                packed_log_probs=False):
        """transformer.py:48-103 -> (log_probs[:,1:], tgt_lengths, length_loss, gen_error, allpos).
        Extension: nll_targets [B,T] int64 -> the first result is -log_probs[:,1:].gather(-1, nll_targets) [B,T]
        (what seq2seq.py:453 / :504 compute from the log-probs), without materialising their gradient."""
        batch_size = inp_tokens.size(0)
        max_enc_len = inp_tokens.size(1)
        if self.check_lengths and not ops.FIXED_WIDTH:
            assert torch.max(inp_lengths) <= inp_tokens.shape[1]
        assert max_enc_len <= self.max_length

        src_mask = _pad_mask(inp_lengths, max_enc_len, 1)
        if not self.args.use_learnable:
            # Without the length predictor the decoder works on the SOURCE positions and masks by the source length
            # (tgt_mask = src_mask, transformer.py:90) while the loss covers the gold length, and the generated length is
            # read off the output itself (first <eos>, seq2seq.py:545-549): rows past the source length are live, so this
            # model keeps the padded execution.  With the length predictor every consumer masks by explicit lengths.
            del src_mask.nast_lengths
        soft_rows = self.embed.soft_rows(inp_tokens) if inp_tokens.dim() == 3 else None
        enc_input = self.embed.assemble(inp_tokens, [(self.style_embed.weight, style)], soft_rows)
        memory = self.encoder(enc_input, src_mask)

        if self.args.decoder_input == "encoded":
            dec_input_emb = memory
        elif self.args.decoder_input == "raw":
            dec_input_emb = enc_input
        else:
            raise NotImplementedError()

        if self.args.use_learnable:
            if generate:
                dec_input_emb, tgt_lengths, allpos = self.length_predictor(
                    dec_input_emb, inp_lengths, embed=self.embed, use_inference_z=use_inference_z)
                length_loss = None
                gen_error = None
            else:
                dec_input_emb, tgt_lengths, length_loss, gen_error = self.length_predictor(
                    dec_input_emb, inp_lengths, inp_tokens, gold_tokens, gold_lengths, embed=self.embed,
                    use_inference_z=use_inference_z, origin_soft_rows=soft_rows)
                allpos = None
            tgt_mask = _pad_mask(tgt_lengths, dec_input_emb.size(1) - 1, 1)
        else:
            tgt_mask = src_mask
            tgt_lengths = length_loss = gen_error = allpos = None

        # packed_log_probs (extension): when the decoder runs on packed rows, hand the log-probs back as ops.PackedSoft
        # (real positions only) instead of the padded [B, T, V] tensor
        log_probs = self.decoder(dec_input_emb, memory, src_mask, tgt_mask, temperature, skip_first=True,
                                 nll_targets=nll_targets, packed_out=packed_log_probs)
        return log_probs, tgt_lengths, length_loss, gen_error, allpos


class Discriminator(nn.Module):
    def __init__(self, param):
        super().__init__()
        self.param = param
        self.args = config = param.args
        dm = self.param.volatile.dm
        self.pad_idx = dm.pad_id
        self.style_embed = Embedding(config.num_styles, config.d_model)
        self.embed = EmbeddingLayer(dm, config.d_model, config.max_length, self.pad_idx, config.learned_pos_embed)
        self.cls_token = nn.Parameter(torch.randn(config.d_model))
        self.encoder = Encoder(config.num_layers, config.d_model, dm.frequent_vocab_size, config.h, config.dropout)
        self.classifier = Linear(config.d_model, config.num_classes)

    def forward(self, inp_tokens, inp_lengths, style=None):
        """transformer.py:130-153"""
        B = inp_tokens.size(0)
        num_extra = 1 if style is None else 2
        L = inp_tokens.size(1)
        mask = _pad_mask(inp_lengths, L, num_extra)
        extra = [(self.cls_token, None)]
        if style is not None:
            extra.append((self.style_embed.weight, style))
        enc_input = self.embed.assemble(inp_tokens, extra)
        # only row 0 (the cls position) of the final LayerNorm is consumed (transformer.py:151)
        feats = self.encoder(enc_input, mask, rows=(1, L + num_extra, 0, B))
        logits = self.classifier(feats)
        return ops.log_softmax(logits)


# ------------------------------------------------------------------------------------------------
# LengthPredictor
# ------------------------------------------------------------------------------------------------
class HiddenDecoder(nn.Module):
    def __init__(self, num_layers, d_model, h, dropout):
        super().__init__()
        self.layers = nn.ModuleList([DecoderLayer(d_model, h, dropout) for _ in range(num_layers)])
        self.norm = LayerNorm(d_model)

    def forward(self, x, memory, src_mask, tgt_mask, causal_lengths=None):
        """transformer.py:359-364 -- NB: no final LayerNorm here (the reference's train/inference asymmetry).
        causal_lengths [B] (extension): tgt_mask is the plain causal mask and only the first causal_lengths[b] positions of
        sentence b are consumed downstream -- the stack may then run on those rows only (ops.PackedSeq); the other
        positions come back as zeros."""
        assert x.size(1) == tgt_mask.size(-1)
        dk = x.shape[-1] // self.layers[0].self_attn.h
        pm = ops.packed_of(src_mask, memory, dk) if causal_lengths is not None else None
        if pm is not None and ops.pack_len_ok(x.size(1), dk):
            pt = ops.PackedSeq(causal_lengths, x.size(1))
            y, mem = ops.pack_rows(x, pt), ops.pack_rows(memory, pm)
            self_pair, cross_pair = ops.PackedPair(pt, pt, causal=True), ops.PackedPair(pt, pm)
            mem = ops.shared_kv(mem)
            for layer in self.layers:
                y = layer(y, mem, cross_pair, self_pair)
            return ops.unpack_rows(y, pt)
        y = x
        memory = ops.shared_kv(memory)      # one gradient buffer for all layers' cross-attention (ops.SharedKVFn)
        for layer in self.layers:
            y = layer(y, memory, src_mask, tgt_mask)
        return y

    def incremental_forward(self, x, memory, src_mask, tgt_mask, prev_states=None):
        """transformer.py:366-377 (state-list API; the fast path is `KVCacheDecoder` below)."""
        y = x
        new_states = []
        for i, layer in enumerate(self.layers):
            y, sub = layer.incremental_forward(y, memory, src_mask, tgt_mask, prev_states[i] if prev_states else None)
            new_states.append(sub)
        new_states.append(torch.cat((prev_states[-1], y), 1) if prev_states else y)
        y = self.norm(new_states[-1])[:, -1:]
        return y, new_states


class KVCacheDecoder:
    """Step-wise evaluation of a HiddenDecoder under no_grad with a real K/V cache.

    Equivalent to HiddenDecoder.incremental_forward (which re-normalises and re-projects the whole prefix
    at every step, transformer.py:263-271): keys/values of position j depend only on the layer input at j,
    so they are projected once and appended to a [B, max_steps, d] cache that the attention kernel reads in
    place (batch stride = max_steps*d).  Cross-attention K/V of `memory` are projected once for all layers
    in one grouped GEMM launch."""

    def __init__(self, dec, memory, src_mask, max_steps):
        self.dec, self.src_mask = dec, src_mask
        B, ne, d = memory.shape
        self.B, self.d, self.h = B, d, dec.layers[0].self_attn.h
        dev = memory.device
        nl = len(dec.layers)
        self.kc = torch.empty(nl, B, max_steps, d, device=dev)
        self.vc = torch.empty(nl, B, max_steps, d, device=dev)
        self.mk = torch.empty(nl, B, ne, d, device=dev)
        self.mv = torch.empty(nl, B, ne, d, device=dev)
        mem2 = memory.reshape(-1, d)
        descs = []
        for i, layer in enumerate(dec.layers):
            p = layer.src_attn.head_projs
            descs.append(ops.d_fwd(mem2, p[1].weight, self.mk[i].view(-1, d), bias=p[1].bias))
            descs.append(ops.d_fwd(mem2, p[2].weight, self.mv[i].view(-1, d), bias=p[2].bias))
        ops.gemm(descs)
        self.max_steps = max_steps

    def step(self, x, k):
        """x [B,d] = input of position k -> hidden [B,d] (after the final LayerNorm, transformer.py:376)."""
        B, d, h = self.B, self.d, self.h
        L = self.max_steps
        for i, layer in enumerate(self.dec.layers):
            sa, ca = layer.self_attn, layer.src_attn
            z, _, _ = ops.ln_fwd(x, layer.sublayer[0].layer_norm.weight, layer.sublayer[0].layer_norm.bias)
            q = torch.empty(B, d, device=x.device)
            kslot, vslot = self.kc[i, :, k], self.vc[i, :, k]        # [B,d] views, row stride L*d
            p = sa.head_projs
            dk_, dv_ = ops.d_fwd(z, p[1].weight, kslot, bias=p[1].bias), ops.d_fwd(z, p[2].weight, vslot, bias=p[2].bias)
            dk_.ldc = dv_.ldc = L * d
            ops.gemm([ops.d_fwd(z, p[0].weight, q, bias=p[0].bias), dk_, dv_])
            a, _ = ops.mha_fwd(q.view(B, 1, d), self.kc[i, :, :k + 1], self.vc[i, :, :k + 1], None, h)
            x1 = torch.empty_like(x)
            ops.gemm([ops.d_fwd(a.view(B, d), sa.fc.weight, x1, bias=sa.fc.bias, residual=x)])
            z, _, _ = ops.ln_fwd(x1, layer.sublayer[1].layer_norm.weight, layer.sublayer[1].layer_norm.bias)
            ops.gemm([ops.d_fwd(z, ca.head_projs[0].weight, q, bias=ca.head_projs[0].bias)])
            a, _ = ops.mha_fwd(q.view(B, 1, d), self.mk[i], self.mv[i], self.src_mask, h)
            x2 = torch.empty_like(x)
            ops.gemm([ops.d_fwd(a.view(B, d), ca.fc.weight, x2, bias=ca.fc.bias, residual=x1)])
            z, _, _ = ops.ln_fwd(x2, layer.sublayer[2].layer_norm.weight, layer.sublayer[2].layer_norm.bias)
            f = layer.pw_ffn.mlp
            hmid = torch.empty(B, f[0].weight.shape[0], device=x.device)
            ops.gemm([ops.d_fwd(z, f[0].weight, hmid, bias=f[0].bias, relu=True)])
            x = torch.empty_like(x2)
            ops.gemm([ops.d_fwd(hmid, f[3].weight, x, bias=f[3].bias, residual=x2)])
        y, _, _ = ops.ln_fwd(x, self.dec.norm.weight, self.dec.norm.bias)
        return y


class LengthPredictor(nn.Module):
    def __init__(self, param):
        super().__init__()
        self.param = param
        self.args = args = param.args
        self.pos_embed = Embedding(args.max_length + 1, args.d_model, padding_idx=0)
        self.decoder = HiddenDecoder(3, args.d_model, args.h, args.dropout)
        self.max_skip = 4
        self.sqrt_d_model = np.sqrt(self.args.d_model)
        self.linear = HipLinear(args.d_model, self.max_skip)
        self.wq = HipLinear(args.d_model, args.d_model)
        self.wk = HipLinear(args.d_model, args.d_model)

    # -- alignment DP -------------------------------------------------------------------------
    def match(self, origin_token, origin_length, target_token, target_length, embed, origin_soft_rows=None):
        """transformer.py:394-449 -> allpos int64 [B, T+1].  Cost tables + Viterbi run in two kernels
        (one wavefront per target row / per sentence); indices are bit-exact given the same costs."""
        unk_id = self.param.volatile.dm.unk_id
        E = embed.token_embed.weight.detach()
        with torch.no_grad():
            if origin_token.dim() == 2:
                cost0, cost = ops.align_costs(E, target_token, src_ids=origin_token, unk_id=unk_id)
            else:
                # soft source: (p with p[unk] folded into p[0]) @ E == p @ E + p[unk] * (E[0] - E[unk]);
                # p @ E is the embedding GEMM the encoder input already needed (re-used when given)
                packed = isinstance(origin_token, ops.PackedSoft)
                if origin_soft_rows is not None:
                    rows = origin_soft_rows.detach()
                else:
                    rows = origin_token.soft_embed(E).detach() if packed else ops.matmul_nn(origin_token.detach(), E)
                unk_w = origin_token.column(unk_id) if packed else origin_token.detach()[..., unk_id]
                cost0, cost = ops.align_costs(E, target_token, src_emb=rows, src_unk_w=unk_w,
                                              unk_id=unk_id)
            return ops.align_viterbi(cost0, cost, origin_length, target_length)

    static_decode_workspace = False     # True: nast_lp_decode_static_f32 (the library's built-in workspace, no allocation)
    sample_u = None                     # test hook: [max_length, B] uniforms for z_decode_mode == "sample"

    def _native_decode(self, memory, origin_length, eos, unk, allpos_buf, static=None, sample_u=None):
        from ._lib import DecLayerWeights, LPDecodeArgs, load
        import ctypes as C
        lib = load()
        B, S1, d = memory.shape
        L = self.args.max_length
        layers = self.decoder.layers
        dff = layers[0].pw_ffn.mlp[0].weight.shape[0]
        arr = (DecLayerWeights * len(layers))()
        for i, l in enumerate(layers):
            sa, ca, f, sl = l.self_attn, l.src_attn, l.pw_ffn.mlp, l.sublayer
            vals = [sl[0].layer_norm.weight, sl[0].layer_norm.bias,
                    sa.head_projs[0].weight, sa.head_projs[0].bias, sa.head_projs[1].weight, sa.head_projs[1].bias,
                    sa.head_projs[2].weight, sa.head_projs[2].bias, sa.fc.weight, sa.fc.bias,
                    sl[1].layer_norm.weight, sl[1].layer_norm.bias,
                    ca.head_projs[0].weight, ca.head_projs[0].bias, ca.head_projs[1].weight, ca.head_projs[1].bias,
                    ca.head_projs[2].weight, ca.head_projs[2].bias, ca.fc.weight, ca.fc.bias,
                    sl[2].layer_norm.weight, sl[2].layer_norm.bias, f[0].weight, f[0].bias, f[3].weight, f[3].bias]
            for (name, _), v in zip(DecLayerWeights._fields_, vals):
                setattr(arr[i], name, v.data_ptr())
        static = self.static_decode_workspace if static is None else static
        ws = None
        if not static:
            nbytes = lib.nast_lp_decode_workspace_bytes(B, S1, d, dff, L, len(layers))
            ws = torch.empty(nbytes // 4 + 64, device=memory.device, dtype=torch.float32)
        memory, origin_length, eos, unk = memory.contiguous(), origin_length.contiguous(), eos.contiguous(), unk.contiguous()
        a = LPDecodeArgs()
        a.B, a.S1, a.d, a.h, a.dff, a.L, a.nlayers = B, S1, d, layers[0].self_attn.h, dff, L, len(layers)
        a.layers = arr
        a.norm_w, a.norm_b = self.decoder.norm.weight.data_ptr(), self.decoder.norm.bias.data_ptr()
        a.wq_w, a.wq_b, a.wk_w, a.wk_b = (t.data_ptr() for t in (self.wq.weight, self.wq.bias, self.wk.weight, self.wk.bias))
        a.lin_w, a.lin_b = self.linear.weight.data_ptr(), self.linear.bias.data_ptr()
        a.pos_w = self.pos_embed.weight.data_ptr()
        a.eos, a.unk = eos.data_ptr(), unk.data_ptr()
        a.memory, a.origin_length, a.allpos = memory.data_ptr(), origin_length.data_ptr(), allpos_buf.data_ptr()
        a.workspace = None if static else ws.data_ptr() + (-ws.data_ptr()) % 256
        if sample_u is not None:
            assert sample_u.shape == (L, B) and sample_u.dtype == torch.float32 and sample_u.is_contiguous()
        a.sample_u = None if sample_u is None else sample_u.data_ptr()
        steps = C.c_int(-1 if ops.ASYNC_DECODE else 0)       # -1: no early-exit read-backs, the call does not synchronise
        ops.call("nast_lp_decode_static_f32" if static else "nast_lp_decode_f32", C.byref(a), C.byref(steps), ops._stream())
        self._decode_keepalive = (ws, memory, eos, unk, sample_u)       # the enqueued kernels still read them
        return steps.value

    # -- forward ------------------------------------------------------------------------------
    def forward(self, origin_input, origin_length, origin_token=None, target_token=None, target_length=None, embed=None,
                *, use_inference_z, origin_soft_rows=None):
        """transformer.py:452-582.  origin_input [B,S+1,d] is the encoder memory (style token first)."""
        dm = self.param.volatile.dm
        batch_size = origin_input.shape[0]
        S1 = origin_input.shape[1]
        d = self.args.d_model
        dev = origin_input.device
        eos_embed = ops.table_row(embed.token_embed.weight, dm.eos_id)      # :468-469 (live parameters)
        unk_embed = ops.table_row(embed.token_embed.weight, dm.unk_id)
        src_mask = _pad_mask(origin_length, S1 - 1, 1)
        memory = origin_input

        if target_token is not None:
            if not use_inference_z:                                           # :480-484
                assert ops.FIXED_WIDTH or (origin_length == target_length).sum() == batch_size    # (a device read)
                z = torch.zeros(1, dtype=torch.float, device=dev)[0]
                return origin_input, target_length, z, z.clone()
            allpos = self.match(origin_token, origin_length, target_token, target_length, embed, origin_soft_rows)
            T1 = target_token.shape[1] + 1
            # aligned decoder input (:488-491) and its positional variant (:493) from one gather each
            dec_input = ops.lp_gather(origin_input, origin_length, allpos, eos_embed, unk_embed, None, 0, True)
            dec_input_new = ops.lp_gather(origin_input, origin_length, allpos, eos_embed, unk_embed, self.pos_embed.weight, 0, True)
            Lm = self.args.max_length + 1
            causal = (torch.ones((Lm, Lm), device=dev).tril() == 0).view(1, 1, Lm, Lm)   # :463-464
            # positions 0..target_length of a sentence are the ones its loss reads (seq2seq.py:461); a causal stack never
            # lets a later position influence them
            live_len = target_length + 1
            if hasattr(target_length, "nast_total"):
                ops.set_total(live_len, target_length.nast_total + batch_size)
            hidden = self.decoder(dec_input_new, memory, src_mask, causal[:, :, :T1, :T1], causal_lengths=live_len)
            # skip logits (:504-507): wk is linear, so wk(candidates)[allpos+j] == wk applied once per source row
            cand_idx = torch.arange(S1 + 5, device=dev).unsqueeze(0).expand(batch_size, -1)
            cand_in = ops.lp_gather(origin_input, origin_length, cand_idx, eos_embed, unk_embed, None, 0, False)
            KC = self.wk(cand_in)                                             # [B,S1+5,d]
            kunk = self.wk(unk_embed.view(1, d))                              # [1,d]
            qh = self.wq(hidden)                                              # [B,T1,d]
            logits = ops.lp_logits(qh, KC, kunk.view(d), allpos)                                      # [B,T1,4], already / sqrt(d)
            logits = logits + self.linear(hidden)
            target = torch.cat([allpos[:, 1:], origin_length.unsqueeze(-1) + 1], dim=1) - allpos     # :509
            gen_err = (target >= 4).sum().float() / target.shape[0]                                  # :511
            target = torch.clamp(target, max=3)
            length_loss = F.cross_entropy(logits.transpose(1, 2), target, reduction="none")         # :517
            return dec_input, target_length, length_loss, gen_err

        if not use_inference_z:
            return origin_input, origin_length                                # :521-522 (as the reference)

        # ---- skip decoding (:525-570): the whole loop runs in native code (nast_lp_decode_f32) ----
        if self.args.z_decode_mode not in ("max", "sample"):
            raise NotImplementedError("no such z_decode_mode")                # :562-563
        L = self.args.max_length
        with torch.no_grad():
            allpos_buf = torch.empty(batch_size, L + 1, dtype=torch.long, device=dev)
            # "sample" (:560-561): one uniform per sentence and step from torch's generator, drawn up front; the skip
            # kernel turns it into a multinomial draw by inverse CDF.  `sample_u` (test hook) injects the uniforms.
            u = None
            if self.args.z_decode_mode == "sample":
                u = self.sample_u if self.sample_u is not None else torch.rand(L, batch_size, device=dev)
            steps = self._native_decode(origin_input.detach(), origin_length, eos_embed.detach(), unk_embed.detach(), allpos_buf,
                                        sample_u=u)
            ops.run_deferred()       # host work that can proceed while the decode runs (ops.async_decode)
            allpos = allpos_buf[:, :steps + 1].contiguous()
        dec_input = ops.lp_gather(origin_input, origin_length, allpos, eos_embed, unk_embed, None, 0, True)   # :576-578
        target_length = (allpos < (origin_length + 1).unsqueeze(-1)).sum(dim=1) - 1                  # :580
        if ops.FIXED_WIDTH:          # graph replay: all max_length + 1 positions stay, the lengths mask the rest
            return dec_input, target_length, allpos
        # one device-to-host read for the width (the reference's .max(), :582) and the total (packed rows downstream)
        t_max, t_sum = torch.stack((target_length.max(), target_length.sum())).tolist()
        ops.set_total(target_length, t_sum)
        return dec_input[:, :t_max + 1], target_length, allpos
