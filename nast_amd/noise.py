"""Input noise of the self-reconstruction pass (reference: styletransformer/noise.py:8-230).

Host-side (numpy) restatement: the noise layer runs once per f_step on [L, B] integer ids, its outputs are
INPUTS of the GPU hot path (a device-side version is a "next" row).  Random
numbers are drawn from numpy's global generator in the same order and shapes as the reference, so a run with
the same numpy seed produces the same blank / dropout decisions; the added words come from torch.multinomial
as in the reference.  Every stage is a pure function of its random draws (`*_apply`), which is what the
tests exercise.
"""
import numpy as np
import torch
from torch import nn


class NoiseLayer(nn.Module):
    def __init__(self, word_blank, word_dropout, word_shuffle, word_repeat, word_add, *,
                 keep_go=True, keep_eos=True, max_length, dm):
        super().__init__()
        self.blank_prob, self.dropout_prob = word_blank, word_dropout
        self.shuffle_weight, self.repeat_prob, self.add_prob = word_shuffle, word_repeat, word_add
        self.pad_index, self.blank_index, self.eos_index = dm.pad_id, dm.unk_id, dm.eos_id
        self.keep_go, self.keep_eos = keep_go, keep_eos
        self.max_length = max_length
        self.dm = dm
        self.register_buffer("cnt", torch.tensor(self._add_distribution(dm), dtype=torch.float))

    @staticmethod
    def _add_distribution(dm):
        """tf-idf weighted unigram distribution per style used by word_add (noise.py:33-58)."""
        V = dm.frequent_vocab_size
        freq, doc = [], np.zeros(V)
        n_sent = 0
        for domain in (0, 1):
            sents = dm.get_all_batch(f"train_{domain}")["sent"]
            tf = np.zeros(V)
            df = np.zeros(V)
            for s in sents:
                ids = np.asarray(s, dtype=np.int64)
                np.add.at(tf, ids, 1)
                df[np.unique(ids)] += 1
            tf[:4] = 0
            freq.append(tf / tf.sum())
            doc += df / len(sents)
            n_sent += len(sents)
        idf = np.maximum(np.log(0.3 / (doc + 1e-10)), 0)
        out = []
        for tf in freq:
            p = tf * idf
            with np.errstate(invalid="ignore", divide="ignore"):
                out.append(p / p.sum())          # a style without tf-idf mass gives NaN here as in the reference (noise.py:52-56);
        return np.array(out)                     # _checked_cnt() refuses to draw from such a row

    def _checked_cnt(self):
        """`cnt` for word_add, validated once per content version: torch.multinomial on a NaN / zero-mass row is undefined
        behaviour on the device (the reference would draw from it silently)."""
        ver = (self.cnt._version, self.cnt.data_ptr())
        if getattr(self, "_cnt_ok", None) != ver:
            c = self.cnt
            bad = ~(torch.isfinite(c).all(1) & (c >= 0).all(1) & (c.sum(1) > 0))
            if bool(bad.any()):
                raise ValueError(f"NoiseLayer.word_add: the word distribution of style(s) {bad.nonzero().flatten().tolist()} has no "
                                 "tf-idf mass (every word of that style occurs in > 30 % of the sentences, or the style is empty): "
                                 "set word_add = 0 or provide NoiseLayer.cnt")
            self._cnt_ok = ver
        return self.cnt

    # ---- stages as pure functions of their random draws ------------------------------------
    def blank_apply(self, x, l, u):
        """x [seq,B] ids, l [B], u [seq,B] uniforms: positions before the final token are replaced by <unk>
        where u < blank_prob (noise.py:103-135 with keep_go=False, keep_eos=True)."""
        keep = u >= self.blank_prob
        pos = np.arange(x.shape[0])[:, None]
        if self.keep_go:
            keep[0] = True
        keep |= pos >= (l[None, :] - 1 if self.keep_eos else l[None, :])
        return np.where(keep, x, self.blank_index), l

    def add_apply(self, x, l, u, new_words):
        """Insert words: u [M-1,B] uniforms (u < add_prob -> the slot holds an inserted word),
        new_words [M,B] candidates (noise.py:76-101)."""
        M = self.max_length
        B = x.shape[1]
        orig = np.concatenate([np.ones((1, B), dtype=bool), u >= self.add_prob], 0)        # True: slot shows a source word
        src = np.cumsum(orig, 0) - 1                                                       # which source word
        src = np.maximum(src, np.arange(-M, 0)[:, None] + l[None, :])                      # the tail must still fit
        past_end = src >= l[None, :]
        l_new = (~past_end).sum(0)
        last_inserted = ~orig[np.maximum(l_new - 1, 0), np.arange(B)]                      # never end on an inserted word
        l_new = l_new - ((l_new > 1) & last_inserted)
        src = np.minimum(src, l[None, :] - 1)
        out = np.take_along_axis(x, src, 0)
        out = np.where(orig, out, new_words)
        out = np.where(past_end, self.eos_index, out)
        return out[:l_new.max()], l_new

    def dropout_apply(self, x, l, u_keep, u_repeat, pick):
        """Drop (u_keep < dropout_prob) / repeat words, re-append <eos>, re-pad (noise.py:137-194).
        pick(n) -> index in [0,n) used when every word of a sentence was dropped."""
        M = self.max_length
        B = x.shape[1]
        keep = u_keep >= self.dropout_prob
        adv = np.concatenate([np.zeros((1, B), dtype=np.int64), np.cumsum(u_repeat >= self.repeat_prob, 0)], 0)
        # whole-batch form of the reference's per-sentence loop (same draws, same results; the loop costs 3 ms at 512
        # sentences while the GPU waits for these inputs)
        S = x.shape[0]
        cols = np.arange(B)
        n = np.where(x[l - 1, cols] == self.eos_index, l - 1, l)                          # words before the final <eos>
        km = keep[:S] & (np.arange(S)[:, None] < n[None, :])
        order = np.argsort(~km, axis=0, kind="stable")                                     # kept positions first, in order
        kept = np.take_along_axis(x, order, 0)
        ksz = km.sum(0)
        for i in np.nonzero(ksz == 0)[0]:                                                  # every word dropped: keep one
            kept[0, i] = x[pick(n[i]), i]
            ksz[i] = 1
        idx = np.maximum(adv, np.arange(-M, 0)[:, None] + ksz[None, :])                    # [M,B], non-decreasing per column
        cnt = (idx < ksz[None, :]).sum(0)
        out = np.take_along_axis(kept, np.minimum(idx, ksz[None, :] - 1)[:, :], 0) if S >= 1 else kept
        x2 = np.full((M + 1, B), self.pad_index, dtype=np.int64)
        x2[:M] = out
        x2[np.arange(M + 1)[:, None] >= cnt[None, :]] = self.pad_index
        add_eos = (cnt < M) & (x2[cnt - 1, cols] != self.eos_index)
        x2[cnt[add_eos], cols[add_eos]] = self.eos_index
        l2 = cnt + add_eos
        return np.ascontiguousarray(x2[:l2.max()]), l2.astype(np.int64)

    def shuffle_apply(self, x, l, noise):
        """Local shuffle: sort positions by index + noise (noise.py:196-230)."""
        out = x.copy()
        for i in range(x.shape[1]):
            s = 1 if self.keep_go else 0
            t = l[i] - 1 if self.keep_eos else l[i]
            perm = (np.arange(t - s) + noise[s:t, i]).argsort()
            out[s:t, i] = x[s:t, i][perm]
        return out, l

    # ---- device path ---------------------------------------------------
    # True: the device path whenever the drop/repeat stage is on (the configurations of the learnable model); "always": also
    # without it (blank / add only); False: always the host (numpy) path, whose draws follow the reference's numpy stream.
    # The fixed-width execution (ops.FIXED_WIDTH, graph replay) implies "always": it cannot leave the device.
    device_side = True

    def _forward_device(self, words, lengths, rev_domain):
        """Same stages, same functions of the draws (`ops.noise_tokens` == `*_apply`, bit-exact given the draws), but the
        draws come from torch's device generator instead of numpy's global one: no device->host copy of the batch, no
        host arithmetic while the GPU waits, one read-back (width and total of the new lengths)."""
        from . import ops
        dev = words.device
        S, B = words.shape
        M = self.max_length
        kw = {}
        if self.blank_prob != 0:
            assert 0 < self.blank_prob <= 1
            kw.update(u_blank=torch.rand(S, B, dtype=torch.float64, device=dev), blank_prob=self.blank_prob)
        if self.add_prob != 0:
            assert self.add_prob < 1
            dist = self._checked_cnt()[rev_domain.to(self.cnt.device)].to(dev)
            kw.update(u_add=torch.rand(M - 1, B, dtype=torch.float64, device=dev), add_prob=self.add_prob,
                      new_words=torch.multinomial(dist, M, True).transpose(0, 1).contiguous())
        if self.dropout_prob != 0 or self.repeat_prob != 0:
            assert self.dropout_prob < 1 and self.repeat_prob < 1
            assert not self.keep_go and self.keep_eos
            kw.update(u_keep=torch.rand(M, B, dtype=torch.float64, device=dev), u_rep=torch.rand(M - 1, B, dtype=torch.float64, device=dev),
                      u_pick=torch.rand(B, dtype=torch.float64, device=dev), dropout_prob=self.dropout_prob, repeat_prob=self.repeat_prob)
        out, new_len = ops.noise_tokens(words, lengths, M, pad=self.pad_index, unk=self.blank_index, eos=self.eos_index,
                                        keep_go=self.keep_go, keep_eos=self.keep_eos, **kw)
        if ops.FIXED_WIDTH:                     # graph replay: keep the full width (new_len <= M), read nothing back
            return out[:M], new_len
        width, total = torch.stack((new_len.max(), new_len.sum())).tolist()
        new_len.nast_total = int(total)
        return out[:width], new_len

    # ---- reference call signature -------------------------------------------------------------
    def forward(self, words, lengths, rev_domain):
        """words [seq,B] int64, lengths [B], rev_domain [B] -> (noisy words [seq',B], lengths [B]) on words.device.
        Stage order and draw order follow noise.py:60-74 (shuffle, blank, add, dropout)."""
        from . import ops
        dev = words.device
        if self.device_side and words.is_cuda and self.shuffle_weight == 0 and (
                self.dropout_prob != 0 or self.repeat_prob != 0 or self.device_side == "always" or ops.FIXED_WIDTH):
            return self._forward_device(words, lengths, rev_domain)
        if ops.FIXED_WIDTH:
            raise NotImplementedError("NoiseLayer: the fixed-width (graph) execution needs the device path "
                                      "(CUDA input, word_shuffle == 0)")
        x = words.detach().cpu().numpy()
        l = lengths.detach().cpu().numpy().astype(np.int64)
        M = self.max_length
        if self.shuffle_weight != 0:
            assert self.shuffle_weight > 1
            x, l = self.shuffle_apply(x, l, np.random.uniform(0, self.shuffle_weight, size=x.shape))
        if self.blank_prob != 0:
            assert 0 < self.blank_prob <= 1
            x, l = self.blank_apply(x, l, np.random.rand(*x.shape))
        if self.add_prob != 0:
            assert self.add_prob < 1
            u = np.random.rand(M - 1, x.shape[1])
            dist = self._checked_cnt()[rev_domain.to(self.cnt.device)]
            new_words = torch.multinomial(dist, M, True).transpose(0, 1).cpu().numpy()
            x, l = self.add_apply(x, l, u, new_words)
        if self.dropout_prob != 0 or self.repeat_prob != 0:
            assert self.dropout_prob < 1 and self.repeat_prob < 1
            assert not self.keep_go and self.keep_eos
            u_keep = np.random.rand(M, x.shape[1])
            u_rep = np.random.rand(M - 1, x.shape[1])
            x, l = self.dropout_apply(x, l, u_keep, u_rep, lambda n: np.random.randint(0, n))
        lengths = torch.from_numpy(l).to(dev)
        lengths.nast_total = int(l.sum())       # known on the host: lets the packed-row layers skip a device read
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev), lengths
