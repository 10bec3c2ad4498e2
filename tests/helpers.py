"""Shared test scaffolding: a fake data manager + args Storage for the product modules."""
import numpy as np
import torch

from nast_amd.utils import Storage


class FakeDM:
    pad_id, go_id, eos_id, unk_id = 0, 1, 2, 3

    def __init__(self, vocab, max_length, seed=17, n_sent=50):
        rng = np.random.default_rng(seed)
        self.frequent_vocab_size = vocab
        self._sents = {}
        for dom in (0, 1):
            self._sents[dom] = [[1] + [int(x) for x in rng.integers(4, vocab, size=int(rng.integers(2, max_length)))] + [2]
                                for _ in range(n_sent)]

    def get_all_batch(self, key):
        return {"sent": self._sents[int(key[-1])]}


def make_param(*, V, d, h, N, L, learnable, disc="Cond", device="cpu", **over):
    a = Storage()
    a.num_styles, a.num_layers, a.d_model, a.max_length, a.h, a.dropout = 2, N, d, L, h, 0
    a.learned_pos_embed, a.use_learnable, a.decoder_input, a.z_decode_mode = True, learnable, "encoded", "max"
    a.discriminator_method = disc
    a.num_classes = 3 if disc == "Multi" else 2
    if learnable:
        a.word_blank, a.word_dropout, a.word_repeat, a.word_add, a.word_shuffle = 0.1, 0.05, 0, 1 / 0.95 - 1, 0
    else:
        a.word_blank, a.word_dropout, a.word_repeat, a.word_add, a.word_shuffle = 0.2, 0, 0, 0, 0
    a.gen_approx_mode = a.dis_approx_mode = "gumbel"
    a.slf_factor = a.cyc_factor = 0.5
    a.length_factor, a.fw_adv_factor, a.bw_adv_factor = 1.0, 1.5, 1.0
    a.lr_F = a.lr_D = 1e-4
    a.L2, a.radam, a.pre_z_batch, a.F_pretrain_iter = 0, False, 0, 0
    a.temperature_config = [(1, 0)]
    a.batch_size, a.iter_D, a.iter_F, a.log_steps = 2, 10, 5, 5
    a.cuda = device
    for k, v in over.items():
        a[k] = v
    p = Storage()
    p.args = a
    p.volatile = Storage()
    p.volatile.dm = FakeDM(V, L)
    return p


def cfg_of(g):
    kw = dict(V=int(g["cfg/V"]), d=int(g["cfg/d"]), h=int(g["cfg/h"]), N=int(g["cfg/N"]), L=int(g["cfg/L"]))
    if "cfg/learnable" in g:
        kw["learnable"] = bool(g["cfg/learnable"])
    if "cfg/disc" in g:
        kw["disc"] = str(g["cfg/disc"])
    return kw


def load_weights(module, arrays, prefix=""):
    """Copy golden arrays {state_dict key: ndarray} into a module; every key must exist with the same shape."""
    sd = module.state_dict()
    for k, v in arrays.items():
        k = prefix + k
        assert k in sd, f"unexpected key {k}"
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(np.asarray(v)))
    return set(sd) - {prefix + k for k in arrays}


def synth_batch(rng, B, L, V, full_first=True):
    lens = rng.integers(3, L + 1, size=B)
    if full_first:
        lens[0] = L
    sent = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = int(lens[b])
        sent[b, :n - 1] = rng.integers(4, V, size=n - 1)
        sent[b, n - 1] = 2
    return torch.tensor(sent), torch.tensor(lens, dtype=torch.long)
