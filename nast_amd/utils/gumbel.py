"""utils/gumbel.py call signatures (reference: utils/gumbel.py:9-69) on the fused HIP row kernel.

Noise: the reference draws `torch.empty_like(logits).exponential_()`.  Production calls use the kernel's
Philox generator (keyed by torch's CUDA generator seed/offset); parity tests inject the Exp(1) draw with
`with inject_noise([E0, E1, ...]):` -- each gumbel call pops one tensor."""
import contextlib

import torch

from .. import ops

_injected = []


@contextlib.contextmanager
def inject_noise(draws):
    """Queue of Exp(1) tensors consumed, in call order, by the gumbel functions inside the block."""
    global _injected
    saved, _injected = _injected, list(draws)
    try:
        yield
    finally:
        _injected = saved


def _next_noise():
    return _injected.pop(0) if _injected else None


def _last_dim(dim, t):
    if dim not in (-1, t.dim() - 1):
        raise NotImplementedError("the HIP gumbel kernels reduce over the last dimension only")


def gumbel_softmax(logits, tau=1, dim=-1):
    _last_dim(dim, logits)
    return ops.gumbel_softmax_sample(logits, _next_noise(), None, tau)[0]


def gumbel_soft_and_index(logits, tau=1):
    """(y_soft, argmax) -- what Seq2seq.approx / get_lengths consume (seq2seq.py:407-414, :489-490)."""
    return ops.gumbel_softmax_sample(logits, _next_noise(), None, tau)


def _straight_through(y_soft, idx):
    y_hard = torch.zeros_like(y_soft).scatter_(-1, idx.unsqueeze(-1), 1.0)
    return y_hard - y_soft.detach() + y_soft


def gumbel_max(logits, tau=1, dim=-1):
    """-> (ret, y_soft): ret is the straight-through one-hot, y_soft the soft sample (utils/gumbel.py:35-44)."""
    _last_dim(dim, logits)
    y_soft, idx = ops.gumbel_softmax_sample(logits, _next_noise(), None, tau)
    return _straight_through(y_soft, idx), y_soft


def gumbel_max_with_mask(logits, mask, tau=1, dim=-1):
    _last_dim(dim, logits)
    y_soft, idx = ops.gumbel_softmax_sample(logits, _next_noise(), mask, tau)
    return _straight_through(y_soft, idx)


def straight_max(logits, dim=-1, allow_gradient=False):
    _last_dim(dim, logits)
    y_soft, idx = ops.gumbel_softmax_sample(logits, None, None, 1.0, True)
    y_hard = torch.zeros_like(y_soft).scatter_(-1, idx.unsqueeze(-1), 1.0)
    if allow_gradient:
        y_hard = y_hard - y_soft.detach() + y_soft
    return y_hard, y_soft


def onehot(sent, vocab_size):
    return torch.zeros(*sent.shape, vocab_size, device=sent.device).scatter_(-1, sent.unsqueeze(-1), 1.0)


# Exported by the reference's utils/__init__.py:17 but never called by styletransformer:
# kept as thin pass-throughs so that `from utils import ...` keeps working; no kernels behind them.
def cdist(x1, x2, p=2):
    return torch.cdist(x1, x2, p)


def recenter_gradient(*a, **k):
    raise NotImplementedError("recenter_gradient is unused by the NAST-StyTrans training step (out of scope)")


class RebarGradient(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("RebarGradient is unused by the NAST-StyTrans training step (out of scope)")
