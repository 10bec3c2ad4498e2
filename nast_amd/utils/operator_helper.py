import numpy as np

from .cuda_helper import Tensor


def generateMask(seqlen, length, type=int, device=None):
    """[seqlen, B] mask, 1 where position < length (utils/operator_helper.py:99-101)."""
    return Tensor((np.arange(seqlen)[:, None] < np.asarray(length)[None, :]).astype(type), device=device)
