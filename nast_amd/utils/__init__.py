"""Minimal counterpart of the reference's `utils` package: only what the generator training step needs
(/root/reference/utils/__init__.py exports)."""
from .storage import Storage
from .cuda_helper import cuda, zeros, ones, Tensor, LongTensor
from .cuda_helper import init as cuda_init
from .network_helper import BaseNetwork, BaseModule
from .operator_helper import generateMask
from .gumbel import (gumbel_max, gumbel_max_with_mask, gumbel_softmax, straight_max, onehot, cdist,
                     recenter_gradient, RebarGradient, inject_noise)
from .model_helper import BaseModel

__all__ = ["Storage", "cuda", "zeros", "ones", "Tensor", "LongTensor", "cuda_init", "BaseNetwork", "BaseModule",
           "generateMask", "gumbel_max", "gumbel_max_with_mask", "gumbel_softmax", "straight_max", "onehot", "cdist",
           "recenter_gradient", "RebarGradient", "inject_noise", "BaseModel"]
