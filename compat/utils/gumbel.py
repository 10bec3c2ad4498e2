"""`utils.gumbel` (utils/gumbel.py:9-69 call signatures) -> nast_amd.utils.gumbel."""
from nast_amd.utils.gumbel import *       # noqa: F401,F403
from nast_amd.utils.gumbel import (gumbel_softmax, gumbel_max, gumbel_max_with_mask, straight_max, onehot, cdist,  # noqa: F401
                                   recenter_gradient, RebarGradient, inject_noise)
