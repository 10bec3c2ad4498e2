"""nast_amd -- MI355X-native (gfx950) implementation of the NAST-StyTrans generator training step.

Python modules mirror the reference's surface (transformer.py / seq2seq.py / utils/gumbel.py); the
arithmetic runs in hand-written HIP kernels behind the C-ABI of include/nast_hip.h.
"""
__version__ = "0.1.0"
