"""ctypes loader of the plain-C alignment oracle (oracle/align_viterbi.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_build", "liboracle_align.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", SO, os.path.join(_HERE, "align_viterbi.c")])
    return SO


def viterbi_align(cost0, cost, olen, tlen):
    if not os.path.exists(SO):
        build()
    lib = ctypes.CDLL(SO)
    cost0 = np.ascontiguousarray(cost0, dtype=np.float32)
    cost = np.ascontiguousarray(cost, dtype=np.float32)
    olen = np.ascontiguousarray(olen, dtype=np.int64)
    tlen = np.ascontiguousarray(tlen, dtype=np.int64)
    B, T, S = cost.shape
    out = np.empty((B, T + 1), dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.oracle_align_viterbi(p(cost0), p(cost), p(olen), p(tlen), p(out), B, T, S)
    assert rc == 0
    return out
