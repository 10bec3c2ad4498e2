"""ctypes binding of libnast_hip.so (the C-ABI in include/nast_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel reports an error the
call raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or `make -C nast_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NAST_HIP_LIB") or os.path.join(_HERE, "lib", "libnast_hip.so")   # (override: lab builds, tools/*_stamps*)

GEMM_MAX_GROUP = 8
GEMM_MAX_SRC = 3
GEMM_RELU, GEMM_ATOMIC, GEMM_ACCUM, GEMM_BF16 = 1, 2, 4, 8

c_fp = C.c_void_p   # all device pointers travel as void*


class GemmDesc(C.Structure):
    _fields_ = [("A", c_fp * GEMM_MAX_SRC), ("B", c_fp * GEMM_MAX_SRC), ("C", c_fp), ("bias", c_fp),
                ("residual", c_fp), ("relu_src", c_fp), ("colsum", c_fp), ("ln_gamma", c_fp), ("ln_beta", c_fp),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("transA", C.c_int32), ("transB", C.c_int32), ("nsrc", C.c_int32), ("flags", C.c_int32),
                ("alpha", C.c_float)]


_DECLAYER_FIELDS = ["ln0_w", "ln0_b", "sa_wq", "sa_bq", "sa_wk", "sa_bk", "sa_wv", "sa_bv", "sa_wo", "sa_bo",
                    "ln1_w", "ln1_b", "ca_wq", "ca_bq", "ca_wk", "ca_bk", "ca_wv", "ca_bv", "ca_wo", "ca_bo",
                    "ln2_w", "ln2_b", "ff_w1", "ff_b1", "ff_w2", "ff_b2"]


class DecLayerWeights(C.Structure):
    _fields_ = [(n, c_fp) for n in _DECLAYER_FIELDS]


class LPDecodeArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "S1", "d", "h", "dff", "L", "nlayers")] + \
               [("layers", C.POINTER(DecLayerWeights))] + \
               [(n, c_fp) for n in ("norm_w", "norm_b", "wq_w", "wq_b", "wk_w", "wk_b", "lin_w", "lin_b", "pos_w", "eos", "unk",
                                    "memory", "origin_length", "allpos", "workspace", "sample_u")]


class Rows(C.Structure):
    _fields_ = [("ld", C.c_int64), ("grp", C.c_int64), ("grp_stride", C.c_int64)]


_i, _i64, _f, _u64, _d = C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_double

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/nast_hip.h one to one
SIGNATURES = {
    "nast_last_error": [],
    "nast_version": [],
    "nast_gemm_f32": [C.POINTER(GemmDesc), _i, c_fp],
    "nast_layernorm_fwd_f32": [c_fp] * 6 + [_i, _i, _f, _i, _i, _i, c_fp],
    "nast_layernorm_bwd_f32": [c_fp] * 9 + [_i, _i, _i, _i, _i, _i, c_fp],
    "nast_mha_fwd_f32": [c_fp] * 6 + [_i] * 9 + [_i64] * 6 + [_f, c_fp],
    "nast_mha_bwd_f32": [c_fp] * 10 + [_i] * 9 + [_i64] * 6 + [_f, c_fp],
    "nast_mha_fwd_varlen_f32": [c_fp] * 5 + [_i] * 9 + [c_fp] * 4 + [_i] * 5 + [_f, c_fp],
    "nast_mha_bwd_varlen_f32": [c_fp] * 9 + [_i] * 9 + [c_fp] * 4 + [_i] * 5 + [_f, c_fp],
    "nast_log_softmax_fwd_f32": [c_fp, Rows, c_fp, Rows, _i, _i, _f, c_fp],
    "nast_log_softmax_bwd_f32": [c_fp, Rows, c_fp, Rows, c_fp, Rows, _i, _i, _f, c_fp],
    "nast_gumbel_softmax_fwd_f32": [c_fp, Rows, c_fp, Rows, c_fp, c_fp, Rows, c_fp, _i, _i, _f, _u64, _u64, _i, c_fp],
    "nast_softmax_bwd_f32": [c_fp, Rows, c_fp, Rows, c_fp, Rows, _i, _i, _f, c_fp],
    "nast_first_eos_length_i64": [c_fp, c_fp, _i, _i, _i64, c_fp],
    "nast_embed_fwd_f32": [c_fp, _i, _i, _i, _i] + [c_fp] * 8 + [c_fp],
    "nast_embed_bwd_f32": [c_fp, _i, _i, _i, _i] + [c_fp] * 8 + [c_fp],
    "nast_align_cost_f32": [c_fp, _i, _i, c_fp, c_fp, c_fp, c_fp, _i64, _i64, c_fp, c_fp, _i, _i, _i, c_fp],
    "nast_align_viterbi_f32": [c_fp] * 5 + [_i, _i, _i, c_fp],
    "nast_align_lse_fwd_f32": [c_fp] * 4 + [_f, c_fp, c_fp, _i, _i, _i, c_fp],
    "nast_align_lse_bwd_f32": [c_fp] * 4 + [_f] + [c_fp] * 5 + [_i, _i, _i, c_fp],
    "nast_pack_index_i64": [c_fp, _i, _i, _i, _i] + [c_fp] * 8 + [c_fp],
    "nast_pad_mask_u8": [c_fp, c_fp, _i, _i, _i, c_fp],
    "nast_rows_gather_f32": [c_fp, _i64, c_fp, _i, _i, _i, c_fp, _i64, c_fp],
    "nast_rows_unpack_f32": [c_fp, _i64, _i, c_fp, c_fp, _i, _i, _i, c_fp, _i64, c_fp],
    "nast_noise_i64": [c_fp, c_fp, _i, _i, _i, c_fp, _d, c_fp, c_fp, _d, c_fp, c_fp, c_fp, _d, _d, _i64, _i64, _i64, _i, _i, c_fp, c_fp, c_fp],
    "nast_lp_gather_fwd_f32": [c_fp] * 7 + [_i] * 6 + [c_fp],
    "nast_lp_gather_bwd_f32": [c_fp] * 7 + [_i] * 7 + [c_fp],
    "nast_lp_step_f32": [c_fp] * 16 + [_i] * 6 + [c_fp],
    "nast_nll_gather_f32": [c_fp, Rows, c_fp, c_fp, _i, _i, c_fp],
    "nast_nll_log_softmax_bwd_f32": [c_fp, c_fp, Rows, c_fp, c_fp, Rows, _i, _i, _f, c_fp],
    "nast_lse_nll_f32": [c_fp, Rows, c_fp, c_fp, c_fp, _i, _i, _f, c_fp],
    "nast_nll_logits_bwd_f32": [c_fp, c_fp, Rows, c_fp, c_fp, c_fp, Rows, _i, _i, _f, c_fp],
    "nast_lp_logits_fwd_f32": [c_fp] * 5 + [_i] * 4 + [c_fp],
    "nast_lp_logits_bwd_f32": [c_fp] * 8 + [_i] * 4 + [c_fp],
    "nast_lp_decode_workspace_bytes": [_i] * 6,
    "nast_lp_decode_f32": [C.POINTER(LPDecodeArgs), C.POINTER(C.c_int), c_fp],
    "nast_lp_decode_static_f32": [C.POINTER(LPDecodeArgs), C.POINTER(C.c_int), c_fp],
    "nast_lp_decode_cluster_status": [C.POINTER(C.c_int)],
    "nast_multi_sqnorm_f32": [c_fp, c_fp, _i, c_fp, c_fp, c_fp, _i, _i, c_fp],
    "nast_multi_adam_f32": [c_fp] * 5 + [_i, c_fp] + [_f] * 8 + [c_fp, c_fp, c_fp, _i, _i, c_fp],
    "nast_prof_begin": [_i],
    "nast_prof_end": [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "nast_prof_get": [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)],
}
_RESTYPES = {"nast_last_error": C.c_char_p, "nast_lp_decode_workspace_bytes": C.c_int64}

_lib = None


class NastHipError(RuntimeError):
    pass


def load():
    """Load the library once; raises NastHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NastHipError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
            "nast_amd has no CPU or PyTorch fallback")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def call(name, *args):
    lib = _lib or load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise NastHipError(f"{name} failed ({rc}): {lib.nast_last_error().decode()}")
