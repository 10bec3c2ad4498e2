"""torch.autograd.Function wrappers over libnast_hip.so.

PyTorch is used here for device memory (torch.empty), the current HIP stream and autograd bookkeeping only;
every numeric operation is a hand-written gfx950 kernel reached through the C-ABI (nast_amd/_lib.py).
There is no CPU / eager fallback: a non-CUDA tensor raises.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
import contextlib

from ._lib import GemmDesc, Rows, GEMM_RELU, GEMM_ATOMIC, GEMM_ACCUM, GEMM_BF16, call

F32 = torch.float32


# ------------------------------------------------------------------------------------------------
# plumbing
# ------------------------------------------------------------------------------------------------
def _p(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_index = None


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~9 us of Python per call (3.3 ms per
    step at 64 sentences, where the step is bound by the host); the raw accessor is a plain C call."""
    global _dev_index
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()      # one process drives one GPU (bench.py / torchrun set it first)
    return _raw_stream(_dev_index)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.NastHipError("nast_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != F32:
            raise _lib.NastHipError(f"fp32 tensor expected, got {t.dtype}")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _flat_rows(n_rows, ld):
    return Rows(ld, 0, 0)


# GEMM precision of the module-level path: "fp32" (v_mfma_f32_32x32x2_f32, the reference's dtype and the default) or
# "bf16" (operands rounded to bf16 in registers, v_mfma_f32_32x32x16_bf16, fp32 accumulate; LayerNorm, softmax,
# attention, log-sum-exp statistics and the alignment DP stay fp32 -- north_star's "bf16 variant within 2e-2").
_GEMM_FLAGS = 0


def set_gemm_precision(name):
    global _GEMM_FLAGS
    if name not in ("fp32", "bf16"):
        raise ValueError("gemm precision must be 'fp32' or 'bf16'")
    _GEMM_FLAGS = GEMM_BF16 if name == "bf16" else 0


@contextlib.contextmanager
def gemm_precision(name):
    global _GEMM_FLAGS
    saved = _GEMM_FLAGS
    set_gemm_precision(name)
    try:
        yield
    finally:
        _GEMM_FLAGS = saved


def desc(A, B, Cm, M, N, K, lda, ldb, ldc, transA, transB, bias=None, residual=None, relu_src=None, colsum=None,
         flags=0, alpha=1.0):
    """Build one nast_gemm_desc. A / B may be a tensor or a list of up to 3 tensors (summed products)."""
    d = GemmDesc()
    As = A if isinstance(A, (list, tuple)) else (A,)
    Bs = B if isinstance(B, (list, tuple)) else (B,)
    for i, (a, b) in enumerate(zip(As, Bs)):
        d.A[i] = a.data_ptr()
        d.B[i] = b.data_ptr()
    d.nsrc = len(As)
    d.C = Cm.data_ptr()
    d.bias = _p(bias)
    d.residual = _p(residual)
    d.relu_src = _p(relu_src)
    d.colsum = _p(colsum)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.transA, d.transB = transA, transB
    d.flags = flags | _GEMM_FLAGS
    d.alpha = alpha
    return d


def gemm(descs):
    """Launch up to 8 GEMMs as one grouped kernel."""
    for i in range(0, len(descs), _lib.GEMM_MAX_GROUP):
        chunk = descs[i:i + _lib.GEMM_MAX_GROUP]
        arr = (GemmDesc * len(chunk))(*chunk)
        call("nast_gemm_f32", arr, len(chunk), _stream())


# descriptor shorthands for the three layouts of a Linear layer: y[M,N] = x[M,K] @ W[N,K]^T
def d_fwd(x2, W, y2, bias=None, residual=None, relu=False):
    M, K = x2.shape
    N = W.shape[0]
    return desc(x2, W, y2, M, N, K, K, K, N, 0, 1, bias=bias, residual=residual, flags=GEMM_RELU if relu else 0)


def d_dgrad(dy2_list, W_list, dx2, relu_src=None, accumulate=False):
    """dx[M,K] (+)= sum_i dy_i[M,N] @ W_i[N,K]"""
    dy0, W0 = dy2_list[0], W_list[0]
    M, N = dy0.shape
    K = W0.shape[1]
    return desc(list(dy2_list), list(W_list), dx2, M, K, N, N, K, K, 0, 0, relu_src=relu_src, flags=GEMM_ACCUM if accumulate else 0)


def d_wgrad(dy2, x2, dW, db=None):
    """dW[N,K] += dy[M,N]^T @ x[M,K]  (atomic, split-K), db[N] += colsum(dy)"""
    assert dW is not None
    M, N = dy2.shape
    K = x2.shape[1]
    return desc(dy2, x2, dW, N, K, M, N, K, K, 1, 0, colsum=db, flags=GEMM_ATOMIC)


def wgrad_descs(dy2, x2, dW, db):
    """[] when the weight is frozen (a frozen weight with a live bias falls back to a column sum)."""
    if dW is not None:
        return [d_wgrad(dy2, x2, dW, db)]
    if db is not None:
        db += dy2.sum(0)
    return []


def zeros_like_many(tensors):
    """One zero-filled flat buffer carved into views shaped like `tensors` (a single fill kernel)."""
    sizes = [t.numel() for t in tensors]
    # keep every view 16-B aligned
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 3) // 4 * 4
    flat = torch.zeros(tot, device=tensors[0].device, dtype=F32)
    return [flat[o:o + n].view(t.shape) for o, n, t in zip(offs, sizes, tensors)]


# ------------------------------------------------------------------------------------------------
# parameter gradients are accumulated IN PLACE
# ------------------------------------------------------------------------------------------------
# The weight-gradient GEMMs, LayerNorm / embedding backward kernels accumulate with fp32 atomics, so for a leaf
# parameter they add straight into `param.grad` (created zero-filled on first use) and the autograd Function
# returns None for it: no per-parameter zero-fill, no `grad += new` kernel for the second backward of f_step
# (seq2seq.py:464, :530) and no index_put through a [V, d] temporary for single embedding rows.
# `grad_listener` (set by nast_amd.ddp.GradReducer) is told about every in-place contribution, because
# autograd's post-accumulate hooks do not fire for gradients that never pass through AccumulateGrad.
# `sink_hook` (same owner) is asked BEFORE an in-place kernel is enqueued and may hand out another buffer to add into (a
# parameter whose gradient bucket is already being all-reduced).  Every parameter that is handed a sink is marked
# `_nast_touched`: nast_amd.optim.FusedAdam steps only the parameters that took part since its zero_grad(), as
# torch.optim.Adam skips `grad is None` -- the persistent, zero-filled buffers used here would otherwise be stepped.
INPLACE_GRADS = True
grad_listener = None
sink_hook = None
# Streams whose backward nodes add into the same .grad buffers (nast_amd.seq2seq: main + second stream).  A buffer that
# is created lazily is zero-filled on ONE of them; the others must not start adding before that fill has run.
peer_streams = ()


def _sink(p):
    if not INPLACE_GRADS or p is None or not p.is_leaf:
        return None
    p._nast_touched = True
    if sink_hook is not None:
        other = sink_hook(p)
        if other is not None:
            return other
    if p.grad is None:
        p.grad = torch.zeros_like(p)
        if p.is_cuda and peer_streams:
            cur = torch.cuda.current_stream(p.device)
            for s in peer_streams:
                if s != cur and s.device == p.device:
                    s.wait_stream(cur)
                    p.grad.record_stream(s)
    return p.grad


def live(*params):
    """Record at FORWARD time which parameters take a gradient (frozen ones -> None), like ctx.needs_input_grad."""
    return tuple(p if (p is not None and p.requires_grad) else None for p in params)


def grad_buffers(params):
    """For each entry of `params` (tensor or None): (buffer to accumulate into or None when no gradient is needed,
    value to return from backward).  Leaves get their .grad, everything else a slice of one zero-filled buffer."""
    bufs, rets, fresh = [None] * len(params), [None] * len(params), []
    for i, p in enumerate(params):
        if p is None:
            continue
        sk = _sink(p)
        if sk is not None:
            bufs[i] = sk
        else:
            fresh.append(i)
    if fresh:
        for i, z in zip(fresh, zeros_like_many([params[i] for i in fresh])):
            bufs[i] = rets[i] = z
    return bufs, rets


def grads_written(params):
    if grad_listener is not None:
        for p in params:
            if p is not None and p.is_leaf and INPLACE_GRADS:
                grad_listener(p)


# ------------------------------------------------------------------------------------------------
# raw kernel calls (no autograd)
# ------------------------------------------------------------------------------------------------
def ln_fwd(x2, gamma, beta, rows_out=None, grp=0, grp_stride=0, grp_off=0, eps=1e-6):
    rows_in, d = x2.shape
    rows_out = rows_in if rows_out is None else rows_out
    y = torch.empty(rows_out, d, device=x2.device, dtype=F32)
    mean = torch.empty(rows_out, device=x2.device, dtype=F32)
    rstd = torch.empty(rows_out, device=x2.device, dtype=F32)
    call("nast_layernorm_fwd_f32", _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows_out, d, eps,
         grp, grp_stride, grp_off, _stream())
    return y, mean, rstd


def ln_bwd(dy2, x2, gamma, mean, rstd, dgamma, dbeta, dx_add=None, grp=0, grp_stride=0, grp_off=0):
    rows_in, d = x2.shape
    dx = torch.empty_like(x2)
    call("nast_layernorm_bwd_f32", _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta),
         _p(dx_add), dy2.shape[0], rows_in, d, grp, grp_stride, grp_off, _stream())
    return dx


def _mask_args(mask, B, nq, nk):
    """bool mask broadcastable to [B, h, nq, nk] with size-1 head dim -> (ptr holder, batch stride, query stride)."""
    if mask is None:
        return None, 0, 0
    if mask.dtype != torch.bool and mask.dtype != torch.uint8:
        mask = mask != 0
    if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[3] != nk or mask.shape[0] not in (1, B) or mask.shape[2] not in (1, nq):
        raise _lib.NastHipError(f"unsupported attention mask shape {tuple(mask.shape)} for B={B} nq={nq} nk={nk}")
    if mask.stride(3) != 1 and nk > 1:
        mask = mask.contiguous()
    bs = mask.stride(0) if mask.shape[0] > 1 else 0
    qs = mask.stride(2) if mask.shape[2] > 1 else 0
    return mask, bs, qs


def mha_fwd(q, k, v, mask, h):
    """q [B,nq,D], k/v [B,nk,D] (last dim contiguous, any row/batch strides) -> out [B,nq,D], lse [B,h,nq]"""
    B, nq, D = q.shape
    nk = k.shape[1]
    dk = D // h
    out = torch.empty(B, nq, D, device=q.device, dtype=F32)
    lse = torch.empty(B, h, nq, device=q.device, dtype=F32)
    m, bs, qs = _mask_args(mask, B, nq, nk)
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    call("nast_mha_fwd_f32", _p(q), _p(k), _p(v), _p(m), _p(out), _p(lse), B, h, nq, nk, dk,
         q.stride(1), k.stride(1), v.stride(1), D, q.stride(0), k.stride(0), v.stride(0), nq * D, bs, qs,
         1.0 / math.sqrt(dk), _stream())
    return out, lse


def mha_bwd(q, k, v, mask, h, out, dout, lse):
    B, nq, D = q.shape
    nk = k.shape[1]
    dk = D // h
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    m, bs, qs = _mask_args(mask, B, nq, nk)
    call("nast_mha_bwd_f32", _p(q), _p(k), _p(v), _p(m), _p(out), _p(dout), _p(lse), _p(dq), _p(dk_), _p(dv),
         B, h, nq, nk, dk, D, D, D, D, nq * D, nk * D, nk * D, nq * D, bs, qs, 1.0 / math.sqrt(dk), _stream())
    return dq, dk_, dv


def rows_of(t):
    """nast_rows for a tensor whose last dim is the (contiguous) vocabulary axis: [R,V] or [B,L,V] views."""
    if t.stride(-1) != 1:
        raise _lib.NastHipError("vocabulary axis must be contiguous")
    if t.dim() == 2:
        return Rows(t.stride(0), 0, 0), t.shape[0]
    if t.dim() == 3:
        return Rows(t.stride(1), t.shape[1], t.stride(0)), t.shape[0] * t.shape[1]
    raise _lib.NastHipError(f"row kernels take 2-D or 3-D tensors, got {t.dim()}-D")


# ------------------------------------------------------------------------------------------------
# autograd Functions: fine-grained (public module API) ...
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b  [+ residual] [ReLU]   (nn.Linear, transformer.py:280-281, :309-312, :390-392)"""

    @staticmethod
    def forward(ctx, x, W, b, residual, relu):
        _chk(x, W, b, residual)
        x2 = _c(x).view(-1, x.shape[-1])
        W0 = W
        W = _c(W)
        y2 = torch.empty(x2.shape[0], W.shape[0], device=x.device, dtype=F32)
        r2 = _c(residual).view(-1, W.shape[0]) if residual is not None else None
        gemm([d_fwd(x2, W, y2, bias=b, residual=r2, relu=relu)])
        ctx.save_for_backward(x2, W, y2 if relu else None)
        ctx.params = live(W0, b)
        ctx.has_res, ctx.relu, ctx.xshape = residual is not None, relu, x.shape
        return y2.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W, y2 = ctx.saved_tensors
        pW, pb = ctx.params
        dy2 = _c(dy).view(-1, W.shape[0])
        if ctx.relu:
            dy2 = dy2 * (y2 > 0)   # only on the fine-grained path; the fused FFN masks inside the dgrad epilogue
        (dW, db), (rW, rb) = grad_buffers([pW, pb])
        descs = []
        if dW is not None:
            descs.append(d_wgrad(dy2, x2, dW, db))
        elif db is not None:
            db += dy2.sum(0)
        dx2 = None
        if ctx.needs_input_grad[0]:
            dx2 = torch.empty_like(x2)
            descs.append(d_dgrad([dy2], [W], dx2))
        if descs:
            gemm(descs)
        grads_written([pW, pb])
        return (dx2.view(ctx.xshape) if dx2 is not None else None, rW, rb, dy if ctx.has_res else None, None)


def linear(x, W, b=None, residual=None, relu=False):
    return LinearFn.apply(x, W, b, residual, relu)


class MatmulNNFn(torch.autograd.Function):
    """y[M,N] = x[M,K] @ E[K,N]   (soft tokens x embedding table, transformer.py:229, :403)"""

    @staticmethod
    def forward(ctx, x, E, holder=None):
        _chk(x, E)
        ctx.holder = holder          # SharedKVFn holder of x: its consumers' gradients meet in one buffer (see PackedSoft)
        x2 = _c(x).view(-1, x.shape[-1])
        E0 = E
        E = _c(E)
        M, K = x2.shape
        N = E.shape[1]
        y2 = torch.empty(M, N, device=x.device, dtype=F32)
        gemm([desc(x2, E, y2, M, N, K, K, N, N, 0, 0)])
        ctx.save_for_backward(x2, E)
        ctx.params = live(E0)
        ctx.xshape = x.shape
        return y2.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, E = ctx.saved_tensors
        M, K = x2.shape
        N = E.shape[1]
        dy2 = _c(dy).view(M, N)
        descs, dx2 = [], None
        if ctx.needs_input_grad[0]:
            hold = ctx.holder
            if hold is None:
                dx2 = torch.empty_like(x2)                       # dX = dY E^T
                descs.append(desc(dy2, E, dx2, M, K, N, N, N, K, 0, 1))
            elif hold.buf is None:                               # first consumer to run backward: plain store
                hold.buf = torch.empty_like(x2)
                descs.append(desc(dy2, E, hold.buf, M, K, N, N, N, K, 0, 1))
            else:                                                # the others add in the GEMM epilogue
                descs.append(desc(dy2, E, hold.buf.view(M, K), M, K, N, N, N, K, 0, 1, flags=GEMM_ACCUM))
        (dE,), (rE,) = grad_buffers(list(ctx.params))
        if dE is not None:                                   # dE += X^T dY
            descs.append(desc(x2, dy2, dE, K, N, M, K, N, N, 1, 0, flags=GEMM_ATOMIC))
        if descs:
            gemm(descs)
        grads_written(ctx.params)
        return (dx2.view(ctx.xshape) if dx2 is not None else None), rE, None


def matmul_nn(x, E, holder=None):
    return MatmulNNFn.apply(x, E, holder)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(d, eps=1e-6) (transformer.py:348-350) with an optional output-row map."""

    @staticmethod
    def forward(ctx, x, gamma, beta, grp, grp_stride, grp_off, rows_out):
        _chk(x, gamma, beta)
        x2 = _c(x).view(-1, x.shape[-1])
        y, mean, rstd = ln_fwd(x2, gamma, beta, rows_out if grp else None, grp, grp_stride, grp_off)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.params = live(gamma, beta)
        ctx.map = (grp, grp_stride, grp_off)
        ctx.xshape = x.shape
        return y if grp else y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd = ctx.saved_tensors
        (dg, db), (rg, rb) = ln_param_buffers(ctx.params[0], ctx.params[1], gamma)
        dy2 = _c(dy).view(-1, x2.shape[1])
        dx = ln_bwd(dy2, x2, gamma, mean, rstd, dg, db, None, *ctx.map)
        grads_written(ctx.params)
        return dx.view(ctx.xshape), rg, rb, None, None, None, None


def ln_param_buffers(gamma, beta, like):
    """Buffers the LayerNorm backward kernel accumulates the parameter gradients into (None = frozen parameter: the
    kernel takes NULL and skips that sum)."""
    return grad_buffers([gamma, beta])


def layer_norm(x, gamma, beta):
    return LayerNormFn.apply(x, gamma, beta, 0, 0, 0, 0)


def layer_norm_rows(x, gamma, beta, grp, grp_stride, grp_off, rows_out):
    """LayerNorm of selected rows only: output row r <- input row (r//grp)*grp_stride + grp_off + r%grp."""
    return LayerNormFn.apply(x, gamma, beta, grp, grp_stride, grp_off, rows_out)


# ------------------------------------------------------------------------------------------------
# host work deferred into the shadow of the greedy decode
# ------------------------------------------------------------------------------------------------
# The greedy skip decoding is ~800 tiny dependent launches: ~9 ms during which the GPU is almost idle and the host, which
# needs the generated lengths next, can only wait.  With ASYNC_DECODE the decode is enqueued without its early-exit
# read-backs, and `run_deferred()` -- called by LengthPredictor right after that enqueue -- runs a callable the training
# step left in `defer`: the step uses it to issue the whole self-reconstruction pass on the main stream, so the decode
# (side stream) executes beside real work and is finished by the time the host asks for the lengths.
ASYNC_DECODE = False
_deferred = None


class async_decode:
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global ASYNC_DECODE, _deferred
        self.prev = (ASYNC_DECODE, _deferred)
        ASYNC_DECODE, _deferred = True, self.fn
        return self

    def __exit__(self, *exc):
        global ASYNC_DECODE, _deferred
        pending = _deferred
        ASYNC_DECODE, _deferred = self.prev
        if pending is not None and exc[0] is None:
            pending()                      # nobody reached the decode (e.g. no length predictor): run it now


def run_deferred():
    global _deferred
    fn, _deferred = _deferred, None
    if fn is not None:
        with torch.enable_grad():
            fn()


# ------------------------------------------------------------------------------------------------
# packed rows: the position-wise work of a ragged batch without its padding
# ------------------------------------------------------------------------------------------------
# A [B, n, d] batch of sentences of lengths len[b] <= n spends n - len[b] rows per sentence on padding that every
# consumer masks (transformer.py:56-61) and that carries no loss and no gradient.  Encoder / Decoder stacks therefore
# run on the R = sum(len) real rows only ("packed": sentence b owns rows off[b] .. off[b]+len[b]-1): LayerNorm, the
# projections and the FFN are row-wise, attention uses the packed entry points (nast_mha_*_varlen_f32).  Results on
# real rows are the padded computation's; padded rows come back as zeros.  Enabled by Seq2seq for its training steps
# (PACK_ROWS); module-level calls keep the reference's padded behaviour, including the values at padded positions.
PACK_ROWS = False
PACK_MAX_LEN = 64            # limit of the packed attention kernels for any head size ...
PACK_MAX_LEN_LONG = 144      # ... and for head sizes 16 / 32 / 64 (the *_long kernels of attention_mfma.hip)


def pack_len_ok(width, dk):
    return width <= PACK_MAX_LEN or (width <= PACK_MAX_LEN_LONG and dk in (16, 32, 64))

# Fixed-shape execution (nast_amd/graph.py: HIP-graph replay of a training step).  The step normally trims tensors to the
# longest sentence of the batch -- NoiseLayer's output, the generated sentences -- which costs a device-to-host read each and
# makes every step a different sequence of launches.  With FIXED_WIDTH those tensors keep their maximum width (max_length;
# max_length + 1 decoder positions), nothing is read back, and the padding is masked by the lengths as it always is.
# GRAPH_NOISE: the Gumbel noise is drawn by torch (exponential_, whose generator state a captured graph advances on every
# replay) instead of the row kernel's own Philox stream, whose seed and offset are kernel arguments fixed at capture.
FIXED_WIDTH = False
GRAPH_NOISE = False


class fixed_width:
    """with ops.fixed_width(): ...   maximum-width tensors, no read-backs (see FIXED_WIDTH)."""

    def __init__(self, on=True, graph_noise=None):
        self.on, self.noise = bool(on), bool(on if graph_noise is None else graph_noise)

    def __enter__(self):
        global FIXED_WIDTH, GRAPH_NOISE
        self.prev = (FIXED_WIDTH, GRAPH_NOISE)
        FIXED_WIDTH, GRAPH_NOISE = self.on, self.noise
        return self

    def __exit__(self, *exc):
        global FIXED_WIDTH, GRAPH_NOISE
        FIXED_WIDTH, GRAPH_NOISE = self.prev


class packed_rows:
    """with ops.packed_rows(True): ...  switches the packed execution of Encoder / Decoder stacks on."""

    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        global PACK_ROWS
        self.prev, PACK_ROWS = PACK_ROWS, self.on

    def __exit__(self, *exc):
        global PACK_ROWS
        PACK_ROWS = self.prev


def set_total(lengths, total):
    """Remember sum(lengths) (known on the host) on the tensor, so that packing needs no device-to-host read."""
    lengths.nast_total = int(total)
    return lengths


NATIVE_PACK_INDEX = os.environ.get("NAST_PACK_TORCH") != "1"      # False: build the index maps with torch ops (what CPU tensors always use)
_pack_flags = {}              # device -> int32 [1], set by nast_pack_index_i64 when a host-side total was wrong (sticky)


def _pack_flag(dev):
    f = _pack_flags.get(dev)
    if f is None:
        f = _pack_flags[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return f


def check_packed():
    """Raise if any packed layout built since the last check was given a row total that disagrees with its lengths (its
    index maps were zero-filled).  One 4-byte read per device: the training steps call it where they synchronise anyway."""
    for dev, f in _pack_flags.items():
        if int(f.item()):
            f.zero_()
            raise _lib.NastHipError("PackedSeq: the host-side row total disagrees with the device lengths")


def check_decode():
    """Raise if a cluster skip decode (csrc/decode_cluster.hip) gave up at one of its barriers since the library was loaded:
    some workgroup of its grid was not resident (a CU mask, a partitioned device), the positions of that call are garbage.
    NAST_DECODE_CLUSTER=0 selects the step-by-step path.  One synchronous 4-byte read: called where the steps synchronise."""
    if not torch.cuda.is_available():
        return
    import ctypes as C
    st = C.c_int(0)
    call("nast_lp_decode_cluster_status", C.byref(st))
    if st.value:
        raise _lib.NastHipError("nast_lp_decode_f32: a cluster decode timed out at a barrier (grid not resident); "
                                "set NAST_DECODE_CLUSTER=0 for the step-by-step path")


class PackedSeq:
    ROW_MULTIPLE = 64        # packed buffers are padded to a multiple of this (weight-gradient GEMMs: K = rows, K % 32 == 0
    #                          keeps them on the fast path).  Padding rows duplicate row 0 on the way in, are kept finite in
    #                          the forward pass and carry exactly zero gradient in the backward pass.

    def __init__(self, lengths, width, total=None):
        """lengths [B] int64 (device), width = n of the padded layout, total = sum(lengths) if the host knows it."""
        self.B, self.width = lengths.shape[0], width
        if total is None:
            total = getattr(lengths, "nast_total", None)
        if total is None:
            total = int(lengths.sum())                       # device-to-host read
        self.total = total
        self.rows = (total + self.ROW_MULTIPLE - 1) // self.ROW_MULTIPLE * self.ROW_MULTIPLE
        dev = lengths.device
        self._tail = None
        if lengths.is_cuda and lengths.dtype == torch.int64 and 1 <= self.B <= 8192 and NATIVE_PACK_INDEX:
            # one launch instead of ~18 (cumsum, repeat_interleave, arange, gathers, cat): nast_pack_index_i64
            B, rows = self.B, self.rows
            i64 = torch.empty(B + 3 * total + rows, dtype=torch.int64, device=dev)
            i32 = torch.empty(2 * B, dtype=torch.int32, device=dev)
            self.off, self.b_of, self.pos = i64[:B], i64[B:B + total], i64[B + total:B + 2 * total]
            self.idx, self.idx_p = i64[B + 2 * total:B + 3 * total], i64[B + 3 * total:]
            self.off32, self.len32, self.bad = i32[:B], i32[B:], _pack_flag(dev)
            call("nast_pack_index_i64", _p(_c(lengths)), B, int(width), int(total), int(rows), _p(self.off), _p(self.off32), _p(self.len32),
                 _p(self.b_of), _p(self.pos), _p(self.idx), _p(self.idx_p), _p(self.bad), _stream())
            return
        off = torch.cumsum(lengths, 0) - lengths
        self.len32, self.off32 = lengths.to(torch.int32), off.to(torch.int32)
        self.off = off
        self.b_of = torch.repeat_interleave(torch.arange(self.B, device=dev), lengths, output_size=total)
        self.pos = torch.arange(total, device=dev) - off[self.b_of]
        self.idx = self.b_of * width + self.pos              # packed row -> row of the padded [B*width] layout
        self.idx_p = _pad_index(self.idx, self.rows)
        self._tail = None

    def tail(self):
        """The rows after each sentence's first one: (their packed row numbers, padded like every packed buffer, and a
        PackedSeq of lengths-1 / width-1 describing them)."""
        if self._tail is None:
            sub = PackedSeq(self.len32.to(torch.int64) - 1, self.width - 1, self.total - self.B)
            self._tail = (_pad_index(self.off[sub.b_of] + sub.pos + 1, sub.rows), sub)
        return self._tail


def _pad_index(idx, rows):
    if rows == idx.shape[0]:
        return idx
    return torch.cat((idx, idx.new_zeros(rows - idx.shape[0])))


class PackedPair:
    """query-side and key-side layouts of one attention call"""

    def __init__(self, q, k, causal=False):
        self.q, self.k, self.causal = q, k, causal


class PackedSoft:
    """Soft tokens [B, T, V] of a ragged batch held as packed rows: `rows` [ps.rows, V] (autograd tensor), sentence b's
    T_b = ps.len positions at rows ps.off[b]...  Quacks like the [B, T, V] tensor for the shape queries the model makes;
    consumers take the rows through `soft_embed` (x @ table on real rows only) and `column` (one vocabulary column)."""

    def __init__(self, rows, ps, unnormalised=False):
        # unnormalised: the rows are generator logits, not log-probabilities (GeneratorFn normalise=False): only a consumer
        # that is invariant under a shift of the row may read them as they are (Seq2seq._approx_packed)
        self.unnormalised = unnormalised
        # the rows feed two embedding tables (the generator's cycle pass and the discriminator, seq2seq.py:492-519): their
        # [rows, V] gradients meet in one buffer instead of a 350-MB autograd add (SharedKVFn)
        self.rows, self.ps = shared_kv(rows), ps
        self.shape = (ps.B, ps.width, rows.shape[-1])
        self.device, self.is_cuda = rows.device, rows.is_cuda

    def dim(self):
        return 3

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def soft_embed(self, table):
        """[B, T, d] = soft tokens @ table (zeros at the padded positions)"""
        return unpack_rows(matmul_nn(self.rows, table, getattr(self.rows, "nast_kv_holder", None)), self.ps)

    def column(self, j):
        """[B, T] weights of vocabulary entry j (no gradient), zeros at the padded positions"""
        return unpack_rows(self.rows.detach()[:, j:j + 1], self.ps).squeeze(-1)


def packed_of(mask, x, dk=None):
    """PackedSeq for a padding mask made by transformer._pad_mask, or None when packing does not apply.
    dk = head size of the attention that will read the rows (decides the length limit)."""
    if not PACK_ROWS or mask is None or not x.is_cuda:
        return None
    ps = getattr(mask, "nast_packed", None)
    if ps is not None:
        return ps
    src = getattr(mask, "nast_lengths", None)
    if src is None or not pack_len_ok(mask.shape[-1], dk):
        return None
    lengths, extra = src
    total = getattr(lengths, "nast_total", None)
    ps = PackedSeq(lengths + extra, mask.shape[-1], None if total is None else total + extra * lengths.shape[0])
    mask.nast_packed = ps
    return ps


def _rows_gather(x2, idx, n_idx, rows_out):
    """y[r] = x2[idx[r]] for r < n_idx, zeros up to rows_out (x2: 2-d, unit stride along the row)"""
    y = torch.empty(rows_out, x2.shape[1], device=x2.device, dtype=F32)
    call("nast_rows_gather_f32", _p(x2), x2.stride(0), _p(idx), n_idx, rows_out, x2.shape[1], _p(y), y.stride(0), _stream())
    return y


def _rows_unpack(xp, ps):
    """y[b * width + p] = xp[off[b] + p] for p < len[b], zeros elsewhere (xp: 2-d, unit stride along the row)"""
    y = torch.empty(ps.B * ps.width, xp.shape[1], device=xp.device, dtype=F32)
    call("nast_rows_unpack_f32", _p(xp), xp.stride(0), xp.shape[0], _p(ps.off), _p(ps.len32), ps.B, ps.width, xp.shape[1], _p(y), y.stride(0),
         _stream())
    return y


def _rows_2d(t):
    t = t.reshape(-1, t.shape[-1])
    return t if t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


class PackRowsFn(torch.autograd.Function):
    """index_select by ps.idx_p; the backward pass is the gather `unpack` instead of zeros + index_add"""

    @staticmethod
    def forward(ctx, x, ps):
        ctx.ps, ctx.xshape = ps, x.shape
        return _rows_gather(_rows_2d(x), ps.idx_p, ps.rows, ps.rows)

    @staticmethod
    def backward(ctx, g):
        return _rows_unpack(_rows_2d(g), ctx.ps).view(ctx.xshape), None


class UnpackRowsFn(torch.autograd.Function):
    """packed rows to their [B, width] positions, zeros elsewhere, in one launch; backward = gather by ps.idx, zero tail"""

    @staticmethod
    def forward(ctx, xp, ps):
        ctx.ps, ctx.rows = ps, xp.shape[0]
        return _rows_unpack(_rows_2d(xp), ps).view(ps.B, ps.width, xp.shape[-1])

    @staticmethod
    def backward(ctx, g):
        ps = ctx.ps
        return _rows_gather(_rows_2d(g), ps.idx, ps.total, ctx.rows), None


def pack_rows(x, ps):
    """[B, n, d] -> [ps.rows, d]  (ps.total real rows, then copies of row 0 as padding)"""
    if x.is_cuda and x.dtype == F32:
        return PackRowsFn.apply(x, ps)
    return x.reshape(-1, x.shape[-1]).index_select(0, ps.idx_p)


def unpack_rows(xp, ps):
    """[ps.rows, d] -> [B, n, d], zeros at the padded positions"""
    if xp.is_cuda and xp.dtype == F32:
        return UnpackRowsFn.apply(xp, ps)
    out = torch.zeros(ps.B * ps.width, xp.shape[-1], device=xp.device, dtype=xp.dtype)
    return out.index_copy(0, ps.idx, xp[:ps.total]).view(ps.B, ps.width, xp.shape[-1])


def mha_fwd_packed(q, k, v, pair, h):
    """q [Rq, D], k/v [Rk, D] packed -> out [Rq, D], lse [B, h, max_nq]"""
    D = q.shape[1]
    dk = D // h
    B = pair.q.B
    out = torch.empty_like(q)
    lse = torch.empty(B, h, pair.q.width, device=q.device, dtype=F32)
    call("nast_mha_fwd_varlen_f32", _p(q), _p(k), _p(v), _p(out), _p(lse), B, h, pair.q.width, pair.k.width, dk, D, D, D, D,
         _p(pair.q.off32), _p(pair.k.off32), _p(pair.q.len32), _p(pair.k.len32), int(pair.causal),
         pair.q.total, q.shape[0], pair.k.total, k.shape[0], 1.0 / math.sqrt(dk), _stream())
    return out, lse                                # (padding rows, which belong to no sentence, come back as zeros)


def mha_bwd_packed(q, k, v, pair, h, dout, lse, out=None):
    """out: the forward result [Rq, D] when the caller still has it (the kernel then takes delta from dO * O and runs
    its dQ and dK / dV halves side by side)."""
    D = q.shape[1]
    dk = D // h
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    call("nast_mha_bwd_varlen_f32", _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(dq), _p(dk_), _p(dv), pair.q.B, h, pair.q.width,
         pair.k.width, dk, D, D, D, D, _p(pair.q.off32), _p(pair.k.off32), _p(pair.q.len32), _p(pair.k.len32),
         int(pair.causal), pair.q.total, q.shape[0], pair.k.total, k.shape[0], 1.0 / math.sqrt(dk), _stream())
    return dq, dk_, dv                             # (padding rows get exactly zero gradient, written by the kernel)


class AttentionFn(torch.autograd.Function):
    """scaled_attention on [B,n,h*dk] projections (transformer.py:296-303 + the transposes of :287-292)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, h):
        _chk(q, k, v)
        q, k, v = _c(q), _c(k), _c(v)
        out, lse = mha_fwd(q, k, v, mask, h)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.mask, ctx.h = mask, h
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk_, dv = mha_bwd(q, k, v, ctx.mask, ctx.h, out, _c(dout), lse)
        return dq, dk_, dv, None, None


def attention(q, k, v, mask, h):
    return AttentionFn.apply(q, k, v, mask, h)


# ------------------------------------------------------------------------------------------------
# ... and fused sublayers (what the Encoder/Decoder layers actually run)
# ------------------------------------------------------------------------------------------------
class _KVHolder:
    """Gradient of one encoder memory tensor, shared by the cross-attention sublayers of all decoder layers that read it."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class SharedKVFn(torch.autograd.Function):
    """Identity on the encoder memory that collects its gradient.  Every decoder layer's cross-attention reads the same memory
    (transformer.py:256), so autograd would receive one [rows, d] gradient per layer and sum them with N - 1 full-size adds
    (49 per f_step).  Instead the sublayers' dgrad GEMMs write / accumulate (NAST_GEMM_ACCUM epilogue) into ONE buffer kept
    on the holder and return no gradient of their own; this node runs once all of them have (the autograd engine counts an
    undefined gradient as a satisfied dependency) and hands the buffer on."""

    @staticmethod
    def forward(ctx, memory, holder):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return memory.view_as(memory)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.holder.buf = ctx.holder.buf, None
        if g is not None:                     # a consumer that is not a cross-attention sublayer
            buf = g if buf is None else buf.add_(g.reshape(buf.shape))
        return buf, None


def shared_kv(memory):
    """-> the memory to hand to a stack of decoder layers (see SharedKVFn); the tensor itself when no gradient is needed."""
    if not (torch.is_grad_enabled() and memory.requires_grad and memory.is_cuda):
        return memory
    holder = _KVHolder()
    m = SharedKVFn.apply(memory, holder)
    m.nast_kv_holder = holder
    return m


class AttnSublayerFn(torch.autograd.Function):
    """x + fc(MHA(LN(x) Wq, kv Wk, kv Wv))  -- SublayerConnection around MultiHeadAttention
    (transformer.py:324-326 + :284-294).  kv_src None = self-attention on LN(x); otherwise cross-attention
    on `kv_src` (the encoder memory, not normalised, transformer.py:256)."""

    @staticmethod
    def forward(ctx, x, kv_src, mask, h, ln_w, ln_b, Wq, bq, Wk, bk, Wv, bv, Wo, bo, kv_holder=None):
        _chk(x, kv_src, ln_w, ln_b, Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        ctx.kv_holder = kv_holder
        packed = isinstance(mask, PackedPair)         # x [Rq, d] (and kv_src [Rk, d]) are packed rows
        d = x.shape[-1]
        x2 = _c(x).view(-1, d)
        z, mean, rstd = ln_fwd(x2, ln_w, ln_b)
        self_attn = kv_src is None
        kv2 = z if self_attn else _c(kv_src).view(-1, d)
        rq, rk = x2.shape[0], kv2.shape[0]
        q = torch.empty(rq, d, device=x.device, dtype=F32)
        k = torch.empty(rk, d, device=x.device, dtype=F32)
        v = torch.empty(rk, d, device=x.device, dtype=F32)
        gemm([d_fwd(z, Wq, q, bias=bq), d_fwd(kv2, Wk, k, bias=bk), d_fwd(kv2, Wv, v, bias=bv)])
        if packed:
            a, lse = mha_fwd_packed(q, k, v, mask, h)
        else:
            B, nq = x.shape[0], x.shape[1]
            nk = nq if self_attn else kv_src.shape[1]
            a, lse = mha_fwd(q.view(B, nq, d), k.view(B, nk, d), v.view(B, nk, d), mask, h)
        y = torch.empty_like(x2)
        gemm([d_fwd(a.view(-1, d), Wo, y, bias=bo, residual=x2)])
        ctx.save_for_backward(x2, z, mean, rstd, kv2 if not self_attn else None, q, k, v, a, lse,
                              ln_w, Wq, Wk, Wv, Wo)
        ctx.mask, ctx.h, ctx.self_attn, ctx.packed = mask, h, self_attn, packed
        ctx.xshape, ctx.kvshape = x.shape, (None if self_attn else kv_src.shape)
        ctx.params = live(ln_w, ln_b, Wq, bq, Wk, bk, Wv, bv, Wo, bo)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, z, mean, rstd, kv2, q, k, v, a, lse, ln_w, Wq, Wk, Wv, Wo = ctx.saved_tensors
        d = x2.shape[1]
        if kv2 is None:
            kv2 = z
        dy2 = _c(dy).view(-1, d)
        (_, _, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo), rets = grad_buffers(list(ctx.params))
        (dlnw, dlnb), _ = ln_param_buffers(ctx.params[0], ctx.params[1], ln_w)
        da = torch.empty_like(a).view(-1, d)
        gemm([d_dgrad([dy2], [Wo], da)] + wgrad_descs(dy2, a.view(-1, d), dWo, dbo))
        if ctx.packed:
            dq, dk_, dv = mha_bwd_packed(q, k, v, ctx.mask, ctx.h, da, lse, a.view(-1, d))
        else:
            B, nq = ctx.xshape[0], ctx.xshape[1]
            nk = nq if ctx.self_attn else ctx.kvshape[1]
            dq, dk_, dv = mha_bwd(q.view(B, nq, d), k.view(B, nk, d), v.view(B, nk, d), ctx.mask, ctx.h, a, da.view(B, nq, d), lse)
        dq2, dk2, dv2 = dq.view(-1, d), dk_.view(-1, d), dv.view(-1, d)
        dz = torch.empty_like(z)
        descs = wgrad_descs(dq2, z, dWq, dbq) + wgrad_descs(dk2, kv2, dWk, dbk) + wgrad_descs(dv2, kv2, dWv, dbv)
        dkv = None
        if ctx.self_attn:
            descs.append(d_dgrad([dq2, dk2, dv2], [Wq, Wk, Wv], dz))
        else:
            descs.append(d_dgrad([dq2], [Wq], dz))
            if ctx.needs_input_grad[1]:
                hold = ctx.kv_holder
                if hold is None:
                    dkv = torch.empty_like(kv2)
                    descs.append(d_dgrad([dk2, dv2], [Wk, Wv], dkv))
                elif hold.buf is None:                     # first layer to run backward: plain store into the shared buffer
                    hold.buf = torch.empty(ctx.kvshape, device=kv2.device, dtype=F32)
                    descs.append(d_dgrad([dk2, dv2], [Wk, Wv], hold.buf.view(-1, d)))
                else:                                      # the others add to it in the GEMM epilogue
                    descs.append(d_dgrad([dk2, dv2], [Wk, Wv], hold.buf.view(-1, d), accumulate=True))
        gemm(descs)
        dx = ln_bwd(dz, x2, ln_w, mean, rstd, dlnw, dlnb, dx_add=dy2)
        grads_written(ctx.params)
        return (dx.view(ctx.xshape), dkv.view(ctx.kvshape) if dkv is not None else None, None, None) + tuple(rets) + (None,)


class FFNSublayerFn(torch.autograd.Function):
    """x + W2 relu(W1 LN(x) + b1) + b2   (transformer.py:324-326 around :305-316)"""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, W1, b1, W2, b2):
        _chk(x, ln_w, ln_b, W1, b1, W2, b2)
        shape = x.shape
        d = shape[-1]
        x2 = _c(x).view(-1, d)
        z, mean, rstd = ln_fwd(x2, ln_w, ln_b)
        hmid = torch.empty(x2.shape[0], W1.shape[0], device=x.device, dtype=F32)
        gemm([d_fwd(z, W1, hmid, bias=b1, relu=True)])
        y = torch.empty_like(x2)
        gemm([d_fwd(hmid, W2, y, bias=b2, residual=x2)])
        ctx.save_for_backward(x2, z, mean, rstd, hmid, ln_w, W1, W2)
        ctx.params = live(ln_w, ln_b, W1, b1, W2, b2)
        ctx.shape = shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, z, mean, rstd, hmid, ln_w, W1, W2 = ctx.saved_tensors
        d = x2.shape[1]
        dy2 = _c(dy).view(-1, d)
        (_, _, dW1, db1, dW2, db2), rets = grad_buffers(list(ctx.params))
        (dlnw, dlnb), _ = ln_param_buffers(ctx.params[0], ctx.params[1], ln_w)
        dh = torch.empty_like(hmid)
        gemm([d_dgrad([dy2], [W2], dh, relu_src=hmid)] + wgrad_descs(dy2, hmid, dW2, db2))
        dz = torch.empty_like(z)
        gemm([d_dgrad([dh], [W1], dz)] + wgrad_descs(dh, z, dW1, db1))
        dx = ln_bwd(dz, x2, ln_w, mean, rstd, dlnw, dlnb, dx_add=dy2)
        grads_written(ctx.params)
        return (dx.view(ctx.shape),) + tuple(rets)


def attn_sublayer(x, kv_src, mask, h, ln, mha):
    p = mha.head_projs
    return AttnSublayerFn.apply(x, kv_src, mask, h, ln.weight, ln.bias, p[0].weight, p[0].bias, p[1].weight, p[1].bias,
                                p[2].weight, p[2].bias, mha.fc.weight, mha.fc.bias,
                                getattr(kv_src, "nast_kv_holder", None) if kv_src is not None else None)


def ffn_sublayer(x, ln, ffn):
    return FFNSublayerFn.apply(x, ln.weight, ln.bias, ffn.mlp[0].weight, ffn.mlp[0].bias, ffn.mlp[3].weight, ffn.mlp[3].bias)


# ------------------------------------------------------------------------------------------------
# generator / softmax family
# ------------------------------------------------------------------------------------------------
class GeneratorFn(torch.autograd.Function):
    """log_softmax((x W^T + b) / temperature)   (Generator.forward, transformer.py:213-214).
    normalise=False (temperature 1 only): the logits themselves.  For a consumer that is invariant under a shift of the row
    -- the Gumbel-softmax, softmax((log_probs + g) / tau) = softmax((logits + g) / tau) -- the log-sum-exp pass over
    [rows, V] and its backward (the identity on gradients that sum to zero over the row, which a softmax's are) drop out:
    SURVEY section 8f rank 1, the part of it that pays on this GPU."""

    @staticmethod
    def forward(ctx, x, W, b, temperature, normalise=True):
        _chk(x, W, b)
        assert normalise or temperature == 1.0
        ctx.normalise = normalise
        x2 = _c(x).view(-1, x.shape[-1])
        V = W.shape[0]
        y = torch.empty(x2.shape[0], V, device=x.device, dtype=F32)
        gemm([d_fwd(x2, W, y, bias=b)])
        r = Rows(V, 0, 0)
        if normalise:
            call("nast_log_softmax_fwd_f32", _p(y), r, _p(y), r, y.shape[0], V, 1.0 / temperature, _stream())
        ctx.save_for_backward(x2, W, y if normalise else None)
        ctx.params = live(W, b)
        ctx.it, ctx.xshape = 1.0 / temperature, x.shape
        return y.view(*x.shape[:-1], V)

    @staticmethod
    def backward(ctx, gy):
        x2, W, y = ctx.saved_tensors
        V = W.shape[0]
        gy2 = _c(gy).view(-1, V)
        if ctx.normalise:
            gl = torch.empty_like(y)
            r = Rows(V, 0, 0)
            call("nast_log_softmax_bwd_f32", _p(gy2), r, _p(y), r, _p(gl), r, y.shape[0], V, ctx.it, _stream())
        else:
            gl = gy2
        (dW, db), (rW, rb) = grad_buffers(list(ctx.params))
        dx = torch.empty_like(x2)
        gemm([d_dgrad([gl], [W], dx)] + wgrad_descs(gl, x2, dW, db))
        grads_written(ctx.params)
        return dx.view(ctx.xshape), rW, rb, None, None


class GeneratorNLLFn(torch.autograd.Function):
    """-log_softmax((x W^T + b) / temperature)[token]  per row: Generator (transformer.py:213-214) + NLLLoss(reduction
    ='none') (seq2seq.py:453, :504) as one node, so the gradient of the [rows, V] log-probabilities never exists."""

    @staticmethod
    def forward(ctx, x, W, b, temperature, tokens):
        _chk(x, W, b)
        x2 = _c(x).view(-1, x.shape[-1])
        V = W.shape[0]
        tokens = _c(tokens).view(-1)
        assert tokens.numel() == x2.shape[0] and tokens.dtype == torch.int64
        y = torch.empty(x2.shape[0], V, device=x.device, dtype=F32)
        gemm([d_fwd(x2, W, y, bias=b)])
        r = Rows(V, 0, 0)
        # the logits are read once for two floats per row (log-sum-exp, NLL); the log-probabilities are never written
        nll = torch.empty(x2.shape[0], device=x.device, dtype=F32)
        lse = torch.empty(x2.shape[0], device=x.device, dtype=F32)
        call("nast_lse_nll_f32", _p(y), r, _p(tokens), _p(lse), _p(nll), y.shape[0], V, 1.0 / temperature, _stream())
        ctx.save_for_backward(x2, W, y, tokens, lse)
        ctx.params = live(W, b)
        ctx.it, ctx.xshape = 1.0 / temperature, x.shape
        return nll.view(x.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        x2, W, y, tokens, lse = ctx.saved_tensors
        V = W.shape[0]
        g = _c(g).view(-1)
        r = Rows(V, 0, 0)
        gl = torch.empty_like(y)
        call("nast_nll_logits_bwd_f32", _p(g), _p(y), r, _p(lse), _p(tokens), _p(gl), r, y.shape[0], V, ctx.it, _stream())
        (dW, db), (rW, rb) = grad_buffers(list(ctx.params))
        dx = torch.empty_like(x2)
        gemm([d_dgrad([gl], [W], dx)] + wgrad_descs(gl, x2, dW, db))
        grads_written(ctx.params)
        return dx.view(ctx.xshape), rW, rb, None, None


def generator_nll(x, W, b, temperature, tokens):
    return GeneratorNLLFn.apply(x, W, b, float(temperature), tokens)


def generator(x, W, b, temperature, normalise=True):
    return GeneratorFn.apply(x, W, b, float(temperature), bool(normalise))


class LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x2 = _c(x).view(-1, x.shape[-1])
        V = x2.shape[1]
        y = torch.empty_like(x2)
        r = Rows(V, 0, 0)
        call("nast_log_softmax_fwd_f32", _p(x2), r, _p(y), r, y.shape[0], V, 1.0, _stream())
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        V = y.shape[1]
        gy2 = _c(gy).view(-1, V)
        gx = torch.empty_like(y)
        r = Rows(V, 0, 0)
        call("nast_log_softmax_bwd_f32", _p(gy2), r, _p(y), r, _p(gx), r, y.shape[0], V, 1.0, _stream())
        return gx.view(gy.shape)


def log_softmax(x):
    return LogSoftmaxFn.apply(x)


class GumbelSoftmaxFn(torch.autograd.Function):
    """y_soft = softmax((logits + g) / tau), g = min(-log(E), 1e10)   (utils/gumbel.py:28-30, :36-38).
    E = injected Exp(1) draw (parity) or None (in-kernel Philox keyed by torch's CUDA generator)."""

    @staticmethod
    def forward(ctx, logits, E, mask, tau, noise_off):
        _chk(logits, E)
        V = logits.shape[-1]
        lr, rows = rows_of(logits)
        y = torch.empty(logits.shape, device=logits.device, dtype=F32)
        yr, _ = rows_of(y)
        idx = torch.empty(logits.shape[:-1], device=logits.device, dtype=torch.int64)
        seed = offset = 0
        er = Rows(0, 0, 0)
        if E is not None:
            if E.shape != logits.shape:
                raise _lib.NastHipError("noise tensor must have the shape of logits")
            er, _ = rows_of(E)
        elif not noise_off and GRAPH_NOISE:
            E = torch.empty_like(logits, memory_format=torch.contiguous_format).exponential_()     # utils/gumbel.py:28
            er, _ = rows_of(E)
        elif not noise_off:
            gen = torch.cuda.default_generators[logits.device.index]
            seed, offset = gen.initial_seed(), gen.get_offset()
            gen.set_offset(offset + ((logits.numel() + 3) // 4 + 3) // 4 * 4)
        m = None
        if mask is not None:
            m = _c(mask.expand(logits.shape).to(torch.int64))
        call("nast_gumbel_softmax_fwd_f32", _p(logits), lr, _p(E), er, _p(m), _p(y), yr, _p(idx), rows, V, 1.0 / tau,
             seed & (2 ** 64 - 1), offset, 1 if noise_off else 0, _stream())
        ctx.save_for_backward(y)
        ctx.it = 1.0 / tau
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    def backward(ctx, gy, _gidx):
        (y,) = ctx.saved_tensors
        V = y.shape[-1]
        gy = _c(gy)
        gx = torch.empty_like(y)
        yr, rows = rows_of(y)
        call("nast_softmax_bwd_f32", _p(gy), yr, _p(y), yr, _p(gx), yr, rows, V, ctx.it, _stream())
        return gx, None, None, None, None


def gumbel_softmax_sample(logits, E=None, mask=None, tau=1.0, noise_off=False):
    """-> (y_soft, argmax index)"""
    return GumbelSoftmaxFn.apply(logits, E, mask, float(tau), noise_off)


def first_eos_length(tokens, eos):
    tokens = _c(tokens)
    out = torch.empty(tokens.shape[0], device=tokens.device, dtype=torch.int64)
    call("nast_first_eos_length_i64", _p(tokens), _p(out), tokens.shape[0], tokens.shape[1], int(eos), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# embedding assembly
# ------------------------------------------------------------------------------------------------
class EmbedFn(torch.autograd.Function):
    """[extra rows | token + position] -> [B, n_extra + L, d]  (transformer.py:63-65, :141-148, :225-231).
    tokens: int64 ids [B,L]  or  fp32 rows [B,L,d] already multiplied by the table (soft tokens)."""

    @staticmethod
    def forward(ctx, tokens, tok_w, pos_w, ew0, ei0, ew1, ei1):
        hard = tokens.dtype == torch.int64
        B, L = tokens.shape[0], tokens.shape[1]
        d = pos_w.shape[1]
        ne = (ew0 is not None) + (ew1 is not None)
        tokens = _c(tokens)
        out = torch.empty(B, ne + L, d, device=pos_w.device, dtype=F32)
        call("nast_embed_fwd_f32", _p(out), B, L, d, ne, _p(tokens) if hard else None, _p(tok_w) if hard else None,
             None if hard else _p(tokens), _p(pos_w), _p(ew0), _p(ei0), _p(ew1), _p(ei1), _stream())
        ctx.save_for_backward(tokens if hard else None, ei0, ei1)
        ctx.meta = (hard, B, L, d, ne)
        ctx.params = live(tok_w if hard else None, pos_w, ew0, ew1)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, ei0, ei1 = ctx.saved_tensors
        hard, B, L, d, ne = ctx.meta
        dout = _c(dout)
        (d_tokw, d_pos, d_e0, d_e1), (r_tokw, r_pos, r_e0, r_e1) = grad_buffers(list(ctx.params))
        d_rows = torch.empty(B, L, d, device=dout.device, dtype=F32) if not hard else None
        call("nast_embed_bwd_f32", _p(dout), B, L, d, ne, _p(ids), _p(d_tokw), _p(d_rows), _p(d_pos), _p(d_e0), _p(ei0),
             _p(d_e1), _p(ei1), _stream())
        grads_written(ctx.params)
        return (d_rows if not hard else None), r_tokw, r_pos, r_e0, None, r_e1, None


def embed_assemble(tokens, tok_w, pos_w, extra=()):
    """extra: up to two (weight [n,d], index [B] int64 or None) pairs placed in front of the tokens."""
    ew0 = ei0 = ew1 = ei1 = None
    if len(extra) > 0:
        ew0, ei0 = extra[0]       # the parameter itself ([n,d] table or a single [d] row): gradients go to its .grad
    if len(extra) > 1:
        ew1, ei1 = extra[1]
    return EmbedFn.apply(tokens, tok_w, pos_w, ew0, ei0, ew1, ei1)


# ------------------------------------------------------------------------------------------------
# alignment
# ------------------------------------------------------------------------------------------------
def align_costs(E, tgt_ids, src_ids=None, src_emb=None, src_unk_w=None, unk_id=3):
    """cost0 [B,T], cost [B,T,S]   (transformer.py:396-412)"""
    B, T = tgt_ids.shape
    S = src_ids.shape[1] if src_ids is not None else src_emb.shape[1]
    V, d = E.shape
    cost0 = torch.empty(B, T, device=E.device, dtype=F32)
    cost = torch.empty(B, T, S, device=E.device, dtype=F32)
    stride = 0
    if src_unk_w is not None:
        # a strided view like soft[..., unk]: element (b,s) at base + (b*S+s)*stride
        assert src_unk_w.dim() == 2 and src_unk_w.stride(0) == S * src_unk_w.stride(1)
        stride = src_unk_w.stride(1)
    call("nast_align_cost_f32", _p(_c(E)), V, d, _p(_c(tgt_ids)), _p(_c(src_ids)) if src_ids is not None else None,
         _p(_c(src_emb)) if src_emb is not None else None, _p(src_unk_w), stride, unk_id, _p(cost0), _p(cost), B, T, S, _stream())
    return cost0, cost


def align_viterbi(cost0, cost, origin_length, target_length):
    """allpos int64 [B,T+1]   (transformer.py:413-448), bit-exact given the same cost tables"""
    B, T, S = cost.shape
    allpos = torch.empty(B, T + 1, device=cost.device, dtype=torch.int64)
    call("nast_align_viterbi_f32", _p(_c(cost0)), _p(_c(cost)), _p(_c(origin_length)), _p(_c(target_length)), _p(allpos),
         B, T, S, _stream())
    return allpos


class AlignLSEFn(torch.autograd.Function):
    """Soft alignment value (extension; no counterpart in the reference)."""

    @staticmethod
    def forward(ctx, cost0, cost, origin_length, target_length, tau):
        cost0, cost = _c(cost0), _c(cost)
        B, T, S = cost.shape
        value = torch.empty(B, device=cost.device, dtype=F32)
        alpha = torch.empty(B, T, S, device=cost.device, dtype=F32)
        ol, tl = _c(origin_length), _c(target_length)
        call("nast_align_lse_fwd_f32", _p(cost0), _p(cost), _p(ol), _p(tl), tau, _p(value), _p(alpha), B, T, S, _stream())
        ctx.save_for_backward(cost0, cost, ol, tl, value, alpha)
        ctx.tau = tau
        return value

    @staticmethod
    def backward(ctx, dvalue):
        cost0, cost, ol, tl, value, alpha = ctx.saved_tensors
        B, T, S = cost.shape
        dc0, dc = torch.empty_like(cost0), torch.empty_like(cost)
        call("nast_align_lse_bwd_f32", _p(cost0), _p(cost), _p(ol), _p(tl), ctx.tau, _p(value), _p(alpha), _p(_c(dvalue)),
             _p(dc0), _p(dc), B, T, S, _stream())
        return dc0, dc, None, None, None


def align_lse(cost0, cost, origin_length, target_length, tau=1.0):
    return AlignLSEFn.apply(cost0, cost, origin_length, target_length, float(tau))


class TableRowFn(torch.autograd.Function):
    """weight[idx] as a [d] tensor (embed.token_embed.weight[eos_id], transformer.py:468-469); the gradient is added
    to row idx of weight.grad in place instead of going through a zero-filled [V, d] temporary."""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.params = live(weight)
        ctx.idx = int(idx)
        return weight.detach()[ctx.idx].clone()

    @staticmethod
    def backward(ctx, g):
        (dW,), (rW,) = grad_buffers(list(ctx.params))
        if dW is not None:
            dW[ctx.idx] += g
        grads_written(ctx.params)
        return rW, None


def table_row(weight, idx):
    return TableRowFn.apply(weight, idx)


class LPGatherFn(torch.autograd.Function):
    """LengthPredictor candidate gather (transformer.py:471-473, :488-491, :504-505)."""

    @staticmethod
    def forward(ctx, origin_input, origin_length, allpos, eos, unk, pos_w, shift, use_stay, pos_skip0):
        _chk(origin_input, eos, unk, pos_w)
        origin_input, allpos, origin_length = _c(origin_input), _c(allpos), _c(origin_length)
        eos, unk = _c(eos), _c(unk)
        B, S1, d = origin_input.shape
        T1 = allpos.shape[1]
        out = torch.empty(B, T1, d, device=origin_input.device, dtype=F32)
        call("nast_lp_gather_fwd_f32", _p(origin_input), _p(origin_length), _p(allpos), _p(eos), _p(unk), _p(pos_w), _p(out),
             B, S1, T1, d, shift, use_stay, _stream())
        ctx.save_for_backward(origin_length, allpos)
        ctx.meta = (B, S1, T1, d, shift, use_stay, pos_skip0, pos_w.shape if pos_w is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        origin_length, allpos = ctx.saved_tensors
        B, S1, T1, d, shift, use_stay, pos_skip0, pws = ctx.meta
        dout = _c(dout)
        dev = dout.device
        d_oin = torch.zeros(B, S1, d, device=dev, dtype=F32) if ctx.needs_input_grad[0] else None
        d_eos = torch.zeros(d, device=dev, dtype=F32)
        d_unk = torch.zeros(d, device=dev, dtype=F32)
        d_pos = torch.zeros(pws, device=dev, dtype=F32) if pws is not None else None
        call("nast_lp_gather_bwd_f32", _p(dout), _p(origin_length), _p(allpos), _p(d_oin), _p(d_eos), _p(d_unk), _p(d_pos),
             B, S1, T1, d, shift, use_stay, pos_skip0, _stream())
        return d_oin, None, None, d_eos, d_unk, d_pos, None, None, None


def lp_gather(origin_input, origin_length, allpos, eos, unk, pos_w=None, shift=0, use_stay=True, pos_skip0=True):
    return LPGatherFn.apply(origin_input, origin_length, allpos, eos, unk, pos_w, int(shift), int(use_stay), int(pos_skip0))


class LPLogitsFn(torch.autograd.Function):
    """Bilinear part of the skip logits (transformer.py:504-507) without the [B,T1,3,d] gather / einsum tensors."""

    @staticmethod
    def forward(ctx, qh, KC, kunk, allpos):
        _chk(qh, KC, kunk)
        qh, KC, kunk, allpos = _c(qh), _c(KC), _c(kunk), _c(allpos)
        B, T1, d = qh.shape
        NC = KC.shape[1]
        out = torch.empty(B, T1, 4, device=qh.device, dtype=F32)
        call("nast_lp_logits_fwd_f32", _p(qh), _p(KC), _p(kunk), _p(allpos), _p(out), B, T1, NC, d, _stream())
        ctx.save_for_backward(qh, KC, kunk, allpos)
        return out

    @staticmethod
    def backward(ctx, g):
        qh, KC, kunk, allpos = ctx.saved_tensors
        B, T1, d = qh.shape
        NC = KC.shape[1]
        g = _c(g)
        dqh = torch.empty_like(qh)
        dKC = torch.zeros_like(KC)
        dkunk = torch.zeros_like(kunk)
        call("nast_lp_logits_bwd_f32", _p(g), _p(qh), _p(KC), _p(kunk), _p(allpos), _p(dqh), _p(dKC), _p(dkunk), B, T1, NC, d,
             _stream())
        return dqh, dKC, dkunk, None


def lp_logits(qh, KC, kunk, allpos):
    return LPLogitsFn.apply(qh, KC, kunk, allpos)


# ------------------------------------------------------------------------------------------------
# NoiseLayer on the device (noise.py:60-194)
# ------------------------------------------------------------------------------------------------
def noise_tokens(x, lengths, max_length, *, u_blank=None, blank_prob=0.0, u_add=None, new_words=None, add_prob=0.0,
                 u_keep=None, u_rep=None, u_pick=None, dropout_prob=0.0, repeat_prob=0.0, pad, unk, eos,
                 keep_go=False, keep_eos=True):
    """x int64 [S,B] (time-major), lengths int64 [B]; float64 uniform draws per stage (None = stage off).
    -> (out int64 [max_length+1, B], new lengths int64 [B]); the caller cuts the width."""
    S, B = x.shape
    M = int(max_length)
    for t, shape in ((u_blank, (S, B)), (u_add, (M - 1, B)), (new_words, (M, B)), (u_keep, (M, B)), (u_rep, (M - 1, B)), (u_pick, (B,))):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or not t.is_cuda):
            raise _lib.NastHipError(f"noise_tokens: expected a contiguous device tensor of shape {shape}, got {tuple(t.shape)}")
    for t in (u_blank, u_add, u_keep, u_rep, u_pick):
        if t is not None and t.dtype != torch.float64:
            raise _lib.NastHipError("noise_tokens: uniform draws must be float64")
    x, lengths = _c(x), _c(lengths)
    out = torch.empty(M + 1, B, dtype=torch.long, device=x.device)
    out_len = torch.empty(B, dtype=torch.long, device=x.device)
    call("nast_noise_i64", _p(x), _p(lengths), S, B, M, _p(u_blank), float(blank_prob), _p(u_add), _p(new_words), float(add_prob),
         _p(u_keep), _p(u_rep), _p(u_pick), float(dropout_prob), float(repeat_prob), int(pad), int(unk), int(eos),
         int(bool(keep_go)), int(bool(keep_eos)), _p(out), _p(out_len), _stream())
    return out, out_len
