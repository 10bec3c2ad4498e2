"""GPU: each C-ABI kernel family against a plain fp32 PyTorch (CPU) statement of the same op.
Tolerance: fp32, 1e-4 relative (north_star) unless noted; index outputs exact."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs a GPU"
    from nast_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def close(a, b, rtol=1e-4, atol=1e-5):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=rtol, atol=atol)


def relclose(a, b, tol=1e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-30
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (70, 40, 32), (1088, 256, 256), (300, 1024, 256), (2112, 9600, 256), (33, 2, 32), (5, 7, 9)])
def test_gemm_linear_fwd_bwd(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    R = torch.randn(M, N, generator=g)
    xr, Wr, br, rr = (t.clone().requires_grad_(True) for t in (x, W, b, res))
    yref = torch.nn.functional.linear(xr, Wr, br) + rr
    (yref * R).sum().backward()
    xg, Wg, bg, rg = (dev(t).requires_grad_(True) for t in (x, W, b, res))
    y = ops.linear(xg, Wg, bg, residual=rg)
    (y * dev(R)).sum().backward()
    relclose(y, yref)
    relclose(xg.grad, xr.grad)
    relclose(Wg.grad, Wr.grad)
    relclose(bg.grad, br.grad)
    relclose(rg.grad, rr.grad)


@pytest.mark.parametrize("M", [1040, 2049, 9733])
@pytest.mark.parametrize("N,transB", [(128, 1), (256, 1), (1024, 1), (256, 0), (1024, 0)])
def test_gemm_weight_stationary(ops, M, N, transB, monkeypatch):
    """The (opt-in, NAST_GEMM_BSTAT=1) weight-stationary kernel (gemm_bstat.hip: K = 256, plain row-major A, N % 128 == 0, M > 1024) through the C-ABI:
    every epilogue it fuses (bias, residual, ReLU, ReLU mask), both weight layouts, row counts that are not a multiple of the
    16-row tile, a strided A, and a grouped launch of three problems -- against float64 on the CPU."""
    from nast_amd._lib import GEMM_RELU
    monkeypatch.setenv("NAST_GEMM_BSTAT", "1")
    K = 256
    g = torch.Generator().manual_seed(M + N + transB)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / 16                      # [N, K]
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    msk = torch.randn(M, N, generator=g)
    xd, bd, rd, md = dev(x), dev(b), dev(res), dev(msk)
    Bd = dev(W if transB else W.t().contiguous())                # transB=0: stored [K, N]
    ldb = K if transB else N
    ref = x.double() @ W.double().t()

    def run(**kw):
        y = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm([ops.desc(xd, Bd, y, M, N, K, K, ldb, N, 0, transB, **kw)])
        return y

    relclose(run(), ref)
    relclose(run(bias=bd), ref + b.double())
    relclose(run(bias=bd, residual=rd), ref + b.double() + res.double())
    relclose(run(bias=bd, flags=GEMM_RELU), torch.relu(ref + b.double()))
    relclose(run(relu_src=md), ref * (msk > 0))
    relclose(run(bias=bd, residual=rd, relu_src=md, alpha=0.5), (0.5 * ref + b.double() + res.double()) * (msk > 0))
    # A as a column block of a wider buffer (lda > K)
    wide = torch.randn(M, K + 64, generator=g)
    wd = dev(wide)
    y = torch.empty(M, N, device="cuda")
    ops.gemm([ops.desc(wd[:, 32:], Bd, y, M, N, K, K + 64, ldb, N, 0, transB)])
    relclose(y, wide[:, 32:32 + K].double() @ W.double().t())
    # three problems in one launch (QKV), different weights and outputs
    Ws = [torch.randn(N, K, generator=g) / 16 for _ in range(3)]
    ys = [torch.empty(M, N, device="cuda") for _ in range(3)]
    Wd = [dev(w if transB else w.t().contiguous()) for w in Ws]            # (kept alive: a descriptor holds a bare pointer)
    ops.gemm([ops.desc(xd, w, y_, M, N, K, K, ldb, N, 0, transB, bias=bd) for w, y_ in zip(Wd, ys)])
    for w, y_ in zip(Ws, ys):
        relclose(y_, x.double() @ w.double().t() + b.double())


def test_gemm_in_place_residual_with_long_k(ops):
    """y = x W^T + b + y with the residual aliasing the output (an in-place residual update) at a K long enough for the
    split-K plan (>= 2048): a cut output is zero-filled before its slices are added, which would erase the residual, so such a
    problem must not be cut.  Next to it the same product with a separate residual (which IS cut)."""
    M, N, K = 96, 64, 4096
    g = torch.Generator().manual_seed(9)
    x, W, b, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 64, torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = x.double() @ W.double().t() + b.double() + r.double()
    xd, Wd, bd = dev(x), dev(W), dev(b)
    y = dev(r).clone()
    ops.gemm([ops.d_fwd(xd, Wd, y, bias=bd, residual=y)])
    relclose(y, ref)
    y2 = torch.empty(M, N, device="cuda")
    ops.gemm([ops.d_fwd(xd, Wd, y2, bias=bd, residual=dev(r))])
    relclose(y2, ref)


@pytest.mark.parametrize("M,N,K", [(64, 1024, 1024), (40, 1024, 4096), (64, 4096, 1024), (7, 1056, 1040), (96, 3072, 1024), (130, 520, 512)])
def test_gemm_small_m_wide_model(ops, M, N, K):
    """The small-M path's 16x16-tile kernel (every K >= 512 and a multiple of 16, at most 512 tiles of 32x32: the decode of a
    d = 1024 model at a small batch): bias, residual, ReLU, an in-place residual, a grouped launch, and a LayerNorm of the A rows
    fused into the product (K <= 1024) -- against float64 on the CPU."""
    g = torch.Generator().manual_seed(M + N + K)
    x, W, b, r = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = x.double() @ W.double().t()
    xd, Wd, bd, rd = dev(x), dev(W), dev(b), dev(r)

    def run(ln=None, **kw):
        y = torch.full((M, N), float("nan"), device="cuda")
        d = ops.d_fwd(xd, Wd, y, **kw)
        if ln is not None:
            d.ln_gamma, d.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
        ops.gemm([d])
        return y

    relclose(run(), ref)
    relclose(run(bias=bd, residual=rd), ref + b.double() + r.double())
    relclose(run(bias=bd, relu=True), torch.relu(ref + b.double()))
    y = rd.clone()
    ops.gemm([ops.d_fwd(xd, Wd, y, bias=bd, residual=y)])
    relclose(y, ref + b.double() + r.double())
    if N * ((M + 31) // 32) * 3 <= 512 * 32:
        Ws = [torch.randn(N, K, generator=g) / math.sqrt(K) for _ in range(3)]
        Wds = [dev(w) for w in Ws]
        ys = [torch.empty(M, N, device="cuda") for _ in range(3)]
        ops.gemm([ops.d_fwd(xd, w, y_, bias=bd) for w, y_ in zip(Wds, ys)])
        for w, y_ in zip(Ws, ys):
            relclose(y_, x.double() @ w.double().t() + b.double())
    if K <= 1024 and K % 16 == 0:
        gam, bet = torch.randn(K, generator=g), torch.randn(K, generator=g)
        xs = x * 3 + 0.5
        xd = dev(xs)
        xn = torch.nn.functional.layer_norm(xs.double(), (K,), gam.double(), bet.double(), 1e-6)
        relclose(run(ln=(dev(gam), dev(bet)), bias=bd, residual=rd), xn @ W.double().t() + b.double() + r.double())


def test_gemm_relu_and_matmul_nn(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(130, 48, generator=g)
    W = torch.randn(72, 48, generator=g) / 7
    b = torch.randn(72, generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    R = torch.randn(130, 72, generator=g)
    yref = torch.relu(torch.nn.functional.linear(xr, Wr, br))
    (yref * R).sum().backward()
    xg, Wg, bg = (dev(t).requires_grad_(True) for t in (x, W, b))
    y = ops.linear(xg, Wg, bg, relu=True)
    (y * dev(R)).sum().backward()
    relclose(y, yref); relclose(xg.grad, xr.grad); relclose(Wg.grad, Wr.grad); relclose(bg.grad, br.grad)
    # soft tokens x table
    s = torch.softmax(torch.randn(6, 11, 40, generator=g), -1)
    E = torch.randn(40, 32, generator=g)
    sr, Er = s.clone().requires_grad_(True), E.clone().requires_grad_(True)
    R = torch.randn(6, 11, 32, generator=g)
    (torch.matmul(sr, Er) * R).sum().backward()
    sg, Eg = dev(s).requires_grad_(True), dev(E).requires_grad_(True)
    y = ops.matmul_nn(sg, Eg)
    (y * dev(R)).sum().backward()
    relclose(y, torch.matmul(s, E)); relclose(sg.grad, sr.grad); relclose(Eg.grad, Er.grad)


@pytest.mark.parametrize("rows,d", [(7, 32), (1088, 256), (100, 1024), (9, 48),
                                    # more rows than one pass of the backward grid holds (4096 x rows-in-flight per wavefront), ragged last pass
                                    (13001, 256), (9001, 512), (4100, 1024), (12290, 64)])
def test_layernorm(ops, rows, d):
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, d, generator=g) * 2 + 0.5
    w = torch.rand(d, generator=g) + 0.5
    b = torch.randn(d, generator=g)
    R = torch.randn(rows, d, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yref = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-6)
    (yref * R).sum().backward()
    xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
    y = ops.layer_norm(xg, wg, bg)
    (y * dev(R)).sum().backward()
    relclose(y, yref); relclose(xg.grad, xr.grad); relclose(wg.grad, wr.grad); relclose(bg.grad, br.grad)


def test_layernorm_row_map(ops):
    g = torch.Generator().manual_seed(5)
    B, n, d = 5, 9, 64
    x = torch.randn(B, n, d, generator=g)
    w, b = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    R = torch.randn(B, n - 1, d, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yref = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-6)[:, 1:]
    (yref * R).sum().backward()
    xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
    y = ops.layer_norm_rows(xg, wg, bg, n - 1, n, 1, B * (n - 1)).view(B, n - 1, d)
    (y * dev(R)).sum().backward()
    relclose(y, yref); relclose(xg.grad, xr.grad); relclose(wg.grad, wr.grad); relclose(bg.grad, br.grad)


def ref_attention(q, k, v, mask, h):
    B, nq, D = q.shape
    dk = D // h
    qh, kh, vh = (t.view(B, -1, h, dk).transpose(1, 2) for t in (q, k, v))
    s = qh.matmul(kh.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        s = s.masked_fill(mask, float("-inf"))
    return torch.softmax(s, -1).matmul(vh).transpose(1, 2).contiguous().view(B, nq, D)


@pytest.mark.parametrize("B,nq,nk,h,dk,kind", [(3, 7, 9, 4, 8, "pad"), (4, 17, 17, 4, 64, "pad"), (2, 33, 33, 4, 64, "causal"),
                                                (2, 33, 34, 4, 64, "pad"), (2, 70, 70, 2, 64, "causal"), (1, 130, 130, 2, 64, "pad"),
                                                (2, 1, 12, 4, 16, "none"), (2, 9, 9, 2, 32, "causal"),
                                                (2, 40, 20, 4, 16, "pad"), (3, 64, 64, 2, 64, "pad"), (2, 20, 50, 2, 8, "pad"), (2, 48, 49, 2, 32, "none"),
                                                # 65 .. 144 positions: the *_long kernels (K/V resident, Q/dO in blocks of 64 queries)
                                                (3, 129, 129, 2, 64, "pad"), (2, 100, 130, 2, 32, "pad"), (2, 80, 80, 2, 16, "causal"),
                                                (2, 144, 144, 1, 64, "none"), (2, 65, 129, 2, 64, "causal"), (2, 130, 40, 2, 64, "pad"),
                                                (2, 150, 150, 1, 64, "pad"),
                                                # one query, more than 64 keys: the streaming single-query kernel (decode of long sentences)
                                                (5, 1, 129, 4, 64, "pad"), (3, 1, 65, 2, 64, "none"), (4, 1, 100, 2, 32, "pad"), (3, 1, 144, 2, 16, "pad"),
                                                (70, 1, 97, 16, 64, "pad"),
                                                # >= 1024 (sentence, head) items: the backward's split form (delta from dO * O)
                                                (260, 33, 33, 4, 64, "pad"), (300, 20, 30, 4, 32, "causal"), (520, 12, 16, 2, 16, "pad"),
                                                (600, 9, 11, 2, 8, "pad"), (1030, 40, 48, 1, 64, "none")])
def test_attention(ops, B, nq, nk, h, dk, kind):
    g = torch.Generator().manual_seed(B * 100 + nq)
    D = h * dk
    q, k, v = (torch.randn(B, n, D, generator=g) for n in (nq, nk, nk))
    if kind == "pad":
        lens = torch.randint(1, nk + 1, (B,), generator=g)
        mask = (torch.arange(nk)[None] >= lens[:, None]).view(B, 1, 1, nk)
    elif kind == "causal":
        big = (torch.ones(nk + 3, nk + 3).tril() == 0).view(1, 1, nk + 3, nk + 3)
        mask = big[:, :, :nq, :nk]   # non-contiguous slice, as transformer.py:497 passes it
    else:
        mask = None
    R = torch.randn(B, nq, D, generator=g)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    yref = ref_attention(qr, kr, vr, mask, h)
    (yref * R).sum().backward()
    qg, kg, vg = (dev(t).requires_grad_(True) for t in (q, k, v))
    y = ops.attention(qg, kg, vg, dev(mask) if mask is not None else None, h)
    (y * dev(R)).sum().backward()
    relclose(y, yref); relclose(qg.grad, qr.grad); relclose(kg.grad, kr.grad); relclose(vg.grad, vr.grad)


def test_attention_single_query_forward_beyond_tile_limits(ops):
    """One query against 300 keys (more than any tile kernel holds): the streaming single-query kernel loops over the keys."""
    g = torch.Generator().manual_seed(5)
    B, nk, h, dk = 3, 300, 2, 64
    q, k, v = torch.randn(B, 1, h * dk, generator=g), torch.randn(B, nk, h * dk, generator=g), torch.randn(B, nk, h * dk, generator=g)
    lens = torch.tensor([300, 1, 177])
    mask = (torch.arange(nk)[None] >= lens[:, None]).view(B, 1, 1, nk)
    with torch.no_grad():
        y = ops.attention(dev(q), dev(k), dev(v), dev(mask), h)
    relclose(y, ref_attention(q, k, v, mask, h))


@pytest.mark.parametrize("B,n,h,dk,holes", [(6, 33, 4, 64, False), (5, 48, 2, 32, False), (4, 33, 4, 64, True), (3, 20, 4, 16, True),
                                            (5, 129, 2, 64, False), (4, 100, 2, 32, True), (3, 144, 1, 16, True),
                                            (280, 33, 4, 64, True), (300, 40, 4, 32, False)])    # (>= 1024 items: split form)
def test_attention_ragged_backward(ops, B, n, h, dk, holes):
    """The backward kernel skips key tiles masked for every query and query tiles whose cotangent rows are all zero
    (what a ragged batch looks like: padded positions carry no loss).  Both short cuts must be exact, also for a key
    mask with holes (unmasked keys after masked ones) and for lengths that end inside a 16-row tile."""
    g = torch.Generator().manual_seed(7 * B + n)
    D = h * dk
    q, k, v = (torch.randn(B, n, D, generator=g) for _ in range(3))
    lens = torch.randint(1, n + 1, (B,), generator=g)
    lens[0], lens[-1] = n, 1
    mask = (torch.arange(n)[None] >= lens[:, None])
    if holes:
        mask = mask | (torch.rand(B, n, generator=g) < 0.3)
        mask[:, 0] = False
    mask = mask.view(B, 1, 1, n)
    R = torch.randn(B, n, D, generator=g)
    R = R * (torch.arange(n)[None] < lens[:, None]).unsqueeze(-1)        # zero cotangent at padded queries
    if holes:
        R[1, 3:9] = 0                                                    # zero rows inside a live tile
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    yref = ref_attention(qr, kr, vr, mask, h)
    (yref * R).sum().backward()
    qg, kg, vg = (dev(t).requires_grad_(True) for t in (q, k, v))
    y = ops.attention(qg, kg, vg, dev(mask), h)
    (y * dev(R)).sum().backward()
    relclose(y, yref); relclose(qg.grad, qr.grad); relclose(kg.grad, kr.grad); relclose(vg.grad, vr.grad)


@pytest.mark.parametrize("rows,V", [(6, 40), (33, 9600), (5, 1001), (3, 32000)])
def test_log_softmax_and_generator(ops, rows, V):
    g = torch.Generator().manual_seed(V)
    x = torch.randn(rows, V, generator=g) * 3
    R = torch.randn(rows, V, generator=g)
    xr = x.clone().requires_grad_(True)
    yref = torch.log_softmax(xr, -1)
    (yref * R).sum().backward()
    xg = dev(x).requires_grad_(True)
    y = ops.log_softmax(xg)
    (y * dev(R)).sum().backward()
    close(y, yref, 1e-5, 1e-5); relclose(xg.grad, xr.grad)
    if V <= 9600:
        d = 32
        h = torch.randn(rows, d, generator=g)
        W, b = torch.randn(V, d, generator=g) / 4, torch.randn(V, generator=g)
        hr, Wr, br = (t.clone().requires_grad_(True) for t in (h, W, b))
        yref = torch.log_softmax(torch.nn.functional.linear(hr, Wr, br) / 0.7, -1)
        (yref * R).sum().backward()
        hg, Wg, bg = (dev(t).requires_grad_(True) for t in (h, W, b))
        y = ops.generator(hg, Wg, bg, 0.7)
        (y * dev(R)).sum().backward()
        close(y, yref, 1e-4, 1e-4); relclose(hg.grad, hr.grad); relclose(Wg.grad, Wr.grad); relclose(bg.grad, br.grad)


@pytest.mark.parametrize("shape", [(3, 5, 40), (2, 16, 9600), (4, 3, 1001)])
def test_gumbel_softmax(ops, shape):
    g = torch.Generator().manual_seed(shape[-1])
    logits = torch.log_softmax(torch.randn(*shape, generator=g) * 2, -1)
    E = torch.empty(*shape).exponential_(generator=g)
    R = torch.randn(*shape, generator=g)
    for tau in (1.0, 0.5):
        lr = logits.clone().requires_grad_(True)
        gum = torch.min(-E.log(), torch.ones(1) * 1e10)
        yref = ((lr + gum) / tau).softmax(-1)
        (yref * R).sum().backward()
        lg = dev(logits).requires_grad_(True)
        y, idx = ops.gumbel_softmax_sample(lg, dev(E), None, tau)
        (y * dev(R)).sum().backward()
        relclose(y, yref); relclose(lg.grad, lr.grad)
        assert torch.equal(idx.cpu(), yref.argmax(-1))
    # sliced (non-contiguous batch stride) input, like log_probs[:, 1:]
    full = torch.randn(shape[0], shape[1] + 1, shape[2], generator=g)
    y, idx = ops.gumbel_softmax_sample(dev(full)[:, 1:], dev(E), None, 1.0)
    relclose(y, ((full[:, 1:] + torch.min(-E.log(), torch.ones(1) * 1e10))).softmax(-1))
    # in-kernel Philox: a valid distribution, reproducible for a fixed generator state, different across calls
    torch.cuda.manual_seed(123)
    y1, _ = ops.gumbel_softmax_sample(dev(logits))
    y2, _ = ops.gumbel_softmax_sample(dev(logits))
    torch.cuda.manual_seed(123)
    y3, _ = ops.gumbel_softmax_sample(dev(logits))
    assert torch.allclose(y1.sum(-1).cpu(), torch.ones(shape[:-1]), atol=1e-4)
    assert torch.equal(y1, y3) and not torch.equal(y1, y2)
    # mask variant + noise-free variant
    mask = (torch.rand(*shape, generator=g) > 0.3).long(); mask[..., 0] = 1
    ym, _ = ops.gumbel_softmax_sample(dev(logits), dev(E), dev(mask), 1.0)
    relclose(ym, (logits + torch.min(-E.log(), torch.ones(1) * 1e10)).masked_fill(mask == 0, -1e9).softmax(-1))
    ys, _ = ops.gumbel_softmax_sample(dev(logits), None, None, 1.0, True)
    relclose(ys, logits.softmax(-1))


def test_gumbel_rng_statistics(ops):
    # argmax of (logits + Gumbel) must follow softmax(logits): chi-square-ish check on a small vocabulary
    torch.cuda.manual_seed(7)
    p = torch.tensor([0.5, 0.25, 0.125, 0.125])
    logits = p.log().view(1, 1, 4).expand(1, 40000, 4).contiguous()
    _, idx = ops.gumbel_softmax_sample(dev(logits))
    freq = torch.bincount(idx.view(-1).cpu(), minlength=4).float() / 40000
    assert (freq - p).abs().max() < 0.01, freq


def test_first_eos_length(ops):
    toks = torch.tensor([[5, 6, 7, 8], [2, 5, 6, 7], [5, 6, 7, 2], [5, 2, 6, 2], [2, 2, 2, 2]])
    assert ops.first_eos_length(dev(toks), 2).cpu().tolist() == [4, 1, 4, 2, 1]


@pytest.mark.parametrize("soft", [False, True])
def test_embed_assemble(ops, soft):
    g = torch.Generator().manual_seed(9)
    B, L, d, V = 5, 7, 32, 40
    tokw, posw = torch.randn(V, d, generator=g), torch.randn(L + 2, d, generator=g)
    stylew, cls = torch.randn(2, d, generator=g), torch.randn(d, generator=g)
    style = torch.tensor([0, 1, 1, 0, 1])
    ids = torch.randint(0, V, (B, L), generator=g)
    rows = torch.randn(B, L, d, generator=g)
    R = torch.randn(B, L + 2, d, generator=g)
    tr, pr, sr, cr, rr = (t.clone().requires_grad_(True) for t in (tokw, posw, stylew, cls, rows))
    tok = rr if soft else tr[ids]
    yref = torch.cat([cr.view(1, 1, d).expand(B, 1, d), sr[style][:, None], tok + pr[:L][None]], 1)
    (yref * R).sum().backward()
    tg, pg, sg, cg, rg = (dev(t).requires_grad_(True) for t in (tokw, posw, stylew, cls, rows))
    y = ops.embed_assemble(rg if soft else dev(ids), None if soft else tg, pg, extra=[(cg, None), (sg, dev(style))])
    (y * dev(R)).sum().backward()
    close(y, yref, 1e-6, 1e-6)
    relclose(pg.grad, pr.grad); relclose(sg.grad, sr.grad); relclose(cg.grad, cr.grad)
    if soft:
        relclose(rg.grad, rr.grad)
    else:
        relclose(tg.grad, tr.grad)


def _lp_gather_ref(oin, olen, allpos, eos, unk, pos, shift, use_stay):
    """plain-torch restatement of transformer.py:471-473, :488-491 (candidate table + aligned gather)"""
    B, S1, d = oin.shape
    T1 = allpos.shape[1]
    rows = []
    for b in range(B):
        for t in range(T1):
            a = int(allpos[b, t])
            prev = int(allpos[b, t - 1]) if t > 0 else -1
            if use_stay and a == prev:
                r = unk
            else:
                p = a + shift
                r = oin[b, p] if (p == 0 or (0 < p < S1 and p - 1 < int(olen[b]))) else eos
            rows.append(r + (pos[t] if pos is not None else 0))
    return torch.stack(rows).view(B, T1, d)


@pytest.mark.parametrize("B,S1,T1,d,shift,use_stay,with_pos", [(5, 7, 6, 32, 0, True, True), (70, 12, 14, 64, 0, True, False),
                                                               (37, 9, 9, 48, 2, False, True), (6, 7, 6, 30, 0, True, True),
                                                               (200, 33, 34, 256, 0, True, True)])
def test_lp_gather_fwd_bwd(ops, B, S1, T1, d, shift, use_stay, with_pos):
    g = torch.Generator().manual_seed(B + T1)
    oin = torch.randn(B, S1, d, generator=g)
    olen = torch.randint(1, S1, (B,), generator=g)
    steps = torch.randint(0, 3, (B, T1), generator=g)
    allpos = torch.cumsum(steps, 1).clamp(max=S1 + 1)
    allpos[:, 0] = 0
    eos, unk = torch.randn(d, generator=g), torch.randn(d, generator=g)
    pos = torch.randn(T1 + 2, d, generator=g) if with_pos else None
    R = torch.randn(B, T1, d, generator=g)
    ref_in = [t.clone().requires_grad_(True) if t is not None else None for t in (oin, eos, unk, pos)]
    yref = _lp_gather_ref(ref_in[0], olen, allpos, ref_in[1], ref_in[2], ref_in[3], shift, use_stay)
    (yref * R).sum().backward()
    gi = [dev(t).requires_grad_(True) if t is not None else None for t in (oin, eos, unk, pos)]
    y = ops.lp_gather(gi[0], dev(olen), dev(allpos), gi[1], gi[2], gi[3], shift, use_stay, pos_skip0=False)
    (y * dev(R)).sum().backward()
    relclose(y, yref)
    for a, b in zip(gi, ref_in):
        if a is not None:
            relclose(a.grad, b.grad if b.grad is not None else torch.zeros_like(b))


@pytest.mark.parametrize("B,T1,S1,d", [(3, 5, 6, 32), (70, 9, 11, 64), (33, 17, 17, 256), (5, 6, 7, 30), (200, 34, 33, 256)])
def test_lp_skip_logits(ops, B, T1, S1, d):
    """transformer.py:504-507: einsum("bld,bltd->blt", wq(hidden), wk(candidates_output)[allpos]) / sqrt(d)"""
    g = torch.Generator().manual_seed(B * 7 + d)
    NC = S1 + 5
    qh, KC, kunk = torch.randn(B, T1, d, generator=g), torch.randn(B, NC, d, generator=g), torch.randn(d, generator=g)
    allpos = torch.cumsum(torch.randint(0, 3, (B, T1), generator=g), 1).clamp(max=S1)
    R = torch.randn(B, T1, 4, generator=g)
    rq, rK, rk = (t.clone().requires_grad_(True) for t in (qh, KC, kunk))
    idx = allpos.unsqueeze(-1) + torch.arange(1, 4)
    sel = torch.gather(rK, 1, idx.view(B, -1, 1).expand(-1, -1, d)).view(B, T1, 3, d)
    ref = torch.cat((torch.matmul(rq, rk.view(d, 1)), torch.einsum("bld,bljd->blj", rq, sel)), -1) / math.sqrt(d)
    (ref * R).sum().backward()
    gq, gK, gk = (dev(t).requires_grad_(True) for t in (qh, KC, kunk))
    out = ops.lp_logits(gq, gK, gk, dev(allpos))
    (out * dev(R)).sum().backward()
    relclose(out, ref); relclose(gq.grad, rq.grad); relclose(gK.grad, rK.grad); relclose(gk.grad, rk.grad)


@pytest.mark.parametrize("rows,d,V,temp", [(6, 32, 40, 1.0), (33, 256, 9600, 0.7), (5, 64, 1001, 1.0)])
def test_generator_nll_fused(ops, rows, d, V, temp):
    """Generator (transformer.py:213-214) + NLLLoss(reduction='none') (seq2seq.py:453) as one node."""
    g = torch.Generator().manual_seed(V + rows)
    x, W, b = torch.randn(rows, d, generator=g), torch.randn(V, d, generator=g) * 0.1, torch.randn(V, generator=g) * 0.1
    tok = torch.randint(0, V, (rows,), generator=g)
    w = torch.rand(rows, generator=g)
    rx, rW, rb = (t.clone().requires_grad_(True) for t in (x, W, b))
    ref = torch.nn.functional.nll_loss(torch.log_softmax((rx @ rW.t() + rb) / temp, -1), tok, reduction="none")
    (ref * w).sum().backward()
    gx, gW, gb = (dev(t).requires_grad_(True) for t in (x, W, b))
    out = ops.generator_nll(gx, gW, gb, temp, dev(tok))
    (out * dev(w)).sum().backward()
    relclose(out, ref); relclose(gx.grad, rx.grad); relclose(gW.grad, rW.grad); relclose(gb.grad, rb.grad)


@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (1088, 256, 256), (300, 1024, 256), (2112, 9600, 256), (4164, 1024, 1024), (2052, 2048, 96),
                                   (16388, 256, 1024)])
def test_gemm_bf16_variant(ops, M, N, K):
    """bf16-MFMA variant of the grouped GEMM (fp32 storage, fp32 accumulate): forward, dgrad, wgrad, bias grad within the
    bf16 tolerance of north_star (2e-2 relative to the largest element)."""
    g = torch.Generator().manual_seed(M + N)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    rx, rW, rb = (t.clone().requires_grad_(True) for t in (x, W, b))
    ref = torch.nn.functional.linear(rx, rW, rb)
    (ref * R).sum().backward()
    gx, gW, gb = (dev(t).requires_grad_(True) for t in (x, W, b))
    with ops.gemm_precision("bf16"):
        y = ops.linear(gx, gW, gb)
        (y * dev(R)).sum().backward()
    for a, r in ((y, ref), (gx.grad, rx.grad), (gW.grad, rW.grad), (gb.grad, rb.grad)):
        relclose(a, r, 2e-2)
    # and it really is a different arithmetic: not bit-identical to the fp32 path
    y32 = ops.linear(dev(x), dev(W), dev(b))
    if M > 1024:      # (M <= 1024 forward goes through the small-M kernel, which is fp32 only)
        assert (y32 - y).abs().max().item() > 0
    relclose(y32, ref, 1e-4)


@pytest.mark.parametrize("M,N,K", [(4160, 1024, 256), (2112, 2048, 512)])
def test_gemm_bf16_tile_epilogues(ops, M, N, K, monkeypatch):
    """The bf16 tile kernel (csrc/gemm_bf16_tile.h: bf16 K-contiguous LDS images, 128 x 128 x 64) through the epilogues the model
    uses -- bias + ReLU with the ReLU mask in the backward (FFN), bias + residual, soft tokens x table (n-contiguous B) -- against
    fp32 torch at north_star's 2e-2, and against the older fp32-image variant (NAST_GEMM_BF16_TILE=0 is read once per process,
    so the comparison here is with torch only)."""
    g = torch.Generator().manual_seed(M + N + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    res, R = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    # y = relu(x W^T + b): forward epilogue RELU.  The backward is compared with the device's own gates: a pre-activation
    # within the bf16 error of zero opens in one arithmetic and not in the other, which is a property of the rounding,
    # not an error of either product (tests/test_model_gpu.py replays gates for the same reason).
    ref = torch.relu(torch.nn.functional.linear(x, W, b))
    gx, gW, gb = (dev(t).requires_grad_(True) for t in (x, W, b))
    with ops.gemm_precision("bf16"):
        y = ops.linear(gx, gW, gb, None, True)
        (y * dev(R)).sum().backward()
    dy = R * (y.detach().cpu() > 0).float()
    for a, r in ((y, ref), (gx.grad, dy @ W), (gW.grad, dy.t() @ x), (gb.grad, dy.sum(0))):
        relclose(a, r, 2e-2)
    # y = res + x W^T + b
    with ops.gemm_precision("bf16"):
        y2 = ops.linear(dev(x), dev(W), dev(b), dev(res))
    relclose(y2, res + torch.nn.functional.linear(x, W, b), 2e-2)
    # soft tokens x table: y = p E (E [K', d] n-contiguous), dp = dy E^T, dE = p^T dy
    p_, E = torch.softmax(torch.randn(M, N, generator=g), -1), torch.randn(N, K, generator=g)
    rp, rE = p_.clone().requires_grad_(True), E.clone().requires_grad_(True)
    refm = rp @ rE
    Rm = torch.randn(M, K, generator=g)
    (refm * Rm).sum().backward()
    gp, gE = dev(p_).requires_grad_(True), dev(E).requires_grad_(True)
    with ops.gemm_precision("bf16"):
        ym = ops.matmul_nn(gp, gE)
        (ym * dev(Rm)).sum().backward()
    for a, r in ((ym, refm), (gp.grad, rp.grad), (gE.grad, rE.grad)):
        relclose(a, r, 2e-2)


def test_step_bf16_variant_vs_oracle():
    """One learnable f_step with bf16-MFMA GEMMs against the fp32 CPU oracle: losses, gradient norm and every parameter
    gradient (relative to the model's largest) within north_star's 2e-2."""
    from test_model_gpu import _oracle_vs_hip_step
    rep = _oracle_vs_hip_step(dict(V=9600, d=256, h=4, N=4, L=32, learnable=True), B=6, seed=3, precision="bf16", name="bench_b6_bf16")
    assert rep["loss_rel_err"] <= 2e-2 and rep["grad_norm_rel_err"] <= 2e-2, rep
    assert max(c["grad_err_vs_model_max"] for c in rep["classes"].values()) <= 2e-2, rep["classes"]


def test_fused_adam_late_first_gradient(ops):
    """A parameter whose first gradient arrives several steps after the others' (the length predictor while `pre_z_batch`
    steps run without it, seq2seq.py:241-247; restored optimizer states): torch.optim.Adam keeps one step count per
    parameter, so its bias corrections differ from the rest of the group -- FusedAdam splits the launch by step count."""
    from nast_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(11)
    shapes = [(300, 64), (64,), (128, 32), (5,)]
    late = {2: 3, 3: 1}                           # tensor index -> first step with a gradient
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    got = [dev(t.detach().clone()).requires_grad_(True) for t in ref]
    o_ref, o_got = torch.optim.Adam(ref, lr=1e-3), FusedAdam(got, lr=1e-3)
    for step in range(6):
        for i, (a, b) in enumerate(zip(ref, got)):
            if step < late.get(i, 0):
                continue
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), dev(gr.clone())
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        o_ref.step()
        n_got = o_got.clip_and_step(5.0)
        np.testing.assert_allclose(float(n_got), float(n_ref), rtol=1e-5)
        for a, b in zip(ref, got):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=1e-5, atol=2e-7)
    for k, st in o_ref.state_dict()["state"].items():
        assert float(st["step"]) == float(o_got.state_dict()["state"][k]["step"])
    assert {float(st["step"]) for st in o_got.state_dict()["state"].values()} == {6.0, 3.0, 5.0}


def _participation_case(device):
    """FusedAdam.track_participation(): persistent zero-filled .grad buffers (zero_grad(set_to_none=False), the data-parallel
    flat buffer) against torch.optim.Adam with zero_grad() -> None.  Tensor 2 sits out steps 0-1 AND step 4 (the length
    predictor around `pre_z_batch`, seq2seq.py:241-247), tensor 3 never takes part (sos_token); weight decay on: a parameter
    that sits out must not decay, move on its momentum or advance its step count."""
    from nast_amd import ops as O
    from nast_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(21)
    shapes = [(300, 64), (64,), (128, 32), (5,)]
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    got = [t.detach().clone().to(device).requires_grad_(True) for t in ref]
    o_ref, o_got = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-2), FusedAdam(got, lr=1e-3, weight_decay=1e-2)
    o_got.track_participation()
    for b in got:
        b.grad = torch.zeros_like(b)              # buffers exist from the start and are never None again
    for step in range(6):
        o_ref.zero_grad()
        o_got.zero_grad(set_to_none=False)
        for i, (a, b) in enumerate(zip(ref, got)):
            if i == 3 or (i == 2 and step in (0, 1, 4)):
                continue
            gr = torch.randn(a.shape, generator=g)
            a.grad = gr.clone()
            if i == 0:                            # through autograd's AccumulateGrad (post-accumulate hook) ...
                (b * gr.to(device)).sum().backward()
            else:                                 # ... or in place through the kernels' gradient sink
                (buf,), _ = O.grad_buffers([b])
                assert buf.data_ptr() == b.grad.data_ptr()
                buf += gr.to(device)
        torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], 5.0)
        o_ref.step()
        o_got.clip_and_step(5.0)
        for a, b in zip(ref, got):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=1e-5, atol=2e-7, err_msg=f"step {step}")
    sd_ref, sd_got = o_ref.state_dict()["state"], o_got.state_dict()["state"]
    assert sd_ref.keys() == sd_got.keys() == {0, 1, 2}
    assert [float(sd_got[k]["step"]) for k in (0, 1, 2)] == [float(sd_ref[k]["step"]) for k in (0, 1, 2)] == [6.0, 6.0, 3.0]


def test_fused_adam_skips_parameters_that_sat_out(ops):
    _participation_case("cuda")


@pytest.mark.parametrize("max_norm,wd", [(5.0, 0.0), (0.05, 0.0), (None, 0.0), (5.0, 1e-3)])
def test_fused_adam_matches_torch(ops, max_norm, wd):
    """FusedAdam.clip_and_step == clip_grad_norm_ + torch.optim.Adam.step over several steps (seq2seq.py:536-537),
    incl. a parameter that never gets a gradient (sos_token, transformer.py:41) and the same state_dict layout."""
    from nast_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(9600, 64), (256, 256), (256,), (1024, 256), (7,), (1,)]
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    got = [dev(t.detach().clone()).requires_grad_(True) for t in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd)
    o_got = FusedAdam(got, lr=1e-3, weight_decay=wd)
    for step in range(4):
        for i, (a, b) in enumerate(zip(ref, got)):
            if i == len(ref) - 1:
                continue                      # never receives a gradient
            gr = torch.randn(a.shape, generator=g) * (10.0 if step == 1 else 0.1)
            a.grad = gr.clone()
            b.grad = dev(gr.clone())
        n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm) if max_norm is not None else None
        o_ref.step()
        n_got = o_got.clip_and_step(max_norm)
        if max_norm is not None:
            np.testing.assert_allclose(float(n_got), float(n_ref), rtol=1e-5)
            for a, b in zip(ref[:-1], got[:-1]):
                relclose(b.grad, a.grad, 1e-5)                 # clipped in place like clip_grad_norm_
        for a, b in zip(ref, got):
            x, y = b.detach().cpu().numpy(), a.detach().numpy()
            if wd == 0.0:
                np.testing.assert_allclose(x, y, rtol=1e-5, atol=2e-7)
            else:
                # where grad + wd * p nearly cancels, Adam's m / sqrt(v) turns the 1e-5 rounding of the clip coefficient into
                # an O(lr) difference (ill-conditioned in torch itself): allow a few such elements, bounded by lr per step
                bad = ~np.isclose(x, y, rtol=1e-5, atol=2e-7)
                assert bad.mean() < 1e-4 and np.abs(x - y).max() <= 2e-3 * (step + 1)
    sd_ref, sd_got = o_ref.state_dict(), o_got.state_dict()
    assert sd_ref["state"].keys() == sd_got["state"].keys()
    for k in sd_ref["state"]:
        assert sd_ref["state"][k].keys() == sd_got["state"][k].keys()
        assert float(sd_ref["state"][k]["step"]) == float(sd_got["state"][k]["step"])
        # the clip coefficient carries the rounding of two different fp32 reductions of ~3e6 squares (norms agree to 1e-5)
        np.testing.assert_allclose(sd_got["state"][k]["exp_avg_sq"].cpu().numpy(), sd_ref["state"][k]["exp_avg_sq"].numpy(), rtol=5e-5, atol=1e-12)


@pytest.mark.parametrize("B,nq,nk,h,dk", [(6, 33, 33, 4, 64), (5, 20, 48, 2, 32), (3, 64, 17, 4, 16), (4, 9, 9, 2, 8),
                                          (5, 129, 129, 2, 64), (4, 70, 130, 2, 32), (3, 144, 90, 2, 16),
                                          # >= 1024 (sentence, head) items: the backward's split form when the forward result is passed
                                          (300, 34, 34, 4, 64), (260, 20, 48, 4, 32), (600, 16, 9, 2, 16)])
def test_attention_packed_rows(ops, B, nq, nk, h, dk):
    """nast_mha_*_varlen_f32 on packed rows == the padded call with a key-padding mask, on the real rows (forward and
    all three gradients); sentences of length 1 and of full width included."""
    g = torch.Generator().manual_seed(11 * B + nq)
    D = h * dk
    ql = torch.randint(1, nq + 1, (B,), generator=g); ql[0], ql[-1] = nq, 1
    kl = torch.randint(1, nk + 1, (B,), generator=g); kl[0], kl[-1] = nk, 1
    q, k, v = (torch.randn(B, n, D, generator=g) for n in (nq, nk, nk))
    R = torch.randn(B, nq, D, generator=g) * (torch.arange(nq)[None] < ql[:, None]).unsqueeze(-1)
    mask = (torch.arange(nk)[None] >= kl[:, None]).view(B, 1, 1, nk)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    (ref_attention(qr, kr, vr, mask, h) * R).sum().backward()
    yref = ref_attention(q, k, v, mask, h)
    pq, pk = ops.PackedSeq(dev(ql), nq), ops.PackedSeq(dev(kl), nk)
    pair = ops.PackedPair(pq, pk)
    qp, kp, vp = ops.pack_rows(dev(q), pq), ops.pack_rows(dev(k), pk), ops.pack_rows(dev(v), pk)
    out, lse = ops.mha_fwd_packed(qp, kp, vp, pair, h)
    qmask = (torch.arange(nq)[None] < ql[:, None]).unsqueeze(-1)
    kmask = (torch.arange(nk)[None] < kl[:, None]).unsqueeze(-1)
    relclose(ops.unpack_rows(out, pq), yref * qmask)
    for with_out in (None, out):       # without the forward result: two-phase kernel; with it: delta from dO * O, halves side by side
        dq, dk_, dv = ops.mha_bwd_packed(qp, kp, vp, pair, h, ops.pack_rows(dev(R), pq), lse, with_out)
        for t, ps_ in ((out, pq), (dq, pq), (dk_, pk), (dv, pk)):         # buffer padding rows come back as exact zeros
            assert t.shape[0] == ps_.rows and float(t[ps_.total:].abs().max()) == 0.0
        relclose(ops.unpack_rows(dq, pq), qr.grad * qmask)
        relclose(ops.unpack_rows(dk_, pk), kr.grad * kmask)
        relclose(ops.unpack_rows(dv, pk), vr.grad * kmask)


@pytest.mark.parametrize("n,h,dk,B", [(33, 4, 64, 5), (129, 2, 64, 5), (100, 2, 32, 5), (33, 4, 64, 280)])
def test_attention_packed_rows_causal(ops, n, h, dk, B):
    """causal flag of the packed attention entry points == the padded call with the causal mask, on the real rows
    (280 sentences x 4 heads: the backward's split form)"""
    g = torch.Generator().manual_seed(5)
    D = h * dk
    ln = torch.randint(1, n + 1, (B,), generator=g); ln[0], ln[-1] = n, 1
    q, k, v = (torch.randn(B, n, D, generator=g) for _ in range(3))
    live = (torch.arange(n)[None] < ln[:, None]).unsqueeze(-1)
    R = torch.randn(B, n, D, generator=g) * live
    mask = (torch.ones(n, n).tril() == 0).view(1, 1, n, n)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    (ref_attention(qr, kr, vr, mask, h) * R).sum().backward()
    yref = ref_attention(q, k, v, mask, h)
    ps = ops.PackedSeq(dev(ln), n)
    pair = ops.PackedPair(ps, ps, causal=True)
    qp, kp, vp = (ops.pack_rows(dev(t), ps) for t in (q, k, v))
    out, lse = ops.mha_fwd_packed(qp, kp, vp, pair, h)
    relclose(ops.unpack_rows(out, ps), yref * live)
    for with_out in (None, out):
        dq, dk_, dv = ops.mha_bwd_packed(qp, kp, vp, pair, h, ops.pack_rows(dev(R), ps), lse, with_out)
        relclose(ops.unpack_rows(dq, ps), qr.grad * live)
        relclose(ops.unpack_rows(dk_, ps), kr.grad * live)
        relclose(ops.unpack_rows(dv, ps), vr.grad * live)


@pytest.mark.parametrize("B,width,d,zero_frac", [(1, 5, 8, 0.0), (7, 9, 6, 0.3), (513, 33, 256, 0.0), (300, 17, 64, 0.1), (5, 130, 1024, 0.0)])
def test_pack_unpack_rows_equal_torch_index_ops(ops, B, width, d, zero_frac):
    """ops.pack_rows / ops.unpack_rows (gather kernels, forward and backward) against index_select / zeros + index_copy and
    what autograd makes of those -- plain copies, so bit-exact; also a strided one-column view as PackedSoft.column passes it."""
    g = torch.Generator().manual_seed(B * 7 + width)
    lengths = torch.randint(1, width + 1, (B,), generator=g)
    lengths[torch.rand(B, generator=g) < zero_frac] = 0
    lengths[0] = width
    lengths = lengths.cuda()
    ps = ops.PackedSeq(lengths, width, int(lengths.sum()))
    x = torch.randn(B, width, d, generator=g).cuda()
    Rp = torch.randn(ps.rows, d, generator=g).cuda()
    Ru = torch.randn(B, width, d, generator=g).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = ops.pack_rows(xa, ps)
    yb = xb.reshape(-1, d).index_select(0, ps.idx_p)
    assert torch.equal(ya, yb)
    Rp0 = Rp.clone(); Rp0[ps.total:] = 0                      # (the rows past `total` carry zero gradient in the model)
    (ya * Rp0).sum().backward(); (yb * Rp0).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    pa, pb = ya.detach().clone().requires_grad_(True), ya.detach().clone().requires_grad_(True)
    ua = ops.unpack_rows(pa, ps)
    ub = torch.zeros(B * width, d, device="cuda").index_copy(0, ps.idx, pb[:ps.total]).view(B, width, d)
    assert torch.equal(ua, ub)
    (ua * Ru).sum().backward(); (ub * Ru).sum().backward()
    assert torch.equal(pa.grad, pb.grad)
    col = ya.detach()[:, d // 2:d // 2 + 1]                     # strided [rows, 1] view
    assert torch.equal(ops.unpack_rows(col, ps), torch.zeros(B * width, 1, device="cuda").index_copy(0, ps.idx, col[:ps.total]).view(B, width, 1))


@pytest.mark.parametrize("B,width,zero_frac", [(1, 5, 0.0), (7, 9, 0.3), (513, 33, 0.0), (2048, 17, 0.1), (64, 64, 0.0)])
def test_pack_index_kernel_equals_torch_maps(ops, B, width, zero_frac):
    """nast_pack_index_i64 (one launch) against the torch formulation of ops.PackedSeq -- integer maps, bit-exact."""
    g = torch.Generator().manual_seed(B * 31 + width)
    lengths = torch.randint(1, width + 1, (B,), generator=g)
    lengths[torch.rand(B, generator=g) < zero_frac] = 0          # (the tail layout of one-word sentences has zero lengths)
    lengths = lengths.cuda()
    total = int(lengths.sum())
    nat = ops.PackedSeq(lengths, width, total)
    assert hasattr(nat, "bad")
    ops.check_packed()                                           # (does not raise)
    ops.NATIVE_PACK_INDEX = False
    try:
        ref = ops.PackedSeq(lengths, width, total)
    finally:
        ops.NATIVE_PACK_INDEX = True
    assert not hasattr(ref, "bad")
    assert nat.rows == ref.rows and nat.total == ref.total
    for name in ("off", "off32", "len32", "b_of", "pos", "idx", "idx_p"):
        a, b = getattr(nat, name), getattr(ref, name)
        assert a.dtype == b.dtype and a.shape == b.shape, name
        assert torch.equal(a, b), name
    if B > 1 and zero_frac == 0.0:         # tail() describes lengths - 1: every sentence needs a first row
        t_nat = nat.tail()
        ops.NATIVE_PACK_INDEX = False
        try:
            t_ref = ref.tail()
        finally:
            ops.NATIVE_PACK_INDEX = True
        assert torch.equal(t_nat[0], t_ref[0]) and torch.equal(t_nat[1].idx_p, t_ref[1].idx_p)
    # a wrong host-side total is reported, not silently used
    from nast_amd import _lib
    wrong = ops.PackedSeq(lengths, width, total + 64)
    assert int(wrong.idx_p.abs().max().item()) == 0              # zero-filled, in bounds
    with pytest.raises(_lib.NastHipError):
        ops.check_packed()
    ops.check_packed()                                           # the flag was reset


@pytest.mark.parametrize("B,width,extra", [(1, 1, 0), (7, 9, 1), (513, 33, 1), (64, 129, 2), (300, 17, 0)])
def test_pad_mask_kernel_equals_torch_formulation(ops, B, width, extra):
    """transformer._pad_mask through nast_pad_mask_u8 (one launch) against arange / compare / zeros / cat, lengths from 0 to
    beyond the width."""
    from nast_amd import transformer as tf
    g = torch.Generator().manual_seed(B + width)
    lengths = torch.randint(0, width + 3, (B,), generator=g)
    got = tf._pad_mask(lengths.cuda(), width, extra)
    ref = tf._pad_mask(lengths, width, extra)                  # CPU tensors take the torch formulation
    assert got.dtype == torch.bool and got.shape == (B, 1, 1, extra + width) and got.is_contiguous()
    assert torch.equal(got.cpu(), ref)
    assert got.nast_lengths[1] == extra
