"""GPU parity tests: every HIP kernel, called through the C-ABI, against the CPU oracle.

Tolerances: f32 path 1e-4 abs on O(1) values (exact-f32 MFMA, different summation order);
bf16 path = deviation from the f32 oracle with bf16-rounded operands, 2e-2 relative.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from mfp.hip import ops
    return ops


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def assert_close(got, want, atol, rtol, what=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), "%s: max err %.3e (tol %.1e/%.1e), %d/%d bad, worst want=%.4f got=%.4f" % (
        what, err.max().item(), atol, rtol, int(bad.sum()), bad.numel(),
        want.flatten()[err.argmax()].item(), got.flatten()[err.argmax()].item())


# ---------------------------------------------------------------------------- hardware probe
def test_tr_read_lane_mapping():
    """ds_read_b64_tr_b16: out[lane i][j] = in[lane 4j + i/4][i%4] per 16-lane group."""
    ops = _ops()
    lanes = torch.arange(64)
    # (a) linear addresses: lane l -> byte 8*l
    out = ops.tr_probe((lanes * 8).to(torch.int32).to(DEV)).cpu().to(torch.int64)
    l = lanes[:, None]
    j = torch.arange(4)[None, :]
    assert torch.equal(out, (l & 15) + j * 16 + (l >> 4) * 64)
    # (b) the row-strided pattern the kernels use: rows of 40 elements
    ld = 40
    li, lg = lanes & 15, lanes >> 4
    addr = ((8 * lg + (li >> 2)) * ld + (li & 3) * 4) * 2
    out = ops.tr_probe(addr.to(torch.int32).to(DEV)).cpu().to(torch.int64)
    want = (8 * lg[:, None] + j) * ld + li[:, None]  # element [k = 8g + j][col = i]
    assert torch.equal(out, want)


# ------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(256, 128, 64), (200, 136, 72), (128, 256, 512), (1000, 376, 128), (64, 8, 8)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_plain(dtype, layout, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    if layout == "wgrad" and M % 8:
        pytest.skip("wgrad needs M % 8 == 0")
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    if dtype == torch.bfloat16:
        A, Bm = bf16_round(A), bf16_round(Bm)
    want = A.double() @ Bm.double()
    if layout == "fwd":
        a_dev, b_dev, ak, bk = A, Bm, True, False
    elif layout == "dgrad":
        a_dev, b_dev, ak, bk = A, Bm.t().contiguous(), True, True
    else:
        a_dev, b_dev, ak, bk = A.t().contiguous(), Bm, False, False
    got = ops.gemm(a_dev.to(DEV, dtype), b_dev.to(DEV, dtype), M, N, K, a_kmajor=ak, b_kmajor=bk,
                   out_dtype=torch.float32, splitk=3 if layout == "wgrad" else 1)
    assert_close(got, want, 2e-4 * math.sqrt(K), 1e-5, "gemm %s %s" % (layout, dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 256, 128
    A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) * 0.1
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    if dtype == torch.bfloat16:
        A, W = bf16_round(A), bf16_round(W)
    Ad, Wd = A.to(DEV, dtype), W.to(DEV, dtype)
    base = A.double() @ W.double() + bias.double()
    tol = dict(atol=3e-4, rtol=1e-5)
    # bias + relu -> cdt output
    got = ops.gemm(Ad, Wd, M, N, K, a_kmajor=True, b_kmajor=False, bias=bias.to(DEV), relu=True, out_dtype=dtype)
    want = base.clamp(min=0)
    if dtype == torch.bfloat16:
        assert_close(got, want, 1e-2, 1e-2, "bias+relu bf16 out")
    else:
        assert_close(got, want, what="bias+relu", **tol)
    # bias + residual (f32 out)
    got = ops.gemm(Ad, Wd, M, N, K, a_kmajor=True, b_kmajor=False, bias=bias.to(DEV), residual=res.to(DEV),
                   out_dtype=torch.float32)
    assert_close(got, base + res.double(), what="bias+residual", **tol)
    # accumulate + rowskip
    code = (torch.rand(M, generator=g) < 0.3).to(torch.uint8)
    C0 = torch.randn(M, N, generator=g)
    out = C0.clone().to(DEV)
    ops.gemm(Ad, Wd, M, N, K, a_kmajor=True, b_kmajor=False, bias=bias.to(DEV), out=out, accum=True,
             rowskip=code.to(DEV))
    want = C0.double() + base * (code == 0)[:, None]
    assert_close(out, want, what="accum+rowskip", **tol)
    # dgrad with relu mask
    H = torch.randn(M, K, generator=g)
    dY = torch.randn(M, N, generator=g)
    if dtype == torch.bfloat16:
        dY = bf16_round(dY)
    got = ops.gemm(dY.to(DEV, dtype), Wd, M, K, N, a_kmajor=True, b_kmajor=True, out_dtype=dtype,
                   relu_bwd_aux=H.to(DEV, dtype))
    want = (dY.double() @ W.double().t()) * (H.to(dtype).double() > 0)
    if dtype == torch.bfloat16:
        assert_close(got, want, 2e-2, 1e-2, "relu_bwd bf16")
    else:
        assert_close(got, want, what="relu_bwd", **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["fwd", "dgrad"])
@pytest.mark.parametrize("M,N,K", [(2304, 264, 256), (4100, 768, 128), (2050, 136, 512), (3000, 1384, 256),
                                   (2200, 256, 768), (2048, 128, 72), (2100, 512, 1024), (1500, 200, 1024)])
def test_gemm_large_m_epilogues(dtype, layout, M, N, K):
    """Train-step-like skinny shapes (large M, small N/K) through every epilogue flag."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(K, N, generator=g) * 0.1
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    if dtype == torch.bfloat16:
        A, W = bf16_round(A), bf16_round(W)
    Bdev = (W.t().contiguous() if layout == "fwd" else W).to(DEV, dtype)
    bk = layout == "fwd"
    want = A.double() @ W.double()
    tol = 2e-4 * math.sqrt(K)
    got = ops.gemm(A.to(DEV, dtype), Bdev, M, N, K, a_kmajor=True, b_kmajor=bk, out_dtype=torch.float32)
    assert_close(got, want, tol, 1e-5, "apanel plain")
    code = (torch.rand(M, generator=g) < 0.2).to(torch.uint8)
    C0 = torch.randn(M, N, generator=g)
    out = C0.clone().to(DEV)
    ops.gemm(A.to(DEV, dtype), Bdev, M, N, K, a_kmajor=True, b_kmajor=bk, out=out, bias=bias.to(DEV), accum=True,
             rowskip=code.to(DEV))
    assert_close(out, C0.double() + (want + bias.double()) * (code == 0)[:, None], tol, 1e-5, "apanel accum+rowskip")
    got = ops.gemm(A.to(DEV, dtype), Bdev, M, N, K, a_kmajor=True, b_kmajor=bk, bias=bias.to(DEV), relu=True,
                   residual=None, out_dtype=dtype)
    ref = (want + bias.double()).clamp(min=0)
    if dtype == torch.bfloat16:
        assert_close(got, ref, 2e-2, 1e-2, "apanel relu bf16")
    else:
        assert_close(got, ref, tol, 1e-5, "apanel relu")
    p, seed, off = 0.1, 99, 5
    plain = ops.gemm(A.to(DEV, dtype), Bdev, M, N, K, a_kmajor=True, b_kmajor=bk, bias=bias.to(DEV),
                     out_dtype=torch.float32)
    dropped = ops.gemm(A.to(DEV, dtype), Bdev, M, N, K, a_kmajor=True, b_kmajor=bk, bias=bias.to(DEV),
                       residual=res.to(DEV), dropout=(p, seed, off), out_dtype=torch.float32)
    colsum = torch.empty(N, device=DEV)
    via_bwd = ops.dropout_bwd(plain, torch.float32, colsum, p, seed, off)
    assert_close(dropped, via_bwd.double().cpu() + res.double(), 1e-5, 1e-5, "apanel dropout+residual")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_wgrad_colsum_rowskip(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    T, M, N = 1500, 264, 128
    X, dY = torch.randn(T, M, generator=g), torch.randn(T, N, generator=g)
    if dtype == torch.bfloat16:
        X, dY = bf16_round(X), bf16_round(dY)
    code = (torch.rand(T, generator=g) < 0.25).to(torch.uint8)
    keep = (code == 0).double()[:, None]
    colsum = torch.empty(M, device=DEV)
    got = ops.gemm(X.to(DEV, dtype), dY.to(DEV, dtype), M, N, T, a_kmajor=False, b_kmajor=False,
                   out_dtype=torch.float32, colsum=colsum, rowskip_a=code.to(DEV), splitk=5)
    assert_close(got, (X.double() * keep).t() @ dY.double(), 2e-3, 1e-5, "wgrad")
    assert_close(colsum, (X.double() * keep).sum(0), 2e-3, 1e-5, "colsum")


def test_gemm_dropout_matches_dropout_bwd():
    ops = _ops()
    M, N, K, p, seed, off = 512, 256, 64, 0.1, 1234, 77
    A = torch.randn(M, K).to(DEV)
    W = torch.randn(K, N).to(DEV)
    plain = ops.gemm(A, W, M, N, K, a_kmajor=True, b_kmajor=False)
    dropped = ops.gemm(A, W, M, N, K, a_kmajor=True, b_kmajor=False, dropout=(p, seed, off))
    colsum = torch.empty(N, device=DEV)
    via_bwd = ops.dropout_bwd(plain, torch.float32, colsum, p, seed, off)
    assert torch.equal(dropped, via_bwd)  # identical counter-based stream (common.h: hash RNG keyed by seed, site, step, row, column)
    keep = (dropped != 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01
    assert_close(colsum, via_bwd.double().sum(0).cpu(), 1e-2, 1e-5, "dropout colsum")
    kept = dropped != 0
    assert_close(dropped[kept], plain[kept] / (1 - p), 1e-5, 1e-5, "scale")
    other = ops.gemm(A, W, M, N, K, a_kmajor=True, b_kmajor=False, dropout=(p, seed, off + 1))
    assert not torch.equal(other, dropped)


# ------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,D", [(37, 128), (1000, 256), (130, 512), (24601, 256)])
def test_layernorm(dtype, T, D):
    """(24 601 rows: the backward kernel's 32-rows-per-workgroup form -- below 3 x 256 x 32 rows it runs 16 per workgroup.)"""
    ops = _ops()
    g = torch.Generator().manual_seed(T + D)
    x = (torch.randn(T, D, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(D, generator=g)).requires_grad_(True)
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    y = (x - mean) / torch.sqrt(var + 1e-3) * gamma + beta
    dy = torch.randn(T, D, generator=g)
    if dtype == torch.bfloat16:
        dy = bf16_round(dy)
    dres = torch.randn(T, D, generator=g)
    y.backward(dy)
    yd, md, rd = ops.layernorm_fwd(x.detach().to(DEV), gamma.detach().to(DEV), beta.detach().to(DEV), dtype)
    if dtype == torch.float32:
        assert_close(yd, y, 1e-5, 1e-5, "ln fwd")
    else:
        assert_close(yd, y, 2e-2, 1e-2, "ln fwd bf16")
    assert_close(md, mean[:, 0], 1e-5, 1e-5, "mean")
    dg, db = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV, dtype), x.detach().to(DEV), gamma.detach().to(DEV), md, rd,
                           dres.to(DEV), dg, db)
    assert_close(dx, x.grad + dres, 2e-5, 1e-5, "ln dx")
    assert_close(dg, gamma.grad, 1e-3, 1e-5, "ln dgamma")
    assert_close(db, beta.grad, 1e-3, 1e-5, "ln dbeta")
    # fused Dropout-backward consumer == mfp_dropout_bwd on the produced dx (same Philox stream)
    p, seed, off = 0.1, 77, 3
    cs = torch.empty(D, device=DEV)
    dx2, dd = ops.layernorm_bwd(dy.to(DEV, dtype), x.detach().to(DEV), gamma.detach().to(DEV), md, rd,
                                dres.to(DEV), dg, db, drop=(cs, p, seed, off, None))
    assert torch.equal(dx2, dx)
    cs_ref = torch.empty(D, device=DEV)
    dd_ref = ops.dropout_bwd(dx, dtype, cs_ref, p, seed, off)
    assert torch.equal(dd, dd_ref)
    assert_close(cs, cs_ref, 1e-3, 1e-4, "fused colsum")


# ------------------------------------------------------------------------------- attention
def _attn_ref(qkv, nvalid, B, S, H):
    D = qkv.shape[1] // 3
    hd = D // H
    q, k, v = [t.reshape(B, S, H, hd).permute(0, 2, 1, 3) for t in qkv.split(D, dim=1)]
    score = q @ k.transpose(-1, -2) / math.sqrt(hd)
    mask = (torch.arange(S)[None, :] < nvalid[:, None]).to(qkv.dtype)[:, None, None, :]
    score = score + -1e9 * (1.0 - mask)
    w = torch.softmax(score, dim=-1)
    return (w @ v).permute(0, 2, 1, 3).reshape(B * S, D)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,S,H,hd", [(2, 32, 8, 16), (3, 128, 8, 32), (2, 50, 8, 32), (1, 7, 4, 32),
                                       (2, 256, 2, 64), (2, 100, 2, 16), (3, 200, 8, 64), (2, 256, 8, 64), (1, 129, 8, 64)])
def test_attention(dtype, B, S, H, hd):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + S)
    D = H * hd
    qkv = torch.randn(B * S, 3 * D, generator=g)
    dout = torch.randn(B * S, D, generator=g)
    if dtype == torch.bfloat16:
        qkv, dout = bf16_round(qkv), bf16_round(dout)
    nvalid = torch.randint(1, S + 1, (B,), generator=g)
    nvalid[0] = S
    qkv64 = qkv.double().requires_grad_(True)
    ref = _attn_ref(qkv64, nvalid, B, S, H)
    ref.backward(dout.double())
    nv = nvalid.to(torch.int32).to(DEV)
    out, lse = ops.attention_fwd(qkv.to(DEV, dtype), nv, B, S, H)
    if dtype == torch.float32:
        assert_close(out, ref, 2e-5, 1e-5, "attn fwd")
    else:
        assert_close(out, ref, 2e-2, 2e-2, "attn fwd bf16")
    # backward consumes the kernel's own forward output (as the train step does)
    dqkv = ops.attention_bwd(qkv.to(DEV, dtype), nv, out, dout.to(DEV, dtype), lse, B, S, H)
    if dtype == torch.float32:
        assert_close(dqkv, qkv64.grad, 5e-5, 1e-4, "attn bwd")
    else:
        assert_close(dqkv, qkv64.grad, 6e-2, 3e-2, "attn bwd bf16")


# ------------------------------------------------------------------------------- embedding
def test_embed_pool_and_row_flags():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    T, D = 700, 128
    sizes = [9, 66, 66, 10, 18, 18, 18, 2, 2]
    rowoff = torch.tensor([0, 9, 75, 141, 151, 151, 151, 169, 171], dtype=torch.int32)
    ROWS = 173
    tables = torch.randn(ROWS, D, generator=g)
    idx = torch.stack([torch.randint(0, s, (T,), generator=g) for s in sizes], dim=1).to(torch.int32)
    idx[:, 7] = torch.randint(-1, 2, (T,), generator=g)  # special columns may be skipped
    idx[:, 8] = torch.randint(-1, 2, (T,), generator=g)
    want = torch.zeros(T, D, dtype=torch.float64)
    for c in range(len(sizes)):
        ok = idx[:, c] >= 0
        want[ok] += tables.double()[(rowoff[c] + idx[ok, c]).long()]
    got = ops.embed_pool_fwd(idx.to(DEV), rowoff.to(DEV), tables.to(DEV))
    assert_close(got, want, 1e-5, 1e-6, "embed fwd")
    dout = torch.randn(T, D, generator=g)
    dwant = torch.zeros(ROWS, D, dtype=torch.float64)
    for c in range(len(sizes)):
        ok = idx[:, c] >= 0
        dwant.index_add_(0, (rowoff[c] + idx[ok, c]).long(), dout.double()[ok])
    dt = torch.full((ROWS, D), float("nan"), device=DEV)
    ops.embed_pool_bwd(idx.to(DEV), rowoff.to(DEV), dout.to(DEV), dt)
    assert_close(dt, dwant, 2e-4, 1e-5, "embed bwd")
    # row flags
    x = torch.randn(T, 512, generator=g)
    x[::5] = 10.0
    x[1::7] = 0.0
    x[3, 0] = 0.0
    code = torch.empty(T, dtype=torch.uint8, device=DEV)
    sp = torch.full((T, 3), 7, dtype=torch.int32, device=DEV)
    ops.row_flags(x.to(DEV), code, sp[:, 1:], 3)
    want_code = torch.where((x == 0.0).all(1), 2, torch.where((x == 10.0).all(1), 1, 0))
    assert torch.equal(code.cpu().long(), want_code)
    assert torch.equal(sp[:, 1].cpu().long(), want_code - 1)
    assert (sp[:, 0] == 7).all() and (sp[:, 2] == 7).all()


# ---------------------------------------------------------------------------------- losses
@pytest.mark.parametrize("dl_dtype", [torch.float32, torch.bfloat16])
def test_losses_vs_oracle(dl_dtype):
    ops = _ops()
    from oracle import torch_ref
    from mfp.data.spec import make_input_columns, synthetic_batch
    ic = make_input_columns("crello")
    B, S = 5, 12
    batch = synthetic_batch(ic, B, S, seed=4, ragged=True)
    g = torch.Generator().manual_seed(9)
    keys = [k for k, c in ic.items() if c.get("is_sequence") and not c.get("demo_only")]
    pred, masks = {}, {}
    col, layout = 0, {}
    for k in keys:
        c = ic[k]
        n = c["shape"][-1] * c["input_dim"] if c["type"] == "categorical" else c["shape"][-1]
        layout[k] = (col, n)
        col += n
    U = (col + 7) // 8 * 8
    logits = torch.randn(B * S, U, generator=g) * 3
    logits[0, :7] = torch.tensor([40.0, -40, 0, 0, 0, 0, 0])  # forces the 1e-7 clip branch
    for k in keys:
        c = ic[k]
        o, n = layout[k]
        sl = logits[:, o:o + n].reshape(B, S, -1)
        pred[k] = (sl.reshape(B, S, c["shape"][-1], c["input_dim"]) if c["type"] == "categorical" else sl
                   ).double().requires_grad_(True)
        masks[k] = torch.rand(B, S, generator=g) < 0.6
    masks["opacity"][:] = False  # den == 0 -> score 1.0 path
    y64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    loss_total, losses, scores, metrics = torch_ref.loss_layer(ic, y64, pred, masks, S)
    loss_total.backward()
    descr = []
    types = batch["type"].to(DEV)
    dev_keep = [types]
    for k in keys:
        c = ic[k]
        o, n = layout[k]
        tgt = batch[k].to(DEV).contiguous()
        msk = masks[k].to(torch.uint8).to(DEV).contiguous()
        dev_keep += [tgt, msk]
        d = dict(col_off=o, n_feat=c["shape"][-1] if c["type"] == "categorical" else 1,
                 n_class=c["input_dim"] if c["type"] == "categorical" else c["shape"][-1],
                 is_numerical=c["type"] != "categorical", target=tgt, mask=msk)
        if "loss_condition" in c:
            bits = sum(1 << i for i, f in enumerate(c["loss_condition"]["mask"]) if f)
            d.update(cond_idx=types, cond_stride=1, cond_bits=bits)
        descr.append(d)
    nvalid = (batch["length"].reshape(-1) + 1).to(torch.int32).to(DEV)
    sums, dl = ops.loss_fwd_bwd(logits.to(DEV), descr, nvalid, B, S, dl_dtype)
    sums = sums.cpu().double()
    for i, k in enumerate(keys):
        assert abs(sums[i, 0].item() - float(losses[k])) < 1e-4 * max(1.0, abs(float(losses[k]))), k
        assert abs(sums[i, 1].item() - float(scores[k + "_score_num"])) < 1e-4, k
        assert abs(sums[i, 2].item() - float(scores[k + "_score_den"])) < 1e-6, k
        o, n = layout[k]
        want = pred[k].grad.reshape(B * S, n)
        if dl_dtype == torch.float32:
            assert_close(dl[:, o:o + n], want, 1e-6, 1e-4, "dlogits " + k)
        else:
            assert_close(dl[:, o:o + n], want, 1e-4, 1e-2, "dlogits bf16 " + k)
    assert (dl[:, col:] == 0).all()


# ------------------------------------------------------------- RICO position-sorted loss
def _flat_layout(ic, keys):
    col, layout = 0, {}
    for k in keys:
        c = ic[k]
        n = c["shape"][-1] * c["input_dim"] if c["type"] == "categorical" else c["shape"][-1]
        layout[k] = (col, n)
        col += (n + 7) // 8 * 8
    return layout, col


@pytest.mark.parametrize("dataset,B,S", [("rico", 6, 20), ("crello", 4, 33), ("rico", 3, 300)])
def test_sort_positions_vs_oracle(dataset, B, S):
    """mfp_sort_positions == argsort of tensor_utils.py:14-44, from labels and from logits, with
    ties (stable), ragged lengths and unflagged documents (identity)."""
    ops = _ops()
    from oracle import np_ref, torch_ref
    from mfp.data.spec import make_input_columns, synthetic_batch
    ic = make_input_columns(dataset)
    batch = synthetic_batch(ic, B, S, seed=2, ragged=True)
    for k in np_ref.SORT_KEYS:          # ties: positions 1 and 3 equal position 0 on every sort key
        batch[k][:, 1] = batch[k][:, 0]
        batch[k][:, min(3, S - 1)] = batch[k][:, 0]
    keys = [k for k, c in ic.items() if c.get("is_sequence") and not c.get("demo_only")]
    layout, U = _flat_layout(ic, keys)
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(B * S, U, generator=g)
    logits[1] = logits[0]
    logits[S + 2] = logits[S]
    logits[5, layout["left"][0]:layout["left"][0] + 4] = logits[5, layout["left"][0]:layout["left"][0] + 64].max() + 1  # argmax tie -> first
    flag = torch.ones(B, dtype=torch.bool)
    flag[1] = False
    nvalid = (batch["length"].reshape(-1) + 1).to(torch.int32)
    ident = torch.arange(S)[None].expand(B, S)
    want_t = torch.where(flag[:, None], torch_ref.sort_indices(batch, ic, False, S), ident)
    pred = {k: logits[:, layout[k][0]:layout[k][0] + ic[k]["input_dim"]].reshape(B, S, 1, -1) for k in np_ref.SORT_KEYS}
    pred["length"] = batch["length"]
    want_p = torch.where(flag[:, None], torch_ref.sort_indices(pred, ic, True, S), ident)
    # cross-check the rank formulation against the numpy argsort restatement
    np_sorted = np_ref.sort_inputs({k: v.numpy() for k, v in batch.items()}, np_ref.valid_columns(ic), maxlen=S)
    assert np.array_equal(np_sorted["left"], np.take_along_axis(batch["left"].numpy(),
                                                                  torch_ref.sort_indices(batch, ic, False, S).numpy()[:, :, None], 1))
    base = (torch.arange(B) * S)[:, None]
    fl = flag.to(torch.uint8).to(DEV)
    labels = [batch[k].to(torch.int32).to(DEV).contiguous() for k in np_ref.SORT_KEYS]
    got_t = ops.sort_positions(nvalid.to(DEV), fl, B, S, labels=labels).cpu().view(B, S)
    heads = [(layout[k][0], ic[k]["input_dim"]) for k in np_ref.SORT_KEYS]
    got_p = ops.sort_positions(nvalid.to(DEV), fl, B, S, logits=logits.to(DEV), heads=heads).cpu().view(B, S)
    assert torch.equal(got_t.long(), want_t + base)
    assert torch.equal(got_p.long(), want_p + base)


@pytest.mark.parametrize("dataset", ["rico", "crello"])
@pytest.mark.parametrize("ignore", [None, "gt", "pred"])
@pytest.mark.parametrize("dl_dtype", [torch.float32, torch.bfloat16])
def test_sorted_losses_vs_oracle(dataset, ignore, dl_dtype):
    """LossLayer with sort_flag (metrics.py:180-211): per-key sums and d(loss)/d(logits) -- the
    gradient lands on the ORIGINAL logits rows -- against the torch restatement."""
    ops = _ops()
    from oracle import np_ref, torch_ref
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.hip.functions import loss_row_maps
    from mfp.models.metrics import build_loss_keys, build_loss_sort
    ic = make_input_columns(dataset)
    B, S = 5, 16
    batch = synthetic_batch(ic, B, S, seed=6, ragged=True)
    keys = [k for k, c in ic.items() if c.get("is_sequence") and not c.get("demo_only")]
    layout, U = _flat_layout(ic, keys)
    g = torch.Generator().manual_seed(10)
    logits = torch.randn(B * S, U, generator=g) * 3
    pred, masks = {}, {}
    for k in keys:
        c = ic[k]
        o, n = layout[k]
        sl = logits[:, o:o + n].reshape(B, S, -1)
        pred[k] = (sl.reshape(B, S, c["shape"][-1], c["input_dim"]) if c["type"] == "categorical" else sl
                   ).double().requires_grad_(True)
        masks[k] = torch.rand(B, S, generator=g) < 0.6
    flag = torch.tensor([True, True, False, True, False])
    y64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    loss_total, losses, scores, _ = torch_ref.loss_layer(ic, y64, pred, masks, S, sort_flag=flag, ignore_sort=ignore)
    loss_total.backward()
    dev = lambda d: {k: v.to(DEV) for k, v in d.items()}
    yt = dev(batch)
    descr = build_loss_keys(ic, layout, yt, dev(masks))
    nvalid = (batch["length"].reshape(-1) + 1).to(torch.int32).to(DEV)
    lg = logits.to(DEV)
    sort = build_loss_sort(ic, layout, yt, flag.to(DEV), ignore)
    pred_row, true_row = loss_row_maps(sort, lg, nvalid, B, S)
    assert (pred_row is None) == (ignore == "pred") and (true_row is None) == (ignore == "gt")
    sums, dl = ops.loss_fwd_bwd(lg, descr, nvalid, B, S, dl_dtype, pred_row=pred_row, true_row=true_row)
    sums = sums.cpu().double()
    plain = ops.loss_fwd_bwd(lg, descr, nvalid, B, S, None)[0].cpu().double()
    assert not torch.allclose(plain[:, 0], sums[:, 0])      # the sort does change the loss
    for i, k in enumerate(keys):
        assert abs(sums[i, 0].item() - float(losses[k])) < 1e-4 * max(1.0, abs(float(losses[k]))), k
        assert abs(sums[i, 1].item() - float(scores[k + "_score_num"])) < 1e-4, k
        assert abs(sums[i, 2].item() - float(scores[k + "_score_den"])) < 1e-6, k
        o, n = layout[k]
        want = pred[k].grad.reshape(B * S, n)
        if dl_dtype == torch.float32:
            assert_close(dl[:, o:o + n], want, 1e-6, 1e-4, "dlogits " + k)
        else:
            assert_close(dl[:, o:o + n], want, 1e-4, 1e-2, "dlogits bf16 " + k)


# ------------------------------------------------------------------------------- optimizer
def test_adam_keras_vs_oracle():
    ops = _ops()
    from oracle import np_ref
    rng = np.random.default_rng(0)
    sizes = [5000, 7, 4096, 12289, 256]
    l2 = [1e-2, 1e-2, 0.0, 1e-2, 0.0]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    n = int(off[-1])
    w = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    wd, md, vd = [torch.from_numpy(a.copy()).to(DEV) for a in (w, m, v)]
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    chunks = ops.AdamChunks(off.tolist(), DEV)
    seg_l2 = torch.tensor(l2, dtype=torch.float32, device=DEV)
    stats = torch.empty(len(sizes), 2, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    w64, m64, v64 = w.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for t in range(1, 4):
        gnp = (rng.standard_normal(n) * (10.0 if t == 2 else 0.01)).astype(np.float32)
        ops.adam_keras(wd, torch.from_numpy(gnp).to(DEV), md, vd, shadow, chunks, seg_l2, stats, step,
                       lr=1e-2, clipnorm=1.0, grad_scale=0.5)
        wsq = []
        for s in range(len(sizes)):
            sl = slice(off[s], off[s + 1])
            ge = gnp[sl].astype(np.float64) * 0.5 + 2 * l2[s] * w64[sl]
            wsq.append((w64[sl] ** 2).sum())
            gc = np_ref.clip_by_norm(ge, 1.0)
            w64[sl], m64[sl], v64[sl] = np_ref.adam_keras_step(w64[sl], gc, m64[sl], v64[sl], t, lr=1e-2)
        assert int(step.item()) == t
        np.testing.assert_allclose(stats[:, 1].cpu().numpy(), np.array(wsq), rtol=1e-4)
        np.testing.assert_allclose(wd.cpu().numpy(), w64, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(md.cpu().numpy(), m64, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(vd.cpu().numpy(), v64, rtol=1e-4, atol=1e-10)
    assert torch.equal(shadow.cpu(), wd.cpu().to(torch.bfloat16))


def test_cast_and_colsum():
    ops = _ops()
    x = torch.randn(1003, 264)
    xd = x.to(DEV)
    dst = torch.empty(x.numel(), dtype=torch.bfloat16, device=DEV)
    ops.cast_bf16(xd.reshape(-1), dst)
    assert torch.equal(dst.cpu(), x.reshape(-1).to(torch.bfloat16))
    out = torch.empty(264, device=DEV)
    ops.colsum(xd, out, 1003, 264)
    assert_close(out, x.double().sum(0), 1e-3, 1e-5, "colsum f32")
    ops.colsum(dst.reshape(1003, 264), out, 1003, 264)
    assert_close(out, x.to(torch.bfloat16).double().sum(0), 1e-3, 1e-5, "colsum bf16")


def test_transpose_cast_bf16():
    """Transposed bf16 shadow of matrices inside a flat buffer (dgrad weights of the ws GEMM)."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    segs, off = [], 40
    for r, c in [(256, 512), (512, 256), (256, 256), (40, 72)]:
        segs.append((off, r, c))
        off += r * c + 24
    # three [64][32] matrices transposed side by side into one [32][192] block
    trio = off
    for jx in range(3):
        segs.append((off, 64, 32, trio + jx * 64, 192))
        off += 64 * 32
    w = torch.randn(off, generator=g)
    out = torch.full((off,), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.transpose_cast_bf16(w.to(DEV), out, ops.TransposeTable(segs, DEV))
    got = out.float().cpu()
    want = torch.full((off,), 7.0)
    for sg in segs:
        o, r, c = sg[:3]
        if len(sg) == 3:
            want[o:o + r * c] = bf16_round(w[o:o + r * c].view(r, c).t().contiguous()).reshape(-1)
        else:
            jx = (sg[3] - trio) // 64
            want[trio:trio + 32 * 192].view(32, 192)[:, jx * 64:(jx + 1) * 64] = bf16_round(w[o:o + r * c].view(r, c).t())
    assert torch.equal(got, want)


def test_embed_onehot_table_gradient():
    """bf16 path: table gradient = onehot^T dh on the wgrad GEMM; exact against a double reference
    on the bf16-rounded dh (counts are exact, accumulation is f32)."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    T, D, dims = 2048, 64, [5, 9, 9, 20, 3]
    ncol = len(dims) + 1                      # last column repeats table 1 (shared table, like color)
    offs = [0]
    for d in dims:
        offs.append(offs[-1] + d)
    rowoff = torch.tensor(offs[:-1] + [offs[1]], dtype=torch.int32)
    rows = offs[-1]
    rows_pad = (rows + 7) // 8 * 8
    idx = torch.stack([torch.randint(-1, d, (T,), generator=g) for d in dims + [dims[1]]], dim=1).to(torch.int32)
    dh = bf16_round(torch.randn(T, D, generator=g))
    P = ops.embed_onehot(idx.to(DEV), rowoff.to(DEV), rows_pad)
    want_P = torch.zeros(T, rows_pad)
    for c in range(ncol):
        ok = idx[:, c] >= 0
        want_P[ok.nonzero()[:, 0], (rowoff[c] + idx[ok, c]).long()] += 1
    assert torch.equal(P.float().cpu(), want_P)
    out = torch.full((rows_pad, D), 9.0, device=DEV)
    ops.gemm(P, dh.to(DEV, torch.bfloat16), rows_pad, D, T, a_kmajor=False, b_kmajor=False, out=out,
             splitk=ops.wgrad_splitk(T, rows_pad, D))
    want = want_P.double().t() @ dh.double()
    assert_close(out, want, 1e-3, 1e-5, "onehot table gradient")


def test_layernorm_bwd_deferred_partials():
    """mfp_layernorm_bwd with dgamma = dbeta = NULL leaves the partials; mfp_reduce_partials sums
    them later (ops.layernorm_bwd(defer=...)) -- same result as the in-line reduction."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    T, D = 1000, 256
    x = torch.randn(T, D, generator=g).to(DEV)
    gamma, beta = (torch.rand(D, generator=g) + 0.5).to(DEV), torch.randn(D, generator=g).to(DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, torch.bfloat16)
    dy = torch.randn(T, D, generator=g).to(DEV, torch.bfloat16)
    dres = torch.randn(T, D, generator=g).to(DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    outs = []
    for deferred in (False, True):
        dg, db, cs = torch.empty(D, device=DEV), torch.empty(D, device=DEV), torch.empty(D, device=DEV)
        pending = []
        dx, dd = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, drop=(cs, 0.1, 7, 3, step),
                                   defer=(lambda fn, *t: pending.append(fn)) if deferred else None)
        for fn in pending:
            fn()
        outs.append((dx, dd, dg, db, cs))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("T,D", [(1000, 256), (24700, 512)])
def test_layernorm_bwd_res16(T, D):
    """mfp_layernorm_bwd_res16 (residual gradient stream in bf16: dres read and dx written as bf16) against
    mfp_layernorm_bwd on the same bf16-valued residual: dx equal to bf16 rounding, the masked copy, the parameter-gradient
    and bias-gradient sums identical (they are formed from the f32 values before the store)."""
    ops = _ops()
    g = torch.Generator().manual_seed(22)
    x = torch.randn(T, D, generator=g).to(DEV)
    gamma, beta = (torch.rand(D, generator=g) + 0.5).to(DEV), torch.randn(D, generator=g).to(DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, torch.bfloat16)
    dy = torch.randn(T, D, generator=g).to(DEV, torch.bfloat16)
    dres16 = torch.randn(T, D, generator=g).to(DEV, torch.bfloat16)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    outs = []
    for dres in (dres16.float(), dres16):
        dg, db, cs = torch.empty(D, device=DEV), torch.empty(D, device=DEV), torch.empty(D, device=DEV)
        dx, dd = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, drop=(cs, 0.1, 7, 3, step))
        outs.append((dx, dd, dg, db, cs))
    (dx32, dd32, dg32, db32, cs32), (dx16, dd16, dg16, db16, cs16) = outs
    assert dx16.dtype == torch.bfloat16 and dx32.dtype == torch.float32
    # (the two instantiations may contract their multiply-adds differently: an f32 last-bit difference now and then lands on the
    #  other side of a bf16 rounding boundary -- one bf16 step on a handful of the 12.6 M values of the larger case)
    want = dx32.to(torch.bfloat16)
    off = dx16 != want
    assert off.float().mean().item() < 1e-5
    assert ((dx16.float() - want.float()).abs() <= want.float().abs() * 2.0 ** -7 + 1e-30).all()
    assert (dd16 != dd32).float().mean().item() < 1e-5
    for a, b in ((dg16, dg32), (db16, db32), (cs16, cs32)):
        assert_close(a, b, 1e-3, 1e-5, "parameter-gradient sums")
    # no residual: the plain bf16 output
    dxn = ops.layernorm_bwd(dy, x, gamma, mean, rstd, None, torch.empty(D, device=DEV), torch.empty(D, device=DEV),
                            dx=torch.empty(T, D, dtype=torch.bfloat16, device=DEV))
    dxf = ops.layernorm_bwd(dy, x, gamma, mean, rstd, None, torch.empty(D, device=DEV), torch.empty(D, device=DEV))
    assert (dxn != dxf.to(torch.bfloat16)).float().mean().item() < 1e-5
    # mfp_layernorm_bwd_xhat: the same from the bf16 stash (x - mean) rstd -- against a double restatement from that stash
    xh = ((x - mean[:, None]) * rstd[:, None]).to(torch.bfloat16)
    dgx, dbx, csx = torch.empty(D, device=DEV), torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    dxx, ddx = ops.layernorm_bwd(dy, None, gamma, None, rstd, dres16, dgx, dbx, drop=(csx, 0.1, 7, 3, step), xhat=xh)
    _check_ln_from_xhat(dy, xh, gamma, rstd, dres16, dxx, ddx, dgx, dbx, csx, 0.1)
    # mfp_dropout_bwd_res16 (bf16 in, bf16 out) == mfp_dropout_bwd on the same values in f32
    c32, c16 = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    m32 = ops.dropout_bwd(dres16.float(), torch.bfloat16, c32, 0.1, 7, 5, step)
    m16 = ops.dropout_bwd(dres16, torch.bfloat16, c16, 0.1, 7, 5, step)
    assert torch.equal(m16, m32) and torch.equal(c16, c32)
    assert 0.05 < (m16 == 0).float().mean().item() < 0.15


@pytest.mark.parametrize("M,N,T,sk", [(256, 512, 4096, 8), (344, 256, 4096, 16), (1384, 256, 2048, 8), (136, 72, 1100, 8)])
def test_gemm_streaming_wgrad(M, N, T, sk):
    """gemm_wg_kernel (bf16, splitk % 8 == 0): C = A^T B over the token dimension with ragged last
    k-tiles / tiles, fused bias-gradient column sums and skipped rows, against a double reference."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + T)
    A = bf16_round(torch.randn(T, M, generator=g))
    Bm = bf16_round(torch.randn(T, N, generator=g))
    code = (torch.rand(T, generator=g) < 0.25).to(torch.uint8)
    out = torch.full((M, N), 5.0, device=DEV)
    colsum = torch.full((M,), 5.0, device=DEV)
    ops.gemm(A.to(DEV, torch.bfloat16), Bm.to(DEV, torch.bfloat16), M, N, T, a_kmajor=False, b_kmajor=False, out=out,
             colsum=colsum, rowskip_a=code.to(DEV), splitk=sk)
    keep = (code == 0).double()[:, None]
    want = (A.double() * keep).t() @ Bm.double()
    assert_close(out, want, 2e-4 * math.sqrt(T), 1e-5, "streaming wgrad")
    assert_close(colsum, (A.double() * keep).sum(0), 2e-4 * math.sqrt(T), 1e-5, "bias gradient")
    out2 = torch.empty((M, N), device=DEV)
    ops.gemm(A.to(DEV, torch.bfloat16), Bm.to(DEV, torch.bfloat16), M, N, T, a_kmajor=False, b_kmajor=False, out=out2,
             splitk=sk)
    assert_close(out2, A.double().t() @ Bm.double(), 2e-4 * math.sqrt(T), 1e-5, "streaming wgrad plain")


@pytest.mark.parametrize("T,splitk", [(4096, None), (1100, 8), (32768, None), (8192, 16), (2048, 2), (1100, 4), (512, 1)])
def test_wgrad_group(T, splitk):
    """mfp_wgrad_group: several products over the same tokens in one launch with the split-K reduction
    inside it (last arriver per tile sums the slabs).  Ragged tiles (M = 344, 1384, N = 72), ragged
    last k-tiles, bias-gradient column sums, masked rows, strided operand views (A and B of one job are
    column slices of wider matrices, as dqkv / the one-hot matrix are), outputs with ldc > N; against a
    double reference.  Launched three times on the same ticket words: results must be BIT-identical
    (fixed summation order, tickets left zero)."""
    ops = _ops()
    g = torch.Generator().manual_seed(T)
    shapes = [(256, 512, True, False), (344, 256, False, False), (1384, 256, True, False), (136, 72, True, True),
              (768, 256, True, False)]
    if T > 8192:   # the timed token count: block-shaped jobs + the encoder tail (masked rows, ragged table tiles)
        shapes = [(256, 512, True, True), (344, 256, False, False), (768, 256, True, False), (512, 256, True, False)]
    wide_a = bf16_round(torch.randn(T, 2048, generator=g)).to(DEV, torch.bfloat16)
    wide_b = bf16_round(torch.randn(T, 1024, generator=g)).to(DEV, torch.bfloat16)
    code = (torch.rand(T, generator=g) < 0.25).to(torch.uint8).to(DEV)
    jobs, want = [], []
    ca = cb = 0
    for M, N, cs, skip in shapes:
        if ca + M > 2048:
            ca = 0
        if cb + N > 1024:
            cb = 0
        A, Bm = wide_a[:, ca:ca + M], wide_b[:, cb:cb + N]
        ca, cb = ca + (M + 7) // 8 * 8, cb + N
        out = torch.full((M, N + 8), 7.0, device=DEV)[:, :N]
        j = dict(A=A, B=Bm, out=out, M=M, N=N)
        keep = torch.ones(T, 1, dtype=torch.float64)
        if cs:
            j["colsum"] = torch.full((M,), 7.0, device=DEV)
        if skip:
            j["rowskip"] = code
            keep = (code.cpu() == 0).double()[:, None]
        jobs.append(j)
        Ad = A.cpu().double() * keep
        want.append((Ad.t() @ Bm.cpu().double(), Ad.sum(0)))
    runs = []
    for _ in range(3):
        for j in jobs:
            j["out"].fill_(7.0)
        ops.wgrad_group(jobs, T, splitk)
        torch.cuda.synchronize()
        runs.append([(j["out"].clone(), j["colsum"].clone() if "colsum" in j else None) for j in jobs])
    tol = 2e-4 * math.sqrt(T)
    for j, (w, wc), (o, c) in zip(jobs, want, runs[0]):
        assert_close(o, w, tol, 1e-5, "grouped wgrad %dx%d" % (j["M"], j["N"]))
        if c is not None:
            assert_close(c, wc, tol, 1e-5, "grouped bias gradient %d" % j["M"])
        assert (j["out"]._base[:, j["N"]:] == 7.0).all()          # nothing written beyond N
    for r in runs[1:]:
        for (o0, c0), (o, c) in zip(runs[0], r):
            assert torch.equal(o0, o) and (c0 is None or torch.equal(c0, c))
    assert int(ops._wgrad_tickets(DEV).abs().sum()) == 0
    # the deferred form (mfp_wgrad_group_partial + mfp_wgrad_reduce: what the train step runs): two groups pending at
    # once, each in its own slab buffer, ONE reduction launch -- bit-identical to the in-launch reduction
    pending = []
    for j in jobs:
        j["out"].fill_(7.0)
        if "colsum" in j:
            j["colsum"].fill_(7.0)
    half = max(1, len(jobs) // 2)
    sk = splitk or ops.wgrad_group_splitk(jobs, T)        # (the split chosen for the whole group above: same summation order)
    ops.wgrad_group(jobs[:half], T, sk, defer=pending)
    ops.wgrad_group(jobs[half:], T, sk, defer=pending)
    assert len(pending) == 2 and all((j["out"] == 7.0).all() for j in jobs)      # nothing written yet
    ops.wgrad_reduce(pending)
    torch.cuda.synchronize()
    assert pending == []
    for j, (o0, c0) in zip(jobs, runs[0]):
        assert torch.equal(j["out"], o0), (j["M"], j["N"])
        if c0 is not None:
            assert torch.equal(j["colsum"], c0)
        assert (j["out"]._base[:, j["N"]:] == 7.0).all()


@pytest.mark.parametrize("T,splitk", [(4096, 4), (3000, 2), (16384, None)])
def test_wgrad_group_macro_tiles(T, splitk):
    """Groups with many tiles (a d_model-512 block's four products: 128 tiles of 128 x 128) take gemm_wgt_kernel in the
    deferred form: 256 x 128 macro tiles, slabs written as the two standard tiles.  Against a double reference, and -- same
    split, same k order per tile -- BIT-identical to the 128 x 128 kernel with the in-launch reduction."""
    ops = _ops()
    D = 512
    g = torch.Generator().manual_seed(T)
    rnd = lambda n: bf16_round(torch.randn(T, n, generator=g)).to(DEV, torch.bfloat16)
    ops_ab = [(rnd(3 * D), rnd(D)), (rnd(2 * D), rnd(D)), (rnd(D), rnd(2 * D)), (rnd(D), rnd(D))]
    def jobs():
        return [dict(A=a, B=b, out=torch.full((a.shape[1], b.shape[1]), 7.0, device=DEV), M=a.shape[1], N=b.shape[1],
                     **({"colsum": torch.full((a.shape[1],), 7.0, device=DEV)} if i != 2 else {})) for i, (a, b) in enumerate(ops_ab)]
    j0 = jobs()
    sk = splitk or ops.wgrad_group_splitk(j0, T)
    assert sk in (1, 2, 4) or sk % 8 == 0
    ops.wgrad_group(j0, T, sk)                       # 128 x 128 units, reduction inside the launch
    j1 = jobs()
    pending = []
    ops.wgrad_group(j1, T, sk, defer=pending)        # macro tiles + mfp_wgrad_reduce
    ops.wgrad_reduce(pending)
    torch.cuda.synchronize()
    tol = 2e-4 * math.sqrt(T)
    for (a, b), x0, x1 in zip(ops_ab, j0, j1):
        want = a.cpu().double().t() @ b.cpu().double()
        assert_close(x1["out"], want, tol, 1e-5, "macro-tile wgrad %dx%d" % (x1["M"], x1["N"]))
        assert torch.equal(x0["out"], x1["out"]), (x1["M"], x1["N"])
        if "colsum" in x1:
            assert_close(x1["colsum"], a.cpu().double().sum(0), tol, 1e-5, "bias gradient")
            assert torch.equal(x0["colsum"], x1["colsum"])


def test_mx_mfma_layout():
    """Pins the operand and scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 that csrc/gemm_fp8.hip relies on (one
    instruction through mfp_debug_mx_probe, e4m3 x e4m3): lane (i = l % 16, g = l / 16) of the A (B) operand holds row
    (column) i, k = 16 g + t for bytes t < 16 and k = 64 + 16 g + (t - 16) for bytes t >= 16; lane (i, s) of the scale
    register (byte 0) scales the CONTIGUOUS block k = 32 s .. 32 s + 31 of row i by 2^(byte - 127); D[i][j] sits in
    lane j + 16 (i / 4), register i % 4."""
    from mfp import hip
    lib = hip.load()
    g = torch.Generator().manual_seed(3)
    A = (torch.randn(16, 128, generator=g)).to(torch.float8_e4m3fn)
    B = (torch.randn(16, 128, generator=g)).to(torch.float8_e4m3fn)
    sa = torch.randint(120, 135, (16, 4), generator=g)
    sb = torch.randint(120, 135, (16, 4), generator=g)

    def pack(M):      # [16 rows][128 k] -> [64 lanes][32 bytes]
        Mb = M.view(torch.uint8)
        out = torch.zeros(64, 32, dtype=torch.uint8)
        for l in range(64):
            i, gg = l % 16, l // 16
            out[l, :16] = Mb[i, 16 * gg:16 * gg + 16]
            out[l, 16:] = Mb[i, 64 + 16 * gg:64 + 16 * gg + 16]
        return out

    def pack_scale(S):      # [16 rows][4 blocks] -> [64 lanes] int32
        return torch.tensor([int(S[l % 16, l // 16]) for l in range(64)], dtype=torch.int32)

    out = torch.zeros(64, 4, device=DEV)
    dA, dB, dsa, dsb = pack(A).to(DEV), pack(B).to(DEV), pack_scale(sa).to(DEV), pack_scale(sb).to(DEV)
    rc = lib.mfp_debug_mx_probe(dA.data_ptr(), dB.data_ptr(), dsa.data_ptr(), dsb.data_ptr(), out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    Ad = (A.double().view(16, 4, 32) * torch.pow(2.0, sa.double() - 127)[:, :, None]).view(16, 128)
    Bd = (B.double().view(16, 4, 32) * torch.pow(2.0, sb.double() - 127)[:, :, None]).view(16, 128)
    want = Ad @ Bd.t()
    got = torch.zeros(16, 16, dtype=torch.float64)
    o = out.cpu().double()
    for l in range(64):
        for r in range(4):
            got[4 * (l // 16) + r, l % 16] = o[l, r]
    assert (got - want).abs().max() <= 1e-4 * want.abs().max(), (got - want).abs().max()      # (measured 2e-5: the unit's accumulation)


def _mx_quantize_host(x):
    """MX (block 32 along the last axis; scale = the smallest power of two with max / scale <= 448) on the host:
    (dequantised values, e4m3 bytes, e8m0 scale bytes)."""
    R, K = x.shape
    xb = x.double().reshape(R, K // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    safe = torch.where(amax > 0, amax, torch.ones_like(amax))
    e = torch.floor(torch.log2(safe))
    e = e + (safe / torch.pow(torch.tensor(2.0, dtype=torch.float64), e) > 1.75).double()      # no saturation: max <= 448
    sb = torch.where(amax > 0, (e + 127 - 8).clamp(min=0, max=254), torch.zeros_like(e))
    scale = torch.pow(torch.tensor(2.0, dtype=torch.float64), sb - 127)
    q = (xb / scale).clamp(-448, 448).float().to(torch.float8_e4m3fn)
    return (q.double() * scale).reshape(R, K), q.reshape(R, K).view(torch.uint8), sb.reshape(R, K // 32).to(torch.uint8)


@pytest.mark.parametrize("M,N,K,relu", [(1000, 1536, 512, False), (2050, 1024, 512, True), (300, 768, 256, False), (64, 8, 128, True)])
def test_gemm_mxfp8(M, N, K, relu):
    """mfp_gemm_mxfp8 (BASELINE config c5, v_mfma_scale_f32_16x16x128_f8f6f4): OCP MX block-scaled e4m3 operands -- weights
    quantised by mfp_quantize_mxfp8 (bytes and scales BIT-identical to the host's MX quantisation), activations quantised on
    the fly -- against the double product of the host-quantised operands (pins the instruction's operand / scale layout),
    and within the expected e4m3 error of the unquantised product.  Rows with very different magnitudes and an all-zero
    block: per-block scales keep every row's mantissa range."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g) * 1.7 * torch.exp2(torch.randint(-6, 7, (M, 1), generator=g).float())
    X[:, 32:64] = 0.0
    X = bf16_round(X)
    W = torch.randn(N, K, generator=g) * 0.05 * torch.exp2(torch.randint(-3, 4, (N, 1), generator=g).float())
    bias = torch.randn(N, generator=g) * 0.1
    Wq = torch.empty(N * K, dtype=torch.uint8, device=DEV)
    Ws = torch.empty(N * K // 32, dtype=torch.uint8, device=DEV)
    ops.quantize_mxfp8(W.to(DEV).reshape(-1), N, K, Wq, Ws)
    Wdq, Wq_ref, Ws_ref = _mx_quantize_host(W)
    assert torch.equal(Ws.cpu().view(N, K // 32), Ws_ref)
    assert torch.equal(Wq.cpu().view(N, K), Wq_ref)
    out = ops.gemm_mxfp8(X.to(DEV, torch.bfloat16), Wq.view(N, K), Ws.view(N, K // 32), M, N, K, bias=bias.to(DEV), relu=relu)
    Xdq, _, _ = _mx_quantize_host(X)
    want = Xdq @ Wdq.t() + bias.double()
    exact = X.double() @ W.double().t() + bias.double()
    if relu:
        want, exact = want.clamp(min=0), exact.clamp(min=0)
    # |error| <= bf16 rounding of the output + f32 accumulation: relative to each ROW's scale (rows differ by 2^12)
    got = out.float().cpu().double()
    row_scale = want.abs().amax(1, keepdim=True).clamp(min=1e-30)
    assert ((got - want).abs() / row_scale).max() < 1e-2, ((got - want).abs() / row_scale).max()
    if N >= 64:
        rel = ((got - exact).norm(dim=1) / exact.norm(dim=1).clamp(min=1e-30)).max()
    else:      # (a row of 8 outputs under a ReLU is no statistic: the whole matrix, rows equalised)
        rs = exact.abs().amax(1, keepdim=True).clamp(min=1e-30)
        rel = ((got - exact) / rs).norm() / (exact / rs).norm()
    # e4m3: 3 mantissa bits on both operands -> ~3.6 % rms per element, ~5 % per product term, and a sum of randomly
    # signed terms inherits the terms' relative error; EVERY row meets it (block scales)
    assert rel < 0.08, rel


@pytest.mark.parametrize("T,p", [(4096, 0.1), (1000, 0.0), (33, 0.1), (128 * 3 + 5, 0.1)])
def test_mlp_fused_fwd(T, p):
    """mfp_mlp_fused_fwd: x2 = x1 + Dropout(relu(LN(x1) W1^T + b1) W2^T + b2) in one launch
    (transformer.py:161-171,222-225) against the three-launch path it replaces (same bf16 rounding points,
    same dropout stream) and a double reference; ragged last row group."""
    ops = _ops()
    D, F = 256, 512
    g = torch.Generator().manual_seed(T)
    x1 = (torch.randn(T, D, generator=g) * (1.0 + torch.rand(T, 1, generator=g)) + 0.3 * torch.randn(T, 1, generator=g))
    gamma, beta = 1.0 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    W1 = bf16_round(torch.randn(F, D, generator=g) * 0.06)
    W2 = bf16_round(torch.randn(D, F, generator=g) * 0.05)
    b1, b2 = torch.randn(F, generator=g) * 0.1, torch.randn(D, generator=g) * 0.1
    xd, gd, bd = x1.to(DEV), gamma.to(DEV), beta.to(DEV)
    W1d, W2d, b1d, b2d = W1.to(DEV, torch.bfloat16), W2.to(DEV, torch.bfloat16), b1.to(DEV), b2.to(DEV)
    step = torch.tensor([3], dtype=torch.int32, device=DEV)
    x2, y2, mean, rstd, h = ops.mlp_fused_fwd(xd, gd, bd, W1d, b1d, W2d, b2d, (p, 1234, 6), step)
    # the path it replaces
    y2u, meanu, rstdu = ops.layernorm_fwd(xd, gd, bd, torch.bfloat16)
    hu = ops.gemm(y2u, W1d, T, F, D, a_kmajor=True, b_kmajor=True, bias=b1d, relu=True, out_dtype=torch.bfloat16)
    x2u = ops.gemm(hu, W2d, T, D, F, a_kmajor=True, b_kmajor=True, bias=b2d, residual=xd,
                   dropout=(p, 1234, 6), step_ptr=step, out_dtype=torch.float32)
    assert_close(mean, meanu.cpu().double(), 1e-5, 1e-5, "mean")
    assert_close(rstd, rstdu.cpu().double(), 1e-5, 1e-5, "rstd")
    assert (y2.float() - y2u.float()).abs().max().item() <= 0.04
    assert (y2 != y2u).float().mean().item() < 0.01
    assert (h != hu).float().mean().item() < 0.02
    assert_close(h, hu.float().cpu().double(), 3e-2, 2e-2, "h vs unfused")
    # the dropout mask is the unfused product's, element for element: dropped entries are exactly x1
    dropped_u, dropped = (x2u == xd), (x2 == xd)
    assert torch.equal(dropped, dropped_u)
    if p > 0:
        assert abs(dropped.float().mean().item() - p) < 0.02
    assert_close(x2, x2u.cpu().double(), 3e-2, 2e-2, "x2 vs unfused")
    # double reference from the kernel's own rounded intermediates
    keep = (~dropped).cpu().double() / (1.0 - p) if p > 0 else torch.ones(T, D, dtype=torch.double)
    hr = (y2.float().cpu().double() @ W1.double().t() + b1.double()).clamp(min=0)
    assert_close(h, hr, 2e-2, 1e-2, "h vs double")
    want = x1.double() + keep * (h.float().cpu().double() @ W2.double().t() + b2.double())
    assert_close(x2, want, 2e-3, 2e-3, "x2 vs double")


@pytest.mark.parametrize("T", [4096, 1000, 33, 128 * 3 + 5])
def test_mlp_fused_bwd(T):
    """mfp_mlp_fused_bwd: dh = (d_o2 W2) * [h > 0], dy2 = dh W1 in one launch against the two products it replaces
    (same bf16 rounding point: dh is rounded before the second product) and a double reference."""
    ops = _ops()
    D, F = 256, 512
    g = torch.Generator().manual_seed(T + 1)
    d_o2 = bf16_round(torch.randn(T, D, generator=g) * 0.5)
    h = bf16_round(torch.randn(T, F, generator=g).clamp(min=0))       # about half the units inactive
    W1 = bf16_round(torch.randn(F, D, generator=g) * 0.06)            # [out = hidden][in]
    W2 = bf16_round(torch.randn(D, F, generator=g) * 0.05)            # [out][in = hidden]
    dd, hd = d_o2.to(DEV, torch.bfloat16), h.to(DEV, torch.bfloat16)
    W2t = W2.t().contiguous().to(DEV, torch.bfloat16)                 # [512][256]
    W1t = W1.t().contiguous().to(DEV, torch.bfloat16)                 # [256][512]
    dh, dy2 = ops.mlp_fused_bwd(dd, hd, W2t, W1t)
    # the two launches it replaces
    dhu = ops.gemm(dd, W2t, T, F, D, a_kmajor=True, b_kmajor=True, out_dtype=torch.bfloat16, relu_bwd_aux=hd)
    dy2u = ops.gemm(dhu, W1t, T, D, F, a_kmajor=True, b_kmajor=True, out_dtype=torch.bfloat16)
    assert torch.equal((dh == 0), (dhu == 0) | (dh == 0))
    assert (dh != dhu).float().mean().item() < 0.02
    assert_close(dh, dhu.float().cpu().double(), 3e-2, 2e-2, "dh vs unfused")
    assert_close(dy2, dy2u.float().cpu().double(), 3e-2, 2e-2, "dy2 vs unfused")
    want_dh = (d_o2.double() @ W2.double()) * (h.double() > 0)
    assert_close(dh, want_dh, 2e-2, 1e-2, "dh vs double")
    assert torch.equal(dh.cpu()[h == 0], torch.zeros_like(dh.cpu()[h == 0]))       # inactive units get exactly zero
    want = dh.float().cpu().double() @ W1.double()
    assert_close(dy2, want, 2e-2, 1e-2, "dy2 vs double")


def _check_ln_from_xhat(dy, xh, gamma, rstd, dres, dx, ddrop, dg, db, cs, p):
    """LayerNorm backward restated in double from the bf16 x-hat stash (what the x-hat form of ln_bwd_tile computes)."""
    d64, x64, g64 = dy.double().cpu(), xh.double().cpu(), gamma.double().cpu()
    gy = d64 * g64
    want = rstd.double().cpu()[:, None] * (gy - gy.mean(-1, keepdim=True) - x64 * (gy * x64).mean(-1, keepdim=True)) + dres.double().cpu()
    assert_close(dx, want, 2e-2, 1e-2, "dx from x-hat")
    assert_close(dg, (d64 * x64).sum(0), 2e-3 * float((d64 * x64).sum(0).abs().max()), 1e-4, "dgamma from x-hat")
    assert_close(db, d64.sum(0), 2e-3 * float(d64.sum(0).abs().max()), 1e-4, "dbeta from x-hat")
    if ddrop is not None:
        keep = (ddrop != 0).cpu()
        scale = 1.0 / (1.0 - p)
        assert abs(keep.double().mean().item() - (1.0 - p)) < 0.02
        assert_close(ddrop.double().cpu()[keep], (want * scale)[keep], 3e-2, 2e-2, "masked copy from x-hat")
        # (the kernel sums the f32 values, the reference here their bf16 roundings: sqrt(T) roundings of ~2^-8 |v|)
        assert_close(cs, ddrop.double().cpu().sum(0), 0.03 * dy.shape[0] ** 0.5, 1e-3, "colsum from x-hat")


@pytest.mark.parametrize("T,p", [(4096, 0.1), (128 * 5, 0.0), (32768, 0.1)])
def test_mlp_bwd_ln(T, p):
    """mfp_mlp_bwd_ln (the backward of LN2 in the epilogue of the MLP half's input-gradient launch) against the two launches
    it replaces, mfp_mlp_fused_bwd + mfp_layernorm_bwd_res16 on the same operands: dh bit-identical (same code), dx / the
    dropout-masked copy equal up to an occasional bf16 step (the row arithmetic may contract differently), the parameter-gradient
    and bias-gradient sums to f32 summation-order noise -- through both reduction routes (in line, batched job)."""
    ops = _ops()
    D, F = 256, 512
    g = torch.Generator().manual_seed(T + 9)
    dd = (torch.randn(T, D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    hd = torch.randn(T, F, generator=g).clamp(min=0).to(DEV, torch.bfloat16)
    W2t = (torch.randn(F, D, generator=g) * 0.05).to(DEV, torch.bfloat16)
    W1t = (torch.randn(D, F, generator=g) * 0.06).to(DEV, torch.bfloat16)
    x = (torch.randn(T, D, generator=g) * 1.5 + 0.2).to(DEV)
    gamma, beta = (torch.rand(D, generator=g) + 0.5).to(DEV), torch.randn(D, generator=g).to(DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, torch.bfloat16)
    dres = torch.randn(T, D, generator=g).to(DEV, torch.bfloat16)
    step = torch.full((1,), 3, dtype=torch.int32, device=DEV)
    new = lambda: (torch.full((D,), float("nan"), device=DEV), torch.full((D,), float("nan"), device=DEV),
                   torch.full((D,), float("nan"), device=DEV))
    dg0, db0, cs0 = new()
    dh0, dy2 = ops.mlp_fused_bwd(dd, hd, W2t, W1t)
    dx0, do0 = ops.layernorm_bwd(dy2, x, gamma, mean, rstd, dres, dg0, db0, drop=(cs0, p, 11, 5, step))
    for batched in (False, True):
        dg1, db1, cs1 = new()
        jobs = [] if batched else None
        dh1, dx1, do1 = ops.mlp_bwd_ln(dd, hd, W2t, W1t, x, gamma, mean, rstd, dres, dg1, db1, (cs1, p, 11, 5, step), jobs=jobs)
        if batched:
            assert len(jobs) == 1
            ops.reduce_partials_batch(jobs)
        assert torch.equal(dh1, dh0)
        assert dx1.dtype == torch.bfloat16 and do1.dtype == torch.bfloat16
        assert (dx1 != dx0).float().mean().item() < 1e-4 and (do1 != do0).float().mean().item() < 1e-4
        assert_close(dx1, dx0.float().cpu().double(), 2e-2, 1e-2, "dx")
        assert torch.equal(do1 == 0, do0 == 0)
        for a, b, what in ((dg1, dg0, "dgamma"), (db1, db0, "dbeta"), (cs1, cs0, "colsum")):
            assert torch.isfinite(a).all(), what
            assert_close(a, b.cpu().double(), 2e-3 * float(b.abs().max()), 1e-4, what)
    # x-hat form: the epilogue reads the bf16 stash (x - mean) rstd instead of x -- against a double restatement from that stash
    xh = ((x - mean[:, None]) * rstd[:, None]).to(torch.bfloat16)
    dg2, db2, cs2 = new()
    dh2, dx2, do2 = ops.mlp_bwd_ln(dd, hd, W2t, W1t, None, gamma, None, rstd, dres, dg2, db2, (cs2, p, 11, 5, step), xhat=xh)
    assert torch.equal(dh2, dh0)
    _check_ln_from_xhat(dy2, xh, gamma, rstd, dres, dx2, do2, dg2, db2, cs2, p)
    # ... on HALF tiles (mfp_mlp_bwd_ln_half: two workgroups per 128-row tile, one row tile per wave): dh, dx and the masked copy
    # bit for bit, the parameter-gradient sums in another grouping; both reduction routes
    for batched in (False, True):
        dg3, db3, cs3 = new()
        jobs = [] if batched else None
        dh3, dx3, do3 = ops.mlp_bwd_ln(dd, hd, W2t, W1t, None, gamma, None, rstd, dres, dg3, db3, (cs3, p, 11, 5, step), jobs=jobs, xhat=xh,
                                       half_tiles=True)
        if batched:
            ops.reduce_partials_batch(jobs)
        assert torch.equal(dh3, dh0) and torch.equal(dx3, dx2) and torch.equal(do3, do2)
        for a, b, what in ((dg3, dg2, "dgamma"), (db3, db2, "dbeta"), (cs3, cs2, "colsum")):
            assert_close(a, b.cpu().double(), 1e-4 * float(b.abs().max()), 1e-5, "half tiles: " + what)


@pytest.mark.parametrize("T", [4096, 1000, 33, 128 * 3 + 5])
def test_qkv_fused_fwd(T):
    """mfp_qkv_fused_fwd: qkv = LN(x) Wqkv^T + b in one launch (transformer.py:216-217,85-90) against ln_fwd + the
    product it replaces (same bf16 rounding point) and a double reference; ragged last row group."""
    ops = _ops()
    D, N = 256, 768
    g = torch.Generator().manual_seed(T + 2)
    x = (torch.randn(T, D, generator=g) * (1.0 + torch.rand(T, 1, generator=g)) + 0.3 * torch.randn(T, 1, generator=g))
    gamma, beta = 1.0 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    W = bf16_round(torch.randn(N, D, generator=g) * 0.06)
    bias = torch.randn(N, generator=g) * 0.1
    xd, gd, bd, Wd, biasd = x.to(DEV), gamma.to(DEV), beta.to(DEV), W.to(DEV, torch.bfloat16), bias.to(DEV)
    qkv, y1, mean, rstd = ops.qkv_fused_fwd(xd, gd, bd, Wd, biasd)
    y1u, meanu, rstdu = ops.layernorm_fwd(xd, gd, bd, torch.bfloat16)
    qkvu = ops.gemm(y1u, Wd, T, N, D, a_kmajor=True, b_kmajor=True, bias=biasd, out_dtype=torch.bfloat16)
    assert_close(mean, meanu.cpu().double(), 1e-5, 1e-5, "mean")
    assert_close(rstd, rstdu.cpu().double(), 1e-5, 1e-5, "rstd")
    assert (y1.float() - y1u.float()).abs().max().item() <= 0.04
    assert (y1 != y1u).float().mean().item() < 0.01
    assert (qkv != qkvu).float().mean().item() < 0.02
    assert_close(qkv, qkvu.float().cpu().double(), 3e-2, 2e-2, "qkv vs unfused")
    want = y1.float().cpu().double() @ W.double().t() + bias.double()
    assert_close(qkv, want, 2e-2, 1e-2, "qkv vs double")


@pytest.mark.parametrize("T,p", [(4096, 0.1), (64 * 5, 0.0), (16384, 0.1), (1024, None)])
def test_dgrad_qkv_ln_half(T, p):
    """mfp_dgrad_qkv_ln_half: dy1 = dqkv Wqkv on 64-row tiles with the x-hat backward of LN1 on the tile, against the launch pair
    it replaces (mfp_dgrad_qkv on the half-size workgroups + mfp_layernorm_bwd_xhat) -- dx and the masked copy up to an occasional
    bf16 step, the parameter-gradient sums to summation-order noise -- and a double restatement from the pair's own dy1;
    p = None: no masked copy (block 0); both reduction routes."""
    import os
    ops = _ops()
    D, K = 256, 768
    g = torch.Generator().manual_seed(T + 31)
    dq = (torch.randn(T, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    Wt = (torch.randn(D, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
    xh = torch.randn(T, D, generator=g).to(DEV, torch.bfloat16)
    gamma, rstd = (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.rand(T, generator=g) + 0.5).to(DEV)
    dres = torch.randn(T, D, generator=g).to(DEV, torch.bfloat16)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    new = lambda: tuple(torch.full((D,), 3.0, device=DEV) for _ in range(3))
    old = os.environ.get("MFP_FUSED_HALF")
    os.environ["MFP_FUSED_HALF"] = "1"
    try:
        dy1 = ops.dgrad_qkv(dq, Wt)
    finally:
        if old is None:
            os.environ.pop("MFP_FUSED_HALF", None)
        else:
            os.environ["MFP_FUSED_HALF"] = old
    dg0, db0, cs0 = new()
    r0 = ops.layernorm_bwd(dy1, None, gamma, None, rstd, dres, dg0, db0, drop=(cs0, p, 11, 5, step) if p is not None else None, xhat=xh)
    dx0, do0 = r0 if p is not None else (r0, None)
    for batched in (False, True):
        dg1, db1, cs1 = new()
        jobs = [] if batched else None
        r1 = ops.dgrad_qkv_ln_half(dq, Wt, xh, gamma, rstd, dres, dg1, db1, drop=(cs1, p, 11, 5, step) if p is not None else None, jobs=jobs)
        if batched:
            ops.reduce_partials_batch(jobs)
        dx1, do1 = r1 if p is not None else (r1, None)
        assert (dx1 != dx0).float().mean().item() < 1e-4
        assert_close(dx1, dx0.float().cpu().double(), 2e-2, 1e-2, "dx")
        if p is not None:
            assert (do1 != do0).float().mean().item() < 1e-4 and torch.equal(do1 == 0, do0 == 0)
        for a, b, what in ((dg1, dg0, "dgamma"), (db1, db0, "dbeta")) + (((cs1, cs0, "colsum"),) if p is not None else ()):
            assert torch.isfinite(a).all(), what
            assert_close(a, b.cpu().double(), 2e-3 * float(b.abs().max()), 1e-4, what)
    _check_ln_from_xhat(dy1, xh, gamma, rstd, dres, dx1, do1, dg1, db1, cs1, p if p is not None else 0.0)


@pytest.mark.parametrize("T", [4096, 1000, 33, 128 * 3 + 5])
def test_dgrad_qkv(T):
    """mfp_dgrad_qkv: dy1 = dqkv Wqkv (K = 768) in the activation-stationary kernel against the product it
    replaces and a double reference; ragged last row group."""
    ops = _ops()
    D, K = 256, 768
    g = torch.Generator().manual_seed(T + 3)
    dq = bf16_round(torch.randn(T, K, generator=g) * 0.5)
    W = bf16_round(torch.randn(K, D, generator=g) * 0.05)             # [out = 768][in = 256]
    Wt = W.t().contiguous().to(DEV, torch.bfloat16)                   # [256][768]
    dqd = dq.to(DEV, torch.bfloat16)
    dy = ops.dgrad_qkv(dqd, Wt)
    dyu = ops.gemm(dqd, Wt, T, D, K, a_kmajor=True, b_kmajor=True, out_dtype=torch.bfloat16)
    assert_close(dy, dyu.float().cpu().double(), 2e-2, 1e-2, "vs the weight-stationary product")
    assert_close(dy, dq.double() @ W.double(), 2e-2, 1e-2, "vs double")


@pytest.mark.parametrize("half", ["0", "1"])
@pytest.mark.parametrize("K", [768, 256])
def test_dgrad_workgroup_sizes(K, half, monkeypatch):
    """The 128-row and the half-size (64 rows, two workgroups per CU: the form a 128-document batch -- BASELINE config
    c4 per GPU -- selects) kernels of mfp_dgrad_qkv / mfp_dgrad_d256 against a double reference, and against each
    other BIT FOR BIT (same k order inside every accumulator); ragged last row group."""
    ops = _ops()
    D, T = 256, 128 * 5 + 37
    g = torch.Generator().manual_seed(K)
    dq = bf16_round(torch.randn(T, K, generator=g) * 0.5)
    W = bf16_round(torch.randn(K, D, generator=g) * 0.05)
    Wt = W.t().contiguous().to(DEV, torch.bfloat16)
    dqd = dq.to(DEV, torch.bfloat16)
    fn = ops.dgrad_qkv if K == 768 else ops.dgrad_d256
    monkeypatch.setenv("MFP_FUSED_HALF", half)
    dy = fn(dqd, Wt)
    assert_close(dy, dq.double() @ W.double(), 2e-2, 1e-2, "vs double")
    monkeypatch.setenv("MFP_FUSED_HALF", "1" if half == "0" else "0")
    other = fn(dqd, Wt)
    assert torch.equal(dy.view(torch.int16), other.view(torch.int16))


@pytest.mark.parametrize("T", [4096, 1000, 33, 128 * 3 + 5])
def test_encoder_dense2(T):
    """mfp_encoder_dense2: h += sum_j [code_j == 0] (x_j W_j^T + b_j) in one launch (encoder.py:156-160,174-175)
    against the two row-skipping accumulate products it replaces and a double reference; ragged last row group."""
    ops = _ops()
    D, K = 256, 512
    g = torch.Generator().manual_seed(T + 4)
    xs = [bf16_round(torch.randn(T, K, generator=g)) for _ in range(2)]
    Ws = [bf16_round(torch.randn(D, K, generator=g) * 0.05) for _ in range(2)]
    bs = [torch.randn(D, generator=g) * 0.1 for _ in range(2)]
    codes = [(torch.rand(T, generator=g) < 0.3).to(torch.uint8) * torch.randint(1, 3, (T,), generator=g, dtype=torch.uint8) for _ in range(2)]
    h0 = torch.randn(T, D, generator=g)
    xd = [x.to(DEV, torch.bfloat16) for x in xs]
    Wd = [w.to(DEV, torch.bfloat16) for w in Ws]
    bd = [b.to(DEV) for b in bs]
    cd = [c.to(DEV) for c in codes]
    h = ops.encoder_dense2(xd, Wd, bd, cd, h0.to(DEV))
    hu = h0.to(DEV)
    for j in range(2):
        ops.gemm(xd[j], Wd[j], T, D, K, a_kmajor=True, b_kmajor=True, out=hu, accum=True, rowskip=cd[j], bias=bd[j])
    assert_close(h, hu.cpu().double(), 1e-3, 1e-3, "vs the two accumulate products")
    want = h0.double()
    for j in range(2):
        want = want + (codes[j] == 0).double()[:, None] * (xs[j].double() @ Ws[j].double().t() + bs[j].double())
    assert_close(h, want, 1e-3, 1e-3, "vs double")


@pytest.mark.parametrize("T,K", [(4096, 1384), (1000, 1384), (33, 200), (128 * 3 + 5, 128), (300, 1536)])
def test_dgrad_rows(T, K):
    """mfp_dgrad_rows: C = A[:, :K] Wt^T with a zero-padded transposed weight copy (decoder heads input gradient),
    run-time number of 128-column pieces, the last piece reaching past the row end; against a double reference and
    the LDS-tiled product it replaces."""
    ops = _ops()
    D = 256
    g = torch.Generator().manual_seed(T + K)
    lda = (K + 7) // 8 * 8
    ldw = (K + 127) // 128 * 128
    A = bf16_round(torch.randn(T, lda, generator=g) * 0.5)
    A[:, K:] = 0          # (pad columns of the logits gradient are zero by construction)
    W = bf16_round(torch.randn(K, D, generator=g) * 0.05)             # [out = K][in = 256]
    Wt = torch.zeros(D, ldw)
    Wt[:, :K] = W.t()
    Ad, Wtd = A.to(DEV, torch.bfloat16), Wt.to(DEV, torch.bfloat16)
    C = ops.dgrad_rows(Ad, Wtd, K)
    want = A[:, :K].double() @ W.double()
    assert_close(C, want, 2e-3, 2e-3, "vs double")
    Cu = ops.gemm(Ad, W.to(DEV, torch.bfloat16), T, D, K, a_kmajor=True, b_kmajor=False, lda=lda, out_dtype=torch.float32)
    assert_close(C, Cu.cpu().double(), 1e-3, 1e-3, "vs the tiled product")


@pytest.mark.parametrize("T", [4096, 1000, 33])
def test_dgrad_d256(T):
    """mfp_dgrad_d256: dx = dy W for the 256 -> 256 attention output projection, against a double reference."""
    ops = _ops()
    D = 256
    g = torch.Generator().manual_seed(T + 9)
    dy = bf16_round(torch.randn(T, D, generator=g) * 0.5)
    W = bf16_round(torch.randn(D, D, generator=g) * 0.06)             # [out][in]
    dx = ops.dgrad_d256(dy.to(DEV, torch.bfloat16), W.t().contiguous().to(DEV, torch.bfloat16))
    assert_close(dx, dy.double() @ W.double(), 2e-2, 1e-2, "vs double")


@pytest.mark.parametrize("B,p", [(3, 0.0), (5, 0.1), (1, 0.0)])
def test_attn_block_fwd(B, p):
    """mfp_attn_block_fwd: x1 = x + Dropout(MHSA(LN1(x)) Wo^T + bo) in ONE launch (transformer.py:211-221,60-99; documents
    of exactly 128 positions) against the three launches it replaces (mfp_qkv_fused_fwd, mfp_attention_fwd, the output
    projection with residual and the same dropout stream) and a double reference; ragged key-padding masks."""
    ops = _ops()
    S, D, H = 128, 256, 8
    T = B * S
    g = torch.Generator().manual_seed(100 * B + int(p * 10))
    x = torch.randn(T, D, generator=g) * (1.0 + torch.rand(T, 1, generator=g)) + 0.3 * torch.randn(T, 1, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    Wqkv = bf16_round(torch.randn(3 * D, D, generator=g) * 0.08)
    bqkv = torch.randn(3 * D, generator=g) * 0.1
    Wo = bf16_round(torch.randn(D, D, generator=g) * 0.06)
    bo = torch.randn(D, generator=g) * 0.1
    nvalid = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32)
    nvalid[0] = S
    step = torch.full((1,), 2, dtype=torch.int32, device=DEV)
    dev = lambda t, dt=None: t.to(DEV, dt) if dt else t.to(DEV)
    xd, gd, bd, bqd, bod, nvd = dev(x), dev(gamma), dev(beta), dev(bqkv), dev(bo), dev(nvalid)
    Wqd, Wod = dev(Wqkv, torch.bfloat16), dev(Wo, torch.bfloat16)
    x1, y1, mean, rstd, qkv, a, lse = ops.attn_block_fwd(xd, gd, bd, Wqd, bqd, Wod, bod, nvd, B, S, H, (p, 7, 3), step)
    # the three launches
    qkvu, y1u, meanu, rstdu = ops.qkv_fused_fwd(xd, gd, bd, Wqd, bqd)
    au, lseu = ops.attention_fwd(qkvu, nvd, B, S, H)
    x1u = ops.gemm(au, Wod, T, D, D, a_kmajor=True, b_kmajor=True, bias=bod, residual=xd, dropout=(p, 7, 3), step_ptr=step,
                   out_dtype=torch.float32)
    assert torch.equal(y1.view(torch.int16), y1u.view(torch.int16)) and torch.equal(mean, meanu) and torch.equal(rstd, rstdu)
    assert torch.equal(qkv.view(torch.int16), qkvu.view(torch.int16))          # same products, same order
    assert_close(lse, lseu.cpu().double(), 1e-4, 1e-5, "lse vs attention kernel")
    assert_close(a, au.float().cpu().double(), 2e-2, 2e-2, "a vs attention kernel")
    assert (a != au).float().mean().item() < 0.15       # (two key blocks with a running maximum: last-bit differences)
    # double reference from the kernel's own bf16 q | k | v
    q64 = qkv.float().cpu().double().view(B, S, 3, H, 32)
    sc = torch.einsum("bqhd,bkhd->bhqk", q64[:, :, 0], q64[:, :, 1]) / 32 ** 0.5
    km = (torch.arange(S)[None, :] >= nvalid[:, None]).double() * -1e9
    sc = sc + km[:, None, None, :]
    want_lse = torch.logsumexp(sc, dim=-1)
    want_a = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, dim=-1), q64[:, :, 2]).reshape(T, D)
    assert_close(lse, want_lse, 1e-3, 1e-4, "lse vs double")
    assert_close(a, want_a, 2e-2, 2e-2, "a vs double")
    if p == 0.0:
        want_x1 = x.double() + a.float().cpu().double() @ Wo.double().t() + bo.double()
        assert_close(x1, want_x1, 2e-3, 1e-3, "x1 vs double")
    # same dropout mask as the projection kernel: where both kept / dropped the values agree
    assert_close(x1, x1u.cpu().double(), 3e-2, 2e-2, "x1 vs the three launches")


@pytest.mark.parametrize("dataset,B,S,want_logits,p", [("crello", 3, 128, True, 0.0), ("crello", 5, 12, True, 0.1),
                                                        ("crello", 4, 96, False, 0.1), ("rico", 6, 40, True, 0.1)])
def test_heads_loss_fused(dataset, B, S, want_logits, p):
    """mfp_heads_loss_fwd_bwd: heads forward + LossLayer + heads input gradient in ONE launch (decoder.py:95-111,
    metrics.py:213-299) against the launches it replaces on the same bf16 operands (mfp_gemm heads forward ->
    mfp_loss_fwd_bwd -> dlogits W) and a double reference for dx; Crello heads in ModelLayout's 8-aligned layout, ragged
    documents, a key without any loss, a loss condition, a tile that is not full (T % 128 != 0)."""
    ops = _ops()
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.params import ModelLayout
    ic = make_input_columns(dataset)
    lay = ModelLayout(ic, 256, 1)
    U, D, T = lay.Upad, 256, B * S
    batch = synthetic_batch(ic, B, S, seed=4, ragged=True)
    g = torch.Generator().manual_seed(90 + B)
    x = bf16_round(torch.randn(T, D, generator=g))
    W = torch.zeros(U, D)
    bias = torch.zeros(U)
    descr, keep_alive = [], []
    types = batch["type"].to(DEV)
    for k in lay.head_order:
        c = ic[k]
        o, n = lay.head_cols[k]
        W[o:o + n] = bf16_round(torch.randn(n, D, generator=g) * (0.15 if c["type"] == "categorical" else 0.05))
        bias[o:o + n] = torch.randn(n, generator=g) * 0.1
        tgt = batch[k].to(DEV).contiguous()
        msk = (torch.rand(B, S, generator=g) < 0.3).to(torch.uint8).to(DEV).contiguous()
        if k == lay.head_order[-1] and dataset == "rico" or k == "opacity":
            msk.zero_()          # a key that carries no loss at all
        keep_alive += [tgt, msk]
        d = dict(col_off=o, n_feat=c["shape"][-1] if c["type"] == "categorical" else 1,
                 n_class=c["input_dim"] if c["type"] == "categorical" else c["shape"][-1],
                 is_numerical=c["type"] != "categorical", target=tgt, mask=msk)
        if "loss_condition" in c:
            bits = sum(1 << i for i, f in enumerate(c["loss_condition"]["mask"]) if f)
            d.update(cond_idx=types, cond_stride=1, cond_bits=bits)
        descr.append(d)
    assert ops.heads_loss_fused_ok(descr, U, D)
    nvalid = (batch["length"].reshape(-1) + 1).to(torch.int32).to(DEV)
    xd, Wd, bd = x.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16), bias.to(DEV)
    step = torch.full((1,), 3, dtype=torch.int32, device=DEV)
    drop = (p, 11, 6, step) if p > 0 else None
    part, dl, logits, dx, dxd = ops.heads_loss_fused(xd, Wd, bd, descr, nvalid, B, S, want_logits=want_logits, drop=drop)
    sums = torch.zeros(len(descr) * 3, device=DEV)
    ops.reduce_partials(part, sums, 3 * len(descr))
    sums = sums.view(len(descr), 3).cpu().double()
    # the launches it replaces
    logits_u = ops.gemm(xd, Wd, T, U, D, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, bias=bd)
    sums_u, dl_u = ops.loss_fwd_bwd(logits_u, descr, nvalid, B, S, torch.bfloat16)
    if want_logits:
        assert_close(logits, logits_u.cpu().double(), 1e-5, 1e-4, "logits vs the heads GEMM")
    else:
        assert logits is None
    su = sums_u.cpu().double()
    for i, k in enumerate(lay.head_order):
        assert abs(sums[i, 0] - su[i, 0]) <= 1e-4 * max(1.0, abs(su[i, 0])), (k, sums[i], su[i])
        assert abs(sums[i, 1] - su[i, 1]) <= 1e-3 * max(1.0, abs(su[i, 1])), (k, sums[i], su[i])
        assert sums[i, 2] == su[i, 2], (k, sums[i], su[i])
    if dataset == "crello":
        assert sums[lay.head_order.index("opacity"), 2] == 0
    assert_close(dl, dl_u.float().cpu().double(), 1e-2, 1e-5, "dlogits vs the loss kernels")
    pad = torch.ones(U, dtype=torch.bool)
    for k in lay.head_order:
        o, n = lay.head_cols[k]
        pad[o:o + n] = False
    assert (dl[:, pad.to(DEV)] == 0).all()
    # dx = dlogits W from the kernel's own bf16 dlogits
    want_dx = dl.float().cpu().double() @ W.double()
    assert_close(dx, want_dx, 2e-3, 1e-6 + 2e-3 * want_dx.abs().max().item(), "dx vs double")
    if p > 0:
        kept = dxd != 0
        frac = kept.float().mean().item()
        nz = (dx != 0).float().mean().item()
        assert abs(frac - (1 - p) * nz) < 0.02, (frac, nz)
        assert_close(dxd[kept], (dx[kept] / (1 - p)).cpu().double(), 1e-2, 1e-6, "masked copy")
        # the same mask as mfp_dgrad_rows draws for this (seed, offset, step)
        ldw = (U + 127) // 128 * 128
        Wt = torch.zeros(D, ldw, dtype=torch.bfloat16, device=DEV)
        Wt[:, :U] = Wd.t()
        _, dxd_u = ops.dgrad_rows(dl, Wt, U, drop=drop)
        both = (dx != 0) & (ops.dgrad_rows(dl, Wt, U) != 0)
        assert torch.equal((dxd != 0) & both, (dxd_u != 0) & both)
    # round 6: the same launch on 64-row tiles (mfp_heads_loss_fwd_bwd_half; c4's per-GPU shape) -- every row walks the same
    # instructions: logits, d(logits), dx and its masked copy bit for bit, the per-key sums the same terms in twice as many rows
    for bf in (False, True):      # (f32 dx, then the bf16 residual-gradient stream's form)
        dt = torch.bfloat16 if bf else torch.float32
        full = ops.heads_loss_fused(xd, Wd, bd, descr, nvalid, B, S, want_logits=want_logits, drop=drop, dx_dtype=dt, half_tiles=False)
        half = ops.heads_loss_fused(xd, Wd, bd, descr, nvalid, B, S, want_logits=want_logits, drop=drop, dx_dtype=dt, half_tiles=True)
        assert half[0].shape[0] == (T + 63) // 64 and full[0].shape[0] == (T + 127) // 128
        for a_, b_, what in zip(full[1:], half[1:], ("dlogits", "logits", "dx", "dx_drop")):
            assert (a_ is None) == (b_ is None), what
            if a_ is not None:
                assert torch.equal(a_, b_), what
        sh = torch.zeros(len(descr) * 3, device=DEV)
        ops.reduce_partials(half[0], sh, 3 * len(descr))
        sh = sh.view(len(descr), 3).cpu().double()
        for i, k in enumerate(lay.head_order):
            assert abs(sh[i, 0] - sums[i, 0]) <= 1e-5 * max(1.0, abs(sums[i, 0])) and abs(sh[i, 1] - sums[i, 1]) <= 1e-5 * max(1.0, abs(sums[i, 1])), k
            assert sh[i, 2] == sums[i, 2], k


@pytest.mark.parametrize("B,S", [(1, 128), (5, 128), (2, 64), (6, 64)])
def test_attn_block_bwd(B, S):
    """mfp_attn_block_bwd: da = d_o1 Wo, dqkv = MHSA'(...; da), dy1 = dqkv Wqkv in ONE launch (autodiff of
    transformer.py:216-221,60-99; documents of 128 positions, or two documents of 64 per tile) against the three launches it
    replaces (mfp_dgrad_d256, mfp_attention_bwd, mfp_dgrad_qkv) and a double reference built from the same bf16 inputs;
    ragged key masks."""
    ops = _ops()
    D, H = 256, 8
    T = B * S
    g = torch.Generator().manual_seed(700 + B)
    rn = lambda *s: torch.randn(*s, generator=g)
    bf = torch.bfloat16
    d = lambda t, dt=None: t.to(DEV, dt) if dt else t.to(DEV)
    qkv = d(rn(T, 3 * D) * 0.7, bf)
    nvalid = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32)
    nvalid[0] = S
    nvd = d(nvalid)
    a, lse = ops.attention_fwd(qkv, nvd, B, S, H)
    Wo = bf16_round(rn(D, D) * 0.06)
    Wqkv = bf16_round(rn(3 * D, D) * 0.08)
    d_o1 = d(rn(T, D) * 0.5, bf)
    Wot, Wqt = d(Wo.t().contiguous(), bf), d(Wqkv.t().contiguous(), bf)
    dqkv, dy1 = ops.attn_block_bwd(d_o1, Wot, qkv, a, lse, nvd, Wqt, B, S, H)
    # the three launches
    da_u = ops.dgrad_d256(d_o1, Wot)
    dqkv_u = ops.attention_bwd(qkv, nvd, a, da_u, lse, B, S, H)
    dy1_u = ops.dgrad_qkv(dqkv_u, Wqt)
    assert_close(dqkv, dqkv_u.float().cpu().double(), 2e-2, 2e-2, "dqkv vs the three launches")
    assert_close(dy1, dy1_u.float().cpu().double(), 2e-2, 2e-2, "dy1 vs the three launches")
    assert (dqkv != dqkv_u).float().mean().item() < 0.15          # same operands; only the summation order of dQ differs
    # double reference: da rounded to bf16 as both paths hold it, softmax from the saved lse
    q64 = qkv.float().cpu().double().view(B, S, 3, H, 32)
    da64 = bf16_round((d_o1.float().cpu() @ Wo).float()).double().view(B, S, H, 32)
    a64 = a.float().cpu().double().view(B, S, H, 32)
    sc = torch.einsum("bqhd,bkhd->bhqk", q64[:, :, 0], q64[:, :, 1]) / 32 ** 0.5
    sc = sc + ((torch.arange(S)[None, :] >= nvalid[:, None]).double() * -1e9)[:, None, None, :]
    P = torch.exp(sc - lse.cpu().double()[..., None])
    dP = torch.einsum("bqhd,bkhd->bhqk", da64, q64[:, :, 2])
    delta = (da64 * a64).sum(-1).permute(0, 2, 1)                  # [B,H,S]
    dS = P * (dP - delta[..., None]) / 32 ** 0.5
    want = torch.stack([torch.einsum("bhqk,bkhd->bqhd", dS, q64[:, :, 1]), torch.einsum("bhqk,bqhd->bkhd", dS, q64[:, :, 0]),
                        torch.einsum("bhqk,bqhd->bkhd", P, da64)], dim=2).reshape(T, 3 * D)
    assert_close(dqkv, want, 3e-2, 3e-2, "dqkv vs double")
    want_dy1 = dqkv.float().cpu().double() @ Wqkv.double()
    assert_close(dy1, want_dy1, 1e-2, 1e-2, "dy1 vs double (from the kernel's own dqkv)")


@pytest.mark.parametrize("B,S,drop", [(5, 128, True), (6, 64, True), (256, 128, True), (4, 128, False)])
def test_attn_block_bwd_ln(B, S, drop):
    """mfp_attn_block_bwd_ln (the backward of LN1 in the epilogue of the attention half's input-gradient launch) against the two
    launches it replaces, mfp_attn_block_bwd + mfp_layernorm_bwd_res16: dqkv bit-identical (same code), dx / the dropout-masked
    copy equal up to an occasional bf16 step, parameter-gradient sums to summation-order noise; with and without the masked copy
    (block 0 has no consumer for one)."""
    ops = _ops()
    D, H = 256, 8
    T = B * S
    g = torch.Generator().manual_seed(900 + B)
    rn = lambda *s: torch.randn(*s, generator=g)
    bf = torch.bfloat16
    qkv = (rn(T, 3 * D) * 0.7).to(DEV, bf)
    nvalid = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32).to(DEV)
    a, lse = ops.attention_fwd(qkv, nvalid, B, S, H)
    Wot, Wqt = (rn(D, D) * 0.06).to(DEV, bf), (rn(D, 3 * D) * 0.08).to(DEV, bf)
    d_o1 = (rn(T, D) * 0.5).to(DEV, bf)
    x = (rn(T, D) * 1.5 + 0.2).to(DEV)
    gamma, beta = (torch.rand(D, generator=g) + 0.5).to(DEV), rn(D).to(DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, bf)
    dres = rn(T, D).to(DEV, bf)
    step = torch.full((1,), 2, dtype=torch.int32, device=DEV)
    new = lambda: (torch.full((D,), float("nan"), device=DEV), torch.full((D,), float("nan"), device=DEV),
                   torch.full((D,), float("nan"), device=DEV))
    dg0, db0, cs0 = new()
    dqkv0, dy1 = ops.attn_block_bwd(d_o1, Wot, qkv, a, lse, nvalid, Wqt, B, S, H)
    if drop:
        dx0, do0 = ops.layernorm_bwd(dy1, x, gamma, mean, rstd, dres, dg0, db0, drop=(cs0, 0.1, 13, 4, step))
    else:
        dx0, do0 = ops.layernorm_bwd(dy1, x, gamma, mean, rstd, dres, dg0, db0), None
    for batched in (False, True):
        dg1, db1, cs1 = new()
        jobs = [] if batched else None
        dqkv1, dx1, do1 = ops.attn_block_bwd_ln(d_o1, Wot, qkv, a, lse, nvalid, Wqt, B, S, H, x, gamma, mean, rstd, dres, dg1, db1,
                                                drop=(cs1, 0.1, 13, 4, step) if drop else None, jobs=jobs)
        if batched:
            ops.reduce_partials_batch(jobs)
        assert torch.equal(dqkv1, dqkv0)
        assert (dx1 != dx0).float().mean().item() < 1e-4
        assert_close(dx1, dx0.float().cpu().double(), 2e-2, 1e-2, "dx")
        sums = [(dg1, dg0, "dgamma"), (db1, db0, "dbeta")]
        if drop:
            assert (do1 != do0).float().mean().item() < 1e-4 and torch.equal(do1 == 0, do0 == 0)
            sums.append((cs1, cs0, "colsum"))
        else:
            assert do1 is None
        for u, v, what in sums:
            assert torch.isfinite(u).all(), what
            assert_close(u, v.cpu().double(), 2e-3 * float(v.abs().max()), 1e-4, what)
    # x-hat form
    xh = ((x - mean[:, None]) * rstd[:, None]).to(bf)
    dg2, db2, cs2 = new()
    dqkv2, dx2, do2 = ops.attn_block_bwd_ln(d_o1, Wot, qkv, a, lse, nvalid, Wqt, B, S, H, None, gamma, None, rstd, dres, dg2, db2,
                                            drop=(cs2, 0.1, 13, 4, step) if drop else None, xhat=xh)
    assert torch.equal(dqkv2, dqkv0)
    _check_ln_from_xhat(dy1, xh, gamma, rstd, dres, dx2, do2, dg2, db2, cs2, 0.1)


@pytest.mark.parametrize("T", [4096, 32768])
def test_wgrad_group_xhat_operand(T):
    """mfp_wgrad_job::n_affine: the B operand of a weight-gradient job is x-hat = (x - mean) rstd of a LayerNorm instead of
    its output y = x-hat gamma + beta; mfp_wgrad_reduce writes gamma[n] (A^T x-hat)[m][n] + beta[n] colsum[m].  Block-shaped
    group (both x-hat jobs + two plain ones, 128 x 128 tiles at 4 096 tokens, macro tiles at 32 768) against A^T y in double
    with y NOT rounded to bf16; the plain jobs of the same launch are untouched."""
    ops = _ops()
    D = 256
    g = torch.Generator().manual_seed(T + 77)
    rn = lambda *s: torch.randn(*s, generator=g)
    bf = torch.bfloat16
    dqkv, dh, d_o2, d_o1 = (rn(T, 3 * D) * 0.3).to(DEV, bf), (rn(T, 2 * D) * 0.3).to(DEV, bf), (rn(T, D) * 0.3).to(DEV, bf), (rn(T, D) * 0.3).to(DEV, bf)
    xh1, xh2 = rn(T, D).to(DEV, bf), rn(T, D).to(DEV, bf)
    h, a = rn(T, 2 * D).clamp(min=0).to(DEV, bf), rn(T, D).to(DEV, bf)
    ga1 = torch.cat([1.0 + 0.3 * rn(D), 0.2 * rn(D)]).to(DEV)      # gamma | beta
    ga2 = torch.cat([1.0 + 0.3 * rn(D), 0.2 * rn(D)]).to(DEV)
    nan = lambda *s: torch.full(s, float("nan"), device=DEV)
    Wq, W1, W2, Wo = nan(3 * D, D), nan(2 * D, D), nan(D, 2 * D), nan(D, D)
    bq, b1 = nan(3 * D), nan(2 * D)
    pending = []
    ops.wgrad_group([dict(A=dqkv, B=xh1, out=Wq, M=3 * D, N=D, colsum=bq, naffine=ga1),
                     dict(A=dh, B=xh2, out=W1, M=2 * D, N=D, colsum=b1, naffine=ga2),
                     dict(A=d_o2, B=h, out=W2, M=D, N=2 * D),
                     dict(A=d_o1, B=a, out=Wo, M=D, N=D)], T, defer=pending)
    ops.wgrad_reduce(pending)
    d64 = lambda t: t.double().cpu()
    y1 = d64(xh1) * d64(ga1[:D]) + d64(ga1[D:])
    y2 = d64(xh2) * d64(ga2[:D]) + d64(ga2[D:])
    for got, want, what in ((Wq, d64(dqkv).t() @ y1, "dWqkv"), (W1, d64(dh).t() @ y2, "dW1"), (W2, d64(d_o2).t() @ d64(h), "dW2"),
                            (Wo, d64(d_o1).t() @ d64(a), "dWo"), (bq, d64(dqkv).sum(0), "bqkv"), (b1, d64(dh).sum(0), "b1")):
        assert torch.isfinite(got).all(), what
        assert_close(got, want, 2e-4 * float(want.abs().max()), 1e-4, what)


@pytest.mark.parametrize("B,p", [(3, 0.0), (4, 0.1), (9, 0.1)])      # (9: one full group of 16 half-document workgroups + a pair)
def test_block_fwd(B, p):
    """mfp_block_fwd: a whole DeepSVG block forward in ONE launch against mfp_attn_block_fwd + mfp_mlp_fused_fwd (same
    dropout streams): the attention half bit for bit, the MLP half within bf16 rounding (LN2 statistics are summed in
    another order), plus a double reference of the MLP half from the kernel's own x1."""
    ops = _ops()
    S, D, H = 128, 256, 8
    T = B * S
    g = torch.Generator().manual_seed(300 + B)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(T, D) * (1.0 + torch.rand(T, 1, generator=g))
    g1, b1_, g2, b2_ = 1.0 + 0.2 * rn(D), 0.1 * rn(D), 1.0 + 0.2 * rn(D), 0.1 * rn(D)
    Wqkv, bqkv, Wo, bo = bf16_round(rn(3 * D, D) * 0.08), rn(3 * D) * 0.1, bf16_round(rn(D, D) * 0.06), rn(D) * 0.1
    W1, c1, W2, c2 = bf16_round(rn(2 * D, D) * 0.06), rn(2 * D) * 0.1, bf16_round(rn(D, 2 * D) * 0.05), rn(D) * 0.1
    nvalid = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32)
    step = torch.full((1,), 1, dtype=torch.int32, device=DEV)
    d = lambda t, dt=None: t.to(DEV, dt) if dt else t.to(DEV)
    bf = torch.bfloat16
    args = (d(x), d(g1), d(b1_), d(Wqkv, bf), d(bqkv), d(Wo, bf), d(bo), d(nvalid))
    x2c = torch.empty(T, D, dtype=bf, device=DEV)
    x2, (y1, m1, r1, qkv, a, lse, x1, y2, m2, r2, h) = ops.block_fwd(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H,
                                                                       p, 7, 3, 4, step, x2_c=x2c)
    x1u, y1u, m1u, r1u, qkvu, au, lseu = ops.attn_block_fwd(*args, B, S, H, (p, 7, 3), step)
    for got, want in ((y1, y1u), (qkv, qkvu), (a, au)):
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(x1, x1u) and torch.equal(lse, lseu) and torch.equal(m1, m1u) and torch.equal(r1, r1u)
    x2u, y2u, m2u, r2u, hu = ops.mlp_fused_fwd(x1u, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), (p, 7, 4), step)
    assert_close(m2, m2u.cpu().double(), 1e-5, 1e-5, "mean2")
    assert_close(r2, r2u.cpu().double(), 1e-5, 1e-5, "rstd2")
    assert (y2 != y2u).float().mean().item() < 0.01 and (y2.float() - y2u.float()).abs().max().item() <= 0.04
    assert_close(h, hu.float().cpu().double(), 3e-2, 2e-2, "h vs mlp_fused")
    assert_close(x2, x2u.cpu().double(), 3e-2, 2e-2, "x2 vs mlp_fused")
    assert torch.equal(x2c.view(torch.int16), x2.to(bf).view(torch.int16))
    if p == 0.0:
        y2d = y2.float().cpu().double()
        hd = torch.relu(y2d @ W1.double().t() + c1.double())
        assert_close(h, hd, 2e-2, 1e-2, "h vs double")
        want = x1.cpu().double() + h.float().cpu().double() @ W2.double().t() + c2.double()
        assert_close(x2, want, 5e-3, 2e-3, "x2 vs double")
        # mfp_block_infer: the same launch with nothing saved (inference callers) -- x2 bit for bit
        x2i = ops.block_infer(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H)
        assert torch.equal(x2i, x2)
    # mfp_block_fwd_xhat: the y1 / y2 buffers receive x-hat = (x - mean) rstd; everything else bit for bit
    x2x, (xh1, m1x, r1x, qkvx, ax, lsex, x1x, xh2, m2x, r2x, hx) = ops.block_fwd(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2),
                                                                              B, S, H, p, 7, 3, 4, step, xhat_stash=True)
    for got, want in ((x2x, x2), (qkvx, qkv), (ax, a), (lsex, lse), (x1x, x1), (hx, h), (m1x, m1), (r1x, r1), (m2x, m2), (r2x, r2)):
        assert torch.equal(got, want)
    for xh, xin, m, r, what in ((xh1, d(x), m1, r1, "x-hat 1"), (xh2, x1, m2, r2, "x-hat 2")):
        want = ((xin.double() - m.double()[:, None]) * r.double()[:, None]).cpu()
        assert_close(xh, want, 1e-2, 8e-3, what)
        # (bf16 roundings of f32 values that may differ in the last bit)
        assert (xh != want.to(DEV, torch.float32).to(bf)).float().mean().item() < 2e-3, what
    # mfp_block_fwd_xhat_half: two four-wave workgroups per document (the other half's K / V recomputed) -- every output of the
    # x-hat form bit for bit (the same instruction sequence on the same operands produces every value)
    # (4 waves: two row tiles per wave, one wave per SIMD; 8 waves: one row tile per wave, one (query tile, head) per wave)
    for waves in (4, 8):
        x2c_h = torch.zeros(T, D, dtype=bf, device=DEV)
        x2h, saved_h = ops.block_fwd(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H, p, 7, 3, 4, step, x2_c=x2c_h,
                                     xhat_stash=True, half_tiles=waves)
        names = ("xhat1", "mean1", "rstd1", "qkv", "a", "lse", "x1", "xhat2", "mean2", "rstd2", "h")
        assert torch.equal(x2h, x2x), "half tiles (%d waves): x2" % waves
        assert torch.equal(x2c_h.view(torch.int16), x2c.view(torch.int16)), "half tiles (%d waves): x2 bf16" % waves
        for name, got, want in zip(names, saved_h, (xh1, m1x, r1x, qkvx, ax, lsex, x1x, xh2, m2x, r2x, hx)):
            assert torch.equal(got, want), "half tiles (%d waves): %s" % (waves, name)


# ------------------------------------------------------------------------------------ d_model 512 (csrc/block_d512.hip)
@pytest.mark.parametrize("T,N,relu", [(256, 1536, False), (1000, 1024, True), (16384, 1536, False), (384, 128, True)])
def test_ln_dense_d512(T, N, relu):
    """mfp_ln_dense_d512 (LN1 + Q|K|V, LN2 + FFN1 + ReLU of a d_model-512 block: transformer.py:216-217,222-223,161-166) against
    a double reference: LN statistics and y, then the product FROM THE KERNEL'S OWN bf16 y (so the bound is the product's)."""
    ops = _ops()
    D = 512
    g = torch.Generator().manual_seed(500 + N + T)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(T, D) * (0.5 + torch.rand(T, 1, generator=g)) + 0.3 * rn(T, 1)
    gam, bet = 1.0 + 0.2 * rn(D), 0.1 * rn(D)
    W, b = bf16_round(rn(N, D) * 0.05), rn(N) * 0.1
    out, y, mean, rstd = ops.ln_dense_d512(x.to(DEV), gam.to(DEV), bet.to(DEV), W.to(DEV, torch.bfloat16), b.to(DEV), N, relu=relu)
    xd = x.double()
    mu = xd.mean(1)
    var = ((xd - mu[:, None]) ** 2).mean(1)
    rs = 1.0 / torch.sqrt(var + 1e-3)
    assert_close(mean, mu, 1e-5, 1e-5, "mean")
    assert_close(rstd, rs, 1e-5, 1e-4, "rstd")
    yw = (xd - mu[:, None]) * rs[:, None] * gam.double() + bet.double()
    assert_close(y, yw, 2e-2, 8e-3, "y = LN(x)")
    want = y.float().cpu().double() @ W.double().t() + b.double()
    if relu:
        want = torch.relu(want)
    assert_close(out, want, 2e-2, 8e-3, "out vs double (from the kernel's own y)")
    # mfp_ln_dense_d512_xhat: the y buffer receives x-hat = (x - mean) rstd; the product and the statistics bit for bit
    out2, xh, mean2, rstd2 = ops.ln_dense_d512(x.to(DEV), gam.to(DEV), bet.to(DEV), W.to(DEV, torch.bfloat16), b.to(DEV), N, relu=relu,
                                               xhat_stash=True)
    assert torch.equal(out2, out) and torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    assert_close(xh, (xd - mu[:, None]) * rs[:, None], 1e-2, 8e-3, "x-hat")


@pytest.mark.parametrize("T,N", [(256, 1024), (1000, 1024), (16384, 1024), (128, 256)])
def test_dense_relumask_d512(T, N):
    """mfp_dense_relumask_d512: dh = (d_o2 W2) * [h > 0] (autodiff of transformer.py:161-171) against a double reference."""
    ops = _ops()
    g = torch.Generator().manual_seed(600 + T)
    rn = lambda *s: torch.randn(*s, generator=g)
    A, W = bf16_round(rn(T, 512)), bf16_round(rn(N, 512) * 0.05)
    h = bf16_round(torch.relu(rn(T, N)))
    out = ops.dense_relumask_d512(A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16), h.to(DEV, torch.bfloat16))
    want = (A.double() @ W.double().t()) * (h > 0).double()
    assert_close(out, want, 2e-2, 8e-3, "dh vs double")
    assert torch.equal(out.cpu() == 0, (h == 0) | (out.cpu() == 0)) and bool((out.cpu()[h == 0] == 0).all())


@pytest.mark.parametrize("T,K,p", [(256, 512, 0.0), (1000, 1024, 0.1), (16384, 1024, 0.1), (64, 1536, 0.0)])
def test_dense_n512_res(T, K, p):
    """mfp_dense_n512_res (attention output projection / FFN2 of a d_model-512 block with dropout and residual:
    transformer.py:218-221,224-225) against a double reference; the dropout mask is mfp_gemm's for the same stream (the
    backward kernels regenerate it from (seed, offset, step, row, column))."""
    ops = _ops()
    g = torch.Generator().manual_seed(700 + T + K)
    rn = lambda *s: torch.randn(*s, generator=g)
    A, W, b, res = bf16_round(rn(T, K)), bf16_round(rn(512, K) * 0.04), rn(512) * 0.1, rn(T, 512)
    step = torch.full((1,), 3, dtype=torch.int32, device=DEV)
    oc = torch.empty(T, 512, dtype=torch.bfloat16, device=DEV)
    Ad, Wd, bd, rd = A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16), b.to(DEV), res.to(DEV)
    out = ops.dense_n512_res(Ad, Wd, bd, rd, (p, 11, 5), step, out_bf16=oc)
    lin = A.double() @ W.double().t() + b.double()
    if p == 0.0:
        assert_close(out, res.double() + lin, 2e-2, 3e-3, "out vs double")
    else:
        ref = ops.gemm(Ad, Wd, T, 512, K, a_kmajor=True, b_kmajor=True, bias=bd, residual=rd, dropout=(p, 11, 5), step_ptr=step,
                       out_dtype=torch.float32)
        keep = ((ref.cpu().double() - res.double()).abs() > 1e-12)      # mfp_gemm's mask for the same stream
        frac = keep.double().mean().item()
        assert abs(frac - (1 - p)) < 0.01, frac
        assert_close(out, res.double() + keep.double() * lin / (1 - p), 3e-2, 3e-3, "out vs double under mfp_gemm's mask")
    assert torch.equal(oc.view(torch.int16), out.to(torch.bfloat16).view(torch.int16))
    # the copy is optional
    out2 = ops.dense_n512_res(Ad, Wd, bd, rd, (p, 11, 5), step)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("T,K", [(256, 512), (1000, 1024), (16384, 1536), (64, 1024)])
def test_dense_n512(T, K):
    """mfp_dense_n512: the input gradients da = d_o1 Wo, dy2 = dh W1, dy1 = dqkv Wqkv of a d_model-512 block (autodiff of
    transformer.py:85-99,161-171) on the transposed weight shadows, against a double reference."""
    ops = _ops()
    g = torch.Generator().manual_seed(800 + T + K)
    rn = lambda *s: torch.randn(*s, generator=g)
    A, W = bf16_round(rn(T, K)), bf16_round(rn(512, K) * 0.04)
    out = ops.dense_n512(A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16))
    assert_close(out, A.double() @ W.double().t(), 2e-2, 8e-3, "out vs double")


@pytest.mark.parametrize("reserve", [0, 8])
@pytest.mark.parametrize("T,K,p", [(1024, 1024, 0.1), (384, 1536, 0.0), (16384, 1536, 0.1), (2048, 1024, None),
                                   (128, 1024, 0.1), (896, 1536, 0.1), (1152, 1024, 0.0)])
def test_dense_n512_lnb(T, K, p, reserve):
    """(round 6, ADVICE r05: T / 128 = 1, 7 and 9 row tiles -- grids that are not multiples of 16, where the two workgroups of a
    row tile are neighbours in dispatch order and sit on different XCDs -- and every shape again with 8 CUs reserved for
    RCCL (MFP_DP_RESERVE_CUS): the flag exchange must not depend on the partner's placement.)"""
    ops = _ops()
    ops.set_reserved_cus(reserve)
    try:
        _dense_n512_lnb_body(T, K, p)
    finally:
        ops.set_reserved_cus(0)


def _dense_n512_lnb_body(T, K, p):
    """mfp_dense_n512_lnb: dy = A W^T (the gradient of a LayerNorm output at d_model 512: dy2 = dh W1, dy1 = dqkv Wqkv) with the
    x-hat LayerNorm backward on the f32 result in the same launch -- the two workgroups of a row tile exchange their halves of the
    row sums through global memory.  Against a double restatement from the double product (dy is never rounded to bf16 here), and
    against the launch pair it replaces (mfp_dense_n512 + mfp_layernorm_bwd_xhat, whose dy IS bf16) within that rounding; three
    launches back to back on the same flags (they must be zero again after every launch); T = 384: three row tiles, i.e. the
    unpaired workgroup mapping; p = None: no masked copy (block 0)."""
    ops = _ops()
    D = 512
    g = torch.Generator().manual_seed(820 + T + K)
    rn = lambda *s: torch.randn(*s, generator=g)
    A, W = bf16_round(rn(T, K)), bf16_round(rn(D, K) * 0.04)
    xh = bf16_round(rn(T, D))
    gamma, rstd = (torch.rand(D, generator=g) + 0.5), (torch.rand(T, generator=g) + 0.5)
    dres = bf16_round(rn(T, D))
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    d = lambda t, dt=None: t.to(DEV, dt) if dt else t.to(DEV)
    bf = torch.bfloat16
    Ad, Wd, xhd, gd, rd, drd = d(A, bf), d(W, bf), d(xh, bf), d(gamma), d(rstd), d(dres, bf)
    outs = []
    for rep in range(3):
        dg, db, cs = (torch.full((D,), 7.0, device=DEV) for _ in range(3))
        drop = (cs, p, 7, 3, step) if p is not None else None
        r = ops.dense_n512_lnb(Ad, Wd, xhd, gd, rd, drd, dg, db, drop=drop)
        dx, dd = r if p is not None else (r, None)
        outs.append((dx.clone(), None if dd is None else dd.clone(), dg.clone(), db.clone(), cs.clone()))
    torch.cuda.synchronize()
    _, flags = ops._LNB_WS[(torch.device(DEV).index if torch.device(DEV).index is not None else torch.cuda.current_device(), T,
                            int(torch.cuda.current_stream().cuda_stream))]
    assert int(flags.abs().sum().item()) == 0
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert (a is None and b is None) or torch.equal(a, b)
    dx, dd, dg, db, cs = outs[0]
    dy64 = A.double() @ W.double().t()
    _check_ln_from_xhat(dy64, xh, gamma, rstd, dres, dx, dd, dg, db, cs, p if p is not None else 0.0)
    # the launch pair it replaces
    dy = ops.dense_n512(Ad, Wd)
    dg2, db2, cs2 = (torch.empty(D, device=DEV) for _ in range(3))
    r2 = ops.layernorm_bwd(dy, None, gd, None, rd, drd, dg2, db2, drop=(cs2, p, 7, 3, step) if p is not None else None, xhat=xhd)
    dx2, dd2 = r2 if p is not None else (r2, None)
    assert_close(dx, dx2.double().cpu(), 6e-2, 2e-2, "dx vs dense_n512 + layernorm_bwd_xhat")
    if dd is not None:
        assert torch.equal(dd == 0, dd2 == 0) or ((dd == 0) != (dd2 == 0)).float().mean().item() < 1e-4      # (the same dropout mask)
    assert_close(dg, dg2.double().cpu(), 2e-2 * float(dg2.abs().max()), 1e-3, "dgamma vs the pair")
    assert_close(db, db2.double().cpu(), 2e-2 * float(db2.abs().max()), 1e-3, "dbeta vs the pair")


@pytest.mark.parametrize("T,U", [(384, 1384), (16384, 1384), (200, 1400)])
def test_dense_n512_lda(T, U):
    """mfp_dense_n512_lda: the decoder heads' input gradient at d_model 512 -- A = d(logits) [T][U] with NO padding columns, W = the
    transposed heads [512][U rounded up to 128] whose pad columns are zero (decoder.py:39-43 under autodiff) -- against a double
    reference; the k-range read past a row's end (the head of the next row, zeros behind the last) must not leak."""
    ops = _ops()
    K = (U + 127) // 128 * 128
    g = torch.Generator().manual_seed(810 + T + U)
    rn = lambda *s: torch.randn(*s, generator=g)
    A = bf16_round(rn(T, U) * 3.0)                                  # (large next-row values: a leak would show)
    W = torch.zeros(512, K)
    W[:, :U] = bf16_round(rn(512, U) * 0.04)
    out = ops.dense_n512_lda(A.to(DEV, torch.bfloat16), W.to(DEV, torch.bfloat16))
    assert_close(out, A.double() @ W[:, :U].double().t(), 6e-2, 8e-3, "out vs double")


@pytest.mark.parametrize("B,p", [(2, 0.0), (6, 0.1)])
def test_block_fwd_two_documents_per_tile(B, p):
    """mfp_block_fwd / mfp_block_infer at S = 64 (two documents per 128-row tile: the shape of real Crello / RICO batches,
    whose sequences are at most 51 positions long -- data/crello-spec.yml:6-13) against the separate launches on the same
    dropout streams: LN1 + Q|K|V (mfp_qkv_fused_fwd), attention per document (mfp_attention_fwd), output projection
    (mfp_gemm), MLP half (mfp_mlp_fused_fwd); ragged lengths, so both documents of a tile carry their own key mask."""
    ops = _ops()
    S, D, H = 64, 256, 8
    T = B * S
    g = torch.Generator().manual_seed(900 + B)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(T, D) * (1.0 + torch.rand(T, 1, generator=g))
    g1, b1_, g2, b2_ = 1.0 + 0.2 * rn(D), 0.1 * rn(D), 1.0 + 0.2 * rn(D), 0.1 * rn(D)
    Wqkv, bqkv, Wo, bo = bf16_round(rn(3 * D, D) * 0.08), rn(3 * D) * 0.1, bf16_round(rn(D, D) * 0.06), rn(D) * 0.1
    W1, c1, W2, c2 = bf16_round(rn(2 * D, D) * 0.06), rn(2 * D) * 0.1, bf16_round(rn(D, 2 * D) * 0.05), rn(D) * 0.1
    nvalid = torch.randint(1, 52, (B,), generator=g).to(torch.int32)
    nvalid[0] = S
    step = torch.full((1,), 1, dtype=torch.int32, device=DEV)
    d = lambda t, dt=None: t.to(DEV, dt) if dt else t.to(DEV)
    bf = torch.bfloat16
    args = (d(x), d(g1), d(b1_), d(Wqkv, bf), d(bqkv), d(Wo, bf), d(bo), d(nvalid))
    x2, (y1, m1, r1, qkv, a, lse, x1, y2, m2, r2, h) = ops.block_fwd(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H,
                                                                       p, 7, 3, 4, step)
    qkvu, y1u, m1u, r1u = ops.qkv_fused_fwd(d(x), d(g1), d(b1_), d(Wqkv, bf), d(bqkv))
    for got, want in ((y1, y1u), (qkv, qkvu)):
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(m1, m1u) and torch.equal(r1, r1u)
    au, lseu = ops.attention_fwd(qkvu, d(nvalid), B, S, H)
    assert_close(a, au.float().cpu().double(), 1e-2, 1e-2, "attention output vs mfp_attention_fwd")
    assert_close(lse, lseu.cpu().double(), 2e-3, 1e-3, "lse vs mfp_attention_fwd")
    x1u = ops.gemm(a, d(Wo, bf), T, D, D, a_kmajor=True, b_kmajor=True, bias=d(bo), residual=d(x), dropout=(p, 7, 3), step_ptr=step,
                   out_dtype=torch.float32)
    assert_close(x1, x1u.cpu().double(), 2e-3, 1e-3, "x1 vs mfp_gemm on the kernel's own a")
    x2u, y2u, m2u, r2u, hu = ops.mlp_fused_fwd(x1, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), (p, 7, 4), step)
    assert_close(h, hu.float().cpu().double(), 3e-2, 2e-2, "h vs mlp_fused")
    assert_close(x2, x2u.cpu().double(), 3e-2, 2e-2, "x2 vs mlp_fused")
    # a double reference of the attention from the kernel's own qkv: every query sees only ITS document's valid keys
    q64 = qkv.float().cpu().double().view(B, S, 3, H, 32)
    sc = torch.einsum("bqhd,bkhd->bhqk", q64[:, :, 0], q64[:, :, 1]) / 32 ** 0.5
    sc = sc + ((torch.arange(S)[None, :] >= nvalid[:, None]).double() * -1e9)[:, None, None, :]
    want_a = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(sc, -1), q64[:, :, 2]).reshape(T, D)
    assert_close(a, want_a, 1e-2, 1e-2, "attention output vs double")
    if p == 0.0:
        x2i = ops.block_infer(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H)
        assert torch.equal(x2i, x2)
    # x-hat form, and the same on HALF tiles -- at S = 64 a half tile is ONE document (eight waves, one row tile each; nothing
    # recomputed): every output bit for bit
    x2x, saved_x = ops.block_fwd(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H, p, 7, 3, 4, step, xhat_stash=True)
    x2h, saved_h = ops.block_fwd(*args, d(g2), d(b2_), d(W1, bf), d(c1), d(W2, bf), d(c2), B, S, H, p, 7, 3, 4, step, xhat_stash=True,
                                 half_tiles=True)
    assert torch.equal(x2x, x2) and torch.equal(x2h, x2)
    for name, got, want in zip(("xhat1", "mean1", "rstd1", "qkv", "a", "lse", "x1", "xhat2", "mean2", "rstd2", "h"), saved_h, saved_x):
        assert torch.equal(got, want), "half tiles at S = 64: " + name
