"""Size-independent properties of the hot path at BASELINE.json's full size (config c2: Crello,
256 documents x 128 elements, d_model 256, 4 blocks) -- the oracle cannot run there in seconds, so
parity is anchored on what the reference's model guarantees by construction:

* documents are independent (attention within a document, LayerNorm per element): a batch equals
  its two halves -- logits bit for bit, score / denominator sums exactly, gradients by linearity;
* padding is inert: what sits in positions >= length changes nothing at the valid positions;
* the input is a SET (input_dtype="set", no position embedding): permuting the elements of a
  document permutes its logits and leaves the losses where they were;
* the loss denominators are counts of (mask & valid & condition): integers, recomputed on the host;
* the fused masking kernel at this size: masks only valid positions, leaves unmasked inputs
  untouched, masks 15 % of the valid fields (binomial 5-sigma band).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, S, D, L = 256, 128, 256, 4


def _model(dtype):
    from mfp.data.spec import make_input_columns
    from mfp.models.model import Model
    ic = make_input_columns("crello")
    model = Model(ic, num_blocks=L, latent_dim=D, dropout=0.0, l2=1e-2, dtype=dtype, device=DEV, seed=3)
    return ic, model


def _masked_batch(ic, seed=0, ragged=True, B=B):
    """Crello-shaped batch + (modified inputs, masks) from the reference-shaped torch masking."""
    from mfp.data.spec import synthetic_batch
    from mfp.models import masking
    from mfp.models.architecture.mask import get_seq_mask
    batch = synthetic_batch(ic, B, S, seed=seed, ragged=ragged)
    nd = {k: v for k, v in ic.items() if not v.get("demo_only")}
    gen = torch.Generator().manual_seed(seed)
    seq_mask = get_seq_mask(batch["length"], maxlen=S)
    filtered = masking.filter_padding(batch, nd, seq_mask)
    modified, masks = {}, {}
    for k, c in nd.items():
        if not c["is_sequence"]:
            modified[k] = filtered[k]
            continue
        m = seq_mask & (torch.rand(B, S, generator=gen) < 0.15)
        modified[k], masks[k] = masking.apply_token(filtered[k], c, m, "masked"), m
    modified["length"] = batch["length"]
    return batch, modified, masks, seq_mask


def _dev(d):
    return {k: v.to(DEV) for k, v in d.items()}


def _take(d, sl):
    return {k: v[sl] for k, v in d.items()}


def _loss_grads(model, ic, batch, modified, masks):
    from mfp.models.metrics import build_loss_keys
    model.store.g.zero_()
    keys = build_loss_keys(ic, model.layout.head_cols, _dev(batch), _dev(masks))
    loss, sums, outputs = model.forward_loss(_dev(modified), keys, training=True)
    loss.backward()
    for side in model.side_streams:
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return sums.clone(), outputs["_flat_logits"].clone(), model.store.g.clone()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_batch_equals_its_halves(dtype):
    ic, model = _model(dtype)
    batch, modified, masks, _ = _masked_batch(ic, seed=1)
    sums, logits, g = _loss_grads(model, ic, batch, modified, masks)
    h = B // 2
    parts = [_loss_grads(model, ic, _take(batch, sl), _take(modified, sl), _take(masks, sl))
             for sl in (slice(0, h), slice(h, B))]
    # forward: a document's logits do not depend on which batch it sits in
    assert torch.equal(logits, torch.cat([p[1] for p in parts]))
    # score numerators / denominators are plain sums over documents
    assert torch.equal(sums[:, 2], parts[0][0][:, 2] + parts[1][0][:, 2])
    assert torch.allclose(sums[:, 1], parts[0][0][:, 1] + parts[1][0][:, 1], rtol=1e-6, atol=1e-3)
    # losses are batch means (metrics.py:277): mean(full) = (mean(h1) + mean(h2)) / 2; so are the gradients
    assert torch.allclose(sums[:, 0], 0.5 * (parts[0][0][:, 0] + parts[1][0][:, 0]), rtol=2e-5, atol=1e-5)
    want = 0.5 * (parts[0][2] + parts[1][2])
    err = (g - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= (2e-4 if dtype == "fp32" else 2e-2) * scale, (err, scale)


def test_c4_per_gpu_shape_equals_its_halves_and_the_c2_batch():
    """BASELINE config c4 (batch 1024 over 8 GPUs): the per-GPU shard, 128 documents x 128 elements, on the bf16 path.
    T / 128 = 128 workgroups < 256 CUs: the launches that have a half-size form take it (mfp_dgrad_qkv / _d256).  The
    shard's logits equal, bit for bit, those of the same documents inside a c2 batch of 256 (other workgroup sizes,
    other grid) and inside shards of 64; its gradients are the mean of its halves' (linearity of the batch mean)."""
    ic, model = _model("bf16")
    batch, modified, masks, _ = _masked_batch(ic, seed=4)
    _, logits_full, _ = _loss_grads(model, ic, batch, modified, masks)
    n = B // 2
    sl = slice(0, n)
    sums, logits, g = _loss_grads(model, ic, _take(batch, sl), _take(modified, sl), _take(masks, sl))
    assert torch.equal(logits, logits_full[:n * S])
    parts = [_loss_grads(model, ic, _take(batch, q), _take(modified, q), _take(masks, q))
             for q in (slice(0, n // 2), slice(n // 2, n))]
    assert torch.equal(logits, torch.cat([p[1] for p in parts]))
    assert torch.equal(sums[:, 2], parts[0][0][:, 2] + parts[1][0][:, 2])
    assert torch.allclose(sums[:, 0], 0.5 * (parts[0][0][:, 0] + parts[1][0][:, 0]), rtol=2e-5, atol=1e-5)
    want = 0.5 * (parts[0][2] + parts[1][2])
    err, scale = (g - want).abs().max().item(), want.abs().max().item()
    assert err <= 2e-2 * scale, (err, scale)


def test_attention_backward_in_one_launch_equals_the_three_launches():
    """c2 batch (256 documents: the chip is full, so the one-launch attention backward of csrc/block_attn_bwd.hip is what
    the step runs) against the three launches it replaces (mfp_dgrad_d256, mfp_attention_bwd, mfp_dgrad_qkv): same bf16
    operands, only the summation order of dQ over the keys differs -> gradients agree to bf16 rounding of dqkv."""
    from mfp.hip import functions as F
    ic, model = _model("bf16")
    batch, modified, masks, _ = _masked_batch(ic, seed=7)
    keep = F.ATTN_BLOCK_BWD
    try:
        F.ATTN_BLOCK_BWD = "1"
        sums1, logits1, g1 = _loss_grads(model, ic, batch, modified, masks)
        F.ATTN_BLOCK_BWD = "0"
        sums0, logits0, g0 = _loss_grads(model, ic, batch, modified, masks)
    finally:
        F.ATTN_BLOCK_BWD = keep
    assert torch.equal(logits1, logits0) and torch.allclose(sums1, sums0, rtol=1e-5, atol=1e-6)    # (sums: f32 atomics)
    err, scale = (g1 - g0).abs().max().item(), g0.abs().max().item()
    assert err <= 2e-3 * scale, (err, scale)


def test_c4_per_gpu_shape_captured_step_equals_eager_step():
    """c4's per-GPU shard through the product's own train step: the hipGraph replay (what bench.py --config c4 times)
    leaves exactly the parameters the eager step leaves -- same kernels, same counter-based masks and dropout."""
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.mfp import MFP
    ic = make_input_columns("crello")
    batch = synthetic_batch(ic, 128, S, seed=6, ragged=True, device=DEV)
    ws = []
    for graph in (False, True):
        m = MFP(ic, num_blocks=L, latent_dim=D, dropout=0.1, l2=1e-2, masking_method="random", dtype="bf16", device=DEV, seed=5)
        m.compile(learning_rate=1e-3)
        if graph:
            m.capture_train_step(batch, warmup=0)
        for _ in range(3):
            m.train_step(batch)
        torch.cuda.synchronize()
        ws.append(m.model.store.w.clone())
    assert torch.isfinite(ws[0]).all()
    assert torch.equal(ws[0], ws[1])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_padding_is_inert(dtype):
    ic, model = _model(dtype)
    batch, modified, masks, seq_mask = _masked_batch(ic, seed=2)
    sums, logits, g = _loss_grads(model, ic, batch, modified, masks)
    # scribble over every padding position of the model INPUT (valid token ids / finite vectors)
    gen = torch.Generator().manual_seed(5)
    scribbled = dict(modified)
    pad = ~seq_mask
    for k, c in ic.items():
        if c.get("demo_only") or not c.get("is_sequence"):
            continue
        x = modified[k].clone()
        if c["type"] == "categorical":
            noise = torch.randint(0, c["input_dim"], x.shape, generator=gen, dtype=torch.int64).to(x.dtype)
        else:
            noise = torch.randn(x.shape, generator=gen)
        x[pad] = noise[pad]
        scribbled[k] = x
    sums2, logits2, g2 = _loss_grads(model, ic, batch, scribbled, masks)
    valid = seq_mask.reshape(-1).to(DEV)
    assert torch.equal(logits[valid], logits2[valid])
    # per-key sums are accumulated with float atomics across workgroups: equal up to summation order
    assert torch.equal(sums[:, 2], sums2[:, 2])
    assert torch.allclose(sums, sums2, rtol=1e-6, atol=1e-4)
    # parameter gradients: padding rows only reach the encoder tables they index (with zero
    # upstream gradient), so everything is unchanged
    assert torch.equal(g, g2)


def test_set_input_is_permutation_equivariant():
    ic, model = _model("fp32")
    batch, modified, masks, seq_mask = _masked_batch(ic, seed=3, ragged=True)
    rng = np.random.default_rng(0)
    n = (batch["length"].reshape(-1) + 1).tolist()
    perm = torch.tensor(np.stack([np.concatenate([rng.permutation(int(k)), np.arange(int(k), S)]) for k in n]))

    def shuffle(d):
        out = {}
        for k, v in d.items():
            if v.dim() >= 2 and v.shape[1] == S:
                idx = perm.reshape(perm.shape + (1,) * (v.dim() - 2)).expand(-1, -1, *v.shape[2:])
                out[k] = torch.gather(v, 1, idx)
            else:
                out[k] = v
        return out

    sums, logits, g = _loss_grads(model, ic, batch, modified, masks)
    sums_p, logits_p, g_p = _loss_grads(model, ic, shuffle(batch), shuffle(modified), shuffle(masks))
    U = logits.shape[1]
    idx = perm.to(DEV)[:, :, None].expand(-1, -1, U)
    want = torch.gather(logits.view(B, S, U), 1, idx).view(B * S, U)
    valid = seq_mask.reshape(-1).to(DEV)
    # key order inside the softmax sums changes -> f32 rounding only
    assert (logits_p[valid] - want[valid]).abs().max().item() < 2e-4
    assert torch.equal(sums_p[:, 2], sums[:, 2])
    assert torch.allclose(sums_p[:, 0], sums[:, 0], rtol=1e-4, atol=1e-5)
    assert (g_p - g).abs().max().item() <= 2e-4 * g.abs().max().item()


def test_loss_denominators_are_counts():
    from mfp.models.metrics import loss_key_names
    ic, model = _model("bf16")
    batch, modified, masks, seq_mask = _masked_batch(ic, seed=4)
    sums, _, _ = _loss_grads(model, ic, batch, modified, masks)
    for i, k in enumerate(loss_key_names(ic)):
        c = ic[k]
        w = masks[k] & seq_mask
        if "loss_condition" in c:
            cond = torch.tensor(c["loss_condition"]["mask"])[batch[c["loss_condition"]["key"]].long()[..., 0]]
            w = w & cond
        n_feat = c["shape"][-1] if c["type"] == "categorical" else 1
        assert float(sums[i, 2]) == float(w.sum()) * n_feat, k
        assert 0.0 <= float(sums[i, 1]) <= float(sums[i, 2]) + 1e-3, k


def test_fused_masking_at_full_size():
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.architecture.mask import get_seq_mask
    from mfp.models.mfp import MFP
    ic = make_input_columns("crello")
    batch = synthetic_batch(ic, B, S, seed=6, ragged=True, device=DEV)
    model = MFP(ic, num_blocks=1, latent_dim=128, dropout=0.0, dtype="bf16", device=DEV, seed=2, masking_method="random")
    ctx = model.model.make_ctx(batch, True)
    tasks = torch.zeros(B, dtype=torch.int32, device=DEV)
    idx_all, codes, xs, masks = model._masker(batch, tasks, ctx.nvalid, ctx.B, ctx.S, None)
    torch.cuda.synchronize()
    seq_mask = get_seq_mask(batch["length"], maxlen=S)
    nvalid_fields, nmasked = 0, 0
    for k, m in masks.items():
        m = m.bool()
        assert not (m & ~seq_mask).any(), k                     # never on padding
        nvalid_fields += int(seq_mask.sum())
        nmasked += int(m.sum())
    p = nmasked / nvalid_fields
    sigma = (0.15 * 0.85 / nvalid_fields) ** 0.5
    assert abs(p - 0.15) < 5 * sigma + 0.01, (p, sigma)         # loss_condition columns mask a little less
    # categorical index columns: unmasked valid positions carry the input value untouched
    col = model.model.layout.idx_cols.index(("left", 0))
    keep = (seq_mask & ~masks["left"].bool()).reshape(-1)
    assert torch.equal(idx_all[:, col][keep].long(), batch["left"].reshape(-1)[keep].long())


def test_c5_per_gpu_shape_with_reserved_cus_equals_its_halves_and_padding_is_inert():
    """BASELINE config c5 at its full per-GPU size (64 documents x 256 positions, d_model 512, 8 blocks; bf16) with 8 CUs
    reserved for RCCL (MFP_DP_RESERVE_CUS=8, what the driver's 8-GPU run may set): the d_model-512 kernels of
    csrc/block_d512.hip, whose LayerNorm-backward epilogue exchanges row sums between PAIRED workgroups through global memory
    (mfp_dense_n512_lnb: 128 row tiles = 256 workgroups, 16 launches per step).  Size-independent properties: the batch
    equals its two halves (logits bit for bit, denominators exactly, losses / gradients by linearity of the batch mean), and
    what sits in the padded positions changes nothing."""
    from mfp.data.spec import make_input_columns
    from mfp.hip import ops
    from mfp.models.model import Model
    global B, S
    oldB, oldS = B, S
    ic = make_input_columns("crello")
    ops.set_reserved_cus(8)
    try:
        B, S = 64, 256
        model = Model(ic, num_blocks=8, latent_dim=512, dropout=0.0, l2=1e-2, dtype="bf16", device=DEV, seed=5)
        batch, modified, masks, seq_mask = _masked_batch(ic, seed=6, B=B)
        sums, logits, g = _loss_grads(model, ic, batch, modified, masks)
        assert torch.isfinite(logits).all() and torch.isfinite(g).all()
        h = B // 2
        parts = [_loss_grads(model, ic, _take(batch, sl), _take(modified, sl), _take(masks, sl)) for sl in (slice(0, h), slice(h, B))]
        assert torch.equal(logits, torch.cat([p[1] for p in parts]))
        assert torch.equal(sums[:, 2], parts[0][0][:, 2] + parts[1][0][:, 2])
        assert torch.allclose(sums[:, 0], 0.5 * (parts[0][0][:, 0] + parts[1][0][:, 0]), rtol=2e-5, atol=1e-5)
        want = 0.5 * (parts[0][2] + parts[1][2])
        err, scale = (g - want).abs().max().item(), want.abs().max().item()
        assert err <= 2e-2 * scale, (err, scale)
        # padding: other values in the positions >= length (sequence columns only) -- same logits at the valid positions, same sums
        gen = torch.Generator().manual_seed(9)
        pad = ~seq_mask
        noisy = dict(modified)
        for k, c in ic.items():
            if c.get("demo_only") or not c["is_sequence"]:
                continue
            v = modified[k].clone()
            if c["type"] == "categorical":
                v[pad] = torch.randint(0, c["input_dim"], v[pad].shape, generator=gen).to(v.dtype)
            else:
                v[pad] = torch.randn(v[pad].shape, generator=gen)
            noisy[k] = v
        sums2, logits2, g2 = _loss_grads(model, ic, batch, noisy, masks)
        valid = seq_mask.reshape(-1).to(DEV)
        assert torch.equal(logits[valid], logits2[valid])
        assert torch.equal(sums[:, 2], sums2[:, 2]) and torch.allclose(sums[:, 0], sums2[:, 0], rtol=1e-6, atol=1e-6)
    finally:
        ops.set_reserved_cus(0)
        B, S = oldB, oldS
