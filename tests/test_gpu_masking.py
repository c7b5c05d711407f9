"""Fused masking kernel (mfp_mask_tokens) vs the reference-shaped torch masking functions.

Deterministic parts (filter_padding's <UNUSED>, attribute-group and element tasks) must match the
torch implementation exactly; the random task is checked against the reference's probabilities
(MASK_PROB .15; of those 90 % changed, 1/9 of the changed get a random token).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(dataset, B, S, dtype="fp32"):
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.mfp import MFP
    ic = make_input_columns(dataset)
    batch = synthetic_batch(ic, B, S, seed=3, ragged=True, device=DEV)
    model = MFP(ic, num_blocks=1, latent_dim=128, dropout=0.0, l2=1e-2, dtype=dtype, device=DEV, seed=11,
                masking_method="random_elem_pos_attr")
    return ic, batch, model


def _run(model, batch, tasks):
    ctx = model.model.make_ctx(batch, True)
    idx_all, codes, xs, masks = model._masker(batch, tasks, ctx.nvalid, ctx.B, ctx.S, None)
    torch.cuda.synchronize()
    return ctx, idx_all, codes, xs, masks


@pytest.mark.parametrize("dataset", ["crello", "rico"])
def test_group_tasks_match_torch_masking(dataset):
    from mfp.models.mfp import preprocess_for_train
    B, S = 6, 20
    ic, batch, model = _setup(dataset, B, S)
    L = model.model.layout
    ngroups = len(model.task_names) - 2
    tasks = (torch.arange(B, device=DEV) % ngroups + 2).to(torch.int32)
    ctx, idx_all, codes, xs, masks = _run(model, batch, tasks)
    targets, modified, ref_masks = preprocess_for_train(batch, model.input_columns, tasks)
    T = B * S
    pos = 0
    for k in L.cat_keys:
        n = L.columns[k]["shape"][-1]
        assert torch.equal(idx_all[:, pos:pos + n], modified[k].reshape(T, n).to(torch.int32)), k
        pos += n
    for j, k in enumerate(L.num_keys):
        m = modified[k].reshape(T, -1)
        want_code = torch.where((m == 0.0).all(1), 2, torch.where((m == 10.0).all(1), 1, 0))
        assert torch.equal(codes[j].long(), want_code), k
        want_sp = torch.where(want_code == 1, 0, torch.where(want_code == 2, 1, -1))
        assert torch.equal(idx_all[:, L.special_col[k]].long(), want_sp), k
        assert torch.equal(xs[j], m), k
    for k in masks:
        assert torch.equal(masks[k].bool(), ref_masks[k]), k


def test_elem_task_masks_one_valid_element():
    B, S = 64, 16
    ic, batch, model = _setup("crello", B, S)
    L = model.model.layout
    tasks = torch.ones(B, dtype=torch.int32, device=DEV)
    ctx, idx_all, codes, xs, masks = _run(model, batch, tasks)
    nv = ctx.nvalid.long()
    ref = None
    for k, m in masks.items():
        m = m.bool()
        assert (m.sum(1) == 1).all(), k
        sel = m.float().argmax(1)
        assert (sel < nv).all()
        ref = sel if ref is None else ref
        assert torch.equal(sel, ref), k          # same element for every attribute
    # the selected element carries <MASK> in every column, everything else is the filtered input
    t_sel = torch.arange(B, device=DEV) * S + ref
    pos = 0
    for k in L.cat_keys:
        n = L.columns[k]["shape"][-1]
        assert (idx_all[t_sel, pos:pos + n] == L.columns[k]["input_dim"]).all(), k
        pos += n
    for j, k in enumerate(L.num_keys):
        assert (codes[j][t_sel] == 1).all() and (xs[j][t_sel] == 10.0).all()
    assert ref.float().std() > 0.5  # not always the same slot


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_random_task_statistics(dtype):
    from mfp.models import masking
    from mfp.models.architecture.mask import get_seq_mask
    B, S = 256, 64
    ic, batch, model = _setup("crello", B, S, dtype)
    L = model.model.layout
    tasks = torch.zeros(B, dtype=torch.int32, device=DEV)
    ctx, idx_all, codes, xs, masks = _run(model, batch, tasks)
    seq_mask = get_seq_mask(batch["length"], maxlen=S)
    filtered = masking.filter_padding(batch, model.input_columns, seq_mask)
    valid = seq_mask.reshape(-1)
    nvalid = valid.sum().item()
    pos = 0
    for k in L.cat_keys:
        col = L.columns[k]
        n, C = col["shape"][-1], col["input_dim"]
        m = masks[k].bool().reshape(-1)
        assert not (m & ~valid).any()
        frac = m.sum().item() / nvalid
        assert abs(frac - 0.15) < 0.015, (k, frac)
        got = idx_all[:, pos:pos + n].long()
        want = filtered[k].reshape(-1, n).long()
        changed = (got != want).any(1)
        assert not (changed & ~m).any(), k                      # only masked positions change
        is_mask_tok = (got == C).all(1) & m
        f_mask = is_mask_tok.sum().item() / max(m.sum().item(), 1)
        assert abs(f_mask - 0.8) < 0.04, (k, f_mask)
        rnd = m & ~is_mask_tok
        assert (got[rnd] <= C + 1).all() and (got[rnd] >= 0).all()
        pos += n
    for j, k in enumerate(L.num_keys):
        m = masks[k].bool().reshape(-1)
        frac = m.sum().item() / nvalid
        assert abs(frac - 0.15) < 0.015, (k, frac)
        want = filtered[k].reshape(-1, 512)
        x = xs[j].float()
        masked_rows = (codes[j] == 1)
        assert not (masked_rows & ~m).any()
        assert abs(masked_rows.sum().item() / max(m.sum().item(), 1) - 0.8) < 0.04
        keep = ~m
        tol = 0 if dtype == "fp32" else 1e-2
        assert (x[keep] - want[keep]).abs().max().item() <= tol
        unused_rows = (want == 0.0).all(1) & keep
        assert (codes[j][unused_rows] == 2).all()
        changed = m & (codes[j] == 0) & ((x - want).abs().max(1).values > 0.02)
        if changed.sum() > 10:   # random-noise rows: N(0, 0.1)
            assert abs(x[changed].std().item() - 0.1) < 0.01
            assert abs(changed.sum().item() / m.sum().item() - 0.1) < 0.04


def test_masks_change_with_step_counter():
    B, S = 32, 32
    ic, batch, model = _setup("crello", B, S)
    model.compile(learning_rate=1e-4)
    tasks = torch.zeros(B, dtype=torch.int32, device=DEV)
    ctx = model.model.make_ctx(batch, True)
    a = model._masker(batch, tasks, ctx.nvalid, B, S, model.optimizer.step_t)[3]["left"].clone()
    b = model._masker(batch, tasks, ctx.nvalid, B, S, model.optimizer.step_t)[3]["left"].clone()
    model.optimizer.step_t += 1
    c = model._masker(batch, tasks, ctx.nvalid, B, S, model.optimizer.step_t)[3]["left"].clone()
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_sample_tasks_distribution_and_step_offset():
    """mfp_sample_tasks: Categorical(probs) by inverse CDF; zero-probability tasks never drawn; the
    device step counter moves the stream."""
    from mfp.hip import ops
    probs = [0.0, 1.0, 1.0, 0.0, 2.0]
    B = 200000
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    t0 = ops.sample_tasks(probs, B, 1234, 1, step, "cuda")
    cnt = torch.bincount(t0.long(), minlength=5).double().cpu() / B
    want = torch.tensor(probs, dtype=torch.float64) / sum(probs)
    assert cnt[0] == 0 and cnt[3] == 0
    assert (cnt - want).abs().max() < 5e-3, cnt
    assert torch.equal(t0, ops.sample_tasks(probs, B, 1234, 1, step, "cuda"))
    step += 1
    assert not torch.equal(t0, ops.sample_tasks(probs, B, 1234, 1, step, "cuda"))
