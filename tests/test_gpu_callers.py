"""GPU parity of the CALLERS of the hot path against the oracle:

* the fused HIP masking kernel (``mfp_mask_tokens``) against ``oracle/np_masking.py`` -- the kernel
  draws from its own counter-based stream, so the draws are INFERRED from its output (which
  positions it masked, which got <MASK> / a random token / stayed) and replayed through the oracle,
  which must then reproduce the kernel's output bit for bit for every task type and the per-document
  task select (reference masking.py:24-155,227-269, mfp.py:95-138);
* ``MFP.test_step`` / ``MFP.__call__(training=False)`` (Keras validation / test metrics:
  mfp.py:298-347, metrics.py:213-299) per metric key against the oracle's model + LossLayer on the
  very masks the call drew -- Crello and RICO (position-sorted loss), with and without the model's
  padded ``_flat_logits`` buffer;
* ``iterative_decode`` (mfp.py:141-207) through ``MFP.__call__(demo_args={"num_iter": n})`` with
  the real model on the GPU against the oracle's decode driving the oracle's f64 model.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


def _nd(ic):
    return {k: v for k, v in ic.items() if not v.get("demo_only", False)}


# --------------------------------------------------------------------------- fused masking kernel
def _infer_draws(nd, filtered, got_x, got_m, task0):
    """Draws that make the reference's random_masking produce what the kernel produced (documents
    of other tasks: nothing masked).  Asserts the kernel's tokens are legal on the way."""
    draws = {}
    for k, c in nd.items():
        if not c["is_sequence"]:
            continue
        m = got_m[k] & task0[:, None]
        x, f = got_x[k], filtered[k]
        if c["type"] == "categorical":
            is_mask_tok = (x == c["input_dim"]).all(-1)
        else:
            is_mask_tok = (x == 10.0).all(-1)
        changed = (x != f).any(-1)
        rnd = m & changed & ~is_mask_tok
        if c["type"] == "categorical":
            assert ((x[rnd] >= 0) & (x[rnd] < c["input_dim"])).all(), k          # randint(0, input_dim)
        u_mask = np.where(m, 0.0, 1.0)
        u_chg = np.where(m & (is_mask_tok | rnd), 0.0, 1.0)
        u_tok = np.where(rnd, 0.0, 1.0)
        draws[k] = dict(u_mask=u_mask, u_chg=u_chg, u_tok=u_tok, random=x)
    return draws


@pytest.mark.parametrize("dataset,method", [("crello", "random_elem_type_pos_attr_img_txt"),
                                            ("crello", "elem_pos_attr_img_txt"), ("rico", "random_elem_type_pos_attr")])
def test_fused_masking_equals_oracle_on_inferred_draws(dataset, method):
    from oracle import np_masking as om
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.mfp import MFP
    ic = make_input_columns(dataset)
    nd = _nd(ic)
    B, S = 48, 40
    batch = synthetic_batch(ic, B, S, seed=5, ragged=True, device=DEV)
    model = MFP(ic, num_blocks=1, latent_dim=128, dropout=0.0, l2=1e-2, dtype="fp32", device=DEV, seed=13,
                masking_method=method)
    L = model.model.layout
    probs = om.task_probs(om.get_task_names(nd), method)
    assert probs == model.task_probs
    tasks = om.sample_tasks(probs, np.random.default_rng(2).permutation(B) / B + 0.5 / B)
    assert set(tasks.tolist()) == {i for i, p in enumerate(probs) if p > 0}
    ctx = model.model.make_ctx(batch, True)
    idx_all, codes, xs, masks = model._masker(batch, torch.from_numpy(tasks).to(DEV), ctx.nvalid, B, S, None)
    torch.cuda.synchronize()
    # the kernel's outputs in the reference's (modified_inputs, masks) form
    got_x, got_m, pos = {}, {}, 0
    for k in L.cat_keys:
        n = L.columns[k]["shape"][-1]
        got_x[k] = idx_all[:, pos:pos + n].reshape(B, S, n).cpu().numpy()
        pos += n
    for j, k in enumerate(L.num_keys):
        got_x[k] = xs[j].reshape(B, S, -1).float().cpu().numpy()
    got_m = {k: v.bool().cpu().numpy() for k, v in masks.items()}
    nb = _np(batch)
    seq_mask = om.get_seq_mask(nb["length"], S)
    filtered = om.filter_padding(nb, nd, seq_mask)
    draws = _infer_draws(nd, filtered, got_x, got_m, tasks == 0) if (tasks == 0).any() else None
    # elem task: the selected slot, as the uniform that selects it (masking.py:108: int(u * length))
    first = L.cat_keys[0]
    sel = got_m[first].argmax(1)
    length = seq_mask.sum(1)
    u_elem = ((sel + 0.5) / length).astype(np.float32)
    _, want_x, want_m = om.preprocess_for_train(nb, nd, tasks, draws, u_elem, maxlen=S)
    for k, c in nd.items():
        if not c["is_sequence"]:
            continue
        assert np.array_equal(got_m[k], want_m[k]), (k, "mask")
        assert np.array_equal(got_x[k], want_x[k]), (k, "tokens")
        assert not (got_m[k] & ~seq_mask).any(), k
    # row codes / special-token indices of the numerical columns follow from the rows (encoder.py:165-175)
    for j, k in enumerate(L.num_keys):
        rows = want_x[k].reshape(B * S, -1)
        code = np.where((rows == 0.0).all(1), 2, np.where((rows == 10.0).all(1), 1, 0))
        assert np.array_equal(codes[j].cpu().numpy(), code), k
    # per-task sanity of what was just compared: elem documents mask exactly one valid element everywhere
    for b in np.nonzero(tasks == 1)[0]:
        for k in got_m:
            assert got_m[k][b].sum() == 1 and got_m[k][b].argmax() == sel[b] and sel[b] < length[b]


@pytest.mark.parametrize("B,S,dtype", [(48, 40, "fp32"), (5, 13, "bf16"), (64, 128, "bf16")])
def test_mask_kernel_forms_agree(B, S, dtype):
    """mask_kernel (round 5: sixteen tokens per workgroup, numerical rows streamed with their loads up front) against the
    wave-per-token form it replaces (MFP_MASK_WAVE=1): Philox is keyed by (seed, token, column / feature / float4 index, step), so
    every output must be bit-identical -- every task type in the mix, ragged lengths, a token count that is not a multiple of 16."""
    import os
    from oracle import np_masking as om
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.mfp import MFP
    ic = make_input_columns("crello")
    method = "random_elem_type_pos_attr_img_txt"
    batch = synthetic_batch(ic, B, S, seed=6, ragged=True, device=DEV)
    model = MFP(ic, num_blocks=1, latent_dim=128, dropout=0.0, l2=1e-2, dtype=dtype, device=DEV, seed=13, masking_method=method)
    probs = om.task_probs(om.get_task_names(_nd(ic)), method)
    tasks = torch.from_numpy(om.sample_tasks(probs, np.random.default_rng(3).permutation(B) / B + 0.5 / B)).to(DEV)
    ctx = model.model.make_ctx(batch, True)
    outs = []
    old = os.environ.get("MFP_MASK_WAVE")
    try:
        for flag in ("1", "0"):
            os.environ["MFP_MASK_WAVE"] = flag
            idx_all, codes, xs, masks = model._masker(batch, tasks, ctx.nvalid, B, S, None)
            torch.cuda.synchronize()
            outs.append((idx_all.clone(), [c.clone() for c in codes], [x.clone() for x in xs], {k: v.clone() for k, v in masks.items()}))
    finally:
        if old is None:
            os.environ.pop("MFP_MASK_WAVE", None)
        else:
            os.environ["MFP_MASK_WAVE"] = old
    (i0, c0, x0, m0), (i1, c1, x1, m1) = outs
    assert torch.equal(i0, i1)
    for a, b in zip(c0 + x0, c1 + x1):
        assert torch.equal(a.view(torch.uint8) if a.dtype == torch.bfloat16 else a, b.view(torch.uint8) if b.dtype == torch.bfloat16 else b)
    for k in m0:
        assert torch.equal(m0[k], m1[k]), k
    assert any(int((x == 10.0).all(-1).sum()) > 0 for x in x1)      # (something was masked)


# ------------------------------------------------------------------ validation / test metric path
def _oracle_metrics(ic, params, L, S, targets, modified, masks, sort_flag=None):
    from oracle import np_ref
    out = np_ref.model_fwd(params, ic, _np(modified), L, maxlen=S)
    return np_ref.loss_layer(ic, _np(targets), out, _np(masks), maxlen=S, sort_flag=sort_flag), out


@pytest.mark.parametrize("dataset,method", [("crello", "random"), ("crello", "elem_pos_attr_img_txt"),
                                            ("rico", "random_pos_elem")])
def test_test_step_and_call_match_oracle(dataset, method, monkeypatch):
    """The Keras validation/test metrics (``val_*``, best-checkpoint selection, the printed test
    metrics) are computed by MFP.test_step / MFP.__call__(training=False): per key <key>_loss,
    <key>_score and total_score must match the oracle to 1e-4 on the masks the call drew."""
    from oracle import np_ref
    import mfp.models.mfp as mfp_mod
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.metrics import loss_key_names
    ic = make_input_columns(dataset)
    B, S, D, L = 6, 14, 128, 2
    params = np_ref.init_params(ic, D, L, seed=-5)
    batch = synthetic_batch(ic, B, S, seed=8, ragged=True, device=DEV)
    model = mfp_mod.MFP(ic, num_blocks=L, latent_dim=D, dropout=0.1, l2=1e-2, dtype="fp32", device=DEV,
                        masking_method=method)
    model.model.store.load_state_dict(params)
    seen = {}
    real = mfp_mod.preprocess_for_train

    def spy(inputs, input_columns, tasks, **kw):
        out = real(inputs, input_columns, tasks, **kw)
        seen["tasks"], seen["out"] = tasks, out
        return out
    monkeypatch.setattr(mfp_mod, "preprocess_for_train", spy)
    keys = loss_key_names(ic)

    def check(metrics):
        targets, modified, masks = seen["out"]
        flag = None
        if model.sort_pos:
            flag = (seen["tasks"] == model.task_names.index("pos")).cpu().numpy()
        (loss_total, losses, scores, want), _ = _oracle_metrics(ic, params, L, S, targets, modified, masks, flag)
        for k in keys:
            assert abs(float(metrics[k + "_loss"]) - losses[k]) <= 1e-4 * max(1.0, abs(losses[k])), (k, "loss")
            assert abs(float(metrics[k + "_score"]) - want[k + "_score"]) <= 1e-4, (k, "score")
        assert abs(float(metrics["total_score"]) - want["total_score"]) <= 1e-4
        return loss_total

    torch.manual_seed(3)
    sums = model.test_step(batch)
    loss_total = check(model.loss_layer.metrics)
    assert abs(float(sums[:, 0].sum()) - loss_total) <= 1e-4 * max(1.0, loss_total)
    torch.manual_seed(4)
    out = model(batch, training=False)                      # mfp.py:298-347 with is_demo False
    check(model.loss_layer.metrics)
    assert "_flat_logits" not in out
    # LossLayer on the split per-key logits (no padded buffer) gives the same sums as on the buffer
    targets, modified, masks = seen["out"]
    y = model.model(modified, training=False)
    flag = (seen["tasks"] == model.task_names.index("pos")) if model.sort_pos else None
    model.loss_layer((targets, dict(y), masks), False, flag)
    with_flat = model.loss_layer.sums.clone()
    y.pop("_flat_logits")
    model.loss_layer((targets, y, masks), False, flag)
    assert torch.allclose(with_flat, model.loss_layer.sums, rtol=1e-6, atol=1e-6)
    # and a LossLayer built without the model's layout (eval.py:49) must not be fooled by the buffer
    from mfp.models.metrics import LossLayer
    plain = LossLayer(ic)
    plain((targets, dict(model.model(modified, training=False)), masks), False, flag)
    assert torch.allclose(with_flat, plain.sums, rtol=1e-6, atol=1e-6)


def test_evaluate_is_mean_over_batches_of_oracle_metrics(monkeypatch):
    """Keras ``evaluate``: mean over batches of the per-batch metric values."""
    from oracle import np_ref
    import mfp.models.mfp as mfp_mod
    from mfp.data.spec import make_input_columns, synthetic_batch
    ic = make_input_columns("crello")
    B, S, D, L = 4, 10, 128, 1
    params = np_ref.init_params(ic, D, L, seed=-6)
    model = mfp_mod.MFP(ic, num_blocks=L, latent_dim=D, dropout=0.0, l2=1e-2, dtype="fp32", device=DEV)
    model.model.store.load_state_dict(params)
    batches = [synthetic_batch(ic, B if i < 2 else 3, S, seed=20 + i, ragged=True, device=DEV) for i in range(3)]
    seen = []
    real = mfp_mod.preprocess_for_train

    def spy(inputs, input_columns, tasks, **kw):
        out = real(inputs, input_columns, tasks, **kw)
        seen.append(out)
        return out
    monkeypatch.setattr(mfp_mod, "preprocess_for_train", spy)
    res = model.evaluate(batches, return_dict=True)
    want_total, want_loss = 0.0, 0.0
    for targets, modified, masks in seen:
        (loss_total, losses, scores, m), _ = _oracle_metrics(ic, params, L, S, targets, modified, masks)
        want_total += m["total_score"] / len(seen)
        want_loss += loss_total / len(seen)
    assert abs(res["total_score"] - want_total) <= 1e-4
    assert abs(res["loss"] - want_loss) <= 1e-4 * max(1.0, want_loss)


# ----------------------------------------------------------------------------- iterative decode
@pytest.mark.parametrize("dataset,num_iter", [("crello", 3), ("rico", 2)])
def test_iterative_decode_gpu_vs_oracle(dataset, num_iter):
    from oracle import np_masking as om, np_ref
    from mfp.data.spec import make_input_columns, synthetic_batch
    from mfp.models.mfp import MFP
    ic = make_input_columns(dataset)
    nd = _nd(ic)
    B, S, D, L = 3, 12, 128, 2
    params = np_ref.init_params(ic, D, L, seed=-7)
    # sharpen the heads so that confidences are well separated (thresholding is not tolerance-sensitive then)
    for k in list(params):
        if k.startswith("decoder/") and k.endswith("/kernel"):
            params[k] = (params[k] * 4.0).astype(np.float32)
    batch = synthetic_batch(ic, B, S, seed=12, ragged=True, device=DEV)
    model = MFP(ic, num_blocks=L, latent_dim=D, dropout=0.1, l2=1e-2, dtype="fp32", device=DEV)
    model.model.store.load_state_dict(params)
    nb = _np(batch)
    seq_mask = om.get_seq_mask(nb["length"], S)
    rng = np.random.default_rng(4)
    masks = {k: ((rng.random((B, S)) < 0.6) & seq_mask if c["is_sequence"] else np.ones(B, bool)) for k, c in nd.items()}
    tmasks = {k: torch.from_numpy(v).to(DEV) for k, v in masks.items()}
    got = model(batch, training=False, demo_args={"masks": tmasks, "num_iter": num_iter})

    def oracle_model(x):
        x = dict(x)
        x["length"] = nb["length"]
        return np_ref.model_fwd(params, ic, x, L, maxlen=S)
    final = om.iterative_decode(oracle_model, masks, nb, nd, om.preprocess_for_test(nb, nd, masks, maxlen=S),
                                num_iter, maxlen=S)
    want = om.merge_inputs_and_prediction(nb, nd, masks, final)
    once = oracle_model(om.preprocess_for_test(nb, nd, masks, maxlen=S))
    differs = False
    for k, c in nd.items():
        if not c["is_sequence"]:
            continue
        err = np.abs(got[k].cpu().double().numpy() - np.asarray(want[k], np.float64)).max()
        assert err < 2e-4, (k, err)
        differs |= not np.allclose(final[k], once[k], atol=1e-3)
    assert differs          # decoding in several iterations changed the prediction of some field
