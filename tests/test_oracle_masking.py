"""Caller-side producers of the hot path: the oracle restatement
``oracle/np_masking.py`` of reference models/masking.py + models/mfp.py:34-207 --

* hand-derived known answers of the oracle itself,
* the product's reference-shaped torch functions (``mfp.models.masking``, ``mfp.models.mfp``) against
  the oracle on REPLAYED draws (the torch generator's stream is regenerated in the order the
  reference draws: per attribute u_mask, u_chg, u_tok, replacement tokens),
* ``iterative_decode`` against the oracle with a deterministic stand-in model.

CPU only (torch masking is plain torch); the fused HIP masking kernel has the same comparison in
tests/test_gpu_masking.py.
"""
import numpy as np
import pytest
import torch

from oracle import np_masking as om


def _cols(dataset):
    from mfp.data.spec import make_input_columns
    ic = make_input_columns(dataset)
    return ic, {k: v for k, v in ic.items() if not v.get("demo_only", False)}


def _batch(ic, B, S, seed=0):
    from mfp.data.spec import synthetic_batch
    return synthetic_batch(ic, B, S, seed=seed, ragged=True)


def _np(d):
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


# ------------------------------------------------------------------------------- known answers
def test_constants_and_task_mix():
    ic, nd = _cols("crello")
    assert om.get_task_names(nd) == ["random", "elem", "type", "pos", "attr", "img", "txt"]
    assert om.task_probs(om.get_task_names(nd), "random") == [1.0, 0, 0, 0, 0, 0, 0]
    p = om.task_probs(om.get_task_names(nd), "elem_pos_attr_img_txt")       # Ours-EXP (BASELINE config 3)
    assert p == [0, 0.2, 0, 0.2, 0.2, 0.2, 0.2]
    ic, nd = _cols("rico")
    assert om.get_task_names(nd) == ["random", "elem", "type", "pos", "attr"]
    assert abs(om.THRESH - 1.0 / 9.0) < 1e-12 and om.CHANGE_PROB == 0.9
    t = om.sample_tasks([0, 0.2, 0, 0.2, 0.2, 0.2, 0.2], np.array([0.0, 0.19, 0.2, 0.5, 0.99999]))
    assert t.tolist() == [1, 1, 3, 4, 6]                                    # zero-probability tasks never drawn


def test_apply_token_known_answers():
    col_c = {"type": "categorical", "input_dim": 7}
    x = np.array([[[1], [2], [3]]], np.int32)
    m = np.array([[True, False, True]])
    assert om.apply_token(x, col_c, m, "masked")[0, :, 0].tolist() == [7, 2, 7]      # <MASK> = C
    assert om.apply_token(x, col_c, m, "unused")[0, :, 0].tolist() == [8, 2, 8]      # <UNUSED> = C + 1
    assert om.apply_token(x, col_c, m, "random", np.full(x.shape, 5))[0, :, 0].tolist() == [5, 2, 5]
    col_n = {"type": "numerical"}
    y = np.arange(6, dtype=np.float32).reshape(1, 3, 2)
    out = om.apply_token(y, col_n, m, "masked")
    assert out[0, 0].tolist() == [10.0, 10.0] and out[0, 1].tolist() == [2.0, 3.0]
    assert om.apply_token(y, col_n, m, "unused")[0, 2].tolist() == [0.0, 0.0]


def test_select_single_element_known_answers():
    mask = om.get_seq_mask(np.array([[3], [0], [1]]), 5)            # lengths (zero-based) -> 4, 1, 2 valid
    sel = om.select_single_element(mask, np.array([0.0, 0.999, 0.5], np.float32))
    assert sel.argmax(1).tolist() == [0, 0, 1] and sel.sum(1).tolist() == [1, 1, 1]
    sel = om.select_single_element(mask, np.array([0.9999, 0.0, 0.49], np.float32))
    assert sel.argmax(1).tolist() == [3, 0, 0]
    assert om.select_single_element(mask, select_last=True).argmax(1).tolist() == [3, 0, 1]
    empty = np.zeros((2, 4), bool)
    assert not om.select_single_element(empty, np.array([0.3, 0.7], np.float32)).any()


def test_filter_padding_marks_padding_and_missing_attributes():
    ic, nd = _cols("crello")
    b = _np(_batch(ic, 4, 9, seed=2))
    seq_mask = om.get_seq_mask(b["length"], 9)
    f = om.filter_padding(b, nd, seq_mask)
    for k, c in nd.items():
        if not c["is_sequence"]:
            continue
        want_unused = ~seq_mask
        if "loss_condition" in c:
            cond = c["loss_condition"]
            bad = ~np.asarray(cond["mask"])[b[cond["key"]][..., 0]]
            want_unused = want_unused | bad
        if c["type"] == "categorical":
            assert (f[k][want_unused] == c["input_dim"] + 1).all(), k
        else:
            assert (f[k][want_unused] == 0.0).all(), k
        assert np.array_equal(f[k][~want_unused], b[k][~want_unused]), k


# ------------------------------------------------------------------- product torch functions vs oracle
def _replay_draws(nd, batch, seed):
    """The stream the product's torch ``random_masking`` consumes from Generator(seed), regenerated in
    the reference's draw order (masking.py:248,252-255, apply_token :83/:91)."""
    gen = torch.Generator().manual_seed(seed)
    draws = {}
    for k, c in nd.items():
        if not c["is_sequence"]:
            continue
        shape = batch[k].shape[:-1]
        d = dict(u_mask=torch.rand(shape, generator=gen).numpy(), u_chg=torch.rand(shape, generator=gen).numpy(),
                 u_tok=torch.rand(shape, generator=gen).numpy())
        if c["type"] == "categorical":
            d["random"] = torch.randint(0, c["input_dim"], batch[k].shape, generator=gen).numpy()
        else:
            d["random"] = (0.1 * torch.randn(batch[k].shape, generator=gen)).numpy()
        draws[k] = d
    return draws


@pytest.mark.parametrize("dataset", ["crello", "rico"])
def test_torch_random_masking_equals_oracle_on_replayed_draws(dataset):
    from mfp.models import masking
    from mfp.models.architecture.mask import get_seq_mask
    ic, nd = _cols(dataset)
    B, S = 16, 24
    batch = _batch(ic, B, S, seed=4)
    seq_mask = get_seq_mask(batch["length"], maxlen=S)
    filtered = masking.filter_padding(batch, nd, seq_mask)
    got_x, got_m = masking.random_masking(filtered, nd, seq_mask, generator=torch.Generator().manual_seed(77))
    o_filtered = om.filter_padding(_np(batch), nd, om.get_seq_mask(_np(batch)["length"], S))
    want_x, want_m = om.random_masking(o_filtered, nd, seq_mask.numpy(), _replay_draws(nd, batch, 77))
    n_masked = 0
    for k, c in nd.items():
        assert np.array_equal(filtered[k].numpy(), o_filtered[k]), k
        assert np.array_equal(got_m[k].numpy(), want_m[k]), k
        assert np.array_equal(got_x[k].numpy(), want_x[k]), k
        if c["is_sequence"]:
            n_masked += int(want_m[k].sum())
            assert not (want_m[k] & ~seq_mask.numpy()).any()
    assert n_masked > 20                                           # the comparison is not vacuous


@pytest.mark.parametrize("dataset,method", [("crello", "random_elem_type_pos_attr_img_txt"), ("rico", "elem_pos_attr")])
def test_torch_preprocess_for_train_equals_oracle(dataset, method):
    """Task select (mfp.py:119-134) over every task variant, with the random and elem draws replayed."""
    from mfp.models.mfp import preprocess_for_train
    ic, nd = _cols(dataset)
    B, S = 21, 12
    batch = _batch(ic, B, S, seed=9)
    names = om.get_task_names(nd)
    probs = om.task_probs(names, method)
    tasks = om.sample_tasks(probs, np.random.default_rng(1).permutation(B) / B + 0.5 / B)   # stratified uniforms
    assert set(tasks.tolist()) == {i for i, p in enumerate(probs) if p > 0}      # every active task occurs
    gen = torch.Generator().manual_seed(5)
    active = [i for i, p in enumerate(probs) if p > 0]
    _, got_x, got_m = preprocess_for_train(batch, nd, torch.from_numpy(tasks), active_tasks=active, generator=gen)
    # the product draws random_masking's stream first (if task 0 is active), then elem's one uniform per document
    gen2 = torch.Generator().manual_seed(5)
    draws = None
    if 0 in active:
        draws = {}
        for k, c in nd.items():
            if not c["is_sequence"]:
                continue
            shape = batch[k].shape[:-1]
            d = dict(u_mask=torch.rand(shape, generator=gen2).numpy(), u_chg=torch.rand(shape, generator=gen2).numpy(),
                     u_tok=torch.rand(shape, generator=gen2).numpy())
            d["random"] = (torch.randint(0, c["input_dim"], batch[k].shape, generator=gen2).numpy()
                           if c["type"] == "categorical" else (0.1 * torch.randn(batch[k].shape, generator=gen2)).numpy())
            draws[k] = d
    u_elem = torch.rand((B,), generator=gen2).numpy()
    _, want_x, want_m = om.preprocess_for_train(_np(batch), nd, tasks, draws, u_elem, maxlen=S)
    for k, c in nd.items():
        assert np.array_equal(got_x[k].numpy(), want_x[k]), k
        if c["is_sequence"]:
            assert np.array_equal(got_m[k].numpy(), want_m[k]), k
    assert np.array_equal(got_x["task"].numpy(), want_x["task"])


def test_torch_preprocess_for_test_equals_oracle():
    from mfp.models.mfp import preprocess_for_test
    ic, nd = _cols("crello")
    B, S = 5, 10
    batch = _batch(ic, B, S, seed=3)
    rng = np.random.default_rng(0)
    seq_mask = om.get_seq_mask(_np(batch)["length"], S)
    masks = {k: (rng.random((B, S)) < 0.4) & seq_mask for k, c in nd.items() if c["is_sequence"]}
    for k, c in nd.items():
        if not c["is_sequence"]:
            masks[k] = np.ones(B, bool)
    got = preprocess_for_test(batch, nd, {k: torch.from_numpy(v) for k, v in masks.items()})
    want = om.preprocess_for_test(_np(batch), nd, masks, maxlen=S)
    for k in nd:
        assert np.array_equal(got[k].numpy(), want[k]), k


# ----------------------------------------------------------------------------- iterative decode
class _StandInModel:
    """Deterministic stand-in for ``Model.__call__``: logits are a fixed random projection of the
    (masked) inputs, so that unmasking a field changes the next iteration's prediction."""

    def __init__(self, nd, seed=0):
        self.nd = nd
        rng = np.random.default_rng(seed)
        self.seq_keys = [k for k, c in nd.items() if c["is_sequence"]]
        self.w = {k: rng.standard_normal((len(self.seq_keys), (c["shape"][-1] * c["input_dim"]
                                                               if c["type"] == "categorical" else c["shape"][-1])))
                  for k, c in nd.items() if c["is_sequence"]}

    def features(self, x):
        f = []
        for k in self.seq_keys:
            v = np.asarray(x[k], np.float64)
            f.append(np.cos(v.sum(-1) * 0.37 + 0.1 * len(f)))
        f = np.stack(f, -1)                                  # (B,S,nkeys)
        return f + 0.3 * f.mean(axis=1, keepdims=True)      # couple positions of a document

    def np_call(self, x):
        f = self.features(x)
        out = {}
        for k in self.seq_keys:
            c = self.nd[k]
            y = 3.0 * f @ self.w[k]
            out[k] = (y.reshape(y.shape[:2] + (c["shape"][-1], c["input_dim"])) if c["type"] == "categorical" else y)
        return out

    def __call__(self, x, training=False):
        return {k: torch.from_numpy(v) for k, v in self.np_call(_np(x)).items()}


@pytest.mark.parametrize("dataset,num_iter", [("crello", 3), ("rico", 4), ("crello", 2)])
def test_iterative_decode_equals_oracle(dataset, num_iter):
    from mfp.models.mfp import iterative_decode, preprocess_for_test
    ic, nd = _cols(dataset)
    B, S = 4, 11
    batch = _batch(ic, B, S, seed=6)
    seq_mask = om.get_seq_mask(_np(batch)["length"], S)
    rng = np.random.default_rng(3)
    masks = {k: ((rng.random((B, S)) < 0.6) & seq_mask if c["is_sequence"] else np.ones(B, bool)) for k, c in nd.items()}
    tmasks = {k: torch.from_numpy(v) for k, v in masks.items()}
    model = _StandInModel(nd)
    got = iterative_decode(model, tmasks, batch, nd, preprocess_for_test(batch, nd, tmasks), num_iter)
    want = om.iterative_decode(model.np_call, masks, _np(batch), nd,
                               om.preprocess_for_test(_np(batch), nd, masks, maxlen=S), num_iter, maxlen=S)
    once = model.np_call(om.preprocess_for_test(_np(batch), nd, masks, maxlen=S))
    differs = False
    for k in want:
        assert np.allclose(got[k].numpy(), want[k], rtol=0, atol=1e-12), k
        differs |= not np.allclose(want[k], once[k])
    assert differs                                                   # iterating changed something


def test_merge_inputs_and_prediction_equals_oracle():
    from mfp.models.mfp import merge_inputs_and_prediction
    ic, nd = _cols("crello")
    B, S = 3, 7
    batch = _batch(ic, B, S, seed=1)
    rng = np.random.default_rng(2)
    masks = {k: rng.random((B, S)) < 0.5 for k, c in nd.items() if c["is_sequence"]}
    model = _StandInModel(nd)
    pred = model.np_call(_np(batch))
    got = merge_inputs_and_prediction(batch, nd, {k: torch.from_numpy(v) for k, v in masks.items()},
                                      {k: torch.from_numpy(v.copy()) for k, v in pred.items()})
    want = om.merge_inputs_and_prediction(_np(batch), nd, masks, pred)
    for k in want:
        assert np.allclose(np.asarray(got[k]), np.asarray(want[k])), k
