"""CPU tests of the ORACLE itself (numpy f64 vs torch restatement, known answers, finite-difference
gradients, golden fixtures).  The reference holds no tests/golden vectors for this path and
TensorFlow is not importable here, so parity is UNPINNED against the reference (see oracle/np_ref.py);
these tests pin the two independent restatements against each other and against hand-derived values.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _ic(name="crello"):
    from mfp.data.spec import make_input_columns
    return make_input_columns(name)


def _batch(ic, B, S, seed=1):
    from mfp.data.spec import synthetic_batch
    return synthetic_batch(ic, B, S, seed=seed, ragged=True)


@pytest.mark.parametrize("dataset,D,L", [("crello", 16, 2), ("rico", 32, 1)])
def test_numpy_and_torch_restatements_agree(dataset, D, L):
    ic = _ic(dataset)
    B, S = 3, 7
    params = np_ref.init_params(ic, D, L, seed=-1)
    batch = _batch(ic, B, S)
    nb = {k: v.numpy() for k, v in batch.items()}
    out = np_ref.model_fwd(params, ic, nb, L, maxlen=S)
    p = torch_ref.to_torch(params, torch.float64)
    out_t = torch_ref.model_fwd(p, ic, batch, L, maxlen=S)
    for k in out:
        np.testing.assert_allclose(out[k], out_t[k].detach().numpy(), rtol=1e-10, atol=1e-12)
    g = torch.Generator().manual_seed(0)
    masks = {k: torch.rand(B, S, generator=g) < 0.5 for k in out}
    lt, losses, scores, metrics = np_ref.loss_layer(ic, nb, out, {k: v.numpy() for k, v in masks.items()}, S)
    lt2, losses2, scores2, metrics2 = torch_ref.loss_layer(ic, batch, out_t, masks, S)
    assert abs(lt - float(lt2)) < 1e-9 * abs(lt)
    for k in losses:
        assert abs(losses[k] - float(losses2[k])) < 1e-9 * max(1.0, abs(losses[k]))
    assert abs(metrics["total_score"] - float(metrics2["total_score"])) < 1e-12
    assert abs(np_ref.l2_loss(params, 1e-2) - float(torch_ref.l2_loss(p, 1e-2))) < 1e-9


def test_parameter_count_matches_survey():
    """2 812 770 parameters at D=256, L=4 (Crello, C_t=7, C_f=35)."""
    shapes = np_ref.param_shapes(_ic("crello"), 256, 4)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 2812770


def test_known_answers():
    # uniform logits -> CE = ln C, argmax = 0 (first index)
    C = 16
    loss, score = np_ref.categorical_metric(np.array([[3]]), np.zeros((1, 1, C)))
    assert abs(loss[0, 0] - math.log(C)) < 1e-12 and score[0, 0] == 0.0
    loss, score = np_ref.categorical_metric(np.array([[0]]), np.zeros((1, 1, C)))
    assert score[0, 0] == 1.0
    # Keras clip: a confident wrong prediction costs -log(1e-7) + log(sum of clipped p)
    z = np.array([[[60.0, -60.0, 0.0]]])
    loss, _ = np_ref.categorical_metric(np.array([[1]]), z)
    p = np.clip(np_ref.softmax(z), 1e-7, 1 - 1e-7)
    assert abs(loss[0, 0] - (-math.log(1e-7) + math.log(p.sum()))) < 1e-9
    # LayerNorm of a constant row -> beta
    beta = np.array([0.5, -1.0, 2.0, 0.0])
    y = np_ref.layer_norm(np.full((2, 4), 3.0), np.ones(4), beta)
    np.testing.assert_allclose(y, np.broadcast_to(beta, (2, 4)), atol=1e-12)
    # cosine score: identical -> 1.0, opposite -> 0.0; mse x width = SSE
    a = np.random.default_rng(0).standard_normal((2, 3, 8))
    l, s = np_ref.continuous_metric(a, a)
    np.testing.assert_allclose(s, 1.0, atol=1e-12)
    np.testing.assert_allclose(l, 0.0, atol=1e-12)
    l, s = np_ref.continuous_metric(a, -a)
    np.testing.assert_allclose(s, 0.0, atol=1e-12)
    np.testing.assert_allclose(l * 8, (4 * a * a).sum(-1), rtol=1e-12)
    # sequence mask is zero-based (mask.py:29)
    m = np_ref.get_seq_mask(np.array([[0], [2]]))
    assert m.tolist() == [[True, False, False], [True, True, True]]


def test_loss_layer_edge_cases():
    ic = _ic("crello")
    B, S, D, L = 2, 5, 16, 1
    params = np_ref.init_params(ic, D, L, seed=-2)
    batch = {k: v.numpy() for k, v in _batch(ic, B, S, seed=3).items()}
    out = np_ref.model_fwd(params, ic, batch, L, maxlen=S)
    none = {k: np.zeros((B, S), bool) for k in out}
    lt, losses, scores, metrics = np_ref.loss_layer(ic, batch, out, none, S)
    assert lt == 0.0
    for k in out:                                   # den == 0 -> normalized score 1.0 (metrics.py:281)
        assert metrics[k + "_score"] == 1.0 and scores[k + "_score_den"] == 0.0
    assert abs(metrics["total_score"] - len(out) / len(ic)) < 1e-12   # divisor counts ALL columns (:298)
    # padded keys get attention weight exactly 0: outputs of valid elements do not depend on padding
    b2 = dict(batch)
    b2["left"] = batch["left"].copy()
    pad = ~np_ref.get_seq_mask(batch["length"], S)
    b2["left"][pad] = 5
    out2 = np_ref.model_fwd(params, ic, b2, L, maxlen=S)
    valid = ~pad
    for k in out:
        np.testing.assert_allclose(out[k][valid], out2[k][valid], rtol=0, atol=1e-12)


def test_encoder_special_tokens():
    ic = _ic("crello")
    D = 16
    params = np_ref.init_params(ic, D, 1, seed=-3)
    batch = {k: v.numpy() for k, v in _batch(ic, 1, 3, seed=4).items()}
    batch["length"][:] = 2
    zero = {k: (np.zeros_like(v) if v.dtype.kind == "f" else v) for k, v in batch.items()}
    m = dict(zero)
    m["image_embedding"] = zero["image_embedding"].copy()
    m["image_embedding"][0, 0] = 10.0        # <MASK> row -> exactly special[0]
    h_m, _ = np_ref.encoder_fwd(params, ic, m, 3)
    h_0, _ = np_ref.encoder_fwd(params, ic, zero, 3)
    sp = params["encoder/input_image_embedding_special/embeddings"].astype(np.float64)
    np.testing.assert_allclose(h_m[0, 0] - h_0[0, 0], sp[0] - sp[1], atol=1e-12)   # all-zero row -> <UNUSED>


def test_gradients_by_finite_differences():
    ic = _ic("rico")
    B, S, D, L = 2, 4, 16, 1
    params = np_ref.init_params(ic, D, L, seed=-4)
    batch = _batch(ic, B, S, seed=5)
    g = torch.Generator().manual_seed(1)
    keys = [k for k, c in ic.items() if c.get("is_sequence")]
    masks = {k: torch.rand(B, S, generator=g) < 0.7 for k in keys}
    state = torch_ref.TrainState(params, l2=1e-2, clipnorm=None, dtype=torch.float64)
    info, grads = torch_ref.loss_and_grads(state, ic, batch, batch, masks, L, maxlen=S)
    nb = {k: v.numpy() for k, v in batch.items()}
    nm = {k: v.numpy() for k, v in masks.items()}

    def total(p):
        out = np_ref.model_fwd(p, ic, nb, L, maxlen=S)
        return np_ref.loss_layer(ic, nb, out, nm, S)[0] + np_ref.l2_loss(p, 1e-2)

    assert abs(total({k: v.astype(np.float64) for k, v in params.items()}) - float(info["total"])) < 1e-9
    rng = np.random.default_rng(0)
    for name in ["encoder/input_left/embeddings", "blocks/seq2seq_0/attn/dense_key/kernel",
                 "blocks/seq2seq_0/norm2/gamma", "blocks/seq2seq_0/mlp/dense_0/bias", "decoder/decoder_type/kernel"]:
        p64 = {k: v.astype(np.float64) for k, v in params.items()}
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in p64[name].shape)
            eps = 1e-6
            old = p64[name][idx]
            p64[name][idx] = old + eps
            up = total(p64)
            p64[name][idx] = old - eps
            dn = total(p64)
            p64[name][idx] = old
            fd = (up - dn) / (2 * eps)
            an = float(grads[name][idx])
            assert abs(fd - an) < 1e-5 * max(1.0, abs(an)), (name, idx, fd, an)


def test_adam_keras_and_clipnorm():
    g = np.array([3.0, 4.0])
    np.testing.assert_allclose(np_ref.clip_by_norm(g, 1.0), g / 5.0)
    np.testing.assert_allclose(np_ref.clip_by_norm(g * 0.01, 1.0), g * 0.01)
    w, m, v = np_ref.adam_keras_step(np.zeros(2), np.array([1.0, -2.0]), np.zeros(2), np.zeros(2), 1, lr=0.1)
    np.testing.assert_allclose(w, [-0.1, 0.1], rtol=1e-5)     # first step: lr * sign(g) up to eps
    state = torch_ref.TrainState({"a/kernel": np.ones((2, 2), np.float32)}, lr=0.1, l2=None, clipnorm=1.0,
                                 dtype=torch.float64)
    torch_ref.apply_gradients(state, {"a/kernel": torch.full((2, 2), 10.0, dtype=torch.float64)})
    wn, mn, vn = np_ref.adam_keras_step(np.ones((2, 2)), np_ref.clip_by_norm(np.full((2, 2), 10.0)), 0.0, 0.0, 1, lr=0.1)
    np.testing.assert_allclose(state.p["a/kernel"].detach().numpy(), wn, rtol=1e-12)


@pytest.mark.parametrize("name", ["crello_d8_l2", "rico_d16_l1"])
def test_golden_fixtures(name):
    """tests/golden/*.npz were generated by tests/golden/make_fixtures.py from the numpy-f64 oracle
    (they pin the oracle against accidental edits; they are NOT reference outputs)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    ic = _ic(meta["dataset"])
    params = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    batch = {k[len("batch:"):]: z[k] for k in z.files if k.startswith("batch:")}
    modified = {k[len("modified:"):]: z[k] for k in z.files if k.startswith("modified:")}
    masks = {k[len("mask:"):]: z[k] for k in z.files if k.startswith("mask:")}
    S = meta["S"]
    out = np_ref.model_fwd(params, ic, modified, meta["L"], maxlen=S)
    for k in out:
        np.testing.assert_allclose(out[k], z["logits:" + k], rtol=1e-9, atol=1e-10)
    lt, losses, scores, metrics = np_ref.loss_layer(ic, batch, out, masks, S)
    for k in losses:
        assert abs(losses[k] - float(z["loss:" + k])) < 1e-9 * max(1.0, abs(losses[k]))
    assert abs(np_ref.l2_loss(params, meta["l2"]) - float(z["reg_loss"])) < 1e-9
    # the torch restatement reproduces the fixture's gradients and Adam step (stored as f32)
    state = torch_ref.TrainState(params, lr=meta["lr"], l2=meta["l2"], clipnorm=1.0, dtype=torch.float64)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    tm = {k: torch.from_numpy(v) for k, v in modified.items()}
    tk = {k: torch.from_numpy(v) for k, v in masks.items()}
    info, grads = torch_ref.loss_and_grads(state, ic, tb, tm, tk, meta["L"], maxlen=S)
    for k in params:
        np.testing.assert_allclose(grads[k].numpy(), z["grad:" + k], rtol=2e-6, atol=1e-9)
    torch_ref.apply_gradients(state, grads)
    for k in params:
        np.testing.assert_allclose(state.p[k].detach().numpy(), z["adam1:" + k], rtol=2e-6, atol=1e-8)


# ------------------------------------------------------------------ RICO position-sorted loss
def test_sort_inputs_known_answer():
    """tensor_utils.py:14-44 on a hand-made document: key order (type, left, top, width, height),
    ties keep their order, padding positions go last."""
    ic = _ic("rico")
    S = 6
    col = lambda *v: np.array(v, np.int64).reshape(1, S, 1)
    inputs = {"length": np.array([[3]]),            # 4 valid positions
              "type": col(2, 1, 2, 1, 0, 0), "left": col(5, 9, 5, 9, 0, 0), "top": col(1, 0, 0, 0, 0, 0),
              "width": col(0, 3, 7, 3, 0, 0), "height": col(0, 0, 0, 0, 0, 0)}
    inputs["clickable"] = col(10, 11, 12, 13, 14, 15)
    out = np_ref.sort_inputs(inputs, {k: ic[k] for k in inputs if k != "length"} | {"length": ic["length"]})
    # priorities: p0 = (2,5,1,0,0) p1 = (1,9,0,3,0) p2 = (2,5,0,7,0) p3 = p1 -> order 1,3,2,0 then padding 4,5
    assert out["clickable"].reshape(-1).tolist() == [11, 13, 12, 10, 14, 15]
    assert out["type"].reshape(-1).tolist() == [1, 1, 2, 2, 0, 0]
    t = {k: torch.as_tensor(v) for k, v in inputs.items()}
    assert torch_ref.sort_indices(t, ic).reshape(-1).tolist() == [1, 3, 2, 0, 4, 5]


def test_sorted_loss_restatements_agree_and_are_order_free():
    ic = _ic("rico")
    B, S, D, L = 4, 9, 16, 1
    params = np_ref.init_params(ic, D, L, seed=-7)
    batch = _batch(ic, B, S, seed=3)
    nb = {k: v.numpy() for k, v in batch.items()}
    out = np_ref.model_fwd(params, ic, nb, L, maxlen=S)
    out_t = {k: torch.as_tensor(v) for k, v in out.items()}
    g = torch.Generator().manual_seed(0)
    masks = {k: torch.rand(B, S, generator=g) < 0.6 for k in out}
    nm = {k: v.numpy() for k, v in masks.items()}
    flag = np.array([True, False, True, True])
    for ignore in (None, "gt", "pred"):
        lt, losses, scores, _ = np_ref.loss_layer(ic, nb, out, nm, S, sort_flag=flag, ignore_sort=ignore)
        lt2, losses2, scores2, _ = torch_ref.loss_layer(ic, batch, out_t, masks, S, sort_flag=torch.as_tensor(flag),
                                                        ignore_sort=ignore)
        assert abs(lt - float(lt2)) < 1e-9 * abs(lt)
        for k in losses:
            assert abs(losses[k] - float(losses2[k])) < 1e-9 * max(1.0, abs(losses[k]))
            assert abs(scores[k + "_score_num"] - float(scores2[k + "_score_num"])) < 1e-9
    # no flag set == the plain loss
    lt0 = np_ref.loss_layer(ic, nb, out, nm, S)[0]
    assert np_ref.loss_layer(ic, nb, out, nm, S, sort_flag=np.zeros(B, bool))[0] == lt0
    # order-free: predictions that are a within-document shuffle of the (one-hot) targets score
    # perfectly under the sorted loss, and badly under the positional one
    rng = np.random.default_rng(0)
    seqkeys = [k for k, c in ic.items() if c.get("is_sequence") and not c.get("demo_only")]
    perm = np.stack([np.concatenate([rng.permutation(int(n) + 1), np.arange(int(n) + 1, S)])
                     for n in nb["length"].reshape(-1)])
    pred = {}
    for k in seqkeys:
        y = np.take_along_axis(nb[k], perm[:, :, None], axis=1)
        pred[k] = 30.0 * np.eye(ic[k]["input_dim"])[y]            # (B, S, N, C) confident logits
    full = {k: np.ones((B, S), bool) for k in seqkeys}
    _, _, sc_sorted, _ = np_ref.loss_layer(ic, nb, pred, full, S, sort_flag=np.ones(B, bool))
    _, _, sc_plain, _ = np_ref.loss_layer(ic, nb, pred, full, S)
    # elements that tie on the five sort keys may still swap their other attributes
    for k in ["type", "left", "top", "width", "height"]:
        assert sc_sorted[k + "_score_num"] == sc_sorted[k + "_score_den"]
    assert sc_plain["left_score_num"] < sc_plain["left_score_den"]


def test_host_sort_inputs_matches_oracle():
    from mfp.models.tensor_utils import sort_inputs
    ic = _ic("rico")
    batch = _batch(ic, 5, 8, seed=11)
    cols = np_ref.valid_columns(ic)
    want = np_ref.sort_inputs({k: v.numpy() for k, v in batch.items()}, cols, maxlen=8)
    got = sort_inputs(batch, cols)
    for k, v in want.items():
        np.testing.assert_array_equal(got[k].numpy(), v)


def test_position_token_of_shuffled_set():
    """input_dtype != "set" adds PositionEmbedding(range(S)) to every element (encoder.py:241-242)."""
    ic = _ic("rico")
    B, S, D, L = 2, 6, 16, 1
    params = np_ref.init_params(ic, D, L, seed=-2, input_dtype="shuffled_set")
    assert params["encoder/input_const/embeddings"].shape == (ic["length"]["input_dim"] + 1, D)
    batch = _batch(ic, B, S, seed=2)
    nb = {k: v.numpy() for k, v in batch.items()}
    plain = {k: v for k, v in params.items() if k != "encoder/input_const/embeddings"}
    h0, _ = np_ref.encoder_fwd(plain, ic, nb, S)
    h1, _ = np_ref.encoder_fwd(params, ic, nb, S)
    np.testing.assert_allclose(h1 - h0, np.broadcast_to(params["encoder/input_const/embeddings"][:S].astype(np.float64), (B, S, D)), atol=1e-12)
    p = torch_ref.to_torch(params, torch.float64)
    ht, _ = torch_ref.encoder_fwd(p, ic, batch, S)
    np.testing.assert_allclose(ht.detach().numpy(), h1, rtol=1e-10, atol=1e-12)


def test_context_token_numpy_vs_torch_and_known_answer():
    """context="id" / "length" (encoder.py:226-248, decoder.py:74-76): both restatements agree; a zero
    task table with zero attention output weights leaves the sequence logits what they are without it."""
    import torch
    from oracle import np_ref, torch_ref
    from mfp.data.spec import make_input_columns, synthetic_batch
    ic = make_input_columns("rico")
    for context in ("id", "length"):
        params = np_ref.init_params(ic, 16, 2, seed=-2, context=context)
        assert params["encoder/input_task/embeddings"].shape == ((5 if context == "id" else ic["length"]["input_dim"]), 16)
        b = synthetic_batch(ic, 3, 7, seed=1, ragged=True)
        b["task"] = torch.tensor([[1], [4], [0]])
        nb = {k: v.numpy() for k, v in b.items()}
        o = np_ref.model_fwd(params, ic, nb, 2, maxlen=7, context=context)
        t = torch_ref.model_fwd(torch_ref.to_torch(params, torch.float64, False), ic, b, 2, maxlen=7, context=context)
        for k in o:
            assert o[k].shape[1] == 7 and np.abs(o[k] - t[k].numpy()).max() < 1e-10, k
        # the token only talks to the elements through attention: with V = 0 in every block it is inert
        inert = dict(params)
        for i in range(2):
            inert["blocks/seq2seq_%d/attn/dense_value/kernel" % i] = np.zeros_like(inert["blocks/seq2seq_%d/attn/dense_value/kernel" % i])
            inert["blocks/seq2seq_%d/attn/dense_value/bias" % i] = np.zeros_like(inert["blocks/seq2seq_%d/attn/dense_value/bias" % i])
        with_tok = np_ref.model_fwd(inert, ic, nb, 2, maxlen=7, context=context)
        without = np_ref.model_fwd({k: v for k, v in inert.items() if "input_task" not in k}, ic, nb, 2, maxlen=7)
        for k in with_tok:
            assert np.abs(with_tok[k] - without[k]).max() < 1e-12, k


def _load_summary_fixture(name="rico_d128_l1_summary"):
    import importlib.util
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    spec = importlib.util.spec_from_file_location("make_fixtures", os.path.join(GOLDEN, "make_fixtures.py"))
    mf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mf)
    ic = _ic(meta["dataset"])
    params = np_ref.init_params(ic, meta["D"], meta["L"], seed=meta["param_seed"])
    for k, v in params.items():      # the regenerated parameters ARE the fixture's (per-variable sums)
        assert abs(float(v.astype(np.float64).sum()) - float(z["summary:" + k][0])) <= 1e-9 * max(1.0, abs(float(z["summary:" + k][0]))), k
    take = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return z, meta, ic, params, take("batch:"), take("modified:"), take("mask:"), mf.sign_vector


def test_golden_summary_fixture_at_a_size_the_device_runs():
    """tests/golden/rico_d128_l1_summary.npz (d_model 128: the smallest width the HIP path takes): parameters regenerated from
    the seed and pinned by per-variable sums, logits / losses in full, gradients and the Adam step as per-variable norms and
    projections.  The same file is read by tests/test_gpu_model.py::test_hip_f32_path_vs_committed_golden_fixture, so the
    chain committed data <-> oracle <-> device kernels closes on data that no code of this repository regenerates at test time."""
    z, meta, ic, params, batch, modified, masks, sign_vector = _load_summary_fixture()
    S, L = meta["S"], meta["L"]
    out = np_ref.model_fwd(params, ic, modified, L, maxlen=S)
    for k in out:
        np.testing.assert_allclose(out[k], z["logits:" + k], rtol=1e-9, atol=1e-10)
    lt, losses, _, _ = np_ref.loss_layer(ic, batch, out, masks, S)
    for k in losses:
        assert abs(losses[k] - float(z["loss:" + k])) < 1e-9 * max(1.0, abs(losses[k]))
    state = torch_ref.TrainState(params, lr=meta["lr"], l2=meta["l2"], clipnorm=1.0, dtype=torch.float64)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    info, grads = torch_ref.loss_and_grads(state, ic, t(batch), t(modified), t(masks), L, maxlen=S)
    before = {k: v.detach().clone() for k, v in state.p.items()}
    torch_ref.apply_gradients(state, grads)
    for k in params:
        g = grads[k].numpy().astype(np.float64).reshape(-1)
        d = (state.p[k].detach() - before[k]).numpy().astype(np.float64).reshape(-1)
        sv = sign_vector(k, g.size)
        want = z["summary:" + k]
        got = np.array([np.linalg.norm(g), g @ sv, np.linalg.norm(d), d @ sv])
        np.testing.assert_allclose(got, want[1:], rtol=1e-7, atol=1e-10 * max(1.0, float(np.abs(want[1:]).max())), err_msg=k)
