"""CPU tests of the host-side logic that mirrors the reference API: schema, masking, task mix,
argument surface, parameter layout / Keras-named state dict."""
import json

import numpy as np
import pytest
import torch

from mfp.data.spec import (ATTRIBUTE_GROUPS, DataSpec, get_attribute_groups, get_dataset_name,
                           get_valid_input_columns, make_input_columns, synthetic_batch)
from mfp.models import masking
from mfp.models.architecture.mask import get_seq_mask
from mfp.models.params import ModelLayout


def test_input_columns_contract():
    ic = make_input_columns("crello")
    assert list(ic)[:2] == ["id", "length"] and ic["id"] == {"demo_only": True, "shape": (1,), "is_sequence": False,
                                                           "primary_label": None}
    assert ic["left"] == {"type": "categorical", "input_dim": 64, "shape": (1,), "is_sequence": True,
                          "primary_label": None}
    assert ic["color"]["shape"] == (3,) and ic["color"]["input_dim"] == 16
    assert ic["image_embedding"]["type"] == "numerical" and ic["image_embedding"]["shape"] == (512,)
    assert ic["length"]["input_dim"] == 50
    cond = ic["text_embedding"]["loss_condition"]
    assert cond["key"] == "type" and sum(cond["mask"]) == 1
    assert sum(ic["image_embedding"]["loss_condition"]["mask"]) == 3
    valid = get_valid_input_columns(ic)
    assert list(valid) == ["type", "left", "top", "width", "height", "opacity", "color", "image_embedding",
                           "text_embedding", "font_family"]
    assert get_dataset_name(ic.keys()) == "crello" and get_dataset_name(make_input_columns("rico").keys()) == "rico"
    assert get_attribute_groups(ic.keys()) == ATTRIBUTE_GROUPS["crello"]
    assert len(ic) == 18                       # total_score divisor (metrics.py:298)
    rico = make_input_columns("rico")
    assert rico["clickable"]["input_dim"] == 2 and "loss_condition" not in rico["icon"]


def test_synthetic_batch_layout():
    ic = make_input_columns("crello")
    b = synthetic_batch(ic, 4, 9, seed=0, ragged=True)
    assert b["length"].shape == (4, 1) and b["length"].dtype == torch.int32 and int(b["length"].max()) == 8
    assert b["color"].shape == (4, 9, 3) and b["color"].dtype == torch.int32
    assert b["image_embedding"].shape == (4, 9, 512) and b["image_embedding"].dtype == torch.float32
    mask = get_seq_mask(b["length"], maxlen=9)
    assert (b["left"][~mask] == 0).all() and (b["image_embedding"][~mask] == 0).all()   # zero padding
    assert (b["type"][mask] >= 1).all()
    ds = DataSpec("rico", "synthetic:12:16", batch_size=8)
    batches = list(ds.make_dataset("train"))
    assert len(batches) == ds.steps_per_epoch("train") == 2 and batches[0]["left"].shape == (8, 12, 1)


def test_seq_mask_and_apply_token():
    m = get_seq_mask(torch.tensor([[0], [2]]))
    assert m.tolist() == [[True, False, False], [True, True, True]]
    col = {"type": "categorical", "input_dim": 5}
    x = torch.tensor([[[1], [2], [3]]], dtype=torch.int32)
    sel = torch.tensor([[True, False, True]])
    assert masking.apply_token(x, col, sel, "masked")[0, :, 0].tolist() == [5, 2, 5]
    assert masking.apply_token(x, col, sel, "unused")[0, :, 0].tolist() == [6, 2, 6]
    r = masking.apply_token(x, col, sel, "random")
    assert r[0, 1, 0] == 2 and 0 <= int(r[0, 0, 0]) < 5
    num = {"type": "numerical"}
    xf = torch.ones(1, 3, 4)
    assert (masking.apply_token(xf, num, sel, "masked")[0, 0] == 10.0).all()
    assert (masking.apply_token(xf, num, sel, "unused")[0, 2] == 0.0).all()


def test_filter_padding_and_task_masking():
    ic = {k: v for k, v in make_input_columns("crello").items() if not v.get("demo_only")}
    B, S = 16, 12
    batch = synthetic_batch(ic, B, S, seed=2, ragged=True)
    seq_mask = get_seq_mask(batch["length"], maxlen=S)
    f = masking.filter_padding(batch, ic, seq_mask)
    assert (f["left"][~seq_mask] == 65).all()                       # <UNUSED> = C + 1
    text_idx = [i for i, fl in enumerate(ic["text_embedding"]["loss_condition"]["mask"]) if fl][0]
    not_text = (batch["type"][..., 0] != text_idx) | ~seq_mask
    assert (f["text_embedding"][not_text] == 0.0).all() and (f["font_family"][not_text] == 36).all()
    assert torch.equal(f["text_embedding"][~not_text], batch["text_embedding"][~not_text])
    # attribute-group task: every valid position of the group's keys masked, nothing else
    mod, masks = masking.feat_masking(f, ic, seq_mask, ["left", "top", "width", "height"])
    assert (mod["left"][seq_mask] == 64).all() and torch.equal(mod["type"], f["type"])
    assert torch.equal(masks["left"], seq_mask) and not masks["type"].any()
    # element task: one valid element per document, all attributes
    mod, masks = masking.elem_masking(f, ic, seq_mask)
    assert (masks["left"].sum(1) == 1).all() and (masks["left"] & ~seq_mask).sum() == 0
    assert torch.equal(masks["left"], masks["image_embedding"])
    assert (mod["image_embedding"][masks["left"]] == 10.0).all()
    # random task: rate and composition (masking.py:11-15)
    g = torch.Generator().manual_seed(0)
    big = synthetic_batch(ic, 256, 64, seed=3)
    sm = get_seq_mask(big["length"], maxlen=64)
    fb = masking.filter_padding(big, ic, sm)
    mod, masks = masking.random_masking(fb, ic, sm, generator=g)
    frac = masks["left"].float().mean().item()
    assert abs(frac - 0.15) < 0.01
    is_mask_tok = (mod["left"][..., 0] == 64) & masks["left"]
    assert abs(is_mask_tok.sum().item() / masks["left"].sum().item() - 0.8) < 0.03
    assert masking.get_task_names(ic) == ["random", "elem", "type", "pos", "attr", "img", "txt"]
    # eval.py's call signature works here (raises TypeError in the reference)
    masking.random_masking(fb, ic, sm, replace_prob=0.0, unchange_prob=0.0)


def test_args_surface():
    from mfp.args import TrainArgs
    a = TrainArgs().parse_args(["--dataset_name", "crello", "--job-dir", "/tmp/j"])
    d = vars(a)
    expect = dict(dataset_name="crello", data_dir=None, weights=None, latent_dim=256, num_blocks=4,
                  arch_type="oneshot", block_type="deepsvg", l2=1e-2, dropout=0.1, masking_method="random",
                  seq_type="default", log_level="INFO", verbose=2, seed=0, mult=1.0, context=None,
                  input_dtype="set", batch_size=256, job_dir="/tmp/j", num_epochs=500, learning_rate=1e-4,
                  enable_profile=False, validation_freq=10)
    for k, v in expect.items():
        assert d[k] == v, k
    json.dumps(d)
    with pytest.raises(SystemExit):
        TrainArgs().parse_args(["--dataset_name", "nope", "--job-dir", "x"])


def test_param_layout_and_state_dict_roundtrip():
    from oracle import np_ref
    ic = make_input_columns("crello")
    L = ModelLayout(ic, 256, 4)
    shapes = np_ref.param_shapes(ic, 256, 4)
    assert L.U == 1378 and L.Upad == 1384 and L.table_rows == 342 and len(L.idx_cols) == 12
    real = [s for s in L.segments.values() if "/_pad/" not in s.name]
    assert sum(s.size for s in real) == 2812770 == sum(int(np.prod(s)) for s in shapes.values())
    for s in real:                                    # Keras name + Keras shape (kernels stored transposed)
        keras_shape = tuple(reversed(s.shape)) if s.transposed else s.shape
        assert shapes[s.name] == keras_shape, s.name
    offs = L.seg_offsets()
    assert offs == sorted(offs) and offs[-1] == L.numel and L.numel % 8 == 0
    # fused QKV rows and concatenated heads are contiguous per variable
    q = L.segments["blocks/seq2seq_0/attn/dense_query/kernel"]
    k = L.segments["blocks/seq2seq_0/attn/dense_key/kernel"]
    assert k.offset == q.offset + 256 * 256
    # gamma | beta of a LayerNorm layer are neighbours, 16-byte aligned: ONE pointer serves mfp_wgrad_job::n_affine (the x-hat
    # stash: the Q|K|V / FFN1 weight gradients are corrected by gamma[n] (.) + beta[n] colsum[m] in the split-K reduction)
    for D_, L_ in ((256, 4), (512, 8)):
        LL = ModelLayout(ic, D_, L_)
        for i in range(L_):
            for ln in ("norm1", "norm2"):
                g = LL.segments["blocks/seq2seq_%d/%s/gamma" % (i, ln)]
                b = LL.segments["blocks/seq2seq_%d/%s/beta" % (i, ln)]
                assert b.offset == g.offset + D_ and g.offset % 4 == 0 and g.size == b.size == D_
    from mfp.models.params import ParamStore
    st = ParamStore(ModelLayout(ic, 64, 1), "cpu", torch.float32, l2=1e-2, seed=3)
    params = np_ref.init_params(ic, 64, 1, seed=-5)
    st.load_state_dict(params)
    back = st.state_dict()
    for name, v in params.items():
        np.testing.assert_array_equal(back[name].numpy(), v)
    assert float(st.seg_l2.max()) == pytest.approx(1e-2) and float(st.seg_l2.min()) == 0.0   # LN gamma/beta unregularised
    # the coefficient table comes from the reference-shaped regulariser options (architecture/utils.py:8-22): Dense
    # kernels and biases and embedding tables are regularised, LayerNormalization variables and pad rows are not
    from mfp.models.architecture.utils import make_dense_options, make_emb_options, variable_l2
    assert make_dense_options(None) == {} and make_emb_options(None) == {}
    assert make_dense_options(0.5) == {"kernel": 0.5, "bias": 0.5} and make_emb_options(0.5) == {"embeddings": 0.5}
    for seg, c in zip(st.layout.segments.values(), st.seg_l2.tolist()):
        kind = seg.name.rsplit("/", 1)[-1]
        want = 1e-2 if (seg.l2 and kind in ("kernel", "bias", "embeddings")) else 0.0
        assert c == pytest.approx(want), seg.name
        assert variable_l2(seg.name, None) == 0.0


def test_task_probabilities():
    from mfp.models.mfp import get_task_probs
    names = ["random", "elem", "type", "pos", "attr", "img", "txt"]
    assert get_task_probs(names, "random") == [1, 0, 0, 0, 0, 0, 0]
    p = get_task_probs(names, "elem_pos_attr_img_txt")
    assert p[0] == 0 and p[2] == 0 and abs(sum(p) - 1) < 1e-12 and p[1] == pytest.approx(0.2)


def test_fused_path_hint():
    """mfp.train prints one line when a run falls off the fused kernels, and names the flag that puts it back."""
    from mfp.train import fused_path_hint
    assert fused_path_hint("bf16", 256, 128, 256) is None
    assert fused_path_hint("bf16", 256, 64, 256) is None
    assert fused_path_hint("bf16", 512, None, 64) is None
    assert fused_path_hint("fp32", 256, None, 8) is None
    assert "--seq_len 64" in fused_path_hint("bf16", 256, None, 256)
    assert "even" in fused_path_hint("bf16", 256, 64, 3)
    assert "latent_dim 128" in fused_path_hint("bf16", 128, 32, 8)
    assert fused_path_hint("bf16", 256, (64, 128), 256) is None and "odd" in fused_path_hint("bf16", 256, (64, 128), 7)


def test_default_seq_len_pads_onto_the_document_tiles(tmp_path):
    """The reference's own command line (no --seq_len; bin/train_mfp.sh:16-20) must land on the document-tile kernels: unset ->
    buckets (64, 128) on the bf16 / fp8 path at d_model 256, --seq_len 0 -> the reference's per-batch lengths
    (src/mfp/mfp/data/spec.py:255-276), an explicit value -> itself; the TFRecord reader pads a batch to the smallest bucket that
    holds its longest document and leaves longer batches alone."""
    import torch
    from mfp.data.spec import DataSpec, write_synthetic_tfrecords
    from mfp.train import default_seq_len
    assert default_seq_len("bf16", 256, None) == (64, 128) and default_seq_len("fp8", 256, None) == (64, 128)
    assert default_seq_len("fp32", 256, None) is None and default_seq_len("bf16", 128, None) is None
    assert default_seq_len("bf16", 512, None) is None      # (csrc/block_d512.hip takes any sequence length)
    assert default_seq_len("bf16", 256, 0) is None and default_seq_len("bf16", 256, 50) == 50
    data = str(tmp_path / "crello")
    write_synthetic_tfrecords(data, "crello", {"train": 12, "val": 4, "test": 4}, seq_len=9, seed=3)
    ragged = next(iter(DataSpec("crello", data, batch_size=4).make_dataset("train")))
    padded = next(iter(DataSpec("crello", data, batch_size=4, seq_len=(64, 128)).make_dataset("train")))
    S0 = ragged["left"].shape[1]
    assert S0 <= 9 and padded["left"].shape[1] == 64 and torch.equal(padded["length"], ragged["length"])
    for k, v in ragged.items():
        if v.dim() >= 2 and v.shape[1] == S0 and k != "length":
            assert torch.equal(padded[k][:, :S0], v), k
            assert not padded[k][:, S0:].any(), k      # zero padding, as the batched TF parser pads
    # a batch longer than every bucket keeps its own length
    v0 = next(iter(DataSpec("crello", data, batch_size=4).make_dataset("val")))["left"].shape[1]
    assert v0 > 2 and next(iter(DataSpec("crello", data, batch_size=4, seq_len=(2,)).make_dataset("val")))["left"].shape[1] == v0


def test_half_tile_routing_predicates(monkeypatch):
    """When the kernels run on HALF tiles (mfp/hip/functions.py): the one-launch block forward when two workgroups per 128-row tile
    still fit the chip in one round -- half a document at S = 128, ONE document at S = 64 (the reference's default batch of 256
    at --seq_len 64 = 128 tiles on 256 CUs) -- the MLP half's backward likewise, and the switches that force either way; the
    64-row activation-stationary workgroups follow MFP_FUSED_HALF as csrc/block_fused.hip's half_mode() does."""
    from types import SimpleNamespace
    from mfp.hip import functions, ops
    monkeypatch.setattr(ops, "cu_count", lambda device=None: 256)
    ctx = SimpleNamespace(store=SimpleNamespace(w=SimpleNamespace(device=None)))
    monkeypatch.setattr(functions, "BLOCK_HALF", "")
    monkeypatch.setattr(functions, "MLP_BWD_HALF", "")
    assert functions._block_half_on(ctx, 128, 128) and not functions._block_half_on(ctx, 129, 128)      # c4 | one document more
    assert not functions._block_half_on(ctx, 256, 128)                                                   # c2: the chip is full
    assert functions._block_half_on(ctx, 256, 64) and not functions._block_half_on(ctx, 512, 64)         # default batch | c2 at S = 64
    assert not functions._block_half_on(ctx, 8, 32)                                                      # no document tile at S = 32
    assert functions._mlp_bwd_half_on(ctx, 128 * 128) and not functions._mlp_bwd_half_on(ctx, 256 * 128)
    monkeypatch.setattr(functions, "BLOCK_HALF", "0")
    assert not functions._block_half_on(ctx, 4, 128)
    monkeypatch.setattr(functions, "BLOCK_HALF", "1")
    assert functions._block_half_on(ctx, 256, 128) and not functions._block_half_on(ctx, 8, 32)
    monkeypatch.delenv("MFP_FUSED_HALF", raising=False)
    assert ops.fused_half_mode(128 * 128) and not ops.fused_half_mode(256 * 128)
    monkeypatch.setenv("MFP_FUSED_HALF", "0")
    assert not ops.fused_half_mode(128)
    monkeypatch.setenv("MFP_FUSED_HALF", "1")
    assert ops.fused_half_mode(1 << 20)
