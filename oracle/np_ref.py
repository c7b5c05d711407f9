"""ORACLE (test infrastructure, NOT product code) -- numpy float64 restatement of the MFP hot path.

PARITY UNPINNED: the reference (CyberAgentAILab/flex-dm, TensorFlow 2.8 / Keras) holds no
tests, golden vectors or fixtures for this path and TensorFlow cannot be
imported in the build container, so this restatement could not be checked against outputs of
the reference itself.  It is pinned instead by (1) agreement with the independently written
torch restatement ``oracle/torch_ref.py`` (tests/test_oracle.py), (2) hand-derived
known-answer tests, (3) finite-difference checks of the gradients.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``flex-dm_amd/``) never does.

The arithmetic of the path lives in un-vendored third-party TensorFlow/Keras (pip
``tensorflow-gpu``, unpinned in the reference's requirements.txt:1; README.md:10 says 2.8).
Keras semantics assumed here [TF-EXT], from TF/Keras 2.8 behaviour:

* Dense: ``y = x @ K + b`` with ``K`` of shape ``(in, out)``.
* Embedding: row gather; float indices cast to int (relied on at encoder.py:167-172).
* LayerNormalization(): last axis, eps = 1e-3, biased variance, gamma/beta.
* Dropout(r), training: ``x * keep / (1 - r)``; eval: identity.
* softmax over the last axis; the ``-1e9`` key mask is additive (transformer.py:73).
* ``sparse_categorical_crossentropy(y, probs)`` (eager): ``clip(p, 1e-7, 1-1e-7)`` -> ``log``
  -> ``sparse_softmax_cross_entropy_with_logits`` (renormalises: subtracts ``log sum p~``).
* ``mean_squared_error``: mean over the last axis; ``cosine_similarity``:
  ``-sum(l2n(a) * l2n(b))`` with ``l2n(x) = x * rsqrt(max(sum x^2, 1e-12))``.
* ``regularizers.l2(l)``: ``l * sum(w^2)`` on every kernel, bias and embedding table.
* Adam (Keras OptimizerV2): ``lr_t = lr*sqrt(1-b2^t)/(1-b1^t)``; ``m = b1 m + (1-b1) g``;
  ``v = b2 v + (1-b2) g^2``; ``w -= lr_t * m / (sqrt(v) + eps)``; eps = 1e-7.
* ``clipnorm``: per-variable ``g * c / max(||g||, c)``.

Every function cites the reference lines it follows (paths relative to
``/root/reference/src/mfp/mfp/``).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

MASK_VALUE = 10.0  # models/masking.py:8
NULL_VALUE = 0.0   # models/masking.py:9
NUM_HEADS = 8      # models/architecture/transformer.py:147 (never overridden by Blocks)
LN_EPS = 1e-3      # [TF-EXT] keras LayerNormalization default epsilon
F64 = np.float64


# ----------------------------------------------------------------------------- schema helpers
def valid_columns(input_columns: Dict) -> Dict:
    """data/spec.py:391-403 with use_canvas=False."""
    out = {}
    for key, col in input_columns.items():
        if key == "length" or col.get("demo_only", False) or not col["is_sequence"]:
            continue
        out[key] = col
    return out


def context_rows(input_columns: Dict, context: Optional[str]) -> int:
    """Rows of the context-token table ``input_task`` (architecture/encoder.py:96-110): the number of
    tasks for context="id" (models/masking.py:18-21: random, elem + the attribute groups), the
    ``length`` vocabulary for context="length" (both layers are NAMED input_task in the reference)."""
    if context == "id":
        n_groups = 3 if "clickable" in input_columns else 5     # data/spec.py:364-391
        return 2 + n_groups
    if context == "length":
        return int(input_columns["length"]["input_dim"])
    assert context is None, "context=%r: canvas contexts are not on the MFP path" % context
    return 0


def param_shapes(input_columns: Dict, latent_dim: int, num_blocks: int,
                 input_dtype: str = "set", context: Optional[str] = None) -> Dict[str, Tuple[int, ...]]:
    """Variables created by Encoder/Blocks/Decoder, in creation order.

    encoder.py:72-92 (Embedding(C+2, D) per categorical; Embedding(2, D) + Dense(D) per
    numerical); transformer.py:43-57,161-173 (4 attention Dense, 2 MLP Dense, 2 LN);
    decoder.py:33-43 (Dense(N*C) or Dense(shape[-1])).
    """
    D = latent_dim
    shapes: Dict[str, Tuple[int, ...]] = {}
    cols = valid_columns(input_columns)
    for key, col in cols.items():
        if col["type"] == "categorical":
            shapes["encoder/input_%s/embeddings" % key] = (col["input_dim"] + 2, D)
        else:
            shapes["encoder/input_%s_special/embeddings" % key] = (2, D)
            shapes["encoder/input_%s/kernel" % key] = (col["shape"][-1], D)
            shapes["encoder/input_%s/bias" % key] = (D,)
    if input_dtype != "set":   # encoder.py:47-55: PositionEmbedding(maxlen = length.input_dim) -> maxlen + 1 rows
        shapes["encoder/input_const/embeddings"] = (int(input_columns["length"]["input_dim"]) + 1, D)
    if context is not None:    # encoder.py:96-110
        shapes["encoder/input_task/embeddings"] = (context_rows(input_columns, context), D)
    for i in range(num_blocks):
        p = "blocks/seq2seq_%d/" % i
        for name in ("dense_query", "dense_key", "dense_value", "combine_heads"):
            shapes[p + "attn/%s/kernel" % name] = (D, D)
            shapes[p + "attn/%s/bias" % name] = (D,)
        shapes[p + "mlp/dense_0/kernel"] = (D, 2 * D)
        shapes[p + "mlp/dense_0/bias"] = (2 * D,)
        shapes[p + "mlp/dense_1/kernel"] = (2 * D, D)
        shapes[p + "mlp/dense_1/bias"] = (D,)
        for n in ("norm1", "norm2"):
            shapes[p + n + "/gamma"] = (D,)
            shapes[p + n + "/beta"] = (D,)
    for key, col in cols.items():
        units = col["shape"][-1] * col["input_dim"] if col["type"] == "categorical" \
            else col["shape"][-1]
        shapes["decoder/decoder_%s/kernel" % key] = (D, units)
        shapes["decoder/decoder_%s/bias" % key] = (units,)
    return shapes


def is_regularized(name: str) -> bool:
    """architecture/utils.py:8-22: L2 on kernels, biases, embeddings; LN gamma/beta are
    created without options (transformer.py:172-173)."""
    return not (name.endswith("/gamma") or name.endswith("/beta"))


def init_params(input_columns: Dict, latent_dim: int, num_blocks: int, seed: int = 0, input_dtype: str = "set",
                context: Optional[str] = None) -> Dict[str, np.ndarray]:
    """[TF-EXT] Keras default initialisers: Dense glorot_uniform / zeros bias; Embedding
    U(-0.05, 0.05); LN gamma 1, beta 0.  (Biases are drawn small-random instead of zero when
    ``seed`` is negative so that tests exercise the bias paths.)"""
    rng = np.random.default_rng(abs(seed))
    params = {}
    for name, shape in param_shapes(input_columns, latent_dim, num_blocks, input_dtype, context).items():
        if name.endswith("/embeddings"):
            w = rng.uniform(-0.05, 0.05, size=shape)
        elif name.endswith("/kernel"):
            limit = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rng.uniform(-limit, limit, size=shape)
        elif name.endswith("/gamma"):
            w = np.ones(shape) if seed >= 0 else 1.0 + 0.1 * rng.standard_normal(shape)
        else:  # bias, beta
            w = np.zeros(shape) if seed >= 0 else 0.05 * rng.standard_normal(shape)
        params[name] = w.astype(np.float32)
    return params


# ----------------------------------------------------------------------------- primitives
def get_seq_mask(length: np.ndarray, maxlen: Optional[int] = None) -> np.ndarray:
    """architecture/mask.py:21-33: sequence_mask(reshape(length,-1)+1, maxlen)."""
    length = np.asarray(length).reshape(-1).astype(np.int64) + 1
    if maxlen is None:
        maxlen = int(length.max())
    return np.arange(maxlen)[None, :] < length[:, None]


def dense(x, kernel, bias):
    return x @ kernel.astype(F64) + bias.astype(F64)


def layer_norm(x, gamma, beta, eps=LN_EPS):
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * gamma.astype(F64) + beta.astype(F64)


def softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def dropout(x, rate, keep_mask):
    """[TF-EXT] tf.nn.dropout: x * keep / (1 - rate); ``keep_mask`` None == eval mode."""
    if keep_mask is None or rate == 0.0:
        return x
    return x * keep_mask.astype(F64) / (1.0 - rate)


# ----------------------------------------------------------------------------- model
def context_ids(inputs, context):
    """The index of the context token (encoder.py:231-240): the task id or the (zero-based) length."""
    v = np.asarray(inputs["task" if context == "id" else "length"])
    return (v[:, 0] if v.ndim == 2 else v).astype(np.int64)


def encoder_fwd(params, input_columns, inputs, maxlen=None, context=None):
    """architecture/encoder.py:147-199 (fusion="add"); context in {None, "id", "length"} (:226-248: a
    token looked up in ``input_task`` is PREPENDED and the sequence mask grows by one)."""
    seq_mask = get_seq_mask(inputs["length"], maxlen)
    seq = 0.0
    for key, col in valid_columns(input_columns).items():
        if col["type"] == "categorical":
            table = params["encoder/input_%s/embeddings" % key].astype(F64)
            x = table[np.asarray(inputs[key]).astype(np.int64)]  # (B,S,N,D)  :157
            x = x.sum(axis=2)                                     # :160
        else:
            xin = np.asarray(inputs[key]).astype(F64)
            is_masked = np.all(xin == MASK_VALUE, axis=2)         # :165
            is_unused = np.all(xin == NULL_VALUE, axis=2)         # :166
            special = params["encoder/input_%s_special/embeddings" % key].astype(F64)
            x = dense(xin, params["encoder/input_%s/kernel" % key],
                      params["encoder/input_%s/bias" % key])      # :173
            x = np.where(is_masked[..., None], special[0], x)     # :174
            x = np.where(is_unused[..., None], special[1], x)     # :175 (unused wins)
        seq = seq + x                                             # :195-197
    if "encoder/input_const/embeddings" in params:                # :241-242, transformer.py:24-30 (dropout 0)
        S = seq.shape[1]
        seq = seq + params["encoder/input_const/embeddings"].astype(F64)[np.arange(S)][None]
    if context is not None:                                       # :231-248
        canvas = params["encoder/input_task/embeddings"].astype(F64)[context_ids(inputs, context)]
        seq = np.concatenate([canvas[:, None, :], seq], axis=1)
        seq_mask = get_seq_mask(np.asarray(inputs["length"]) + 1, None if maxlen is None else maxlen + 1)
    return seq, seq_mask


def attention_fwd(params, prefix, x, seq_mask):
    """architecture/transformer.py:60-99."""
    B, S, D = x.shape
    H = NUM_HEADS
    hd = D // H

    def heads(name):
        y = dense(x, params[prefix + "attn/%s/kernel" % name], params[prefix + "attn/%s/bias" % name])
        return y.reshape(B, S, H, hd).transpose(0, 2, 1, 3)      # :78-80

    q, k, v = heads("dense_query"), heads("dense_key"), heads("dense_value")
    score = q @ k.transpose(0, 1, 3, 2)                           # :61
    scaled = score / np.sqrt(float(hd))                           # :63
    m = seq_mask.astype(F64)[:, None, None, :]
    scaled = scaled + -1e9 * (1.0 - m)                            # :73
    w = softmax(scaled, axis=-1)                                  # :74
    out = w @ v                                                   # :75
    out = out.transpose(0, 2, 1, 3).reshape(B, S, D)              # :92-97
    return dense(out, params[prefix + "attn/combine_heads/kernel"],
                 params[prefix + "attn/combine_heads/bias"])      # :98


def block_fwd(params, i, x, seq_mask, rate=0.0, keep1=None, keep2=None):
    """DeepSVGBlock.call, architecture/transformer.py:211-229 (conditional=None, no pooling)."""
    p = "blocks/seq2seq_%d/" % i
    y = layer_norm(x, params[p + "norm1/gamma"], params[p + "norm1/beta"])   # :216
    y = attention_fwd(params, p, y, seq_mask)                                  # :217
    y = dropout(y, rate, keep1)                                                # :218
    x = x + y                                                                  # :219
    y = layer_norm(x, params[p + "norm2/gamma"], params[p + "norm2/beta"])   # :222
    y = dense(y, params[p + "mlp/dense_0/kernel"], params[p + "mlp/dense_0/bias"])
    y = np.maximum(y, 0.0)                                                     # :163-166 relu
    y = dense(y, params[p + "mlp/dense_1/kernel"], params[p + "mlp/dense_1/bias"])
    y = dropout(y, rate, keep2)                                                # :224
    return x + y                                                               # :225


def decoder_fwd(params, input_columns, h, context=None):
    """architecture/decoder.py:72-111 (detachment="default"); with a context token the first position
    is split off (:74-76) and, the canvas heads being off this path, dropped."""
    if context is not None:
        h = h[:, 1:]
    B, S, _ = h.shape
    outputs = {}
    for key, col in valid_columns(input_columns).items():
        y = dense(h, params["decoder/decoder_%s/kernel" % key], params["decoder/decoder_%s/bias" % key])
        if col["type"] == "categorical":
            outputs[key] = y.reshape(B, S, col["shape"][-1], col["input_dim"])
        else:
            outputs[key] = y.reshape(B, S, col["shape"][-1])
    return outputs


def model_fwd(params, input_columns, inputs, num_blocks, rate=0.0, keep_masks=None, maxlen=None, context=None):
    """_OneShot.call, models/model.py:26-30."""
    h, seq_mask = encoder_fwd(params, input_columns, inputs, maxlen, context)
    for i in range(num_blocks):
        k1 = keep_masks[(i, 1)] if keep_masks else None
        k2 = keep_masks[(i, 2)] if keep_masks else None
        h = block_fwd(params, i, h, seq_mask, rate, k1, k2)
    return decoder_fwd(params, input_columns, h, context)


# ----------------------------------------------------------------------------- losses
def categorical_metric(y_true, logits):
    """models/metrics.py:36-49 + [TF-EXT] keras sparse_categorical_crossentropy from probs."""
    p = softmax(logits, axis=-1)
    pred = p.argmax(axis=-1)
    pc = np.clip(p, 1e-7, 1.0 - 1e-7)
    logp = np.log(pc)
    lse = np.log(np.exp(logp).sum(axis=-1))
    picked = np.take_along_axis(logp, y_true[..., None].astype(np.int64), axis=-1)[..., 0]
    loss = -(picked - lse)
    score = (y_true == pred).astype(F64)
    return loss, score


def continuous_metric(y_true, y_pred):
    """models/metrics.py:52-57."""
    loss = ((y_true - y_pred) ** 2).mean(axis=-1)

    def l2n(x):
        return x / np.sqrt(np.maximum((x ** 2).sum(axis=-1, keepdims=True), 1e-12))

    cos = -(l2n(y_true) * l2n(y_pred)).sum(axis=-1)
    score = -0.5 * cos + 0.5
    return loss, score


SORT_KEYS = ["type", "left", "top", "width", "height"]   # models/tensor_utils.py:11


def sort_inputs(inputs, input_columns, from_logits=False, maxlen=None):
    """sort_inputs, models/tensor_utils.py:14-44: reorder every sequence tensor of ``inputs`` by
    the lexicographic key (type, left, top, width, height), padding positions last.

    [TF-EXT] ``tf.argsort`` (ascending, stable=False) is ``top_k`` of the negated values, which
    lists equal values lowest index first, i.e. the order of a stable sort; ``tf.argmax``
    returns the first maximal index.
    """
    CONST = 100                                                                    # :15
    data = {}
    for key, col in input_columns.items():                                         # :24-28
        if key not in inputs:
            continue
        v = np.asarray(inputs[key])
        if col["is_sequence"] and col["type"] == "categorical":
            if from_logits:
                v = v.argmax(axis=-1)
            v = v.astype(np.int64)
        data[key] = v
    for key in SORT_KEYS:
        assert input_columns[key]["input_dim"] < CONST                             # :21
    S = data[SORT_KEYS[0]].shape[1]
    invalid = ~get_seq_mask(inputs["length"], maxlen or S)                         # :30
    priority = np.zeros(data[SORT_KEYS[0]].shape[:2], np.int64)
    for key in SORT_KEYS:                                                          # :32-33
        priority = priority * CONST + data[key][..., 0]
    priority = priority + invalid.astype(np.int64) * CONST ** len(SORT_KEYS)       # :34
    indices = np.argsort(priority, axis=-1, kind="stable")                         # :35
    out = {}
    for key, val in inputs.items():                                                # :37-43
        val = np.asarray(val)
        if key in input_columns and input_columns[key]["is_sequence"]:
            idx = indices.reshape(indices.shape + (1,) * (val.ndim - 2))
            out[key] = np.take_along_axis(val, idx, axis=1)
        else:
            out[key] = val
    return out


def sorted_loss_inputs(input_columns, y_true, y_pred, sort_flag, ignore_sort=None, maxlen=None):
    """The ``sort_flag`` prologue of LossLayer.call, models/metrics.py:180-211: documents whose
    flag is set have targets and predictions re-ordered independently (targets by their labels,
    predictions by their own argmax), the others are left as they are.  ``mfp_masks`` are NOT
    re-ordered (:251 uses them as they came in)."""
    assert ignore_sort in ("gt", "pred", None)                                     # :181
    cols = valid_columns(input_columns)
    t_sort = y_true if ignore_sort == "gt" else sort_inputs(y_true, cols, maxlen=maxlen)
    y_pred = dict(y_pred)
    y_pred["length"] = y_true["length"]                                            # :188
    p_sort = y_pred if ignore_sort == "pred" else sort_inputs(y_pred, cols, from_logits=True, maxlen=maxlen)
    flag = np.asarray(sort_flag).astype(bool)
    yt, yp = {}, {}
    for key in y_true.keys():                                                      # :196-211
        col = input_columns[key]
        if col.get("demo_only", False):
            continue
        if col["is_sequence"]:
            f = flag[:, None, None]
            yt[key] = np.where(f, np.asarray(t_sort[key]), np.asarray(y_true[key]))
            if col["type"] == "categorical":
                f = f[:, None]
            yp[key] = np.where(f, np.asarray(p_sort[key]), np.asarray(y_pred[key]))
        else:
            yt[key] = y_true[key]
            if key in y_pred:
                yp[key] = y_pred[key]
    return yt, yp


def loss_layer(input_columns, y_true, y_pred, mfp_masks, maxlen=None, sort_flag=None, ignore_sort=None):
    """LossLayer.call, models/metrics.py:172-299 (predict_context=False); ``sort_flag`` (B,) bool
    selects the RICO position-sorted variant (:180-211).

    Returns ``(loss_total, losses{key}, scores{key_score_num/_den}, metrics{...})``.
    """
    if sort_flag is not None:
        y_true, y_pred = sorted_loss_inputs(input_columns, y_true, y_pred, sort_flag, ignore_sort, maxlen)
    seq_mask = get_seq_mask(y_true["length"], maxlen)
    losses, scores, metrics = {}, {}, {}
    score_total = 0.0
    for key, col in input_columns.items():
        if col.get("demo_only", False) or not col["is_sequence"]:
            continue
        prediction = np.asarray(y_pred[key]).astype(F64)[:, : seq_mask.shape[1]]   # :231
        if col["type"] == "categorical":
            yt = np.asarray(y_true[key]).astype(np.int64)
            assert yt.max() <= col["input_dim"] - 1 and yt.min() >= 0             # :236-237
            loss, score = categorical_metric(yt, prediction)
        else:
            loss, score = continuous_metric(np.asarray(y_true[key]).astype(F64), prediction)
            loss = loss[..., None] * float(col["shape"][-1])                       # :247-248
            score = score[..., None]
        w = np.asarray(mfp_masks[key]).astype(F64)[..., None]                     # :251
        loss = loss * w
        score = score * w
        den = np.ones_like(loss) * w
        if "loss_condition" in col:                                                # :256-261
            cond = col["loss_condition"]
            cw = np.asarray(cond["mask"])[np.asarray(y_true[cond["key"]]).astype(np.int64)]
            cw = cw.astype(F64)
            loss, score, den = loss * cw, score * cw, den * cw
        sw = seq_mask.astype(F64)[:, :, None]                                      # :263-267
        loss = (loss * sw).sum(axis=1).sum(axis=1)
        score = (score * sw).sum(axis=1).sum(axis=1)
        den = (den * sw).sum(axis=1).sum(axis=1)
        loss = loss.mean()                                                         # :277
        score, den = score.sum(), den.sum()
        normalized = 1.0 if den == 0.0 else score / den                            # :281
        score_total += normalized
        metrics[key + "_score"] = normalized
        scores[key + "_score_num"] = score
        scores[key + "_score_den"] = den
        losses[key] = loss
    loss_total = 0.0
    for key, loss in losses.items():
        metrics[key + "_loss"] = loss
        loss_total += loss
    metrics["total_score"] = score_total / len(input_columns)                      # :298
    return loss_total, losses, scores, metrics


def l2_loss(params, l2):
    """architecture/utils.py:8-22 + [TF-EXT] regularizers.l2: l2 * sum(w^2), no 1/2."""
    if l2 is None:
        return 0.0
    return sum(l2 * (w.astype(F64) ** 2).sum() for n, w in params.items() if is_regularized(n))


# ----------------------------------------------------------------------------- optimizer
def clip_by_norm(g, clip=1.0):
    """[TF-EXT] tf.clip_by_norm as Keras ``clipnorm`` applies it per variable."""
    n = np.sqrt((g.astype(F64) ** 2).sum())
    return g * clip / max(n, clip)


def adam_keras_step(w, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7):
    """[TF-EXT] Keras OptimizerV2 Adam dense update at (1-based) step ``t``;
    configured at train.py:71-77."""
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    w = w - lr_t * m / (np.sqrt(v) + eps)
    return w, m, v
