"""ORACLE (test infrastructure, NOT product code) -- eager torch restatement of the MFP hot path.

PARITY UNPINNED (see ``oracle/np_ref.py`` header: the reference has no tests/golden vectors
and TensorFlow is not importable here).  This file is the second, independently structured
restatement: op-for-op and deliberately un-fused like the reference's eager TF execution
(separate Q/K/V Dense, materialised (B,H,S,S) scores, per-attribute Python loops for
embedding / heads / losses, per-variable clip + Adam).  It provides

* autograd gradients for the parity tests of the HIP backward kernels,
* the full train step (masking -> fwd -> losses + L2 -> grads -> per-variable clipnorm ->
  Keras Adam) that ``bench.py`` times as ``cpu_baseline`` (kind "port").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
it.  Reference line citations are relative to ``/root/reference/src/mfp/mfp/``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .np_ref import (LN_EPS, MASK_VALUE, NULL_VALUE, NUM_HEADS, is_regularized,
                     valid_columns)

MASK_PROB = 0.15                       # models/masking.py:11
CHANGE_PROB = 1.0 - 0.1                # :13-14
THRESH = 0.1 / CHANGE_PROB             # :15


def to_torch(params, dtype=torch.float32, requires_grad=True):
    out = {}
    for k, v in params.items():
        t = torch.as_tensor(v).detach().clone().to(dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def get_seq_mask(length, maxlen=None):
    """architecture/mask.py:21-33."""
    length = length.reshape(-1).to(torch.int64) + 1
    if maxlen is None:
        maxlen = int(length.max())
    return torch.arange(maxlen)[None, :] < length[:, None]


def _dense(x, p, name):
    return x @ p[name + "/kernel"] + p[name + "/bias"]


def _layer_norm(x, gamma, beta):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + LN_EPS) * gamma + beta


def _dropout(x, rate, keep):
    if keep is None or rate == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - rate)


def encoder_fwd(p, input_columns, inputs, maxlen=None, context=None):
    """architecture/encoder.py:147-248 (fusion="add"; context in {None, "id", "length"})."""
    dtype = next(iter(p.values())).dtype
    seq_mask = get_seq_mask(inputs["length"], maxlen)
    data_s = []
    for key, col in valid_columns(input_columns).items():
        if col["type"] == "categorical":
            x = p["encoder/input_%s/embeddings" % key][inputs[key].to(torch.int64)]
            x = x.sum(dim=2)
        else:
            xin = inputs[key].to(dtype)
            is_masked = (xin == MASK_VALUE).all(dim=2)
            is_unused = (xin == NULL_VALUE).all(dim=2)
            special = p["encoder/input_%s_special/embeddings" % key]
            masked_emb = special[torch.zeros(seq_mask.shape, dtype=torch.int64)]
            unused_emb = special[torch.ones(seq_mask.shape, dtype=torch.int64)]
            x = _dense(xin, p, "encoder/input_%s" % key)
            x = torch.where(is_masked[..., None], masked_emb, x)
            x = torch.where(is_unused[..., None], unused_emb, x)
        data_s.append(x)
    seq = 0.0
    for d in data_s:
        seq = seq + d
    if "encoder/input_const/embeddings" in p:    # encoder.py:241-242 (PositionEmbedding, dropout rate 0)
        positions = torch.arange(seq.shape[1])
        seq = seq + p["encoder/input_const/embeddings"][positions][None, :, :]
    if context is not None:                                   # encoder.py:231-248
        ids = inputs["task" if context == "id" else "length"]
        ids = (ids[:, 0] if ids.dim() == 2 else ids).to(torch.int64)
        seq = torch.cat([p["encoder/input_task/embeddings"][ids][:, None, :], seq], dim=1)
        seq_mask = get_seq_mask(inputs["length"] + 1, None if maxlen is None else maxlen + 1)
    return seq, seq_mask


def attention_fwd(p, prefix, x, seq_mask):
    """architecture/transformer.py:60-99."""
    B, S, D = x.shape
    H, hd = NUM_HEADS, D // NUM_HEADS

    def heads(name):
        return _dense(x, p, prefix + "attn/" + name).reshape(B, S, H, hd).permute(0, 2, 1, 3)

    q, k, v = heads("dense_query"), heads("dense_key"), heads("dense_value")
    score = q @ k.transpose(-1, -2)
    scaled = score / math.sqrt(float(hd))
    m = seq_mask.to(x.dtype)[:, None, None, :]
    scaled = scaled + -1e9 * (1.0 - m)
    w = torch.softmax(scaled, dim=-1)
    out = (w @ v).permute(0, 2, 1, 3).reshape(B, S, D)
    return _dense(out, p, prefix + "attn/combine_heads")


def block_fwd(p, i, x, seq_mask, rate=0.0, keep1=None, keep2=None):
    """DeepSVGBlock.call, architecture/transformer.py:211-229."""
    pre = "blocks/seq2seq_%d/" % i
    y = _layer_norm(x, p[pre + "norm1/gamma"], p[pre + "norm1/beta"])
    y = attention_fwd(p, pre, y, seq_mask)
    y = _dropout(y, rate, keep1)
    x = x + y
    y = _layer_norm(x, p[pre + "norm2/gamma"], p[pre + "norm2/beta"])
    y = torch.relu(_dense(y, p, pre + "mlp/dense_0"))
    y = _dense(y, p, pre + "mlp/dense_1")
    y = _dropout(y, rate, keep2)
    return x + y


def decoder_fwd(p, input_columns, h, context=None):
    """architecture/decoder.py:72-111."""
    if context is not None:
        h = h[:, 1:]
    B, S, _ = h.shape
    outputs = {}
    for key, col in valid_columns(input_columns).items():
        y = _dense(h, p, "decoder/decoder_%s" % key)
        if col["type"] == "categorical":
            outputs[key] = y.reshape(B, S, col["shape"][-1], col["input_dim"])
        else:
            outputs[key] = y.reshape(B, S, col["shape"][-1])
    return outputs


def model_fwd(p, input_columns, inputs, num_blocks, rate=0.0, keep_masks=None, maxlen=None,
              return_hidden=False, context=None):
    """_OneShot.call, models/model.py:26-30."""
    h, seq_mask = encoder_fwd(p, input_columns, inputs, maxlen, context)
    hidden = [h]
    for i in range(num_blocks):
        k1 = keep_masks[(i, 1)] if keep_masks else None
        k2 = keep_masks[(i, 2)] if keep_masks else None
        h = block_fwd(p, i, h, seq_mask, rate, k1, k2)
        hidden.append(h)
    out = decoder_fwd(p, input_columns, h, context)
    return (out, hidden) if return_hidden else out


def categorical_metric(y_true, logits):
    """models/metrics.py:36-49 + [TF-EXT] CE-from-probabilities."""
    prob = torch.softmax(logits, dim=-1)
    pred = prob.argmax(dim=-1)
    logp = torch.log(torch.clamp(prob, 1e-7, 1.0 - 1e-7))
    lse = torch.logsumexp(logp, dim=-1)
    picked = torch.gather(logp, -1, y_true[..., None].to(torch.int64))[..., 0]
    return -(picked - lse), (y_true == pred).to(logits.dtype)


def continuous_metric(y_true, y_pred):
    """models/metrics.py:52-57."""
    loss = ((y_true - y_pred) ** 2).mean(dim=-1)

    def l2n(x):
        return x * torch.rsqrt(torch.clamp((x ** 2).sum(dim=-1, keepdim=True), min=1e-12))

    cos = -(l2n(y_true) * l2n(y_pred)).sum(dim=-1)
    return loss, -0.5 * cos + 0.5


SORT_KEYS = ["type", "left", "top", "width", "height"]   # models/tensor_utils.py:11


def sort_indices(inputs, input_columns, from_logits=False, maxlen=None):
    """The permutation of sort_inputs (models/tensor_utils.py:15-35), written rank-wise instead of
    with an argsort: position i goes to slot #{j : p_j < p_i or (p_j == p_i and j < i)}."""
    S = inputs[SORT_KEYS[0]].shape[1]
    prio = torch.zeros(inputs[SORT_KEYS[0]].shape[:2], dtype=torch.int64)
    for key in SORT_KEYS:
        v = inputs[key]
        v = v.argmax(dim=-1) if from_logits else v
        prio = prio * 100 + v[..., 0].to(torch.int64)
    prio = prio + (~get_seq_mask(inputs["length"], maxlen or S)).to(torch.int64) * 100 ** len(SORT_KEYS)
    pos = torch.arange(S)
    before = (prio[:, None, :] < prio[:, :, None]) | ((prio[:, None, :] == prio[:, :, None]) & (pos[None, None, :] < pos[None, :, None]))
    rank = before.sum(dim=-1)                       # slot of position i
    return torch.argsort(rank, dim=-1)              # position held by slot r (rank is a permutation)


def sorted_loss_inputs(input_columns, y_true, y_pred, sort_flag, ignore_sort=None, maxlen=None):
    """LossLayer.call's sort prologue, models/metrics.py:180-211 (differentiable in y_pred: the
    gather passes gradients back to the rows it picked)."""
    flag = torch.as_tensor(sort_flag).to(torch.bool)
    S = y_true[SORT_KEYS[0]].shape[1]
    ident = torch.arange(S)[None, :].expand(flag.shape[0], S)
    it = ident if ignore_sort == "gt" else sort_indices(y_true, input_columns, False, maxlen)
    yp_len = dict(y_pred)
    yp_len["length"] = y_true["length"]
    ip = ident if ignore_sort == "pred" else sort_indices({k: v.detach() if torch.is_tensor(v) else v for k, v in yp_len.items()},
                                                          input_columns, True, maxlen)
    it = torch.where(flag[:, None], it, ident)
    ip = torch.where(flag[:, None], ip, ident)

    def take(val, idx):
        idx = idx.reshape(idx.shape + (1,) * (val.dim() - 2)).expand(-1, -1, *val.shape[2:])
        return torch.gather(val, 1, idx)

    yt, yp = {}, {}
    for key, col in input_columns.items():
        if col.get("demo_only", False):
            continue
        if col["is_sequence"]:
            yt[key] = take(y_true[key], it)
            yp[key] = take(y_pred[key][:, :S], ip)
        else:
            if key in y_true:
                yt[key] = y_true[key]
            if key in y_pred:
                yp[key] = y_pred[key]
    yt["length"] = y_true["length"]
    return yt, yp


def loss_layer(input_columns, y_true, y_pred, mfp_masks, maxlen=None, sort_flag=None, ignore_sort=None):
    """LossLayer.call, models/metrics.py:172-299; ``sort_flag`` (B,) selects the RICO
    position-sorted variant (:180-211)."""
    if sort_flag is not None:
        y_true, y_pred = sorted_loss_inputs(input_columns, y_true, y_pred, sort_flag, ignore_sort, maxlen)
    seq_mask = get_seq_mask(y_true["length"], maxlen)
    losses, scores, metrics = {}, {}, {}
    score_total = 0.0
    for key, col in input_columns.items():
        if col.get("demo_only", False) or not col["is_sequence"]:
            continue
        prediction = y_pred[key][:, : seq_mask.shape[1]]
        dtype = prediction.dtype
        if col["type"] == "categorical":
            loss, score = categorical_metric(y_true[key].to(torch.int64), prediction)
        else:
            loss, score = continuous_metric(y_true[key].to(dtype), prediction)
            loss = loss[..., None] * float(col["shape"][-1])
            score = score[..., None]
        w = mfp_masks[key].to(dtype)[..., None]
        loss, score = loss * w, score * w
        den = torch.ones_like(loss) * w
        if "loss_condition" in col:
            cond = col["loss_condition"]
            cw = torch.tensor(cond["mask"])[y_true[cond["key"]].to(torch.int64)].to(dtype)
            loss, score, den = loss * cw, score * cw, den * cw
        sw = seq_mask.to(dtype)[:, :, None]
        loss = (loss * sw).sum(dim=1).sum(dim=1)
        score = (score * sw).sum(dim=1).sum(dim=1)
        den = (den * sw).sum(dim=1).sum(dim=1)
        loss = loss.mean()
        score, den = score.sum(), den.sum()
        normalized = torch.where(den == 0.0, torch.ones_like(den), score / den)
        score_total = score_total + normalized
        metrics[key + "_score"] = normalized
        scores[key + "_score_num"] = score
        scores[key + "_score_den"] = den
        losses[key] = loss
    loss_total = 0.0
    for key, loss in losses.items():
        metrics[key + "_loss"] = loss
        loss_total = loss_total + loss
    metrics["total_score"] = score_total / len(input_columns)
    return loss_total, losses, scores, metrics


def l2_loss(p, l2):
    """architecture/utils.py:8-22; l2 * sum(w^2) per regularised variable."""
    if l2 is None:
        return 0.0
    total = 0.0
    for name, w in p.items():
        if is_regularized(name):
            total = total + l2 * (w ** 2).sum()
    return total


# ----------------------------------------------------------------------------- masking
def _apply_token(x, col, mask, token, gen=None):
    """models/masking.py:68-95."""
    m = mask[..., None]
    if col["type"] == "categorical":
        if token == "masked":
            data = col["input_dim"]
        elif token == "unused":
            data = col["input_dim"] + 1
        else:
            data = torch.randint(0, col["input_dim"], x.shape, generator=gen).to(x.dtype)
        return torch.where(m, torch.as_tensor(data, dtype=x.dtype), x)
    if token == "masked":
        data = torch.full_like(x, MASK_VALUE)
    elif token == "unused":
        data = torch.full_like(x, NULL_VALUE)
    else:
        data = 0.1 * torch.randn(x.shape, generator=gen)
    return torch.where(m, data, x)


def filter_padding(inputs, input_columns, seq_mask):
    """models/masking.py:24-53."""
    out = {}
    for key, col in input_columns.items():
        if col.get("demo_only", False):
            continue
        if not col["is_sequence"]:
            out[key] = inputs[key]
            continue
        unused = ~seq_mask
        if "loss_condition" in col:
            cond = col["loss_condition"]
            bad = torch.tensor([not f for f in cond["mask"]])
            unused = unused | bad[inputs[cond["key"]][..., 0].to(torch.int64)]
        out[key] = _apply_token(inputs[key], col, unused, "unused")
    return out


def random_masking(inputs, input_columns, seq_mask, gen=None):
    """models/masking.py:227-269 (task "random")."""
    modified, masks = {}, {}
    for key, col in input_columns.items():
        if col.get("demo_only", False):
            continue
        if not col["is_sequence"]:
            modified[key] = inputs[key]
            continue
        shape = inputs[key].shape[:-1]
        mfp = seq_mask & (torch.rand(shape, generator=gen) < MASK_PROB)
        chg = mfp & (torch.rand(shape, generator=gen) < CHANGE_PROB)
        r = torch.rand(shape, generator=gen)
        x = _apply_token(inputs[key], col, chg & (r >= THRESH), "masked")
        x = _apply_token(x, col, chg & (r < THRESH), "random", gen)
        modified[key], masks[key] = x, mfp
    return modified, masks


# ----------------------------------------------------------------------------- train step
class TrainState:
    """Parameters + Keras-Adam slots for the CPU restatement of the train step."""

    def __init__(self, params, lr=1e-4, l2=1e-2, clipnorm=1.0, dtype=torch.float32):
        self.p = to_torch(params, dtype)
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.t = 0
        self.lr, self.l2, self.clipnorm = lr, l2, clipnorm


def loss_and_grads(state: TrainState, input_columns, targets, modified_inputs, masks,
                   num_blocks, rate=0.0, keep_masks=None, maxlen=None, sort_flag=None, context=None):
    """Keras Model.train_step with no compiled loss: total = sum(model.losses)
    = LossLayer add_loss (metrics.py:297) + every L2 regulariser (utils.py:8-22)."""
    for w in state.p.values():
        w.grad = None
    out = model_fwd(state.p, input_columns, modified_inputs, num_blocks, rate, keep_masks, maxlen, context=context)
    loss_total, losses, scores, metrics = loss_layer(input_columns, targets, out, masks, maxlen, sort_flag=sort_flag)
    reg = l2_loss(state.p, state.l2)
    total = loss_total + reg
    total.backward()
    grads = {k: w.grad.detach().clone() for k, w in state.p.items()}
    return dict(total=total.detach(), data_loss=torch.as_tensor(loss_total).detach(),
                reg_loss=torch.as_tensor(reg).detach(), losses=losses, scores=scores,
                metrics=metrics, outputs=out), grads


def apply_gradients(state: TrainState, grads, b1=0.9, b2=0.999, eps=1e-7):
    """Per-variable clipnorm then Keras Adam (train.py:71-77; [TF-EXT])."""
    state.t += 1
    t = state.t
    lr_t = state.lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    with torch.no_grad():
        for k, w in state.p.items():
            g = grads[k]
            if state.clipnorm is not None:
                n = torch.sqrt((g * g).sum())
                g = g * state.clipnorm / torch.clamp(n, min=state.clipnorm)
            state.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
            state.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            w.sub_(lr_t * state.m[k] / (torch.sqrt(state.v[k]) + eps))


def train_step(state: TrainState, input_columns, batch, num_blocks, rate=0.0, gen=None,
               maxlen=None):
    """MFP.call(training=True) for masking_method="random" (mfp.py:298-340) + optimizer."""
    S = maxlen or batch[next(k for k, c in input_columns.items()
                             if c.get("is_sequence") and not c.get("demo_only"))].shape[1]
    seq_mask = get_seq_mask(batch["length"], S)
    filtered = filter_padding(batch, input_columns, seq_mask)
    modified, masks = random_masking(filtered, input_columns, seq_mask, gen)
    modified["length"] = batch["length"]
    keep = None
    if rate > 0.0:
        B = batch["length"].shape[0]
        D = state.p["blocks/seq2seq_0/norm1/gamma"].shape[0]
        keep = {(i, j): torch.rand((B, S, D), generator=gen) >= rate
                for i in range(num_blocks) for j in (1, 2)}
    info, grads = loss_and_grads(state, input_columns, batch, modified, masks, num_blocks,
                                 rate, keep, S)
    apply_gradients(state, grads)
    return info
