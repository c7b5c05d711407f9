"""LossLayer: per-attribute masked CE / MSE losses and scores (reference models/metrics.py).

``LossLayer.call`` (metrics.py:173-299) is one fused HIP launch pair (``mfp_loss_fwd_bwd``): the
per-key loss, score numerator and denominator come back as a ``[nkeys][3]`` tensor and the
Keras metric names / definitions are reproduced from it:

* ``<key>_loss``  = mean over B of sum over (S, N) of weighted loss          (metrics.py:265-277)
* ``<key>_score`` = score_sum / den_sum, 1.0 when den == 0                    (metrics.py:279-281)
* ``total_score`` = sum of <key>_score / len(ALL input_columns)               (metrics.py:298)
* returned ``[scores]`` holds ``<key>_score_num`` / ``<key>_score_den``        (metrics.py:287-288)

``BeautyLayer`` / ``LayoutMetricLayer`` / ``mae_from_logits`` are unused by train.py and eval.py
 and are not provided.  The RICO position-sorted variant (``sort_flag``,
metrics.py:180-211) runs on the device too: ``mfp_sort_positions`` turns the two ``sort_inputs``
calls into row maps that ``mfp_loss_fwd_bwd_sorted`` reads logits / targets through.
"""
from typing import Dict, List, Union

import torch

from mfp.data.spec import get_valid_input_columns
from mfp.hip import ops


def loss_key_names(input_columns: Dict) -> List[str]:
    return [k for k, c in input_columns.items()
            if not c.get("demo_only", False) and c.get("is_sequence", False)]


def build_loss_keys(input_columns: Dict, head_cols: Dict, y_true: Dict, mfp_masks: Dict) -> List[dict]:
    """Descriptors for ``mfp_loss_fwd_bwd`` (one per sequence attribute, LossLayer order)."""
    keys = []
    for key in loss_key_names(input_columns):
        column = input_columns[key]
        off, units = head_cols[key]
        categorical = column["type"] == "categorical"
        target = y_true[key]
        if categorical:
            target = target.to(torch.int32)
        else:
            target = target.to(torch.float32)
        mask = mfp_masks[key]
        mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
        d = dict(col_off=off,
                 n_feat=column["shape"][-1] if categorical else 1,
                 n_class=column["input_dim"] if categorical else column["shape"][-1],
                 is_numerical=not categorical,
                 target=target.contiguous(), mask=mask.contiguous())
        if "loss_condition" in column:
            cond = column["loss_condition"]
            assert len(cond["mask"]) <= 32, "loss_condition vocabularies above 32 entries unsupported"
            ck = y_true[cond["key"]].to(torch.int32).contiguous()
            d.update(cond_idx=ck, cond_stride=ck.shape[-1],
                     cond_bits=sum(1 << i for i, f in enumerate(cond["mask"]) if f))
        keys.append(d)
    return keys


SORT_KEYS = ["type", "left", "top", "width", "height"]   # reference models/tensor_utils.py:11


def build_loss_sort(input_columns: Dict, head_cols: Dict, y_true: Dict, sort_flag: torch.Tensor,
                    ignore_sort: str = None) -> dict:
    """Descriptor of the position-sorted loss (reference metrics.py:180-211, tensor_utils.py:14-44)
    for ``ops.sort_positions``: the documents whose ``sort_flag`` is set have targets ordered by
    (type, left, top, width, height) and predictions ordered by the argmax of those heads."""
    assert ignore_sort in ("gt", "pred", None)                      # metrics.py:181
    heads = []
    for key in SORT_KEYS:
        assert key in y_true and input_columns[key]["input_dim"] < 100   # tensor_utils.py:19-21
        heads.append((head_cols[key][0], input_columns[key]["input_dim"]))   # feature 0 of the head
    flag = sort_flag.view(torch.uint8) if sort_flag.dtype == torch.bool else (sort_flag != 0).to(torch.uint8)
    labels = [y_true[key].to(torch.int32).contiguous() for key in SORT_KEYS]
    return dict(flag=flag.contiguous(), labels=labels, heads=heads, ignore_sort=ignore_sort)


def metrics_from_sums(input_columns: Dict, sums: torch.Tensor):
    """-> (losses{key}, scores{key_score_num/_den}, metrics{...}) from the [nkeys][3] sums."""
    names = loss_key_names(input_columns)
    den = sums[:, 2]
    normalized = torch.where(den == 0.0, torch.ones_like(den), sums[:, 1] / den.clamp(min=1e-30))
    losses, scores, metrics = {}, {}, {}
    for i, key in enumerate(names):
        metrics[key + "_score"] = normalized[i]
        scores[key + "_score_num"] = sums[i, 1]
        scores[key + "_score_den"] = sums[i, 2]
        losses[key] = sums[i, 0]
    for key, loss in losses.items():
        metrics[key + "_loss"] = loss
    metrics["total_score"] = normalized.sum() / len(input_columns)
    return losses, scores, metrics


class LossLayer:
    def __init__(self, input_columns: Dict, name: str = "loss_layer", predict_context: bool = False,
                 model_layout=None, **kwargs):
        """``model_layout`` (new, optional): the ``ModelLayout`` of the model whose outputs this layer
        scores.  The model's concatenated logits buffer ``_flat_logits`` pads every head to a multiple
        of 8 columns; it is only consumed in place when the layout that produced it is known --
        otherwise the per-key logits of ``y_pred`` are re-concatenated (unpadded)."""
        if predict_context:
            raise NotImplementedError("predict_context is off the MFP hot path")
        self.name = name
        self._input_columns = input_columns
        self._valid_input_columns = get_valid_input_columns(input_columns)
        self.losses: List[torch.Tensor] = []
        self.metrics: Dict[str, torch.Tensor] = {}
        self.sums = None
        col, self._head_cols = 0, {}
        for k, c in self._valid_input_columns.items():
            units = c["shape"][-1] * c["input_dim"] if c["type"] == "categorical" else c["shape"][-1]
            self._head_cols[k] = (col, units)
            col += units
        self._U = col
        self._model_head_cols, self._model_ld = None, None
        if model_layout is not None:
            assert sorted(model_layout.head_cols) == sorted(self._head_cols)      # (the model keeps categorical heads first)
            self._model_head_cols, self._model_ld = dict(model_layout.head_cols), model_layout.Upad

    def _flat_logits(self, y_pred: Dict, B: int, S: int):
        """-> (logits [B*S][ld] f32, head_cols).  The model's own buffer is used in place only
        together with the model's (8-aligned) column offsets."""
        flat = y_pred.get("_flat_logits")
        if (flat is not None and self._model_head_cols is not None and flat.shape[0] == B * S
                and flat.shape[1] == self._model_ld):
            return flat, self._model_head_cols
        parts = [y_pred[k][:, :S].reshape(B * S, -1).to(torch.float32) for k in self._head_cols]
        return torch.cat(parts, dim=1).contiguous(), self._head_cols

    def __call__(self, inputs, training=False, sort_flag: Union[bool, torch.Tensor] = None,
                 ignore_sort: str = None):
        y_true, y_pred, mfp_masks = inputs
        first = next(iter(self._head_cols))
        B, S = y_true[first].shape[:2]
        logits, head_cols = self._flat_logits(y_pred, B, S)
        nvalid = (y_true["length"].reshape(-1) + 1).to(torch.int32)
        keys = build_loss_keys(self._input_columns, head_cols, y_true, mfp_masks)
        pred_row = true_row = None
        if torch.is_tensor(sort_flag):                               # metrics.py:180-211
            from mfp.hip.functions import loss_row_maps
            sort = build_loss_sort(self._input_columns, head_cols, y_true, sort_flag, ignore_sort)
            pred_row, true_row = loss_row_maps(sort, logits, nvalid, B, S)
        sums, _ = ops.loss_fwd_bwd(logits, keys, nvalid, B, S, None, pred_row=pred_row, true_row=true_row)
        losses, scores, metrics = metrics_from_sums(self._input_columns, sums)
        self.losses = [sums[:, 0].sum()]
        self.metrics = metrics
        self.sums = sums
        return [scores]
