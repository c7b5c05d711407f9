"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the CALLER side of the MFP
hot path: the masking producers, the task mix and the MaskGIT-style iterative decode.

PARITY UNPINNED (see ``oracle/np_ref.py``: the reference has no tests / golden vectors and
TensorFlow cannot be imported here).  Only ``tests/`` and ``__graft_entry__.smoke()`` import this.

The reference draws its randomness from TensorFlow's stateful generators (``tf.random.uniform``,
``tf.random.normal``, ``tfp`` Categorical) whose streams cannot be reproduced without TensorFlow.
Every function here therefore takes its random DRAWS as explicit arguments (uniforms in [0, 1),
replacement tokens, task ids): the arithmetic that turns draws into masks and tokens is what the
reference defines, and that is restated bit for bit.  A test replays the draws an implementation
made (or infers them from its output) through these functions and demands identical results; the
probabilities (MASK_PROB etc.) are checked statistically.

Reference line citations are relative to ``/root/reference/src/mfp/mfp/``.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np

MASK_VALUE = 10.0                        # models/masking.py:8
NULL_VALUE = 0.0                         # :9
MASK_PROB = 0.15                         # :11
REPLACE_PROB = 0.1                       # :12
UNCHANGE_PROB = 0.1                      # :13
CHANGE_PROB = 1.0 - UNCHANGE_PROB        # :14
THRESH = REPLACE_PROB / CHANGE_PROB      # :15

ATTRIBUTE_GROUPS = {                     # data/spec.py:364-377
    "rico": {"type": ["type"], "pos": ["left", "top", "width", "height"],
             "attr": ["icon", "clickable", "text_button"]},
    "crello": {"type": ["type"], "pos": ["left", "top", "width", "height"],
               "attr": ["opacity", "color", "font_family"], "img": ["image_embedding"],
               "txt": ["text_embedding"]},
}


def get_attribute_groups(keys) -> Dict[str, List[str]]:
    """data/spec.py:380-391."""
    return ATTRIBUTE_GROUPS["rico" if "clickable" in keys else "crello"]


def get_task_names(input_columns) -> List[str]:
    """models/masking.py:18-21."""
    return ["random", "elem"] + list(get_attribute_groups(input_columns.keys()).keys())


def task_probs(task_names: List[str], masking_method: str) -> List[float]:
    """models/mfp.py:34-43: the probabilities behind the tfp Categorical."""
    used = masking_method.split("_")
    probs = [1.0 if n in used else 0.0 for n in task_names]
    total = sum(probs)
    assert total > 0.0
    return [p / total for p in probs]


def sample_tasks(probs: List[float], u: np.ndarray) -> np.ndarray:
    """Categorical(probs) by inverse CDF on uniforms ``u`` (B,) -> int32 task ids.  ([TF-EXT] tfp
    samples with the Gumbel trick on its own stream; any exact sampler has the same law.)"""
    cdf = np.cumsum(np.asarray(probs, np.float64))
    cdf[-1] = 1.0 + 1e-12
    return np.searchsorted(cdf, np.asarray(u, np.float64), side="right").astype(np.int32)


def get_seq_mask(length, maxlen=None) -> np.ndarray:
    """models/architecture/mask.py:21-33."""
    length = np.asarray(length).reshape(-1).astype(np.int64) + 1
    if maxlen is None:
        maxlen = int(length.max())
    return np.arange(maxlen)[None, :] < length[:, None]


def apply_token(x, column, mask, token_type: str, random_data=None) -> np.ndarray:
    """models/masking.py:68-95: ``x * (1 - m) + data * m`` with the token of ``token_type``.
    ``random_data`` (shape of ``x``) supplies the draws of the "random" token: integers in
    [0, input_dim) for a categorical column, N(0, 0.1) floats for a numerical one."""
    assert token_type in ("masked", "unused", "random")
    x = np.asarray(x)
    mask = np.asarray(mask).astype(bool)
    assert mask.ndim == 2 and x.ndim == 3                                       # :72-73
    m = mask[..., None]
    if column["type"] == "categorical":
        data = {"masked": column["input_dim"], "unused": column["input_dim"] + 1}.get(token_type)   # :80-83
        if token_type == "random":
            data = np.asarray(random_data).astype(x.dtype)
        mi = m.astype(x.dtype)
        return (x * (1 - mi) + data * mi).astype(x.dtype)                       # :85
    data = {"masked": MASK_VALUE, "unused": NULL_VALUE}.get(token_type)         # :88-92
    if token_type == "random":
        data = np.asarray(random_data, np.float32)
    mf = m.astype(np.float32)
    return (x.astype(np.float32) * (np.float32(1.0) - mf) + np.float32(1.0) * data * mf).astype(np.float32)   # :93


def filter_padding(inputs, input_columns, mask) -> Dict[str, np.ndarray]:
    """models/masking.py:24-53: <UNUSED> on padding and on attributes the element's type lacks."""
    out = {}
    unused = ~np.asarray(mask).astype(bool)
    for key, column in input_columns.items():
        if column["is_sequence"]:
            if "loss_condition" in column:
                cond = column["loss_condition"]
                m = np.zeros(unused.shape, bool)
                for i, flag in enumerate(cond["mask"]):
                    if not flag:
                        m |= (np.asarray(inputs[cond["key"]]) == i)[..., 0]      # :41-43
                m |= unused
            else:
                m = unused
            out[key] = apply_token(inputs[key], column, m, "unused")
        else:
            out[key] = np.asarray(inputs[key])
    return out


def get_initial_masks(input_columns, mask) -> Dict[str, np.ndarray]:
    """models/masking.py:56-65."""
    mask = np.asarray(mask)
    return {k: (np.zeros(mask.shape, bool) if c["is_sequence"] else np.ones(mask.shape[:1], bool))
            for k, c in input_columns.items()}


def select_single_element(mask, u=None, select_last: bool = False) -> np.ndarray:
    """models/masking.py:98-113: slot ``int(u * length)`` of each document (``u`` (B,) uniforms, f32
    arithmetic as in the reference), all-False when the document is empty."""
    mask = np.asarray(mask).astype(bool)
    length = mask.astype(np.int64).sum(axis=1).astype(np.float32)
    if select_last:
        arr = (length - 1).astype(np.int32)
    else:
        arr = (np.asarray(u, np.float32) * length).astype(np.int32)
    new = np.arange(mask.shape[1])[None, :] == arr[:, None]                     # one_hot(arr, S); -1 -> no slot
    return new & (length > 0.0)[:, None]


def feat_masking(inputs, input_columns, mask, feat_group: List[str]):
    """models/masking.py:116-133: <MASK> on every valid position of the group's attributes."""
    modified = {k: np.asarray(v) for k, v in inputs.items()}
    masks = get_initial_masks(input_columns, mask)
    for key in feat_group:
        modified[key] = apply_token(modified[key], input_columns[key], mask, "masked")
        masks[key] = np.asarray(mask).astype(bool)
    return modified, masks


def elem_masking(inputs, input_columns, mask, u=None, is_autoreg: bool = False):
    """models/masking.py:136-155: one element of each document gets <MASK> in all attributes."""
    masks = get_initial_masks(input_columns, mask)
    selected = select_single_element(mask, u, is_autoreg)
    modified = {}
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            modified[key] = np.asarray(inputs[key])
        else:
            modified[key] = apply_token(inputs[key], column, selected, "masked")
            masks[key] = selected
    return modified, masks


def random_masking(inputs, input_columns, mask, draws: Dict[str, dict]):
    """models/masking.py:227-269.  ``draws[key]`` = dict(u_mask, u_chg, u_tok: (B,S) uniforms,
    random: replacement tokens shaped like the input) -- per attribute, in this order, as the
    reference draws them (:248, :252-255, and inside apply_token :83/:91)."""
    modified, masks = {}, {}
    mask = np.asarray(mask).astype(bool)
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            modified[key] = np.asarray(inputs[key])
            masks[key] = np.ones(np.asarray(inputs[key]).shape, bool)            # :243
            continue
        d = draws[key]
        mfp_mask = mask & (np.asarray(d["u_mask"]) < MASK_PROB)                 # :248-249
        chg_mask = mfp_mask & (np.asarray(d["u_chg"]) < CHANGE_PROB)            # :252-254
        r = np.asarray(d["u_tok"])                                               # :255
        x = apply_token(inputs[key], column, chg_mask & (r >= THRESH), "masked")
        x = apply_token(x, column, chg_mask & (r < THRESH), "random", d.get("random"))
        modified[key], masks[key] = x, mfp_mask
    return modified, masks


def preprocess_for_train(inputs, input_columns, tasks, draws: Dict[str, dict], u_elem=None,
                         is_autoreg: bool = False, maxlen: Optional[int] = None):
    """models/mfp.py:95-138 for ``input_dtype="set"``: all task variants are built and the per-document
    task id selects among them (task 0 = random, 1 = elem, 2.. = the attribute groups in order).
    ``draws`` may be None when task 0 is never selected (its variant is then irrelevant)."""
    tasks = np.asarray(tasks)
    assert tasks.ndim == 1                                                       # :102
    groups = get_attribute_groups(input_columns.keys())
    seq_mask = get_seq_mask(inputs["length"], maxlen)
    filtered = filter_padding(inputs, input_columns, seq_mask)
    if draws is None:
        assert not (tasks == 0).any()
        modified = {k: np.asarray(v) for k, v in filtered.items()}
        masks = {k: (np.zeros(seq_mask.shape, bool) if c["is_sequence"] else np.ones(np.asarray(inputs[k]).shape, bool))
                 for k, c in input_columns.items()}
    else:
        modified, masks = random_masking(filtered, input_columns, seq_mask, draws)   # :113
    if u_elem is None:
        assert not (tasks == 1).any()
        u_elem = np.zeros(tasks.shape[0], np.float32)
    data = [elem_masking(filtered, input_columns, seq_mask, u_elem, is_autoreg)]     # :114
    for group in groups.values():                                                # :115-117
        data.append(feat_masking(filtered, input_columns, seq_mask, group))
    for key in list(modified.keys()):                                            # :119-134
        for i, (mod_tmp, masks_tmp) in enumerate(data):
            cond = tasks == (i + 1)
            if input_columns[key]["is_sequence"]:
                cond = cond[..., None]
            modified[key] = np.where(cond[..., None], mod_tmp[key], modified[key])
            if input_columns[key]["is_sequence"]:
                masks[key] = np.where(cond, masks_tmp[key], masks[key])
    modified["task"] = tasks[..., None]                                          # :137
    return inputs, modified, masks


def preprocess_for_test(inputs, input_columns, masks, tasks=None, maxlen: Optional[int] = None):
    """models/mfp.py:72-92."""
    seq_mask = get_seq_mask(inputs["length"], maxlen)
    filtered = filter_padding(inputs, input_columns, seq_mask)
    modified = {}
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            modified[key] = filtered[key]
            continue
        modified[key] = apply_token(filtered[key], column, masks[key], "masked")
    if tasks is None:
        tasks = np.zeros(np.asarray(inputs["left"]).shape[0])
    modified["task"] = np.asarray(tasks)[..., None]
    return modified


def merge_inputs_and_prediction(inputs, input_columns, masks, prediction):
    """models/mfp.py:46-69: unmasked positions of the prediction are overwritten by the ground truth."""
    prediction = dict(prediction)
    for key, column in input_columns.items():
        if column.get("demo_only", False):
            continue
        if not column["is_sequence"]:
            prediction[key] = inputs[key]
        elif key not in masks:
            continue
        elif column["type"] == "numerical":
            cond = np.asarray(masks[key])[..., None]
            prediction[key] = np.where(cond, prediction[key], np.asarray(inputs[key]))
        else:
            gt = np.eye(column["input_dim"])[np.asarray(inputs[key]).astype(np.int64)]
            cond = np.asarray(masks[key])[..., None, None]
            prediction[key] = np.where(cond, prediction[key], gt)
    return prediction


def _softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def iterative_decode(model: Callable[[Dict], Dict], masks, inputs, input_columns, modified_inputs,
                     num_iter: int, maxlen: Optional[int] = None):
    """MaskGIT-like decoding, models/mfp.py:141-207.  ``model(modified_inputs) -> {key: logits}``.

    One deviation from the letter of the reference, stated: ``confidence[key] >= threshold``
    (mfp.py:187) compares a (B,S) tensor with a (B,) tensor, which broadcasts along the LAST axis
    and is only what the authors meant for B = 1 (the notebook's use) -- here the per-document
    threshold is applied per row (``threshold[:, None]``), identical for B = 1."""
    masks = {k: np.asarray(v).astype(bool) for k, v in masks.items()}
    seq_mask = get_seq_mask(inputs["length"], maxlen)
    filtered = filter_padding(inputs, input_columns, seq_mask)
    cat_keys = [k for k, v in input_columns.items() if v["is_sequence"] and v.get("type") == "categorical"]
    num_masked = sum(masks[k].astype(int).sum(-1) for k in cat_keys)              # :151
    num_update = (num_masked / num_iter).round().astype(int)                     # :152 (round half to even)
    modified_inputs = dict(modified_inputs)
    final, outputs = None, None
    for i in range(num_iter):
        outputs = model(modified_inputs)
        if i == 0:
            final = dict(outputs)
        conf = {k: np.where(masks[k], _softmax(np.asarray(outputs[k], np.float64)).max(-1).mean(-1), 0.0)
                for k in cat_keys}                                              # :160-171
        conf_sorted = -np.sort(-np.concatenate([conf[k] for k in cat_keys], axis=-1), axis=-1)   # :172-176
        threshold = np.stack([conf_sorted[b, k] for b, k in enumerate(num_update)])           # :177-179
        for key in cat_keys:
            pred = np.asarray(outputs[key]).argmax(-1).astype(np.int32)          # :183
            update = (conf[key] >= threshold[:, None]) & (conf[key] > 0)        # :184 (see docstring)
            filtered[key] = np.where(update[:, :, None], pred, filtered[key])
            masks[key] = np.where(masks[key] == update, False, masks[key])       # :188
            if i > 0:
                final[key] = np.where(update[:, :, None, None], outputs[key], final[key])
        for key, column in input_columns.items():                                # :196-200
            if column["is_sequence"]:
                modified_inputs[key] = apply_token(filtered[key], column, masks[key], "masked")
    for key in ("image_embedding", "text_embedding"):                            # :203-205
        if key in outputs:
            final[key] = outputs[key]
    return final
