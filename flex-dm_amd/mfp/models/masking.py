"""BERT-style masking of document attributes (reference models/masking.py).

These are the *caller-side producers* of the hot path's inputs: cheap
integer / element-wise work on ``(B,S)`` masks that stays in torch on whatever device the
batch lives on.  Semantics, constants and call signatures follow the reference:

* constants ``masking.py:8-15``; ``get_task_names`` ``:18-21``; ``filter_padding`` ``:24-53``;
  ``get_initial_masks`` ``:56-65``; ``apply_token`` ``:68-95``; ``select_single_element``
  ``:98-113``; ``feat_masking`` ``:116-133``; ``elem_masking`` ``:136-155``;
  ``random_masking`` ``:227-269``.

``random_masking`` additionally accepts (and ignores when ``None``) ``replace_prob`` /
``unchange_prob`` so that ``eval.py --task_mode random`` works (the reference raises
``TypeError`` there: eval.py:59-65 vs masking.py:227-231).
"""
from typing import Any, Dict, List, Optional, Tuple

import torch

from mfp.data.spec import get_attribute_groups

MASK_VALUE = 10.0
NULL_VALUE = 0.0

MASK_PROB = 0.15
REPLACE_PROB = 0.1
UNCHANGE_PROB = 0.1
CHANGE_PROB = 1.0 - UNCHANGE_PROB
THRESH = REPLACE_PROB / CHANGE_PROB


def get_task_names(input_columns):
    task_names = ["random", "elem"]
    task_names += list(get_attribute_groups(input_columns.keys()).keys())
    return task_names


def _rand(shape, device, generator=None):
    return torch.rand(tuple(shape), device=device, generator=generator)


def filter_padding(inputs: Dict[str, torch.Tensor], input_columns: Dict,
                   mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Set the ``<UNUSED>`` token on padding and on attributes an element type lacks."""
    modified_inputs = {}
    unused_mask = ~mask
    for key, column in input_columns.items():
        input_ = inputs[key]
        if column["is_sequence"]:
            if "loss_condition" in column:
                cond = column["loss_condition"]
                mask_ = torch.zeros_like(mask)
                for i, flag in enumerate(cond["mask"]):
                    if not flag:
                        mask_ = mask_ | (inputs[cond["key"]] == i)[..., 0]
                mask_ = mask_ | unused_mask
            else:
                mask_ = unused_mask
            modified_inputs[key] = apply_token(input_, column, mask_, "unused")
        else:
            modified_inputs[key] = input_
    return modified_inputs


def get_initial_masks(input_columns: Dict, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    masks = {}
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            masks[key] = torch.ones(mask.shape[:1], dtype=torch.bool, device=mask.device)
        else:
            masks[key] = torch.zeros_like(mask)
    return masks


def apply_token(input_: torch.Tensor, column: Dict[str, Any], mask: torch.Tensor,
                token_type: str, generator=None) -> torch.Tensor:
    assert token_type in ["masked", "unused", "random"]
    assert mask.dim() == 2
    assert input_.dim() == 3
    mask = mask[..., None]
    if column["type"] == "categorical":
        x = mask.to(input_.dtype)
        if token_type == "masked":
            data = column["input_dim"]
        elif token_type == "unused":
            data = column["input_dim"] + 1
        else:
            data = torch.randint(0, column["input_dim"], input_.shape, device=input_.device,
                                 generator=generator).to(input_.dtype)
        output = input_ * (1 - x) + data * x
    else:
        x = mask.to(torch.float32)
        if token_type == "masked":
            data = MASK_VALUE
        elif token_type == "unused":
            data = NULL_VALUE
        else:
            data = 0.1 * torch.randn(input_.shape, device=input_.device, generator=generator)
        output = input_ * (1.0 - x) + data * x
    return output


def select_single_element(mask: torch.Tensor, select_last: bool = False,
                          generator=None) -> torch.Tensor:
    assert mask.dim() == 2
    length = mask.to(torch.int64).sum(dim=1).to(torch.float32)
    if select_last:
        arr = (length - 1).to(torch.int64)
    else:
        arr = (_rand(mask.shape[:1], mask.device, generator) * length).to(torch.int64)
    ar = torch.arange(mask.shape[1], device=mask.device)
    new_mask = ar[None, :] == arr[:, None]  # one_hot(arr, S); arr=-1 -> all False
    new_mask = new_mask & (length > 0.0)[:, None]
    return new_mask


def feat_masking(inputs, input_columns, mask, feat_group: List[str]
                 ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    modified_inputs = {key: inputs[key] for key in inputs.keys()}
    masks = get_initial_masks(input_columns, mask)
    for key in feat_group:
        column = input_columns[key]
        modified_inputs[key] = apply_token(modified_inputs[key], column, mask, "masked")
        masks[key] = mask
    return modified_inputs, masks


def elem_masking(inputs, input_columns, mask, is_autoreg: bool = False, generator=None
                 ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    masks = get_initial_masks(input_columns, mask)
    selected_mask = select_single_element(mask, is_autoreg, generator)
    modified_inputs = {}
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            modified_inputs[key] = inputs[key]
        else:
            modified_inputs[key] = apply_token(inputs[key], column, selected_mask, "masked")
            masks[key] = selected_mask
    return modified_inputs, masks


def random_masking(inputs, input_columns, mask, replace_prob: Optional[float] = None,
                   unchange_prob: Optional[float] = None, generator=None
                   ) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """15 % of valid positions per attribute; of those 90 % changed: 8/9 <MASK>, 1/9 random."""
    change_prob = CHANGE_PROB if unchange_prob is None else 1.0 - unchange_prob
    rep = REPLACE_PROB if replace_prob is None else replace_prob
    thresh = rep / change_prob if change_prob > 0 else 0.0
    modified_inputs = {}
    masks = {}
    for key, column in input_columns.items():
        if not column["is_sequence"]:
            modified_inputs[key] = inputs[key]
            masks[key] = torch.ones(inputs[key].shape, dtype=torch.bool,
                                    device=inputs[key].device)
            continue
        shape = inputs[key].shape[:-1]
        dev = inputs[key].device
        mfp_mask = mask & (_rand(shape, dev, generator) < MASK_PROB)
        chg_mask = mfp_mask & (_rand(shape, dev, generator) < change_prob)
        rand_arr = _rand(shape, dev, generator)
        masked_input = apply_token(inputs[key], column, chg_mask & (rand_arr >= thresh), "masked")
        masked_input = apply_token(masked_input, column, chg_mask & (rand_arr < thresh),
                                   "random", generator)
        modified_inputs[key] = masked_input
        masks[key] = mfp_mask
    return modified_inputs, masks
