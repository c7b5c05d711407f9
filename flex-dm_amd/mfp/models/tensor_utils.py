"""Host-side tensor helpers (reference models/tensor_utils.py).

Only what the hot path's callers reach: ``sort_inputs`` (RICO position-sorted loss,
tensor_utils.py:14-44), ``shuffle_inputs`` (:47-78), ``reorganize_indices`` (:81-108) and the
dict split/merge helpers (:111-129).
"""
from typing import Dict, List, Union

import torch

from mfp.models.architecture.mask import get_seq_mask

KEYS = ["type", "left", "top", "width", "height"]


def _gather_seq(val: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    idx = indices
    while idx.dim() < val.dim():
        idx = idx.unsqueeze(-1)
    return torch.gather(val, 1, idx.expand(-1, -1, *val.shape[2:]))


def sort_inputs(inputs: Dict, input_columns: Dict, from_logits: bool = False):
    CONST = 100
    assert "length" in inputs
    for key in KEYS:
        assert key in inputs
        assert input_columns[key]["input_dim"] < CONST
    data = {k: v for k, v in inputs.items()}
    for key, column in input_columns.items():
        if column.get("demo_only", False) or key not in data:
            continue
        if column["is_sequence"] and column["type"] == "categorical":
            if from_logits:
                data[key] = data[key].argmax(dim=-1)
            data[key] = data[key].to(torch.int64)
    S = inputs[KEYS[0]].shape[1]
    invalid = ~get_seq_mask(data["length"], maxlen=S)
    priority = torch.zeros_like(data[KEYS[0]][..., 0])
    for key in KEYS:
        priority = priority * CONST + data[key][..., 0]
    priority = priority + invalid.to(torch.int64) * (CONST ** len(KEYS))
    indices = torch.argsort(priority, dim=-1, stable=True)
    new_inputs = {}
    for key, val in inputs.items():
        if key in input_columns and input_columns[key].get("is_sequence", False):
            new_inputs[key] = _gather_seq(val, indices)
        else:
            new_inputs[key] = val
    return new_inputs


def shuffle_inputs(inputs: Dict):
    """Random order of the valid elements of every document, padding left in place (reference
    tensor_utils.py:47-78 walks the batch in Python with ``random.shuffle``; here the permutation
    is an argsort of uniform keys on the device -- same distribution, no host sync, so the step
    stays capturable in a hipGraph)."""
    assert "length" in inputs and "left" in inputs
    B, S = inputs["left"].shape[:2]
    dev = inputs["left"].device
    valid = get_seq_mask(inputs["length"], maxlen=S)
    keys = torch.rand((B, S), device=dev)
    keys = torch.where(valid, keys, 2.0 + torch.arange(S, device=dev, dtype=torch.float32)[None, :])
    indices = torch.argsort(keys, dim=1)
    new_inputs = {}
    for key, val in inputs.items():
        if val.dim() >= 2 and val.shape[1] == S:
            new_inputs[key] = _gather_seq(val, indices)
        else:
            new_inputs[key] = val
    return new_inputs


def reorganize_indices(from_inds: torch.Tensor, n_elems: torch.Tensor,
                       maxlen: Union[int, None] = None):
    assert from_inds.dim() == 2 and n_elems.dim() == 2
    B = from_inds.shape[0]
    if not maxlen:
        maxlen = int(n_elems.max().item()) + 1
    data = []
    for i in range(B):
        from_ind = int(from_inds[i, 0])
        n_elem = int(n_elems[i, 0])
        ids = list(range(maxlen))
        del ids[from_ind]
        data.append(ids[:n_elem] + [from_ind] + ids[n_elem:])
    return torch.tensor(data, device=from_inds.device)


def merge_list_of_dict_of_tensors(inputs: List[Dict[str, torch.Tensor]], axis: int = 0):
    return {k: torch.cat([x[k] for x in inputs], dim=axis) for k in inputs[0].keys()}


def split_dict_of_tensors(inputs: Dict[str, torch.Tensor], num_splits: int = 1, axis: int = 0):
    result = [{} for _ in range(num_splits)]
    for k, v in inputs.items():
        assert v.shape[axis] % num_splits == 0
        for i, x in enumerate(torch.chunk(v, num_splits, dim=axis)):
            result[i][k] = x
    return result
