"""Sequence mask from zero-based lengths (reference architecture/mask.py:21-33)."""
from typing import Optional

import torch


def get_seq_mask(inputs: torch.Tensor, from_logits: bool = False,
                 maxlen: Optional[int] = None) -> torch.Tensor:
    """``sequence_mask(reshape(length,-1)+1, maxlen)`` -> ``(B, maxlen)`` bool.

    ``length`` is zero-based (mask.py:29).  With ``maxlen=None`` the width is
    ``max(length)+1`` exactly as ``tf.sequence_mask`` does; that forces one device->host
    read, so hot-path callers pass ``maxlen`` (= the padded S of the batch).
    """
    if from_logits:
        length = inputs.argmax(dim=-1).reshape(-1)
    else:
        length = inputs.reshape(-1)
    length = length.to(torch.int64) + 1
    if maxlen is None:
        maxlen = int(length.max().item()) if length.numel() else 0
    ar = torch.arange(maxlen, device=length.device)
    seq_mask = ar[None, :] < length[:, None]
    assert seq_mask.dim() == 2
    return seq_mask
