"""Decoder: one Dense head per attribute (reference architecture/decoder.py:9-111), run as ONE
concatenated-heads GEMM ``(T,D) x (D,U)``; outputs are views of the ``[T][Upad]`` logits buffer
reshaped to ``(B,S,N,C)`` / ``(B,S,512)`` as decoder.py:97-110 does.  Only
``detachment="default"`` is provided; with ``context in {"id", "length"}`` the prepended token's position is
split off first (decoder.py:74-76).
"""
from typing import Dict, Union

from mfp.data.spec import get_valid_input_columns
from mfp.hip.functions import DecoderFn, StepCtx


def split_logits(logits, layout, input_columns, B, S):
    outputs = {}
    for key, (off, units) in layout.head_cols.items():
        column = input_columns[key]
        sl = logits[:, off:off + units]
        if column["type"] == "categorical":
            outputs[key] = sl.reshape(B, S, column["shape"][-1], column["input_dim"])
        else:
            outputs[key] = sl.reshape(B, S, column["shape"][-1])
    return outputs


class Decoder:
    def __init__(self, input_columns: Dict, store, context: Union[str, None] = None,
                 detachment: str = "default", latent_dim: int = 256, dropout: float = 0.1,
                 l2: float = None, **kwargs):
        assert detachment in ["default", "flat", "none"]
        if context not in (None, "id", "length") or detachment != "default":
            raise NotImplementedError("only context in {None, 'id', 'length'}, detachment='default' are provided")
        self.context = context
        self.input_columns = input_columns
        self.valid_input_columns = get_valid_input_columns(input_columns, False)
        self.store, self.latent_dim = store, latent_dim

    def __call__(self, inputs, ctx: StepCtx):
        if self.context is not None:      # decoder.py:74-76: the context position carries no sequence head
            inputs = inputs[:, 1:]
        B, S, D = inputs.shape
        logits = DecoderFn.apply(inputs.reshape(B * S, D), ctx)
        outputs = split_logits(logits, self.store.layout, self.input_columns, B, S)
        outputs["_flat_logits"] = logits
        return outputs
