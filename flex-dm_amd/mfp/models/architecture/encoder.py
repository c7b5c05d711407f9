"""Encoder: per-attribute embedding + sum fusion (reference architecture/encoder.py:14-265).

Only the hot-path configuration is implemented: ``fusion="add"``, ``context=None``,
``input_dtype="set"``, no element-wise noise (encoder.py:72-92,147-199,260-265); the ablation
paths (flat/concat/none fusion, context tokens, position tokens) are out of scope
 and raise ``NotImplementedError``.
"""
from typing import Dict, Union

import torch

from mfp.data.spec import get_valid_input_columns
from mfp.hip.functions import EncoderFn, StepCtx
from mfp.models.architecture.mask import get_seq_mask

CONTEXT_NAMES = [None, "id", "canvas", "length", "canvas_add"]


class Encoder:
    def __init__(self, input_columns: Dict, store, context: Union[str, None] = None,
                 input_dtype: str = "set", use_elemwise_noise: bool = False, fusion: str = "add",
                 latent_dim: int = 128, dropout: float = 0.1, l2: float = None, **kwargs):
        assert context in CONTEXT_NAMES
        assert fusion in ["add", "concat", "flat", "none"]
        if context is not None or input_dtype != "set" or use_elemwise_noise or fusion != "add":
            raise NotImplementedError(
                "only context=None, input_dtype='set', fusion='add' is on the accelerated MFP path")
        self.input_columns = input_columns
        self.valid_input_columns = get_valid_input_columns(input_columns, False)
        self.context, self.fusion, self.latent_dim = context, fusion, latent_dim
        self.store = store

    def __call__(self, inputs: Dict, ctx: StepCtx):
        L = self.store.layout
        S = ctx.S
        seq_mask = get_seq_mask(inputs["length"], maxlen=S)
        cat_inputs = [inputs[k] for k in L.cat_keys]
        num_inputs = [inputs[k] for k in L.num_keys]
        h = EncoderFn.apply(self.store.anchor, ctx, cat_inputs, num_inputs)
        return h.view(ctx.B, S, L.D), seq_mask
