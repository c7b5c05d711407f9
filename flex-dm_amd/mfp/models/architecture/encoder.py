"""Encoder: per-attribute embedding + sum fusion (reference architecture/encoder.py:14-265).

Implemented: ``fusion="add"``, no element-wise noise (encoder.py:72-92,147-199,260-265), with
``input_dtype="set"`` (the hot path) or ``"shuffled_set"`` (adds the learned position token
``input_const``, encoder.py:47-55,241-242), and the context tokens ``context in {"id", "length"}``
(encoder.py:96-110,226-248: a task / length embedding prepended to the sequence; with
``input_dtype="set"`` only).  The other ablation paths (flat/concat/none fusion, the canvas contexts)
raise ``NotImplementedError``.
"""
from typing import Dict, Union

import torch

from mfp.data.spec import get_valid_input_columns
from mfp.hip.functions import ContextTokenFn, EncoderFn, PosConstFn, StepCtx
from mfp.models.architecture.mask import get_seq_mask

CONTEXT_NAMES = [None, "id", "canvas", "length", "canvas_add"]


class Encoder:
    def __init__(self, input_columns: Dict, store, context: Union[str, None] = None,
                 input_dtype: str = "set", use_elemwise_noise: bool = False, fusion: str = "add",
                 latent_dim: int = 128, dropout: float = 0.1, l2: float = None, **kwargs):
        assert context in CONTEXT_NAMES
        assert fusion in ["add", "concat", "flat", "none"]
        if (context not in (None, "id", "length") or input_dtype not in ("set", "shuffled_set") or use_elemwise_noise
                or fusion != "add" or (context is not None and input_dtype != "set")):
            raise NotImplementedError(
                "provided: context in {None, 'id', 'length'} (the latter two with input_dtype='set'), "
                "input_dtype in {'set', 'shuffled_set'}, fusion='add'")
        self.use_pos_token = input_dtype != "set"            # encoder.py:41
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
        h = EncoderFn.apply(self.store.anchor, ctx, cat_inputs, num_inputs).view(ctx.B, S, L.D)
        if self.use_pos_token:                               # encoder.py:241-242
            if S > L.pos_rows:
                raise ValueError("sequence length %d exceeds the %d rows of the position table" % (S, L.pos_rows))
            h = PosConstFn.apply(h, self.store.anchor, ctx)
        if self.context is not None:                         # encoder.py:226-248
            h = self.add_context_token(h, inputs, ctx)
            seq_mask = get_seq_mask(inputs["length"] + 1, maxlen=S + 1)
        return h, seq_mask

    def context_ids(self, inputs: Dict) -> torch.Tensor:
        ids = inputs["task"] if self.context == "id" else inputs["length"]
        return (ids[:, 0] if ids.dim() == 2 else ids).to(torch.int64)

    def add_context_token(self, h, inputs: Dict, ctx: StepCtx):
        return ContextTokenFn.apply(h, self.context_ids(inputs), self.store.anchor, ctx)
