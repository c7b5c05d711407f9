"""``Model`` = Encoder -> Blocks -> Decoder (reference models/model.py:9-52, ``_OneShot``/``Model``).

The baselines in the reference's model.py (VanillaTransformer, AutoReg, BART) are unreachable
from train.py/eval.py (mfp.py:230 asserts ``arch_type == "oneshot"``) and are not provided.
New, additive constructor arguments: ``dtype`` ("fp32" parity path / "bf16" MFMA path / "fp8": bf16 with
e4m3 QKV and FFN1 forward products, BASELINE config c5),
``device``, ``seed``.
"""
from typing import Dict, Optional, Union

import torch

from mfp import dp
from mfp.hip.functions import DecoderLossFn, EncoderPreFn, StepCtx
from mfp.models.architecture.decoder import Decoder, split_logits
from mfp.models.architecture.encoder import Encoder
from mfp.models.architecture.transformer import Blocks
from mfp.models.params import ModelLayout, ParamStore

DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
          "fp8": torch.bfloat16}   # fp8: bf16 everywhere except the QKV / FFN1 forward products (e4m3 operands)


def _first_seq_key(input_columns):
    for k, c in input_columns.items():
        if c.get("is_sequence") and not c.get("demo_only"):
            return k
    raise ValueError("no sequence column")


class Model:
    def __init__(self, input_columns: Dict, num_blocks: int = 4, block_type: str = "deepsvg",
                 context: Union[str, None] = None, input_dtype: str = "set",
                 use_elemwise_noise: bool = False, latent_dim: int = 256, dropout: float = 0.1,
                 l2: Optional[float] = None, dtype: str = "fp32", device: str = "cuda", seed: int = 0,
                 **kwargs):
        self.arch_type = "oneshot"
        self.input_columns = input_columns
        self.dropout, self.seed = dropout, seed
        self.layout = ModelLayout(input_columns, latent_dim, num_blocks, input_dtype, context)
        self.context = context
        self.store = ParamStore(self.layout, device, DTYPES[dtype], l2=l2, seed=seed, fp8=(dtype == "fp8"))
        self.blocks = Blocks(self.store, num_blocks=num_blocks, block_type=block_type,
                             latent_dim=latent_dim, dropout=dropout, l2=l2)
        self.encoder = Encoder(input_columns, self.store, context=context, input_dtype=input_dtype,
                               use_elemwise_noise=use_elemwise_noise, latent_dim=latent_dim,
                               dropout=dropout, l2=l2)
        self.decoder = Decoder(input_columns, self.store, context=context, latent_dim=latent_dim,
                               dropout=dropout, l2=l2)
        self.step_ptr = None  # device int32 step counter (set by the optimizer) for dropout offsets
        # everything runs on ONE stream (weight gradients on side streams were measured 85 us per step slower with the grouped
        # launches, rounds 1-4; the switch is gone)
        self.side_stream = None
        self.side_streams = []
        self._first = _first_seq_key(input_columns)

    def make_ctx(self, inputs: Dict, training: bool, nvalid: Optional[torch.Tensor] = None) -> StepCtx:
        B, S = inputs[self._first].shape[:2]
        if nvalid is None:      # (the train step gets it from the step prologue kernel)
            nvalid = (inputs["length"].reshape(-1) + 1).to(torch.int32)
        return StepCtx(self.store, B, S, nvalid, training, self.dropout, dp.rank_seed(self.seed), self.step_ptr,
                       self.side_streams if training and self.side_streams else None)

    def hidden(self, inputs: Dict, training: bool = False, ctx: Optional[StepCtx] = None):
        ctx = ctx or self.make_ctx(inputs, training)
        h, mask = self.encoder(inputs, ctx)
        bctx = ctx.with_context_token() if self.context is not None else ctx    # the blocks see S + 1 positions
        h = self.blocks(h, mask, bctx)
        ctx.mid, ctx.cuts = bctx.mid, bctx.cuts
        return h, ctx

    def __call__(self, inputs: Dict, training: bool = False) -> Dict[str, torch.Tensor]:
        """_OneShot.call (model.py:26-30): dict of per-attribute logits."""
        h, ctx = self.hidden(inputs, training)
        return self.decoder(h, ctx)

    def forward_loss(self, inputs: Dict, loss_keys, training: bool = True, premasked=None, ctx=None,
                     loss_sort=None):
        """Train-step path: heads + LossLayer fused.  Returns (loss_total, sums, outputs).  When ``ctx.tail["sums"]``
        carries the step prologue's accumulator (MFP._forward) ``loss_total`` is a placeholder root and ``sums`` becomes
        valid with the end-of-backward reduction (``ctx.flush_ln_jobs``); every other caller gets both at once.
        ``premasked`` = (idx_all, codes, xs) from the fused masking kernel replaces ``inputs``."""
        if premasked is not None:
            # the last block feeds the heads directly (no context token to strip): its MLP kernel writes the heads'
            # bf16 operand and the heads' input-gradient kernel the masked gradient it starts its backward from
            ctx.tail["fuse"] = self.context is None and training
            h = EncoderPreFn.apply(self.store.anchor, ctx, *premasked).view(ctx.B, ctx.S, self.layout.D)
            bctx = ctx
            if self.context is not None:
                h = self.encoder.add_context_token(h, inputs, ctx)
                bctx = ctx.with_context_token()
            h = self.blocks(h, None, bctx)
            ctx.mid, ctx.cuts = bctx.mid, bctx.cuts
        else:
            h, ctx = self.hidden(inputs, training)
        if self.context is not None:      # decoder.py:74-76
            h = h[:, 1:]
        B, S, D = h.shape
        ctx.loss_sort = loss_sort
        loss, sums, logits = DecoderLossFn.apply(h.reshape(B * S, D), ctx, loss_keys)
        if logits.shape[0] == 0:      # (the train step asked for no logits: ctx.tail["want_logits"] = False)
            return loss, sums, {}
        outputs = split_logits(logits, self.layout, self.input_columns, B, S)
        outputs["_flat_logits"] = logits
        return loss, sums, outputs
