"""Transformer blocks (reference architecture/transformer.py).

``DeepSVGBlock`` (pre-norm, transformer.py:208-229) with ``MultiHeadSelfAttention``
(:33-99, 8 heads, key-padding mask) and the ReLU MLP (:161-171) runs as one autograd node of
HIP kernels (``mfp.hip.functions.BlockFn``).  ``Blocks`` (:239-280) stacks them.  The post-norm
``TransformerBlock``, ``PositionEmbedding`` and the cross-attention classes are only reachable
from baselines/ablations and are not provided.
"""
from mfp.hip.functions import BlockFn, StepCtx
from mfp.models.architecture.utils import make_dense_options


class DeepSVGBlock:
    def __init__(self, store, index: int, emb_size=64, num_heads=8, dropout=0.1, conditional=None,
                 pooling=None, dense_options=None, lookahead=True, name=None):
        if emb_size % num_heads != 0:
            raise ValueError(f"embedding dimension = {emb_size} should be divisible by "
                             f"number of heads = {num_heads}.")
        if conditional or pooling or not lookahead or num_heads != 8:
            raise NotImplementedError("conditional / pooling / causal blocks are off the MFP hot path")
        self.store, self.index, self.name = store, index, name

    def __call__(self, x, ctx: StepCtx):
        B, S, D = x.shape
        return BlockFn.apply(x.reshape(B * S, D), ctx, self.index).view(B, S, D)


def get_seq_block(layer_type):
    if layer_type != "deepsvg":
        raise NotImplementedError("block_type=%r: only 'deepsvg' is on the MFP hot path" % layer_type)
    return DeepSVGBlock


class Blocks:
    def __init__(self, store, latent_dim=128, num_blocks=1, block_type="deepsvg", conditional=None,
                 lookahead=True, dropout=0.1, l2=None, **kwargs):
        self.seq2seq = {}
        self.latent_dim, self.num_blocks, self.conditional = latent_dim, num_blocks, conditional
        layer_fn = get_seq_block(block_type)
        for i in range(num_blocks):
            self.seq2seq["seq2seq_%d" % i] = layer_fn(
                store, i, latent_dim, dropout=dropout, conditional=conditional,
                dense_options=make_dense_options(l2), lookahead=lookahead, name="seq2seq_%d" % i)

    def __call__(self, seq, mask, ctx: StepCtx):
        # the key-padding mask enters the kernels as ctx.nvalid (= length + 1 per document)
        # ctx.mid: the activation entering block L/2 -- where the data-parallel step cuts its backward
        # pass in two, so that the gradients of the upper half are all-reduced under the lower half
        for i, layer in enumerate(self.seq2seq.values()):
            if i == self.num_blocks // 2 and i > 0:
                ctx.mid = seq
            if i > 0:
                ctx.cuts[i] = seq
            seq = layer(seq, ctx)
        return seq
