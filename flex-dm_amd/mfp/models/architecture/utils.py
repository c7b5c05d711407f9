"""L2 regulariser options (reference architecture/utils.py:8-22).

In the reference these are Keras regulariser objects attached per Dense / Embedding; here they
are per-variable coefficients consumed by the fused Adam kernel (``ParamStore.seg_l2``):
``l2 * sum(w^2)`` on every kernel, bias and embedding table, none on LayerNorm gamma/beta.
"""


def make_dense_options(l2):
    if l2 is None:
        return {}
    return dict(kernel_regularizer=("l2", l2), bias_regularizer=("l2", l2))


def make_emb_options(l2):
    if l2 is None:
        return {}
    return dict(embeddings_regularizer=("l2", l2))
