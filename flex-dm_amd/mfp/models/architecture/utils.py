"""L2 regulariser options (reference architecture/utils.py:8-22).

In the reference these are Keras regulariser objects attached per Dense / Embedding layer; here they are the
per-variable coefficients the fused Adam kernel applies (``ParamStore.seg_l2``): ``l2 * sum(w^2)`` on every Dense
kernel and bias and on every embedding table, none on LayerNormalization gamma / beta.  ``variable_l2`` is the one
place that decides which variable kinds are regularised; ``ParamStore`` builds its coefficient table through it.
"""


def make_dense_options(l2):
    """Dense(kernel_regularizer=l2, bias_regularizer=l2) -> {variable kind: coefficient}."""
    if l2 is None:
        return {}
    return {"kernel": float(l2), "bias": float(l2)}


def make_emb_options(l2):
    """Embedding(embeddings_regularizer=l2) -> {variable kind: coefficient}."""
    if l2 is None:
        return {}
    return {"embeddings": float(l2)}


def variable_l2(name: str, l2) -> float:
    """L2 coefficient of the variable ``name`` (its last path component is the Keras variable kind)."""
    kind = name.rsplit("/", 1)[-1]
    return {**make_dense_options(l2), **make_emb_options(l2)}.get(kind, 0.0)
