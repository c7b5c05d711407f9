"""Column schema + synthetic ``DataSpec`` for the MFP hot path.

Mirrors the *contract* of the reference's ``DataSpec`` (reference
``src/mfp/mfp/data/spec.py:24-361``) that the hot path depends on:

* ``make_input_columns()`` -> the ``input_columns`` dict that parameterises every
  shape on the path (reference ``spec.py:144-211``),
* ``get_attribute_groups`` / ``get_dataset_name`` / ``get_valid_input_columns`` /
  ``ATTRIBUTE_GROUPS`` (reference ``spec.py:364-403``),
* ``make_dataset(split, ...)`` yielding dict batches with the reference's dtypes and
  layout (int32 ``(B,S,N)`` categorical, float32 ``(B,S,512)`` numerical, zero-based
  ``length (B,1)``, zero padding past ``length``; reference ``spec.py:255-285``).

The TFRecord reader itself (``tf.data`` + Keras preprocessing layers) is out of scope for
this round: batches are *synthetic*, drawn with the distributions of
SURVEY.md §8d.  The schema is held as Python tables below rather than YAML; the values
(bins, shapes, loss conditions, column order) are those of the reference's
``crello-spec.yml`` / ``rico-spec.yml``.  Vocabulary sizes that the reference reads from
``vocabulary.json`` (absent from the reference tree) default to the synthetic values of
SURVEY.md §8 and are overridden by ``<path>/vocabulary.json`` when that file exists.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Iterator, List, Optional

import numpy as np
import torch

logger = logging.getLogger(__name__)

# --------------------------------------------------------------------------- schema tables
# kind: "demo" | "lookup" | "discretize" | "int" | "float"
# Column order matters: Encoder/Decoder/LossLayer iterate in this order (reference
# encoder.py:72, decoder.py:35, metrics.py:222) and the fp32 fusion sum follows it.


def _disc(bins, shape=(1,)):
    return dict(kind="discretize", bins=bins, shape=shape, is_sequence=True)


_CRELLO_TYPES = ["", "coloredBackground", "imageElement", "maskElement", "svgElement",
                 "textElement", "humanElement"]

_SCHEMAS = {
    "crello": [
        ("id", dict(kind="demo")),
        ("length", dict(kind="lookup", vocab=list(range(1, 51)))),
        ("group", dict(kind="lookup", vocab=[""] + ["group%d" % i for i in range(6)])),
        ("format", dict(kind="lookup", vocab=[""] + ["format%d" % i for i in range(67)])),
        ("canvas_width", dict(kind="lookup", vocab=[-1] + list(range(41)))),
        ("canvas_height", dict(kind="lookup", vocab=[-1] + list(range(46)))),
        ("category", dict(kind="lookup", vocab=[""] + ["cat%d" % i for i in range(23)])),
        ("type", dict(kind="lookup", vocab=_CRELLO_TYPES, is_sequence=True,
                      primary_label="")),
        ("left", _disc(64)),
        ("top", _disc(64)),
        ("width", _disc(64)),
        ("height", _disc(64)),
        ("opacity", _disc(8)),
        ("color", dict(_disc(16, shape=(3,)),
                       loss_condition=("type", ["textElement", "coloredBackground"]))),
        ("image_embedding", dict(kind="float", shape=(512,), is_sequence=True,
                                 loss_condition=("type", ["svgElement", "imageElement",
                                                          "maskElement"]))),
        ("text_embedding", dict(kind="float", shape=(512,), is_sequence=True,
                                loss_condition=("type", ["textElement"]))),
        ("font_family", dict(kind="lookup", is_sequence=True,
                             vocab=["[UNK]"] + ["font%d" % i for i in range(34)],
                             loss_condition=("type", ["textElement"]))),
        ("uuid", dict(kind="demo", is_sequence=True)),
    ],
    "rico": [
        ("length", dict(kind="lookup", vocab=list(range(1, 51)))),
        ("left", _disc(64)),
        ("top", _disc(64)),
        ("width", _disc(64)),
        ("height", _disc(64)),
        ("clickable", dict(kind="int", max=1, is_sequence=True)),
        ("type", dict(kind="lookup", is_sequence=True, primary_label="",
                      vocab=["[UNK]"] + ["component%d" % i for i in range(26)])),
        ("icon", dict(kind="lookup", is_sequence=True,
                      vocab=["[UNK]"] + ["icon%d" % i for i in range(59)])),
        ("text_button", dict(kind="lookup", is_sequence=True,
                             vocab=["[UNK]"] + ["text%d" % i for i in range(29)])),
    ],
}

ATTRIBUTE_GROUPS = {
    "rico": {
        "type": ["type"],
        "pos": ["left", "top", "width", "height"],
        "attr": ["icon", "clickable", "text_button"],
    },
    "crello": {
        "type": ["type"],
        "pos": ["left", "top", "width", "height"],
        "attr": ["opacity", "color", "font_family"],
        "img": ["image_embedding"],
        "txt": ["text_embedding"],
    },
}


def get_dataset_name(keys) -> str:
    """reference spec.py:378-383"""
    return "rico" if "clickable" in keys else "crello"


def get_attribute_groups(keys) -> Dict[str, List[str]]:
    """reference spec.py:386-388"""
    return ATTRIBUTE_GROUPS[get_dataset_name(keys)]


def get_valid_input_columns(input_columns: Dict, use_canvas: bool = False) -> Dict:
    """Columns the encoder/decoder/loss iterate over (reference spec.py:391-403)."""
    outputs = {}
    for key, column in input_columns.items():
        if key == "length":
            continue
        if column.get("demo_only", False):
            continue
        if not column["is_sequence"] and not use_canvas:
            continue
        outputs[key] = column
    return outputs


def set_visual_default(decoded_data: Dict):
    """reference spec.py:16-21 (visualisation helper imported by eval.py)."""
    for element in decoded_data["elements"]:
        element["color"] = [0.0, 0.0, 0.0]
        element["opacity"] = 1.0
        element["font_family"] = "DummyFont"
    return decoded_data


def make_input_columns(name: str, vocabulary: Optional[Dict] = None) -> Dict:
    """Build the ``input_columns`` dict of reference ``spec.py:144-211`` from the tables."""
    schema = _resolve_schema(name, vocabulary)
    inputs: Dict[str, Dict] = {}
    for key, col in schema:
        kind = col["kind"]
        if kind == "demo":
            inputs[key] = {"demo_only": True}
            # the reference still fills shape/is_sequence for demo columns (spec.py:183-184)
        elif kind == "discretize":
            inputs[key] = {"type": "categorical", "input_dim": col["bins"]}
        elif kind == "lookup":
            inputs[key] = {"type": "categorical", "input_dim": len(col["vocab"])}
        elif kind == "int":
            inputs[key] = {"type": "categorical", "input_dim": col["max"] + 1}
        elif kind == "float":
            inputs[key] = {"type": "numerical"}
        else:  # pragma: no cover
            raise NotImplementedError(kind)
        inputs[key]["shape"] = tuple(col.get("shape", (1,)))
        inputs[key]["is_sequence"] = bool(col.get("is_sequence", False))
        if "primary_label" in col:
            inputs[key]["primary_label"] = col["vocab"].index(col["primary_label"]) \
                if col["primary_label"] in col["vocab"] else 0
        else:
            inputs[key]["primary_label"] = None
    by_name = dict(schema)
    for key, col in schema:
        if "loss_condition" in col:
            cond_key, values = col["loss_condition"]
            vocab = by_name[cond_key]["vocab"]
            inputs[key]["loss_condition"] = {
                "key": cond_key,
                "mask": [v in values for v in vocab],
            }
    return inputs


def _resolve_schema(name: str, vocabulary: Optional[Dict]):
    if name not in _SCHEMAS:
        raise ValueError("unknown dataset %r (expected one of %s)" % (name, list(_SCHEMAS)))
    schema = [(k, dict(c)) for k, c in _SCHEMAS[name]]
    if vocabulary:
        for key, col in schema:
            if col["kind"] == "lookup" and key in vocabulary:
                vocab = vocabulary[key]
                if isinstance(vocab, dict):
                    vocab = [k for k, v in vocab.items() if v >= col.get("min_freq", 1)]
                lead = [col["vocab"][0]] if col["vocab"] and col["vocab"][0] in ("", "[UNK]", -1) \
                    else []
                col["vocab"] = lead + [v for v in vocab if v not in lead]
    return schema


# --------------------------------------------------------------------------- synthetic data
def synthetic_batch(
    input_columns: Dict,
    batch_size: int,
    seq_len: int,
    seed: int = 0,
    ragged: bool = False,
    device: str = "cpu",
) -> Dict[str, torch.Tensor]:
    """One Crello/RICO-shaped batch with the distributions of SURVEY.md §8d.

    ``ragged=False``: every document has ``seq_len`` elements (timing runs; elements are
    unambiguous).  ``ragged=True``: ``length+1 ~ U{1..seq_len}`` with at least one
    full-length document so that ``sequence_mask``'s implied maxlen equals ``seq_len``
    (reference mask.py:31), and positions past ``length`` zero-padded (spec.py:255-276).
    """
    rng = np.random.default_rng(seed)
    B, S = batch_size, seq_len
    if ragged:
        n = rng.integers(1, S + 1, size=(B,))
        n[rng.integers(0, B)] = S
    else:
        n = np.full((B,), S)
    valid = (np.arange(S)[None, :] < n[:, None])
    batch: Dict[str, torch.Tensor] = {}
    for key, col in input_columns.items():
        if col.get("demo_only", False):
            continue
        if key == "length":
            batch[key] = torch.from_numpy((n - 1).astype(np.int32)).view(B, 1)
            continue
        shape = col["shape"]
        if not col["is_sequence"]:
            x = rng.integers(0, col["input_dim"], size=(B,) + shape).astype(np.int32)
            batch[key] = torch.from_numpy(x)
            continue
        if col["type"] == "categorical":
            lo = 1 if col.get("primary_label", None) is not None else 0
            x = rng.integers(lo, col["input_dim"], size=(B, S) + shape).astype(np.int32)
            x = x * valid[:, :, None]
            batch[key] = torch.from_numpy(x.astype(np.int32))
        else:
            x = rng.standard_normal(size=(B, S) + shape).astype(np.float32)
            x /= np.linalg.norm(x, axis=-1, keepdims=True)  # CLIP-like unit rows
            x = x * valid[:, :, None]
            batch[key] = torch.from_numpy(x.astype(np.float32))
    if device != "cpu":
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


class _SyntheticDataset:
    """Iterable of synthetic batches; stands in for the reference's ``tf.data`` pipeline."""

    def __init__(self, input_columns, batch_size, seq_len, num_batches, seed, ragged, repeat,
                 device):
        self._args = (input_columns, batch_size, seq_len)
        self._num_batches = num_batches
        self._seed = seed
        self._ragged = ragged
        self._repeat = repeat
        self._device = device

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        i = 0
        while True:
            for b in range(self._num_batches):
                yield synthetic_batch(*self._args, seed=self._seed + b, ragged=self._ragged,
                                      device=self._device)
            i += 1
            if not self._repeat:
                return

    def __len__(self):
        return self._num_batches


class DataSpec(object):
    """Drop-in for the reference ``DataSpec(name, path, batch_size)`` (spec.py:24-76).

    ``path`` is either ``"synthetic"`` / ``"synthetic:<seq_len>[:<docs>]"`` or a directory;
    a directory is consulted for ``vocabulary.json`` / ``count.json`` (spec.py:74-86) and
    batches are still synthetic until the TFRecord reader lands.
    """

    def __init__(self, name, path, batch_size=8, seq_len: Optional[int] = None,
                 device: str = "cpu", ragged: bool = True):
        self._name = name
        self._path = path or "synthetic"
        self._batch_size = batch_size
        self._device = device
        self._ragged = ragged
        docs = None
        vocabulary = None
        if self._path.startswith("synthetic"):
            parts = self._path.split(":")
            if len(parts) > 1 and parts[1]:
                seq_len = seq_len or int(parts[1])
            if len(parts) > 2 and parts[2]:
                docs = int(parts[2])
        else:
            vpath = os.path.join(self._path, "vocabulary.json")
            if os.path.exists(vpath):
                with open(vpath) as f:
                    vocabulary = json.load(f)
            cpath = os.path.join(self._path, "count.json")
            if os.path.exists(cpath):
                with open(cpath) as f:
                    self._splits = json.load(f)
        self._seq_len = seq_len or 50
        if not hasattr(self, "_splits"):
            docs = docs or 4 * batch_size
            self._splits = {"train": docs, "val": max(batch_size, docs // 4),
                            "test": max(batch_size, docs // 4)}
        self._schema = _resolve_schema(name, vocabulary)
        self._vocabulary = vocabulary
        self._preprocessor = {k: c for k, c in self._schema if c["kind"] in ("lookup", "discretize")}

    @property
    def columns(self):
        return dict(self._schema)

    @property
    def preprocessor(self):
        return self._preprocessor

    def size(self, split):
        return self._splits[split]

    def steps_per_epoch(self, split, batch_size=None):
        return int(np.ceil(self.size(split) / (batch_size or self._batch_size)))

    def make_input_columns(self):
        return make_input_columns(self._name, self._vocabulary)

    def make_dataset(self, split, batch_size=None, shuffle=None, repeat=False, prefetch=None,
                     parallel=None, cache=None):
        assert split in self._splits, "split must be one of (%s)" % ", ".join(self._splits)
        bs = batch_size or self._batch_size
        seed = {"train": 0, "val": 100003, "test": 200003}.get(split, 300007)
        return _SyntheticDataset(self.make_input_columns(), bs, self._seq_len,
                                 self.steps_per_epoch(split, bs), seed, self._ragged, repeat,
                                 self._device)
