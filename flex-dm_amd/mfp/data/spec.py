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

Batches are *synthetic* (distributions of SURVEY.md §8d) unless ``path`` holds the reference's
``<split>-*.tfrecord`` files: those are read without TensorFlow by ``mfp.data.tfrecord`` and
preprocessed here with the semantics of the reference's Keras layers (SURVEY.md §8f row 3;
reference spec.py:88-134,213-287, discretizer.py:5-31).  The schema is held as Python tables
below rather than YAML; the values
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
        ("color", dict(_disc(16, shape=(3,)), min=0.0, max=255.0, raw="int64",   # crello-spec.yml:80-92
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


# --------------------------------------------------------------------------- real data (TFRecord)
def _raw_kind(col: Dict) -> str:
    """Feature type a column is stored with in the SequenceExample (the spec's ``dtype``)."""
    if "raw" in col:
        return col["raw"]
    k = col["kind"]
    if k == "demo":
        return "bytes"
    if k == "lookup":
        return "bytes" if any(isinstance(v, str) for v in col["vocab"]) else "int64"
    if k in ("discretize", "float"):
        return "float"
    return "int64"


def lookup_indices(values, vocab: List, name: str = "") -> np.ndarray:
    """Keras StringLookup / IntegerLookup as the reference configures them (spec.py:103-134): the index
    of a token is its position in ``get_vocabulary()`` (= ``vocab`` here: mask token or OOV token
    first when the layer has one); tokens outside it map to the OOV slot 0 when the vocabulary
    starts with ``[UNK]`` / ``-1`` and are an error otherwise (``num_oov_indices: 0``)."""
    table = {(v.encode() if isinstance(v, str) else v): i for i, v in enumerate(vocab)}
    has_oov = len(vocab) > 0 and vocab[0] in ("[UNK]", -1)
    flat = np.asarray(values, dtype=object).reshape(-1)
    out = np.empty(flat.shape, dtype=np.int32)
    for i, v in enumerate(flat):
        if isinstance(v, (np.integer,)):
            v = int(v)
        j = table.get(v)
        if j is None:
            if not has_oov:
                raise ValueError("column %s: %r is not in the vocabulary and the lookup has no OOV slot" % (name, v))
            j = 0
        out[i] = j
    return out.reshape(np.asarray(values, dtype=object).shape)


def discretize(x: np.ndarray, lo: float, hi: float, bins: int) -> np.ndarray:
    """Keras Discretization(linspace(lo, hi, bins)[1:]) (spec.py:95-100): class = number of
    boundaries <= x (tf Bucketize), i.e. 0 .. bins-1."""
    boundaries = np.linspace(lo, hi, bins)[1:]
    return np.searchsorted(boundaries, np.asarray(x, dtype=np.float32), side="right").astype(np.int32)


def parse_examples(records: List[bytes], schema, device: str = "cpu", include_demo: bool = False,
                   seq_len: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """``tf.io.parse_sequence_example`` on a batch of serialized records + the per-column
    preprocessing + the int64 -> int32 cast of reference spec.py:255-285.  Sequence features are
    zero / empty-string padded to the longest document of the batch BEFORE preprocessing, as the
    batched TF parser does.  demo_only columns come back as Python lists of bytes."""
    from mfp.data import tfrecord
    parsed = [tfrecord.parse_sequence_example(r) for r in records]
    B = len(parsed)
    S = 1
    for _, lists in parsed:
        for steps in lists.values():
            S = max(S, len(steps))
    if isinstance(seq_len, (tuple, list)):      # buckets (mfp.train's default): the smallest that holds the batch, else its own length
        S = next((b for b in sorted(seq_len) if b >= S), S)
    elif seq_len is not None:       # fixed-shape batches (hipGraph replay): pad every batch to seq_len
        if S > seq_len:
            raise ValueError("document of %d elements does not fit seq_len=%d" % (S, seq_len))
        S = seq_len
    out: Dict[str, torch.Tensor] = {}
    for key, col in schema:
        if col["kind"] == "demo" and not include_demo:
            continue
        raw, width = _raw_kind(col), int(np.prod(col.get("shape", (1,))))
        pad = b"" if raw == "bytes" else 0
        is_seq = bool(col.get("is_sequence", False))
        if is_seq:
            arr = np.empty((B, S, width), dtype=object if raw == "bytes" else (np.float32 if raw == "float" else np.int64))
            arr[...] = pad
            for b, (_, lists) in enumerate(parsed):
                for t, step in enumerate(lists.get(key, [])):
                    vals = list(step)
                    if len(vals) != width:
                        raise ValueError("column %s: step of %d values, expected %d" % (key, len(vals), width))
                    arr[b, t, :] = vals
        else:
            arr = np.empty((B, width), dtype=object if raw == "bytes" else (np.float32 if raw == "float" else np.int64))
            arr[...] = pad
            for b, (ctx, _) in enumerate(parsed):
                vals = list(ctx.get(key, []))
                if len(vals) != width:
                    raise ValueError("column %s: %d context values, expected %d" % (key, len(vals), width))
                arr[b, :] = vals
        kind = col["kind"]
        if kind == "demo":
            out[key] = arr.tolist()
            continue
        if kind == "lookup":
            arr = lookup_indices(arr, col["vocab"], key)
        elif kind == "discretize":
            arr = discretize(arr, col.get("min", 0.0), col.get("max", 1.0), col["bins"])
        elif kind == "float":
            arr = arr.astype(np.float32)
        else:
            arr = arr.astype(np.int32)
        out[key] = torch.from_numpy(np.ascontiguousarray(arr))
    if device != "cpu":
        out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in out.items()}
    return out


class _TFRecordDataset:
    """Iterable over the batches of ``<path>/<split>-*.tfrecord`` (reference make_dataset,
    spec.py:213-253): list files, (shuffle), (repeat), batch, parse + preprocess."""

    def __init__(self, files, schema, batch_size, shuffle, repeat, device, seed=0, seq_len=None):
        self._seq_len = seq_len
        self._files, self._schema, self._bs = files, schema, batch_size
        self._shuffle, self._repeat, self._device = bool(shuffle), repeat, device
        self._rng = np.random.default_rng(seed)
        self._count = None

    def _records(self) -> List[bytes]:
        from mfp.data import tfrecord
        recs: List[bytes] = []
        for f in self._files:
            recs.extend(tfrecord.read_records(f))
        return recs

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        # shuffle -> repeat -> batch (reference spec.py:244-248): batching runs over the REPEATED
        # record stream, so with repeat=True a batch straddles pass boundaries and every batch is
        # full; only a non-repeated pass ends with a short batch
        recs = self._records()
        self._count = len(recs)
        pending: List[bytes] = []
        while recs:
            order = self._rng.permutation(len(recs)) if self._shuffle else np.arange(len(recs))
            for j in order:
                pending.append(recs[j])
                if len(pending) == self._bs:
                    yield parse_examples(pending, self._schema, self._device, seq_len=self._seq_len)
                    pending = []
            if not self._repeat:
                break
        if pending:
            yield parse_examples(pending, self._schema, self._device, seq_len=self._seq_len)

    def __len__(self):
        if self._count is None:
            self._count = len(self._records())
        return int(np.ceil(self._count / self._bs))


def write_synthetic_tfrecords(data_dir: str, name: str, docs: Dict[str, int], seq_len: int = 12, seed: int = 0,
                              shards: int = 2) -> Dict[str, list]:
    """Write ``<split>-0000i.tfrecord`` shards + ``count.json`` with random RAW documents (strings,
    unbucketed floats) of dataset ``name`` -- the format the reference's converters produce.
    Returns the raw documents per split (for tests)."""
    from mfp.data import tfrecord
    rng = np.random.default_rng(seed)
    schema = _resolve_schema(name, None)
    os.makedirs(data_dir, exist_ok=True)
    raw_docs: Dict[str, list] = {}
    for split, n_docs in docs.items():
        out = []
        for d in range(n_docs):
            n = int(rng.integers(1, seq_len + 1))
            ctx, lists = {}, {}
            for key, col in schema:
                raw, width = _raw_kind(col), int(np.prod(col.get("shape", (1,))))

                def draw():
                    if key == "length":
                        return np.array([n], dtype=np.int64)
                    if col["kind"] == "lookup":
                        vocab = [v for v in col["vocab"] if v not in ("", "[UNK]", -1)]
                        pick = [vocab[int(i)] for i in rng.integers(0, len(vocab), size=width)]
                        return pick if raw == "bytes" else np.array(pick, dtype=np.int64)
                    if col["kind"] == "demo":
                        return ["doc%d" % d] * width
                    if col["kind"] == "int":
                        return rng.integers(0, col["max"] + 1, size=width).astype(np.int64)
                    if col["kind"] == "discretize":
                        lo, hi = col.get("min", 0.0), col.get("max", 1.0)
                        if raw == "int64":
                            return rng.integers(int(lo), int(hi) + 1, size=width).astype(np.int64)
                        return (lo + (hi - lo) * rng.random(width)).astype(np.float32)
                    v = rng.standard_normal(width).astype(np.float32)
                    return v / np.linalg.norm(v)
                if col.get("is_sequence", False):
                    lists[key] = [draw() for _ in range(n)]
                else:
                    ctx[key] = draw()
            out.append((ctx, lists))
        raw_docs[split] = out
        per = int(np.ceil(n_docs / shards))
        for sh in range(shards):
            chunk = out[sh * per:(sh + 1) * per]
            tfrecord.write_records(os.path.join(data_dir, "%s-%05d.tfrecord" % (split, sh)),
                                   (tfrecord.encode_sequence_example(c, l) for c, l in chunk))
    with open(os.path.join(data_dir, "count.json"), "w") as f:
        json.dump({k: v for k, v in docs.items()}, f)
    return raw_docs


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
    a directory is consulted for ``vocabulary.json`` / ``count.json`` (spec.py:74-86) and, when it
    holds ``<split>-*.tfrecord`` files, ``make_dataset`` reads them (TensorFlow-free reader,
    ``mfp.data.tfrecord``); otherwise batches are synthetic.
    """

    def __init__(self, name, path, batch_size=8, seq_len=None,
                 device: str = "cpu", ragged: bool = True):
        # seq_len: None = pad each batch to its longest document (the reference, spec.py:255-276); an int = pad every batch to
        # it; a tuple of buckets, e.g. (64, 128) = pad each batch to the smallest bucket that holds it (padding is inert: masked
        # keys, masked losses) so that it lands on the document-tile kernels
        buckets = tuple(seq_len) if isinstance(seq_len, (tuple, list)) else None
        if buckets:
            seq_len = None
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
        self._fixed_seq_len = buckets or seq_len      # None: pad each batch to its longest document (the reference)
        self._seq_len = seq_len or (min(buckets) if buckets else 50)
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
        if not self._path.startswith("synthetic"):
            from mfp.data import tfrecord
            files = tfrecord.list_split_files(self._path, split)
            if files:
                logger.info("TFRecord from %s (%d files)", os.path.join(self._path, split + "-*.tfrecord"), len(files))
                return _TFRecordDataset(files, self._schema, bs, shuffle, repeat, self._device,
                                        seq_len=self._fixed_seq_len)
        seed = {"train": 0, "val": 100003, "test": 200003}.get(split, 300007)
        return _SyntheticDataset(self.make_input_columns(), bs, self._seq_len,
                                 self.steps_per_epoch(split, bs), seed, self._ragged, repeat,
                                 self._device)
