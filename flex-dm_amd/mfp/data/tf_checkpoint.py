"""TensorFlow-free reader of TF2 checkpoints (``best.ckpt.index`` + ``best.ckpt.data-00000-of-00001``).

The reference restores pretrained weights with ``model.load_weights(path)`` (reference
``src/mfp/mfp/train.py:67-69``, ``eval.py:169-172``, ``notebooks/util.py:24-26``) and writes them
with ``model.save_weights`` / Keras ``ModelCheckpoint`` (``train.py:95-97``) in TensorFlow's
"tensor bundle" format.  TensorFlow is not available to this engine, and the format is small:

* ``<prefix>.index`` -- an SSTable in LevelDB's table format (tensorflow/core/lib/io/table*.cc):
  ``data blocks | metaindex block | index block | 48-byte footer``.  A block is a run of
  prefix-compressed entries ``varint shared | varint non_shared | varint value_len | key tail |
  value`` followed by ``uint32 restart[n] | uint32 n``, and on disk is trailed by one compression
  byte (0 = raw, 1 = snappy) and a masked CRC-32C.  The footer holds the (offset, size) handles of
  the metaindex and index blocks as varints, zero padding, and the magic ``0xdb4775248b80fb57``.
  Key ``""`` maps to a ``BundleHeaderProto``, every other key (a checkpoint key such as
  ``model/encoder/input_layer/left/embeddings/.ATTRIBUTES/VARIABLE_VALUE``) to a
  ``BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}``
  (tensorflow/core/protobuf/tensor_bundle.proto).
* ``<prefix>.data-SSSSS-of-NNNNN`` -- the tensors' little-endian bytes back to back.

Checkpoint keys follow the Keras object graph of the reference's model: attribute names, dict keys
for the dict-of-layers containers (``encoder.py:45``, ``transformer.py:256``, ``decoder.py:32``)
and ``layer_with_weights-N`` inside ``tf.keras.Sequential`` (``transformer.py:161-171``).
:func:`canonical_name` maps them onto this engine's variable names.  UNPINNED like the oracle: no
checkpoint of the reference exists in this environment, so the key grammar is restated from TF
2.8's object-graph naming rules, not checked against a real file; :func:`read_state_dict`
therefore matches strictly (every model variable exactly once, shapes equal) and accepts an
explicit ``name_map`` override.

The small writer at the bottom exists for tests and fixtures.
"""
from __future__ import annotations

import os
import re
import struct
from collections import OrderedDict
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np

from .tfrecord import _enc_ld, _enc_varint, _fields, _varint, masked_crc32c

TABLE_MAGIC = 0xDB4775248B80FB57
_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
DT_STRING, DT_BFLOAT16 = 7, 14
_DT_CODE = {np.dtype(v).str: k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------------------------- snappy
def snappy_decompress(buf: bytes) -> bytes:
    """Raw snappy block format (google/snappy format_description.txt)."""
    n, pos = _varint(buf, 0)
    out = bytearray()
    end = len(buf)
    while pos < end:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:                                   # copy, 1-byte offset
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:                                 # copy, 2-byte offset
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:                                           # copy, 4-byte offset
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("snappy: bad copy offset")
        start = len(out) - off
        if off >= ln:
            out += out[start:start + ln]
        else:                                           # overlapping copy = run-length fill
            for i in range(ln):
                out.append(out[start + i])
    if len(out) != n:
        raise ValueError("snappy: length mismatch (%d vs %d)" % (len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------------------ SSTable
def _read_block(data: bytes, offset: int, size: int, check_crc: bool) -> bytes:
    body = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(body) < size or len(trailer) < 5:
        raise ValueError("table: truncated block")
    if check_crc and masked_crc32c(body + trailer[:1]) != struct.unpack("<I", trailer[1:])[0]:
        raise ValueError("table: block checksum mismatch")
    if trailer[0] == 0:
        return body
    if trailer[0] == 1:
        return snappy_decompress(body)
    raise ValueError("table: unknown block compression %d" % trailer[0])


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise ValueError("table: block too small")
    nrestart = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * nrestart
    if limit < 0:
        raise ValueError("table: bad restart count")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key):
            raise ValueError("table: bad shared prefix length")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, check_crc: bool = True) -> "OrderedDict[bytes, bytes]":
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s: not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)          # metaindex handle (unused: no filter policy in bundles)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out: "OrderedDict[bytes, bytes]" = OrderedDict()
    for _, handle in _block_entries(_read_block(data, ioff, isize, check_crc)):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        for k, v in _block_entries(_read_block(data, boff, bsize, check_crc)):
            out[k] = v
    return out


# --------------------------------------------------------------------------------- the bundle
class BundleEntry:
    __slots__ = ("dtype", "shape", "shard_id", "offset", "size", "crc32c", "sliced")

    def __init__(self, buf: bytes):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c = 0, (), 0, 0, 0, None
        self.sliced = False
        for num, wt, val in _fields(buf):
            if num == 1 and wt == 0:
                self.dtype = val
            elif num == 2 and wt == 2:                   # TensorShapeProto { repeated Dim dim = 2 {size = 1} }
                dims = []
                for n2, w2, v2 in _fields(val):
                    if n2 == 2 and w2 == 2:
                        size = 0
                        for n3, w3, v3 in _fields(v2):
                            if n3 == 1 and w3 == 0:
                                size = v3
                        dims.append(size)
                self.shape = tuple(dims)
            elif num == 3 and wt == 0:
                self.shard_id = val
            elif num == 4 and wt == 0:
                self.offset = val
            elif num == 5 and wt == 0:
                self.size = val
            elif num == 6 and wt == 5:
                self.crc32c = struct.unpack("<I", val)[0]
            elif num == 7:
                self.sliced = True


class TFCheckpointReader:
    """``tf.train.load_checkpoint(prefix)`` look-alike: ``keys()``, ``get_tensor(key)``."""

    def __init__(self, prefix: str, check_crc: bool = True):
        index = prefix + ".index"
        if not os.path.exists(index):
            raise FileNotFoundError(index)
        self.prefix = prefix
        self.check_crc = check_crc
        table = read_table(index, check_crc)
        self.num_shards = 1
        header = table.get(b"")
        if header is not None:
            for num, wt, val in _fields(header):
                if num == 1 and wt == 0:
                    self.num_shards = val
                elif num == 2 and wt == 0 and val != 0:
                    raise ValueError("big-endian checkpoints are not supported")
        self.entries: "OrderedDict[str, BundleEntry]" = OrderedDict(
            (k.decode("utf-8"), BundleEntry(v)) for k, v in table.items() if k != b"")
        self._shards: Dict[int, np.memmap] = {}

    def keys(self) -> List[str]:
        return list(self.entries.keys())

    def variable_keys(self) -> List[str]:
        """Model variables only: no optimizer slots / counters / the object-graph proto."""
        return [k for k in self.entries if k.endswith(_SUFFIX) and "/.OPTIMIZER_SLOT/" not in k
                and not k.startswith("optimizer/") and not k.startswith("save_counter/")]

    def _shard(self, i: int):
        if i not in self._shards:
            path = "%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards)
            self._shards[i] = np.memmap(path, dtype=np.uint8, mode="r")
        return self._shards[i]

    def get_tensor(self, key: str) -> np.ndarray:
        e = self.entries[key]
        if e.sliced:
            raise ValueError("%s: partitioned variables are not supported" % key)
        raw = bytes(self._shard(e.shard_id)[e.offset:e.offset + e.size])
        if len(raw) != e.size:
            raise ValueError("%s: data shard is truncated" % key)
        if self.check_crc and e.crc32c is not None and e.dtype != DT_STRING and masked_crc32c(raw) != e.crc32c:
            raise ValueError("%s: tensor checksum mismatch" % key)
        if e.dtype == DT_BFLOAT16:
            u = np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16
            return u.view(np.float32).reshape(e.shape)
        if e.dtype == DT_STRING:
            raise ValueError("%s: string tensors are not model variables" % key)
        if e.dtype not in _DTYPES:
            raise ValueError("%s: unsupported dtype code %d" % (key, e.dtype))
        arr = np.frombuffer(raw, dtype=_DTYPES[e.dtype])
        if arr.size != int(np.prod(e.shape, dtype=np.int64)):
            raise ValueError("%s: %d elements for shape %s" % (key, arr.size, e.shape))
        return arr.reshape(e.shape)


# ----------------------------------------------------------------------- key grammar -> names
_RULES = [
    # PositionEmbedding holds its table in an attribute called "embeddings" (transformer.py:17)
    (re.compile(r"^encoder/input_layer/(const)/embeddings/(embeddings)$"), r"encoder/input_\1/\2"),
    (re.compile(r"^encoder/input_layer/([^/]+)/(embeddings|kernel|bias)$"), r"encoder/input_\1/\2"),
    (re.compile(r"^blocks/seq2seq/(seq2seq_\d+)/mlp/layer_with_weights-(\d+)/(kernel|bias)$"), r"blocks/\1/mlp/dense_\2/\3"),
    (re.compile(r"^blocks/seq2seq/(seq2seq_\d+)/(attn/[^/]+/(?:kernel|bias)|norm[12]/(?:gamma|beta))$"), r"blocks/\1/\2"),
    (re.compile(r"^decoder/decoders/([^/]+)/(kernel|bias)$"), r"decoder/decoder_\1/\2"),
]


def _unescape(component: str) -> str:
    # object-graph edge names escape "." as ".." and then "/" as ".S" (trackable/base.py)
    out, i = [], 0
    while i < len(component):
        if component[i] == "." and i + 1 < len(component) and component[i + 1] in ".S":
            out.append("." if component[i + 1] == "." else "/")
            i += 2
        else:
            out.append(component[i])
            i += 1
    return "".join(out)


def canonical_name(key: str) -> Optional[str]:
    """Checkpoint key -> this engine's variable name, or None if the key is not a model variable.

    ``MFP.model`` is the attribute that holds the network (reference mfp.py:231-284), so keys saved
    from the outer model start with ``model/``; keys saved from the inner model do not.
    """
    if not key.endswith(_SUFFIX):
        return None
    path = key[:-len(_SUFFIX)]
    if "/.OPTIMIZER_SLOT/" in path or path.startswith("optimizer/") or path.startswith("save_counter"):
        return None
    if path.startswith("model/"):
        path = path[len("model/"):]
    for rx, repl in _RULES:
        m = rx.match(path)
        if m:   # dict keys (attribute names of the dataset) may carry escaped "." or "/"
            return re.sub(r"\\(\d)", lambda g: _unescape(m.group(int(g.group(1)))), repl)
    return None


def checkpoint_key(name: str, outer: bool = True) -> str:
    """Inverse of :func:`canonical_name` (fixtures / documentation)."""
    m = re.match(r"^encoder/input_(.+)/(embeddings|kernel|bias)$", name)
    if name == "encoder/input_const/embeddings":
        path = "encoder/input_layer/const/embeddings/embeddings"
    elif m:
        path = "encoder/input_layer/%s/%s" % m.groups()
    else:
        m = re.match(r"^blocks/(seq2seq_\d+)/mlp/dense_(\d+)/(kernel|bias)$", name)
        if m:
            path = "blocks/seq2seq/%s/mlp/layer_with_weights-%s/%s" % m.groups()
        else:
            m = re.match(r"^blocks/(seq2seq_\d+)/(.+)$", name)
            if m:
                path = "blocks/seq2seq/%s/%s" % m.groups()
            else:
                m = re.match(r"^decoder/decoder_(.+)/(kernel|bias)$", name)
                if not m:
                    raise KeyError(name)
                path = "decoder/decoders/%s/%s" % m.groups()
    return ("model/" if outer else "") + path + _SUFFIX


def read_state_dict(prefix: str, expected: Optional[Dict[str, Tuple[int, ...]]] = None,
                    name_map: Optional[Dict[str, str]] = None, check_crc: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Variables of a TF checkpoint under this engine's names (Keras layouts: kernel ``(in, out)``).

    ``expected``: name -> shape of every variable the model needs; all must be found exactly once
    with that shape, and model-variable keys of the checkpoint that map to nothing are an error.
    ``name_map``: explicit ``checkpoint key -> variable name`` overrides (applied first).
    """
    reader = TFCheckpointReader(prefix, check_crc)
    name_map = name_map or {}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    unknown = []
    for key in reader.variable_keys():
        name = name_map.get(key) or canonical_name(key)
        if name is None:
            unknown.append(key)
            continue
        if name in out:
            raise ValueError("%s: two checkpoint keys map to %s" % (prefix, name))
        if expected is not None and name not in expected:
            unknown.append(key)
            continue
        out[name] = np.array(reader.get_tensor(key), dtype=np.float32)   # writable copy
    if expected is not None:
        missing = [n for n in expected if n not in out]
        bad = [(n, out[n].shape, tuple(expected[n])) for n in out if tuple(out[n].shape) != tuple(expected[n])]
        if missing or unknown or bad:
            raise ValueError(
                "%s does not match the model: missing %s; unmatched checkpoint keys %s; shape mismatches %s "
                "(pass name_map={checkpoint key: variable name} to override the key grammar)"
                % (prefix, missing[:8], unknown[:8], bad[:8]))
    return out


# ------------------------------------------------------------------------ writer (fixtures/tests)
def _build_block(entries: Iterable[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts, last, n = [], b"", 0
    for key, val in entries:
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(val)) + key[shared:] + val
        last = key
        n += 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def snappy_literal(raw: bytes) -> bytes:
    """A valid (if pointless) snappy stream of ``raw``: literals only, <= 60 bytes each."""
    out = bytearray(_enc_varint(len(raw)))
    for i in range(0, len(raw), 60):
        piece = raw[i:i + 60]
        out.append((len(piece) - 1) << 2)
        out += piece
    return bytes(out)


def write_table(path: str, items: "Dict[bytes, bytes]", block_size: int = 4096, snappy: bool = False) -> None:
    keys = sorted(items)
    blob = bytearray()
    index: List[Tuple[bytes, bytes]] = []

    def emit(block: bytes) -> bytes:
        off = len(blob)
        kind = b"\x01" if snappy else b"\x00"
        if snappy:
            block = snappy_literal(block)
        blob.extend(block)
        blob.extend(kind + struct.pack("<I", masked_crc32c(block + kind)))
        return _enc_varint(off) + _enc_varint(len(block))

    cur: List[Tuple[bytes, bytes]] = []
    size = 0
    for k in keys:
        cur.append((k, items[k]))
        size += len(k) + len(items[k]) + 3
        if size >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, size = [], 0
    if cur:
        index.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, restart_interval=1))
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(path, "wb") as f:
        f.write(bytes(blob) + footer)


def write_bundle(prefix: str, tensors: "Dict[str, np.ndarray]", snappy: bool = False) -> None:
    """Single-shard tensor bundle holding ``tensors`` under the given checkpoint keys."""
    items: Dict[bytes, bytes] = {b"": _enc_varint(1 << 3) + _enc_varint(1)            # num_shards = 1
                                 + _enc_ld(3, _enc_varint(1 << 3) + _enc_varint(1))}  # version.producer = 1
    data = bytearray()
    for key in sorted(tensors):
        arr = np.asarray(tensors[key], order="C")
        code = _DT_CODE.get(arr.dtype.str)
        if code is None:
            raise ValueError("%s: dtype %s not supported" % (key, arr.dtype))
        raw = arr.tobytes()
        shape = b"".join(_enc_ld(2, _enc_varint(1 << 3) + _enc_varint(int(d))) for d in arr.shape)
        entry = _enc_varint(1 << 3) + _enc_varint(code) + _enc_ld(2, shape)
        entry += _enc_varint(4 << 3) + _enc_varint(len(data)) + _enc_varint(5 << 3) + _enc_varint(len(raw))
        entry += _enc_varint((6 << 3) | 5) + struct.pack("<I", masked_crc32c(raw))
        items[key.encode("utf-8")] = entry
        data += raw
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    write_table(prefix + ".index", items, snappy=snappy)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
