"""TensorFlow-free reader (and writer, for fixtures) of the reference's dataset files.

The reference feeds training from ``<data_dir>/<split>-*.tfrecord`` through
``tf.data.TFRecordDataset`` + ``tf.io.parse_sequence_example`` (reference
``src/mfp/mfp/data/spec.py:213-287``).  Neither TensorFlow nor protobuf is available to this
engine, and neither is needed: the two formats are small and public.

* **TFRecord framing** (tensorflow/core/lib/io/record_writer.cc): per record
  ``uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)``,
  little endian, ``masked(c) = ((c >> 15) | (c << 17)) + 0xa282ead8  (mod 2^32)``,
  CRC-32C = Castagnoli polynomial (reflected 0x82F63B78).
* **tf.train.SequenceExample** (tensorflow/core/example/example.proto, feature.proto), protobuf
  wire format::

      SequenceExample { Features context = 1; FeatureLists feature_lists = 2; }
      Features        { map<string, Feature> feature = 1; }
      FeatureLists    { map<string, FeatureList> feature_list = 1; }
      FeatureList     { repeated Feature feature = 1; }
      Feature         { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2;
                                     Int64List int64_list = 3; } }
      BytesList { repeated bytes value = 1; }  FloatList { repeated float value = 1 [packed]; }
      Int64List { repeated int64 value = 1 [packed]; }

  (a ``map<K,V>`` field is a repeated message ``{K key = 1; V value = 2;}``).

Only what the reader needs is implemented; unknown fields are skipped by wire type.
"""
from __future__ import annotations

import glob
import os
import struct
from typing import Dict, Iterable, Iterator, List, Tuple, Union

import numpy as np

# ------------------------------------------------------------------------------------ CRC-32C
_POLY = 0x82F63B78


def _make_table():
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (_POLY if c & 1 else 0)
        tab[i] = c
    return [int(x) for x in tab]


_TABLE = _make_table()


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli): crc32c(b"123456789") == 0xE3069283."""
    c = 0xFFFFFFFF
    tab = _TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- TFRecord framing
def read_records(path: str, check_crc: bool = True) -> Iterator[bytes]:
    """Yield the payload of every record of one TFRecord file."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("%s: truncated record header" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if check_crc and masked_crc32c(head[:8]) != lcrc:
                raise ValueError("%s: corrupted record length" % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError("%s: truncated record" % path)
            if check_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise ValueError("%s: corrupted record data" % path)
            yield data


def write_records(path: str, records: Iterable[bytes]) -> int:
    n = 0
    with open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head)
            f.write(struct.pack("<I", masked_crc32c(head)))
            f.write(data)
            f.write(struct.pack("<I", masked_crc32c(data)))
            n += 1
    return n


def list_split_files(data_dir: str, split: str) -> List[str]:
    """``<data_dir>/<split>-*.tfrecord`` (reference spec.py:228)."""
    return sorted(glob.glob(os.path.join(data_dir, split + "-*.tfrecord")))


# ------------------------------------------------------------------------- protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: bytes) -> Iterator[Tuple[int, int, Union[int, bytes]]]:
    """(field number, wire type, value) of one message; value is int (varint / fixed) or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def _parse_feature(buf: bytes):
    """Feature -> list of bytes | np.float32 array | np.int64 array."""
    for num, wt, val in _fields(buf):
        if wt != 2:
            continue
        if num == 1:      # BytesList
            return [v for n2, w2, v in _fields(val) if n2 == 1 and w2 == 2]
        if num == 2:      # FloatList: packed (wire type 2) or one fixed32 per element
            out = []
            for n2, w2, v in _fields(val):
                if n2 != 1:
                    continue
                out.append(np.frombuffer(v, dtype="<f4"))
            return np.concatenate(out) if out else np.zeros(0, np.float32)
        if num == 3:      # Int64List: packed varints or one varint per element
            out: List[int] = []
            for n2, w2, v in _fields(val):
                if n2 != 1:
                    continue
                if w2 == 0:
                    out.append(v)
                else:
                    p = 0
                    while p < len(v):
                        x, p = _varint(v, p)
                        out.append(x)
            arr = np.array(out, dtype=np.uint64).astype(np.int64)   # two's complement negatives
            return arr
    return []


def _parse_map(buf: bytes, value_parser) -> Dict[str, object]:
    out = {}
    for num, wt, entry in _fields(buf):
        if num != 1 or wt != 2:
            continue
        key, value = None, None
        for n2, w2, v in _fields(entry):
            if n2 == 1 and w2 == 2:
                key = v.decode("utf-8")
            elif n2 == 2 and w2 == 2:
                value = value_parser(v)
        if key is not None:
            out[key] = value if value is not None else []
    return out


def _parse_feature_list(buf: bytes):
    return [_parse_feature(v) for num, wt, v in _fields(buf) if num == 1 and wt == 2]


def parse_sequence_example(buf: bytes) -> Tuple[Dict[str, object], Dict[str, list]]:
    """-> (context: name -> values, feature_lists: name -> [values per step])."""
    context, lists = {}, {}
    for num, wt, val in _fields(buf):
        if wt != 2:
            continue
        if num == 1:
            context = _parse_map(val, _parse_feature)
        elif num == 2:
            lists = _parse_map(val, _parse_feature_list)
    return context, lists


# ----------------------------------------------------------------------------- encoder (fixtures)
def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_feature(values) -> bytes:
    if isinstance(values, np.ndarray) and values.dtype.kind == "f":
        return _enc_ld(2, _enc_ld(1, values.astype("<f4").tobytes()))
    if isinstance(values, np.ndarray) and values.dtype.kind in "iu":
        return _enc_ld(3, _enc_ld(1, b"".join(_enc_varint(int(x)) for x in values)))
    vals = list(values)
    if vals and isinstance(vals[0], (bytes, str)):
        return _enc_ld(1, b"".join(_enc_ld(1, v.encode() if isinstance(v, str) else v) for v in vals))
    if vals and isinstance(vals[0], float):
        return _enc_feature(np.asarray(vals, dtype=np.float32))
    return _enc_feature(np.asarray(vals, dtype=np.int64))


def encode_sequence_example(context: Dict[str, object], feature_lists: Dict[str, list]) -> bytes:
    ctx = b"".join(_enc_ld(1, _enc_ld(1, k.encode()) + _enc_ld(2, _enc_feature(v))) for k, v in context.items())
    fl = b"".join(
        _enc_ld(1, _enc_ld(1, k.encode()) + _enc_ld(2, b"".join(_enc_ld(1, _enc_feature(step)) for step in steps)))
        for k, steps in feature_lists.items())
    return _enc_ld(1, ctx) + _enc_ld(2, fl)
