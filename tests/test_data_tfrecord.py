"""TensorFlow-free dataset path: TFRecord framing, SequenceExample wire
format, the Keras-layer preprocessing semantics and DataSpec.make_dataset over real files."""
import os
import struct

import numpy as np
import pytest
import torch

from mfp.data import tfrecord
from mfp.data.spec import DataSpec, _resolve_schema, discretize, lookup_indices, write_synthetic_tfrecords


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 / the usual CRC-32C check values
    assert tfrecord.crc32c(b"123456789") == 0xE3069283
    assert tfrecord.crc32c(b"") == 0
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tfrecord.crc32c(bytes(range(32))) == 0x46DD794E
    c = tfrecord.crc32c(b"abc")
    assert tfrecord.masked_crc32c(b"abc") == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


def test_record_framing_roundtrip_and_corruption(tmp_path):
    path = str(tmp_path / "x.tfrecord")
    recs = [b"", b"a", os.urandom(1000), b"\x00" * 17]
    assert tfrecord.write_records(path, recs) == 4
    assert list(tfrecord.read_records(path)) == recs
    raw = bytearray(open(path, "rb").read())
    # layout of the second record: 8 length + 4 crc + 1 data + 4 crc after the 16-byte empty first one
    assert struct.unpack("<Q", raw[16:24])[0] == 1
    raw[28] ^= 0x01                      # flip a data bit of record 2
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="corrupted record data"):
        list(tfrecord.read_records(path))
    assert len(list(tfrecord.read_records(path, check_crc=False))) == 4
    open(path, "wb").write(bytes(raw[:20]))
    with pytest.raises(ValueError, match="truncated"):
        list(tfrecord.read_records(path))


def test_sequence_example_wire_format():
    # hand-assembled bytes: context {"n": int64 [3]}, feature_lists {"x": [[1.5], [-2.0]]}
    f_int = bytes([0x1A, 0x03, 0x0A, 0x01, 0x03])                       # Feature{int64_list{value:[3] packed}}
    ctx = bytes([0x0A, 0x0A, 0x0A, 0x01]) + b"n" + bytes([0x12, len(f_int)]) + f_int
    def f_float(v):
        p = struct.pack("<f", v)
        return bytes([0x12, 0x06, 0x0A, 0x04]) + p                     # Feature{float_list{value packed}}
    steps = b"".join(bytes([0x0A, 0x08]) + f_float(v) for v in (1.5, -2.0))
    fl = bytes([0x0A, 3 + 2 + len(steps), 0x0A, 0x01]) + b"x" + bytes([0x12, len(steps)]) + steps
    buf = bytes([0x0A, len(ctx)]) + ctx + bytes([0x12, len(fl)]) + fl
    context, lists = tfrecord.parse_sequence_example(buf)
    assert list(context["n"]) == [3]
    assert [list(s) for s in lists["x"]] == [[1.5], [-2.0]]
    # encoder <-> parser, incl. bytes, negative int64 and unpacked int64 elements
    enc = tfrecord.encode_sequence_example({"id": [b"doc"], "k": np.array([-1, 2 ** 40], dtype=np.int64)},
                                           {"t": [[b"a"], [b""]], "f": [np.array([0.25, 0.5], np.float32)]})
    context, lists = tfrecord.parse_sequence_example(enc)
    assert context["id"] == [b"doc"] and list(context["k"]) == [-1, 2 ** 40]
    assert lists["t"] == [[b"a"], [b""]] and list(lists["f"][0]) == [0.25, 0.5]
    unpacked = bytes([0x1A, 0x04, 0x08, 0x05, 0x08, 0x07])                # int64_list{value:5, value:7} unpacked
    assert list(tfrecord._parse_feature(unpacked)) == [5, 7]


def test_preprocessing_semantics():
    # Discretization(linspace(0,1,64)[1:]): 64 classes, boundaries inclusive on the left
    b = np.linspace(0.0, 1.0, 64)[1:]
    x = np.array([0.0, b[0] - 1e-6, b[0], 0.5, 1.0, 2.0, -1.0], dtype=np.float32)
    assert discretize(x, 0.0, 1.0, 64).tolist() == [0, 0, 1, int(np.searchsorted(b, np.float32(0.5), "right")), 63, 63, 0]
    assert discretize(np.array([0, 16, 17, 255]), 0.0, 255.0, 16).tolist() == [0, 0, 1, 15]
    # lookups: mask token first / OOV slot first / strict
    assert lookup_indices([b"", b"textElement"], ["", "coloredBackground", "textElement"], "type").tolist() == [0, 2]
    assert lookup_indices([b"nope", b"f1"], ["[UNK]", "f0", "f1"], "font").tolist() == [0, 2]
    assert lookup_indices(np.array([7, 1]), [-1, 1, 7], "w").tolist() == [2, 1]
    with pytest.raises(ValueError, match="no OOV slot"):
        lookup_indices([b"zzz"], ["", "a"], "type")


@pytest.mark.parametrize("name", ["crello", "rico"])
def test_dataspec_reads_tfrecords(tmp_path, name):
    d = str(tmp_path / name)
    raw = write_synthetic_tfrecords(d, name, {"train": 11, "val": 3, "test": 3}, seq_len=6, seed=4)
    spec = DataSpec(name, d, batch_size=4)
    assert spec.size("train") == 11 and spec.steps_per_epoch("train") == 3
    ic = spec.make_input_columns()
    schema = dict(_resolve_schema(name, None))
    batches = list(spec.make_dataset("train", shuffle=False))
    assert [b["length"].shape[0] for b in batches] == [4, 4, 3] and len(spec.make_dataset("train")) == 3
    docs = raw["train"]
    i = 0
    for batch in batches:
        B = batch["length"].shape[0]
        S = max(len(next(iter(l.values()))) for _, l in docs[i:i + B])
        for key, col in ic.items():
            if col.get("demo_only"):
                assert key not in batch
                continue
            t = batch[key]
            if col["is_sequence"]:
                assert t.shape == (B, S) + tuple(col["shape"]), key
            assert t.dtype == (torch.float32 if col["type"] == "numerical" else torch.int32), key
            if col["type"] == "categorical":
                assert int(t.min()) >= 0 and int(t.max()) < col["input_dim"], key
        for b in range(B):
            ctx, lists = docs[i + b]
            n = len(lists["left"])
            assert int(batch["length"][b, 0]) == n - 1                   # zero-based (IntegerLookup over 1..50)
            assert torch.equal(batch["left"][b, :n, 0],
                               torch.from_numpy(discretize(np.concatenate(lists["left"]), 0.0, 1.0, 64)))
            assert int(batch["left"][b, n:].abs().sum()) == 0           # zero padding past the document
            want_type = [schema["type"]["vocab"].index(s[0]) for s in lists["type"]]
            assert batch["type"][b, :n, 0].tolist() == want_type and int(batch["type"][b, n:].sum()) == 0
            if name == "crello":
                assert torch.allclose(batch["image_embedding"][b, :n], torch.from_numpy(np.stack(lists["image_embedding"])))
                assert torch.equal(batch["color"][b, :n],
                                   torch.from_numpy(discretize(np.stack(lists["color"]), 0.0, 255.0, 16)))
        i += B
    # shuffled epochs visit every document once
    seen = sorted(int(x) for b in spec.make_dataset("train", shuffle=True) for x in b["length"][:, 0])
    assert seen == sorted(len(l["left"]) - 1 for _, l in docs)
