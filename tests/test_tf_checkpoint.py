"""TensorFlow-free TF2 checkpoint reader: LevelDB table format, snappy,
tensor-bundle entries and the Keras object-graph key grammar of the reference's model."""
import os
import struct

import numpy as np
import pytest

from mfp.data import tf_checkpoint as tfc
from mfp.data.tfrecord import masked_crc32c


def test_snappy_known_streams():
    # format_description.txt: literal "abcd", copy(1-byte offset, len 4, off 4), copy(2-byte, len 6, off 2)
    stream = bytes([14]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([0b00000001, 4]) + bytes([((6 - 1) << 2) | 2, 2, 0])
    assert tfc.snappy_decompress(stream) == b"abcdabcdcdcdcd"
    # long literal (length in one extra byte) + overlapping run-length copy
    lit = bytes(range(100))
    stream = bytes([164, 1]) + bytes([60 << 2, 99]) + lit + bytes([((64 - 1) << 2) | 2, 1, 0])
    assert tfc.snappy_decompress(stream) == lit + bytes([99]) * 64
    # 4-byte-offset copy
    stream = bytes([8]) + bytes([(4 - 1) << 2]) + b"wxyz" + bytes([((4 - 1) << 2) | 3, 4, 0, 0, 0])
    assert tfc.snappy_decompress(stream) == b"wxyzwxyz"
    with pytest.raises(ValueError):
        tfc.snappy_decompress(bytes([8]) + bytes([(4 - 1) << 2]) + b"wxyz" + bytes([((4 - 1) << 2) | 2, 9, 0]))
    raw = os.urandom(1000)
    assert tfc.snappy_decompress(tfc.snappy_literal(raw)) == raw


@pytest.mark.parametrize("snappy", [False, True])
def test_table_round_trip_multi_block(tmp_path, snappy):
    items = {b"": b"header"}
    for i in range(500):   # long shared prefixes -> prefix compression, several 4 KB blocks
        items[("model/blocks/seq2seq/seq2seq_%d/attn/dense_%03d/kernel" % (i % 7, i)).encode()] = os.urandom(i % 40)
    path = str(tmp_path / "t.index")
    tfc.write_table(path, items, snappy=snappy)
    got = tfc.read_table(path)
    assert list(got.keys()) == sorted(items) and all(got[k] == items[k] for k in items)
    blob = bytearray(open(path, "rb").read())
    assert struct.unpack("<Q", blob[-8:])[0] == 0xDB4775248B80FB57
    blob[10] ^= 1                     # corrupt the first data block
    open(path, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_table(path)
    open(path, "wb").write(b"not a table" * 10)
    with pytest.raises(ValueError, match="magic"):
        tfc.read_table(path)


def test_hand_assembled_table_block():
    """A block written byte by byte from the format description (not by write_table)."""
    e = lambda shared, tail, val: bytes([shared, len(tail), len(val)]) + tail + val
    block = e(0, b"apple", b"1") + e(3, b"ly", b"22") + e(0, b"bat", b"") + struct.pack("<II", 0, 1)
    assert list(tfc._block_entries(block)) == [(b"apple", b"1"), (b"apply", b"22"), (b"bat", b"")]


def _model_tensors(dataset, D, L, seed=0):
    from oracle import np_ref
    from mfp.data.spec import make_input_columns
    ic = make_input_columns(dataset)
    return ic, np_ref.init_params(ic, D, L, seed=seed)


@pytest.mark.parametrize("dataset,outer", [("crello", True), ("rico", False)])
def test_bundle_round_trip_with_reference_key_grammar(tmp_path, dataset, outer):
    ic, params = _model_tensors(dataset, 16, 2)
    tensors = {tfc.checkpoint_key(k, outer): v.astype(np.float32) for k, v in params.items()}
    assert ("model/" if outer else "") + "blocks/seq2seq/seq2seq_1/mlp/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE" in tensors
    assert ("model/" if outer else "") + "encoder/input_layer/left/embeddings/.ATTRIBUTES/VARIABLE_VALUE" in tensors
    assert ("model/" if outer else "") + "decoder/decoders/type/kernel/.ATTRIBUTES/VARIABLE_VALUE" in tensors
    # what a Keras checkpoint carries besides the variables
    first = next(iter(tensors))
    tensors[first.replace("/.ATTRIBUTES", "/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES")] = np.zeros_like(tensors[first])
    tensors["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(7, np.int64)
    tensors["save_counter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(1, np.int64)
    prefix = str(tmp_path / "ckpt" / "best.ckpt")
    tfc.write_bundle(prefix, tensors, snappy=outer)
    reader = tfc.TFCheckpointReader(prefix)
    assert set(reader.keys()) == set(tensors)
    assert int(reader.get_tensor("optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE")) == 7
    expected = {k: v.shape for k, v in params.items()}
    got = tfc.read_state_dict(prefix, expected)
    assert list(sorted(got)) == sorted(params)
    for k, v in params.items():
        np.testing.assert_array_equal(got[k], v.astype(np.float32))
    # strictness: a missing variable, a foreign variable and a wrong shape are all errors
    with pytest.raises(ValueError, match="missing"):
        tfc.read_state_dict(prefix, dict(expected, **{"decoder/decoder_extra/bias": (3,)}))
    fewer = dict(expected)
    fewer.pop("decoder/decoder_type/bias")
    with pytest.raises(ValueError, match="unmatched"):
        tfc.read_state_dict(prefix, fewer)
    with pytest.raises(ValueError, match="shape"):
        tfc.read_state_dict(prefix, dict(expected, **{"decoder/decoder_type/bias": (99,)}))
    # corrupt one tensor byte -> checksum error
    shard = prefix + ".data-00000-of-00001"
    blob = bytearray(open(shard, "rb").read())
    blob[len(blob) // 2] ^= 0x40
    open(shard, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_state_dict(prefix, expected)


def test_name_map_override_and_entry_fields(tmp_path):
    prefix = str(tmp_path / "x.ckpt")
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    tfc.write_bundle(prefix, {"weird/path/.ATTRIBUTES/VARIABLE_VALUE": a,
                              "model/decoder/decoders/left/bias/.ATTRIBUTES/VARIABLE_VALUE": np.ones(4, np.float32)})
    r = tfc.TFCheckpointReader(prefix)
    e = r.entries["weird/path/.ATTRIBUTES/VARIABLE_VALUE"]
    assert (e.dtype, e.shape, e.shard_id, e.size) == (1, (2, 3), 0, 24) and e.crc32c == masked_crc32c(a.tobytes())
    with pytest.raises(ValueError, match="unmatched"):
        tfc.read_state_dict(prefix, {"decoder/decoder_left/bias": (4,), "encoder/input_x/kernel": (2, 3)})
    got = tfc.read_state_dict(prefix, {"decoder/decoder_left/bias": (4,), "encoder/input_x/kernel": (2, 3)},
                              name_map={"weird/path/.ATTRIBUTES/VARIABLE_VALUE": "encoder/input_x/kernel"})
    np.testing.assert_array_equal(got["encoder/input_x/kernel"], a)
    assert tfc.canonical_name("model/encoder/input_layer/a..b.Sc/kernel" + tfc._SUFFIX) == "encoder/input_a.b/c/kernel"
    assert tfc.canonical_name("_CHECKPOINTABLE_OBJECT_GRAPH") is None
    # PositionEmbedding of the shuffled_set input type keeps its table in an attribute "embeddings"
    k = "model/encoder/input_layer/const/embeddings/embeddings" + tfc._SUFFIX
    assert tfc.canonical_name(k) == "encoder/input_const/embeddings"
    assert tfc.checkpoint_key("encoder/input_const/embeddings") == k
