"""TensorFlow checkpoint-bundle reader/writer (nmrgnn_amd/tfbundle.py), pinned by the reference's own
bundle index (tests/golden/bundle_index.json <- nmrgnn/models/baseline/variables/variables.index)."""
import hashlib
import json
import os
import struct
import zlib

import numpy as np
import pytest

from nmrgnn_amd import tfbundle as tb

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref_index():
    with open(os.path.join(HERE, "golden", "bundle_index.json")) as f:
        doc = json.load(f)
    doc["items"] = [(k.encode(), bytes.fromhex(v)) for k, v in doc["entries"]]
    return doc


def test_table_writer_reproduces_reference_index_bytes(ref_index):
    blob = tb.build_table(ref_index["items"])
    assert len(blob) == ref_index["length"]
    assert hashlib.sha256(blob).hexdigest() == ref_index["sha256"]
    assert tb.read_table(blob, verify=True) == ref_index["items"]       # block CRCs verify too


def test_entry_codec_round_trips_every_reference_entry(ref_index):
    assert ref_index["items"][0] == (b"", tb.HEADER_VALUE)
    for k, v in ref_index["items"][1:]:
        e = tb.decode_entry(v)
        assert tb.encode_entry(e) == v, k


def test_crc32c_against_checksums_stored_in_reference_index(ref_index):
    ent = {k.decode(): tb.decode_entry(v) for k, v in ref_index["items"][1:]}
    known = {"optimizer/decay": np.float32(0.0), "optimizer/beta_1": np.float32(0.9),
             "optimizer/beta_2": np.float32(0.999), "optimizer/learning_rate": np.float32(1e-4)}
    for name, val in known.items():        # Keras Adam defaults and model.py:44 (lr 1e-4)
        e = ent[name + tb.VALUE_SUFFIX]
        assert e.size == 4 and tb.mask_crc(tb.crc32c(val.tobytes())) == e.crc32c, name
    assert tb.crc32c(b"123456789") == 0xE3069283                      # the standard CRC-32C check value
    assert tb.unmask_crc(tb.mask_crc(0xDEADBEEF)) == 0xDEADBEEF


def test_crc32c_lane_path_equals_serial():
    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, 1 << 16, dtype=np.uint8).tobytes() + b"tail!"
    serial = tb._crc_raw(data, 0xFFFFFFFF) ^ 0xFFFFFFFF
    assert tb.crc32c(data) == serial
    assert tb.crc32c(data[1000:], tb.crc32c(data[:1000])) == serial   # continuation


def test_reference_architecture_from_index(ref_index):
    ent = {k.decode(): tb.decode_entry(v) for k, v in ref_index["items"][1:]}
    arch, num_elem = tb.infer_hypers(ent)
    assert arch == {"atom_feature_size": 256, "edge_feature_size": 3, "edge_hidden_size": 128,
                    "edge_fc_layers": 4, "mp_layers": 4, "fc_layers": 4}
    assert num_elem == 10
    # our parameter shapes == the bundle's shapes under the name map; trainable total 1,070,477
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from nmrgnn_amd.params import param_shapes
    hp = declare_gnn_space(HyperParameters(**arch))
    names, total = tb.variable_names(hp), 0
    for ours, shape in param_shapes(hp, num_elem):
        e = ent[names[ours] + tb.VALUE_SUFFIX]
        assert e.shape == tuple(shape) and e.dtype == tb.DT_FLOAT and e.size == 4 * int(np.prod(shape)), ours
        for s in ("m", "v"):
            assert ent[f"{names[ours]}/.OPTIMIZER_SLOT/optimizer/{s}{tb.VALUE_SUFFIX}"].shape == tuple(shape)
        total += int(np.prod(shape))
    assert total == 1070477


def test_bundle_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"a/w" + tb.VALUE_SUFFIX: rng.standard_normal((7, 5, 3)).astype(np.float32),
               "a/b" + tb.VALUE_SUFFIX: rng.standard_normal(9000).astype(np.float32),       # > lane threshold
               "step" + tb.VALUE_SUFFIX: np.asarray(12345, dtype=np.int64),
               "flag": np.array([True, False]), "empty": np.zeros((0, 4), np.float32)}
    prefix = str(tmp_path / "variables" / "variables")
    tb.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path / "variables")) == ["variables.data-00000-of-00001", "variables.index"]
    back = tb.read_bundle(prefix)
    assert list(back) == sorted(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape
        np.testing.assert_array_equal(back[k], v)
    # corruption is detected
    p = prefix + ".data-00000-of-00001"
    raw = bytearray(open(p, "rb").read())
    raw[10] ^= 1
    open(p, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(prefix)
    os.remove(p)
    with pytest.raises(FileNotFoundError, match="data shard is missing"):
        tb.read_bundle(prefix)


def test_gnn_bundle_round_trip_with_optimizer(tmp_path):
    from nmrgnn_amd.hypers import HyperParameters, declare_gnn_space
    from nmrgnn_amd.params import param_shapes
    hp = declare_gnn_space(HyperParameters(atom_feature_size=32, edge_feature_size=2, edge_hidden_size=16,
                                           mp_layers=2, fc_layers=3, edge_fc_layers=2))
    rng = np.random.default_rng(2)
    state = {k: rng.standard_normal(s).astype(np.float32) for k, s in param_shapes(hp, 10)}
    opt = {"m": {k: v * 0.1 for k, v in state.items()}, "v": {k: v * v for k, v in state.items()},
           "iter": 77, "learning_rate": 1e-4, "beta_1": 0.9, "beta_2": 0.999}
    prefix = str(tmp_path / "variables")
    tb.save_gnn_bundle(prefix, state, hp, optimizer=opt)
    st, arch, C, o = tb.load_gnn_bundle(prefix, with_optimizer=True)
    assert C == 10 and arch["atom_feature_size"] == 32 and arch["mp_layers"] == 2 and arch["fc_layers"] == 3
    for k in state:
        np.testing.assert_array_equal(st[k], state[k])
        np.testing.assert_array_equal(o["m"][k], opt["m"][k])
        np.testing.assert_array_equal(o["v"][k], opt["v"][k])
    assert o["iter"] == 77 and o["beta_1"] == pytest.approx(0.9)


def test_rejects_non_bundle(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"\x00" * 100)
    with pytest.raises(ValueError, match="magic"):
        tb.read_index(str(p))
