"""TF-checkpoint (tensor bundle) reader: round trip through the bundle writer that follows the same published format, CRC
checks, optimizer-slot filtering.  (No TensorFlow-written file exists offline: see the module's PINNING STATUS.)"""
import numpy as np
import pytest


def test_crc32c_known_answers():
    from articulated_pose_amd.checkpoint import crc32c, mask_crc
    assert crc32c(b"123456789") == 0xE3069283                     # the standard CRC-32C check value
    assert crc32c(b"\x00" * 32) == 0x8A9136AA                      # RFC 3720 B.4
    assert crc32c(b"\xff" * 32) == 0x62A8AB43
    assert mask_crc(0) == 0xa282ead8


def test_bundle_round_trip_and_filter(tmp_path):
    from articulated_pose_amd import checkpoint as ck
    from articulated_pose_amd.weights import synthetic_weights
    w = synthetic_weights(3, seed=9)
    extra = dict(w)
    some = sorted(w)[0]
    extra[some + "/Adam"] = np.zeros_like(w[some]); extra[some + "/Adam_1"] = np.ones_like(w[some])
    extra["beta1_power"] = np.float32(0.9); extra["global_step"] = np.int64(1234)
    extra["odd/int"] = np.arange(7, dtype=np.int32).reshape(7)
    prefix = str(tmp_path / "model.ckpt-7")
    ck.write_bundle(prefix, extra, block_entries=5)
    entries, header = ck.read_index(prefix + ".index")
    assert header["num_shards"] == 1 and set(entries) == set(extra)
    got = ck.read_tf_checkpoint(prefix)
    assert set(got) == set(extra)
    for k in extra:
        np.testing.assert_array_equal(got[k], np.asarray(extra[k]))
        assert got[k].dtype == np.asarray(extra[k]).dtype
    out = ck.convert(prefix, str(tmp_path / "w.npz"))
    assert set(out) == set(w) | {"odd/int"}
    from articulated_pose_amd.weights import load_npz
    back = load_npz(str(tmp_path / "w.npz"))
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])


def test_corruption_is_detected(tmp_path):
    from articulated_pose_amd import checkpoint as ck
    prefix = str(tmp_path / "m")
    ck.write_bundle(prefix, {"a/weights": np.arange(12, dtype=np.float32).reshape(3, 4), "b/biases": np.ones(4, np.float32)})
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[5] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="tensor crc"):
        ck.read_tf_checkpoint(prefix)
    assert ck.read_tf_checkpoint(prefix, verify=False)["b/biases"].sum() == 4
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        ck.read_index(prefix + ".index")
    with pytest.raises(ValueError, match="magic"):
        open(prefix + ".index", "wb").write(b"x" * 64)
        ck.read_index(prefix + ".index")


def test_crc32c_lane_parallel_equals_serial():
    """Tensors above 128 KiB take the numpy lane-parallel CRC (registers chained by the GF(2) zeros operator): equal to the byte
    loop at sizes around the switch and for chained calls; RFC 3720 B.4 vectors."""
    from articulated_pose_amd import checkpoint as C
    assert C.crc32c(b"\x00" * 32) == 0x8A9136AA and C.crc32c(b"\xff" * 32) == 0x62A8AB43 and C.crc32c(bytes(range(32))) == 0x46DD794E
    rng = np.random.RandomState(0)
    for n in (0, 1, 131071, 131072, 131073, 700001):
        d = rng.randint(0, 256, n, dtype=np.uint8).tobytes()
        want = C._crc_serial(d, 0xFFFFFFFF) ^ 0xFFFFFFFF
        assert C.crc32c(d) == want
        assert C.crc32c(d[n // 3:], C.crc32c(d[:n // 3])) == want
