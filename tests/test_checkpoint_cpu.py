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


def test_hand_assembled_bundle():
    """tests/golden/tf_bundle/: a bundle laid out byte by byte by tests/golden/gen_tf_bundle_golden.py from the published
    tensor-bundle / table formats -- its own bit-wise CRC-32C, its own varints, prefix-compressed keys, BundleWriter's
    header (version{producer: 1}), an int64 scalar with an empty shape message -- i.e. NOT produced by checkpoint.write_bundle.
    (Still not a TensorFlow-written file: none exists offline.)"""
    import os
    from articulated_pose_amd.checkpoint import is_model_variable, read_index, read_tf_checkpoint
    prefix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle", "model.ckpt-7")
    entries, header = read_index(prefix + ".index")
    assert header["num_shards"] == 1 and header.get("endianness", 0) == 0 and header["version"] == b"\x08\x01"
    assert sorted(entries) == ["SPFN/fc1/biases", "SPFN/fc1/weights", "global_step"]
    assert entries["SPFN/fc1/weights"]["offset"] == 12 and entries["SPFN/fc1/weights"]["shape"] == (2, 3)
    w = read_tf_checkpoint(prefix)                        # verifies every block crc and every tensor crc
    np.testing.assert_array_equal(w["SPFN/fc1/biases"], np.asarray([0.5, -1.25, 3.0], np.float32))
    np.testing.assert_array_equal(w["SPFN/fc1/weights"], np.asarray([[1.0, 2.0, 3.0], [-4.0, 5.5, -6.25]], np.float32))
    assert w["global_step"].dtype == np.int64 and w["global_step"].shape == () and int(w["global_step"]) == 50000
    assert sorted(read_tf_checkpoint(prefix, include=is_model_variable)) == ["SPFN/fc1/biases", "SPFN/fc1/weights"]
    # a flipped payload byte must be caught by the tensor crc, a flipped index byte by the block crc
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for f in os.listdir(os.path.dirname(prefix)):
            shutil.copy(os.path.join(os.path.dirname(prefix), f), d)
        p2 = os.path.join(d, "model.ckpt-7")
        raw = bytearray(open(p2 + ".data-00000-of-00001", "rb").read())
        raw[5] ^= 1
        open(p2 + ".data-00000-of-00001", "wb").write(bytes(raw))
        with pytest.raises(ValueError, match="crc"):
            read_tf_checkpoint(p2)
        idx = bytearray(open(p2 + ".index", "rb").read())
        idx[20] ^= 1
        open(p2 + ".index", "wb").write(bytes(idx))
        with pytest.raises(ValueError, match="crc"):
            read_index(p2 + ".index")


def test_tf_written_checkpoint():
    """The one thing SURVEY 8(f)-1 still lacks: a checkpoint written by TensorFlow itself (none ships with the reference, no TensorFlow
    in the image).  Drop `model.ckpt-<step>.index` + `.data-00000-of-00001` written by the reference's tf.train.Saver (TensorFlow 1.10,
    main.py:81-97) into tests/golden/tf_written/ -- optionally with expected.npz = {variable name: array} from
    tf.train.load_checkpoint -- and this test pins the reader against it: every tensor's CRC verifies, the reference's variable
    inventory (weights.layer_table) is present with its shapes, and the values equal expected.npz when that is there."""
    import glob
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_written")
    idx = sorted(glob.glob(os.path.join(here, "*.index")))
    if not idx:
        pytest.skip("no TensorFlow-written checkpoint under tests/golden/tf_written/ (see INTEGRATION.md section 3)")
    from articulated_pose_amd import checkpoint as ck
    from articulated_pose_amd.weights import layer_table
    prefix = idx[0][:-len(".index")]
    got = ck.read_tf_checkpoint(prefix)                       # verifies every block and tensor CRC
    assert len(got) > 0
    for K in (2, 3, 4):
        names = [full + "/weights" for full, _cin, _cout, _bn, _kind in layer_table(K, True, True, "SPFN")]
        if all(n in got for n in names):
            for full, cin, cout, _bn, _kind in layer_table(K, True, True, "SPFN"):
                assert tuple(got[full + "/weights"].shape[-2:]) == (cin, cout), full
            break
    else:
        pytest.fail("the checkpoint holds the ANCSH variable inventory for none of K = 2, 3, 4")
    exp = os.path.join(here, "expected.npz")
    if os.path.exists(exp):
        want = np.load(exp)
        for k in want.files:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
