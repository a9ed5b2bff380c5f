#!/usr/bin/env python
"""A TensorFlow-1 "tensor bundle" checkpoint assembled BYTE BY BYTE from the published formats, independently of
articulated-pose_amd/checkpoint.py (this script imports nothing from the package: its own bit-wise CRC-32C, its own varints,
every block laid out by hand below), so that the reader is checked against something other than its own writer:

    python tests/golden/gen_tf_bundle_golden.py   ->  tests/golden/tf_bundle/model.ckpt-7.{index,data-00000-of-00001}

NOT a TensorFlow-written file -- there is no TensorFlow in the build image and the reference ships no checkpoint
(README.md:80-92) -- but laid out the way tensorflow/core/util/tensor_bundle's BundleWriter lays one out:
  data shard   raw little-endian tensor bytes back to back, in key order, no alignment;
  index        one LevelDB-format table (tensorflow/core/lib/io/table_builder.cc, kNoCompression, restart interval 16):
               a single data block holding key "" -> BundleHeaderProto and one BundleEntryProto per variable with
               PREFIX-COMPRESSED keys, an empty metaindex block, an index block whose key is the short successor of the last
               data key, every block followed by <type 0><masked crc32c>, and the 48-byte footer.
What the reference does with such a file: tf.train.Saver().restore(sess, ckpt.model_checkpoint_path), main.py:81-97,
lib/network.py:409-419."""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tf_bundle")


def crc32c_bitwise(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), one bit at a time."""
    c = 0xFFFFFFFF
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def masked(c):                       # tensorflow/core/lib/hash/crc32c.h: rotate right by 15, add a constant
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def main():
    assert crc32c_bitwise(b"123456789") == 0xE3069283            # RFC 3720 B.4 check value
    # ---- tensors (values chosen exactly representable) ---------------------------------------------------------------
    biases = struct.pack("<3f", 0.5, -1.25, 3.0)                                    # SPFN/fc1/biases   float32 (3,)
    weights = struct.pack("<6f", 1.0, 2.0, 3.0, -4.0, 5.5, -6.25)                   # SPFN/fc1/weights  float32 (2, 3)
    step = struct.pack("<q", 50000)                                                 # global_step       int64 scalar
    shard = biases + weights + step
    # ---- BundleEntryProto: dtype=1 varint, shape=2 message, shard_id=3 (0: omitted), offset=4 (0: omitted), size=5, crc32c=6 fixed32
    dim = lambda n: b"\x12" + varint(2) + b"\x08" + varint(n)                     # TensorShapeProto.dim{size=n}  (n < 128)
    e_biases = b"\x08\x01" + b"\x12" + varint(len(dim(3))) + dim(3) + b"\x28" + varint(12) + b"\x35" + struct.pack("<I", masked(crc32c_bitwise(biases)))
    e_weights = b"\x08\x01" + b"\x12" + varint(len(dim(2) + dim(3))) + dim(2) + dim(3) + b"\x20" + varint(12) + b"\x28" + varint(24) + \
        b"\x35" + struct.pack("<I", masked(crc32c_bitwise(weights)))
    e_step = b"\x08\x09" + b"\x12\x00" + b"\x20" + varint(36) + b"\x28" + varint(8) + b"\x35" + struct.pack("<I", masked(crc32c_bitwise(step)))
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"                  # num_shards = 1; endianness LITTLE (0, omitted); version{producer = 1}
    # ---- data block: entries <shared><non_shared><value_len><key suffix><value>, restart array [0], count 1 ------------------
    k1, k2, k3 = b"SPFN/fc1/biases", b"SPFN/fc1/weights", b"global_step"
    shared12 = 9                                                                    # "SPFN/fc1/" is common to k1 and k2
    assert k1[:shared12] == k2[:shared12] and k1[shared12] != k2[shared12]
    block = b"".join([
        varint(0) + varint(0) + varint(len(header)) + header,                                       # key ""
        varint(0) + varint(len(k1)) + varint(len(e_biases)) + k1 + e_biases,
        varint(shared12) + varint(len(k2) - shared12) + varint(len(e_weights)) + k2[shared12:] + e_weights,
        varint(0) + varint(len(k3)) + varint(len(e_step)) + k3 + e_step,
        struct.pack("<I", 0), struct.pack("<I", 1)])
    trailer = lambda b: b"\x00" + struct.pack("<I", masked(crc32c_bitwise(b + b"\x00")))      # type kNoCompression + masked crc
    f = block + trailer(block)
    meta_off = len(f)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)                                       # empty block: restarts [0], count 1
    f += meta + trailer(meta)
    index_off = len(f)
    handle = varint(0) + varint(len(block))
    sep = b"h"                                                                               # FindShortSuccessor("global_step")
    index = varint(0) + varint(len(sep)) + varint(len(handle)) + sep + handle + struct.pack("<I", 0) + struct.pack("<I", 1)
    f += index + trailer(index)
    footer = varint(meta_off) + varint(len(meta)) + varint(index_off) + varint(len(index))
    f += footer + b"\x00" * (40 - len(footer)) + struct.pack("<II", 0x8B80FB57, 0xDB477524)    # magic, low word first
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "model.ckpt-7.index"), "wb").write(f)
    open(os.path.join(OUT, "model.ckpt-7.data-00000-of-00001"), "wb").write(shard)
    open(os.path.join(OUT, "checkpoint"), "w").write('model_checkpoint_path: "model.ckpt-7"\nall_model_checkpoint_paths: "model.ckpt-7"\n')
    print("index %d bytes, shard %d bytes" % (len(f), len(shard)))


if __name__ == "__main__":
    main()
