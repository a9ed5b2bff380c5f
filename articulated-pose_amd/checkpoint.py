"""TensorFlow-1 checkpoint -> `.npz` weight converter without TensorFlow.

The reference restores its pretrained models with `tf.train.Saver.restore` (lib/network.py:409-419, main.py:81-97); this
build loads `{TF variable name: ndarray}` dictionaries (`weights.load_npz`).  A TF checkpoint `<prefix>.index` +
`<prefix>.data-0000N-of-0000M` is a "tensor bundle":

  * `.index` is a LevelDB-format sorted table (tensorflow/core/lib/io/table*, same layout and magic as LevelDB's
    table/format.cc): data blocks of prefix-compressed (key, value) entries + restart array, each followed by a 5-byte
    trailer (compression type, masked crc32c), an index block of block handles and a 48-byte footer
    (metaindex handle, index handle, padding, magic 0xdb4775248b80fb57);
  * key "" holds a BundleHeaderProto (num_shards = 1, endianness = 2, version = 3), every other key is a variable name
    whose value is a BundleEntryProto (dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6, slices = 7);
  * the data shards hold the raw little-endian tensor bytes at [offset, offset + size).

PINNING STATUS: restated from the published formats above; no TensorFlow and no TF-written checkpoint exist in the build
image, so the reader is exercised against bundles produced by `write_bundle` below (same specification) and verifies the
per-tensor and per-block CRCs TensorFlow stores, which a real file must satisfy.  Unpinned against a TF-written file.

    python -m articulated_pose_amd.checkpoint results/model/3.9/model.ckpt-50000 eyeglasses_ancsh.npz
"""
import struct
import sys

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}

# ---- crc32c (Castagnoli), TensorFlow's masking ------------------------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def _crc_serial(data, c):
    t = _crc_table()
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c


def _zeros_operator(nbytes):
    """32x32 GF(2) matrix (as 32 column words) that advances a raw CRC register over `nbytes` zero bytes: the register update
    is linear, so appending data of known length to a message = operator(len) applied to its register, XOR the data's register."""
    # one zero BIT: c -> (c >> 1) ^ (poly if c & 1)
    op = [0x82F63B78] + [1 << (i - 1) for i in range(1, 32)]          # column i = image of bit i

    def apply(m, v):
        r, i = 0, 0
        while v:
            if v & 1:
                r ^= m[i]
            v >>= 1
            i += 1
        return r

    def square(m):
        return [apply(m, m[i]) for i in range(32)]

    result = None
    nbits = nbytes * 8
    while nbits:
        if nbits & 1:
            result = op if result is None else [apply(op, result[i]) for i in range(32)]
        op = square(op)
        nbits >>= 1
    return result if result is not None else [1 << i for i in range(32)]


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli).  Multi-megabyte tensors are split into equal lanes advanced TOGETHER by numpy table look-ups (one
    vector operation per byte position instead of one Python iteration per byte); the lanes' registers are then chained with
    the GF(2) operator that advances a register over a lane's length of zeros."""
    buf = np.frombuffer(bytes(data), np.uint8)
    c = crc ^ 0xFFFFFFFF
    n = buf.size
    lanes = 2048
    if n < 64 * lanes:
        return _crc_serial(buf.tobytes(), c) ^ 0xFFFFFFFF
    L = n // lanes
    t = _crc_table()
    block = buf[:L * lanes].reshape(lanes, L)
    reg = np.zeros(lanes, np.uint32)
    reg[0] = c                                            # the running register enters the first lane
    for i in range(L):
        reg = t[(reg ^ block[:, i]) & 0xFF] ^ (reg >> np.uint32(8))
    op = _zeros_operator(L)
    cols = np.asarray(op, np.uint64)
    c = int(reg[0])
    for j in range(1, lanes):                             # c = advance(c, L zero bytes) ^ lane register (lanes start from 0)
        v, r, i = c, 0, 0
        while v:
            if v & 1:
                r ^= int(cols[i])
            v >>= 1
            i += 1
        c = r ^ int(reg[j])
    c = _crc_serial(buf[L * lanes:].tobytes(), c)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- varints / minimal protobuf ---------------------------------------------------------------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _proto_fields(buf):
    """[(field number, wire type, value)]: value = int (varint / fixed) or bytes (length-delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _parse_shape(buf):
    dims = []
    for num, _wt, v in _proto_fields(buf):
        if num == 2:                                   # TensorShapeProto.dim
            size = 0
            for n2, _w2, v2 in _proto_fields(v):
                if n2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=False)
    for num, _wt, v in _proto_fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            e["shape"] = _parse_shape(v)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] = True
    return e


# ---- LevelDB-format table -----------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify):
    body = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if verify and mask_crc(crc32c(data[offset:offset + size + 1])) != stored:
        raise ValueError("table block at %d: crc mismatch" % offset)
    if ctype != 0:
        raise ValueError("table block at %d is compressed (type %d); only uncompressed index files are supported" % (offset, ctype))
    return body


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path, verify=True):
    """{variable name: entry dict} + the bundle header dict from a `.index` file."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-48:]
    _mo, p = _varint(footer, 0)
    _ms, p = _varint(footer, p)
    io, p = _varint(footer, p)
    isz, p = _varint(footer, p)
    entries, header = {}, {}
    for _sep, handle in _block_entries(_read_block(data, io, isz, verify)):
        off, q = _varint(handle, 0)
        sz, q = _varint(handle, q)
        for key, val in _block_entries(_read_block(data, off, sz, verify)):
            if key == b"":
                for num, _wt, v in _proto_fields(val):
                    header[{1: "num_shards", 2: "endianness", 3: "version"}.get(num, num)] = v
            else:
                entries[key.decode()] = _parse_entry(val)
    return entries, header


def read_tf_checkpoint(prefix, verify=True, include=None):
    """All variables of the checkpoint `prefix` as {name: ndarray}.  include: optional predicate on the name."""
    entries, header = read_index(prefix + ".index", verify)
    if header.get("endianness", 0) != 0:
        raise ValueError("big-endian checkpoints are not supported")
    n_shards = header.get("num_shards", 1)
    shards, out = {}, {}
    for name, e in sorted(entries.items()):
        if include is not None and not include(name):
            continue
        if e["slices"]:
            raise ValueError("variable %s is stored as slices (partitioned variable): not supported" % name)
        if e["dtype"] not in _DTYPES:
            raise ValueError("variable %s has unsupported dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, n_shards), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        dt = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise ValueError("variable %s: %d bytes stored for shape %s of %s" % (name, e["size"], e["shape"], dt))
        if verify and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise ValueError("variable %s: tensor crc mismatch" % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=dt).reshape(e["shape"]).copy()
    return out


def is_model_variable(name):
    """Drop optimizer slots and bookkeeping (Adam moments, beta powers, global_step, EMA shadows)."""
    tail = name.rsplit("/", 1)[-1]
    return not (tail in ("Adam", "Adam_1", "ExponentialMovingAverage") or name.startswith(("beta1_power", "beta2_power", "global_step"))
                or "/Adam" in name or "_power" in tail)


def convert(prefix, out_npz, verify=True, include=is_model_variable):
    w = read_tf_checkpoint(prefix, verify, include)
    np.savez(out_npz, **w)
    return w


# ---- writer (tests / round trips; the same specification the reader follows) ----------------------------------------------------
def _proto_varint_field(num, v):
    return _put_varint(num << 3) + _put_varint(v)


def _proto_bytes_field(num, b):
    return _put_varint((num << 3) | 2) + _put_varint(len(b)) + b


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, block_entries=8):
    """Write {name: ndarray} as a one-shard tensor bundle (`.index` + `.data-00000-of-00001`)."""
    data, entries = bytearray(), []
    for name in sorted(tensors):
        a = np.ascontiguousarray(tensors[name])
        raw = a.tobytes()
        shape = b"".join(_proto_bytes_field(2, _proto_varint_field(1, int(d))) for d in a.shape)
        e = _proto_varint_field(1, _DTYPE_IDS[a.dtype]) + _proto_bytes_field(2, shape) + _proto_varint_field(4, len(data)) + \
            _proto_varint_field(5, len(raw)) + _put_varint((6 << 3) | 5) + struct.pack("<I", mask_crc(crc32c(raw)))
        entries.append((name.encode(), e))
        data += raw
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    items = [(b"", _proto_varint_field(1, 1) + _proto_bytes_field(3, _proto_varint_field(1, 1)))] + entries
    f, index_items = bytearray(), []

    def emit(block):
        off = len(f)
        f.extend(block + b"\x00")
        f.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        index_items.append((chunk[-1][0] + b"\x00", emit(_build_block(chunk))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, 1))
    footer = meta + index
    f.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    open(prefix + ".index", "wb").write(bytes(f))


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit("usage: python -m articulated_pose_amd.checkpoint <checkpoint prefix> <out.npz>")
    got = convert(sys.argv[1], sys.argv[2])
    print("wrote %d variables (%d parameters) to %s" % (len(got), sum(v.size for v in got.values()), sys.argv[2]))
