"""TensorFlow checkpoint "V2" (tensor bundle) files, read and written without TensorFlow
(`saver.save(sess, checkpoint_path + 'model.ckpt', global_step)`,
multigpu_train.py:188-189; `tf.train.get_checkpoint_state` + `saver.restore`, test.py:146-150;
`slim.assign_from_checkpoint_fn(pretrained_model_path, ...)`, multigpu_train.py:149-151).

A checkpoint `<prefix>` is
  <prefix>.index                 an SSTable (TensorFlow's table::Table = the LevelDB table format,
                                 uncompressed for bundles): key "" -> BundleHeaderProto, key <variable
                                 name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, masked crc32c}
  <prefix>.data-00000-of-00001   the raw little-endian tensor bytes at those offsets
plus the directory's `checkpoint` text file naming the latest prefix.

Format restated from the published TensorFlow / LevelDB sources (tensor_bundle.proto, tensor_bundle.cc,
table/format.cc, table/block_builder.cc); no TF checkpoint exists in this container to pin it
against, so the tests hold it to the formats' own known answers (CRC-32C vectors, magic number,
block layout) and to round trips.  Host-side file I/O: nothing here touches the GPU; CRC-32C over the
tensor bytes uses the library's host routine `ocr_crc32c` when it is built (pure Python otherwise)."""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
BLOCK_RESTART_INTERVAL = 16
BLOCK_SIZE = 262144

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_BFLOAT16, DT_HALF = 1, 2, 3, 9, 14, 19
_NP_OF = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"),
          DT_INT64: np.dtype("<i8"), DT_HALF: np.dtype("<f2")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
          np.dtype("int64"): DT_INT64, np.dtype("float16"): DT_HALF}


# ------------------------------------------------------------------------------- CRC-32C
def _make_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC_TABLE = _make_table()


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), as LevelDB's crc32c::Extend(crc, data)."""
    data = bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data
    if len(data) >= 4096:
        try:                                   # host routine of the kernel library (slicing-by-8)
            import ctypes
            from . import _lib
            fn = _lib.load().ocr_crc32c
            fn.restype = ctypes.c_uint32
            buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
            return int(fn(buf, ctypes.c_size_t(len(data)), ctypes.c_uint32(crc)))
        except Exception:
            pass
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """crc32c::Mask: rotate right by 15 and add a constant (CRCs of CRCs are weak)."""
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------- varints / protobuf
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _pb_fields(buf):
    """Iterate (field number, wire type, value) of a serialized protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _pb_varint(fn, v):
    return _put_varint(fn << 3) + _put_varint(v)


def _pb_bytes(fn, b):
    return _put_varint((fn << 3) | 2) + _put_varint(len(b)) + b


def _encode_shape(shape):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }."""
    return b"".join(_pb_bytes(2, _pb_varint(1, int(d)) if d else b"") for d in shape)


def _decode_shape(buf):
    dims = []
    for fn, wt, v in _pb_fields(buf):
        if fn == 2:
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
    return tuple(dims)


def encode_entry(dtype, shape, offset, size, crc_masked, shard_id=0):
    """BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)}."""
    out = _pb_varint(1, dtype) + _pb_bytes(2, _encode_shape(shape))
    if shard_id:
        out += _pb_varint(3, shard_id)
    if offset:
        out += _pb_varint(4, offset)
    if size:
        out += _pb_varint(5, size)
    out += _put_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)
    return out


def decode_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for fn, wt, v in _pb_fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            e["shape"] = _decode_shape(v)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif fn == 7:
            e["slices"] += 1
    return e


def encode_header(num_shards=1):
    """BundleHeaderProto {num_shards=1, endianness=2 (LITTLE=0, omitted), version=3 {producer=1}}."""
    return _pb_varint(1, num_shards) + _pb_bytes(3, _pb_varint(1, 1))


# ------------------------------------------------------------------------------- table blocks
class _BlockBuilder:
    """LevelDB block: prefix-compressed entries, restart points every 16 keys."""

    def __init__(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b""

    def add(self, key, value):
        shared = 0
        if self.counter < BLOCK_RESTART_INTERVAL:
            n = min(len(self.last_key), len(key))
            while shared < n and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + \
            struct.pack("<I", len(self.restarts))


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _snappy_decompress(buf):
    """Raw snappy block format (LevelDB kSnappyCompression); bundles are written uncompressed, kept
    for tables written by other tools."""
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block")
    return bytes(out)


def _read_block(data, offset, size, verify=True):
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
    if verify and unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 0:
        return contents
    if ctype == 1:
        return _snappy_decompress(contents)
    raise ValueError("unknown block compression type %d" % ctype)


def read_table(path):
    """All (key, value) pairs of a table file, in key order."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow table file (bad magic number)" % path)
    footer = data[-48:]
    _, pos = _get_varint(footer, 0)           # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)      # index handle
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize)):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        out.extend(_block_entries(_read_block(data, boff, bsize)))
    return out


def write_table(path, items):
    """items: (key bytes, value bytes) sorted by key.  Uncompressed, 256 KiB blocks (bundle settings)."""
    out = bytearray()
    index = _BlockBuilder()

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes)
        out.append(0)                                                   # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block_bytes + b"\x00"))))
        return off, len(block_bytes)

    blk, last = _BlockBuilder(), None
    pending = None
    for key, value in items:
        if last is not None and key <= last:
            raise ValueError("table keys must be strictly increasing")
        if pending is not None:
            # index key for the finished block: any separator in [last key of the block, next key);
            # the last key itself is always valid (LevelDB shortens it, readers only compare)
            index.add(pending[0], _put_varint(pending[1]) + _put_varint(pending[2]))
            pending = None
        blk.add(key, value)
        last = key
        if blk.size_estimate() >= BLOCK_SIZE:
            off, size = emit(blk.finish())
            pending = (last, off, size)
            blk = _BlockBuilder()
    if blk.counter or not out:
        off, size = emit(blk.finish())
        pending = (last if last is not None else b"", off, size)
    if pending is not None:
        index.add(pending[0], _put_varint(pending[1]) + _put_varint(pending[2]))
    moff, msize = emit(_BlockBuilder().finish())                        # empty metaindex block
    ioff, isize = emit(index.finish())
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


# ------------------------------------------------------------------------------- bundles
def write_bundle(prefix, tensors):
    """tensors: {variable name: numpy array}.  Writes <prefix>.index and <prefix>.data-00000-of-00001
    (tensors laid out in key order, as BundleWriter's callers add them)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = [(b"", encode_header(1))]
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)            # (0-d arrays are contiguous; ascontiguousarray would make them 1-d)
            if a.dtype not in _DT_OF:
                raise TypeError("%s: dtype %s has no checkpoint encoding here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode(), encode_entry(_DT_OF[a.dtype], a.shape, offset, len(raw),
                                                      mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + ".index", items)


def read_bundle(prefix, verify=True):
    """{variable name: numpy array} of a V2 checkpoint prefix."""
    entries = read_table(prefix + ".index")
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    num_shards, big_endian = 1, False
    for fn, _, v in _pb_fields(entries[0][1]):
        if fn == 1:
            num_shards = v
        elif fn == 2:
            big_endian = v == 1
    if big_endian:
        raise ValueError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, val in entries[1:]:
        e = decode_entry(val)
        if e["slices"]:
            raise ValueError("%s: partitioned variables are not supported" % key.decode())
        if e["dtype"] not in _NP_OF:
            raise TypeError("%s: unsupported checkpoint dtype %d" % (key.decode(), e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), dtype=np.uint8, mode="r")
        raw = bytes(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if verify and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("%s: tensor checksum mismatch" % key.decode())
        out[key.decode()] = np.frombuffer(raw, dtype=_NP_OF[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ------------------------------------------------------------------------------- `checkpoint` state file
def update_checkpoint_state(save_dir, model_checkpoint_path, all_paths=None):
    """tf.train.update_checkpoint_state: the text-format CheckpointState proto."""
    rel = os.path.basename(model_checkpoint_path)
    lines = ['model_checkpoint_path: "%s"' % rel]
    for p in (all_paths or [rel]):
        lines.append('all_model_checkpoint_paths: "%s"' % os.path.basename(p))
    with open(os.path.join(save_dir, "checkpoint"), "w") as f:
        f.write("\n".join(lines) + "\n")


def get_checkpoint_state(checkpoint_dir):
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path (absolute), or None."""
    path = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.exists(path):
        return None
    for line in open(path):
        line = line.strip()
        if line.startswith("model_checkpoint_path:"):
            rel = line.split(":", 1)[1].strip().strip('"')
            return rel if os.path.isabs(rel) else os.path.join(checkpoint_dir, rel)
    return None


# ------------------------------------------------------------------------------- V1 checkpoints
# The slim model-zoo files the reference starts from (`--pretrained_model_path
# /data/resnet_v1_50.ckpt`, train.sh:3, multigpu_train.py:149-151) are "V1" checkpoints: ONE table file
# (tensorflow/core/util/tensor_slice_writer.cc, saved_tensor_slice.proto), usually with
# snappy-compressed blocks.  Key "" holds SavedTensorSlices{meta: SavedTensorSliceMeta{tensor:
# SavedSliceMeta{name=1, shape=2, type=3, slice=4}}}; every other entry holds
# SavedTensorSlices{data: SavedSlice{name=1, slice=2, data=3: TensorProto}} with the values in the
# TensorProto's typed repeated field (float_val=5 packed, double_val=6, int_val=7, int64_val=10,
# half_val=13) or in tensor_content=4.  Only whole-tensor slices are supported (no partitioned
# variables), which is what single-device savers write.
_V1_FIELD = {DT_FLOAT: (5, "<f4"), DT_DOUBLE: (6, "<f8"), DT_INT32: (7, None), DT_INT64: (10, None), DT_HALF: (13, None)}


def _packed_varints(buf):
    out, pos = [], 0
    while pos < len(buf):
        v, pos = _get_varint(buf, pos)
        out.append(v - (1 << 64) if v >= (1 << 63) else v)
    return out


def _decode_tensor_proto(buf, dtype, count):
    fld, fmt = _V1_FIELD[dtype]
    content, chunks = None, []
    for fn, wt, v in _pb_fields(buf):
        if fn == 4:
            content = bytes(v)
        elif fn == fld:
            if wt == 2 and fmt:                           # packed fixed-width
                chunks.append(np.frombuffer(bytes(v), dtype=fmt))
            elif wt == 2:                                 # packed varints
                chunks.append(np.array(_packed_varints(v), dtype=np.int64))
            elif wt == 0:
                chunks.append(np.array([v - (1 << 64) if v >= (1 << 63) else v], dtype=np.int64))
            else:                                         # unpacked fixed32 / fixed64
                chunks.append(np.frombuffer(bytes(v), dtype=fmt))
    np_dt = _NP_OF[dtype]
    if content is not None and len(content):
        return np.frombuffer(content, dtype=np_dt).copy()
    if not chunks:
        return np.zeros(count, np_dt)
    a = np.concatenate(chunks)
    if dtype == DT_HALF:                                  # half_val carries the 16 raw bits in an int32
        a = a.astype(np.uint16).view(np.float16)
    a = a.astype(np_dt)
    if a.size == 1 and count > 1:                         # TensorProto's "all elements equal" compression
        a = np.repeat(a, count)
    return a


def read_v1_checkpoint(path):
    """{variable name: numpy array} of a V1 (single-file) TensorFlow checkpoint."""
    entries = read_table(path)
    if not entries or entries[0][0] != b"":
        raise ValueError("%s has no SavedTensorSliceMeta entry" % path)
    meta = {}
    for fn, _, v in _pb_fields(entries[0][1]):
        if fn != 1:
            continue
        for f2, _, v2 in _pb_fields(v):                   # SavedTensorSliceMeta.tensor
            if f2 != 1:
                continue
            name, shape, dtype, nslices = None, (), DT_FLOAT, 0
            for f3, _, v3 in _pb_fields(v2):              # SavedSliceMeta
                if f3 == 1:
                    name = bytes(v3).decode()
                elif f3 == 2:
                    shape = _decode_shape(v3)
                elif f3 == 3:
                    dtype = v3
                elif f3 == 4:
                    nslices += 1
            if nslices > 1:
                raise ValueError("%s: partitioned variables are not supported" % name)
            meta[name] = (shape, dtype)
    out = {}
    for _, val in entries[1:]:
        for fn, _, v in _pb_fields(val):
            if fn != 2:
                continue
            name, data = None, b""
            for f2, _, v2 in _pb_fields(v):               # SavedSlice
                if f2 == 1:
                    name = bytes(v2).decode()
                elif f2 == 3:
                    data = v2
            shape, dtype = meta[name]
            if dtype not in _V1_FIELD:
                raise TypeError("%s: unsupported checkpoint dtype %d" % (name, dtype))
            count = int(np.prod(shape)) if shape else 1
            out[name] = _decode_tensor_proto(data, dtype, count).reshape(shape)
    return out


def is_v1_checkpoint(path):
    """A single table file (V2 checkpoints are a prefix with .index / .data-* files)."""
    if not os.path.isfile(path) or os.path.exists(path + ".index"):
        return False
    with open(path, "rb") as f:
        f.seek(0, 2)
        if f.tell() < 48:
            return False
        f.seek(-8, 2)
        return struct.unpack("<Q", f.read(8))[0] == TABLE_MAGIC
