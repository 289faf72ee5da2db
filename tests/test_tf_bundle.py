"""CPU: TensorFlow V2 checkpoint bundles read / written without TensorFlow (tf_bundle.py,
checkpoint.save_tf_checkpoint / load_tf_checkpoint).  No TF checkpoint exists in this container, so the
format is held to its own published constants: the CRC-32C test vectors of RFC 3720 B.4, LevelDB's
table magic number and footer / block-trailer layout, protobuf wire bytes written out by hand, and
round trips (PARITY UNPINNED against a TensorFlow-written file)."""
import os
import struct

import numpy as np
import pytest

from tensorflow_ocr_amd import checkpoint as C
from tensorflow_ocr_amd import tf_bundle as B


def test_crc32c_known_answers_python_and_library():
    vec = [(b"123456789", 0xE3069283), (bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43),
           (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C)]
    for data, want in vec:
        assert B.crc32c(data) == want
    # the library's host routine (used for >= 4096 bytes) against the bytewise Python loop, and Extend
    import ctypes
    from tensorflow_ocr_amd import _lib
    fn = _lib.load().ocr_crc32c
    fn.restype = ctypes.c_uint32
    for data, want in vec:
        assert fn(data, ctypes.c_size_t(len(data)), ctypes.c_uint32(0)) == want
    big = np.random.default_rng(0).integers(0, 256, 70001).astype(np.uint8).tobytes()
    c = 0xFFFFFFFF
    for b in big:
        c = B._CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    assert B.crc32c(big) == c ^ 0xFFFFFFFF == B.crc32c(big[33333:], B.crc32c(big[:33333]))
    # LevelDB's masking (crc32c.h): rotate right 15, add 0xa282ead8
    assert B.mask_crc(0) == 0xa282ead8 and B.unmask_crc(B.mask_crc(0xDEADBEEF)) == 0xDEADBEEF


def test_proto_wire_bytes():
    # BundleHeaderProto{num_shards: 1, version{producer: 1}}
    assert B.encode_header(1) == bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:3} dim{size:300}}, offset: 16, size: 3600, crc32c: 0x01020304}
    e = B.encode_entry(B.DT_FLOAT, (3, 300), 16, 3600, 0x01020304)
    assert e == bytes([0x08, 0x01, 0x12, 0x09, 0x12, 0x02, 0x08, 0x03, 0x12, 0x03, 0x08, 0xac, 0x02,
                       0x20, 0x10, 0x28, 0x90, 0x1c, 0x35, 0x04, 0x03, 0x02, 0x01])
    d = B.decode_entry(e)
    assert d["dtype"] == 1 and d["shape"] == (3, 300) and d["offset"] == 16 and d["size"] == 3600 and d["crc32c"] == 0x01020304
    assert B.decode_entry(B.encode_entry(B.DT_INT64, (), 0, 8, 7))["shape"] == ()


def test_table_layout_and_round_trip(tmp_path):
    path = str(tmp_path / "t.index")
    items = [(b"", b"hdr")] + [(("k%05d" % i).encode(), os.urandom(i % 50)) for i in range(8000)]
    B.write_table(path, items)
    raw = open(path, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 and raw[-8:] == bytes.fromhex("57fb808b247547db")
    assert B.read_table(path) == items
    # first data block: entry 0 is (shared 0, non_shared 0, value_len 3, "hdr"); its trailer is
    # type 0 + masked crc32c(block + type)
    assert raw[:6] == bytes([0, 0, 3]) + b"hdr"
    raw2 = bytearray(raw)
    raw2[10] ^= 1                                  # corrupt one byte of the first block
    open(path, "wb").write(raw2)
    with pytest.raises(ValueError):
        B.read_table(path)
    with pytest.raises(ValueError):
        B.write_table(path, [(b"b", b""), (b"a", b"")])
    # snappy-compressed blocks from other writers decode too: literal "abcd", copy(offset 4, len 4), literal "!"
    assert B._snappy_decompress(bytes([9, 0x0c]) + b"abcd" + bytes([0x01, 0x04, 0x00]) + b"!") == b"abcdabcd!"


def test_checkpoint_round_trip_with_ema_and_state_file(tmp_path):
    rng = np.random.default_rng(1)
    sd = {"resnet_v1_50/conv1/weights": rng.standard_normal((7, 7, 3, 64)).astype(np.float32),
          "feature_fusion/Conv_5/biases": np.arange(16, dtype=np.float32),
          "resnet_v1_50/conv1/BatchNorm/moving_variance": np.ones(64, np.float32),
          "scalar": np.float32(3.5), "empty": np.zeros((0, 4), np.float32)}
    sd.update({"v%04d/w" % i: rng.standard_normal((int(rng.integers(1, 200)),)).astype(np.float32) for i in range(2500)})
    ema = {"feature_fusion/Conv_5/biases": np.arange(16, dtype=np.float32) * 0.5}
    d = str(tmp_path)
    C.save_tf_checkpoint(d, 1000, sd, ema)
    prefix = C.save_tf_checkpoint(d, 2000, sd, ema)
    assert os.path.basename(prefix) == "model.ckpt-2000"
    assert open(os.path.join(d, "checkpoint")).read().splitlines()[0] == 'model_checkpoint_path: "model.ckpt-2000"'
    assert B.get_checkpoint_state(d) == prefix
    out, step = C.load_tf_checkpoint(d)
    assert step == 2000 and set(out) == set(sd)
    for k in sd:
        assert out[k].dtype == np.asarray(sd[k]).dtype and out[k].shape == np.asarray(sd[k]).shape
        assert np.array_equal(out[k], sd[k])
    avg, _ = C.load_tf_checkpoint(prefix + ".index", use_moving_averages=True)
    assert np.array_equal(avg["feature_fusion/Conv_5/biases"], ema["feature_fusion/Conv_5/biases"])
    assert np.array_equal(avg["scalar"], sd["scalar"])
    keys = [k for k, _ in B.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys == sorted(keys) and b"global_step" in keys
    assert b"feature_fusion/Conv_5/biases/ExponentialMovingAverage" in keys
    # data corruption is caught by the per-tensor checksum
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[100] ^= 0x40
    open(data, "wb").write(raw)
    with pytest.raises(ValueError):
        C.load_tf_checkpoint(prefix)


def _v1_file(path, tensors, compress):
    """A V1 checkpoint written field by field (saved_tensor_slice.proto), optionally with snappy blocks
    (literal-only streams: valid snappy, no compressor needed)."""
    def tensor_proto(a):
        a = np.asarray(a)
        if a.dtype == np.float32:
            return B._pb_bytes(5, a.astype("<f4").tobytes())                       # float_val, packed
        if a.dtype == np.int64:
            return B._pb_bytes(10, b"".join(B._put_varint(int(v)) for v in a.ravel()))   # int64_val, packed
        if a.dtype == np.int32:
            return b"".join(B._pb_varint(7, int(v)) for v in a.ravel())            # int_val, unpacked
        raise TypeError(a.dtype)
    metas, items = b"", []
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        full = b"".join(B._pb_bytes(1, b"") for _ in a.shape)                       # TensorSliceProto: full extents
        sm = B._pb_bytes(1, name.encode()) + B._pb_bytes(2, B._encode_shape(a.shape)) + \
            B._pb_varint(3, B._DT_OF[a.dtype]) + B._pb_bytes(4, full)
        metas += B._pb_bytes(1, sm)
        sl = B._pb_bytes(1, name.encode()) + B._pb_bytes(2, full) + B._pb_bytes(3, tensor_proto(a))
        items.append((b"\x00" + name.encode() + b"\x00\x01", B._pb_bytes(2, sl)))   # stand-in for the ordered-code key
    items = [(b"", B._pb_bytes(1, metas))] + sorted(items)
    B.write_table(path, items)
    if compress:                                   # rewrite every block as a snappy literal stream
        raw = open(path, "rb").read()
        import struct as st
        footer = raw[-48:]
        _, pos = B._get_varint(footer, 0)
        _, pos = B._get_varint(footer, pos)
        ioff, pos = B._get_varint(footer, pos)
        isize, pos = B._get_varint(footer, pos)
        out = bytearray()
        index = B._BlockBuilder()

        def emit(block, ctype):
            off = len(out)
            out.extend(block)
            out.append(ctype)
            out.extend(st.pack("<I", B.mask_crc(B.crc32c(bytes(block) + bytes([ctype])))))
            return off, len(block)

        def snappy_literal(b):
            o = bytearray(B._put_varint(len(b)))
            for i in range(0, len(b), 60):
                chunk = b[i:i + 60]
                o.append((len(chunk) - 1) << 2)
                o.extend(chunk)
            return bytes(o)
        for key, handle in B._block_entries(B._read_block(raw, ioff, isize)):
            boff, p2 = B._get_varint(handle, 0)
            bsize, _ = B._get_varint(handle, p2)
            off, size = emit(snappy_literal(B._read_block(raw, boff, bsize)), 1)
            index.add(key, B._put_varint(off) + B._put_varint(size))
        moff, msize = emit(B._BlockBuilder().finish(), 0)
        ioff2, isize2 = emit(index.finish(), 0)
        f = B._put_varint(moff) + B._put_varint(msize) + B._put_varint(ioff2) + B._put_varint(isize2)
        out.extend(f + b"\x00" * (40 - len(f)) + st.pack("<Q", B.TABLE_MAGIC))
        open(path, "wb").write(out)


@pytest.mark.parametrize("compress", [False, True])
def test_v1_model_zoo_checkpoint_reader(tmp_path, compress):
    rng = np.random.default_rng(2)
    sd = {"resnet_v1_50/conv1/weights": rng.standard_normal((7, 7, 3, 64)).astype(np.float32),
          "resnet_v1_50/block1/unit_1/bottleneck_v1/conv1/BatchNorm/gamma": np.ones(64, np.float32),
          "global_step": np.asarray(12345, np.int64), "counts": np.arange(6, dtype=np.int32).reshape(2, 3)}
    path = str(tmp_path / "resnet_v1_50.ckpt")
    _v1_file(path, sd, compress)
    assert B.is_v1_checkpoint(path)
    out = B.read_v1_checkpoint(path)
    assert set(out) == set(sd)
    for k in sd:
        assert out[k].dtype == np.asarray(sd[k]).dtype and np.array_equal(out[k], sd[k])
    got, step = C.load_tf_checkpoint(path)                 # the --pretrained_model_path route
    assert step == 12345 and np.array_equal(got["resnet_v1_50/conv1/weights"], sd["resnet_v1_50/conv1/weights"])
