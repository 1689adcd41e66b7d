"""CPU: the TensorFlow checkpoint (V2 tensor bundle) reader of SURVEY.md 8f-N1.

No checkpoint written by a real TensorFlow is available (the reference ships none, TF cannot be installed), so these
tests pin the reader against (a) bytes assembled by hand from the published format description, (b) files produced by
the module's own writer and (c) index entries encoded by Google's protobuf runtime from the restated schema (an independent
encoder for the proto layer) -- "parity unpinned", as the module header says."""
import os
import struct

import numpy as np
import pytest

from defensegan_amd import archs, synth, tf_checkpoint as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C
    assert T.crc32c(b"") == 0
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"123456789") == 0xE3069283
    # incremental == one-shot ; mask is a rotation plus a constant (crc32c.h)
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283
    assert T.mask_crc(0) == 0xA282EAD8


def test_varint_and_proto_wire_format():
    assert T._put_varint(0) == b"\x00" and T._put_varint(300) == b"\xac\x02"
    assert T._get_varint(b"\xac\x02\x07", 0) == (300, 2)
    # BundleEntryProto {dtype: DT_FLOAT(1), shape {dim {size: 5} dim {size: 5}}, offset: 16, size: 100, crc32c: fixed32}
    shape = b"\x12\x02\x08\x05" * 2
    buf = b"\x08\x01" + b"\x12" + bytes([len(shape)]) + shape + b"\x20\x10" + b"\x28\x64" + b"\x35" + struct.pack("<I", 0xDEADBEEF)
    e = T._parse_entry("v", buf)
    assert (e.dtype, e.shape, e.shard, e.offset, e.size, e.crc, e.sliced) == (1, (5, 5), 0, 16, 100, 0xDEADBEEF, False)


def test_entries_encoded_by_the_protobuf_runtime():
    """BundleEntryProto / TensorShapeProto bytes produced by Google's protobuf runtime (not by this repository's writer) from the
    published schema (tensorflow/core/protobuf/tensor_bundle.proto, tensor_shape.proto: field numbers restated below) are read by
    the hand-written parser: large dims and offsets (multi-byte varints), a named dim, an unknown field a newer writer might add,
    a sliced entry.  An independent ENCODER for the proto layer; the schema itself stays a restatement."""
    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="dg_tensor_bundle_restated.proto", package="dgt", syntax="proto3")
    shape = fd.message_type.add(name="TensorShapeProto")
    dim = shape.nested_type.add(name="Dim")
    dim.field.add(name="size", number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    dim.field.add(name="name", number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    shape.field.add(name="dim", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".dgt.TensorShapeProto.Dim")
    shape.field.add(name="unknown_rank", number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
    sl = fd.message_type.add(name="TensorSliceProto")
    ext = sl.nested_type.add(name="Extent")
    ext.field.add(name="start", number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ext.field.add(name="length", number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    sl.field.add(name="extent", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".dgt.TensorSliceProto.Extent")
    ent = fd.message_type.add(name="BundleEntryProto")
    ent.field.add(name="dtype", number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)            # enum DataType on the wire: a varint
    ent.field.add(name="shape", number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".dgt.TensorShapeProto")
    ent.field.add(name="shard_id", number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
    ent.field.add(name="offset", number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name="size", number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ent.field.add(name="crc32c", number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
    ent.field.add(name="slices", number=7, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".dgt.TensorSliceProto")
    ent.field.add(name="from_the_future", number=15, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)  # (not in the schema: must be skipped)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    desc = pool.FindMessageTypeByName("dgt.BundleEntryProto")
    Entry = message_factory.GetMessageClass(desc) if hasattr(message_factory, "GetMessageClass") else message_factory.MessageFactory(pool).GetPrototype(desc)

    m = Entry(dtype=1, shard_id=0, offset=(1 << 33) + 12345, size=5 * 5 * 64 * 128 * 4, crc32c=0xFEEDC0DE, from_the_future="x" * 200)
    for d in (5, 5, 64, 128):
        m.shape.dim.add(size=d)
    m.shape.dim[3].name = "filters_out"
    e = T._parse_entry("Generator.2.Filters", m.SerializeToString())
    assert (e.dtype, e.shape, e.shard, e.offset, e.size, e.crc, e.sliced) == (1, (5, 5, 64, 128), 0, (1 << 33) + 12345, 819200, 0xFEEDC0DE, False)

    m0 = Entry(dtype=1, size=4, crc32c=1)                      # a scalar: no dims, defaults (shard 0, offset 0) left off the wire
    e0 = T._parse_entry("s", m0.SerializeToString())
    assert (e0.shape, e0.shard, e0.offset, e0.size) == ((), 0, 0, 4)

    ms = Entry(dtype=1, shard_id=2, offset=7, size=64, crc32c=3)
    ms.shape.dim.add(size=4096)
    ms.slices.add().extent.add(start=0, length=16)
    es = T._parse_entry("partitioned", ms.SerializeToString())
    assert es.sliced and es.shard == 2 and es.shape == (4096,)


def test_hand_assembled_block_with_prefix_compression():
    # entries: "Generator.2" -> "a", "Generator.3" (shares 10 bytes) -> "bc", restart, "z" -> ""
    blk = b"\x00\x0b\x01Generator.2a" + b"\x0a\x01\x023bc"
    r2 = len(blk)
    blk += b"\x00\x01\x00z"
    blk += struct.pack("<III", 0, r2, 2)
    assert T._parse_block(blk) == [(b"Generator.2", b"a"), (b"Generator.3", b"bc"), (b"z", b"")]


def test_snappy_literal_and_copy():
    # "abcdabcdabcd": literal "abcd" + copy(offset 4, length 8)
    src = bytes([12]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4])
    assert T._snappy_uncompress(src) == b"abcdabcdabcd"


def _ref_style_tensors(arch="mnist", use_bn=False):
    p = synth.make_weights(arch, seed=7, gain=1.0, bias_range=0.1, use_bn=use_bn, bn_jitter=0.1 if use_bn else 0.0)
    t = {}
    for name, a in p.items():
        scope = name.rsplit(".", 1)[0]                       # Generator.Input.W -> scope Generator.Input
        t["%s/%s" % (scope, name)] = a
        t["%s/%s/Adam" % (scope, name)] = np.zeros_like(a)   # optimizer slots share the file
        t["%s/%s/Adam_1" % (scope, name)] = np.ones_like(a)
    rs = np.random.RandomState(3)
    for i in range(40):                                      # enough keys for several 4 KB index blocks
        t["Discriminator.%d/Discriminator.%d.Filters" % (i, i)] = rs.standard_normal((3, 3, 2, 2)).astype(np.float32)
    t["global_step"] = np.array(12345, np.int64)
    t["beta1_power"] = np.array(0.5, np.float32)
    return p, t


@pytest.mark.parametrize("arch,use_bn", [("mnist", False), ("celeba", True)])
def test_write_read_round_trip_and_generator_selection(tmp_path, arch, use_bn):
    p, t = _ref_style_tensors(arch, use_bn)
    prefix = str(tmp_path / "ckpt" / "GAN.model-12345")
    T.write_checkpoint(prefix, t)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    with open(prefix + ".index", "rb") as f:
        raw = f.read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57
    listed = T.list_variables(prefix)
    assert [n for n, _, _ in listed] == sorted(t, key=lambda s: s.encode())
    assert dict((n, s) for n, s, _ in listed)["global_step"] == ()
    back = T.read_checkpoint(prefix)
    assert set(back) == set(t)
    for k in t:
        assert back[k].dtype == t[k].dtype and back[k].shape == t[k].shape and np.array_equal(back[k], t[k]), k
    # directory / prefix / index-file spellings all resolve (base_model.py:311-323)
    d = str(tmp_path / "ckpt")
    assert T.latest_checkpoint(d) == prefix
    for spelling in (d, prefix, prefix + ".index", prefix + ".data-00000-of-00001"):
        assert T.resolve_prefix(spelling) == prefix
    a = archs.make_arch(arch)
    got = T.generator_weights(d, archs.weight_shapes(a, use_bn).keys())
    assert set(got) == set(p)
    for k in p:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], p[k]), k


def test_corruption_and_errors_are_reported(tmp_path):
    _, t = _ref_style_tensors("mnist")
    prefix = str(tmp_path / "m")
    T.write_checkpoint(prefix, t, write_state=False)
    assert T.latest_checkpoint(str(tmp_path)) is None
    with pytest.raises(T.CheckpointError):
        T.resolve_prefix(str(tmp_path))
    with pytest.raises(T.CheckpointError):
        T.read_checkpoint(prefix, names=["no/such/variable"])
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[100] ^= 0x40
    open(data, "wb").write(bytes(raw))
    with pytest.raises(T.CheckpointError, match="checksum"):
        T.read_checkpoint(prefix)
    assert len(T.read_checkpoint(prefix, verify=False)) == len(t)       # explicit opt-out still reads
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(T.CheckpointError):
        T.list_variables(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(T.CheckpointError):
        T.list_variables(prefix)


def test_bn_parameters_keep_dims_shapes_load(tmp_path):
    """tflib Batchnorm saves scale / offset with the keep_dims shape of the moments -- [1,4096] for BN1, [1,1,1,C] for
    BN2 / BN3 (/root/reference/tflib/ops/batchnorm.py:83-89): a USE_BN checkpoint of the reference must load."""
    from defensegan_amd.gan import dataset_gan_dict
    p, t = _ref_style_tensors("mnist", use_bn=True)
    for full in list(t):
        leaf = full.split("/")[-1]
        if leaf.endswith(".scale") or leaf.endswith(".offset"):
            if full.endswith("/Adam") or full.endswith("/Adam_1"):
                continue
            c = t[full].size
            t[full] = t[full].reshape((1, c) if "BN1" in leaf else (1, 1, 1, c))
    T.write_checkpoint(str(tmp_path / "GAN.model-1"), t)
    gan = dataset_gan_dict["mnist"](cfg={"USE_BN": True}, test_mode=True)
    assert gan.load_generator(str(tmp_path)) is True and gan.initialized
    for k in p:
        assert gan._weights[k].shape == p[k].shape and np.array_equal(gan._weights[k], p[k]), k
    with pytest.raises(ValueError, match="BN2.scale"):
        gan.set_weights({"Generator.BN2.scale": np.ones((1, 1, 1, 127), np.float32)})
