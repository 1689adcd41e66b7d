"""TensorFlow "tensor bundle" (checkpoint V2) reader, so that generators trained with the reference load without
TensorFlow -- SURVEY.md section 8f row N1.

The reference saves its generator with ``tf.train.Saver`` over ``slim.get_variables('Generator')``
(/root/reference/models/gan.py:80-87) and restores it through ``tf.train.get_checkpoint_state`` + ``saver.restore``
(/root/reference/models/base_model.py:294-335).  TensorFlow itself (1.7, README.md:46) is a third-party dependency
that is absent from /root/reference and cannot be installed here, so the on-disk format is restated from its published
source: ``tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}``, ``tensorflow/core/protobuf/tensor_bundle.proto``,
``tensorflow/core/lib/io/{format,block,block_builder,table_builder}.cc`` and ``table_format.txt`` (the LevelDB table
format).  **No checkpoint written by a real TensorFlow exists in this environment: the reader is pinned only against
files produced by ``write_checkpoint`` below (same restated format) and hand-assembled bytes -- "parity unpinned".**

A checkpoint ``<prefix>`` is ``<prefix>.index`` + ``<prefix>.data-SSSSS-of-NNNNN``:

* ``.index`` is a LevelDB-style sorted string table.  Footer (last 48 bytes): metaindex BlockHandle, index BlockHandle
  (two varint64 each: offset, size), zero padding to 40 bytes, magic ``0xdb4775248b80fb57`` little endian.  A block is
  ``entries | restart offsets (uint32 each) | num_restarts (uint32)`` followed on disk by a 1-byte compression type
  (0 = none, 1 = snappy) and a masked CRC32C; an entry is ``varint32 shared | varint32 non_shared | varint32 value_len |
  key suffix | value`` (keys are prefix-compressed against the previous key, restarts every 16 entries).  The index
  block maps separator keys to the BlockHandles of the data blocks.
* key ``""`` holds a ``BundleHeaderProto`` {1: num_shards, 2: endianness (0 = little), 3: version}; every other key is
  a variable name holding a ``BundleEntryProto`` {1: dtype, 2: TensorShapeProto {2: dim {1: size}}, 3: shard_id,
  4: offset, 5: size, 6: crc32c (fixed32, masked), 7: slices}.
* the data shard holds the raw little-endian tensor bytes at [offset, offset + size).
"""
import os
import re
import struct
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 19: np.float16, 17: np.uint16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------- crc32c
def _make_crc_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return [int(v) for v in t]


_CRC_TABLE = _make_crc_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli), the checksum of the table blocks and of every tensor."""
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    """core/lib/hash/crc32c.h Mask(): rotate right by 15 and add a constant."""
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- varints / protobuf
def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = 0
    val = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if b < 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _put_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Wire-level protobuf parse: field number -> list of raw values (int for varint/fixed, bytes for length-delimited)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise CheckpointError("truncated protobuf field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _field(tag: int, wt: int, payload: bytes) -> bytes:
    return _put_varint((tag << 3) | wt) + payload


# ---------------------------------------------------------------------------------------------- table blocks
def _parse_block(block: bytes) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise CheckpointError("block too small")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise CheckpointError("bad restart array")
    out = []
    key = b""
    pos = 0
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError("corrupt block entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def _read_block(f: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset + size + 5 > len(f):
        raise CheckpointError("block handle beyond end of file")
    raw = f[offset:offset + size]
    ctype = f[offset + size]
    if verify:
        want = struct.unpack_from("<I", f, offset + size + 1)[0]
        if mask_crc(crc32c(f[offset:offset + size + 1])) != want:
            raise CheckpointError("index block checksum mismatch")
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_uncompress(raw)
    raise CheckpointError("unknown block compression %d" % ctype)


def _snappy_uncompress(src: bytes) -> bytes:
    """Raw snappy (format_description.txt): only needed if a writer enabled table compression (TF's BundleWriter does not)."""
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy stream")
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as fh:
        f = fh.read()
    if len(f) < 48:
        raise CheckpointError("%s: too small for a table footer" % path)
    footer = f[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise CheckpointError("%s: bad table magic (not a checkpoint V2 index)" % path)
    pos = 0
    _, pos = _get_varint(footer, pos)            # metaindex handle (unused by tensor bundles)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out: List[Tuple[bytes, bytes]] = []
    for _, handle in _parse_block(_read_block(f, ioff, isize, verify)):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        out.extend(_parse_block(_read_block(f, boff, bsize, verify)))
    return out


# ---------------------------------------------------------------------------------------------- bundle
class _Entry(object):
    __slots__ = ("name", "dtype", "shape", "shard", "offset", "size", "crc", "sliced")


def _parse_entry(name: str, buf: bytes) -> _Entry:
    p = _parse_proto(buf)
    e = _Entry()
    e.name = name
    e.dtype = int(p.get(1, [0])[0])
    shape = []
    for sp in p.get(2, []):
        for dim in _parse_proto(sp).get(2, []):
            sz = _parse_proto(dim).get(1, [0])[0]
            shape.append(sz - (1 << 64) if sz >= (1 << 63) else sz)
    e.shape = tuple(int(s) for s in shape)
    e.shard = int(p.get(3, [0])[0])
    e.offset = int(p.get(4, [0])[0])
    e.size = int(p.get(5, [0])[0])
    e.crc = int(p.get(6, [0])[0])
    e.sliced = 7 in p
    return e


def _index(prefix: str, verify: bool):
    idx = prefix + ".index"
    if not os.path.exists(idx):
        raise CheckpointError("%s not found (a V1 checkpoint or a wrong prefix?)" % idx)
    kv = read_table(idx, verify)
    if not kv or kv[0][0] != b"":
        raise CheckpointError("%s: bundle header entry missing" % idx)
    hdr = _parse_proto(kv[0][1])
    num_shards = int(hdr.get(1, [1])[0])
    if int(hdr.get(2, [0])[0]) != 0:
        raise CheckpointError("big-endian bundles are not supported")
    entries = [_parse_entry(k.decode("utf-8"), v) for k, v in kv[1:]]
    return num_shards, entries


def list_variables(prefix: str) -> List[Tuple[str, Tuple[int, ...], np.dtype]]:
    """(name, shape, dtype) of every tensor in the checkpoint, in name order (tf.train.list_variables)."""
    _, entries = _index(prefix, True)
    return [(e.name, e.shape, np.dtype(_DTYPES[e.dtype]) if e.dtype in _DTYPES else None) for e in entries]


def read_checkpoint(prefix: str, names: Optional[Iterable[str]] = None,
                    name_filter: Optional[Callable[[str], bool]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """name -> array for the selected tensors (all when neither ``names`` nor ``name_filter`` is given).
    ``verify`` checks the CRC32C of the index blocks and of every tensor read."""
    num_shards, entries = _index(prefix, verify)
    want = set(names) if names is not None else None
    shards: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}
    for e in entries:
        if want is not None and e.name not in want:
            continue
        if name_filter is not None and not name_filter(e.name):
            continue
        if e.sliced:
            raise CheckpointError("%s is a partitioned variable (slices are not supported)" % e.name)
        if e.dtype not in _DTYPES:
            raise CheckpointError("%s: unsupported dtype enum %d" % (e.name, e.dtype))
        dt = np.dtype(_DTYPES[e.dtype])
        count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if count * dt.itemsize != e.size:
            raise CheckpointError("%s: size %d does not match shape %r" % (e.name, e.size, e.shape))
        if e.shard not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, e.shard, num_shards)
            if not os.path.exists(path):
                raise CheckpointError("%s not found" % path)
            shards[e.shard] = np.memmap(path, dtype=np.uint8, mode="r")
        raw = shards[e.shard][e.offset:e.offset + e.size]
        if len(raw) != e.size:
            raise CheckpointError("%s: data shard too short" % e.name)
        raw = bytes(raw)
        if verify and mask_crc(crc32c(raw)) != e.crc:
            raise CheckpointError("%s: tensor checksum mismatch" % e.name)
        out[e.name] = np.frombuffer(raw, dtype=dt.newbyteorder("<")).reshape(e.shape).astype(dt)
    if want is not None and want - set(out):
        raise CheckpointError("not in checkpoint: %s" % ", ".join(sorted(want - set(out))))
    return out


def latest_checkpoint(checkpoint_dir: str) -> Optional[str]:
    """``tf.train.latest_checkpoint``: the prefix named by ``model_checkpoint_path`` in the text-format
    ``checkpoint`` state file (relative paths are relative to the directory), as base_model.py:318-323 resolves it."""
    state = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state, "r") as fh:
        m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', fh.read(), re.M)
    if not m:
        return None
    # the reference re-joins the basename with the directory it was given (base_model.py:321-323)
    prefix = os.path.join(checkpoint_dir, os.path.basename(m.group(1)))
    return prefix if os.path.exists(prefix + ".index") else None


def resolve_prefix(path: str) -> str:
    """A directory (state file), a prefix, or any of the prefix's files -> the checkpoint prefix."""
    if os.path.isdir(path):
        p = latest_checkpoint(path)
        if p is None:
            raise CheckpointError("no checkpoint state / index file in %s" % path)
        return p
    m = re.match(r"^(.*)\.(index|data-\d{5}-of-\d{5})$", path)
    if m:
        path = m.group(1)
    if not os.path.exists(path + ".index"):
        raise CheckpointError("%s.index not found" % path)
    return path


def generator_weights(path: str, expected: Iterable[str], verify: bool = True) -> Dict[str, np.ndarray]:
    """The generator parameters of a reference checkpoint, keyed by tflib parameter name.

    Variables live under name scopes (``Generator.Input/Generator.Input.W`` ...); optimizer slots
    (``.../Adam``) and the other networks share the file.  A variable is taken when the LAST component of its name is an
    expected parameter name; the shortest full name wins if several match."""
    prefix = resolve_prefix(path)
    expected = list(expected)
    exp = set(expected)
    best: Dict[str, str] = {}
    for name, _, _ in list_variables(prefix):
        leaf = name.split("/")[-1]
        if leaf in exp and (leaf not in best or len(name) < len(best[leaf])):
            best[leaf] = name
    got = read_checkpoint(prefix, names=best.values(), verify=verify)
    return {leaf: np.ascontiguousarray(got[full], dtype=np.float32) for leaf, full in best.items()}


# ---------------------------------------------------------------------------------------------- writer
class _BlockBuilder(object):
    def __init__(self, restart_interval: int = 16):
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""
        self.interval = restart_interval

    def add(self, key: bytes, value: bytes) -> None:
        shared = 0
        if self.count < self.interval:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self) -> bytes:
        out = bytes(self.buf)
        for r in self.restarts:
            out += struct.pack("<I", r)
        return out + struct.pack("<I", len(self.restarts))


def _emit_block(fh, contents: bytes) -> bytes:
    off = fh.tell()
    fh.write(contents)
    fh.write(b"\x00")
    fh.write(struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
    return _put_varint(off) + _put_varint(len(contents))


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096, write_state: bool = True) -> None:
    """Writes ``tensors`` as a one-shard V2 bundle in the format ``read_checkpoint`` reads (and, to the best of the
    restatement above, the one TensorFlow reads).  Used by the tests and to hand a weight pack back to the reference."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    items: List[Tuple[bytes, bytes]] = []
    version = _field(1, 0, _put_varint(1))                                  # VersionDef.producer = 1
    items.append((b"", _field(1, 0, _put_varint(1)) + _field(2, 0, _put_varint(0)) + _field(3, 2, _put_varint(len(version)) + version)))
    with open("%s.data-00000-of-00001" % prefix, "wb") as dh:
        for name in names:
            a = np.asarray(tensors[name])                  # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_IDS:
                raise CheckpointError("%s: dtype %s not supported" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes(order="C")
            dims = b"".join(_field(2, 2, _put_varint(len(d)) + d) for d in (_field(1, 0, _put_varint(int(s))) for s in a.shape))
            entry = _field(1, 0, _put_varint(_DTYPE_IDS[a.dtype])) + _field(2, 2, _put_varint(len(dims)) + dims)
            entry += _field(4, 0, _put_varint(dh.tell())) + _field(5, 0, _put_varint(len(raw)))
            entry += _field(6, 5, struct.pack("<I", mask_crc(crc32c(raw))))
            items.append((name.encode("utf-8"), entry))
            dh.write(raw)
    with open(prefix + ".index", "wb") as fh:
        index = _BlockBuilder(restart_interval=1)
        blk = _BlockBuilder()
        last_key = b""
        for key, value in items:
            blk.add(key, value)
            last_key = key
            if len(blk.buf) >= block_size:
                index.add(last_key, _emit_block(fh, blk.finish()))
                blk = _BlockBuilder()
        if blk.buf:
            index.add(last_key, _emit_block(fh, blk.finish()))
        meta = _emit_block(fh, _BlockBuilder().finish())
        ih = _emit_block(fh, index.finish())
        footer = meta + ih
        fh.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    if write_state:
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as sh:
            base = os.path.basename(prefix)
            sh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
