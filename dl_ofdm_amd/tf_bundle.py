"""TensorFlow tensor-bundle checkpoint I/O: read and write the ``<prefix>.index`` +
``<prefix>.data-00000-of-00001`` pair ``tf.train.Saver`` produces (dev/py/ofdmreceiver_np.py:191,271,
dev/py/ofdmreceiver_np_mp.py:337,459), so receivers trained with the reference load here and vice versa.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table_*: a LevelDB-style SSTable):
  .index  = data block(s) of prefix-compressed (key, value) entries with a restart point every 16 entries,
            each block followed by a 1-byte compression type (0) and a masked CRC32C; then an empty metaindex
            block, an index block (short successor of each data block's last key -> block handle) and a
            48-byte footer (two varint block handles, zero padding, magic 0xdb4775248b80fb57).
            key "" -> BundleHeaderProto{num_shards=1, version{producer=1}};
            key <variable name> -> BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c(masked)}.
  .data-00000-of-00001 = the tensors' raw little-endian bytes, in key order.
The writer is pinned byte-for-byte against the reference's own ``test_v1/model/*.index`` files
(tests/test_tf_bundle.py); CRC32C comes from libdccn (``dccn_crc32c``, host code).

No ``.meta`` (MetaGraphDef) is written: it describes a TF graph, which this framework does not have;
``tf.train.load_checkpoint`` / ``Saver.restore`` on an existing graph need only the two files above.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import _lib

MAGIC = 0xDB4775248B80FB57
RESTART_INTERVAL = 16
BLOCK_SIZE = 256 * 1024                      # table::Options::block_size the bundle writer uses
MASK_DELTA = 0xA282EAD8
DT = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
DT_OF = {np.dtype(v): k for k, v in DT.items()}
HEADER = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])      # num_shards=1, version{producer=1}


# ---- primitives ------------------------------------------------------------------------------------
def crc32c(data: bytes, crc: int = 0) -> int:
    return int(_lib.load().dccn_crc32c(crc, data, len(data)))


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m: int) -> int:
    rot = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


# ---- protobuf (just the two messages of the bundle) ----------------------------------------------------
def encode_entry(dtype: int, shape: Iterable[int], offset: int, size: int, crc_masked: int, shard_id: int = 0) -> bytes:
    """BundleEntryProto; proto3 omits zero scalars, the shape submessage is always present."""
    sh = b"".join(b"\x12" + _put_varint(len(d)) + d for d in
                  ((b"\x08" + _put_varint(int(n)) if int(n) else b"") for n in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(sh)) + sh
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    if size:
        out += b"\x28" + _put_varint(size)
    if crc_masked:
        out += b"\x35" + struct.pack("<I", crc_masked)
    return out


def _fields(buf: bytes):
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack("<I", buf[pos:pos + 4])[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack("<Q", buf[pos:pos + 8])[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, v


def decode_entry(buf: bytes) -> dict:
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0)
    for fno, v in _fields(buf):
        if fno == 1:
            e["dtype"] = v
        elif fno == 2:
            for f2, dim in _fields(v):
                if f2 == 2:
                    e["shape"].append(next((d for f3, d in _fields(dim) if f3 == 1), 0))
        elif fno == 3:
            e["shard_id"] = v
        elif fno == 4:
            e["offset"] = v
        elif fno == 5:
            e["size"] = v
        elif fno == 6:
            e["crc32c"] = v
        elif fno == 7:
            raise NotImplementedError("partitioned (sliced) variables are not supported")
    return e


# ---- SSTable -------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self, restart_interval: int = RESTART_INTERVAL):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""
        self.interval = restart_interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _short_successor(key: bytes) -> bytes:
    """leveldb BytewiseComparator::FindShortSuccessor"""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _short_separator(start: bytes, limit: bytes) -> bytes:
    """leveldb BytewiseComparator::FindShortestSeparator"""
    m = min(len(start), len(limit))
    d = 0
    while d < m and start[d] == limit[d]:
        d += 1
    if d < m and start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _emit_block(out: bytearray, contents: bytes) -> bytes:
    """append block + trailer (type 0 = uncompressed, masked crc of contents+type); returns its handle"""
    handle = _put_varint(len(out)) + _put_varint(len(contents))
    out += contents + b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00")))
    return handle


def build_table(items: List[Tuple[bytes, bytes]]) -> bytes:
    """sorted (key, value) pairs -> SSTable bytes (what TableBuilder::Finish leaves in the file)"""
    out, index, blk = bytearray(), _BlockBuilder(1), _BlockBuilder()      # index blocks restart at every entry
    pending: Optional[Tuple[bytes, bytes]] = None          # (last key of the flushed block, its handle)
    for key, value in items:
        if pending is not None:
            index.add(_short_separator(pending[0], key), pending[1])
            pending = None
        blk.add(key, value)
        if blk.size() >= BLOCK_SIZE:
            pending = (blk.last, _emit_block(out, blk.finish()))
            blk = _BlockBuilder()
    if blk.count or not items:
        pending = (blk.last, _emit_block(out, blk.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta_handle = _emit_block(out, _BlockBuilder().finish())
    index_handle = _emit_block(out, index.finish())
    footer = meta_handle + index_handle
    out += footer + bytes(40 - len(footer)) + struct.pack("<Q", MAGIC)
    return bytes(out)


def _read_block(buf: bytes, off: int, size: int, verify: bool = True):
    contents, trailer = buf[off:off + size], buf[off + size:off + size + 5]
    if trailer[0] != 0:
        raise NotImplementedError("compressed SSTable blocks are not supported")
    if verify and unmask_crc(struct.unpack("<I", trailer[1:5])[0]) != crc32c(contents + trailer[:1]):
        raise ValueError("SSTable block checksum mismatch at offset %d" % off)
    n_restarts = struct.unpack("<I", contents[-4:])[0]
    end = len(contents) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(contents, pos)
        non_shared, pos = _get_varint(contents, pos)
        vlen, pos = _get_varint(contents, pos)
        key = key[:shared] + contents[pos:pos + non_shared]
        pos += non_shared
        yield key, contents[pos:pos + vlen]
        pos += vlen


def read_table(buf: bytes, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != MAGIC:
        raise ValueError("not a TensorFlow SSTable (bad magic)")
    footer = buf[-48:]
    _, p = _get_varint(footer, 0)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isz, p = _get_varint(footer, p)
    items = []
    for _, handle in _read_block(buf, ioff, isz, verify):
        doff, q = _get_varint(handle, 0)
        dsz, q = _get_varint(handle, q)
        items.extend(_read_block(buf, doff, dsz, verify))
    return items


# ---- bundles -------------------------------------------------------------------------------------------
def read_index(path: str, verify: bool = True) -> Dict[str, dict]:
    """name -> {dtype, shape, shard_id, offset, size, crc32c} of ``<prefix>.index`` (header entry dropped)"""
    with open(path, "rb") as f:
        items = read_table(f.read(), verify)
    return {k.decode(): decode_entry(v) for k, v in items if k}


def index_bytes(entries: Dict[str, dict]) -> bytes:
    items = [(b"", HEADER)]
    for name in sorted(entries, key=lambda s: s.encode()):
        e = entries[name]
        items.append((name.encode(), encode_entry(e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"],
                                                  e.get("shard_id", 0))))
    return build_table(items)


def data_path(prefix: str) -> str:
    return prefix + ".data-00000-of-00001"


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], state_file: bool = True) -> str:
    """tf.train.Saver().save(sess, prefix): variables -> ``prefix.index`` + ``prefix.data-00000-of-00001``
    (+ the ``checkpoint`` state file next to them)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    entries, offset = {}, 0
    with open(data_path(prefix), "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])                       # (ascontiguousarray would turn scalars into [1])
            if a.dtype not in DT_OF:
                raise TypeError("%s: dtype %s has no TensorFlow DataType mapping here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C")
            f.write(raw)
            entries[name] = dict(dtype=DT_OF[a.dtype], shape=list(a.shape), shard_id=0, offset=offset, size=len(raw),
                                 crc32c=mask_crc(crc32c(raw)))
            offset += len(raw)
    with open(prefix + ".index", "wb") as f:
        f.write(index_bytes(entries))
    if state_file:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix


def read_checkpoint(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    """tf.train.load_checkpoint(prefix): name -> array (checksums verified)"""
    entries = read_index(prefix + ".index", verify)
    if any(e["shard_id"] != 0 for e in entries.values()):
        raise NotImplementedError("multi-shard bundles are not supported")
    with open(data_path(prefix), "rb") as f:
        blob = f.read()
    out = {}
    for name, e in entries.items():
        raw = blob[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("%s: data file truncated" % name)
        if verify and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("%s: tensor checksum mismatch" % name)
        if e["dtype"] not in DT:
            raise TypeError("%s: TensorFlow DataType %d not supported" % (name, e["dtype"]))
        out[name] = np.frombuffer(raw, dtype=np.dtype(DT[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    return out


# ---- the receiver's variables <-> TF layouts -------------------------------------------------------------
def rx_to_tf(params: Dict[str, np.ndarray], kin: int) -> Dict[str, np.ndarray]:
    """engine layout -> the shapes tf.layers created: the live C-Conv tap goes to
    position (K-1)//2 of the [1,K,1,K,2F] conv3d kernel (dead taps zero: they never meet data), the 1x1 conv2d
    kernel becomes [1,1,2,m].  Applies to the variables and, key by key, to their Adam slots."""
    out = {}
    for n, a in params.items():
        a = np.asarray(a)
        base = n.split("/Adam")[0]
        if base == "fft_like/conv3d/kernel":
            full = np.zeros((1, kin, 1, kin, a.shape[-1]), dtype=a.dtype)
            full[0, (kin - 1) // 2, 0] = a.reshape(kin, -1)
            a = full
        elif base == "demodulation/conv2d/kernel":
            a = a.reshape(1, 1, *a.shape[-2:])
        out[n] = a
    return out


RX_TRAINABLES = ("fft_like/conv3d/kernel", "fft_like/conv3d/bias", "demodulation/dense/kernel", "demodulation/dense/bias",
                 "demodulation/conv2d/kernel", "demodulation/conv2d/bias", "demodulation/dense_1/kernel",
                 "demodulation/dense_1/bias")
_OPT_SCALARS = ("global_step", "beta1_power", "beta2_power")


def rx_from_tf(tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """TF layouts -> engine layouts, for the model versions this implementation can run:

      * the dev receiver (dev/py/model.py:1222-1292): the eight RX_TRAINABLES (+ Adam slots, step, beta powers);
      * the archived v1 receiver (test_v1/model/*): it carries a second 1x1 convolution ``demodulation/conv2d_1``
        directly behind the first, with no activation in between -- the two are one affine map, so they are folded
        into ``demodulation/conv2d`` (kernel w1.w1b, bias b1.w1b + b1b) and the receiver evaluates exactly the same
        function.  The fold is for inference / fine-tuning from fresh optimizer state: the Adam slots of the two
        factors cannot be folded and are dropped (``load_checkpoint`` then leaves the optimizer at its initial state).

    Any other variable (a different topology) raises instead of being silently ignored."""
    known = set(RX_TRAINABLES) | {"demodulation/conv2d_1/kernel", "demodulation/conv2d_1/bias"}
    for n in tensors:
        base = n.split("/Adam")[0]
        if base not in known and n not in _OPT_SCALARS:
            raise ValueError("checkpoint variable %r does not belong to a receiver topology this implementation runs "
                             "(dev ofdm_dense_rx, or v1 with its extra demodulation/conv2d_1)" % n)
    out = {}
    for n, a in tensors.items():
        base = n.split("/Adam")[0]
        if base == "fft_like/conv3d/kernel" and a.ndim == 5:
            a = a[0, (a.shape[1] - 1) // 2, 0]
        elif base in ("demodulation/conv2d/kernel", "demodulation/conv2d_1/kernel") and a.ndim == 4:
            a = a[0, 0]
        out[n] = np.ascontiguousarray(a) if np.ndim(a) else np.asarray(a)
    if "demodulation/conv2d_1/kernel" in out:
        w1, b1 = out["demodulation/conv2d/kernel"].astype(np.float64), out["demodulation/conv2d/bias"].astype(np.float64)
        w1b, b1b = (out["demodulation/conv2d_1/kernel"].astype(np.float64),
                    out["demodulation/conv2d_1/bias"].astype(np.float64))
        out["demodulation/conv2d/kernel"] = (w1 @ w1b).astype(np.float32)
        out["demodulation/conv2d/bias"] = (b1 @ w1b + b1b).astype(np.float32)
        for n in list(out):
            if n.startswith("demodulation/conv2d_1/") or n.startswith("demodulation/conv2d/kernel/Adam") or \
                    n.startswith("demodulation/conv2d/bias/Adam"):
                del out[n]
    return out


def eq_to_tf(tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """equaliser store shapes -> TF: conv3d kernels [kL,kW,C,2F] gain the unit depth axis [kL,kW,1,C,2F]"""
    out = {}
    for n, a in tensors.items():
        a = np.asarray(a)
        if "/conv3d" in n and "/kernel" in n and a.ndim == 4:
            a = a.reshape(a.shape[0], a.shape[1], 1, a.shape[2], a.shape[3])
        out[n] = a
    return out


def eq_from_tf(tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    out = {}
    for n, a in tensors.items():
        if "/conv3d" in n and "/kernel" in n and a.ndim == 5:
            a = a.reshape(a.shape[0], a.shape[1], a.shape[3], a.shape[4])
        out[n] = np.ascontiguousarray(a) if np.ndim(a) else np.asarray(a)
    return out
