"""tf.train.Saver checkpoint files (format V2, the "tensor bundle"), read and written without TensorFlow.

The reference saves and restores its session with `tf.train.Saver` (main.py:604 `saver = tf.train.Saver(max_to_keep=50)`,
:612 `saver.restore(sess, model_path + '/' + best_model_name)`, :666 `saver.save(sess, ..., global_step=epoch)`).  With
TF >= 1.0 that writes, per checkpoint prefix P,

    P.index                   an SSTable (LevelDB table format) keyed by variable name:
                                ""      -> BundleHeaderProto  {num_shards, endianness, version}
                                <name>  -> BundleEntryProto   {dtype, shape, shard_id, offset, size, crc32c}
    P.data-00000-of-00001     the raw little-endian tensor bytes, concatenated in key order

(and P.meta, the serialized graph, which carries no values and is neither needed nor written here).  This module
implements exactly those two files -- the table format (prefix-compressed blocks with restart points, per-block
masked CRC-32C trailers, metaindex + index blocks, 48-byte footer), the two protobuf messages, and CRC-32C -- so that a
trained checkpoint of the reference maps 1:1 onto `Engine.load_params` (the variable names are the keys of both) and
a training session of this framework can be written back in the reference's format.

TensorFlow is not installed in this build's environment, so the writer is checked against the reader and against the
published format constants (magic number, CRC mask, field numbers), not against TensorFlow itself.

Format references: leveldb doc/table_format.md; tensorflow/core/lib/io/{format,block,table}.cc;
tensorflow/core/protobuf/tensor_bundle.proto; tensorflow/core/util/tensor_bundle/tensor_bundle.cc.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE = 262144            # tensorflow table::Options default used by BundleWriter
RESTART_INTERVAL = 16
CRC_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('bool')}
DTYPE_ENUM = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('uint8'): 4, np.dtype('int8'): 6,
              np.dtype('int64'): 9, np.dtype('bool'): 10}


# ------------------------------------------------------------------------------------------------------------ CRC-32C
def _make_table():
    poly = 0x82F63B78          # Castagnoli, reflected
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if c & 1 else 0)
        t[i] = c
    return t


_TABLE = _make_table()
_native = None


def _native_crc():
    """libjcm exports a hardware CRC-32C (jcm_crc32c); used when the library is built, for the 226 MB data file."""
    global _native
    if _native is None:
        _native = False
        try:
            import ctypes
            from . import _lib
            if os.path.exists(_lib.LIB_PATH):
                lib = ctypes.CDLL(_lib.LIB_PATH)
                fn = lib.jcm_crc32c
                fn.restype = ctypes.c_uint32
                fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
                _native = fn
        except (OSError, AttributeError):
            _native = False
    return _native


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of `data` (bytes-like), continuing from `crc`."""
    buf = np.frombuffer(memoryview(data).cast('B'), np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    fn = _native_crc()
    if fn and buf.size >= 64:
        return int(fn(buf.ctypes.data, buf.size, crc))
    c = (~crc) & 0xffffffff
    table = _TABLE
    for b in buf.tolist():
        c = int(table[(c ^ b) & 0xff]) ^ (c >> 8)
    return (~c) & 0xffffffff


def mask_crc(crc):
    """leveldb/TF store CRCs 'masked' (crc32c::Mask): rotate right by 15 and add a constant."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + CRC_MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - CRC_MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------------------ varints / protobuf
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Iterate (field number, wire type, value) of a serialized protobuf message."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield num, wt, v


def encode_header(num_shards=1):
    """BundleHeaderProto {num_shards = 1; endianness = LITTLE (0, default, not serialized); version {producer = 1}}."""
    out = bytearray()
    out += b'\x08'
    _put_varint(out, num_shards)
    out += b'\x1a\x02\x08\x01'
    return bytes(out)


def encode_entry(dtype_enum, shape, offset, size, crc_masked, shard_id=0):
    """BundleEntryProto: 1 dtype, 2 shape (TensorShapeProto: repeated dim = 2 {size = 1}), 3 shard_id, 4 offset, 5 size,
    6 crc32c (fixed32)."""
    sh = bytearray()
    for d in shape:
        dim = bytearray(b'\x08')
        _put_varint(dim, int(d))
        sh += b'\x12'
        _put_varint(sh, len(dim))
        sh += dim
    out = bytearray(b'\x08')
    _put_varint(out, dtype_enum)
    out += b'\x12'
    _put_varint(out, len(sh))
    out += sh
    if shard_id:
        out += b'\x18'
        _put_varint(out, shard_id)
    if offset:
        out += b'\x20'
        _put_varint(out, offset)
    if size:
        out += b'\x28'
        _put_varint(out, size)
    out += b'\x35' + struct.pack('<I', crc_masked)
    return bytes(out)


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': False}
    for num, _wt, v in _fields(buf):
        if num == 1:
            e['dtype'] = v
        elif num == 2:
            for n2, _w2, dim in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _w3, s in _fields(dim):
                        if n3 == 1:
                            size = s if s < (1 << 63) else s - (1 << 64)
                    e['shape'].append(size)
        elif num == 3:
            e['shard_id'] = v
        elif num == 4:
            e['offset'] = v
        elif num == 5:
            e['size'] = v
        elif num == 6:
            e['crc32c'] = v
        elif num == 7:
            e['slices'] = True
    return e


# ------------------------------------------------------------------------------------------------------------ SSTable
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            n = min(len(self.last_key), len(key))
            while shared < n and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last_key = key
        self.counter += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _block_handle(offset, size):
    out = bytearray()
    _put_varint(out, offset)
    _put_varint(out, size)
    return bytes(out)


def write_table(path, items, block_size=BLOCK_SIZE):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order -> an uncompressed SSTable."""
    with open(path, 'wb') as fh:
        pos = 0

        def emit(block):
            nonlocal pos
            trailer = b'\x00' + struct.pack('<I', mask_crc(crc32c(block + b'\x00')))        # type 0 = no compression
            fh.write(block + trailer)
            handle = _block_handle(pos, len(block))
            pos += len(block) + 5
            return handle

        index = _BlockBuilder(1)
        data = _BlockBuilder(RESTART_INTERVAL)
        prev = None
        for key, value in items:
            if prev is not None and not key > prev:
                raise ValueError('table keys must be strictly increasing (%r after %r)' % (key, prev))
            prev = key
            data.add(key, value)
            if data.size() >= block_size:
                index.add(data.last_key, emit(data.finish()))
                data = _BlockBuilder(RESTART_INTERVAL)
        if data.buf:
            index.add(data.last_key, emit(data.finish()))
        meta_handle = emit(_BlockBuilder(RESTART_INTERVAL).finish())                     # empty metaindex block
        index_handle = emit(index.finish())
        footer = meta_handle + index_handle
        footer += b'\x00' * (40 - len(footer))
        footer += struct.pack('<II', TABLE_MAGIC & 0xffffffff, TABLE_MAGIC >> 32)
        fh.write(footer)


def _snappy_uncompress(buf):
    """Snappy raw format (tables written with kSnappyCompression carry block type 1)."""
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
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
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
            off = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy block')
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('corrupt snappy block (length)')
    return bytes(out)


def _read_block(buf, offset, size, verify=True):
    raw = buf[offset:offset + size]
    typ = buf[offset + size]
    stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(buf[offset:offset + size + 1]):
        raise ValueError('table block at %d fails its checksum' % offset)
    if typ == 0:
        return raw
    if typ == 1:
        return _snappy_uncompress(raw)
    raise ValueError('unknown table block type %d' % typ)


def _block_entries(block):
    nrest = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrest
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of an SSTable, in order."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    if len(buf) < 48 or struct.unpack_from('<II', buf, len(buf) - 8) != (TABLE_MAGIC & 0xffffffff, TABLE_MAGIC >> 32):
        raise ValueError('%s is not an SSTable (bad magic number)' % path)
    foot = buf[len(buf) - 48:]
    _mo, p = _get_varint(foot, 0)
    _ms, p = _get_varint(foot, p)
    io, p = _get_varint(foot, p)
    isz, p = _get_varint(foot, p)
    out = []
    for _k, handle in _block_entries(_read_block(buf, io, isz, verify)):
        off, q = _get_varint(handle, 0)
        sz, q = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(buf, off, sz, verify)))
    return out


# ------------------------------------------------------------------------------------------------------------ tensor bundle
def save_checkpoint(prefix, tensors):
    """Write `tensors` {variable name: array} as the checkpoint `prefix` (prefix.index + prefix.data-00000-of-00001),
    the files `tf.train.Saver.save(sess, prefix)` produces (main.py:666).  float32 stays float32, Python / numpy integers
    become int32 scalars (n_iters)."""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    items = [(b'', encode_header(1))]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as fh:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])
            if a.dtype == np.float64 or a.dtype == np.float16:
                a = a.astype(np.float32)
            if a.dtype not in DTYPE_ENUM:
                raise TypeError('%s: dtype %s has no checkpoint encoding here' % (name, a.dtype))
            shape = a.shape                                    # () for scalars: np.ascontiguousarray would make it (1,)
            raw = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'))).tobytes()
            fh.write(raw)
            items.append((name.encode('utf-8'), encode_entry(DTYPE_ENUM[a.dtype], shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', items)


def list_variables(prefix):
    """[(name, shape, numpy dtype)] of a checkpoint, like tf.train.list_variables."""
    out = []
    for key, value in read_table(prefix + '.index'):
        if key == b'':
            continue
        e = decode_entry(value)
        out.append((key.decode('utf-8'), tuple(e['shape']), DTYPES.get(e['dtype'])))
    return out


def load_checkpoint(prefix, names=None, verify=True):
    """{variable name: array} of the checkpoint `prefix` (what `saver.restore(sess, prefix)` assigns, main.py:612).
    `names`: restrict to these variables.  Every tensor's CRC-32C is checked unless verify is False."""
    entries = read_table(prefix + '.index', verify)
    if not entries or entries[0][0] != b'':
        raise ValueError('%s.index has no bundle header' % prefix)
    num_shards, endianness = 1, 0
    for num, _wt, v in _fields(entries[0][1]):
        if num == 1:
            num_shards = v
        elif num == 2:
            endianness = v
    if endianness != 0:
        raise ValueError('big-endian checkpoints are not supported')
    shards = {}
    out = {}
    for key, value in entries[1:]:
        name = key.decode('utf-8')
        if names is not None and name not in names:
            continue
        e = decode_entry(value)
        if e['slices']:
            raise ValueError('%s is a partitioned variable (tensor slices); the reference does not create any' % name)
        if e['dtype'] not in DTYPES:
            raise ValueError('%s has unsupported dtype enum %d' % (name, e['dtype']))
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), dtype=np.uint8, mode='r')
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        dt = DTYPES[e['dtype']]
        count = int(np.prod(e['shape'])) if e['shape'] else 1
        if raw.size != count * dt.itemsize:
            raise ValueError('%s: %d bytes on disk, shape %s needs %d' % (name, raw.size, e['shape'], count * dt.itemsize))
        if verify and e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(np.asarray(raw)):
            raise ValueError('%s fails its CRC-32C' % name)
        out[name] = np.frombuffer(np.asarray(raw).tobytes(), dtype=dt).reshape(e['shape']).copy()
    return out
