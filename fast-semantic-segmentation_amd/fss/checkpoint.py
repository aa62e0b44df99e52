"""TensorFlow checkpoint-V2 ("tensor bundle") reader and writer, numpy only.

The reference restores weights with `tf.train.Saver().restore(sess, prefix)`
(inference.py:90-91) and walks `checkpoint` state files (eval.py:56-74).  This
module reads the same files without TensorFlow:

  <prefix>.index                 LevelDB-format SSTable: key "" -> BundleHeaderProto,
                                 key <variable name> -> BundleEntryProto
  <prefix>.data-00000-of-0000N   raw little-endian tensor bytes at [offset, offset+size)

Only what `Saver` writes is supported: uncompressed table blocks, whole
(unsliced) tensors.  The writer produces the same layout and exists so the
loader path can be exercised offline (no released checkpoint is available
here; the on-disk format is restated from the LevelDB table format and
tensor_bundle.proto, not validated against a TensorFlow-written file).
"""
import os
import re
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_FOOTER = 48
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 9: np.int64, 10: np.bool_}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------- crc32c
def _make_table():
    tbl = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl[i] = c
    return tbl


_CRC_TABLE = _make_table()


def _crc_bytes(state, data):
    """Advances the raw CRC register `state` over `data`, one byte at a time."""
    tbl = _CRC_TABLE
    for b in bytes(data):
        state = int(tbl[(state ^ b) & 0xFF]) ^ (state >> 8)
    return state


def _gf2_apply(mat, vec):
    """mat: 32 columns (uint32 each) of a GF(2) matrix; returns mat * vec."""
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _zero_advance_matrix(nbytes):
    """The linear map that advancing the CRC register over `nbytes` zero bytes applies."""
    mat = [_crc_bytes(1 << i, b'\x00') for i in range(32)]       # one zero byte
    result = [1 << i for i in range(32)]                         # identity
    while nbytes:
        if nbytes & 1:
            result = [_gf2_apply(mat, c) for c in result]
        mat = [_gf2_apply(mat, c) for c in mat]
        nbytes >>= 1
    return result


_CRC_BLOCK = 4096
_CRC_ADVANCE = None


def crc32c(data):
    """CRC-32C (Castagnoli) of a bytes-like object.  Large inputs are cut into 4 KiB blocks
    whose raw CRCs (register started at 0) are computed in lock step with numpy - one table
    look-up per byte position for all blocks at once - and chained with the register's
    zero-advance matrix: crc(A | B) = advance_len(B)(crc(A)) xor raw(B), the CRC being linear."""
    global _CRC_ADVANCE
    buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray))
                        else data, dtype=np.uint8)
    n = buf.size
    if n < 4 * _CRC_BLOCK:
        return _crc_bytes(0xFFFFFFFF, buf.tobytes()) ^ 0xFFFFFFFF
    nblk = n // _CRC_BLOCK
    blocks = buf[:nblk * _CRC_BLOCK].reshape(nblk, _CRC_BLOCK)
    reg = np.zeros(nblk, np.uint32)
    tbl = _CRC_TABLE
    for col in range(_CRC_BLOCK):
        reg = tbl[(reg ^ blocks[:, col]) & 0xFF] ^ (reg >> 8)
    if _CRC_ADVANCE is None:
        _CRC_ADVANCE = _zero_advance_matrix(_CRC_BLOCK)
    state = 0xFFFFFFFF
    for r in reg.tolist():
        state = _gf2_apply(_CRC_ADVANCE, state) ^ r
    state = _crc_bytes(state, buf[nblk * _CRC_BLOCK:].tobytes())
    return state ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints
def _read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


# ------------------------------------------------------- minimal proto codec
def _parse_fields(buf):
    """Yields (field_number, wire_type, value) of one protobuf message."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, val


def _parse_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, sliced)."""
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for field, wt, val in _parse_fields(buf):
        if field == 1:
            e['dtype'] = val
        elif field == 2:
            dims = []
            for f2, _, v2 in _parse_fields(val):
                if f2 == 2:      # TensorShapeProto.Dim
                    size = 0
                    for f3, _, v3 in _parse_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    dims.append(size)
            e['shape'] = tuple(dims)
        elif field == 3:
            e['shard_id'] = val
        elif field == 4:
            e['offset'] = val
        elif field == 5:
            e['size'] = val
        elif field == 6:
            e['crc32c'] = struct.unpack('<I', val)[0]
        elif field == 7:
            e['sliced'] = True
    return e


def _encode_entry(dtype_code, shape, shard_id, offset, size, crc):
    shape_msg = b''.join(b'\x12' + _varint(len(d)) + d
                         for d in (b'\x08' + _varint(int(s)) for s in shape))
    out = b'\x08' + _varint(dtype_code) + b'\x12' + _varint(len(shape_msg)) + shape_msg
    if shard_id:
        out += b'\x18' + _varint(shard_id)
    if offset:
        out += b'\x20' + _varint(offset)
    out += b'\x28' + _varint(size) + b'\x35' + struct.pack('<I', crc)
    return out


# ---------------------------------------------------------------- SSTable IO
def _read_block(data, offset, size, verify=True):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack('<I', data[offset + size + 1:offset + size + 5])[0]
        if stored != masked_crc32c(data[offset:offset + size + 1]):
            raise ValueError('checkpoint index: block checksum mismatch')
    if ctype != 0:
        raise NotImplementedError('compressed checkpoint index blocks are not supported')
    return block


def _block_entries(block):
    """Decodes the prefix-compressed (key, value) entries of one table block."""
    num_restarts = struct.unpack('<I', block[-4:])[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(prefix, verify=True):
    """Returns (header dict, {variable name: entry dict}) of <prefix>.index."""
    with open(prefix + '.index', 'rb') as f:
        data = f.read()
    if len(data) < _FOOTER or struct.unpack('<Q', data[-8:])[0] != _MAGIC:
        raise ValueError('%s.index is not a tensor-bundle index (bad magic)' % prefix)
    footer = data[-_FOOTER:]
    pos = 0
    _, pos = _read_varint(footer, pos)       # metaindex handle
    _, pos = _read_varint(footer, pos)
    idx_off, pos = _read_varint(footer, pos)
    idx_size, pos = _read_varint(footer, pos)
    entries, header = {}, None
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, verify)):
        boff, p = _read_varint(handle, 0)
        bsize, _ = _read_varint(handle, p)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b'':
                header = {f: v for f, _, v in _parse_fields(value)}
            else:
                entries[key.decode('utf-8')] = _parse_entry(value)
    if header is None:
        raise ValueError('%s.index has no bundle header' % prefix)
    return header, entries


def load_variables(prefix, wanted=None, verify=True):
    """Reads variables of a V2 checkpoint into {name: ndarray}.

    wanted: iterable of names or {name: shape}; every wanted variable must be
    present (and match the shape), variables not asked for (optimizer slots,
    auxiliary heads, global_step) are ignored.  None = read everything."""
    header, entries = read_index(prefix, verify)
    num_shards = header.get(1, 1)
    if header.get(2, 0) != 0:
        raise NotImplementedError('big-endian checkpoints are not supported')
    names = list(entries) if wanted is None else list(wanted)
    shapes = wanted if isinstance(wanted, dict) else {}
    missing = [n for n in names if n not in entries]
    if missing:
        raise KeyError('checkpoint %s is missing %d variables, e.g. %s'
                       % (prefix, len(missing), missing[:3]))
    shards, out = {}, {}
    for name in names:
        e = entries[name]
        if e['sliced']:
            raise NotImplementedError('sliced (partitioned) variable %s' % name)
        if e['dtype'] not in _DTYPES:
            raise NotImplementedError('dtype code %d of %s' % (e['dtype'], name))
        sid = e['shard_id']
        if sid not in shards:
            path = '%s.data-%05d-of-%05d' % (prefix, sid, num_shards)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode='r')
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if verify and e['crc32c'] is not None:
            if masked_crc32c(raw.tobytes()) != e['crc32c']:
                raise ValueError('checksum mismatch in variable %s' % name)
        arr = np.frombuffer(raw.tobytes(), dtype=_DTYPES[e['dtype']]).reshape(e['shape'])
        if name in shapes and tuple(shapes[name]) != tuple(arr.shape):
            raise ValueError('variable %s has shape %s in the checkpoint, expected %s'
                             % (name, arr.shape, tuple(shapes[name])))
        out[name] = arr
    return out


# -------------------------------------------------------------------- writer
class _BlockBuilder(object):
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts)
        return out + struct.pack('<I', len(self.restarts))


def save_variables(prefix, variables, block_size=4096):
    """Writes {name: ndarray} as <prefix>.index + <prefix>.data-00000-of-00001."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items, offset = [], 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(variables):
            arr = np.asarray(variables[name])
            raw = arr.tobytes()       # C order
            f.write(raw)
            items.append((name.encode('utf-8'),
                          _encode_entry(_DTYPE_CODES[arr.dtype], arr.shape, 0, offset, len(raw),
                                        masked_crc32c(raw))))
            offset += len(raw)
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'     # num_shards=1, version{producer=1}
    items = [(b'', header)] + items
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes + b'\x00')
        out.extend(struct.pack('<I', masked_crc32c(block_bytes + b'\x00')))
        return _varint(off) + _varint(len(block_bytes))

    blk, last_key = _BlockBuilder(), b''
    for key, value in items:
        blk.add(key, value)
        last_key = key
        if len(blk.buf) >= block_size:
            index.add(last_key, emit(blk.finish()))
            blk = _BlockBuilder()
    if blk.count:
        index.add(last_key, emit(blk.finish()))
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    out.extend(footer)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))


# ---------------------------------------------------- `checkpoint` state file
def get_checkpoint_state(checkpoint_dir):
    """Parses <dir>/checkpoint -> (model_checkpoint_path, [all_model_checkpoint_paths])
    with relative paths resolved against the directory; None if absent."""
    path = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(path):
        return None
    latest, all_paths = None, []
    with open(path) as f:
        for line in f:
            m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"', line)
            if not m:
                continue
            p = m.group(2)
            if not os.path.isabs(p):
                p = os.path.join(checkpoint_dir, p)
            if m.group(1) == 'model_checkpoint_path':
                latest = p
            else:
                all_paths.append(p)
    if latest is None:
        return None
    return latest, all_paths or [latest]


def latest_checkpoint(checkpoint_dir):
    state = get_checkpoint_state(checkpoint_dir) if checkpoint_dir else None
    return state[0] if state else None


def write_checkpoint_state(checkpoint_dir, prefixes):
    with open(os.path.join(checkpoint_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(prefixes[-1]))
        for p in prefixes:
            f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
