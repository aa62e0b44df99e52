"""TFRecord framing and tf.Example decoding without TensorFlow.

A TFRecord file is a sequence of
    uint64 length | uint32 masked_crc32c(length) | data[length] | uint32 masked_crc32c(data)
(little endian).  A tf.Example is the protobuf
    Example{1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}}.
Restated from the published formats; the reference reads these files through
tf.data.TFRecordDataset + slim's TFExampleDecoder (builders/dataset_builder.py:18-62)
and writes them in dataset_tools/create_cityscapes_tfrecord.py:62-92.
"""
import struct

from fss.checkpoint import _parse_fields, _read_varint, _varint, masked_crc32c


def read_records(path, verify=False):
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError('%s: truncated record header' % path)
            length, = struct.unpack('<Q', head[:8])
            if verify and struct.unpack('<I', head[8:])[0] != masked_crc32c(head[:8]):
                raise ValueError('%s: corrupted record length' % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError('%s: truncated record' % path)
            if verify and struct.unpack('<I', tail)[0] != masked_crc32c(data):
                raise ValueError('%s: corrupted record data' % path)
            yield data


def write_records(path, records):
    with open(path, 'wb') as f:
        for data in records:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', masked_crc32c(head)))
            f.write(data + struct.pack('<I', masked_crc32c(data)))


def parse_example(data):
    """serialized tf.Example -> {key: list of bytes | float | int}."""
    out = {}
    for f1, _, features in _parse_fields(data):
        if f1 != 1:
            continue
        for f2, _, entry in _parse_fields(features):
            if f2 != 1:
                continue
            key, feature = None, b''
            for f3, _, v in _parse_fields(entry):
                if f3 == 1:
                    key = bytes(v).decode('utf-8')
                elif f3 == 2:
                    feature = v
            values = []
            for kind, _, lst in _parse_fields(feature):
                if kind == 1:                                  # BytesList
                    values = [bytes(v) for f, _, v in _parse_fields(lst) if f == 1]
                elif kind == 2:                                # FloatList (packed or not)
                    for f, wt, v in _parse_fields(lst):
                        if f == 1 and wt == 2:
                            values += list(struct.unpack('<%df' % (len(v) // 4), v))
                        elif f == 1:
                            values.append(struct.unpack('<f', v)[0])
                elif kind == 3:                                # Int64List (packed or not)
                    for f, wt, v in _parse_fields(lst):
                        if f == 1 and wt == 2:
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                values.append(x - (1 << 64) if x >= (1 << 63) else x)
                        elif f == 1:
                            values.append(v - (1 << 64) if v >= (1 << 63) else v)
            out[key] = values
    return out


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def make_example(features):
    """{key: bytes | int | [..]} -> serialized tf.Example (bytes / int64 features)."""
    entries = b''
    for key in sorted(features):
        vals = features[key]
        if not isinstance(vals, (list, tuple)):
            vals = [vals]
        if all(isinstance(v, (bytes, bytearray)) for v in vals):
            feature = _ld(1, b''.join(_ld(1, bytes(v)) for v in vals))
        else:
            packed = b''.join(_varint(int(v) & ((1 << 64) - 1)) for v in vals)
            feature = _ld(3, _ld(1, packed))
        entries += _ld(1, _ld(1, key.encode('utf-8')) + _ld(2, feature))
    return _ld(1, entries)
