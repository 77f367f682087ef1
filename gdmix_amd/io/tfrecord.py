"""TFRecord framing and tf.train.{SequenceExample, Example} wire codec, written from the format specs
(no TensorFlow here): what per_entity_grouped_input_fn reads
(gdmix-trainer/src/gdmix/io/input_data_pipeline.py:223-332).

Framing per record:  uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
Compression: whole-file gzip (".gz") or zlib (".deflate") around the same framing (:63-85).

Protobuf (tensorflow/core/example/{example,feature}.proto):
  SequenceExample { Features context = 1; FeatureLists feature_lists = 2; }
  Example         { Features features = 1; }
  Features        { map<string, Feature> feature = 1; }        map entry: key = 1, value = 2
  FeatureLists    { map<string, FeatureList> feature_list = 1; }
  FeatureList     { repeated Feature feature = 1; }
  Feature         { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
  *List           { repeated value = 1; }   packed or unpacked
"""
import gzip
import struct
import zlib

import numpy as np

# ---- CRC-32C (Castagnoli), table driven --------------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        tab = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            tab[i] = c
        _CRC_TABLE = [int(x) for x in tab]
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    tab = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- framing -----------------------------------------------------------------------------------------
def compression_of(filename: str) -> str:
    """'' | 'ZLIB' | 'GZIP' from the file suffix (input_data_pipeline.py:63-85)."""
    suffix = filename.split(".")[-1]
    if suffix == "deflate":
        return "ZLIB"
    if suffix == "gz":
        return "GZIP"
    return ""


def read_file_bytes(filename: str) -> bytes:
    with open(filename, "rb") as f:
        raw = f.read()
    comp = compression_of(filename)
    if comp == "GZIP":
        return gzip.decompress(raw)
    if comp == "ZLIB":
        return zlib.decompress(raw)
    return raw


def iter_records(filename: str, check_crc: bool = False):
    """Yield the payload (memoryview) of every record of a TFRecord file."""
    buf = memoryview(read_file_bytes(filename))
    pos, n = 0, len(buf)
    while pos < n:
        if pos + 12 > n:
            raise ValueError(f"{filename}: truncated record header at byte {pos}")
        (length,) = struct.unpack_from("<Q", buf, pos)
        if check_crc:
            (lcrc,) = struct.unpack_from("<I", buf, pos + 8)
            if lcrc != masked_crc32c(bytes(buf[pos:pos + 8])):
                raise ValueError(f"{filename}: corrupt length CRC at byte {pos}")
        start = pos + 12
        end = start + length
        if end + 4 > n:
            raise ValueError(f"{filename}: truncated record at byte {pos}")
        if check_crc:
            (dcrc,) = struct.unpack_from("<I", buf, end)
            if dcrc != masked_crc32c(bytes(buf[start:end])):
                raise ValueError(f"{filename}: corrupt data CRC at byte {pos}")
        yield buf[start:end]
        pos = end + 4


def write_records(filename: str, payloads) -> None:
    out = bytearray()
    for p in payloads:
        p = bytes(p)
        hdr = struct.pack("<Q", len(p))
        out += hdr + struct.pack("<I", masked_crc32c(hdr)) + p + struct.pack("<I", masked_crc32c(p))
    comp = compression_of(filename)
    data = bytes(out)
    if comp == "GZIP":
        data = gzip.compress(data)
    elif comp == "ZLIB":
        data = zlib.compress(data)
    with open(filename, "wb") as f:
        f.write(data)


# ---- protobuf wire decoding --------------------------------------------------------------------------
def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) of one message; value is int (wt 0/1/5) or memoryview (wt 2)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _signed64(u):
    return u - (1 << 64) if u >= (1 << 63) else u


def _decode_feature(buf):
    """Feature -> ('bytes', [bytes...]) | ('float', float32 array) | ('int64', int64 array)."""
    for fn, wt, v in _fields(buf):
        if wt != 2:
            continue
        if fn == 1:
            return "bytes", [bytes(x) for f2, w2, x in _fields(v) if f2 == 1 and w2 == 2]
        if fn == 2:
            chunks = []
            for f2, w2, x in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 2:   # packed
                    chunks.append(np.frombuffer(x, dtype="<f4"))
                elif w2 == 5:
                    chunks.append(np.frombuffer(x, dtype="<f4"))
            return "float", (np.concatenate(chunks) if chunks else np.zeros(0, np.float32)).astype(np.float32)
        if fn == 3:
            vals = []
            for f2, w2, x in _fields(v):
                if f2 != 1:
                    continue
                if w2 == 2:   # packed varints
                    p, n = 0, len(x)
                    while p < n:
                        u, p = _varint(x, p)
                        vals.append(_signed64(u))
                elif w2 == 0:
                    vals.append(_signed64(x))
            return "int64", np.array(vals, np.int64)
    return "empty", np.zeros(0, np.float32)


def _decode_features_map(buf):
    out = {}
    for fn, wt, entry in _fields(buf):
        if fn != 1 or wt != 2:
            continue
        key, val = None, None
        for f2, w2, x in _fields(entry):
            if f2 == 1 and w2 == 2:
                key = bytes(x).decode("utf-8")
            elif f2 == 2 and w2 == 2:
                val = x
        if key is not None:
            out[key] = _decode_feature(val) if val is not None else ("empty", np.zeros(0, np.float32))
    return out


def decode_sequence_example(buf):
    """-> (context: {name: (kind, values)}, feature_lists: {name: [(kind, values) per step]})."""
    context, flists = {}, {}
    for fn, wt, v in _fields(buf):
        if wt != 2:
            continue
        if fn == 1:
            context = _decode_features_map(v)
        elif fn == 2:
            for f2, w2, entry in _fields(v):
                if f2 != 1 or w2 != 2:
                    continue
                key, steps = None, []
                for f3, w3, x in _fields(entry):
                    if f3 == 1 and w3 == 2:
                        key = bytes(x).decode("utf-8")
                    elif f3 == 2 and w3 == 2:
                        steps = [_decode_feature(y) for f4, w4, y in _fields(x) if f4 == 1 and w4 == 2]
                if key is not None:
                    flists[key] = steps
    return context, flists


def decode_example(buf):
    for fn, wt, v in _fields(buf):
        if fn == 1 and wt == 2:
            return _decode_features_map(v)
    return {}


# ---- protobuf wire encoding (test data and synthetic partitions) ----------------------------------------
def _enc_varint(u):
    out = bytearray()
    u &= (1 << 64) - 1
    while True:
        b = u & 0x7F
        u >>= 7
        if u:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def encode_feature(kind, values):
    if kind == "bytes":
        inner = b"".join(_ld(1, v if isinstance(v, bytes) else str(v).encode("utf-8")) for v in values)
        return _ld(1, inner)
    if kind == "float":
        arr = np.asarray(values, dtype="<f4")
        return _ld(2, _ld(1, arr.tobytes()) if arr.size else b"")
    if kind == "int64":
        packed = b"".join(_enc_varint(int(v)) for v in values)
        return _ld(3, _ld(1, packed) if len(values) else b"")
    raise ValueError(kind)


def _enc_features_map(d):
    out = bytearray()
    for k, (kind, vals) in d.items():
        out += _ld(1, _ld(1, k.encode("utf-8")) + _ld(2, encode_feature(kind, vals)))
    return bytes(out)


def encode_sequence_example(context, feature_lists):
    """context: {name: (kind, values)}, feature_lists: {name: [(kind, values), ...]} -> bytes."""
    fl = bytearray()
    for k, steps in feature_lists.items():
        body = b"".join(_ld(1, encode_feature(kind, vals)) for kind, vals in steps)
        fl += _ld(1, _ld(1, k.encode("utf-8")) + _ld(2, body))
    return _ld(1, _enc_features_map(context)) + _ld(2, bytes(fl))


def encode_example(features):
    return _ld(1, _enc_features_map(features))
