"""Minimal Avro Object Container File reader/writer (fastavro is not installable here), written from
the Avro 1.x specification, for the two record types the random-effect stage reads and writes:

  * BayesianLinearModelAvro — photon-ml model file (gdmix-trainer/src/gdmix/models/schemas.py:3-51),
    written by export_linear_model_to_avro (util/io_utils.py:163-212) and read back for warm start
    (models/custom/random_effect_lr_lbfgs_model.py:256-309);
  * the inference result record (util/io_utils.py:367-375).

Container: magic 'Obj\\x01' | file metadata map {avro.schema, avro.codec} | 16-byte sync | blocks
[long count][long byte_size][payload][sync]. Binary encoding: int/long zig-zag varint; float/double IEEE
little endian; string/bytes = long length + data; array/map = blocks of [long count][items] ending with 0;
union = long branch index + value; record = fields in schema order. Codecs: null and deflate.
"""
import json
import os
import struct
import zlib

MAGIC = b"Obj\x01"

BAYESIAN_LINEAR_MODEL_SCHEMA = {
    "type": "record", "name": "BayesianLinearModelAvro", "namespace": "com.linkedin.photon.avro.generated",
    "doc": "a generic schema to describe a Bayesian linear model with means and variances",
    "fields": [
        {"name": "modelId", "type": "string"},
        {"name": "modelClass", "type": ["null", "string"],
         "doc": "The fully-qualified class name of enclosing GLM model class. E.g.: "
                "com.linkedin.photon.ml.supervised.classification.LogisticRegressionModel", "default": None},
        {"name": "means", "type": {"type": "array", "items": {
            "type": "record", "name": "NameTermValueAvro",
            "doc": "A tuple of name, term and value. Used as feature or model coefficient",
            "fields": [{"name": "name", "type": "string"}, {"name": "term", "type": "string"},
                       {"name": "value", "type": "double"}]}}},
        {"name": "variances", "type": ["null", {"type": "array", "items": "NameTermValueAvro"}], "default": None},
        {"name": "lossFunction", "type": ["null", "string"],
         "doc": "The loss function used for training as the class name. E.g.: "
                "com.linkedin.photon.ml.function.LogisticLossFunction", "default": None},
    ]}


def inference_output_schema(schema_params, has_weight, has_logits_per_coordinate=True):
    """get_inference_output_avro_schema (util/io_utils.py:367-375)."""
    fields = [{"name": schema_params.uid_column_name, "type": "long"},
              {"name": schema_params.prediction_score_column_name, "type": "float"},
              {"name": schema_params.label_column_name, "type": ["null", "float"], "default": None}]
    if has_weight:
        fields.append({"name": schema_params.weight_column_name, "type": "float"})
    if has_logits_per_coordinate:
        fields.append({"name": schema_params.prediction_score_per_coordinate_column_name, "type": "float"})
    return {"name": "validation_result", "type": "record", "fields": fields}


# ---- primitive codecs ----------------------------------------------------------------------------------
def _zz(n):
    return (n << 1) ^ (n >> 63)


def enc_long(n):
    u = _zz(int(n)) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = u & 0x7F
        u >>= 7
        if u:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def dec_long(buf, pos):
    u, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        u |= (b & 0x7F) << shift
        if not b & 0x80:
            break
        shift += 7
    return (u >> 1) ^ -(u & 1), pos


def enc_string(s):
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    return enc_long(len(b)) + b


# ---- schema driven codec ------------------------------------------------------------------------------
class Codec:
    """Encoder/decoder for one (parsed JSON) schema; named types are resolved on first sight."""

    def __init__(self, schema):
        self.schema = schema
        self.named = {}
        self._collect(schema)

    def _collect(self, s):
        if isinstance(s, dict):
            t = s.get("type")
            if t in ("record", "enum", "fixed") and "name" in s:
                self.named[s["name"]] = s
                if "namespace" in s:
                    self.named[s["namespace"] + "." + s["name"]] = s
            if t == "record":
                for f in s["fields"]:
                    self._collect(f["type"])
            elif t == "array":
                self._collect(s["items"])
            elif t == "map":
                self._collect(s["values"])
            elif isinstance(t, (dict, list)):
                self._collect(t)
        elif isinstance(s, list):
            for b in s:
                self._collect(b)

    def _resolve(self, s):
        if isinstance(s, str) and s in self.named:
            return self.named[s]
        return s

    # -- encode
    def encode(self, datum, s=None, out=None):
        top = out is None
        if top:
            out = bytearray()
        s = self._resolve(self.schema if s is None else s)
        if isinstance(s, list):
            idx = self._union_branch(s, datum)
            out += enc_long(idx)
            self.encode(datum, s[idx], out)
        elif isinstance(s, dict):
            t = s["type"]
            if t == "record":
                for f in s["fields"]:
                    if f["name"] in datum:
                        v = datum[f["name"]]
                    elif "default" in f:
                        v = f["default"]
                    else:
                        raise ValueError(f"record {s.get('name')}: field {f['name']!r} missing")
                    self.encode(v, f["type"], out)
            elif t == "array":
                items = list(datum)
                if items:
                    out += enc_long(len(items))
                    for it in items:
                        self.encode(it, s["items"], out)
                out += enc_long(0)
            elif t == "map":
                if datum:
                    out += enc_long(len(datum))
                    for k, v in datum.items():
                        out += enc_string(k)
                        self.encode(v, s["values"], out)
                out += enc_long(0)
            else:
                self.encode(datum, t, out)
        else:
            if s == "null":
                pass
            elif s == "boolean":
                out.append(1 if datum else 0)
            elif s in ("int", "long"):
                out += enc_long(datum)
            elif s == "float":
                out += struct.pack("<f", float(datum))
            elif s == "double":
                out += struct.pack("<d", float(datum))
            elif s in ("string", "bytes"):
                out += enc_string(datum)
            else:
                raise ValueError(f"unsupported avro type {s!r}")
        return bytes(out) if top else None

    def _union_branch(self, branches, datum):
        for i, b in enumerate(branches):
            b = self._resolve(b)
            name = b if isinstance(b, str) else b.get("type")
            if datum is None and name == "null":
                return i
            if datum is None:
                continue
            if name == "null":
                continue
            if name in ("string", "bytes") and isinstance(datum, (str, bytes)):
                return i
            if name in ("float", "double") and isinstance(datum, (int, float)) and not isinstance(datum, bool):
                return i
            if name in ("int", "long") and isinstance(datum, int) and not isinstance(datum, bool):
                return i
            if name == "boolean" and isinstance(datum, bool):
                return i
            if name == "array" and isinstance(datum, (list, tuple)):
                return i
            if name in ("record", "map") and isinstance(datum, dict):
                return i
        # numpy scalars and friends: fall back to the first non-null branch
        for i, b in enumerate(branches):
            if self._resolve(b) != "null" and datum is not None:
                return i
        raise ValueError(f"no union branch for {datum!r}")

    # -- decode
    def decode(self, buf, pos=0, s=None):
        s = self._resolve(self.schema if s is None else s)
        if isinstance(s, list):
            idx, pos = dec_long(buf, pos)
            return self.decode(buf, pos, s[idx])
        if isinstance(s, dict):
            t = s["type"]
            if t == "record":
                rec = {}
                for f in s["fields"]:
                    rec[f["name"]], pos = self.decode(buf, pos, f["type"])
                return rec, pos
            if t == "array":
                items = []
                while True:
                    cnt, pos = dec_long(buf, pos)
                    if cnt == 0:
                        break
                    if cnt < 0:
                        cnt = -cnt
                        _, pos = dec_long(buf, pos)   # block byte size
                    for _ in range(cnt):
                        v, pos = self.decode(buf, pos, s["items"])
                        items.append(v)
                return items, pos
            if t == "map":
                m = {}
                while True:
                    cnt, pos = dec_long(buf, pos)
                    if cnt == 0:
                        break
                    if cnt < 0:
                        cnt = -cnt
                        _, pos = dec_long(buf, pos)
                    for _ in range(cnt):
                        ln, pos = dec_long(buf, pos)
                        k = bytes(buf[pos:pos + ln]).decode("utf-8")
                        pos += ln
                        m[k], pos = self.decode(buf, pos, s["values"])
                return m, pos
            return self.decode(buf, pos, t)
        if s == "null":
            return None, pos
        if s == "boolean":
            return bool(buf[pos]), pos + 1
        if s in ("int", "long"):
            return dec_long(buf, pos)
        if s == "float":
            return struct.unpack_from("<f", buf, pos)[0], pos + 4
        if s == "double":
            return struct.unpack_from("<d", buf, pos)[0], pos + 8
        if s == "string":
            ln, pos = dec_long(buf, pos)
            return bytes(buf[pos:pos + ln]).decode("utf-8"), pos + ln
        if s == "bytes":
            ln, pos = dec_long(buf, pos)
            return bytes(buf[pos:pos + ln]), pos + ln
        raise ValueError(f"unsupported avro type {s!r}")


# ---- container files ----------------------------------------------------------------------------------
_META_CODEC = Codec({"type": "map", "values": "bytes"})


class Writer:
    """Object container file writer; records are buffered into blocks of `block_records` (the reference
    writes 1024-record blocks, util/io_utils.py:299)."""

    def __init__(self, path, schema, codec="null", block_records=1024, sync_marker=None):
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        self.f = open(path, "wb")
        self.codec_name = codec
        self.codec = Codec(schema)
        self.block_records = block_records
        self.sync = sync_marker or os.urandom(16)
        self.buf = bytearray()
        self.count = 0
        self.total = 0
        meta = {"avro.schema": json.dumps(schema).encode("utf-8"), "avro.codec": codec.encode("ascii")}
        self.f.write(MAGIC + _META_CODEC.encode(meta) + self.sync)

    def write(self, record):
        self.codec.encode(record, None, self.buf)
        self.count += 1
        self.total += 1
        if self.count >= self.block_records:
            self.flush()

    def write_encoded(self, payload: bytes, count: int):
        """Append `count` records already encoded with this schema (vectorised encoders)."""
        self.flush()
        self._emit(payload, count)
        self.total += count

    def _emit(self, payload, count):
        if self.codec_name == "deflate":
            payload = zlib.compress(bytes(payload))[2:-4]   # raw deflate, as the Avro spec requires
        self.f.write(enc_long(count) + enc_long(len(payload)) + bytes(payload) + self.sync)

    def flush(self):
        if self.count:
            self._emit(self.buf, self.count)
            self.buf = bytearray()
            self.count = 0

    def close(self):
        self.flush()
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def container_header(schema, codec="null", sync_marker=None):
    """(header bytes, sync marker) of an object container file: what Writer.__init__ emits."""
    sync = sync_marker or os.urandom(16)
    meta = {"avro.schema": json.dumps(schema).encode("utf-8"), "avro.codec": codec.encode("ascii")}
    return MAGIC + _META_CODEC.encode(meta) + sync, sync


def write_file(path, schema, records, codec="null", block_records=1024):
    with Writer(path, schema, codec, block_records) as w:
        for r in records:
            w.write(r)
    return w.total


def read_file(path):
    """Yield the records of an object container file (schema taken from the file)."""
    with open(path, "rb") as f:
        data = f.read()
    buf = memoryview(data)
    if bytes(buf[:4]) != MAGIC:
        raise ValueError(f"{path}: not an Avro object container file")
    meta, pos = _META_CODEC.decode(buf, 4)
    schema = json.loads(meta["avro.schema"].decode("utf-8"))
    codec_name = meta.get("avro.codec", b"null").decode("ascii")
    codec = Codec(schema)
    sync = bytes(buf[pos:pos + 16])
    pos += 16
    n = len(buf)
    while pos < n:
        count, pos = dec_long(buf, pos)
        size, pos = dec_long(buf, pos)
        payload = buf[pos:pos + size]
        pos += size
        if bytes(buf[pos:pos + 16]) != sync:
            raise ValueError(f"{path}: sync marker mismatch")
        pos += 16
        if codec_name == "deflate":
            payload = memoryview(zlib.decompress(bytes(payload), -15))
        elif codec_name != "null":
            raise ValueError(f"{path}: unsupported codec {codec_name}")
        p = 0
        for _ in range(count):
            rec, p = codec.decode(payload, p)
            yield rec


def read_header(path, limit=1 << 22):
    """(schema, codec name, sync marker, offset of the first block) of an object container file."""
    with open(path, "rb") as f:
        data = f.read(limit)
    if data[:4] != MAGIC:
        raise ValueError(f"{path}: not an Avro object container file")
    meta, pos = _META_CODEC.decode(memoryview(data), 4)
    if pos + 16 > len(data):
        raise ValueError(f"{path}: truncated header")
    return (json.loads(meta["avro.schema"].decode("utf-8")), meta.get("avro.codec", b"null").decode("ascii"),
            bytes(data[pos:pos + 16]), pos + 16)


def is_model_schema(schema):
    """True when a writer schema has exactly the field order and types of BAYESIAN_LINEAR_MODEL_SCHEMA (docs, defaults
    and namespaces aside): the layout the native model reader walks."""
    def ntv(t):
        return (isinstance(t, dict) and t.get("type") == "record" and
                [(f.get("name"), f.get("type")) for f in t.get("fields", [])] == [("name", "string"), ("term", "string"), ("value", "double")])
    try:
        if schema.get("type") != "record":
            return False
        f = schema["fields"]
        if [x["name"] for x in f] != ["modelId", "modelClass", "means", "variances", "lossFunction"]:
            return False
        if f[0]["type"] != "string" or f[1]["type"] != ["null", "string"] or f[4]["type"] != ["null", "string"]:
            return False
        m = f[2]["type"]
        if not (isinstance(m, dict) and m.get("type") == "array" and ntv(m["items"])):
            return False
        name = m["items"]["name"]
        full = (m["items"].get("namespace") or schema.get("namespace") or "")
        v = f[3]["type"]
        if not (isinstance(v, list) and len(v) == 2 and v[0] == "null" and isinstance(v[1], dict) and v[1].get("type") == "array"):
            return False
        it = v[1]["items"]
        return it == name or it == (full + "." + name if full else name) or ntv(it)
    except (KeyError, TypeError, AttributeError, IndexError):
        return False


def read_schema(path):
    with open(path, "rb") as f:
        data = f.read(1 << 16)
    meta, _ = _META_CODEC.decode(memoryview(data), 4)
    return json.loads(meta["avro.schema"].decode("utf-8"))
