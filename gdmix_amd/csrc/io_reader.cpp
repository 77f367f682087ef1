// io_reader.cpp — libgdmix_io.so: multi-threaded reader of entity-grouped TFRecord partitions
// (include/gdmix_io.h). Written from the format specifications: TFRecord framing with
// masked CRC-32C, whole-file gzip / zlib, tf.train.SequenceExample protobuf wire format. The record
// semantics follow what the reference does between the file and the solver:
//   per_entity_grouped_input_fn   gdmix-trainer/src/gdmix/io/input_data_pipeline.py:223-332
//   prepare_jobs                  gdmix-trainer/src/gdmix/models/custom/scipy/job_consumers.py:209-258
// and are the same, error for error, as gdmix_amd/io/grouped_reader.py (the Python statement of the same
// rules, kept for hosts without a C++ toolchain); tests/test_native_io.py compares the two array for array.
//
// Two parallel passes over the framed records: (1) locate the columns of every record and size them,
// (2) after a prefix sum, decode straight into the final arrays. No intermediate per-record objects.
#include "../../include/gdmix_io.h"

#include <fcntl.h>
#include <nmmintrin.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace

namespace gdmix_io_detail {
// shared with io_avro.cpp: same thread-local message buffer
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
// threads <= 0 in the API: the host's cores, at most 32 (one trainer process per GPU shares the host with seven
// others, and the decode phases stop scaling before that: tools/io_bench.py); GDMIX_IO_THREADS overrides.
int default_threads() {
  if (const char* e = getenv("GDMIX_IO_THREADS")) {
    const int t = atoi(e);
    if (t > 0) return t;
  }
  int t = (int)std::thread::hardware_concurrency();
  if (t <= 0) t = 1;
  return t < 32 ? t : 32;
}
}  // namespace gdmix_io_detail

namespace {

// ---- CRC-32C ---------------------------------------------------------------------------------------
uint32_t crc_table[256];
std::atomic<int> crc_table_ready{0};

void crc_init() {
  if (crc_table_ready.load(std::memory_order_acquire)) return;
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    crc_table[i] = c;
  }
  crc_table_ready.store(1, std::memory_order_release);
}

__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const uint8_t* p, size_t n) {
  uint64_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c = _mm_crc32_u64(c, v);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = _mm_crc32_u8(c32, *p++);
  return c32 ^ 0xFFFFFFFFu;
}

uint32_t crc32c_sw(const uint8_t* p, size_t n) {
  crc_init();
  uint32_t c = 0xFFFFFFFFu;
  while (n--) c = crc_table[(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

uint32_t crc32c(const uint8_t* p, size_t n) {
  static const bool hw = __builtin_cpu_supports("sse4.2");
  return hw ? crc32c_hw(p, n) : crc32c_sw(p, n);
}

uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// ---- protobuf wire -----------------------------------------------------------------------------------
struct Span {
  const uint8_t* p = nullptr;
  const uint8_t* e = nullptr;
  size_t size() const { return (size_t)(e - p); }
  bool empty() const { return p == e; }
};

inline bool get_varint(Span& s, uint64_t& v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 70; shift += 7) {
    if (s.p >= s.e) return false;
    const uint8_t b = *s.p++;
    if (shift < 64) r |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) { v = r; return true; }
  }
  return false;
}

// One field of a message: number, wire type, payload (wt 2) or value (wt 0) / raw bytes (wt 1, 5).
struct Field {
  uint32_t fn, wt;
  Span payload;
  uint64_t value;
};

inline bool next_field(Span& s, Field& f) {
  uint64_t key;
  if (!get_varint(s, key)) return false;
  f.fn = (uint32_t)(key >> 3);
  f.wt = (uint32_t)(key & 7);
  f.value = 0;
  f.payload = Span();
  switch (f.wt) {
    case 0: return get_varint(s, f.value);
    case 1:
      if (s.size() < 8) return false;
      f.payload.p = s.p; f.payload.e = s.p + 8; s.p += 8;
      return true;
    case 2: {
      uint64_t len;
      if (!get_varint(s, len) || len > s.size()) return false;
      f.payload.p = s.p; f.payload.e = s.p + len; s.p += len;
      return true;
    }
    case 5:
      if (s.size() < 4) return false;
      f.payload.p = s.p; f.payload.e = s.p + 4; s.p += 4;
      return true;
    default: return false;   // groups are not used by these messages
  }
}

enum Kind { K_ABSENT = 0, K_EMPTY, K_BYTES, K_FLOAT, K_INT64 };

// tf.train.Feature -> kind + the *List message (first length-delimited field 1/2/3 wins, as a oneof reader sees)
bool feature_kind(Span feat, Kind& kind, Span& list) {
  kind = K_EMPTY;
  list = Span();
  Field f;
  while (!feat.empty()) {
    if (!next_field(feat, f)) return false;
    if (f.wt != 2) continue;
    if (f.fn == 1) { kind = K_BYTES; list = f.payload; return true; }
    if (f.fn == 2) { kind = K_FLOAT; list = f.payload; return true; }
    if (f.fn == 3) { kind = K_INT64; list = f.payload; return true; }
  }
  return true;
}

// number of values of a FloatList / Int64List message (packed or one field per element)
bool count_floats(Span list, int64_t& n) {
  n = 0;
  Field f;
  while (!list.empty()) {
    if (!next_field(list, f)) return false;
    if (f.fn != 1) continue;
    if (f.wt == 2) { if (f.payload.size() % 4) return false; n += (int64_t)(f.payload.size() / 4); }
    else if (f.wt == 5) n += 1;
  }
  return true;
}

bool count_int64s(Span list, int64_t& n) {
  n = 0;
  Field f;
  while (!list.empty()) {
    if (!next_field(list, f)) return false;
    if (f.fn != 1) continue;
    if (f.wt == 2) {
      for (const uint8_t* q = f.payload.p; q < f.payload.e; ++q) n += !(*q & 0x80);
      if (f.payload.size() && (f.payload.e[-1] & 0x80)) return false;   // truncated varint
    } else if (f.wt == 0) n += 1;
  }
  return true;
}

template <class T>
bool read_floats(Span list, T* out, int64_t n) {   // T = float
  int64_t k = 0;
  Field f;
  while (!list.empty()) {
    if (!next_field(list, f)) return false;
    if (f.fn != 1) continue;
    if (f.wt == 2) {
      const int64_t c = (int64_t)(f.payload.size() / 4);
      if (k + c > n) return false;
      memcpy(out + k, f.payload.p, (size_t)c * 4);
      k += c;
    } else if (f.wt == 5) {
      if (k + 1 > n) return false;
      memcpy(out + k, f.payload.p, 4);
      k += 1;
    }
  }
  return k == n;
}

template <class Fn>
bool each_int64(Span list, Fn&& fn) {
  Field f;
  while (!list.empty()) {
    if (!next_field(list, f)) return false;
    if (f.fn != 1) continue;
    if (f.wt == 2) {
      Span s = f.payload;
      while (!s.empty()) {
        uint64_t v;
        if (!get_varint(s, v)) return false;
        fn((int64_t)v);
      }
    } else if (f.wt == 0) {
      fn((int64_t)f.value);
    }
  }
  return true;
}

bool read_int64s(Span list, int64_t* out, int64_t n) {
  int64_t k = 0;
  bool over = false;
  if (!each_int64(list, [&](int64_t v) { if (k < n) out[k] = v; else over = true; ++k; })) return false;
  return !over && k == n;
}

bool key_is(const Span& k, const char* name, size_t len) { return name && k.size() == len && memcmp(k.p, name, len) == 0; }

// ---- records ---------------------------------------------------------------------------------------
struct Names {
  const char* entity; size_t entity_len;
  const char* uid; size_t uid_len;
  const char* offset; size_t offset_len;
  const char* label; size_t label_len;
  const char* weight; size_t weight_len;
  std::string bag_idx, bag_val;
  bool has_bag;
};

struct RecInfo {
  Span rec;
  int file;
  // located columns (Feature messages) / feature lists (FeatureList messages)
  Span f_entity, f_uid, f_offset, f_label, f_weight, fl_idx, fl_val;
  bool has_label_col;
  int64_t n, nnz, id_len;
};

struct Ctx {
  const gdmix_io_schema* sc;
  Names nm;
  std::vector<std::string> files;
  mutable std::atomic<int> labels_nonbinary{0};   // set by decode() when a label is neither 0 nor 1
};

int rec_error(const Ctx& c, const RecInfo& r, int code, const char* what) {
  return fail(code, "%s: record at byte offset of length %zu: %s", c.files[r.file].c_str(), r.rec.size(), what);
}

// entity id of a located entity column as the reference renders it
int entity_id(const Ctx& c, const RecInfo& r, std::string& out) {
  Kind kind;
  Span list;
  if (!feature_kind(r.f_entity, kind, list)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad entity feature");
  out.clear();
  if (kind == K_BYTES) {
    int cnt = 0;
    Field f;
    while (!list.empty()) {
      if (!next_field(list, f)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad bytes list");
      if (f.fn == 1 && f.wt == 2) { if (cnt++ == 0) out.assign((const char*)f.payload.p, f.payload.size()); }
    }
    if (cnt != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "entity column must be a scalar");
    return GDMIX_IO_OK;
  }
  if (kind == K_INT64) {
    int64_t cnt;
    if (!count_int64s(list, cnt)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad int64 list");
    if (cnt != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "entity column must be a scalar");
    int64_t v = 0;
    if (!read_int64s(list, &v, 1)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad int64 list");
    out = std::to_string((long long)v);
    return GDMIX_IO_OK;
  }
  if (kind == K_FLOAT) {
    int64_t cnt;
    if (!count_floats(list, cnt)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad float list");
    if (cnt != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "entity column must be a scalar");
    float v = 0;
    if (!read_floats(list, &v, 1)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad float list");
    out = std::to_string((long long)v);   // str(int(x))
    return GDMIX_IO_OK;
  }
  return rec_error(c, r, GDMIX_IO_ESCHEMA, "entity column must be a scalar");
}

// length of a dense per-sample column of the given kinds (K_EMPTY counts as length 0)
int column_len(const Ctx& c, const RecInfo& r, Span feat, bool allow_int, bool allow_float, const char* what, int64_t& n) {
  Kind kind;
  Span list;
  if (!feature_kind(feat, kind, list)) return rec_error(c, r, GDMIX_IO_EFORMAT, what);
  n = 0;
  if (kind == K_EMPTY) return GDMIX_IO_OK;
  if (kind == K_INT64 && allow_int) return count_int64s(list, n) ? GDMIX_IO_OK : rec_error(c, r, GDMIX_IO_EFORMAT, what);
  if (kind == K_FLOAT && allow_float) return count_floats(list, n) ? GDMIX_IO_OK : rec_error(c, r, GDMIX_IO_EFORMAT, what);
  return rec_error(c, r, GDMIX_IO_ESCHEMA, what);
}

// Pass 1: find the columns of one record, validate their lengths, size its outputs.
int locate(const Ctx& c, RecInfo& r) {
  const Names& nm = c.nm;
  Span msg = r.rec;
  Span context, flists;
  Field f;
  while (!msg.empty()) {
    if (!next_field(msg, f)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad SequenceExample");
    if (f.wt != 2) continue;
    if (f.fn == 1) context = f.payload;
    else if (f.fn == 2) flists = f.payload;
  }
  bool have_entity = false, have_uid = false, have_offset = false, have_weight = false;
  r.has_label_col = false;
  while (!context.empty()) {
    if (!next_field(context, f)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad context");
    if (f.fn != 1 || f.wt != 2) continue;
    Span entry = f.payload, key, val;
    bool has_key = false;
    Field g;
    while (!entry.empty()) {
      if (!next_field(entry, g)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad context entry");
      if (g.wt != 2) continue;
      if (g.fn == 1) { key = g.payload; has_key = true; }
      else if (g.fn == 2) val = g.payload;
    }
    if (!has_key) continue;
    if (key_is(key, nm.entity, nm.entity_len)) { r.f_entity = val; have_entity = true; }
    else if (key_is(key, nm.uid, nm.uid_len)) { r.f_uid = val; have_uid = true; }
    else if (key_is(key, nm.offset, nm.offset_len)) { r.f_offset = val; have_offset = true; }
    else if (key_is(key, nm.label, nm.label_len)) { r.f_label = val; r.has_label_col = true; }
    else if (key_is(key, nm.weight, nm.weight_len)) { r.f_weight = val; have_weight = true; }
  }
  bool have_idx = false, have_val = false;
  if (nm.has_bag) {
    while (!flists.empty()) {
      if (!next_field(flists, f)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad feature_lists");
      if (f.fn != 1 || f.wt != 2) continue;
      Span entry = f.payload, key, val;
      bool has_key = false;
      Field g;
      while (!entry.empty()) {
        if (!next_field(entry, g)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad feature_lists entry");
        if (g.wt != 2) continue;
        if (g.fn == 1) { key = g.payload; has_key = true; }
        else if (g.fn == 2) val = g.payload;
      }
      if (!has_key) continue;
      if (key_is(key, nm.bag_idx.c_str(), nm.bag_idx.size())) { r.fl_idx = val; have_idx = true; }
      else if (key_is(key, nm.bag_val.c_str(), nm.bag_val.size())) { r.fl_val = val; have_val = true; }
    }
  }
  (void)have_idx; (void)have_val;   // a missing list is an empty list, as for the reference's dict.get(name, [])
  if (!have_entity) return rec_error(c, r, GDMIX_IO_ESCHEMA, "entity column is missing");
  if (!have_uid) return rec_error(c, r, GDMIX_IO_ESCHEMA, "uid column is missing");
  if (!have_offset) return rec_error(c, r, GDMIX_IO_ESCHEMA, "offset column is missing");
  if (nm.weight && !have_weight) return rec_error(c, r, GDMIX_IO_ESCHEMA, "weight column is missing");
  std::string id;
  int rc = entity_id(c, r, id);
  if (rc) return rc;
  r.id_len = (int64_t)id.size();
  int64_t n, m;
  if ((rc = column_len(c, r, r.f_uid, true, false, "uid column must be an int64 list", n))) return rc;
  if ((rc = column_len(c, r, r.f_offset, false, true, "offset column must be a float list", m))) return rc;
  if (m != n) return rec_error(c, r, GDMIX_IO_ESCHEMA, "offsets and uids differ in length");
  if (r.has_label_col) {
    if ((rc = column_len(c, r, r.f_label, true, true, "label column must be an int64 or float list", m))) return rc;
    if (m != n) return rec_error(c, r, GDMIX_IO_ESCHEMA, "labels and uids differ in length");
  }
  if (nm.weight) {
    if ((rc = column_len(c, r, r.f_weight, false, true, "weight column must be a float list", m))) return rc;
    if (m != n) return rec_error(c, r, GDMIX_IO_ESCHEMA, "weights and uids differ in length");
  }
  r.n = n;
  if (!nm.has_bag) { r.nnz = n; return GDMIX_IO_OK; }
  // steps of the two feature lists: same count, same length per step; the sample count the reference derives
  // (last step that owns a feature, job_consumers.py:229-232) must equal the uid count
  Span si = r.fl_idx, sv = r.fl_val;
  int64_t step = 0, last_nonempty = 0, nnz = 0;
  Field fi, fv;
  for (;;) {
    bool gi = false, gv = false;
    while (!si.empty()) {
      if (!next_field(si, fi)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad index feature list");
      if (fi.fn == 1 && fi.wt == 2) { gi = true; break; }
    }
    while (!sv.empty()) {
      if (!next_field(sv, fv)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad value feature list");
      if (fv.fn == 1 && fv.wt == 2) { gv = true; break; }
    }
    if (gi != gv) return rec_error(c, r, GDMIX_IO_ESCHEMA, "index and value feature lists differ in length");
    if (!gi) break;
    Kind ki, kv;
    Span li, lv;
    if (!feature_kind(fi.payload, ki, li) || !feature_kind(fv.payload, kv, lv))
      return rec_error(c, r, GDMIX_IO_EFORMAT, "bad feature list step");
    int64_t a = 0, b = 0;
    if (ki == K_INT64) { if (!count_int64s(li, a)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad int64 list"); }
    else if (ki != K_EMPTY) return rec_error(c, r, GDMIX_IO_ESCHEMA, "feature indices must be int64 lists");
    if (kv == K_FLOAT) { if (!count_floats(lv, b)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad float list"); }
    else if (kv != K_EMPTY) return rec_error(c, r, GDMIX_IO_ESCHEMA, "feature values must be float lists");
    if (a != b) return rec_error(c, r, GDMIX_IO_ESCHEMA, "indices and values of a sample differ in length");
    ++step;
    if (a > 0) last_nonempty = step;
    if (step <= n) nnz += a;
  }
  if (last_nonempty != n) return rec_error(c, r, GDMIX_IO_ESCHEMA, "feature rows (last non-empty) and uids differ in count");
  r.nnz = nnz;
  return GDMIX_IO_OK;
}

// Pass 2: decode one record into the batch arrays at its offsets.
int decode(const Ctx& c, const RecInfo& r, gdmix_io_batch* b, int64_t e, int64_t row0, int64_t nz0, int64_t id0,
           bool keep_label) {
  const Names& nm = c.nm;
  Kind kind;
  Span list;
  std::string id;
  int rc = entity_id(c, r, id);
  if (rc) return rc;
  memcpy(b->ent_id_bytes + id0, id.data(), id.size());
  const int64_t n = r.n;
  feature_kind(r.f_uid, kind, list);
  if (n && !read_int64s(list, b->uid + row0, n)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad uid list");
  feature_kind(r.f_offset, kind, list);
  if (n && !read_floats(list, b->offset + row0, n)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad offset list");
  if (keep_label && n) {
    feature_kind(r.f_label, kind, list);
    if (kind == K_FLOAT) {
      if (!read_floats(list, b->y + row0, n)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad label list");
    } else {
      int64_t k = 0;
      float* y = b->y + row0;
      if (!each_int64(list, [&](int64_t v) { if (k < n) y[k] = (float)v; ++k; }) || k != n)
        return rec_error(c, r, GDMIX_IO_EFORMAT, "bad label list");
    }
    bool binary = true;
    for (int64_t i = 0; i < n; ++i) binary &= (b->y[row0 + i] == 0.0f) | (b->y[row0 + i] == 1.0f);
    if (!binary) c.labels_nonbinary.store(1, std::memory_order_relaxed);
  } else {
    for (int64_t i = 0; i < n; ++i) b->y[row0 + i] = 0.0f;
  }
  if (nm.weight && n) {
    feature_kind(r.f_weight, kind, list);
    if (!read_floats(list, b->weight + row0, n)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad weight list");
  }
  int64_t* rp = b->row_nnz_ptr + row0;   // rp[i] = start of sample i; the closing entry is written by the next record / the caller
  if (!nm.has_bag) {
    for (int64_t i = 0; i < n; ++i) { rp[i] = nz0 + i; b->col_global[nz0 + i] = 0; b->val[nz0 + i] = 0.0f; }
    return GDMIX_IO_OK;
  }
  Span si = r.fl_idx, sv = r.fl_val;
  Field fi, fv;
  int64_t pos = nz0;
  for (int64_t i = 0; i < n; ++i) {
    bool gi = false, gv = false;
    while (!si.empty()) { if (!next_field(si, fi)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad index feature list"); if (fi.fn == 1 && fi.wt == 2) { gi = true; break; } }
    while (!sv.empty()) { if (!next_field(sv, fv)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad value feature list"); if (fv.fn == 1 && fv.wt == 2) { gv = true; break; } }
    if (!gi || !gv) return rec_error(c, r, GDMIX_IO_EFORMAT, "feature list ended early");
    Kind ki, kv;
    Span li, lv;
    feature_kind(fi.payload, ki, li);
    feature_kind(fv.payload, kv, lv);
    rp[i] = pos;
    int64_t k = 0;
    bool bad = false;
    const int64_t nf = c.sc->num_features;
    int64_t* cg = b->col_global;
    const int64_t cap = nz0 + r.nnz;
    if (ki == K_INT64) {
      if (!each_int64(li, [&](int64_t v) {
            if (pos + k < cap) cg[pos + k] = v;
            if (nf > 0 && (v < 0 || v >= nf)) bad = true;
            ++k;
          }))
        return rec_error(c, r, GDMIX_IO_EFORMAT, "bad int64 list");
    }
    if (pos + k > cap) return rec_error(c, r, GDMIX_IO_EFORMAT, "feature list changed between passes");
    if (bad) return rec_error(c, r, GDMIX_IO_ESCHEMA, "feature index outside [0, num_features)");
    if (k && !read_floats(lv, b->val + pos, k)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad float list");
    pos += k;
  }
  if (pos != nz0 + r.nnz) return rec_error(c, r, GDMIX_IO_EFORMAT, "feature list changed between passes");
  return GDMIX_IO_OK;
}

// ---- files -----------------------------------------------------------------------------------------
struct FileBuf {
  const uint8_t* data = nullptr;
  size_t size = 0;
  void* map = nullptr;       // mmap base (raw files)
  size_t map_size = 0;
  uint8_t* heap = nullptr;   // inflated contents
  ~FileBuf() {
    if (map) munmap(map, map_size);
    free(heap);
  }
};

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && memcmp(s.data() + s.size() - n, suf, n) == 0;
}

int load_file(const std::string& path, FileBuf& fb) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return fail(GDMIX_IO_EIO, "%s: cannot open", path.c_str());
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return fail(GDMIX_IO_EIO, "%s: cannot stat", path.c_str()); }
  const size_t sz = (size_t)st.st_size;
  if (sz == 0) { close(fd); return GDMIX_IO_OK; }
  void* m = mmap(nullptr, sz, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);   // populate: one bulk fault-in instead of one per page touched
  close(fd);
  if (m == MAP_FAILED) return fail(GDMIX_IO_EIO, "%s: cannot map", path.c_str());
  fb.map = m;
  fb.map_size = sz;
  // compression by suffix: ".deflate" = zlib stream, ".gz" = gzip (input_data_pipeline.py:63-85)
  const bool gz = ends_with(path, ".gz"), zl = ends_with(path, ".deflate");
  if (!gz && !zl) { fb.data = (const uint8_t*)m; fb.size = sz; return GDMIX_IO_OK; }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, gz ? 15 + 16 : 15) != Z_OK) return fail(GDMIX_IO_EIO, "%s: inflateInit2 failed", path.c_str());
  size_t cap = sz * 4 + (1 << 16), len = 0;
  uint8_t* out = (uint8_t*)malloc(cap);
  if (!out) { inflateEnd(&zs); return fail(GDMIX_IO_ENOMEM, "out of memory"); }
  const uint8_t* in = (const uint8_t*)m;
  size_t in_left = sz;
  int zr = Z_OK;
  while (zr != Z_STREAM_END) {
    if (zs.avail_in == 0 && in_left) {
      const size_t chunk = in_left > (1u << 30) ? (1u << 30) : in_left;
      zs.next_in = const_cast<Bytef*>(in);
      zs.avail_in = (uInt)chunk;
      in += chunk;
      in_left -= chunk;
    }
    if (len == cap) {
      cap *= 2;
      uint8_t* bigger = (uint8_t*)realloc(out, cap);
      if (!bigger) { free(out); inflateEnd(&zs); return fail(GDMIX_IO_ENOMEM, "out of memory"); }
      out = bigger;
    }
    const size_t room = cap - len > (1u << 30) ? (1u << 30) : cap - len;
    zs.next_out = out + len;
    zs.avail_out = (uInt)room;
    zr = inflate(&zs, Z_NO_FLUSH);
    len += room - zs.avail_out;
    if (zr != Z_OK && zr != Z_STREAM_END) {
      free(out);
      inflateEnd(&zs);
      return fail(GDMIX_IO_EIO, "%s: corrupt %s stream", path.c_str(), gz ? "gzip" : "zlib");
    }
    if (zr == Z_OK && zs.avail_in == 0 && in_left == 0 && zs.avail_out != 0) {
      free(out);
      inflateEnd(&zs);
      return fail(GDMIX_IO_EIO, "%s: truncated %s stream", path.c_str(), gz ? "gzip" : "zlib");
    }
  }
  inflateEnd(&zs);
  munmap(fb.map, fb.map_size);
  fb.map = nullptr;
  fb.heap = out;
  fb.data = out;
  fb.size = len;
  return GDMIX_IO_OK;
}

// TFRecord framing: uint64 length | uint32 masked_crc(length) | data | uint32 masked_crc(data)
int index_records(const std::string& path, const FileBuf& fb, int file, bool check_crc, std::vector<RecInfo>& recs) {
  size_t pos = 0;
  recs.reserve(fb.size / 512 + 16);
  while (pos < fb.size) {
    if (fb.size - pos < 12) return fail(GDMIX_IO_EFORMAT, "%s: truncated record header at byte %zu", path.c_str(), pos);
    uint64_t len;
    memcpy(&len, fb.data + pos, 8);
    if (check_crc) {
      uint32_t want;
      memcpy(&want, fb.data + pos + 8, 4);
      if (want != masked(crc32c(fb.data + pos, 8))) return fail(GDMIX_IO_EFORMAT, "%s: corrupt length CRC at byte %zu", path.c_str(), pos);
    }
    const size_t start = pos + 12;
    if (len > fb.size - start || fb.size - start - len < 4)
      return fail(GDMIX_IO_EFORMAT, "%s: truncated record at byte %zu", path.c_str(), pos);
    RecInfo r;
    r.rec.p = fb.data + start;
    r.rec.e = fb.data + start + len;
    r.file = file;
    r.has_label_col = false;
    r.n = r.nnz = r.id_len = 0;
    recs.push_back(r);
    pos = start + len + 4;
  }
  return GDMIX_IO_OK;
}

template <class Fn>
int parallel_for(int64_t count, int threads, Fn&& fn, int64_t chunk = 256) {
  if (count <= 0) return GDMIX_IO_OK;
  std::atomic<int64_t> next{0};
  std::atomic<int> rc{GDMIX_IO_OK};
  std::string first_err;
  std::atomic<int> err_set{0};
  auto work = [&]() {
    for (;;) {
      const int64_t b = next.fetch_add(chunk);
      if (b >= count || rc.load(std::memory_order_relaxed) != GDMIX_IO_OK) return;
      const int64_t e = b + chunk < count ? b + chunk : count;
      for (int64_t i = b; i < e; ++i) {
        const int r = fn(i);
        if (r != GDMIX_IO_OK) {
          int expected = 0;
          if (err_set.compare_exchange_strong(expected, 1)) { first_err = g_err; rc.store(r); }
          return;
        }
      }
    }
  };
  int nt = threads;
  if ((int64_t)nt * chunk > count) nt = (int)((count + chunk - 1) / chunk);
  if (nt <= 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
  }
  if (rc.load() != GDMIX_IO_OK) snprintf(g_err, sizeof(g_err), "%s", first_err.c_str());
  return rc.load();
}

// ---- array pool ---------------------------------------------------------------------------------------------------------
// The arrays of a batch are hundreds of MB: a fresh malloc maps them, the decode pass faults every page in and free() unmaps
// them again — on a 128-thread host that was half the wall time of a read. Blocks of at least POOL_MIN bytes are therefore kept
// (already faulted in) when a batch is freed and handed out again to the next read of a similar size, up to GDMIX_IO_POOL_MB
// of idle blocks — default 4096 MB divided by the number of worker processes on the node (LOCAL_WORLD_SIZE, as
// torch.distributed.run exports it), at least 512 MB: eight workers each holding 4 GB of idle pages would pin 32 GB that free()
// used to return. gdmix_io_pool_trim() releases every idle block (the drivers call it when a stage's partitions are done).
// A 32-byte header in front of every block holds its capacity.
constexpr size_t POOL_MIN = (size_t)1 << 20, POOL_HDR = 32;
struct PoolBlock { void* base; size_t cap; };
std::mutex g_pool_mu;
std::vector<PoolBlock> g_pool;
size_t g_pool_bytes = 0;

size_t pool_limit() {
  static size_t lim = [] {
    const char* e = getenv("GDMIX_IO_POOL_MB");
    long long mb = 4096;
    if (e) {
      mb = atoll(e);
    } else if (const char* l = getenv("LOCAL_WORLD_SIZE")) {
      const long long workers = atoll(l);
      if (workers > 1) mb = 4096 / workers < 512 ? 512 : 4096 / workers;
    }
    return (size_t)(mb < 0 ? 0 : mb) << 20;
  }();
  return lim;
}

void* pool_malloc(size_t bytes) {
  if (bytes >= POOL_MIN) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i)
      if (g_pool[i].cap >= bytes && g_pool[i].cap <= bytes + bytes / 2 && (best < 0 || g_pool[i].cap < g_pool[best].cap)) best = i;
    if (best >= 0) {
      void* base = g_pool[best].base;
      g_pool_bytes -= g_pool[best].cap;
      g_pool.erase(g_pool.begin() + best);
      return (char*)base + POOL_HDR;
    }
  }
  const size_t cap = bytes >= POOL_MIN ? bytes + bytes / 8 : bytes;   // a little slack: the next partition is rarely the same size
  void* base = malloc(cap + POOL_HDR);
  if (!base) return nullptr;
  *(size_t*)base = cap;
  return (char*)base + POOL_HDR;
}

void pool_free(void* p) {
  if (!p) return;
  void* base = (char*)p - POOL_HDR;
  const size_t cap = *(size_t*)base;
  if (cap >= POOL_MIN) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool_bytes + cap <= pool_limit()) {
      g_pool.push_back({base, cap});
      g_pool_bytes += cap;
      return;
    }
  }
  free(base);
}

size_t pool_trim() {
  std::vector<PoolBlock> idle;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    idle.swap(g_pool);
    bytes = g_pool_bytes;
    g_pool_bytes = 0;
  }
  for (auto& b : idle) free(b.base);
  return bytes;
}

}  // namespace
namespace gdmix_io_detail {
// the array pool for io_avro.cpp (model tables come and go like partitions do)
void* pool_alloc(size_t bytes) { return pool_malloc(bytes); }
void pool_release(void* p) { pool_free(p); }
size_t buffers_trim();   // io_avro.cpp: the writers' byte buffers
}  // namespace gdmix_io_detail
namespace {

template <class T>
bool alloc(T*& p, int64_t count) {
  p = (T*)pool_malloc((size_t)(count > 0 ? count : 1) * sizeof(T));
  return p != nullptr;
}

}  // namespace

extern "C" {

GDMIX_IO_API int gdmix_io_abi_version(void) { return GDMIX_IO_ABI_VERSION; }
GDMIX_IO_API const char* gdmix_io_last_error(void) { return g_err; }
GDMIX_IO_API uint32_t gdmix_io_crc32c(const void* data, size_t len) { return crc32c((const uint8_t*)data, len); }
GDMIX_IO_API uint32_t gdmix_io_masked_crc32c(const void* data, size_t len) { return masked(crc32c((const uint8_t*)data, len)); }

GDMIX_IO_API size_t gdmix_io_pool_trim(void) { return pool_trim() + gdmix_io_detail::buffers_trim(); }

GDMIX_IO_API void gdmix_io_free(gdmix_io_batch* b) {
  if (!b) return;
  pool_free(b->ent_row_ptr); pool_free(b->row_nnz_ptr); pool_free(b->col_global); pool_free(b->val); pool_free(b->y);
  pool_free(b->offset); pool_free(b->weight); pool_free(b->uid); pool_free(b->ent_id_ptr); pool_free(b->ent_id_bytes);
  pool_free(b->ent_n); pool_free(b->row_nnz); pool_free(b->col); pool_free(b->y8);
  free(b);
}

GDMIX_IO_API int gdmix_io_narrow(gdmix_io_batch* b, int32_t threads) {
  if (!b) return fail(GDMIX_IO_EINVAL, "batch is NULL");
  if (b->ent_n) return GDMIX_IO_OK;                       // already narrow
  if (!b->ent_row_ptr || !b->row_nnz_ptr || (b->Z > 0 && !b->col_global)) return fail(GDMIX_IO_EINVAL, "batch has no 64-bit arrays to narrow");
  if (threads <= 0) threads = gdmix_io_detail::default_threads();
  const int64_t E = b->E, N = b->N, Z = b->Z;
  // widths: one pass over the per-sample counts and the indices (maxima per chunk, then over the chunks)
  const int64_t CH = 1 << 16;
  const int64_t nch_n = (N + CH - 1) / CH, nch_z = (Z + CH - 1) / CH;
  std::vector<int64_t> kmax((size_t)(nch_n > 0 ? nch_n : 1), 0), cmax((size_t)(nch_z > 0 ? nch_z : 1), 0), cmin((size_t)(nch_z > 0 ? nch_z : 1), 0);
  int rc = parallel_for(nch_n, threads, [&](int64_t c) {
    int64_t m = 0;
    const int64_t e = (c + 1) * CH < N ? (c + 1) * CH : N;
    for (int64_t i = c * CH; i < e; ++i) { const int64_t k = b->row_nnz_ptr[i + 1] - b->row_nnz_ptr[i]; if (k > m) m = k; }
    kmax[(size_t)c] = m;
    return GDMIX_IO_OK;
  }, 1);
  if (rc != GDMIX_IO_OK) return rc;
  rc = parallel_for(nch_z, threads, [&](int64_t c) {
    int64_t m = 0, lo = 0;
    const int64_t e = (c + 1) * CH < Z ? (c + 1) * CH : Z;
    for (int64_t i = c * CH; i < e; ++i) { const int64_t v = b->col_global[i]; if (v > m) m = v; if (v < lo) lo = v; }
    cmax[(size_t)c] = m; cmin[(size_t)c] = lo;
    return GDMIX_IO_OK;
  }, 1);
  if (rc != GDMIX_IO_OK) return rc;
  int64_t km = 0, cm = 0, cl = 0;
  for (int64_t v : kmax) if (v > km) km = v;
  for (int64_t v : cmax) if (v > cm) cm = v;
  for (int64_t v : cmin) if (v < cl) cl = v;
  if (cl < 0 || cm > 0x7fffffffll) return fail(GDMIX_IO_ERANGE, "feature index outside [0, 2^31)");
  for (int64_t e = 0; e < E; ++e)
    if (b->ent_row_ptr[e + 1] - b->ent_row_ptr[e] > 0x7fffffffll) return fail(GDMIX_IO_ERANGE, "an entity has more than 2^31 samples");
  const int kw = km <= 0xff ? 1 : (km <= 0xffff ? 2 : 4);
  const int cw = cm <= 0xffff ? 2 : 4;
  const bool bytes_y = b->has_label && b->labels_binary;
  int32_t* ent_n = nullptr; uint8_t* row_nnz = nullptr; uint8_t* col = nullptr; uint8_t* y8 = nullptr;
  bool ok = alloc(ent_n, E) && alloc(row_nnz, N * kw) && alloc(col, Z * cw) && (!bytes_y || alloc(y8, N));
  if (!ok) { pool_free(ent_n); pool_free(row_nnz); pool_free(col); pool_free(y8); return fail(GDMIX_IO_ENOMEM, "out of memory"); }
  for (int64_t e = 0; e < E; ++e) ent_n[e] = (int32_t)(b->ent_row_ptr[e + 1] - b->ent_row_ptr[e]);
  parallel_for(nch_n, threads, [&](int64_t c) {
    const int64_t e = (c + 1) * CH < N ? (c + 1) * CH : N;
    for (int64_t i = c * CH; i < e; ++i) {
      const int64_t k = b->row_nnz_ptr[i + 1] - b->row_nnz_ptr[i];
      if (kw == 1) row_nnz[i] = (uint8_t)k; else if (kw == 2) ((uint16_t*)row_nnz)[i] = (uint16_t)k; else ((uint32_t*)row_nnz)[i] = (uint32_t)k;
      if (bytes_y) y8[i] = b->y[i] != 0.0f ? 1 : 0;
    }
    return GDMIX_IO_OK;
  }, 1);
  parallel_for(nch_z, threads, [&](int64_t c) {
    const int64_t e = (c + 1) * CH < Z ? (c + 1) * CH : Z;
    if (cw == 2) for (int64_t i = c * CH; i < e; ++i) ((uint16_t*)col)[i] = (uint16_t)b->col_global[i];
    else for (int64_t i = c * CH; i < e; ++i) ((int32_t*)col)[i] = (int32_t)b->col_global[i];
    return GDMIX_IO_OK;
  }, 1);
  pool_free(b->row_nnz_ptr); b->row_nnz_ptr = nullptr;
  pool_free(b->col_global); b->col_global = nullptr;
  b->ent_n = ent_n; b->row_nnz = row_nnz; b->col = col; b->y8 = y8; b->row_nnz_width = kw; b->col_width = cw;
  return GDMIX_IO_OK;
}

GDMIX_IO_API int gdmix_io_read_grouped(const char* const* files, int32_t n_files, const gdmix_io_schema* sc,
                                       gdmix_io_batch** out) {
  if (!out) return fail(GDMIX_IO_EINVAL, "out is NULL");
  *out = nullptr;
  if (!sc || n_files < 0 || (n_files > 0 && !files)) return fail(GDMIX_IO_EINVAL, "NULL argument");
  if (!sc->entity || !sc->offset || !sc->uid) return fail(GDMIX_IO_EINVAL, "schema needs entity, offset and uid column names");
  Ctx c;
  c.sc = sc;
  Names& nm = c.nm;
  nm.entity = sc->entity; nm.entity_len = strlen(sc->entity);
  nm.uid = sc->uid; nm.uid_len = strlen(sc->uid);
  nm.offset = sc->offset; nm.offset_len = strlen(sc->offset);
  nm.label = sc->label; nm.label_len = sc->label ? strlen(sc->label) : 0;
  nm.weight = sc->weight; nm.weight_len = sc->weight ? strlen(sc->weight) : 0;
  nm.has_bag = sc->feature_bag != nullptr;
  if (nm.has_bag) { nm.bag_idx = std::string(sc->feature_bag) + "_indices"; nm.bag_val = std::string(sc->feature_bag) + "_values"; }
  int threads = sc->threads;
  if (threads <= 0) threads = gdmix_io_detail::default_threads();

  const bool timing = getenv("GDMIX_IO_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[gdmix_io] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  std::vector<FileBuf> bufs((size_t)n_files);
  // record descriptors of all files, in entity order: a pooled block (56 MB for 400 k records: zero-filling and faulting a fresh
  // vector cost more than the copy into it)
  struct RecBlock { RecInfo* p = nullptr; ~RecBlock() { pool_free(p); } } recblk;
  RecInfo* recs = nullptr;
  int64_t E = 0;
  std::vector<std::vector<RecInfo>> file_recs((size_t)n_files);
  int64_t bytes = 0;
  for (int f = 0; f < n_files; ++f) {
    if (!files[f]) return fail(GDMIX_IO_EINVAL, "files[%d] is NULL", f);
    c.files.emplace_back(files[f]);
  }
  {
    // inflate / map and index the framing of the files in parallel (one file per task)
    std::atomic<int> nextf{0};
    std::atomic<int> rc{GDMIX_IO_OK};
    std::string err;
    std::atomic<int> err_set{0};
    auto work = [&]() {
      for (;;) {
        const int f = nextf.fetch_add(1);
        if (f >= n_files) return;
        int r = load_file(c.files[f], bufs[f]);
        if (r == GDMIX_IO_OK) r = index_records(c.files[f], bufs[f], f, sc->check_crc != 0, file_recs[f]);
        if (r != GDMIX_IO_OK) {
          int expected = 0;
          if (err_set.compare_exchange_strong(expected, 1)) { err = g_err; rc.store(r); }
        }
      }
    };
    const int nt = threads < n_files ? threads : n_files;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (rc.load() != GDMIX_IO_OK) { snprintf(g_err, sizeof(g_err), "%s", err.c_str()); return rc.load(); }
  }
  lap("load+index");
  {
    // entity order = file order, then record order; the per-file lists are copied into place in parallel (a serial
    // insert of 20 MB of record descriptors was a quarter of a 64-thread read)
    std::vector<size_t> first((size_t)n_files + 1, 0);
    for (int f = 0; f < n_files; ++f) { first[(size_t)f + 1] = first[(size_t)f] + file_recs[f].size(); bytes += (int64_t)bufs[f].size; }
    E = (int64_t)first[(size_t)n_files];
    static_assert(std::is_trivially_copyable<RecInfo>::value, "RecInfo is copied as bytes");
    recblk.p = recs = (RecInfo*)pool_malloc((size_t)(E > 0 ? E : 1) * sizeof(RecInfo));
    if (!recs) return fail(GDMIX_IO_ENOMEM, "out of memory");
    const int64_t piece = 1 << 15;
    std::vector<std::pair<int, size_t>> jobs;   // (file, first record of the piece inside the file)
    for (int f = 0; f < n_files; ++f)
      for (size_t o = 0; o < file_recs[f].size(); o += (size_t)piece) jobs.emplace_back(f, o);
    const int rcc = parallel_for((int64_t)jobs.size(), threads, [&](int64_t k) {
      const int f = jobs[(size_t)k].first;
      const size_t o = jobs[(size_t)k].second;
      const size_t cnt = std::min((size_t)piece, file_recs[f].size() - o);
      std::copy(file_recs[f].begin() + (ptrdiff_t)o, file_recs[f].begin() + (ptrdiff_t)(o + cnt), recs + (ptrdiff_t)(first[(size_t)f] + o));
      return (int)GDMIX_IO_OK;
    }, 1);
    if (rcc != GDMIX_IO_OK) return rcc;
    for (int f = 0; f < n_files; ++f) std::vector<RecInfo>().swap(file_recs[f]);
  }
  lap("concat");
  if (sc->check_crc) {
    const int rc = parallel_for(E, threads, [&](int64_t i) {
      const RecInfo& r = recs[i];
      uint32_t want;
      memcpy(&want, r.rec.e, 4);
      if (want != masked(crc32c(r.rec.p, r.rec.size())))
        return fail(GDMIX_IO_EFORMAT, "%s: corrupt data CRC in record %lld", c.files[r.file].c_str(), (long long)i);
      return (int)GDMIX_IO_OK;
    });
    if (rc != GDMIX_IO_OK) return rc;
  }
  lap("crc");
  int rc = parallel_for(E, threads, [&](int64_t i) { return locate(c, recs[i]); });
  if (rc != GDMIX_IO_OK) return rc;
  lap("locate");

  gdmix_io_batch* b = (gdmix_io_batch*)calloc(1, sizeof(gdmix_io_batch));
  if (!b) return fail(GDMIX_IO_ENOMEM, "out of memory");
  b->E = E;
  bool ok = alloc(b->ent_row_ptr, E + 1) && alloc(b->ent_id_ptr, E + 1);
  std::vector<int64_t> nz0((size_t)E + 1);
  // labels are kept up to the first record that has no label column (the reference reader stops trusting the
  // column from there on and reports the data as unlabelled)
  int64_t first_unlabelled = E;
  if (!sc->label) first_unlabelled = 0;
  if (ok) {
    int64_t N = 0, Z = 0, I = 0;
    for (int64_t i = 0; i < E; ++i) {
      b->ent_row_ptr[i] = N; nz0[(size_t)i] = Z; b->ent_id_ptr[i] = I;
      N += recs[i].n; Z += recs[i].nnz; I += recs[i].id_len;
      if (sc->label && !recs[i].has_label_col && i < first_unlabelled) first_unlabelled = i;
    }
    b->ent_row_ptr[E] = N; nz0[(size_t)E] = Z; b->ent_id_ptr[E] = I;
    b->N = N; b->Z = Z;
    b->has_label = (sc->label && first_unlabelled == E) ? 1 : 0;
    b->labels_binary = 1;
    b->bytes_read = bytes;
    ok = alloc(b->row_nnz_ptr, N + 1) && alloc(b->col_global, Z) && alloc(b->val, Z) && alloc(b->y, N) &&
         alloc(b->offset, N) && alloc(b->uid, N) && alloc(b->ent_id_bytes, I) && (!sc->weight || alloc(b->weight, N));
  }
  if (!ok) { gdmix_io_free(b); return fail(GDMIX_IO_ENOMEM, "out of memory"); }
  b->row_nnz_ptr[b->N] = b->Z;
  rc = parallel_for(E, threads, [&](int64_t i) {
    return decode(c, recs[i], b, i, b->ent_row_ptr[i], nz0[(size_t)i], b->ent_id_ptr[i], i < first_unlabelled);
  });
  if (rc != GDMIX_IO_OK) { gdmix_io_free(b); return rc; }
  b->labels_binary = c.labels_nonbinary.load() ? 0 : 1;
  lap("decode");
  *out = b;
  return GDMIX_IO_OK;
}

// ---- per-record (tf.train.Example) files: the fixed-effect stage's input -------------------------------------------
// One Example per sample (per_record_input_fn, gdmix-trainer/src/gdmix/io/input_data_pipeline.py:129-221): dense
// columns are one-element lists, the sparse bag two lists `<bag>_indices` / `<bag>_values`. schema->entity is unused;
// schema->offset / label / weight may be NULL (column not in the metadata: offset 0, label 0, weight 1, as
// fixed_effect_lr_lbfgs_model.py:255-258 defaults them). The batch comes back with E = 0 and no entity arrays.
GDMIX_IO_API int gdmix_io_read_examples(const char* const* files, int32_t n_files, const gdmix_io_schema* sc,
                                        gdmix_io_batch** out) {
  if (!out) return fail(GDMIX_IO_EINVAL, "out is NULL");
  *out = nullptr;
  if (!sc || n_files < 0 || (n_files > 0 && !files) || !sc->uid) return fail(GDMIX_IO_EINVAL, "NULL argument");
  int threads = sc->threads;
  if (threads <= 0) threads = gdmix_io_detail::default_threads();
  Ctx c;
  c.sc = sc;
  for (int f = 0; f < n_files; ++f) {
    if (!files[f]) return fail(GDMIX_IO_EINVAL, "files[%d] is NULL", f);
    c.files.emplace_back(files[f]);
  }
  std::vector<FileBuf> bufs((size_t)n_files);
  std::vector<std::vector<RecInfo>> file_recs((size_t)n_files);
  {
    std::atomic<int> nextf{0}, rc{GDMIX_IO_OK}, err_set{0};
    std::string err;
    auto work = [&]() {
      for (;;) {
        const int f = nextf.fetch_add(1);
        if (f >= n_files) return;
        int r = load_file(c.files[f], bufs[f]);
        if (r == GDMIX_IO_OK) r = index_records(c.files[f], bufs[f], f, sc->check_crc != 0, file_recs[f]);
        if (r != GDMIX_IO_OK) { int e = 0; if (err_set.compare_exchange_strong(e, 1)) { err = g_err; rc.store(r); } }
      }
    };
    const int nt = threads < n_files ? threads : n_files;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (rc.load() != GDMIX_IO_OK) { snprintf(g_err, sizeof(g_err), "%s", err.c_str()); return rc.load(); }
  }
  std::vector<RecInfo> recs;
  int64_t bytes = 0;
  for (int f = 0; f < n_files; ++f) {
    recs.insert(recs.end(), file_recs[f].begin(), file_recs[f].end());
    std::vector<RecInfo>().swap(file_recs[f]);
    bytes += (int64_t)bufs[f].size;
  }
  const int64_t N = (int64_t)recs.size();
  const std::string bag_i = sc->feature_bag ? std::string(sc->feature_bag) + "_indices" : std::string();
  const std::string bag_v = sc->feature_bag ? std::string(sc->feature_bag) + "_values" : std::string();
  const size_t l_uid = strlen(sc->uid), l_off = sc->offset ? strlen(sc->offset) : 0, l_lab = sc->label ? strlen(sc->label) : 0,
               l_w = sc->weight ? strlen(sc->weight) : 0;
  // pass 1: locate the columns of every record (f_uid / f_offset / f_label / f_weight / fl_idx / fl_val hold Feature messages)
  int rc = parallel_for(N, threads, [&](int64_t i) {
    RecInfo& r = recs[i];
    if (sc->check_crc) {
      uint32_t want;
      memcpy(&want, r.rec.e, 4);
      if (want != masked(crc32c(r.rec.p, r.rec.size()))) return rec_error(c, r, GDMIX_IO_EFORMAT, "corrupt data CRC");
    }
    Span msg = r.rec, feats;
    Field f;
    while (!msg.empty()) {
      if (!next_field(msg, f)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad Example");
      if (f.fn == 1 && f.wt == 2) feats = f.payload;
    }
    bool have_uid = false, have_off = false, have_lab = false, have_w = false;
    while (!feats.empty()) {
      if (!next_field(feats, f)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad Features");
      if (f.fn != 1 || f.wt != 2) continue;
      Span entry = f.payload, key, val;
      bool has_key = false;
      Field g;
      while (!entry.empty()) {
        if (!next_field(entry, g)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad Features entry");
        if (g.wt != 2) continue;
        if (g.fn == 1) { key = g.payload; has_key = true; }
        else if (g.fn == 2) val = g.payload;
      }
      if (!has_key) continue;
      if (key_is(key, sc->uid, l_uid)) { r.f_uid = val; have_uid = true; }
      else if (key_is(key, sc->offset, l_off)) { r.f_offset = val; have_off = true; }
      else if (key_is(key, sc->label, l_lab)) { r.f_label = val; have_lab = true; }
      else if (key_is(key, sc->weight, l_w)) { r.f_weight = val; have_w = true; }
      else if (sc->feature_bag && key_is(key, bag_i.c_str(), bag_i.size())) r.fl_idx = val;
      else if (sc->feature_bag && key_is(key, bag_v.c_str(), bag_v.size())) r.fl_val = val;
    }
    if (!have_uid) return rec_error(c, r, GDMIX_IO_ESCHEMA, "uid column is missing");
    if (sc->offset && !have_off) return rec_error(c, r, GDMIX_IO_ESCHEMA, "offset column is missing");
    if (sc->label && !have_lab) return rec_error(c, r, GDMIX_IO_ESCHEMA, "label column is missing");
    if (sc->weight && !have_w) return rec_error(c, r, GDMIX_IO_ESCHEMA, "weight column is missing");
    int64_t m;
    int e;
    if ((e = column_len(c, r, r.f_uid, true, false, "uid column must hold one int64", m))) return e;
    if (m != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "uid column must hold one value per record");
    if (sc->offset) { if ((e = column_len(c, r, r.f_offset, true, true, "bad offset column", m))) return e; if (m != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "offset column must hold one value per record"); }
    if (sc->label) { if ((e = column_len(c, r, r.f_label, true, true, "bad label column", m))) return e; if (m != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "label column must hold one value per record"); }
    if (sc->weight) { if ((e = column_len(c, r, r.f_weight, true, true, "bad weight column", m))) return e; if (m != 1) return rec_error(c, r, GDMIX_IO_ESCHEMA, "weight column must hold one value per record"); }
    int64_t a = 0, b = 0;
    if (sc->feature_bag) {
      if ((e = column_len(c, r, r.fl_idx, true, false, "feature indices must be an int64 list", a))) return e;
      if ((e = column_len(c, r, r.fl_val, false, true, "feature values must be a float list", b))) return e;
      if (a != b) return rec_error(c, r, GDMIX_IO_ESCHEMA, "indices and values of a sample differ in length");
    }
    r.n = 1;
    r.nnz = a;
    return (int)GDMIX_IO_OK;
  });
  if (rc != GDMIX_IO_OK) return rc;
  gdmix_io_batch* b = (gdmix_io_batch*)calloc(1, sizeof(gdmix_io_batch));
  if (!b) return fail(GDMIX_IO_ENOMEM, "out of memory");
  b->E = 0; b->N = N; b->has_label = sc->label ? 1 : 0; b->bytes_read = bytes;
  bool ok = alloc(b->row_nnz_ptr, N + 1);
  int64_t Z = 0;
  if (ok) { for (int64_t i = 0; i < N; ++i) { b->row_nnz_ptr[i] = Z; Z += recs[i].nnz; } b->row_nnz_ptr[N] = Z; b->Z = Z; }
  ok = ok && alloc(b->col_global, Z) && alloc(b->val, Z) && alloc(b->y, N) && alloc(b->offset, N) && alloc(b->uid, N) &&
       alloc(b->weight, N);
  if (!ok) { gdmix_io_free(b); return fail(GDMIX_IO_ENOMEM, "out of memory"); }
  auto one_float = [&](Span feat, float& outv) {   // int64 or float scalar -> float
    Kind kind;
    Span list;
    if (!feature_kind(feat, kind, list)) return false;
    if (kind == K_FLOAT) return read_floats(list, &outv, 1);
    if (kind == K_INT64) { int64_t v = 0; if (!read_int64s(list, &v, 1)) return false; outv = (float)v; return true; }
    return false;
  };
  rc = parallel_for(N, threads, [&](int64_t i) {
    const RecInfo& r = recs[i];
    Kind kind;
    Span list;
    feature_kind(r.f_uid, kind, list);
    if (!read_int64s(list, b->uid + i, 1)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad uid");
    b->offset[i] = 0.0f; b->y[i] = 0.0f; b->weight[i] = 1.0f;
    if (sc->offset && !one_float(r.f_offset, b->offset[i])) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad offset");
    if (sc->label && !one_float(r.f_label, b->y[i])) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad label");
    if (sc->weight && !one_float(r.f_weight, b->weight[i])) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad weight");
    if (sc->feature_bag && r.nnz) {
      const int64_t z0 = b->row_nnz_ptr[i];
      feature_kind(r.fl_idx, kind, list);
      if (!read_int64s(list, b->col_global + z0, r.nnz)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad index list");
      if (sc->num_features > 0)
        for (int64_t k = 0; k < r.nnz; ++k)
          if (b->col_global[z0 + k] < 0 || b->col_global[z0 + k] >= sc->num_features)
            return rec_error(c, r, GDMIX_IO_ESCHEMA, "feature index outside [0, num_features)");
      feature_kind(r.fl_val, kind, list);
      if (!read_floats(list, b->val + z0, r.nnz)) return rec_error(c, r, GDMIX_IO_EFORMAT, "bad value list");
    }
    return (int)GDMIX_IO_OK;
  });
  if (rc != GDMIX_IO_OK) { gdmix_io_free(b); return rc; }
  *out = b;
  return GDMIX_IO_OK;
}

}  // extern "C"
