// io_avro.cpp — libgdmix_io.so: native writers of the two Avro object container files the random-effect stage
// produces (include/gdmix_io.h), from the Avro 1.x specification:
//   model file   BayesianLinearModelAvro per entity   export_linear_model_to_avro / gen_one_avro_model,
//                                                     gdmix-trainer/src/gdmix/util/io_utils.py:102-212
//   score file   validation_result per sample         batched_write_avro, util/io_utils.py:299-334,367-375
// The caller (gdmix_amd/model.py) supplies the container header (magic, metadata map with the schema JSON,
// sync marker) and the pre-encoded constant pieces; this file encodes the records in blocks of
// `block_records` (the reference writes 1024-record blocks), blocks in parallel, and appends them in order.
// Byte for byte what gdmix_amd/io/avro.py + model.py's Python encoders write (tests/test_native_io.py).
#include "../../include/gdmix_io.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

extern "C" const char* gdmix_io_last_error(void);
namespace gdmix_io_detail { int set_error(int code, const char* fmt, ...); }
using gdmix_io_detail::set_error;

namespace {

inline void put_long(std::string& out, int64_t n) {
  uint64_t u = ((uint64_t)n << 1) ^ (uint64_t)(n >> 63);
  while (u >= 0x80) { out.push_back((char)((u & 0x7F) | 0x80)); u >>= 7; }
  out.push_back((char)u);
}
inline void put_double(std::string& out, double v) { out.append((const char*)&v, 8); }
inline void put_float(std::string& out, float v) { out.append((const char*)&v, 4); }

// raw deflate of one block payload (the Avro "deflate" codec: RFC 1951, no zlib header / checksum)
bool raw_deflate(const std::string& in, std::string& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  out.resize(deflateBound(&zs, (uLong)in.size()));
  zs.next_in = (Bytef*)in.data();
  zs.avail_in = (uInt)in.size();
  zs.next_out = (Bytef*)&out[0];
  zs.avail_out = (uInt)out.size();
  const int rc = deflate(&zs, Z_FINISH);
  out.resize(out.size() - zs.avail_out);
  deflateEnd(&zs);
  return rc == Z_STREAM_END;
}

// Encode blocks [0, n_blocks) with `encode(block, payload)` on `threads` threads, a bounded number of blocks in
// memory at a time, and append them to the file in order.
template <class Enc>
int write_blocks(const char* path, const uint8_t* header, int64_t header_len, const uint8_t* sync, int64_t total,
                 int32_t block_records, int32_t deflate_codec, int32_t threads, Enc&& encode) {
  if (!path || !header || !sync || header_len <= 0 || block_records <= 0) return set_error(GDMIX_IO_EINVAL, "bad argument");
  FILE* f = fopen(path, "wb");
  if (!f) return set_error(GDMIX_IO_EIO, "%s: cannot open for writing", path);
  bool ok = fwrite(header, 1, (size_t)header_len, f) == (size_t)header_len;
  const int64_t n_blocks = (total + block_records - 1) / block_records;
  if (threads <= 0) { threads = (int)std::thread::hardware_concurrency(); if (threads <= 0) threads = 1; }
  const int64_t round = (int64_t)threads * 4;
  std::vector<std::string> out((size_t)round);
  std::atomic<int> failed{0};
  for (int64_t b0 = 0; b0 < n_blocks && ok; b0 += round) {
    const int64_t nb = (b0 + round < n_blocks) ? round : n_blocks - b0;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
      std::string payload, packed;
      for (;;) {
        const int64_t k = next.fetch_add(1);
        if (k >= nb) return;
        const int64_t blk = b0 + k;
        const int64_t r0 = blk * block_records;
        const int64_t r1 = (r0 + block_records < total) ? r0 + block_records : total;
        payload.clear();
        encode(r0, r1, payload);
        const std::string* body = &payload;
        if (deflate_codec) {
          if (!raw_deflate(payload, packed)) { failed.store(1); return; }
          body = &packed;
        }
        std::string& o = out[(size_t)k];
        o.clear();
        put_long(o, r1 - r0);
        put_long(o, (int64_t)body->size());
        o.append(*body);
        o.append((const char*)sync, 16);
      }
    };
    const int nt = (int64_t)threads < nb ? threads : (int)nb;
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (failed.load()) { ok = false; break; }
    for (int64_t k = 0; k < nb && ok; ++k) ok = fwrite(out[(size_t)k].data(), 1, out[(size_t)k].size(), f) == out[(size_t)k].size();
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) return set_error(GDMIX_IO_EIO, "%s: write failed", path);
  return GDMIX_IO_OK;
}

}  // namespace

extern "C" {

GDMIX_IO_API int gdmix_io_avro_write_models(const char* path, const uint8_t* header, int64_t header_len, const uint8_t* sync,
                                            const gdmix_io_model_table* t, int32_t block_records, int32_t deflate_codec,
                                            int32_t threads) {
  if (!t || t->E < 0) return set_error(GDMIX_IO_EINVAL, "bad model table");
  if (t->E > 0 && (!t->id_ptr || !t->id_bytes || !t->coef_beg || !t->coef_cnt || !t->mean))
    return set_error(GDMIX_IO_EINVAL, "model table has NULL arrays");
  if (t->prefix_ptr && t->E > 0 && (!t->feat_beg || !t->feat_idx || !t->prefix_bytes))
    return set_error(GDMIX_IO_EINVAL, "model table has NULL feature arrays");
  const int ic = t->has_intercept ? 1 : 0;
  std::atomic<int> bad{0};
  const int rc = write_blocks(path, header, header_len, sync, t->E, block_records, deflate_codec, threads,
                              [&](int64_t r0, int64_t r1, std::string& out) {
    std::string means, vars;
    for (int64_t e = r0; e < r1; ++e) {
      const int64_t c0 = t->coef_beg[e], p = t->coef_cnt[e];
      const double* mean = t->mean + c0;
      const double* var = (t->variance && t->var_beg && t->var_beg[e] >= 0) ? t->variance + t->var_beg[e] : nullptr;
      means.clear();
      vars.clear();
      int64_t items = 0;
      if (ic && p > 0) {
        means.append((const char*)t->icpt_enc, (size_t)t->icpt_len);
        put_double(means, mean[0]);
        if (var) { vars.append((const char*)t->icpt_enc, (size_t)t->icpt_len); put_double(vars, var[0]); }
        ++items;
      }
      if (t->prefix_ptr) {
        const int64_t* idx = t->feat_idx + t->feat_beg[e];
        for (int64_t k = 0; k + ic < p; ++k) {
          const double v = mean[ic + k];
          if (!(v > t->threshold || v < -t->threshold)) continue;   // keep |v| > threshold
          const int64_t g = idx[k];
          if (g < 0 || g >= t->n_prefix) { bad.store(1); continue; }
          const uint8_t* pre = t->prefix_bytes + t->prefix_ptr[g];
          const size_t len = (size_t)(t->prefix_ptr[g + 1] - t->prefix_ptr[g]);
          means.append((const char*)pre, len);
          put_double(means, v);
          if (var) { vars.append((const char*)pre, len); put_double(vars, var[ic + k]); }
          ++items;
        }
      }
      const int64_t i0 = t->id_ptr[e], i1 = t->id_ptr[e + 1];
      put_long(out, i1 - i0);
      out.append(t->id_bytes + i0, (size_t)(i1 - i0));
      out.append((const char*)t->class_enc, (size_t)t->class_len);
      if (items) { put_long(out, items); out.append(means); }
      put_long(out, 0);
      if (var) {
        put_long(out, 1);
        if (items) { put_long(out, items); out.append(vars); }
        put_long(out, 0);
      } else {
        put_long(out, 0);
      }
      out.append((const char*)t->loss_enc, (size_t)t->loss_len);
    }
  });
  if (rc != GDMIX_IO_OK) return rc;
  if (bad.load()) return set_error(GDMIX_IO_ESCHEMA, "%s: a coefficient's global feature index is outside the feature list", path);
  return GDMIX_IO_OK;
}

GDMIX_IO_API int gdmix_io_avro_write_scores(const char* path, const uint8_t* header, int64_t header_len, const uint8_t* sync,
                                            int64_t n, const int64_t* uid, const float* score, const float* label,
                                            const float* weight, const float* per_coord, int32_t block_records,
                                            int32_t deflate_codec, int32_t threads) {
  if (n < 0 || (n > 0 && (!uid || !score || !per_coord))) return set_error(GDMIX_IO_EINVAL, "score arrays are NULL");
  return write_blocks(path, header, header_len, sync, n, block_records, deflate_codec, threads,
                      [&](int64_t r0, int64_t r1, std::string& out) {
    for (int64_t i = r0; i < r1; ++i) {
      put_long(out, uid[i]);
      put_float(out, score[i]);
      if (label) { out.push_back((char)0x02); put_float(out, label[i]); }   // union branch 1: float
      else out.push_back((char)0x00);                                       // union branch 0: null
      if (weight) put_float(out, weight[i]);
      put_float(out, per_coord[i]);
    }
  });
}

}  // extern "C"
