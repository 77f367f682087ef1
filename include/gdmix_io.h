/* gdmix_io.h — C ABI of libgdmix_io.so: native reader of entity-grouped TFRecord partitions.
 *
 * Replaces, for the random-effect path, what the reference does with the TensorFlow runtime before the
 * solver sees any data (SURVEY.md §8 rows a9 + the slicing half of a8, next-row N3):
 *   per_entity_grouped_input_fn      gdmix-trainer/src/gdmix/io/input_data_pipeline.py:223-332
 *   dataset_reader                   gdmix-trainer/src/gdmix/util/io_utils.py:351-364
 *   prepare_jobs (per-entity slices) gdmix-trainer/src/gdmix/models/custom/scipy/job_consumers.py:209-258
 * One SequenceExample per entity: the entity id is a scalar in the context, every
 * dense column a per-sample list in the context, the sparse bag two feature lists `<bag>_indices` /
 * `<bag>_values` with one step per sample. Output: the entity-major ragged arrays gdmix_re_pack consumes
 * (gdmix_re.h, gdmix_re_raw_batch), in host memory.
 *
 * Host code only (no HIP). Every function returns 0 or a negative code and never throws;
 * gdmix_io_last_error() gives the message of the calling thread's last failure. Input files are untrusted:
 * every length and offset is bounds-checked.
 */
#ifndef GDMIX_IO_H
#define GDMIX_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GDMIX_IO_API __attribute__((visibility("default")))
#else
#define GDMIX_IO_API
#endif

#define GDMIX_IO_OK        0
#define GDMIX_IO_EINVAL   (-1)   /* bad argument */
#define GDMIX_IO_EIO      (-2)   /* file cannot be opened / read / inflated */
#define GDMIX_IO_EFORMAT  (-3)   /* framing, CRC or protobuf wire error */
#define GDMIX_IO_ESCHEMA  (-4)   /* a record does not match the schema (missing column, length mismatch, ...) */
#define GDMIX_IO_ENOMEM   (-5)
#define GDMIX_IO_ERANGE   (-6)   /* a value does not fit the 32-bit hand-over form (gdmix_io_narrow) */

#define GDMIX_IO_ABI_VERSION 7

typedef struct gdmix_io_schema {
  const char* entity;        /* context key of the entity id (int64 or bytes scalar)                       */
  const char* feature_bag;   /* sparse bag name; NULL => intercept-only model: one dummy zero feature per
                              * sample (job_consumers.py:213-218)                                          */
  const char* offset;        /* context key, float list [n]                                                */
  const char* uid;           /* context key, int64 list [n]                                                */
  const char* label;         /* context key, int64 or float list [n]; NULL or absent from a record =>
                              * has_label = 0 and y = 0 (inference data)                                   */
  const char* weight;        /* context key, float list [n]; NULL => no weight array                        */
  int64_t num_features;      /* > 0: feature indices must lie in [0, num_features); <= 0: unchecked         */
  int32_t check_crc;         /* verify the masked CRC-32C of every record (length and data)                 */
  int32_t threads;           /* decode threads; <= 0 (here and in every call below): one per online CPU, at most 32;
                              * the environment variable GDMIX_IO_THREADS overrides that default            */
} gdmix_io_schema;

/* Arrays are owned by the library until gdmix_io_free. Entity order = file order, then record order. */
typedef struct gdmix_io_batch {
  int64_t E, N, Z;
  int64_t* ent_row_ptr;   /* [E+1] sample offsets                                  */
  int64_t* row_nnz_ptr;   /* [N+1] non-zero offsets                                */
  int64_t* col_global;    /* [Z]   global feature index                            */
  float*   val;           /* [Z]                                                   */
  float*   y;             /* [N]   labels as float (0 when has_label == 0)         */
  float*   offset;        /* [N]                                                   */
  float*   weight;        /* [N] or NULL                                           */
  int64_t* uid;           /* [N]                                                   */
  int64_t* ent_id_ptr;    /* [E+1] offsets into ent_id_bytes                       */
  char*    ent_id_bytes;  /* entity ids: UTF-8 bytes, or the decimal rendering of an int64 id
                           * (job_consumers.py:235-239)                            */
  int32_t  has_label;
  int64_t  bytes_read;    /* decompressed bytes of TFRecord framing parsed         */
  int32_t  labels_binary; /* every label read is exactly 0 or 1 (what fit() asserts, binary_logistic_regression.py:208);
                           * checked while decoding, so that the caller need not pass over y again */
  /* the 32-bit hand-over form (include/gdmix_re.h: gdmix_re_wire_batch), filled by gdmix_io_narrow; NULL / 0 before */
  int32_t* ent_n;         /* [E] samples per entity                                                          */
  void*    row_nnz;       /* [N] non-zeros per sample, row_nnz_width bytes each (1, 2 or 4: the narrowest that fits) */
  void*    col;           /* [Z] global feature index, col_width bytes each (2 when every index is below 65 536, else 4) */
  uint8_t* y8;            /* [N] labels as bytes when has_label and labels_binary, else NULL                     */
  int32_t  row_nnz_width, col_width;
} gdmix_io_batch;

GDMIX_IO_API int gdmix_io_abi_version(void);
/* (ABI 7) hash of the sources this binary was compiled from (gdmix_amd/build.py: io_source_id()); see gdmix_re_build_id. */
GDMIX_IO_API const char* gdmix_io_build_id(void);
GDMIX_IO_API const char* gdmix_io_last_error(void);

/* Read every record of `files` (".gz" => gzip, ".deflate" => zlib, anything else raw: input_data_pipeline.py:63-85)
 * into one batch. */
GDMIX_IO_API int gdmix_io_read_grouped(const char* const* files, int32_t n_files, const gdmix_io_schema* schema,
                                       gdmix_io_batch** out);
GDMIX_IO_API void gdmix_io_free(gdmix_io_batch* batch);

/* Round 4: turn a decoded batch into the 32-bit form a partition crosses PCIe in — counts instead of 64-bit
 * pointers, int32 / uint16 feature indices, byte labels: 0.47 of the bytes for a C2 partition — in pooled blocks, and give the two large
 * 64-bit arrays (row_nnz_ptr, col_global) back to the pool: they become NULL; ent_row_ptr, val, y, offset, weight, uid and the ids stay.
 * One parallel pass over the arrays. Fails (GDMIX_IO_ERANGE) if a feature index is outside [0, 2^31) or an entity has 2^31 samples. */
GDMIX_IO_API int gdmix_io_narrow(gdmix_io_batch* batch, int32_t threads);

/* The library keeps freed arrays (and the Avro writers' byte buffers) for the next partition instead of returning them to the
 * allocator: up to GDMIX_IO_POOL_MB megabytes of idle blocks (environment; default 4096 / LOCAL_WORLD_SIZE, at least 512) plus up
 * to 1 GB of writer buffers. gdmix_io_pool_trim releases every idle block now and returns the number of bytes released; arrays
 * that are still in use are not touched. */
GDMIX_IO_API size_t gdmix_io_pool_trim(void);

/* Per-record files of the fixed-effect stage: one tf.train.Example per sample (per_record_input_fn,
 * io/input_data_pipeline.py:129-221); dense columns hold one value, the sparse bag is `<bag>_indices` / `<bag>_values`.
 * schema->entity is ignored; offset / label / weight may be NULL (defaults 0 / 0 / 1). Returns a batch with E = 0:
 * row_nnz_ptr [N+1], col_global, val, y, offset, weight, uid. */
GDMIX_IO_API int gdmix_io_read_examples(const char* const* files, int32_t n_files, const gdmix_io_schema* schema,
                                        gdmix_io_batch** out);

/* Write a batch as one SequenceExample per entity (".gz" / ".deflate" suffix => compressed): the layout
 * DataPartitioner produces (DataPartitioner.scala:313-316) and gdmix_io_read_grouped reads. int_entity_ids != 0:
 * entity ids (decimal strings) are written as an int64 scalar instead of bytes. Labels are written as int64
 * when batch->has_label; schema->weight NULL or batch->weight NULL => no weight column; num_features, check_crc
 * and threads of the schema are ignored. */
GDMIX_IO_API int gdmix_io_write_grouped(const char* path, const gdmix_io_batch* batch, const gdmix_io_schema* schema,
                                        int32_t int_entity_ids);

/* ---- Avro object container writers (model and score files) ----------------------------------------------
 * The caller supplies the container header (magic "Obj\x01", the metadata map with avro.schema / avro.codec,
 * the 16-byte sync marker) and the constant, pre-encoded pieces of a record; the library encodes the records in
 * blocks of `block_records` (blocks in parallel, written in order). deflate_codec != 0: raw-deflate every block
 * (avro.codec must then say "deflate"). */
typedef struct gdmix_io_model_table {   /* one BayesianLinearModelAvro per entity, in array order */
  int64_t E;
  const int64_t* id_ptr;        /* [E+1] offsets into id_bytes                                              */
  const char*    id_bytes;      /* modelId strings, UTF-8                                                   */
  const int64_t* coef_beg;      /* [E] first coefficient of the entity in mean (intercept first)            */
  const int64_t* coef_cnt;      /* [E] number of coefficients (features + has_intercept)                    */
  const int64_t* var_beg;       /* [E] first variance in `variance`, < 0 => the entity has none; or NULL    */
  const int64_t* feat_beg;      /* [E] first entry in feat_idx                                              */
  const double*  mean;
  const double*  variance;      /* or NULL                                                                  */
  const int64_t* feat_idx;      /* global feature index of every non-intercept coefficient                  */
  const int64_t* prefix_ptr;    /* [n_prefix+1] offsets into prefix_bytes; NULL => intercept-only file      */
  const uint8_t* prefix_bytes;  /* per global feature: Avro string(name) + string(term)                     */
  int64_t        n_prefix;
  const uint8_t* icpt_enc;      /* string("(INTERCEPT)") + string("")                                       */
  int64_t        icpt_len;
  const uint8_t* class_enc;     /* modelClass union: branch + string                                        */
  int64_t        class_len;
  const uint8_t* loss_enc;      /* lossFunction union: branch + string                                      */
  int64_t        loss_len;
  int32_t        has_intercept;
  double         threshold;     /* features with |value| <= threshold are not written (intercept always is) */
} gdmix_io_model_table;

GDMIX_IO_API int gdmix_io_avro_write_models(const char* path, const uint8_t* header, int64_t header_len, const uint8_t* sync,
                                            const gdmix_io_model_table* table, int32_t block_records, int32_t deflate_codec,
                                            int32_t threads);

/* validation_result records: uid long, score float, label [null, float] (label NULL => null), weight float
 * (weight NULL => the schema has no weight field), per-coordinate score float. */
GDMIX_IO_API int gdmix_io_avro_write_scores(const char* path, const uint8_t* header, int64_t header_len, const uint8_t* sync,
                                            int64_t n, const int64_t* uid, const float* score, const float* label,
                                            const float* weight, const float* per_coord, int32_t block_records,
                                            int32_t deflate_codec, int32_t threads);

/* ---- Avro model file reader ---------------------------------------------------------------------------------
 * Every BayesianLinearModelAvro record of an object container file as flat arrays: replaces the record-by-record
 * loop of RandomEffectLRLBFGSModel._load_weights / _convert_avro_model_record_to_sparse_coefficients
 * (gdmix-trainer/src/gdmix/models/custom/random_effect_lr_lbfgs_model.py:256-309) for the prior model of a warm
 * start and the model of an inference run. The caller parses the container header (data_offset = first byte after
 * the header's sync marker) and checks that the writer schema has the canonical field order; prefix_* is the feature
 * list as pre-encoded string(name) + string(term) (as for gdmix_io_avro_write_models; of equal pairs the last wins,
 * like the reference's dict). has_intercept: the first coefficient of every record must be the intercept
 * (it carries no feature index), everything else must be in the feature list (GDMIX_IO_ESCHEMA otherwise = the reference's
 * AssertionError / KeyError). A variances array must be empty or line up with means. Blocks are decoded in parallel. */
typedef struct gdmix_io_models {
  int64_t  E;             /* records, in file order                                            */
  int64_t  C;             /* coefficients over all records                                     */
  int64_t* id_ptr;        /* [E+1] offsets into id_bytes                                       */
  char*    id_bytes;      /* modelId strings                                                   */
  int64_t* coef_ptr;      /* [E+1] offsets into mean / variance / feat_idx                     */
  double*  mean;          /* [C]                                                               */
  double*  variance;      /* [C] 0 where the record has none                                   */
  int64_t  F;             /* C - E * has_intercept                                             */
  int64_t* feat_idx;      /* [F] global feature index of every non-intercept coefficient, in order: record e
                           * owns feat_idx[coef_ptr[e] - e * has_intercept .. )                            */
  uint8_t* has_variance;  /* [E]                                                               */
  int32_t  any_variance;
} gdmix_io_models;

GDMIX_IO_API int gdmix_io_avro_read_models(const char* path, int64_t data_offset, const uint8_t* sync, int32_t deflate_codec,
                                           const int64_t* prefix_ptr, const uint8_t* prefix_bytes, int64_t n_prefix,
                                           const uint8_t* icpt_enc, int64_t icpt_len, int32_t has_intercept, int32_t threads,
                                           gdmix_io_models** out);
GDMIX_IO_API void gdmix_io_free_models(gdmix_io_models* models);

/* ---- model coefficients in a batch's index space ---------------------------------------------------------------
 * The warm start of prepare_jobs (gdmix-trainer/src/gdmix/models/custom/scipy/job_consumers.py:262-288) and the
 * coefficient vector InferenceJobConsumer scores with, for all entities of a packed batch at once. Entity e of the
 * batch has features cur_idx[cur_ptr[e] .. cur_ptr[e+1]) (global indices) and, when src_row[e] >= 0, a model in row
 * src_row[e] of a table (coefficients prior_theta[prior_coef_ptr[r] ..), intercept first when has_intercept; global
 * feature indices prior_idx[prior_feat_ptr[r] ..)). Writes into theta (laid out [intercept,] features per entity, i.e.
 * entity e starts at cur_ptr[e] + e * has_intercept; zero-filled by the caller, or here when zero_first: in parallel, each
 * stretch by the thread that writes into it next): the model's intercept and, for every
 * batch feature the model has, its coefficient (of equal indices in a model the first listed). Entities without a model
 * are left untouched. */
GDMIX_IO_API int gdmix_io_map_coefficients(int64_t E, const int64_t* cur_ptr, const int64_t* cur_idx, const int64_t* src_row,
                                           const int64_t* prior_coef_ptr, const int64_t* prior_feat_ptr,
                                           const double* prior_theta, const int64_t* prior_idx, int32_t has_intercept,
                                           double* theta, int32_t zero_first, int32_t threads);

/* 1 when the E byte strings bytes[ptr[e] .. ptr[e+1]) are pairwise different, 0 when two are equal, < 0 on error. The model
 * dict of random_effect_lr_lbfgs_model.py:155-162 keeps one entry per entity id; a partition whose ids are all different (the
 * normal case) never has to become that dict on the host. */
GDMIX_IO_API int gdmix_io_ids_unique(const char* bytes, const int64_t* ptr, int64_t E);

/* row_in_a[j] = the i with a[i] == b[j] (byte strings a_bytes[a_ptr[i] .. a_ptr[i+1]) of a table whose ids are all different),
 * or -1: the id lookups of a warm start / an inference run / a carry-over (random_effect_lr_lbfgs_model.py:141-162,
 * job_consumers.py:262-288) for a whole partition at once. */
GDMIX_IO_API int gdmix_io_match_ids(const char* a_bytes, const int64_t* a_ptr, int64_t Ea, const char* b_bytes, const int64_t* b_ptr,
                                    int64_t Eb, int64_t* row_in_a, int32_t threads);

/* CRC-32C (Castagnoli) and TFRecord's masked form ((crc >> 15 | crc << 17) + 0xa282ead8). */
GDMIX_IO_API uint32_t gdmix_io_crc32c(const void* data, size_t len);
GDMIX_IO_API uint32_t gdmix_io_masked_crc32c(const void* data, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* GDMIX_IO_H */
