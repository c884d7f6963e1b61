/*
 * orama_hip.h — C ABI of liborama_hip.so: the MI355X (gfx950) implementation of OramaCore's
 * hybrid-search scoring hot path.  This header IS the drop-in boundary: every entry point cites
 * the reference interface it replaces (paths relative to the oramacore repository root), and
 * INTEGRATION.md shows the Rust `extern "C"` shim that binds it inside the reference's wrappers.
 *
 * Conventions (mirroring the reference's seams — SURVEY.md §8b):
 *   - plain pointers + sizes, no C++/torch types; every function returns an `int` status
 *     (ORAMA_OK == 0) and never throws/aborts across the boundary;  the message of the last
 *     failure on the calling thread is available from orama_last_error() — the shim turns it into
 *     `anyhow::Error` (→ ReadError::Generic → HTTP 500, src/collection_manager/sides/read/mod.rs:136-138).
 *   - borrowed inputs: query / filter / posting pointers are only read during the call;
 *     outputs are caller-allocated buffers with the documented capacity.
 *   - handles are owned by the wrapper struct that created them and are destroyed with it.
 *   - re-entrancy: `*_search*` may be called concurrently from many threads on one handle and
 *     concurrently with insert/delete (the reference calls search(&self) from many tokio workers
 *     under a read lock — src/collection_manager/sides/read/collection.rs:846-884); the library
 *     serialises mutation internally and gives each search its own stream + scratch.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     ORAMA_ERR_HIP.
 *
 * Limits of the implemented envelope.  A request beyond one of them is well-formed but not served:
 * the call returns ORAMA_ERR_UNSUPPORTED (never a truncated answer) and the shim should route it to the
 * reference's own CPU path.  ORAMA_ERR_INVALID is reserved for malformed arguments (null pointers,
 * mismatched dimensions, zero where the reference itself has no meaning for it).
 *   - result size: limit / top_k <= 4096 per query (exact radix select + one-workgroup sort of the
 *     winners); for a shard group additionally shards * k <= 4096 (the merge runs in one workgroup);
 *     group-by max_results <= 1024.  (Whole score maps of ANY size: orama_post_search_scores.)
 *   - query tokens: n_tokens <= 64.  Threshold masks use bit (1 << (t % 32)) like the reference's u32
 *     shift in release mode (token_score.rs:286-288), so > 32 tokens alias there too.
 *   - a vector store holds < 2^32 - 16 rows, a postings store < 2^32 - 1 documents, one query references
 *     < 2^32 - 1 postings (row / local-document indices are 32-bit on the device; DocumentIds are
 *     64-bit everywhere).  Vector dimensions <= 65536.
 *   - Concurrency bound: every search / insert leases per-call scratch (stream + buffers) from its context; at most
 *     ORAMA_MAX_INFLIGHT (default 32) sets are out at a time, further callers wait their turn, and a caller that waited
 *     ORAMA_ACQUIRE_TIMEOUT_MS (default 30 000) without getting its sets fails with ORAMA_ERR_BUSY — nothing was started,
 *     the request can be retried.  Sets kept by long-lived handles (orama_scores) do NOT count against the bound, so open
 *     score maps can never starve the searches; a co-located shard group takes one set per shard per call, so it holds at
 *     most ORAMA_MAX_INFLIGHT / 2 shards (orama_shard_group_create refuses more).
 *   - fp16 query batches: any q; 65..256 queries share one corpus pass, larger batches run in
 *     passes of 256.  Up to 64 queries run as one pass of the LDS-resident form while their fragments fit
 *     (dimensions <= 1024), as passes of 32 above that (dimensions <= 2048 for fp16 storage).
 */
#ifndef ORAMA_HIP_H
#define ORAMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): every search / insert entry point may return ORAMA_ERR_BUSY (5) — acquire times out where version 1 blocked —;
 * orama_ctx_set_f16_wide refuses the comparison kernels (modes 1, 5) in the product library; co-located groups are limited to 16
 * shards; a postings store keeps 4 more bytes per posting (the pre-divided normalised tf).  A shim built against version 1
 * must map status 5 before it is relinked. */
#define ORAMA_ABI_VERSION 2

/* status codes */
#define ORAMA_OK 0
#define ORAMA_ERR_INVALID 1     /* bad argument (null handle, dim mismatch, k == 0 where illegal …) */
#define ORAMA_ERR_HIP 2         /* HIP runtime failure / no device */
#define ORAMA_ERR_OOM 3         /* device or host allocation failed */
#define ORAMA_ERR_UNSUPPORTED 4 /* valid request outside the implemented envelope */
#define ORAMA_ERR_BUSY 5        /* the call could not get its per-call scratch sets in time (see "Concurrency bound") */

/* DistanceMetric — oramacore_fields::embedding::DistanceMetric; the reference hard-codes Cosine
 * (src/collection_manager/sides/read/index/embedding_field.rs:66,88).  L2 is a build-side extension. */
#define ORAMA_METRIC_COSINE 0 /* distance = 1 - cos(q, x) */
#define ORAMA_METRIC_L2SQ 1   /* distance = |q - x|^2 */

/* storage element type of the HBM-resident corpus.  The reference stores f32 (Vec<f32> end to
 * end, src/collection_manager/sides/operation/op.rs:144); f16 is a build-side extension. */
#define ORAMA_DTYPE_F32 0
#define ORAMA_DTYPE_F16 1
/* fp32 rows PLUS an fp16 copy of the same rows ("shadow", +50 % HBM): orama_vec_search (and the request batcher on top
 * of it) runs in two stages — the shadow scan (half the bytes; MFMA for query batches) proposes max(2k, k + 256)
 * candidates, the fp32 rows give their exact distances and the final order.  The answer is the fp32 scan's, bit for bit:
 * a query whose candidate list cannot be PROVEN to contain the exact top-k (error bound of the fp16 image, see
 * DESIGN §4 K1s) is answered by the plain fp32 scan instead.  Cosine metric, dimensions % 4 == 0 and <= 1024, k <= 2048;
 * outside that the store behaves as ORAMA_DTYPE_F32.  orama_hybrid_search and orama_shard_vec_search /
 * orama_shard_hybrid_search take the two stages too (shards begin together and are joined one by one).  The *_device entry
 * points and the pipelined shard session — which never return to the host between their launches — take them for k <= 128
 * with the fallback decided and run ON THE DEVICE: the queries that are not proven (or that the fp16 image cannot serve: a
 * zero vector, components >= 6e4, NaN) are re-answered behind the selection by the fp32 scan's own kernel over a
 * device-made list of queries; with nothing to re-answer those launches end at once. */
#define ORAMA_DTYPE_F32_SHADOW16 2

typedef struct orama_ctx orama_ctx;   /* one per GPU: device ordinal, stream + scratch pools */
typedef struct orama_vec orama_vec;   /* one per embedding field  (EmbeddingFieldStorage) */
typedef struct orama_post orama_post; /* one per index: resident postings of its string fields */

int orama_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). Never NULL. */
const char* orama_last_error(void);

/* ------------------------------------------------------------------ context */
/* GPUs visible to this process (what a shim sizes its shard group with; the reference has no counterpart — its
 * ReadSide is one CPU process, src/collection_manager/sides/read/mod.rs:621-738). */
int orama_device_count(int* out);
int orama_ctx_create(int device_ordinal, orama_ctx** out);
void orama_ctx_destroy(orama_ctx* ctx);
int orama_ctx_synchronize(orama_ctx* ctx);
/* Device facts for reports: name (<= 255 chars), CU count, HBM bytes. Any pointer may be NULL. */
int orama_ctx_device_info(orama_ctx* ctx, char* name256, int* compute_units, uint64_t* hbm_bytes);
/* PCI address of the context's device, "dddd:bb:dd.f" (hipDeviceGetPCIBusId) — what a monitor needs to find THIS device under
 * /sys/bus/pci/devices/ or in an SMI library: HIP ordinals are not DRM card numbers (bench.py's clock / power sampler). */
int orama_ctx_pci_bus_id(orama_ctx* ctx, char* out, int capacity);

/* Launch geometry of the K1 scan (tuning sweeps; defaults are the measured best on MI355X):
 * rows each wave keeps in flight (1/2/4/8), persistent workgroups per CU, nontemporal corpus loads. */
int orama_ctx_set_scan_tuning(orama_ctx* ctx, int rows_per_wave, int blocks_per_cu, int nontemporal);
/* Register-ring geometry of the K2 fp16 scan: k-steps per chunk (8/12/16) and chunks in the ring (2..4). */
int orama_ctx_set_f16_tuning(orama_ctx* ctx, int ksteps_per_chunk, int ring_chunks);
/* Kernel used for fp16 batches of 65..256 queries per corpus pass: 0 = K2 in passes of 64, 1 = K2c (MFMA waves also
 * issue the LDS-DMA), 2 / 3 = K2d producer/consumer kernel, geometry 1 / 2 (vec_f16_pc.hip), 4 = K2q, queries
 * stationary in registers (vec_f16_qs.hip; batches of <= 128 and rows wider than 768 dimensions take K2d), 5 = K2h,
 * two query tiles per wave with the K loop split over a wave pair (vec_f16_kh.hip).  Default 4.  Same results. */
int orama_ctx_set_f16_wide(orama_ctx* ctx, int mode);
/* Scorer of the BM25 searches over a resident store: 1 (default) = K3r, the document-range partitioned scorer that takes
 * whole query batches per launch (bm25_ranges.hip) — for the plain top-k search and, where no OMC applies, for
 * orama_post_search_hybrid; 2 = K3r for the plain search only; 0 = K3, per-document records in HBM (bm25_kernels.hip),
 * which the score-map / precomputed-ntf / fused-hybrid entry points always use.  Modes 0-2 leave the key-list form alone (option
 * "k3r_compact" of orama_ctx_set_option); 3 / 4 are kept as shorthands: mode 1 + "k3r_compact" 0 (round 4's lists, one key slot
 * per posting) / 2 (compact lists for every batch size).  Same results bit for bit. */
int orama_ctx_set_bm25_ranges(orama_ctx* ctx, int on);
/* ORAMA_DTYPE_F32_SHADOW16 stores: 1 (default) = two-stage search where it pays (batches of more than 8 queries, or at
 * least 4 GB of fp32 rows: below that the plain scan is faster than the second stage's launches), 2 = always two stages,
 * 0 = always the plain fp32 scan.  Same results in every mode. */
int orama_ctx_set_two_stage(orama_ctx* ctx, int on);
/* Plain fp32 stores (the reference's own dtype: Vec<f32> rows, embedding_field.rs:66,88,232-237): batches of at least
 * `min_queries` concurrent queries share corpus passes of <= 32 queries on the matrix cores (K1m, vec_f32_mfma.hip:
 * v_mfma_f32_32x32x2_f32 — f32 in, f32 accumulate, exact) with a per-query threshold filter instead of one dense distance
 * array per query; smaller batches take K1 / K1b (<= 8 queries per pass, VALU).  Default 9; 0 = never.  Cosine stores whose
 * dimension is a multiple of 32 and <= 1024 (the query tile lives in LDS), k <= 128; everything else keeps K1 / K1b.  The matrix
 * scan only PROPOSES candidates — K1x (the rows rounded to fp16 in registers, <= 64 queries per pass, HBM-bound) where every row
 * has a sound fp16 image, else K1m (f32 x f32, <= 32 per pass) — K1's own arithmetic decides, and a candidate list that is not
 * proven complete is re-answered by K1 on the device: a query's answer is the single-query scan's, bit for bit, in any batch. */
int orama_ctx_set_f32_batch(orama_ctx* ctx, int min_queries);
/* Tuning / test options of a context, by name — NOT needed by a deployment (the defaults are the measured choices of DESIGN.md) and
 * never read from the environment by the product library (its whole environment is the deployment list of INTEGRATION.md §5b).
 * Every option selects between code paths that return the SAME answers (the parity tests run them against each other):
 *   "fused_topk" 0/1/2, "f32_multi" 0/1, "f32_batch_cvt" 0/1 (fp32 batches: the candidate scan rounds the rows to fp16 in registers — K1x,
 *   <= 64 queries per pass, HBM-bound — or multiplies f32 by f32 — K1m, <= 32 per pass; the answer is K1's either way), "f16_solo" 0..2, "f16_wide" 0..5, "f16_kc" 8|12|16, "f16_nbuf" 2..4,
 *   "f16_head_rows" (0 = 131 072), "f16_cand_mib" (0 = 6 144), "f16_chunk_grow" -1/0/1, "f16_grow_factor" 2..64,
 *   "two_stage_spare" 1..4096, "k3r_target" 16..2048, "k3r_compact" 0/1/2 (key lists: one slot per posting / compact for batches
 *   of >= 8 queries / always compact), "bm25_ranges" 0/1, "bm25_ranges_hybrid" 0/1, "select_wide" 0..3,
 *   "select_pairs" 0/1, "hybrid_device_tail" 0/1, "direct_out" 0/1, "stage_copy" 0/1 (1 = by kernel), "scan_done_event" 0/1.
 * ORAMA_ERR_INVALID for an unknown name or a value outside the option's range.  (Comparison builds — ORAMA_COMPARISON_KERNELS=1,
 * liborama_hip_cmp.so — additionally accept each option as ORAMA_<NAME> in the environment, with the timing ablations and traces.) */
int orama_ctx_set_option(orama_ctx* ctx, const char* name, long long value);

/* Per-kernel HIP-event timing (used by bench.py's roofline leg).  When enabled, the library
 * brackets each launch of the named hot kernels with hipEvents on the launching stream.
 * kernels: "vec_scan_f32", "vec_scan_f16", "topk_select", "bm25_accumulate", "bm25_finalize",
 * "bm25_range_bounds", "bm25_range_df", "bm25_range_score", "shard_all_gather" (one rank per process), "shard_merge". */
int orama_prof_enable(orama_ctx* ctx, int on);
int orama_prof_reset(orama_ctx* ctx);
int orama_prof_get(orama_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);
/* The individual launch durations behind orama_prof_get's sum, oldest first (the most recent 65 536 are kept): the roofline is
 * quoted on the MEDIAN launch.  Writes min(*n, capacity) values; `out_ms` may be NULL to ask for the count. */
int orama_prof_samples(orama_ctx* ctx, const char* kernel, float* out_ms, uint64_t capacity, uint64_t* n);

/* Raw HBM buffers for callers that drive the *_device entry points themselves (tests, bench.py, a shim that keeps
 * queries / results resident): plain hipMalloc / hipMemcpy on the context's device — no torch, no other runtime.
 * `offset` is in bytes.  Copies are synchronous with respect to the host. */
int orama_dev_malloc(orama_ctx* ctx, uint64_t bytes, void** out);
void orama_dev_free(orama_ctx* ctx, void* d_ptr);
int orama_dev_upload(orama_ctx* ctx, void* d_dst, uint64_t offset, const void* src, uint64_t bytes);
int orama_dev_download(orama_ctx* ctx, const void* d_src, uint64_t offset, void* dst, uint64_t bytes);
/* HIP streams owned by the library (priority: 0 normal, 1 high — high-priority streams get their own hardware
 * queues on ROCm, which keeps a launch-bound tail from queueing behind a corpus scan). */
int orama_stream_create(orama_ctx* ctx, int high_priority, void** out_stream);
void orama_stream_destroy(orama_ctx* ctx, void* stream);
int orama_stream_synchronize(orama_ctx* ctx, void* stream);

/* ------------------------------------------------------------------ vector store
 * Replaces oramacore_fields::embedding::EmbeddingStorage behind
 * EmbeddingFieldStorage (src/collection_manager/sides/read/index/embedding_field.rs:63-320). */

/* EmbeddingFieldStorage::new — embedding_field.rs:65-76 (EmbeddingConfig::new(dim, Cosine)). */
int orama_vec_create(orama_ctx* ctx, uint32_t dim, int metric, int dtype, uint64_t reserve_rows,
                     orama_vec** out);
void orama_vec_destroy(orama_vec* v);

/* EmbeddingFieldStorage::insert — embedding_field.rs:232-237 (N vectors per doc allowed).
 * `rows` is n_rows x dim f32 row-major (host); doc_ids[i] is the DocumentId of row i.
 * Rows that are non-finite or have zero norm are rejected (index_vec_vec -> None);
 * *accepted (nullable) receives the number of rows stored. */
int orama_vec_insert(orama_vec* v, const uint64_t* doc_ids, const float* rows, uint64_t n_rows,
                     uint64_t* accepted);
/* EmbeddingFieldStorage::delete — embedding_field.rs:240-242: tombstones every row of each doc. */
int orama_vec_delete(orama_vec* v, const uint64_t* doc_ids, uint64_t n);
/* EmbeddingFieldStorage::compact — embedding_field.rs:286-290: drops tombstoned rows and
 * re-packs the HBM matrix. `version` is recorded (current_version_number, :298-300). */
int orama_vec_compact(orama_vec* v, uint64_t version);
/* EmbeddingFieldStorage::{stats,has_pending_ops,current_version_number} — :281-310. */
typedef struct {
    uint32_t dimensions;
    uint64_t num_embeddings; /* live rows */
    uint64_t num_rows;       /* live + tombstoned rows resident in HBM */
    uint64_t pending_ops;    /* tombstones not yet compacted */
    uint64_t version;
    uint64_t hbm_bytes;      /* bytes of HBM held by this store (incl. its fp16 shadow) */
    uint64_t two_stage_queries;   /* ORAMA_DTYPE_F32_SHADOW16: queries answered by the two-stage plan ... */
    uint64_t two_stage_fallbacks; /* ... of which the candidate list could not be proven complete (fp32 scan instead) */
} orama_vec_info_t;
int orama_vec_info(orama_vec* v, orama_vec_info_t* out);

/* EmbeddingStorage::search / search_with_filter — call site embedding_field.rs:255-266.
 *   queries      : q x dim f32 row-major (host). q == 1 is the reference's shape; q > 1 is the
 *                  batch extension and returns q independent result lists.
 *   k            : `limit` (VectorSearchParams.limit, committed_field/vector.rs:10-15).
 *   allow_bitmap : NULL, or the materialised DocumentFilter (embedding_field.rs:54-61): bit d of the
 *                  little-endian u64 array set <=> doc id d passes; ids >= bitmap_bits never pass.
 *   out_ids/out_dist : q x k (host); out_n : q counts. Row-level results, distance ascending
 *                  (ties: doc id asc), several rows of one doc may appear — the caller sums them
 *                  (embedding_field.rs:268-276), exactly as with the reference's storage. */
int orama_vec_search(orama_vec* v, const float* queries, uint32_t q, uint32_t k,
                     const uint64_t* allow_bitmap, uint64_t bitmap_bits, uint64_t* out_ids,
                     float* out_dist, uint32_t* out_n);

/* Same, with every buffer already in HBM and work enqueued on `hip_stream` (a hipStream_t, NULL =
 * the default stream) without host synchronisation — used by the sharded path so that RCCL can
 * consume the candidates in place.  d_out_rows (nullable) receives store row indices; out ids are
 * DocumentIds.  Unused tail entries (i >= d_out_n[q]) are (UINT64_MAX, +inf). */
int orama_vec_search_device(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                            const uint64_t* d_allow_bitmap, uint64_t bitmap_bits,
                            uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                            void* hip_stream);

/* K6 — merge per-shard candidate lists after the RCCL all-gather: for each of q
 * queries, `lists` shards x k candidates laid out [shard][query][k] -> best k by (distance asc,
 * id asc). All pointers are device memory; enqueued on hip_stream. */
int orama_merge_candidates_device(orama_ctx* ctx, const uint64_t* d_ids, const float* d_dist,
                                  uint32_t lists, uint32_t q, uint32_t k, uint64_t* d_out_ids,
                                  float* d_out_dist, uint32_t* d_out_n, void* hip_stream);

/* Packed exchange block for the sharded path: one rank's candidates for q queries laid out as
 * [q*k u64 ids][q*k f32 distances] and padded to a multiple of 8 bytes, so that ONE all-gather moves
 * ids and distances together.  orama_packed_block_bytes gives the block size;
 * orama_vec_search_packed_device writes the local block (same semantics as
 * orama_vec_search_device); orama_merge_packed_device merges `lists` consecutive blocks. */
uint64_t orama_packed_block_bytes(uint32_t q, uint32_t k);
int orama_vec_search_packed_device(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                                   const uint64_t* d_allow_bitmap, uint64_t bitmap_bits,
                                   void* d_packed_block, uint32_t* d_out_n, void* hip_stream);
/* Two-stream form: the corpus scan(s) are enqueued on `scan_stream`, the top-k tail on `tail_stream` (the
 * library inserts the scan→tail and tail→next-scan event dependencies).  A caller that alternates between two
 * tail streams keeps ONE scan in flight at all times while the launch-bound tail (top-k, all-gather, merge) of
 * the previous query runs beside it — consecutive scans never share HBM bandwidth. */
int orama_vec_search_packed_device2(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                                    const uint64_t* d_allow_bitmap, uint64_t bitmap_bits, void* d_packed_block,
                                    uint32_t* d_out_n, void* scan_stream, void* tail_stream);
int orama_merge_packed_device(orama_ctx* ctx, const void* d_packed_blocks, uint32_t lists, uint32_t q,
                              uint32_t k, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                              void* hip_stream);

/* Test/bench utilities (no reference counterpart). */
/* Fill the store with n synthetic rows generated in HBM: x = u * g/|g|,
 * g ~ N(0,1)^dim, u ~ U(0.5, 2), counter-based RNG keyed by (seed, row, col); doc id = first_doc_id + row. */
int orama_vec_fill_synthetic(orama_vec* v, uint64_t n_rows, uint64_t seed, uint64_t first_doc_id);
/* Read back stored rows (as f32) by store row index. */
int orama_vec_get_rows(orama_vec* v, const uint64_t* row_idx, uint64_t n, float* out_rows,
                       uint64_t* out_doc_ids);

/* ------------------------------------------------------------------ resident allow-bitmaps
 * A filter — FilterResult<DocumentId> (index/filter.rs:344-392), including the "NOT uncommitted-deleted" predicate
 * every search carries while deletes are pending — materialised once as a bit per DocumentId and kept in HBM.
 * orama_allow_token() is what a search takes: ANY `allow_bitmap` argument of this header accepts either host
 * words (uploaded for that call) or the token of a resident bitmap (used in place, no PCIe traffic;
 * `bitmap_bits` must not exceed the bitmap's size).  The shim caches handles per filter hash and flips single bits
 * with orama_allow_set when a document is deleted / re-admitted (under the index's write lock, like the delete
 * itself: a search in flight may see either state).  Destroy only after searches using the token returned.
 * A postings store remembers the document frequencies it had to COUNT under a resident bitmap (corpus_docs.len() of
 * collect_contributions_with_filter, token_score.rs:262-275) per set of lists and content of the bitmap: the next query with
 * the same lists under the same bitmap is scored in one launch, like an unfiltered one (orama_allow_set starts a new
 * content; host words are never remembered). */
typedef struct orama_allow orama_allow;
int orama_allow_create(orama_ctx* ctx, const uint64_t* words, uint64_t bitmap_bits, orama_allow** out);
void orama_allow_destroy(orama_allow* a);
const uint64_t* orama_allow_token(const orama_allow* a);
int orama_allow_set(orama_allow* a, const uint64_t* doc_ids, uint64_t n, int allowed);

/* ------------------------------------------------------------------ request micro-batcher
 * The reference API has no batch entry (EmbeddingFieldStorage::search takes ONE target,
 * embedding_field.rs:250-254); concurrent single-query callers are coalesced here so that one corpus pass serves
 * up to max_batch requests (K2's MFMA path at Q <= 64 per pass, K2d above).  orama_batcher_search blocks the calling
 * thread until its answer is ready and has the semantics of orama_vec_search(q = 1, no filter); max_wait_us = 0 means
 * "never delay": a batch is whatever arrived while the previous pass was running.  Extension — no reference
 * counterpart; the Rust shim would call it from spawn_blocking. */
typedef struct orama_batcher orama_batcher;
int orama_batcher_create(orama_vec* v, uint32_t max_batch, uint32_t max_wait_us, orama_batcher** out);
void orama_batcher_destroy(orama_batcher* b);
int orama_batcher_search(orama_batcher* b, const float* query, uint32_t k, uint64_t* out_ids, float* out_dist,
                         uint32_t* out_n);
/* The same with the request's filter (semantics of orama_vec_search with allow_bitmap — the reference's
 * search_with_filter, embedding_field.rs:255-262).  A batch shares one bitmap: requests are grouped by their
 * (allow_bitmap pointer, bitmap_bits) pair, so pass the SAME resident token (orama_allow_token) for the same filter —
 * e.g. the index's NOT-deleted bitmap, which every search carries while deletes are pending (index/filter.rs:344-392). */
int orama_batcher_search_filtered(orama_batcher* b, const float* query, uint32_t k, const uint64_t* allow_bitmap,
                                  uint64_t bitmap_bits, uint64_t* out_ids, float* out_dist, uint32_t* out_n);
int orama_batcher_stats(orama_batcher* b, uint64_t* requests, uint64_t* batches, uint32_t* largest_batch);

/* ------------------------------------------------------------------ BM25F full-text scoring
 * Replaces the in-tree hot loop search_full_text + BM25Scorer + top_n:
 *   src/collection_manager/sides/read/index/token_score.rs:257-302,
 *   src/collection_manager/bm25.rs:325-525, src/collection_manager/sides/read/sort.rs:260-279. */

/* One posting-list entry = the per_doc_ntf vector returned by one field for one query token
 * (ContributionsResult.token_contributions[i].per_doc_ntf, token_score.rs:264-271). */
typedef struct {
    uint32_t token;      /* query token index (term_index) */
    const uint64_t* doc; /* host, `len` DocumentIds; a doc MAY repeat inside an entry — the reference scorer pushes every
                          * (doc, ntf) pair (bm25.rs:355-366) and sums them in order; repeats are split into extra ranks */
    const float* ntf;    /* host, `len` normalised TFs (boost + length norm already folded in) */
    uint64_t len;
} orama_ntf_entry;

typedef struct {
    float total_documents;  /* N: index document_count as f32 — token_score.rs:221 */
    float k;                /* 1.2 — token_score.rs:283 */
    uint32_t n_tokens;
    int use_threshold;      /* 0: BM25Scorer::plain, 1: with_threshold — token_score.rs:211-218 */
    uint32_t threshold;     /* floor(n_tokens * threshold) */
    uint32_t top_k;         /* limit + offset (sort.rs:24-34) */
} orama_bm25_params;

/* Seam (i): host-provided contributions (exact inputs of the in-tree loop) -> BM25F scores,
 * optional OMC multiply (search.rs:39-48; omc_* nullable, doc ids ascending), match count
 * (token_score_results.len(), search.rs:482) and top-k (score desc, doc id asc; NaN dropped).
 * Entries are consumed in array order inside each token.  out_* hold top_k entries. */
int orama_bm25_score(orama_ctx* ctx, const orama_ntf_entry* entries, uint32_t n_entries,
                     const orama_bm25_params* params, const uint64_t* omc_doc, const float* omc_mul,
                     uint64_t n_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                     uint64_t* out_count);

/* BM25Scorer::get_scores (bm25.rs:416-428) for seam (i): the WHOLE score map of the same computation, any size —
 * *out_n receives the number of entries; with capacity >= *out_n the (DocumentId, score) pairs are written (order
 * unspecified, like a HashMap's); capacity == 0 only counts.  (The selection entry above is limited to top_k <= 4096.) */
int orama_bm25_score_map(orama_ctx* ctx, const orama_ntf_entry* entries, uint32_t n_entries,
                         const orama_bm25_params* params, const uint64_t* omc_doc, const float* omc_mul, uint64_t n_omc,
                         uint64_t capacity, uint64_t* out_ids, float* out_scores, uint64_t* out_n);

/* Seam (ii): HBM-resident postings (SURVEY §8b option ii).  The store mirrors what
 * StringFieldStorage holds per field (src/collection_manager/sides/read/index/string_field.rs):
 * for each (field, term) a doc-sorted list of (doc, tf, field_length).  Term dictionary lookup /
 * prefix / fuzzy expansion stay on the host (third-party a5); the caller passes term ids. */
int orama_post_create(orama_ctx* ctx, orama_post** out);
void orama_post_destroy(orama_post* p);
/* Bulk build.  docs: n_docs DocumentIds ascending (the index's live documents; N = n_docs).
 * For list l in [0, n_lists): field_of_list[l], postings [list_off[l], list_off[l+1]) of
 * (post_doc, post_tf, post_len) with post_doc ascending and present in docs.
 * avg_field_len[f] for f in [0, n_fields) — StringStorage::info().avg_field_length. */
int orama_post_build(orama_post* p, const uint64_t* docs, uint64_t n_docs, uint32_t n_fields,
                     const float* avg_field_len, uint32_t n_lists, const uint32_t* field_of_list,
                     const uint64_t* list_off, const uint64_t* post_doc, const uint32_t* post_tf,
                     const uint32_t* post_len);
/* Live update between commits (StringFieldStorage::insert, string_field.rs:155-170): append
 * n_docs NEW documents (ids ascending and greater than every stored id — DocumentIds are handed out sequentially,
 * write/collection_document_storage.rs:73-77) and n_lists NEW posting lists holding their postings.  The new lists
 * get the ids n_lists_before … n_lists_before + n_lists − 1 (orama_post_info); a query references the committed
 * list AND the delta list(s) of a term with the same `token` in consecutive orama_term_ref entries — exactly how
 * several fields of one token are passed.  avg_field_len[n_fields] replaces the field averages (they move with
 * every insert).  Deletes reach the scorer through the allow bitmap (the reference's NOT-deleted predicate,
 * see orama_allow_*) until the next orama_post_build.  Runs under the store's exclusive lock. */
int orama_post_append(orama_post* p, const uint64_t* docs, uint64_t n_docs, const float* avg_field_len,
                      uint32_t n_lists, const uint32_t* field_of_list, const uint64_t* list_off,
                      const uint64_t* post_doc, const uint32_t* post_tf, const uint32_t* post_len);
/* Bench utility (no reference counterpart): synthetic postings generated in HBM.  n_docs documents
 * with dense ids [first_doc_id, first_doc_id + n_docs), one field, field length ~ LogNormal(4.0, 0.6) clipped to
 * [4, 2000]; list l models the term of Zipf(1.07) rank ranks[l] over a 2^20 vocabulary:
 * df = n_docs * (1 - exp(-avg_len * rank^-1.07 / H)), docs stratified-uniform, tf in 1..3. */
int orama_post_fill_synthetic(orama_post* p, uint64_t n_docs, uint64_t first_doc_id, uint32_t n_lists,
                              const uint32_t* ranks, uint64_t seed, uint64_t* out_total_postings);

/* Read back one posting list (test/bench checker) and the store's shape (StringStorage::info()). */
int orama_post_get_list(orama_post* p, uint32_t list, uint64_t capacity, uint64_t* out_doc, uint32_t* out_tf,
                        uint32_t* out_len, uint64_t* out_n);
int orama_post_info(orama_post* p, uint64_t* n_docs, uint32_t* n_lists, uint64_t* n_postings, float* avg_len0);

/* OMC multipliers of the index (Index::get_all_omc, index/mod.rs:1720-1739); doc ids ascending.  The multipliers are
 * stored densely against the CURRENT doc table: orama_post_build / orama_post_fill_synthetic drop them — call
 * orama_post_set_omc again after every rebuild (orama_post_append keeps them, new documents get 1.0). */
int orama_post_set_omc(orama_post* p, const uint64_t* omc_doc, const float* omc_mul, uint64_t n);

/* One (token, list) reference of a query: token `token` expands to posting list `list`
 * scored with `boost` = the field boost (SearchParams.boost, token_score.rs:234-238) TIMES the exact-match factor of the
 * third-party string store when the list's dictionary term is the query token itself (token_score.rs:182-185, 226-228: "ntf
 * already includes boost + length normalization + exact_match_boost").  The factor's value lives in oramacore_fields 0.2.0, not in
 * the reference checkout; boost_integration.rs:449-491 pins that it is > 1.  The library takes the product as given: the caller
 * (the Rust shim: AcceleratorConfig.exact_match_boost; include/orama/host.hpp orama::term_ref; the Python mirror) applies it. */
typedef struct {
    uint32_t token;
    uint32_t list;
    float boost;
} orama_term_ref;

/* search_full_text over resident postings: ntf = boost * (tf / (1 - b + b * (len / avglen)))
 * (bm25.rs:99-110 with Bm25Params::default b = 0.75 passed in `b`), then exactly as seam (i).
 * allow_bitmap as in orama_vec_search (the filter is applied to postings, so df counts only
 * allowed docs — collect_contributions_with_filter, string_field.rs:215-225). */
int orama_post_search(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                      const orama_bm25_params* params, const uint64_t* allow_bitmap,
                      uint64_t bitmap_bits, int apply_omc, uint64_t* out_ids, float* out_scores,
                      uint32_t* out_n, uint64_t* out_count);

/* Many independent orama_post_search queries in one call (batch extension, no reference counterpart — the reference
 * reaches this path one request per tokio worker): up to `max_parallel` (0 = 8) library threads each take the next
 * query and run the single-query path on their own stream + scratch set, so the launch-bound kernels of different
 * queries overlap on the device.  Results of query i: out_ids / out_scores + i * stride_k (stride_k >= every top_k),
 * out_n[i], out_count[i] — each identical to what orama_post_search returns for that query. */
typedef struct {
    const orama_term_ref* refs;
    uint32_t n_refs;
    orama_bm25_params params;
} orama_post_query_desc;
int orama_post_search_batch(orama_post* p, const orama_post_query_desc* queries, uint32_t n_queries, float b,
                            const uint64_t* allow_bitmap, uint64_t bitmap_bits, int apply_omc, uint32_t max_parallel,
                            uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count);
/* Same, plus one status per query.  Queries of a batch are independent: a query that is malformed, outside the envelope
 * (ORAMA_ERR_UNSUPPORTED) or invalidated by a rebuild gets its own status and out_n = 0, every other query is answered.
 * Both forms return the first failing query's status (message: orama_last_error) and still complete the rest. */
int orama_post_search_batch_status(orama_post* p, const orama_post_query_desc* queries, uint32_t n_queries, float b,
                                   const uint64_t* allow_bitmap, uint64_t bitmap_bits, int apply_omc, uint32_t max_parallel,
                                   uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                                   uint64_t* out_count, int* out_status);

/* Request batcher for full-text searches (the BM25 side of orama_batcher_*).  The reference serves
 * every request alone (search_full_text is reached from many tokio workers, src/collection_manager/sides/read/
 * collection.rs:846-884); the range-partitioned scorer takes 32 queries per set of launches, so concurrent callers of
 * orama_post_batcher_search are coalesced into orama_post_search_batch calls — grouped by the arguments a batch shares:
 * (allow_bitmap, bitmap_bits, b, apply_omc).  Same arguments, results and errors as orama_post_search; the call blocks
 * until the request's batch has been scored.  max_batch in [1, 4096]; max_wait_us = 0: no artificial delay (batches
 * form while the previous one occupies the GPU).  Destroy the batcher before its store. */
typedef struct orama_post_batcher orama_post_batcher;
int orama_post_batcher_create(orama_post* p, uint32_t max_batch, uint32_t max_wait_us, orama_post_batcher** out);
void orama_post_batcher_destroy(orama_post_batcher* b);
int orama_post_batcher_search(orama_post_batcher* b, const orama_term_ref* refs, uint32_t n_refs, float bm25_b,
                              const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                              int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count);
int orama_post_batcher_stats(orama_post_batcher* b, uint64_t* requests, uint64_t* batches, uint32_t* largest_batch);

/* ------------------------------------------------------------------ score map, facets, groups
 * The reference hands the WHOLE HashMap<DocumentId, f32> of a search to facets and groups
 * (src/collection_manager/sides/read/search.rs:355-400 -> index/facet.rs:35-209, index/group.rs:107-170,
 * sort.rs:129-230).  orama_post_search_scores is orama_post_search / orama_post_search_hybrid (hybrid != 0) that ALSO
 * keeps that map resident: the handle owns the scratch set the scorer wrote (candidate list + position index) and a
 * read lock on the store until orama_scores_destroy — destroy it before the next orama_post_build / append.
 * Calls on one handle are serialised; different handles run concurrently.
 *
 *   orama_scores_count   token_score_results.len() (search.rs:482)
 *   orama_scores_export  every (DocumentId, score) entry, any size (what BM25Scorer::get_scores returns)
 *   orama_scores_lookup  token_scores.get(doc) for a list of ids (present[i] = contains_key)
 *
 * A facet FIELD is the resident image of one filter field of the index (BoolFieldStorage / StringFilterFieldStorage /
 * NumberFieldStorage, index/{bool,string_filter,number}_field.rs): its DocumentIds are resolved to the index's local
 * doc order once, at creation (ids the index does not hold are dropped — they can never be keys of a score map); it
 * goes stale when the index is rebuilt (ORAMA_ERR_INVALID from the calls below).
 *   buckets : bucket b = the ids `storage.filter(value_b)` yields — bool: {true, false}; string filter: one bucket per
 *             key; group-by: one bucket per value combination (the intersections GroupContext::execute builds,
 *             group.rs:134-166, are independent of the query — the shim materialises them once per commit)
 *   numbers : one (doc, value) entry per stored number (i32 / f32 widened to f64, exactly representable)
 *
 *   orama_facet_count         counts[b] = |bucket b ∩ keys(map)|                  bool_field.rs:182-208,
 *                                                                                   string_filter_field.rs:175-193
 *   orama_facet_count_ranges  counts[r] = entries with from[r] <= v <= to[r] whose doc is a key of the map
 *                                                                                   number_field.rs:368-387 (Between is
 *                                                                                   inclusive, :604-631)
 *   orama_group_top           per bucket the best `max_results` docs by score among keys of the map with a non-NaN
 *                             score (sort.rs:203-213; DECLARED tie rule: score desc, DocumentId asc).
 *                             out_ids / out_scores: n_buckets x max_results, out_n: n_buckets; max_results <= 1024.
 * The "facets are computed on the UNFILTERED score map when the request has filters" rule (search.rs:361-396) is the
 * caller's: it runs the second search with only the NOT-deleted bitmap and passes that handle. */
typedef struct orama_scores orama_scores;
typedef struct orama_facet_field orama_facet_field;
int orama_post_search_scores(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                             const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                             int hybrid, const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, int apply_omc,
                             uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count,
                             orama_scores** out_map);
void orama_scores_destroy(orama_scores* sm);
int orama_scores_count(orama_scores* sm, uint64_t* out);
int orama_scores_export(orama_scores* sm, uint64_t capacity, uint64_t* out_ids, float* out_scores, uint64_t* out_n);
int orama_scores_lookup(orama_scores* sm, const uint64_t* doc_ids, uint32_t n, float* out_scores, uint8_t* out_present);
int orama_facet_field_create_buckets(orama_post* index, const uint64_t* bucket_off, const uint64_t* bucket_docs,
                                     uint32_t n_buckets, orama_facet_field** out);
int orama_facet_field_create_numbers(orama_post* index, const uint64_t* docs, const double* values, uint64_t n,
                                     orama_facet_field** out);
void orama_facet_field_destroy(orama_facet_field* f);
int orama_facet_count(orama_scores* sm, orama_facet_field* f, uint64_t* out_counts);
int orama_facet_count_ranges(orama_scores* sm, orama_facet_field* f, const double* from, const double* to,
                             uint32_t n_ranges, uint64_t* out_counts);
int orama_group_top(orama_scores* sm, orama_facet_field* f, uint32_t max_results, uint64_t* out_ids, float* out_scores,
                    uint32_t* out_n);

/* ------------------------------------------------------------------ term-dictionary expansion
 * The dictionary step of collect_contributions (third-party in the reference: an FST walked with a prefix /
 * Levenshtein automaton).  Semantics as the reference's tests show them: `exact` -> the term itself; otherwise
 * every term the token is a PREFIX of (src/tests/fulltext_search.rs:603-753) plus, with `tolerance` > 0, every term
 * within that Levenshtein distance (:956-1018).  The sorted terms of one field live in HBM (blob + offsets[n+1],
 * strictly ascending byte-wise); orama_dict_expand scans all of them in one kernel and returns the matching term
 * indexes in ascending order.  out_n = number of matches; when it exceeds `capacity` only an arbitrary subset of
 * `capacity` matches was written — retry with a larger buffer.  Distances are over bytes; tokens <= 64 bytes. */
typedef struct orama_dict orama_dict;
int orama_dict_create(orama_ctx* ctx, const uint8_t* blob, const uint32_t* offsets, uint32_t n_terms,
                      orama_dict** out);
void orama_dict_destroy(orama_dict* d);
int orama_dict_expand(orama_dict* d, const uint8_t* token, uint32_t token_len, int exact, uint32_t tolerance,
                      uint32_t capacity, uint32_t* out_terms, uint32_t* out_n);

/* ------------------------------------------------------------------ hybrid
 * search_hybrid + normalize_and_combine — token_score.rs:357-422, then OMC + count + top-k.
 * vec_doc/vec_score: the vector result MAP after the a2 epilogue (<= limit entries, any order,
 * unique docs).  Full-text side as orama_post_search.  min/max fold from 0.0 over both maps;
 * out = fulltext' (+ vector'), max == min yields NaN scores which top-k drops (sort.rs:264-268)
 * while `out_count` still counts them (search.rs:482). */
int orama_post_search_hybrid(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                             const orama_bm25_params* params, const uint64_t* allow_bitmap,
                             uint64_t bitmap_bits, const uint64_t* vec_doc, const float* vec_score,
                             uint32_t n_vec, int apply_omc, uint64_t* out_ids, float* out_scores,
                             uint32_t* out_n, uint64_t* out_count);

/* search_hybrid as ONE call (token_score.rs:357-387): the vector leg (scan + top-`limit` rows of `query` in `v`)
 * and the full-text leg over `p` run concurrently on two HIP streams; the in-tree epilogue of
 * EmbeddingFieldStorage::search (embedding_field.rs:268-276: similarity = 1 - distance, Model::rescale_score when
 * rescale_e5 != 0, `>= min_similarity` cut-off, per-document sum) is applied to the <= limit hits, then
 * normalize_and_combine + OMC + count + top-k run on the device. `v` and `p` must belong to the same context. */
int orama_hybrid_search(orama_vec* v, orama_post* p, const float* query, uint32_t limit, float min_similarity,
                        int rescale_e5, const orama_term_ref* refs, uint32_t n_refs, float b,
                        const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                        int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count);

/* ------------------------------------------------------------------ one index sharded over several GPUs
 * shard g of an index holds the documents of one doc-id range — its rows in an orama_vec and its
 * postings in an orama_post built from that range only (avg_field_len = the index-wide averages).  A query is the
 * single-GPU search_full_text / search_hybrid (token_score.rs:186-387) cut at the three points where the
 * reference reads index-wide quantities, with the caller running the collective (RCCL over xGMI) in between:
 *
 *   begin  : K3 accumulate over the local postings           d_df[n_tokens] (i32)  <- local df per token
 *            -- all-reduce SUM of d_df:  corpus_docs.len() is index-wide (token_score.rs:262-275) --
 *   score  : idf from the GLOBAL df and params->total_documents (the index's document_count,
 *            token_score.rs:221), K3 finalise                 d_minmax[2] (i64)      <- hybrid only
 *            -- hybrid: all-reduce MAX of d_minmax: min/max fold over the whole map (token_score.rs:398-401) --
 *   finish : [hybrid: K5 with the reduced min/max; vector hits whose document lives on another shard are
 *            skipped here and combined there] OMC, K4 local top-k
 *                                                            d_block <- [k u64 ids][k f32 scores][pad][u64 count]
 *            -- all-gather of the blocks, then orama_post_merge_blocks_device on every rank --
 *
 * Every stage is enqueued on `hip_stream` (the stream the collectives run on); only `score` touches the host
 * (n_tokens log1pf).  df_global is a HOST array: the reduced d_df downloaded by the caller.  The query object
 * holds the store's shared lock and one scratch set; all calls for one query must come from the same thread and
 * end with orama_post_query_end (which waits for the stream).  Results are bit-identical to the single-store
 * search over the union of the shards (tests/test_sharded_fulltext_gpu.py). */
typedef struct orama_post_query orama_post_query;
uint64_t orama_post_block_bytes(uint32_t top_k);
int orama_post_query_begin(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                           const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                           int hybrid, int apply_omc, uint32_t n_vec_cap, void* hip_stream, int32_t* d_df,
                           orama_post_query** out);
int orama_post_query_score(orama_post_query* q, const uint32_t* df_global, int64_t* d_minmax);
int orama_post_query_finish(orama_post_query* q, const int64_t* d_minmax_global, const uint64_t* vec_doc,
                            const float* vec_score, uint32_t n_vec, void* d_block);
void orama_post_query_end(orama_post_query* q);
/* K6 over `lists` gathered blocks: global top-k (score desc, DocumentId asc) and the summed match count. */
int orama_post_merge_blocks_device(orama_ctx* ctx, const void* d_blocks, uint32_t lists, uint32_t top_k,
                                   uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_n,
                                   uint64_t* d_out_count, void* hip_stream);
/* Replace the per-field average lengths (index-wide averages for the shards of one index). */
int orama_post_set_avg_len(orama_post* p, const float* avg_field_len, uint32_t n_fields);

/* ------------------------------------------------------------------ sharded index, exchange inside the library
 * The staged entry points above leave the collectives to the caller.  A shard GROUP owns them: RCCL communicators
 * (loaded with dlopen on first use), one exchange stream and the gathered buffers per local shard, so that a sharded
 * search is ONE call from the reference's single Rust process (ReadSide, src/collection_manager/sides/read/mod.rs:
 * 621-738) — no torch, no second runtime.  Deployments:
 *   orama_shard_group_create(devices[n], n, flags)  one process, n shards:
 *       devices all distinct  -> one RCCL communicator per GPU (ncclCommInitAll), collectives over xGMI;
 *       devices all the same  -> the shards share one GPU: no communicator, every shard writes its block into the
 *                                same gathered buffer and the reductions are device-local kernels (what a 1-GPU box
 *                                and the parity tests run); ORAMA_SHARD_FORCE_RCCL with n == 1 builds a 1-rank
 *                                communicator anyway (plumbing test of the RCCL path on one GPU);
 *   orama_shard_group_create_rank(id, rank, world, device)  one process per GPU (bench.py under
 *       torch.distributed.run): rank 0 makes the 128-byte id with orama_shard_unique_id, the launcher carries it.
 * orama_shard_group_ctx(g, i) is the context of local shard i: create its orama_vec / orama_post with it (shard r of
 * `world` holds one contiguous DocumentId range).  Calls on one group are serialised (collectives must be
 * issued in the same order on every rank); in the one-process-per-GPU form every rank makes the same calls.
 * Index-wide quantities and where they travel: df[n_tokens] all-reduce SUM (token_score.rs:262-275), hybrid
 * {max, min} all-reduce MAX (token_score.rs:398-401), candidates + count all-gather then K6 (sort.rs:260-279,
 * search.rs:482); idf comes from the host libm after ONE pinned read-back of 4*n_tokens bytes.  Results are
 * bit-identical to the single-store search over the union of the shards. */
typedef struct orama_shard_group orama_shard_group;
#define ORAMA_SHARD_FORCE_RCCL 1u
int orama_shard_unique_id(void* out_id128);
int orama_shard_group_create(const int* devices, uint32_t n_shards, uint32_t flags, orama_shard_group** out);
int orama_shard_group_create_rank(const void* id128, int rank, int world, int device, orama_shard_group** out);
void orama_shard_group_destroy(orama_shard_group* g);
orama_ctx* orama_shard_group_ctx(orama_shard_group* g, uint32_t local_shard);
int orama_shard_group_info(orama_shard_group* g, uint32_t* world, uint32_t* n_local_shards, uint32_t* first_rank,
                           int* uses_rccl);
/* Barrier over all shards of the group (local devices drained + one all-reduced word). */
int orama_shard_group_barrier(orama_shard_group* g);
/* max over all ranks of one host double (the slowest rank's elapsed time in bench.py). */
int orama_shard_group_allreduce_max_f64(orama_shard_group* g, double* inout);

/* EmbeddingStorage::search over the row shards of one field (call site embedding_field.rs:255-266): `shards` holds the
 * local shards in rank order; queries / outputs as orama_vec_search (host buffers, q x k).  allow_bitmaps: NULL, or
 * one RESIDENT bitmap token per local shard (orama_allow_token; each lives on its shard's device). */
int orama_shard_vec_search(orama_shard_group* g, orama_vec* const* shards, const float* queries, uint32_t q, uint32_t k,
                           const uint64_t* const* allow_bitmaps, uint64_t bitmap_bits, uint64_t* out_ids,
                           float* out_dist, uint32_t* out_n);
/* search_full_text (hybrid = 0) / the full-text leg of search_hybrid with the GLOBAL vector map (hybrid = 1) over the
 * document shards of one index — token_score.rs:186-303, 357-422.  Every shard store was built from its doc-id range
 * with the SAME list numbering and the index-wide field averages; params->total_documents is the index-wide N;
 * params->top_k >= 1.  Outputs as orama_post_search. */
int orama_shard_post_search(orama_shard_group* g, orama_post* const* shards, const orama_term_ref* refs, uint32_t n_refs,
                            float b, const orama_bm25_params* params, const uint64_t* const* allow_bitmaps,
                            uint64_t bitmap_bits, int apply_omc, int hybrid, const uint64_t* vec_doc,
                            const float* vec_score, uint32_t n_vec, uint64_t* out_ids, float* out_scores,
                            uint32_t* out_n, uint64_t* out_count);
/* search_hybrid over a sharded index in one call: sharded vector leg, the in-tree epilogue of
 * EmbeddingFieldStorage::search on the <= limit global hits (embedding_field.rs:268-276), sharded full-text leg. */
int orama_shard_hybrid_search(orama_shard_group* g, orama_vec* const* vec_shards, orama_post* const* post_shards,
                              const float* query, uint32_t limit, float min_similarity, int rescale_e5,
                              const orama_term_ref* refs, uint32_t n_refs, float b, const orama_bm25_params* params,
                              const uint64_t* const* allow_bitmaps, uint64_t bitmap_bits, int apply_omc,
                              uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count);

/* Concurrency: a group whose shards all live in this process (both orama_shard_group_create forms) serves up to
 * ORAMA_SHARD_LANES (default 8) sharded calls at a time — each on its own exchange streams and buffers; only the issue
 * of a collective is serialised.  A one-process-per-rank group serves one call at a time (the callers of different
 * processes could order concurrent calls differently).  Matches the re-entrancy `search(&self)` assumes
 * (src/collection_manager/sides/read/collection.rs:846-884).
 * The request micro-batcher in front of a group: like orama_batcher_create, every pass is one orama_shard_vec_search
 * over `shards` (the group's local shards, in rank order); orama_batcher_search_filtered then takes, in place of the
 * bitmap words, the address of the caller's array of resident per-shard tokens (const uint64_t* const*; requests
 * carrying the same array share a pass).  Every shard of the group must live in this process (ORAMA_ERR_UNSUPPORTED
 * otherwise: processes would form different batches).  The group and the stores must outlive the batcher. */
/* orama_post_search_batch_status over the document shards of one index (`shards`: the group's local shards in rank order,
 * built as for orama_shard_post_search) — for every shape of group: all shards in this process or one process per rank
 * (every rank then makes the same call: same queries, same order).  Per block of <= 512 queries: every shard's df of every
 * token (list lengths where exact; under a filter or with several lists per token the range scorer's counting launch), ONE
 * all-reduce sums them over the index (corpus_docs.len(), token_score.rs:262-275); every shard scores the block with the range
 * scorer and the index-wide idf; ONE all-gather moves the shards' top-k blocks and every rank merges them by (score desc,
 * DocumentId asc) and sums the counts (sort.rs:260-279, search.rs:482).  Queries the range scorer does not take (more than 64
 * non-empty lists) are answered afterwards by orama_shard_post_search, one at a time.  A systemic failure on any shard
 * (ORAMA_ERR_BUSY, _HIP, _OOM) ends the call with that status for every query on every rank.  allow_bitmaps: NULL or one
 * resident token per local shard.  Results per query as orama_post_search over the union of the shards, bit for bit;
 * out_status may be NULL. */
int orama_shard_post_search_batch(orama_shard_group* g, orama_post* const* shards, const orama_post_query_desc* queries,
                                  uint32_t n_queries, float b, const uint64_t* const* allow_bitmaps, uint64_t bitmap_bits,
                                  int apply_omc, uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                                  uint64_t* out_count, int* out_status);
/* The full-text request batcher in front of a group: like orama_post_batcher_create, every dispatch is one
 * orama_shard_post_search_batch.  A request's filter, if any, is the address of the caller's array of resident per-shard
 * tokens.  Every shard of the group must live in this process.  The group and the stores must outlive the batcher. */
int orama_post_batcher_create_group(orama_shard_group* g, orama_post* const* shards, uint32_t max_batch, uint32_t max_wait_us,
                                    orama_post_batcher** out);
/* Sharded calls the group can run side by side, and how many lanes (exchange streams + buffers per local shard) it has
 * made so far. */
int orama_shard_group_lanes(orama_shard_group* g, uint32_t* max_lanes, uint32_t* lanes_created);
int orama_batcher_create_group(orama_shard_group* g, orama_vec* const* shards, uint32_t max_batch, uint32_t max_wait_us,
                               orama_batcher** out);


/* Pipelined vector-search session (serving loop / bench.py): `n_queries` queries resident in HBM on every local
 * device; orama_shard_session_step(s, i) enqueues step i — queries [i*q, (i+1)*q) modulo the resident set — without
 * host synchronisation: corpus scans of consecutive steps run back to back on ONE scan stream per device, the
 * launch-bound tail (K4, all-gather, K6) on `n_slots` high-priority streams used round-robin, so the tail of step i
 * overlaps the scan of step i+1.  Results stay in HBM; orama_shard_session_result reads the last step of a slot after
 * orama_shard_session_sync.  With one shard and force_exchange == 0 no exchange runs (the block is the answer). */
typedef struct orama_shard_session orama_shard_session;
int orama_shard_session_create(orama_shard_group* g, orama_vec* const* shards, const float* queries, uint32_t n_queries,
                               uint32_t q_per_step, uint32_t k, uint32_t n_slots, int force_exchange,
                               orama_shard_session** out);
void orama_shard_session_destroy(orama_shard_session* s);
int orama_shard_session_step(orama_shard_session* s, uint32_t step);
int orama_shard_session_sync(orama_shard_session* s);
int orama_shard_session_result(orama_shard_session* s, uint32_t slot, uint64_t* out_ids, float* out_dist, uint32_t* out_n);

/* Standalone normalize_and_combine on host-provided maps (both sides small or large) — the
 * literal replacement of token_score.rs:393-422 + top_n for callers that keep seam (i). */
int orama_hybrid_combine(orama_ctx* ctx, const uint64_t* vec_doc, const float* vec_score,
                         uint64_t n_vec, const uint64_t* ft_doc, const float* ft_score,
                         uint64_t n_ft, uint32_t top_k, uint64_t* out_ids, float* out_scores,
                         uint32_t* out_n, uint64_t* out_count);

/* Reciprocal-rank fusion — an EXTRA next to the parity path above (the reference merges by min-max + sum; RRF is
 * what north_star names).  Each list is cut to its best `depth` entries (score desc, DocumentId asc; ranks from 1)
 * and score[doc] = sum 1 / (rrf_k + rank) over the lists holding doc (f32, full-text term first).  out_count =
 * documents in the union of the two cut lists.  rrf_k is conventionally 60. */
int orama_hybrid_rrf(orama_ctx* ctx, const uint64_t* vec_doc, const float* vec_score, uint64_t n_vec,
                     const uint64_t* ft_doc, const float* ft_score, uint64_t n_ft, float rrf_k, uint32_t depth,
                     uint32_t top_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count);

/* ------------------------------------------------------------------ top-n
 * top_n — sort.rs:260-279 over an explicit (doc, score) list: NaN dropped, score desc, doc asc.
 * (+0.0 and -0.0 tie, as NotNan<f32> compares them; a -0.0 score is returned as +0.0.) */
int orama_top_n(orama_ctx* ctx, const uint64_t* doc, const float* score, uint64_t n, uint32_t top_k,
                uint64_t* out_ids, float* out_scores, uint32_t* out_n);

#ifdef __cplusplus
}
#endif
#endif /* ORAMA_HIP_H */
