// vec_kernels.hpp — launchers of the vector-store kernels (K1 scan, row norms, synthetic fill,
// row gather).  Device side of oramacore_fields::embedding::EmbeddingStorage::search as called at
// src/collection_manager/sides/read/index/embedding_field.rs:255-266.
#pragma once

#include "common.hpp"

namespace orama {

// Initial tuning: built-in defaults overridden by ORAMA_SCAN_ROWS / ORAMA_SCAN_BLOCKS_PER_CU /
// ORAMA_SCAN_NT (read once).
ScanTuning default_scan_tuning();
bool scan_tuning_valid(const ScanTuning& t);

struct ScanArgs {
    const float* corpus = nullptr;    // n x dim f32 row-major, rows 16-B aligned when dim % 4 == 0
    const float* inv_norm = nullptr;  // n: 1/|x| (cosine only)
    const float* query = nullptr;     // dim f32 (HBM)
    uint64_t n = 0;
    uint32_t dim = 0;
    int metric = ORAMA_METRIC_COSINE;
    const uint64_t* row_doc = nullptr;   // n DocumentIds (needed when `allow` is set)
    const uint32_t* dead = nullptr;      // nullable bitmap over rows, bit set = tombstoned
    const uint64_t* allow = nullptr;     // nullable bitmap over doc ids
    uint64_t allow_bits = 0;
    float* out_dist = nullptr;           // dense mode: n distances; NaN for excluded rows
    // fused top-k mode (k <= 128): every wave keeps its own best-k of the rows it scans and writes them as
    // 128 composite keys (~ordered(distance) << 32 | ~row, 0 = empty) to wave_lists[wave * 128 ...]
    unsigned long long* wave_lists = nullptr;
    uint32_t topk = 0;
};

// K1: one corpus pass, one query.  Algorithmic HBM traffic: n * dim * 4 bytes.
int launch_vec_scan_f32(orama_ctx* ctx, const ScanArgs& a, hipStream_t stream, hipEvent_t done = nullptr, hipEvent_t* attached = nullptr);
// Number of waves launch_vec_scan_f32 will use for `a` (= number of 128-key lists written in fused mode).
uint32_t vec_scan_f32_waves(orama_ctx* ctx, const ScanArgs& a);
constexpr uint32_t kWaveListKeys = 128;

// K1 for queries chosen on the device: queries d_pick[0 .. *d_n_pick) of a.query (an array of queries), one corpus pass each,
// wave lists of the f-th at a.wave_lists + f * list_stride (vec_scan_f32_picked_waves() lists of 128 keys); fused mode,
// cosine, dimensions on the vectorised path.  *d_n_pick == 0 ends the launch at once.  Distances bit-identical to K1's.
bool vec_scan_f32_picked_supported(const ScanArgs& a);
uint32_t vec_scan_f32_picked_waves(orama_ctx* ctx, const ScanArgs& a);
int launch_vec_scan_f32_picked(orama_ctx* ctx, const ScanArgs& a, const uint32_t* d_pick, const uint32_t* d_n_pick,
                               uint64_t list_stride, hipStream_t stream);

// K1b: one corpus pass, nq in [2, 8] queries (a.query = nq contiguous queries); distances of query j go to
// a.out_dist[j * out_stride + row].  Bit-identical per (row, query) to launch_vec_scan_f32.  Dense mode only.
constexpr uint32_t kScanMultiMaxQ = 8;
bool vec_scan_f32_multi_supported(const ScanArgs& a);
int launch_vec_scan_f32_multi(orama_ctx* ctx, const ScanArgs& a, uint32_t nq, uint64_t out_stride,
                              hipStream_t stream);

// 1/|x| per row (cosine) for rows [first, first + n).
int launch_row_inv_norm_f32(const float* corpus, uint64_t first, uint64_t n, uint32_t dim,
                            float* inv_norm, hipStream_t stream);

// Synthetic rows generated in HBM: x = u * g/|g|, g ~ N(0,1), u ~ U(0.5, 2).
int launch_synth_fill_f32(float* corpus, uint64_t first, uint64_t n, uint32_t dim, uint64_t seed,
                          hipStream_t stream);

// out[i] = corpus[row_idx[i]]  (f32 rows).
int launch_gather_rows_f32(const float* corpus, const uint64_t* d_row_idx, uint64_t n, uint32_t dim,
                           float* d_out, hipStream_t stream);

// Two-stage exact search (fp32 rows + fp16 shadow): exact K1 distances of candidate rows, and the check that the
// candidate lists cannot miss a row of the exact top-k (vec_store.hip).
bool vec_rerank_f32_supported(uint32_t dim);
int launch_rerank_f32(const float* corpus, const float* inv_norm, uint32_t dim, const float* d_queries, uint32_t q,
                      const uint32_t* d_cand_rows, const uint32_t* d_cand_n, uint32_t stride, float* d_out_dist,
                      hipStream_t stream);
int launch_shadow_band(const float* d_shadow_dist, const uint32_t* d_n, uint32_t q, uint32_t k, uint32_t k1, float band,
                       uint32_t* d_flag, hipStream_t stream, const uint32_t* d_also = nullptr);

// ids[i] = first + i
int launch_iota_u64(uint64_t* d_ids, uint64_t n, uint64_t first, hipStream_t stream);

}  // namespace orama
