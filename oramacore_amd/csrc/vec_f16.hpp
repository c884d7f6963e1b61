// vec_f16.hpp — K2: fp16 corpus in an MFMA-fragment-tiled HBM layout + batched-query scan on the
// matrix cores with a fused per-query threshold filter (build-side extension: SURVEY F4, configs C3/C5).
#pragma once

#include "common.hpp"

namespace orama {

// HBM layout of the fp16 corpus: [row_tile = row/32][k_step = k/16][lane 0..63][8 halves], where lane
// l = (kgrp << 5) | (row & 31) holds elements k = 16*k_step + 8*kgrp + 0..7 of its row — exactly the
// A-operand fragment of v_mfma_f32_32x32x16_f16, so every wave-wide 16 B/lane load is one contiguous
// 1 KiB and a row tile (32 rows) is one contiguous run of kpad*64 bytes.  kpad = dim rounded up to 128.
inline uint32_t f16_kpad(uint32_t dim) { return (dim + 127u) & ~127u; }
// Bytes between consecutive row tiles: 32 rows x kpad x 2 B of fragments + an optional skew pad
// (ORAMA_F16_TILE_PAD, bytes).  The pad was an experiment against HBM channel camping (all tiles start at
// multiples of 48 KiB at dim 768); it measured neutral on MI355X, so the default is 0.
uint64_t f16_tile_pad();
inline uint64_t f16_tile_bytes(uint32_t dim) { return (uint64_t)f16_kpad(dim) * 64u + f16_tile_pad(); }
inline uint64_t f16_tiles(uint64_t rows) { return (rows + 31) / 32; }

constexpr uint32_t kF16MaxQ = 64;  // queries per corpus pass (2 MFMA column tiles); larger batches loop
constexpr uint32_t kF16MetaBytes = 256;  // tile metadata record in LDS: 32 norms, the tombstone word, padding
// LDS of the K2 kernel: the batch's B fragments, 1/|q|, and per wave (8 of them) a ring of <= 5 metadata records, a
// 64-bin histogram and a staging area of `stage_entries` passing rows (12 bytes each) on their way to the candidate lists
constexpr size_t kF16LdsLimit = 160 * 1024;
inline size_t vec_scan_f16_lds_bytes(uint32_t dim, int nqt, uint32_t stage_entries) {
    return (size_t)(f16_kpad(dim) / 16) * (size_t)nqt * 1024 + 64 * sizeof(float) + 8 * (5 * kF16MetaBytes + 256) +
           8 * 3 * (size_t)stage_entries * sizeof(uint32_t);
}
// the staging area takes the LDS the fragments leave, in steps of 64 entries, 128..1024 (0: not even 128 fit)
inline uint32_t vec_scan_f16_stage_entries(uint32_t dim, int nqt) {
    const size_t fixed = vec_scan_f16_lds_bytes(dim, nqt, 0);
    if (fixed >= kF16LdsLimit) return 0;
    const size_t e = ((kF16LdsLimit - fixed) / (8 * 3 * sizeof(uint32_t))) & ~(size_t)63;
    return e < 128 ? 0u : (uint32_t)(e > 1024 ? 1024 : e);
}
// queries one K2 pass can take at this dimension: 64 while two column tiles of fragments fit the 160 KiB of LDS
inline uint32_t vec_scan_f16_max_q(uint32_t dim) { return vec_scan_f16_stage_entries(dim, 2) ? kF16MaxQ : 32; }

// rows [first, first+n) of `src` (f32 row-major [n][dim]) → fp16 (RNE) into the tiled store.
int launch_f16_store_rows(void* tiled, const float* src, uint64_t first, uint64_t n, uint32_t dim,
                          hipStream_t stream);
// 1/|x| (cosine) or |x|^2 (L2) of the STORED (fp16-rounded) rows, accumulated in f32.
int launch_f16_inv_norm(const void* tiled, uint64_t first, uint64_t n, uint32_t dim, float* inv_norm,
                        hipStream_t stream, int metric = ORAMA_METRIC_COSINE);
// out[i] = row row_idx[i] converted back to f32.
int launch_f16_gather_rows(const void* tiled, const uint64_t* d_row_idx, uint64_t n, uint32_t dim,
                           float* d_out, hipStream_t stream);
// zero the padding rows of the last tile / whole tiles in [first_row, cap_rows)
int launch_f16_zero_rows(void* tiled, uint64_t first_row, uint64_t end_row, uint32_t dim, hipStream_t stream);

struct F16ScanArgs {
    const void* tiled = nullptr;
    const float* inv_norm = nullptr;
    const float* queries = nullptr;  // q x dim f32 (HBM); converted to fp16 fragments in LDS by every block
    uint32_t q = 0;                  // 1..64
    uint32_t dim = 0;
    // ORAMA_METRIC_COSINE: inv_norm[] holds 1/|x| and the distance is 1 - s/(|x||q|);  ORAMA_METRIC_L2SQ: inv_norm[]
    // holds |x|^2 (of the stored, fp16-rounded row) and the distance is (|q|^2 + |x|^2) - 2 s
    int metric = ORAMA_METRIC_COSINE;
    uint64_t n_rows = 0;             // rows in the store
    uint64_t row_begin = 0, row_end = 0;  // rows scanned by this launch (row_begin % 32 == 0)
    const uint64_t* row_doc = nullptr;
    const uint32_t* dead = nullptr;
    const uint64_t* allow = nullptr;
    uint64_t allow_bits = 0;
    // dense mode: out_dense[j * dense_stride + (row - row_begin)] = distance (NaN when excluded)
    float* out_dense = nullptr;
    uint64_t dense_stride = 0;
    // filter mode: rows with distance < tau[j] are appended to (cand_dist, cand_row)[j * cand_stride + pos]
    const float* tau = nullptr;
    float* cand_dist = nullptr;
    uint32_t* cand_row = nullptr;
    uint32_t* cand_count = nullptr;  // q counters (pre-seeded by the caller)
    uint64_t cand_stride = 0;
    // the caller promises no bit-identity with batched answers (the shadow scan of the two-stage plan): <= 4 queries take
    // K1h, the dot-product kernel without MFMA
    bool solo = false;
    // K1h fused mode (one query): every wave keeps its best min(topk, 64) rows as keys ~ordered(distance) << 32 | ~row and
    // writes them (0 = empty) to wave_lists[wave * 64 ...]; a wave that evicted a row writes the key of its worst kept
    // row to wave_thr[wave] (0 otherwise) — see launch_shadow_wave_check.  No dense output, no candidate lists.
    unsigned long long* wave_lists = nullptr;
    unsigned long long* wave_thr = nullptr;
    uint32_t topk = 0;
    uint32_t stage_cap = 0;  // set by the launcher: vec_scan_f16_stage_entries
    uint32_t dbg = 0;  // timing ablations (ORAMA_K2_DBG): 1 no epilogue (K2), 2 no candidate appends (K2, K2d)
};
// K2. Algorithmic HBM traffic: (row_end - row_begin) * kpad * 2 bytes per launch (serves all q queries).
int launch_vec_scan_f16(orama_ctx* ctx, const F16ScanArgs& a, hipStream_t stream);

// Waves launch_vec_scan_f16 uses in the fused mode for `a` (= number of 256-key lists written); 0 when the fused mode
// does not apply to these arguments.
constexpr uint32_t kF16WaveListKeys = 64;
constexpr uint32_t kSelectMaxKeysFused = 4096;  // topk the fused mode accepts (the merged result is cut there)
uint32_t vec_scan_f16_fused_waves(orama_ctx* ctx, const F16ScanArgs& a);
// flag[0] = 1 when rows evicted by a wave could belong to the k best of the merged result (d_out_dist ascending, d_out_n[0])
int launch_shadow_wave_check(const unsigned long long* d_wave_thr, uint32_t waves, const float* d_out_dist, const uint32_t* d_out_n,
                             uint32_t k, uint32_t* d_flag, hipStream_t stream);

// K2c (vec_f16_wide.hip): the same scan for 65..256 queries per corpus pass — a register-blocked GEMM (block tile
// 256 rows x 256 queries, corpus and query fragments staged through an LDS double buffer).  `d_query_frags`
// (f16_wide_query_bytes(dim) bytes of HBM scratch) receives the fp16 query fragments + 1/|q| when `prepare` is set
// (first launch of a batch); later launches of the same batch reuse them.  Same F16ScanArgs semantics, a.q <= 256.
constexpr uint32_t kF16WideMaxQ = 256;
size_t f16_wide_query_bytes(uint32_t dim);
int launch_vec_scan_f16_wide(orama_ctx* ctx, const F16ScanArgs& a, void* d_query_frags, bool prepare,
                             hipStream_t stream);

// fp16 query fragments [query tile 0..7][k-step][lane][8 halves] + per-query 1/|q| (cosine) or |q|^2 (L2) of the
// fp16-rounded query into `d_query_frags` (f16_wide_query_bytes(dim) bytes) — shared by K2c and K2d.
int launch_f16_prepare_queries(const float* d_queries, uint32_t q, uint32_t dim, int metric, void* d_query_frags,
                               hipStream_t stream);
// K2d (vec_f16_pc.hip): the wide-batch scan as a producer/consumer kernel — dedicated loader waves feed the LDS ring,
// consumer waves only read fragments and issue MFMAs.  Same arguments as K2c; the fragments must have been prepared.
// geometry 1: 12 consumers of 2 x 2 MFMA tiles + 4 loaders (192 rows x 256 queries per block tile) — the default;
// 2: 8 consumers of 3 x 2 tiles + 4 loaders.
int launch_vec_scan_f16_pc(orama_ctx* ctx, const F16ScanArgs& a, void* d_query_frags, hipStream_t stream, int geometry);

// K2q (vec_f16_qs.hip): the wide-batch scan with the queries stationary in registers — every wave keeps ONE 32-query
// tile's B fragments for all k-steps in VGPRs, LDS holds only the corpus ring, the global->LDS traffic is the corpus once.
// kpad <= 768 and 129..256 queries (vec_scan_f16_qs_supports); same arguments as K2d, the fragments must have been prepared.
bool vec_scan_f16_qs_supports(uint32_t dim, uint32_t q);
int launch_vec_scan_f16_qs(orama_ctx* ctx, const F16ScanArgs& a, void* d_query_frags, hipStream_t stream);

// K2h (vec_f16_kh.hip): queries stationary in registers, two query tiles per wave, the K loop of a tile split over two waves
// that hand the accumulators on through LDS (one accumulator chain per output element, as everywhere else).  kpad a
// multiple of 256 and <= 768, 129..256 queries (vec_scan_f16_kh_supports); same arguments as K2d.
bool vec_scan_f16_kh_supports(uint32_t dim, uint32_t q);
int launch_vec_scan_f16_kh(orama_ctx* ctx, const F16ScanArgs& a, void* d_query_frags, hipStream_t stream);

// tau[j] = k-th best distance of list j when the list is full, else +inf; and seed the candidate lists
// with the current best entries: cand[j][0..n_j) = (dist, row), cand_count[j] = n_j.
int launch_f16_seed_candidates(const float* best_dist, const uint32_t* best_row, const uint32_t* best_n,
                               uint32_t q, uint32_t k, float* tau, float* cand_dist, uint32_t* cand_row,
                               uint32_t* cand_count, uint64_t cand_stride, hipStream_t stream, const float* tau_cap = nullptr);
// the ceiling experiment of a threshold shared between shards (ORAMA_F16_TAU_ORACLE=1): cap[j] = query j's final k-th distance, one ulp up
int launch_f16_remember_kth(const float* out_dist, const uint32_t* out_n, uint32_t k, uint32_t q, float* cap, hipStream_t stream);

}  // namespace orama
