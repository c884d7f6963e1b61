// select.hpp — K4: top-k selection over dense or candidate score arrays (device side of
// `top_n`, src/collection_manager/sides/read/sort.rs:260-279, and of the k-NN cut in the
// third-party EmbeddingStorage::search).
#pragma once

#include "common.hpp"

namespace orama {

constexpr uint32_t kSelectMaxK = 4096;  // LDS-resident final sort (16 B x 4096 = 64 KiB)

// Per-query selection state in HBM (zeroed by launch_select with one hipMemsetAsync).
struct SelectState {
    uint32_t hist[6][2048];
    unsigned long long prefix;  // selected high bits of the 64-bit composite key so far
    uint32_t remaining;         // how many elements must still be taken inside `prefix`
    uint32_t done;              // selection threshold fully determined
    uint32_t sel_shift;         // select  (key >> sel_shift) >= prefix
    uint32_t valid;             // non-NaN elements
    uint32_t kprime;            // min(k, valid)
    uint32_t out_count;         // collect cursor
    uint32_t pad;
};

struct SelectPlan {
    // input: q independent lists. List i starts at vals + i*stride (and idx + i*stride).
    const float* vals = nullptr;
    const uint32_t* idx = nullptr;    // nullable: implicit position index
    uint64_t stride = 0;
    const uint32_t* n_dev = nullptr;  // nullable: per-list length in HBM (<= n)
    uint32_t n = 0;                   // (max) list length
    uint32_t n_hint = 0;              // expected length when n_dev is set (grid sizing only; 0 = n)
    uint32_t q = 1;
    uint32_t k = 0;
    bool descending = true;           // true: larger value is better (scores); false: distances
    const uint64_t* id_map = nullptr; // nullable: idx -> 64-bit id used for tie order + out_ids
    // scratch (HBM)
    SelectState* state = nullptr;     // q entries
    unsigned long long* keys = nullptr;  // q x k
    uint64_t keys_capacity = 0;          // keys `keys` can hold; >= q x 4096 enables the two-launch form for short lists
    // output (HBM): q x k each, out_n: q
    uint32_t* out_idx = nullptr;      // nullable
    uint64_t* out_ids = nullptr;      // nullable (id_map[idx] or idx)
    float* out_val = nullptr;
    uint32_t* out_n = nullptr;
};

// Enqueue the selection on `stream`. Order of the result: value (best first), then id asc, then
// idx asc. NaN values are never selected. Requires 1 <= k <= kSelectMaxK.
int launch_select(orama_ctx* ctx, const SelectPlan& plan, hipStream_t stream);

// K6: merge `lists` candidate lists of k (id, dist) pairs per query ([list][query][k] layout);
// entries with id == UINT64_MAX are padding. Result per query: best k by (dist asc, id asc).
int launch_merge_candidates(orama_ctx* ctx, const uint64_t* d_ids, const float* d_dist,
                            uint32_t lists, uint32_t q, uint32_t k, uint64_t* d_out_ids,
                            float* d_out_dist, uint32_t* d_out_n, hipStream_t stream);

// The launchers of select_wide.hip's two kernels (round 5; called by launch_select): <= 512 values of every list in one
// workgroup without a sort; `parts` workgroups per dense list, each taking its <= kSelectWidePartValues values in one round and
// writing its best k keys to p.keys (launch_select orders them with the final kernel).
constexpr uint32_t kSelectWidePartValues = 32u * 1024u;
int launch_select_tiny(const SelectPlan& p, hipStream_t stream);
int launch_pairs_reduce_wide(const SelectPlan& p, uint32_t parts, bool force_narrow, hipStream_t stream);

// Top-k over lists of 64-bit composite keys (value-key << 32 | ~idx, "larger wins", 0 = empty) — the output
// of the fused per-wave top-k of K1.  List i = keys + i*stride, n_keys entries.  Reduction: workgroups sort
// 8192-key chunks in LDS and keep their best k until <= 4096 keys remain, then one workgroup applies the
// final order (value, 64-bit id asc, idx asc).  d_tmp must hold keys_topk_scratch_keys() keys.  d_n_per_list (optional,
// device): list i holds only min(n_keys, d_n_per_list[i]) keys — what lies behind them is never read.  d_tau (optional):
// one ZEROED 64-bit word per list, tau_stride words apart — the running bound the chunks of a list share in the first
// reduction level (a chunk with at most k keys at or above the bound skips its selection).  d_n_active (optional, device):
// only lists 0 .. *d_n_active - 1 exist — the workgroups of the others end at once and their outputs are left untouched.
constexpr uint32_t kKeysChunk = 8192;
// Optional side job of the final launch (round 5): workgroup i copies words [i * words, (i + 1) * words) from `src` to `dst` before
// anything else — K3r hands its per-query result words to the pinned host block this way, and points the outputs there as well:
// the chunk's chain ends with the final launch instead of a read-back behind it.
struct KeysMirror {
    const uint32_t* src = nullptr;
    uint32_t* dst = nullptr;
    uint32_t words = 0;
};
int launch_keys_topk(orama_ctx* ctx, const unsigned long long* d_keys, uint32_t n_keys, uint64_t stride,
                     uint32_t q, uint32_t k, bool descending, const uint64_t* id_map,
                     unsigned long long* d_tmp, uint32_t* out_idx, uint64_t* out_ids, float* out_val,
                     uint32_t* out_n, hipStream_t stream, const uint32_t* d_n_per_list = nullptr,
                     unsigned long long* d_tau = nullptr, uint32_t tau_stride = 0, const uint32_t* d_n_active = nullptr,
                     bool counted = false, uint32_t n_per_list_stride = 1, const KeysMirror* mirror = nullptr);
// Keys of scratch launch_keys_topk needs for lists of n_keys entries.
uint64_t keys_topk_scratch_keys(uint32_t n_keys, uint32_t q, uint32_t k);

// Packed exchange block of one rank: [q*k u64 ids][q*k f32 distances], padded to 8 bytes.
uint64_t packed_block_bytes(uint32_t q, uint32_t k);
// K6 over `lists` consecutive packed blocks (the output of one all-gather).
int launch_merge_packed(orama_ctx* ctx, const void* d_packed, uint32_t lists, uint32_t q, uint32_t k,
                        uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n, hipStream_t stream);
// Same merge over blocks `block_stride` bytes apart (each starts with [q*k u64 ids][q*k f32 values]; trailing
// bytes are the caller's), values ascending (distances) or descending (scores).
int launch_merge_blocks(orama_ctx* ctx, const void* d_blocks, uint64_t block_stride, uint32_t lists, uint32_t q,
                        uint32_t k, bool descending, uint64_t* d_out_ids, float* d_out_val, uint32_t* d_out_n,
                        hipStream_t stream);

}  // namespace orama
