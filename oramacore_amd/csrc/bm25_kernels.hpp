// bm25_kernels.hpp — K3 (postings accumulate → BM25F finalise) and K5 (hybrid min-max combine).
// Device side of search_full_text / BM25Scorer / normalize_and_combine:
//   src/collection_manager/sides/read/index/token_score.rs:257-302, 393-422
//   src/collection_manager/bm25.rs:369-428, 484-524
#pragma once

#include "common.hpp"

namespace orama {

constexpr uint32_t kMaxTokens = 64;  // query tokens per search (mask bit = 1 << (t % 32), see below)

// One (token, posting list) reference processed by the accumulate kernel.
struct Bm25Seg {
    uint64_t post_begin;  // first posting of the list inside the postings arrays
    uint64_t virt_begin;  // first virtual index of this segment inside its launch
    uint32_t len;
    uint32_t token;
    float boost;   // SearchParams.boost of the field (resident mode)
    float inv_avg_unused;
    float avg_len; // field average length (resident mode)
    uint32_t pad;
};

// Query-scoped device state (zeroed per query with one memset).
struct Bm25State {
    uint32_t df[kMaxTokens];   // distinct docs per token (corpus_docs.len(), token_score.rs:262-275)
    uint32_t list_len;         // slots of the candidate list in use (host: one per posting; + appended vector docs)
    uint32_t cand_count;       // docs in the score map (== `count`, search.rs:482)
    uint32_t max_key;          // ordered(max score) over non-NaN candidates, 0 if none
    uint32_t min_key;          // ordered(min score), 0xffffffff if none
};

struct Bm25Accum {
    // postings: word0 = local doc index; word1 = (tf << 16 | field_len) [resident] or f32 ntf bits [PRE]
    const uint32_t* post_doc = nullptr;
    const uint32_t* post_val = nullptr;
    const Bm25Seg* segs = nullptr;  // segments of THIS launch (device)
    uint32_t n_segs = 0;
    uint64_t total = 0;             // virtual postings in this launch
    bool precomputed = false;       // post_val holds ntf
    float b = 0.75f;                // Bm25Params::default().b
    const uint64_t* docs = nullptr;   // local idx -> DocumentId (needed with `allow`)
    const uint64_t* allow = nullptr;  // nullable bitmap over DocumentIds
    uint64_t allow_bits = 0;
    uint32_t epoch = 0;
    uint64_t n_docs = 0;
    // Per-document record of `slots` 8-byte cells (slots = pow2 >= n_tokens + 1, one or a few cache lines):
    // cell t < n_tokens = {epoch:32 | S_t:f32}; the LAST cell's high word = epoch of the last query that touched
    // the doc.  A posting's accumulator update and its first-touch test hit the same line, and finalise reads a
    // document's token cells contiguously.
    unsigned long long* acc = nullptr;  // [n_docs][slots]
    uint32_t slots = 0;
    // touched[virt_base + v] = doc when posting v is the doc's first touch in this query, else 0xffffffff:
    // a slot per posting instead of a compacted list — no returning atomics on a shared cursor (one hot word
    // saturates at ~88 atomics/us on MI355X, which was the whole cost of this kernel)
    uint32_t* touched = nullptr;
    uint64_t virt_base = 0;             // first slot of this launch (launches of one query are laid end to end)
    Bm25State* state = nullptr;
};
int launch_bm25_accumulate(orama_ctx* ctx, const Bm25Accum& a, hipStream_t stream);

struct Bm25Finalize {
    uint32_t n_tokens = 0;
    float k = 1.2f;
    const float* idf_vals = nullptr;   // idf per token (device, n_tokens) — computed by the host libm
    bool use_threshold = false;
    uint32_t threshold = 0;
    bool track_minmax = false;         // hybrid: reduce min/max of the emitted scores into `state`
    const float* omc_dense = nullptr;  // nullable: multiplier per local doc (applied to the final score)
    uint32_t epoch = 0;
    uint64_t n_docs = 0;
    const unsigned long long* acc = nullptr;  // [n_docs][slots]
    uint32_t slots = 0;
    const uint32_t* touched = nullptr;
    uint32_t n_slots = 0;              // slots written by the accumulate launches (= postings referenced)
    Bm25State* state = nullptr;
    // outputs: the score map as a candidate list + dense position index
    float* cand_score = nullptr;
    uint32_t* cand_idx = nullptr;
    unsigned long long* emit = nullptr;  // [n_docs] {epoch:32 | position:32} for docs in the map
};
int launch_bm25_finalize(orama_ctx* ctx, const Bm25Finalize& f, hipStream_t stream);

// K5 — normalize_and_combine on the candidate list (token_score.rs:393-422).
struct HybridCombine {
    float vec_min = 0.0f, vec_max = 0.0f;  // fold(0.0, min/max) over the vector map (host, <= k entries)
    const uint32_t* vec_idx = nullptr;     // local doc of each vector entry (device)
    const float* vec_score = nullptr;
    uint32_t n_vec = 0;
    const float* omc_dense = nullptr;      // nullable, applied after the combine (search.rs:342-343)
    uint32_t epoch = 0;
    uint32_t cand_cap = 0;
    Bm25State* state = nullptr;
    float* cand_score = nullptr;
    uint32_t* cand_idx = nullptr;
    unsigned long long* emit = nullptr;
};
int launch_hybrid_combine(orama_ctx* ctx, const HybridCombine& h, hipStream_t stream);

// Register an uploaded (idx, score) list as the candidate list (standalone orama_hybrid_combine):
// fills `emit`, cand_count and the min/max keys.
int launch_hybrid_ingest(orama_ctx* ctx, uint32_t n, uint32_t epoch, const uint32_t* cand_idx,
                         const float* cand_score, unsigned long long* emit, Bm25State* state,
                         hipStream_t stream);

// Sparse OMC multiply (seam i): for each (local doc, multiplier): score[pos(doc)] *= multiplier.
int launch_omc_sparse(const uint32_t* d_idx, const float* d_mul, uint32_t n, uint32_t epoch,
                      const unsigned long long* emit, float* cand_score, hipStream_t stream);

// ---- sharded index: the words exchanged between the stages of a staged query
int launch_df_export(const Bm25State* st, uint32_t n_tokens, int* d_out, hipStream_t stream);
int launch_minmax_export(const Bm25State* st, long long* d_out, hipStream_t stream);
int launch_minmax_import(Bm25State* st, const long long* d_in, hipStream_t stream);
int launch_count_export(const Bm25State* st, unsigned long long* d_out, hipStream_t stream);
int launch_count_sum(const void* d_blocks, uint64_t stride, uint64_t off, uint32_t lists, unsigned long long* d_out,
                     hipStream_t stream);

// ---- the score map of one search as the facet / group kernels see it (facets.hip)
struct ScoreMapDev {
    const unsigned long long* emit = nullptr;  // [n_docs] {epoch:32 | position:32}
    const float* cand_score = nullptr;         // candidate list = the map's entries (NaN slots are empty)
    const uint32_t* cand_idx = nullptr;
    const uint64_t* docs = nullptr;            // local idx -> DocumentId; nullptr when ids are dense_base + idx
    uint64_t dense_base = 0;
    uint32_t epoch = 0;
};
constexpr uint32_t kGroupMaxK = 1024;  // max_results of a group-by (LDS-resident running top-k)
int launch_facet_count_buckets(const ScoreMapDev& m, const uint32_t* d_entry_doc, const uint64_t* d_bucket_off,
                               uint32_t n_buckets, uint64_t n_entries, unsigned long long* d_counts, hipStream_t s);
int launch_facet_count_ranges(const ScoreMapDev& m, const uint32_t* d_entry_doc, const double* d_entry_val, uint64_t n_entries,
                              const double* d_from, const double* d_to, uint32_t n_ranges, unsigned long long* d_counts,
                              hipStream_t s);
int launch_group_top(const ScoreMapDev& m, const uint32_t* d_entry_doc, const uint64_t* d_bucket_off, uint32_t n_buckets,
                     uint32_t k, uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_n, hipStream_t s);
int launch_scores_export(const ScoreMapDev& m, uint32_t list_len, uint64_t* d_out_ids, float* d_out_scores,
                         uint32_t* d_cursor, hipStream_t s);
int launch_scores_lookup(const ScoreMapDev& m, const uint32_t* d_doc, uint32_t n, float* d_out, uint8_t* d_present,
                         hipStream_t s);

// ---- synthetic postings generated in HBM (bench utility)
// len[doc] ~ LogNormal(4.0, 0.6) clipped to [4, 2000]
int launch_synth_doc_len(uint16_t* d_len, uint64_t n_docs, uint64_t seed, hipStream_t stream);
// list l = postings [list_off[l], list_off[l+1]): docs stratified-uniform over [0, n_docs) (ascending, unique),
// tf in 1..3, field length from d_len.
int launch_synth_postings(uint32_t* post_doc, uint32_t* post_val, const uint64_t* d_list_off, uint32_t n_lists,
                          uint64_t n_docs, const uint16_t* d_len, uint64_t seed, uint64_t total,
                          hipStream_t stream);

}  // namespace orama
