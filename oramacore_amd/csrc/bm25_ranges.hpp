// bm25_ranges.hpp — K3r: BM25F over resident postings by DOCUMENT RANGE, for whole batches of queries.
// Same arithmetic as K3 (bm25_kernels.hip), different data movement: see bm25_ranges.hip.
//   src/collection_manager/sides/read/index/token_score.rs:257-302, src/collection_manager/bm25.rs:369-428
#pragma once

#include "bm25_kernels.hpp"

namespace orama {

// Threads of a scoring workgroup (256 or 512): a lane carries up to 8 postings, so a range holds at most 8 x that many.
#ifndef ORAMA_K3R_WG
#define ORAMA_K3R_WG 256
#endif
constexpr uint32_t kRangeThreads = ORAMA_K3R_WG;
static_assert(kRangeThreads == 256 || kRangeThreads == 512, "scoring workgroups of four or eight waves");
constexpr uint32_t kRangeCap = 8 * kRangeThreads;         // postings one workgroup scores in LDS
constexpr uint32_t kRangeMaxWidth = 128 * kRangeThreads;  // documents per range (one bit each in the workgroup's LDS bitmap)
constexpr uint32_t kRangeMaxRefs = 64;   // non-empty posting lists per query: one bit each in the scoring launch's presence masks
constexpr uint32_t kRangeBatchMax = 32;  // queries scored by one set of launches
constexpr uint32_t kRangeStripes = 8;     // compact key lists: the scoring launch walks the batch in this many passes over its queries
constexpr uint32_t kScorePubRanges = 64;  // compact key lists: ranges of a query that publish a score for the others' floor (one per lane)

// One (token, posting list) reference of one query of the batch.
struct RangeSeg {
    uint64_t post_begin;  // first posting of the list inside the postings arrays
    uint32_t len;
    uint32_t tok_rank;    // token << 10 | rank (position of the list among the token's lists)
    float boost;
    float avg_len;
    // Dense-list accelerator (round 6, orama_post::d_acc): 1 + the offset, in 32-bit words, of this list's
    // [bitmap over the index's documents | exclusive popcount prefix of the bitmap's words] inside RangeBatch::post_acc; 0 = none
    // (a list under n_docs / 128 postings or beyond the store's budget, a posting whose normalised tf is not a tame number)
    uint64_t acc_off;
};

struct RangeQuery {
    uint64_t key_off;     // first slot of this query in the key buffer (one slot per referenced posting)
    uint64_t bounds_base; // first bounds entry of this query
    uint32_t seg_begin, seg_end;
    uint32_t width;       // a range = `width` consecutive local documents (1 .. kRangeMaxWidth, any value)
    uint32_t n_ranges;
    uint32_t n_tokens;
    uint32_t use_threshold, threshold;
    float k;
    uint32_t want_df;     // count df on the device: 1 = a token has several lists (distinct pairs), 2 = only a filter
    uint32_t track_minmax; // hybrid: reduce the largest / smallest non-NaN score into the result words
    uint32_t topk;        // keys the caller will take from this query's list (compact key lists: the scoring launch's floor)
    uint32_t pad2;
};

// Per-query result words (device): 128 bytes apart so that the per-workgroup atomics of different queries and of
// different quantities never share a cache line.
struct RangeResult {
    uint32_t count;       // documents in the score map
    uint32_t pad0[31];
    uint32_t overflow;    // a range held more than kRangeCap postings: rerun with smaller ranges
    uint32_t pad1[31];    // pad1[0]: by how much the worst range was too large, in 1/16 (postings / documents with several postings / cells against what a workgroup takes)
    uint32_t df[kMaxTokens];
    uint32_t max_key;     // hybrid: ordered(largest non-NaN score), 0 = none
    uint32_t pad2[31];
    uint32_t min_inv;     // hybrid: ~ordered(smallest non-NaN score), 0 = none
    uint32_t pad3[31];
    unsigned long long topk_tau;  // running bound of the key list's top-k reduction (launch_keys_topk, zero at launch)
    uint32_t pad4[30];
    uint32_t score_floor; // compact key lists: ordered(score) that at least `topk` documents scored so far reach (monotone)
    uint32_t pad5[31];
    uint32_t n_keys;      // compact key lists: keys appended so far (the cursor; the top-k reads it as the list's length)
    uint32_t pad6[31];
};

struct RangeBatch {
    const RangeSeg* segs = nullptr;      // grouped by query, in (token, reference) order
    const RangeQuery* queries = nullptr;
    uint32_t n_segs = 0, n_queries = 0;
    uint64_t total_postings = 0;         // referenced by the whole batch
    uint32_t max_ranges = 0;             // most ranges of a query of the batch
    uint32_t max_refs = 0;               // most references (non-empty lists) of a query of the batch: > 32 takes 64-bit presence masks
    // the scoring launch is a 1-D grid over the (query, range) pairs that exist: workgroup w scores range
    // w - range_start[q] of the query q with range_start[q] <= w < range_start[q + 1]
    uint32_t range_start[kRangeBatchMax + 1] = {0};
    // compact key lists: the launch scores the batch in kRangeStripes passes — stripe s holds ranges [s n / S, (s + 1) n / S) of
    // every query (n = its ranges), queries in order inside a stripe — so that every query's ranges are scored THROUGHOUT
    // the launch whatever the queries' sizes: stripe_start[s * kRangeBatchMax + q] = first workgroup of (stripe s, query q),
    // unused queries repeat the next entry; stripe_start[S * kRangeBatchMax] = all pairs
    // (a DEVICE table of kRangeStripes * kRangeBatchMax + 1 words, uploaded with the batch's other tables; stripe_total = its last
    // word, for the launcher's check)
    const uint32_t* stripe_start = nullptr;
    uint32_t stripe_total = 0;
    uint64_t max_bound_entries = 0;      // largest references x (ranges + 1) of a query: grid.x of the bounds launch
    const uint32_t* post_doc = nullptr;
    const uint32_t* post_val = nullptr;
    // tf / ((1 - b) + b * len / avg_len) per posting, stored by the store for ITS b and average lengths (nullptr when the
    // query's b differs: the kernels then divide themselves — same operations, same bits)
    const float* post_ntf = nullptr;
    // dense-list accelerators (RangeSeg::acc_off): per list acc_words bitmap words, then acc_words word ranks; nullptr = none
    const uint32_t* post_acc = nullptr;
    uint32_t acc_words = 0;              // ceil(n_docs / 32)
    uint32_t* bounds = nullptr;
    const uint64_t* docs = nullptr;      // local idx -> DocumentId (filter); nullptr when the ids are dense_base + idx
    uint64_t dense_base = 0;
    const uint64_t* allow = nullptr;
    uint64_t allow_bits = 0;
    float b = 0.75f;
    const float* idf = nullptr;          // [n_queries][kMaxTokens]
    const float* omc_dense = nullptr;
    unsigned long long* keys = nullptr;  // ordered(score) << 32 | ~local doc; 0 = empty
    // Compact key lists (round 5; plain top-k batches): results[query].n_keys starts at zero (the bounds launch clears the
    // result words) and the scoring launch APPENDS — a workgroup writes only the keys that reach its floor (a score at least
    // `topk` documents are known to reach), behind one cursor bump per workgroup; the top-k reads n_keys keys instead of one
    // slot per posting.  (The cursors of a batch live 1 KB apart, like every other per-query word: 12 K returning atomics
    // on ONE cache line took the scoring launch from 3.5 to 5.7 us per query — profiles/r05_k3r_compact_ab_v1.log.)
    // 0: one slot per posting (score maps, OMC, hybrid min / max, wide masks).
    uint32_t compact_keys = 0;
    // Compact key lists: what the FIRST kScorePubRanges ranges of each query publish — one word each, the smallest of the
    // workgroup's four wave maxima: at least 4 of the range's documents score that or more — [n_queries][kScorePubRanges] u32,
    // zeroed by the bounds launch.  The ceil(topk / 4)-th largest published word is a floor for every later range of the query
    // (that many ranges hold 4 documents each at or above it).  Plain stores and loads: same-address global atomics (a
    // cursor per query, a histogram of the scores) cost ~50 ns EACH on this chip — profiles/r05_k3r_compact_ab_v*.log.
    uint32_t* score_pub = nullptr;
    RangeResult* results = nullptr;
    // score-map mode (n_queries == 1): besides its key, every slot gets the map entry it stands for — map_idx[slot] = local
    // document (0xffffffff: none), map_score[slot] = its score (after OMC; NaN stays), map_emit[document] = epoch << 32 | slot:
    // the candidate list + position index facets / groups / export / lookup read (ScoreMapDev, bm25_kernels.hpp)
    uint32_t* map_idx = nullptr;
    float* map_score = nullptr;
    unsigned long long* map_emit = nullptr;
    uint32_t map_epoch = 0;
    uint32_t any_minmax = 0;  // a query of the batch tracks the min / max of its scores (hybrid)
    uint32_t debug = 0;  // comparison build only (ORAMA_K3R_DBG): 1 skip the merge, 2 skip the fold, 4 stop after the bounds loads
};

// bounds[query][r][reference] = postings of the reference whose document lies in a range < r.  Also zeroes `results`.
int launch_range_bounds(orama_ctx* ctx, const RangeBatch& b, hipStream_t stream);
// df_only: count distinct (token, document) pairs into results[q].df for the queries that want it.
int launch_range_score(orama_ctx* ctx, const RangeBatch& b, bool df_only, hipStream_t stream);
#if ORAMA_COMPARISON_KERNELS
// The plain top-k batch with compact key lists, round-6 experiment (bm25_ranges_fast.hip): singletons need no ranks, lists whose
// singletons cannot reach the published floor are not scored.  Same answers, bit for bit, fewer instructions — and not faster
// (profiles/r06_k3r_fast_body.md): comparison builds only, option "k3r_fast" (default 0).
int launch_range_score_fast(orama_ctx* ctx, const RangeBatch& b, hipStream_t stream);
// The round-2/3 scoring launch (bm25_ranges_merge.hip: 64-bit keys merged by a merge tree in LDS), kept for A/B runs only.
int launch_range_score_merge(orama_ctx* ctx, const RangeBatch& b, bool df_only, hipStream_t stream);
#endif
// post_ntf[i] = tf_i / ((1 - b) + b * len_i / avg_len[list of i]) for every posting of the store (list_off: n_lists + 1
// offsets, list_avg: the average length of each list's field; both on the device).
#if ORAMA_COMPARISON_KERNELS
// Dense-list accelerators: for each of the n_acc lists (acc_list: list index, its postings at list_off), the bitmap of its
// documents and the exclusive popcount prefix of the bitmap's words at d_acc + i * 2 * acc_words (zeroed by the call), and the
// smallest / largest post_ntf of the list at d_minmax[2 i], [2 i + 1] (a NaN anywhere yields NaN).
int launch_acc_build(const uint32_t* post_doc, const float* post_ntf, const uint64_t* d_list_off, const uint32_t* d_acc_list, uint32_t n_acc,
                     uint32_t acc_words, uint32_t* d_acc, float* d_minmax, hipStream_t stream);
#endif
int launch_ntf_precompute(const uint32_t* post_val, float* post_ntf, const uint64_t* d_list_off, const float* d_list_avg,
                          uint32_t n_lists, uint64_t n_postings, float b, hipStream_t stream);
// Hybrid: the full-text score of `n` given documents (local indices) of query `qi`, by the same fold as the range kernel
// (lists of a token in reference order, tokens ascending): out_score[j] / out_present[j] (in the score map or not).
// (d_n, optional, device: only the first min(n, *d_n) documents exist — the device tail of the one-call hybrid search)
int launch_range_score_docs(orama_ctx* ctx, const RangeBatch& b, uint32_t qi, const uint32_t* d_doc, uint32_t n, float* d_out_score,
                            uint32_t* d_out_present, hipStream_t stream, const uint32_t* d_n = nullptr);

}  // namespace orama
