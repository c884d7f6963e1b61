/*
 * orama_oracle.h — CPU restatement of OramaCore's hybrid-search scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / CPU baseline.  The product path (oramacore_amd + liborama_hip.so) never links,
 * imports or calls it.
 *
 * Parity status:
 *   - BM25F scorer arithmetic (orc_bm25f_*):      PINNED by the reference's own known-answer
 *     tests src/collection_manager/bm25.rs:533-563, 911-983, 985-1043 (tests/golden/bm25_kat.json).
 *   - facet counts, score-ordered groups (orc_facet_count_*, orc_group_top): PINNED by the facet / group results the
 *     reference's own integration tests assert — src/tests/facets.rs:9-576 (8 cases), src/tests/groupby.rs:9-174, 416-467,
 *     580-754 (6 cases) — held as data in tests/golden/reference_facet_cases.json, reference_group_cases.json
 *     (tests/test_reference_cases.py).
 *   - ntf / field boost / exact-match factor / threshold / OMC: CONSTRAINED (inequalities, orders, ratios, hit counts) by nineteen more of
 *     its cases (tests/golden/reference_cases.json); the formulas themselves live in the un-vendored crates below.
 *   - cosine scan / top-k ties / hybrid combine:  PARITY UNPINNED — the arithmetic lives in the
 *     un-vendored crates oramacore_fields 0.2.0 / oramacore_lib 0.4.4 (Cargo.lock:5313-5335) and
 *     the reference holds no numeric test for it; this file restates the published semantics
 *     observed at the in-tree call sites and declares every assumption it makes.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef ORAMA_ORACLE_H
#define ORAMA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- vector path (a1, a2) */

/* Cosine distance of one row, fp32, strictly sequential accumulation (no FMA contraction):
 *   dist = 1 - (q·x) / (sqrt(q·q) * sqrt(x·x))
 * Follows the contract documented at
 * src/collection_manager/sides/read/index/embedding_field.rs:244-249 ("oramacore_fields cosine
 * distance = 1.0 - cosine_similarity").  ASSUMPTION (unpinned): a zero-norm query scores 0
 * against everything (dist = 1). */
float orc_cosine_distance_f32(const float* q, const float* x, uint32_t d);
/* Same in f64 (error-bound reference for the 1e-4 tolerance). */
double orc_cosine_distance_f64(const float* q, const float* x, uint32_t d);
/* Squared-L2 distance (build-side extension; no reference counterpart — SURVEY F4). */
float orc_l2sq_distance_f32(const float* q, const float* x, uint32_t d);

/* All N distances (metric 0 = cosine, 1 = squared L2). */
void orc_distances_f32(const float* corpus, uint64_t n, uint32_t d, const float* q,
                       int metric, float* out_dist);
/* Multi-threaded variant (plain pthreads, row-parallel) for the all-cores CPU baseline. */
void orc_distances_f32_mt(const float* corpus, uint64_t n, uint32_t d, const float* q,
                          int metric, float* out_dist, int threads);

/* Row validity at insert: finite and non-zero norm.  ASSUMPTION (unpinned) for the
 * `Option` returned by EmbeddingIndexer::index_vec_vec (embedding_field.rs:232-237). */
int orc_row_is_valid(const float* x, uint32_t d);

/* a1 — EmbeddingStorage::search / search_with_filter (third-party; call site
 * embedding_field.rs:255-266): per-ROW k nearest by distance.  `row_doc[i]` is the DocumentId of
 * row i (several rows may share one id); `dead` (nullable) is a per-row byte, non-zero = deleted;
 * `allow_bitmap` (nullable) has one bit per doc id (bit i of word i/64), ids >= bitmap_bits are
 * rejected — this is the materialised `DocumentFilter::contains` (embedding_field.rs:54-61).
 * Order (declared tie rule): distance asc, then doc id asc, then row asc; selection at
 * the k boundary among equal distances keeps the lowest ROW indices.
 * Returns the number of results written (<= k). */
uint32_t orc_vector_search(const float* corpus, uint64_t n, uint32_t d, const uint64_t* row_doc,
                           const uint8_t* dead, const float* q, int metric, uint32_t k,
                           const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                           uint64_t* out_doc, float* out_dist, uint64_t* out_row);

/* Model::rescale_score — src/python/embeddings.rs:71-92.  is_e5 != 0 selects the E5 rescale
 * (clamp(s, 0.7, 1.0) - 0.7) / (1.0 - 0.7) with DELTA computed in f32 exactly as the source. */
float orc_rescale_score(float score, int is_e5);

/* a2 — EmbeddingFieldStorage::search epilogue, embedding_field.rs:268-276:
 *   similarity = 1 - distance; score = rescale(similarity); if score >= min_similarity:
 *   out[doc] += score   (entries created at 0.0).
 * in/out map is a pair of parallel arrays sorted by doc id ascending; `*io_n` entries on entry,
 * updated on return (capacity must be >= *io_n + n_hits). */
void orc_embedding_epilogue(const uint64_t* hit_doc, const float* hit_dist, uint32_t n_hits,
                            int is_e5, float min_similarity,
                            uint64_t* io_doc, float* io_score, uint64_t* io_n);

/* ---------------------------------------------------------------- BM25F (a5–a8) */

/* calculate_idf — bm25.rs:78-82: ln_1p((N - df + 0.5) / (df + 0.5)), f32, libm log1pf
 * (Rust's f32::ln_1p lowers to the platform log1pf). */
float orc_bm25_idf(float total_documents, uint64_t df);
/* bm25f_normalized_tf — bm25.rs:99-110: tf / (1 - b + b * (len / avglen)). */
float orc_bm25f_normalized_tf(uint32_t tf, uint32_t field_len, float avg_len, float b);
/* bm25f_score — bm25.rs:124-126: idf * (k + 1) * S / (k + S), evaluated left to right. */
float orc_bm25f_score(float s, float k, float idf);
/* BM25Scorer::add (legacy single-field form) — bm25.rs:248-307; returns the value added to the
 * document score, or NaN when the contribution is skipped. */
float orc_bm25_legacy_add(uint32_t tf, uint32_t field_len, float avg_len, float total_docs_with_field,
                          uint64_t docs_with_term, float k, float weight, float b, float boost);

/* One posting-list "entry": the per_doc_ntf vector one field returns for one query token
 * (token_score.rs:264-271).  Docs are unique inside an entry. */
typedef struct {
    uint32_t token;        /* query token index (term_index, token_score.rs:257) */
    const uint64_t* doc;   /* len entries */
    const float* ntf;      /* len entries; already includes boost + length norm (token_score.rs:268) */
    uint64_t len;
} orc_entry;

/* a6–a8 — search_full_text inner loop + finalize_term + get_scores:
 * token_score.rs:257-302, bm25.rs:369-428 (threshold scorer) / :484-524 (plain scorer).
 * Entries are consumed in array order inside each token (this fixes the f32 order of
 * S = sum(weight * ntf), weight == 1.0; ASSUMPTION: callers pass fields in ascending FieldId, the
 * reference iterates a HashSet — token_score.rs:160-177).  df = distinct docs of the token
 * (max 1), N = total_documents (token_score.rs:221), k = 1.2, phrase boost = 1.0.
 * use_threshold != 0 selects BM25FScorerWithThreshold with `threshold` = floor(n_tokens * t)
 * computed by the caller (token_score.rs:211-216).
 * Output: doc ids ascending + scores; returns the number of docs in the score map. The out arrays
 * must hold sum(len) entries. */
uint64_t orc_search_full_text(const orc_entry* entries, uint32_t n_entries, uint32_t n_tokens,
                              float total_documents, float k, int use_threshold, uint32_t threshold,
                              uint64_t* out_doc, float* out_score);

/* ---------------------------------------------------------------- hybrid / OMC / top-n (a9–a11) */

/* normalize_and_combine — token_score.rs:393-422.  Inputs: two maps as (doc asc, score) arrays.
 * max = fold(0.0, f32::max) and min = fold(0.0, f32::min) over BOTH maps (f32::max/min ignore a
 * NaN operand); v' = (v - min) / (max - min); out = fulltext', out[doc] += vector'[doc]
 * (created at 0.0 when absent).  Output sorted by doc asc; capacity n_v + n_f. Returns size. */
uint64_t orc_normalize_and_combine(const uint64_t* v_doc, const float* v_score, uint64_t n_v,
                                   const uint64_t* f_doc, const float* f_score, uint64_t n_f,
                                   uint64_t* out_doc, float* out_score);

/* apply_omc_multipliers — search.rs:39-48: score[doc] *= omc[doc] when present.
 * omc map given as (doc asc, multiplier). */
void orc_apply_omc(uint64_t* doc, float* score, uint64_t n,
                   const uint64_t* omc_doc, const float* omc_mul, uint64_t n_omc);

/* top_n — sort.rs:260-279: drop NaN, keep `n` best by score desc.  Declared tie rule
 * (reference unpinned): equal scores ordered by doc id asc.  Returns count written. */
uint64_t orc_top_n(const uint64_t* doc, const float* score, uint64_t n_in, uint64_t n,
                   uint64_t* out_doc, float* out_score);

/* ---------------------------------------------------------------- facets / groups over the score map (§8f rank 4) */

/* BoolFieldStorage::calculate_facet (src/collection_manager/sides/read/index/bool_field.rs:182-208) and
 * StringFilterFieldStorage::calculate_facet (string_filter_field.rs:175-193): per field value (bucket b = doc ids
 * bucket_doc[bucket_off[b] .. bucket_off[b+1])) the number of ids that are keys of token_scores (NaN scores count). */
void orc_facet_count_buckets(const uint64_t* map_doc, uint64_t n_map, const uint64_t* bucket_off,
                             const uint64_t* bucket_doc, uint32_t n_buckets, uint64_t* out_counts);
/* NumberFieldStorage::calculate_facet (number_field.rs:368-387): NumberFilter::Between is inclusive on both ends
 * (number_field.rs:604-631); one (doc, value) entry per stored number. */
void orc_facet_count_ranges(const uint64_t* map_doc, uint64_t n_map, const uint64_t* doc, const double* value, uint64_t n,
                            const double* from, const double* to, uint32_t n_ranges, uint64_t* out_counts);
/* sort_groups without sort_by (src/collection_manager/sides/read/sort.rs:203-213): per group the best `max_results`
 * documents of the group that are in token_scores with a non-NaN score.  DECLARED tie rule: score desc, id asc. */
void orc_group_top(const uint64_t* map_doc, const float* map_score, uint64_t n_map, const uint64_t* group_off,
                   const uint64_t* group_doc, uint32_t n_groups, uint32_t max_results, uint64_t* out_doc,
                   float* out_score, uint32_t* out_n);

#ifdef __cplusplus
}
#endif
#endif /* ORAMA_ORACLE_H */
