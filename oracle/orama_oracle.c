/*
 * orama_oracle.c — CPU restatement of OramaCore's hybrid-search scoring path (plain C99).
 * TEST INFRASTRUCTURE ONLY — see orama_oracle.h for the scope statement and parity status.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: every f32 operation rounds once, in the
 * order written, like the reference's scalar Rust).
 */
#include "orama_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */

static int f32_is_normal(float x) { /* Rust f32::is_normal — bm25.rs:387,501 */
    return fpclassify(x) == FP_NORMAL;
}
/* Rust f32::max / f32::min: a NaN operand is ignored (token_score.rs:398-401). */
static float rust_f32_max(float a, float b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    return a > b ? a : b;
}
static float rust_f32_min(float a, float b) {
    if (isnan(a)) return b;
    if (isnan(b)) return a;
    return a < b ? a : b;
}

/* ------------------------------------------------------------------ vector path */

float orc_cosine_distance_f32(const float* q, const float* x, uint32_t d) {
    float dot = 0.0f, nq = 0.0f, nx = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        dot = dot + q[i] * x[i];
        nq = nq + q[i] * q[i];
        nx = nx + x[i] * x[i];
    }
    float den = sqrtf(nq) * sqrtf(nx);
    if (!(den > 0.0f)) return 1.0f; /* zero-norm operand: similarity 0 (declared assumption) */
    return 1.0f - dot / den;
}

double orc_cosine_distance_f64(const float* q, const float* x, uint32_t d) {
    double dot = 0.0, nq = 0.0, nx = 0.0;
    for (uint32_t i = 0; i < d; ++i) {
        dot += (double)q[i] * (double)x[i];
        nq += (double)q[i] * (double)q[i];
        nx += (double)x[i] * (double)x[i];
    }
    double den = sqrt(nq) * sqrt(nx);
    if (!(den > 0.0)) return 1.0;
    return 1.0 - dot / den;
}

float orc_l2sq_distance_f32(const float* q, const float* x, uint32_t d) {
    float acc = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        float t = q[i] - x[i];
        acc = acc + t * t;
    }
    return acc;
}

void orc_distances_f32(const float* corpus, uint64_t n, uint32_t d, const float* q, int metric,
                       float* out_dist) {
    for (uint64_t r = 0; r < n; ++r) {
        const float* x = corpus + r * (uint64_t)d;
        out_dist[r] = metric == 0 ? orc_cosine_distance_f32(q, x, d) : orc_l2sq_distance_f32(q, x, d);
    }
}

typedef struct {
    const float* corpus;
    uint64_t lo, hi;
    uint32_t d;
    const float* q;
    int metric;
    float* out;
} mt_job;

static void* mt_worker(void* p) {
    mt_job* j = (mt_job*)p;
    orc_distances_f32(j->corpus + j->lo * (uint64_t)j->d, j->hi - j->lo, j->d, j->q, j->metric,
                      j->out + j->lo);
    return NULL;
}

void orc_distances_f32_mt(const float* corpus, uint64_t n, uint32_t d, const float* q, int metric,
                          float* out_dist, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    mt_job job[256];
    uint64_t per = (n + (uint64_t)threads - 1) / (uint64_t)threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        uint64_t lo = per * (uint64_t)t, hi = lo + per;
        if (lo >= n) break;
        if (hi > n) hi = n;
        job[t] = (mt_job){corpus, lo, hi, d, q, metric, out_dist};
        pthread_create(&tid[t], NULL, mt_worker, &job[t]);
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
}

int orc_row_is_valid(const float* x, uint32_t d) {
    float n2 = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        if (!isfinite(x[i])) return 0;
        n2 = n2 + x[i] * x[i];
    }
    return isfinite(n2) && n2 > 0.0f;
}

typedef struct {
    float dist;
    uint64_t doc;
    uint64_t row;
} hit_t;

static int hit_cmp_select(const void* a, const void* b) { /* distance asc, row asc */
    const hit_t* x = (const hit_t*)a;
    const hit_t* y = (const hit_t*)b;
    if (x->dist < y->dist) return -1;
    if (x->dist > y->dist) return 1;
    return (x->row > y->row) - (x->row < y->row);
}
static int hit_cmp_final(const void* a, const void* b) { /* distance asc, doc asc, row asc */
    const hit_t* x = (const hit_t*)a;
    const hit_t* y = (const hit_t*)b;
    if (x->dist < y->dist) return -1;
    if (x->dist > y->dist) return 1;
    if (x->doc != y->doc) return x->doc < y->doc ? -1 : 1;
    return (x->row > y->row) - (x->row < y->row);
}

uint32_t orc_vector_search(const float* corpus, uint64_t n, uint32_t d, const uint64_t* row_doc,
                           const uint8_t* dead, const float* q, int metric, uint32_t k,
                           const uint64_t* allow_bitmap, uint64_t bitmap_bits, uint64_t* out_doc,
                           float* out_dist, uint64_t* out_row) {
    hit_t* hits = (hit_t*)malloc(sizeof(hit_t) * (size_t)(n ? n : 1));
    uint64_t m = 0;
    for (uint64_t r = 0; r < n; ++r) {
        if (dead && dead[r]) continue;
        uint64_t doc = row_doc ? row_doc[r] : r;
        if (allow_bitmap) {
            if (doc >= bitmap_bits) continue;
            if (!((allow_bitmap[doc >> 6] >> (doc & 63)) & 1ull)) continue;
        }
        const float* x = corpus + r * (uint64_t)d;
        float dist = metric == 0 ? orc_cosine_distance_f32(q, x, d) : orc_l2sq_distance_f32(q, x, d);
        if (isnan(dist)) continue;
        hits[m++] = (hit_t){dist, doc, r};
    }
    qsort(hits, (size_t)m, sizeof(hit_t), hit_cmp_select);
    uint64_t kk = m < k ? m : k;
    qsort(hits, (size_t)kk, sizeof(hit_t), hit_cmp_final);
    for (uint64_t i = 0; i < kk; ++i) {
        out_doc[i] = hits[i].doc;
        out_dist[i] = hits[i].dist;
        if (out_row) out_row[i] = hits[i].row;
    }
    free(hits);
    return (uint32_t)kk;
}

float orc_rescale_score(float score, int is_e5) {
    if (!is_e5) return score;
    const float MIN = 0.7f, MAX = 1.0f;
    const float DELTA = MAX - MIN;
    float c = score; /* f32::clamp keeps NaN */
    if (c < MIN) c = MIN;
    if (c > MAX) c = MAX;
    return (c - MIN) / DELTA;
}

void orc_embedding_epilogue(const uint64_t* hit_doc, const float* hit_dist, uint32_t n_hits,
                            int is_e5, float min_similarity, uint64_t* io_doc, float* io_score,
                            uint64_t* io_n) {
    uint64_t n = *io_n;
    for (uint32_t i = 0; i < n_hits; ++i) {
        float similarity = 1.0f - hit_dist[i];
        float score = orc_rescale_score(similarity, is_e5);
        if (!(score >= min_similarity)) continue;
        /* sorted insert / accumulate */
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2;
            if (io_doc[mid] < hit_doc[i]) lo = mid + 1; else hi = mid;
        }
        if (lo < n && io_doc[lo] == hit_doc[i]) {
            io_score[lo] = io_score[lo] + score;
        } else {
            memmove(io_doc + lo + 1, io_doc + lo, (size_t)(n - lo) * sizeof(uint64_t));
            memmove(io_score + lo + 1, io_score + lo, (size_t)(n - lo) * sizeof(float));
            io_doc[lo] = hit_doc[i];
            io_score[lo] = 0.0f + score;
            ++n;
        }
    }
    *io_n = n;
}

/* ------------------------------------------------------------------ BM25F */

float orc_bm25_idf(float total_documents, uint64_t df_) {
    float df = (float)df_;
    float ratio = (total_documents - df + 0.5f) / (df + 0.5f);
    return log1pf(ratio);
}

float orc_bm25f_normalized_tf(uint32_t tf_, uint32_t field_len, float avg_len, float b) {
    float tf = (float)tf_;
    float len = (float)field_len;
    return tf / (1.0f - b + b * (len / avg_len));
}

float orc_bm25f_score(float s, float k, float idf) { return idf * (k + 1.0f) * s / (k + s); }

float orc_bm25_legacy_add(uint32_t tf, uint32_t field_len, float avg_len,
                          float total_docs_with_field, uint64_t docs_with_term, float k,
                          float weight, float b, float boost) {
    float ntf = orc_bm25f_normalized_tf(tf, field_len, avg_len, b);
    float weighted = weight * ntf;
    float idf = orc_bm25_idf(total_docs_with_field, docs_with_term);
    float term_score = orc_bm25f_score(weighted, k, idf);
    if (isnan(term_score)) return NAN;
    return term_score * boost;
}

typedef struct {
    uint64_t doc;
    uint32_t order; /* position in the token's contribution stream */
    float ntf;
} contrib_t;

static int contrib_cmp(const void* a, const void* b) {
    const contrib_t* x = (const contrib_t*)a;
    const contrib_t* y = (const contrib_t*)b;
    if (x->doc != y->doc) return x->doc < y->doc ? -1 : 1;
    return (x->order > y->order) - (x->order < y->order);
}

typedef struct {
    uint64_t doc;
    float score;
    uint32_t mask;
} acc_t;

static int acc_cmp(const void* a, const void* b) {
    const acc_t* x = (const acc_t*)a;
    const acc_t* y = (const acc_t*)b;
    return (x->doc > y->doc) - (x->doc < y->doc);
}

typedef struct {
    acc_t a;
    uint64_t idx;
} tagged_t;

static int cmp_tagged(const void* a, const void* b) { /* doc asc, then append (token) order */
    const tagged_t* x = (const tagged_t*)a;
    const tagged_t* y = (const tagged_t*)b;
    int c = acc_cmp(&x->a, &y->a);
    if (c) return c;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

uint64_t orc_search_full_text(const orc_entry* entries, uint32_t n_entries, uint32_t n_tokens,
                              float total_documents, float k, int use_threshold, uint32_t threshold,
                              uint64_t* out_doc, float* out_score) {
    uint64_t total = 0;
    for (uint32_t e = 0; e < n_entries; ++e) total += entries[e].len;
    /* per-(token, doc) applied term scores, appended in token order */
    acc_t* applied = (acc_t*)malloc(sizeof(acc_t) * (size_t)(total ? total : 1));
    uint64_t n_applied = 0;
    contrib_t* buf = (contrib_t*)malloc(sizeof(contrib_t) * (size_t)(total ? total : 1));

    for (uint32_t t = 0; t < n_tokens; ++t) {
        uint64_t m = 0;
        for (uint32_t e = 0; e < n_entries; ++e) {
            if (entries[e].token != t) continue;
            for (uint64_t i = 0; i < entries[e].len; ++i) {
                buf[m].doc = entries[e].doc[i];
                buf[m].order = (uint32_t)m;
                buf[m].ntf = entries[e].ntf[i];
                ++m;
            }
        }
        qsort(buf, (size_t)m, sizeof(contrib_t), contrib_cmp);
        uint64_t df = 0;
        for (uint64_t i = 0; i < m;) { /* corpus_docs.len() — token_score.rs:262-275 */
            uint64_t j = i;
            while (j < m && buf[j].doc == buf[i].doc) ++j;
            ++df;
            i = j;
        }
        if (df < 1) df = 1;
        float idf = orc_bm25_idf(total_documents, df);
        for (uint64_t i = 0; i < m;) {
            uint64_t j = i;
            float s = 0.0f; /* Iterator::sum::<f32>(): folds from 0.0 (from -0.0 since Rust 1.83 — the same value for
                             * every sum that matters here: a sum that stays +-0.0 is not `is_normal` and is skipped) */
            while (j < m && buf[j].doc == buf[i].doc) {
                s = s + 1.0f * buf[j].ntf; /* contrib.weight * contrib.normalized_tf, weight = 1.0 */
                ++j;
            }
            if (f32_is_normal(s)) {
                float term_score = orc_bm25f_score(s, k, idf);
                if (!isnan(term_score)) {
                    float final_score = term_score * 1.0f; /* phrase boost */
                    applied[n_applied].doc = buf[i].doc;
                    applied[n_applied].score = final_score;
                    applied[n_applied].mask = 1u << (t & 31u); /* 1 << term_index on u32: release-mode Rust masks the shift amount */
                    ++n_applied;
                }
            }
            i = j;
        }
    }
    /* fold into document_scores in token order: stable by construction (tokens appended in order,
     * qsort on doc only is not stable → add the append index as tiebreak). */
    tagged_t* tg = (tagged_t*)malloc(sizeof(tagged_t) * (size_t)(n_applied ? n_applied : 1));
    for (uint64_t i = 0; i < n_applied; ++i) { tg[i].a = applied[i]; tg[i].idx = i; }
    qsort(tg, (size_t)n_applied, sizeof(tagged_t), cmp_tagged);
    uint64_t n_out = 0;
    for (uint64_t i = 0; i < n_applied;) {
        uint64_t j = i;
        float score = 0.0f; /* entry().or_insert(0.0) */
        uint32_t mask = 0;
        while (j < n_applied && tg[j].a.doc == tg[i].a.doc) {
            score = score + tg[j].a.score;
            mask |= tg[j].a.mask;
            ++j;
        }
        int keep = 1;
        if (use_threshold) { /* get_scores — bm25.rs:416-428 */
            uint32_t c = (uint32_t)__builtin_popcount(mask);
            keep = c >= threshold;
        }
        if (keep) {
            out_doc[n_out] = tg[i].a.doc;
            out_score[n_out] = score;
            ++n_out;
        }
        i = j;
    }
    free(tg);
    free(buf);
    free(applied);
    return n_out;
}

/* ------------------------------------------------------------------ hybrid / OMC / top-n */

uint64_t orc_normalize_and_combine(const uint64_t* v_doc, const float* v_score, uint64_t n_v,
                                   const uint64_t* f_doc, const float* f_score, uint64_t n_f,
                                   uint64_t* out_doc, float* out_score) {
    float mx = 0.0f, mn = 0.0f;
    for (uint64_t i = 0; i < n_v; ++i) mx = rust_f32_max(mx, v_score[i]);
    {
        float m2 = 0.0f;
        for (uint64_t i = 0; i < n_f; ++i) m2 = rust_f32_max(m2, f_score[i]);
        mx = rust_f32_max(mx, m2);
    }
    for (uint64_t i = 0; i < n_v; ++i) mn = rust_f32_min(mn, v_score[i]);
    {
        float m2 = 0.0f;
        for (uint64_t i = 0; i < n_f; ++i) m2 = rust_f32_min(m2, f_score[i]);
        mn = rust_f32_min(mn, m2);
    }
    /* merge two doc-sorted maps */
    uint64_t i = 0, j = 0, n = 0;
    while (i < n_v || j < n_f) {
        if (j >= n_f || (i < n_v && v_doc[i] < f_doc[j])) {
            float vn = (v_score[i] - mn) / (mx - mn);
            out_doc[n] = v_doc[i];
            out_score[n] = 0.0f + vn; /* entry(k).or_default() += v */
            ++i;
        } else if (i >= n_v || f_doc[j] < v_doc[i]) {
            out_doc[n] = f_doc[j];
            out_score[n] = (f_score[j] - mn) / (mx - mn);
            ++j;
        } else {
            float fn = (f_score[j] - mn) / (mx - mn);
            float vn = (v_score[i] - mn) / (mx - mn);
            out_doc[n] = f_doc[j];
            out_score[n] = fn + vn;
            ++i;
            ++j;
        }
        ++n;
    }
    return n;
}

void orc_apply_omc(uint64_t* doc, float* score, uint64_t n, const uint64_t* omc_doc,
                   const float* omc_mul, uint64_t n_omc) {
    if (n_omc == 0) return;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t lo = 0, hi = n_omc;
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2;
            if (omc_doc[mid] < doc[i]) lo = mid + 1; else hi = mid;
        }
        if (lo < n_omc && omc_doc[lo] == doc[i]) score[i] = score[i] * omc_mul[lo];
    }
}

typedef struct {
    float score;
    uint64_t doc;
} ts_t;

static int ts_cmp(const void* a, const void* b) { /* score desc, doc asc */
    const ts_t* x = (const ts_t*)a;
    const ts_t* y = (const ts_t*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->doc > y->doc) - (x->doc < y->doc);
}

uint64_t orc_top_n(const uint64_t* doc, const float* score, uint64_t n_in, uint64_t n,
                   uint64_t* out_doc, float* out_score) {
    ts_t* a = (ts_t*)malloc(sizeof(ts_t) * (size_t)(n_in ? n_in : 1));
    uint64_t m = 0;
    for (uint64_t i = 0; i < n_in; ++i) {
        if (isnan(score[i])) continue; /* NotNan::new(..) Err → continue — sort.rs:264-268 */
        a[m].score = score[i];
        a[m].doc = doc[i];
        ++m;
    }
    qsort(a, (size_t)m, sizeof(ts_t), ts_cmp);
    uint64_t kk = m < n ? m : n;
    for (uint64_t i = 0; i < kk; ++i) {
        out_doc[i] = a[i].doc;
        out_score[i] = a[i].score;
    }
    free(a);
    return kk;
}

/* ---------------------------------------------------------------- facets / groups */

static int u64_cmp(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x > y) - (x < y);
}

/* token_scores.contains_key(doc) over a sorted copy of the map's keys */
static int map_contains(const uint64_t* sorted, uint64_t n, uint64_t doc) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (sorted[mid] < doc) lo = mid + 1; else hi = mid;
    }
    return lo < n && sorted[lo] == doc;
}

/* BoolFieldStorage::calculate_facet (bool_field.rs:182-208) and StringFilterFieldStorage::calculate_facet
 * (string_filter_field.rs:175-193): for every value of the field (a bucket = the doc ids `storage.filter(value)`
 * yields) count the ids that are keys of token_scores.  A NaN score is still a key. */
void orc_facet_count_buckets(const uint64_t* map_doc, uint64_t n_map, const uint64_t* bucket_off,
                             const uint64_t* bucket_doc, uint32_t n_buckets, uint64_t* out_counts) {
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n_map ? n_map : 1));
    memcpy(keys, map_doc, sizeof(uint64_t) * (size_t)n_map);
    qsort(keys, (size_t)n_map, sizeof(uint64_t), u64_cmp);
    for (uint32_t b = 0; b < n_buckets; ++b) {
        uint64_t c = 0;
        for (uint64_t i = bucket_off[b]; i < bucket_off[b + 1]; ++i) c += (uint64_t)map_contains(keys, n_map, bucket_doc[i]);
        out_counts[b] = c;
    }
    free(keys);
}

/* NumberFieldStorage::calculate_facet (number_field.rs:368-387): per range, the documents the filter
 * NumberFilter::Between((from, to)) yields — BetweenInclusive on both storages (number_field.rs:604-631) — that are keys
 * of token_scores.  One (doc, value) entry per stored number (a doc holding two numbers inside one range is yielded
 * twice by the storage iterator and counted twice, as `.filter(..).count()` does).  Values compared as f64. */
void orc_facet_count_ranges(const uint64_t* map_doc, uint64_t n_map, const uint64_t* doc, const double* value, uint64_t n,
                            const double* from, const double* to, uint32_t n_ranges, uint64_t* out_counts) {
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n_map ? n_map : 1));
    memcpy(keys, map_doc, sizeof(uint64_t) * (size_t)n_map);
    qsort(keys, (size_t)n_map, sizeof(uint64_t), u64_cmp);
    for (uint32_t r = 0; r < n_ranges; ++r) out_counts[r] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (!map_contains(keys, n_map, doc[i])) continue;
        for (uint32_t r = 0; r < n_ranges; ++r)
            if (from[r] <= value[i] && value[i] <= to[r]) out_counts[r] += 1;
    }
    free(keys);
}

static int ts_by_doc(const void* a, const void* b) {
    const ts_t* x = (const ts_t*)a;
    const ts_t* y = (const ts_t*)b;
    return (x->doc > y->doc) - (x->doc < y->doc);
}

/* sort_groups without sort_by (sort.rs:203-213): per group (the doc-id set GroupContext::execute built, group.rs:
 * 107-170) a CappedHeap of `max_results` over the docs that are keys of token_scores with a non-NaN score, best score
 * first.  Declared tie rule (the reference iterates a HashSet into a heap — undefined): score desc, DocumentId asc.
 * out_doc / out_score: n_groups x max_results, out_n: n_groups. */
void orc_group_top(const uint64_t* map_doc, const float* map_score, uint64_t n_map, const uint64_t* group_off,
                   const uint64_t* group_doc, uint32_t n_groups, uint32_t max_results, uint64_t* out_doc,
                   float* out_score, uint32_t* out_n) {
    ts_t* m = (ts_t*)malloc(sizeof(ts_t) * (size_t)(n_map ? n_map : 1));
    for (uint64_t i = 0; i < n_map; ++i) {
        m[i].doc = map_doc[i];
        m[i].score = map_score[i];
    }
    qsort(m, (size_t)n_map, sizeof(ts_t), ts_by_doc);
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint64_t len = group_off[g + 1] - group_off[g];
        ts_t* c = (ts_t*)malloc(sizeof(ts_t) * (size_t)(len ? len : 1));
        uint64_t nc = 0;
        for (uint64_t i = group_off[g]; i < group_off[g + 1]; ++i) {
            uint64_t lo = 0, hi = n_map;
            while (lo < hi) {
                uint64_t mid = (lo + hi) / 2;
                if (m[mid].doc < group_doc[i]) lo = mid + 1; else hi = mid;
            }
            if (lo < n_map && m[lo].doc == group_doc[i] && !isnan(m[lo].score)) c[nc++] = m[lo];
        }
        qsort(c, (size_t)nc, sizeof(ts_t), ts_cmp);
        const uint32_t k = (uint32_t)(nc < max_results ? nc : max_results);
        for (uint32_t i = 0; i < k; ++i) {
            out_doc[(size_t)g * max_results + i] = c[i].doc;
            out_score[(size_t)g * max_results + i] = c[i].score;
        }
        out_n[g] = k;
        free(c);
    }
    free(m);
}
