/* orama_cpu_fast.c — the CPU BASELINE legs bench.py reports beside the GPU numbers ("-O3 -march=native +
 * all cores").  TEST / MEASUREMENT INFRASTRUCTURE ONLY, like everything under oracle/: never linked into, loaded by or
 * shipped with the product.  Built on the machine that runs the benchmark (oracle/cpu_fast.py: gcc -O3 -march=native), so
 * the host's own vector width is used.
 *
 * Unlike orama_oracle.c these functions are NOT order-exact restatements: they are what a competent CPU implementation of
 * the same path would run —
 *   cpf_distances_f32     cosine distance of one query against n rows, 8 independent fused-multiply-add accumulators per
 *                         row (one SIMD register each way: the compiler vectorises the inner loop without any
 *                         reassociation licence), rows split over `threads` pthreads.  The reference's third-party scan (embedding_field.rs:258-265) is
 *                         presumably of this kind; its answers agree with the order-exact oracle to ~1e-6.
 *   cpf_bm25_hashmap      search_full_text as the reference runs it (token_score.rs:257-300, bm25.rs:369-428, 484-520): per token
 *                         a HashSet of the documents (df), the scorer's per-token HashMap doc -> sum of ntf, finalize_term
 *                         draining it into the document HashMap — open addressing with a multiplicative hash here (cheaper than
 *                         std's SipHash: a generous baseline) — then top_n with a bounded heap (sort.rs:260-279).
 * Results of cpf_bm25_hashmap are bit-identical to the oracle's for one list per token (same additions, same order per
 * document); bench.py checks that before it quotes the rate.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vector scan */
static void distances_range(const float* corpus, uint64_t lo, uint64_t hi, uint32_t d, const float* q, float qn, float* out) {
    for (uint64_t r = lo; r < hi; ++r) {
        const float* x = corpus + r * (uint64_t)d;
        float dot[8] = {0}, nn[8] = {0};
        uint32_t k = 0;
        for (; k + 8 <= d; k += 8)
            for (int j = 0; j < 8; ++j) {
                dot[j] = __builtin_fmaf(q[k + j], x[k + j], dot[j]);
                nn[j] = __builtin_fmaf(x[k + j], x[k + j], nn[j]);
            }
        float sd = 0.0f, sn = 0.0f;
        for (int j = 0; j < 8; ++j) {
            sd += dot[j];
            sn += nn[j];
        }
        for (; k < d; ++k) {
            sd += q[k] * x[k];
            sn += x[k] * x[k];
        }
        out[r] = 1.0f - sd / (qn * sqrtf(sn));
    }
}

typedef struct {
    const float* corpus;
    uint64_t lo, hi;
    uint32_t d;
    const float* q;
    float qn;
    float* out;
} scan_job;

static void* scan_thread(void* p) {
    scan_job* j = (scan_job*)p;
    distances_range(j->corpus, j->lo, j->hi, j->d, j->q, j->qn, j->out);
    return NULL;
}

void cpf_distances_f32(const float* corpus, uint64_t n, uint32_t d, const float* q, float* out, int threads) {
    float qq = 0.0f;
    for (uint32_t k = 0; k < d; ++k) qq += q[k] * q[k];
    const float qn = sqrtf(qq);
    if (threads <= 1) {
        distances_range(corpus, 0, n, d, q, qn, out);
        return;
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    scan_job* jobs = (scan_job*)malloc(sizeof(scan_job) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (scan_job){corpus, n * (uint64_t)t / (uint64_t)threads, n * (uint64_t)(t + 1) / (uint64_t)threads, d, q, qn, out};
        pthread_create(&th[t], NULL, scan_thread, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}

/* A copy of `corpus` whose pages are FIRST TOUCHED by the threads that will scan them (the same static split of the rows as
 * cpf_distances_f32 with `threads` threads): on a multi-socket host every thread then streams from its own node's memory.
 * bench.py's sample arrives as one numpy array placed by one thread — the all-core leg peaked at 16-32 threads and FELL
 * beyond (191 GB/s on a 256-thread host).  Free with cpf_free. */
typedef struct {
    const float* src;
    float* dst;
    uint64_t lo, hi;
    uint32_t d;
} place_job;

static void* place_thread(void* p) {
    const place_job* j = (const place_job*)p;
    memcpy(j->dst + j->lo * (uint64_t)j->d, j->src + j->lo * (uint64_t)j->d, (size_t)(j->hi - j->lo) * j->d * sizeof(float));
    return NULL;
}

float* cpf_place_rows(const float* corpus, uint64_t n, uint32_t d, int threads) {
    float* dst = NULL;
    if (posix_memalign((void**)&dst, 4096, (size_t)n * d * sizeof(float) + 4096) != 0) return NULL;  /* untouched pages */
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    place_job* jobs = (place_job*)malloc(sizeof(place_job) * (size_t)threads);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (place_job){corpus, dst, n * (uint64_t)t / (uint64_t)threads, n * (uint64_t)(t + 1) / (uint64_t)threads, d};
        pthread_create(&th[t], NULL, place_thread, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
    return dst;
}

void cpf_free(void* p) { free(p); }

/* ------------------------------------------------------------------ BM25F with hash maps */
typedef struct {
    uint64_t* key;  /* doc + 1 (0 = empty) */
    float* val;
    uint32_t* mask;
    uint64_t cap, n;
} map_t;

static uint64_t pow2_at_least(uint64_t x) {
    uint64_t c = 16;
    while (c < x) c <<= 1;
    return c;
}

static void map_init(map_t* m, uint64_t expect, int with_mask) {
    m->cap = pow2_at_least(expect * 2 + 16);
    m->n = 0;
    m->key = (uint64_t*)calloc((size_t)m->cap, sizeof(uint64_t));
    m->val = (float*)malloc((size_t)m->cap * sizeof(float));
    m->mask = with_mask ? (uint32_t*)malloc((size_t)m->cap * sizeof(uint32_t)) : NULL;
}

static void map_free(map_t* m) {
    free(m->key);
    free(m->val);
    free(m->mask);
}

static inline uint64_t map_slot(const map_t* m, uint64_t doc, int* found) {
    uint64_t i = ((doc + 1) * 0x9E3779B97F4A7C15ull) >> 20 & (m->cap - 1);
    for (;;) {
        if (m->key[i] == doc + 1) {
            *found = 1;
            return i;
        }
        if (m->key[i] == 0) {
            *found = 0;
            return i;
        }
        i = (i + 1) & (m->cap - 1);
    }
}

static inline int f32_is_normal(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t e = (u >> 23) & 0xffu;
    return e != 0u && e != 0xffu;
}

typedef struct {
    uint32_t token;
    const uint64_t* doc;
    const float* ntf;
    uint64_t len;
} cpf_entry;

typedef struct {
    float score;
    uint64_t doc;
} hit_t;

/* "a is a worse hit than b": lower score, or equal score and HIGHER DocumentId (the declared tie rule) */
static inline int worse(hit_t a, hit_t b) { return a.score < b.score || (a.score == b.score && a.doc > b.doc); }

static void heap_sift_down(hit_t* h, uint64_t n, uint64_t i) { /* min-heap by `worse`: the root is the worst kept hit */
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == i) return;
        hit_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static int hit_cmp(const void* a, const void* b) {
    const hit_t* x = (const hit_t*)a;
    const hit_t* y = (const hit_t*)b;
    if (worse(*y, *x)) return -1;
    if (worse(*x, *y)) return 1;
    return 0;
}

/* returns the number of hits written (<= top_k); *out_count = documents in the score map */
uint64_t cpf_bm25_hashmap(const cpf_entry* entries, uint32_t n_entries, uint32_t n_tokens, float total_documents, float k,
                          int use_threshold, uint32_t threshold, uint64_t top_k, uint64_t* out_doc, float* out_score,
                          uint64_t* out_count) {
    uint64_t total = 0, longest = 0;
    for (uint32_t e = 0; e < n_entries; ++e) {
        total += entries[e].len;
        if (entries[e].len > longest) longest = entries[e].len;
    }
    map_t docs; /* document_scores: HashMap<DocumentId, f32> (+ the threshold scorer's mask) */
    map_init(&docs, total, 1);
    for (uint32_t t = 0; t < n_tokens; ++t) {
        uint64_t tok_total = 0;
        for (uint32_t e = 0; e < n_entries; ++e)
            if (entries[e].token == t) tok_total += entries[e].len;
        if (!tok_total) continue;
        map_t term; /* corpus_docs (HashSet) and the scorer's per-token contributions in one table: key = doc, val = sum of ntf */
        map_init(&term, tok_total, 0);
        uint64_t* order = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)tok_total); /* drain order = first-insertion order */
        for (uint32_t e = 0; e < n_entries; ++e) {
            if (entries[e].token != t) continue;
            for (uint64_t i = 0; i < entries[e].len; ++i) {
                int found;
                const uint64_t s = map_slot(&term, entries[e].doc[i], &found);
                if (!found) {
                    term.key[s] = entries[e].doc[i] + 1;
                    term.val[s] = 0.0f + 1.0f * entries[e].ntf[i];
                    order[term.n++] = s;
                } else {
                    term.val[s] = term.val[s] + 1.0f * entries[e].ntf[i];
                }
            }
        }
        float df = (float)(term.n < 1 ? 1 : term.n);
        const float idf = log1pf((total_documents - df + 0.5f) / (df + 0.5f));
        for (uint64_t j = 0; j < term.n; ++j) { /* finalize_term, bm25.rs:484-520 */
            const uint64_t s = order[j];
            const float S = term.val[s];
            if (!f32_is_normal(S)) continue;
            const float term_score = idf * (k + 1.0f) * S / (k + S);
            if (term_score != term_score) continue;
            int found;
            const uint64_t doc = term.key[s] - 1;
            const uint64_t ds = map_slot(&docs, doc, &found);
            if (!found) {
                docs.key[ds] = doc + 1;
                docs.val[ds] = 0.0f;
                docs.mask[ds] = 0u;
                docs.n++;
            }
            docs.val[ds] = docs.val[ds] + term_score * 1.0f;
            docs.mask[ds] |= 1u << (t & 31u);
        }
        free(order);
        map_free(&term);
    }
    /* count + top_n (search.rs:482, sort.rs:260-279): a heap bounded by top_k over the map's entries */
    hit_t* heap = (hit_t*)malloc(sizeof(hit_t) * (size_t)(top_k ? top_k : 1));
    uint64_t hn = 0, count = 0;
    for (uint64_t i = 0; i < docs.cap; ++i) {
        if (!docs.key[i]) continue;
        if (use_threshold && (uint32_t)__builtin_popcount(docs.mask[i]) < threshold) continue;
        ++count;
        const hit_t h = {docs.val[i], docs.key[i] - 1};
        if (h.score != h.score || top_k == 0) continue;
        if (hn < top_k) {
            heap[hn++] = h;
            if (hn == top_k)
                for (uint64_t j = hn / 2; j-- > 0;) heap_sift_down(heap, hn, j);
        } else if (worse(heap[0], h)) {
            heap[0] = h;
            heap_sift_down(heap, hn, 0);
        }
    }
    qsort(heap, (size_t)hn, sizeof(hit_t), hit_cmp);
    for (uint64_t i = 0; i < hn; ++i) {
        out_doc[i] = heap[i].doc;
        out_score[i] = heap[i].score;
    }
    free(heap);
    map_free(&docs);
    if (out_count) *out_count = count;
    return hn;
}
