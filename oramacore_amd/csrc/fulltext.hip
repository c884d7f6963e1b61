// fulltext.hip — BM25F scoring (K3), hybrid combine (K5) and top-n entry points.
//
// Two seams:
//   (i)  orama_bm25_score / orama_hybrid_combine / orama_top_n: the caller still owns the postings
//        (third-party StringStorage) and hands over exactly what the in-tree loop consumes
//        (token_score.rs:257-302); data crosses PCIe per call.
//   (ii) orama_post_*: postings live in HBM; a query is a handful of (token, list, boost) references
//        and never moves postings over PCIe.
// Both run the same kernels (bm25_kernels.hip) and the same top-k (select.hip).
#include <algorithm>
#include <cmath>
#include <shared_mutex>
#include <vector>

#include "bm25_kernels.hpp"
#include "common.hpp"
#include "select.hpp"

using namespace orama;

namespace {

// reserve + zero on (re)allocation: epoch-stamped arrays must never expose uninitialised epochs
int reserve_zeroed(DevBuf& b, size_t bytes, hipStream_t s) {
    if (bytes <= b.cap) return ORAMA_OK;
    ORAMA_TRY(b.reserve(bytes + bytes / 8));
    ORAMA_HIP_TRY(hipMemsetAsync(b.p, 0, b.cap, s));
    return ORAMA_OK;
}

struct QueryBuffers {
    Bm25State* state = nullptr;
    uint32_t* touched = nullptr;
    float* cand_score = nullptr;
    uint32_t* cand_idx = nullptr;
    unsigned long long* acc = nullptr;
    uint32_t* seen = nullptr;
    unsigned long long* emit = nullptr;
    uint32_t epoch = 0;
};

// Carve the per-query device buffers out of the scratch set and start a new epoch.
int prepare_query(Scratch* sc, uint64_t n_docs, uint32_t n_tokens, uint64_t cand_cap, QueryBuffers* qb) {
    hipStream_t s = sc->stream;
    const bool grew = (size_t)n_tokens * n_docs * 8 > sc->bm25_acc.cap || (size_t)n_docs * 4 > sc->bm25_seen.cap ||
                      (size_t)n_docs * 8 > sc->bm25_emit.cap;
    ORAMA_TRY(reserve_zeroed(sc->bm25_acc, (size_t)n_tokens * n_docs * 8, s));
    ORAMA_TRY(reserve_zeroed(sc->bm25_seen, (size_t)n_docs * 4, s));
    ORAMA_TRY(reserve_zeroed(sc->bm25_emit, (size_t)n_docs * 8, s));
    (void)grew;
    if (sc->bm25_epoch == 0xffffffffu) {  // wrap: forget every stamp
        ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_acc.p, 0, sc->bm25_acc.cap, s));
        ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_seen.p, 0, sc->bm25_seen.cap, s));
        ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_emit.p, 0, sc->bm25_emit.cap, s));
        sc->bm25_epoch = 0;
    }
    qb->epoch = ++sc->bm25_epoch;
    ORAMA_TRY(sc->misc2.reserve(sizeof(Bm25State)));
    ORAMA_TRY(sc->misc3.reserve((size_t)(cand_cap ? cand_cap : 1) * 4));      // touched
    ORAMA_TRY(sc->misc4.reserve((size_t)(cand_cap ? cand_cap : 1) * 4));      // cand_score
    ORAMA_TRY(sc->misc5.reserve((size_t)(cand_cap ? cand_cap : 1) * 4));      // cand_idx
    ORAMA_TRY(sc->h_misc.reserve(sizeof(Bm25State) + 4096));
    Bm25State* hs = sc->h_misc.as<Bm25State>();
    memset(hs, 0, sizeof(Bm25State));
    hs->min_key = 0xffffffffu;
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc2.p, hs, sizeof(Bm25State), hipMemcpyHostToDevice, s));
    qb->state = sc->misc2.as<Bm25State>();
    qb->touched = sc->misc3.as<uint32_t>();
    qb->cand_score = sc->misc4.as<float>();
    qb->cand_idx = sc->misc5.as<uint32_t>();
    qb->acc = sc->bm25_acc.as<unsigned long long>();
    qb->seen = sc->bm25_seen.as<uint32_t>();
    qb->emit = sc->bm25_emit.as<unsigned long long>();
    return ORAMA_OK;
}

// top-k over the candidate list + download of (ids, scores, n, count)
int select_and_download(orama_ctx* ctx, Scratch* sc, const QueryBuffers& qb, const uint64_t* d_docs,
                        uint32_t cand_cap, uint32_t top_k, uint64_t* out_ids, float* out_scores,
                        uint32_t* out_n, uint64_t* out_count) {
    hipStream_t s = sc->stream;
    const uint32_t kk = top_k ? top_k : 1;
    ORAMA_TRY(sc->out_ids.reserve((size_t)kk * 8));
    ORAMA_TRY(sc->out_val.reserve((size_t)kk * 4));
    ORAMA_TRY(sc->out_n.reserve(4));
    if (top_k) {
        ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState)));
        ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * kk));
        SelectPlan p;
        p.vals = qb.cand_score;
        p.idx = qb.cand_idx;
        p.stride = 0;
        p.n_dev = &qb.state->cand_count;
        p.n = cand_cap;
        p.q = 1;
        p.k = top_k;
        p.descending = true;
        p.id_map = d_docs;
        p.state = sc->sel_state.as<SelectState>();
        p.keys = sc->sel_keys.as<unsigned long long>();
        p.out_ids = sc->out_ids.as<uint64_t>();
        p.out_val = sc->out_val.as<float>();
        p.out_n = sc->out_n.as<uint32_t>();
        ORAMA_TRY(launch_select(ctx, p, s));
    }
    ORAMA_TRY(sc->h_out.reserve((size_t)kk * 12 + 4 + sizeof(Bm25State)));
    char* h = sc->h_out.as<char>();
    if (top_k) {
        ORAMA_HIP_TRY(hipMemcpyAsync(h, sc->out_ids.p, (size_t)kk * 8, hipMemcpyDeviceToHost, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)kk * 8, sc->out_val.p, (size_t)kk * 4, hipMemcpyDeviceToHost, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)kk * 12, sc->out_n.p, 4, hipMemcpyDeviceToHost, s));
    }
    ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)kk * 12 + 4, qb.state, sizeof(Bm25State), hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    const Bm25State* st = reinterpret_cast<const Bm25State*>(h + (size_t)kk * 12 + 4);
    if (out_count) *out_count = st->cand_count;
    uint32_t cnt = 0;
    if (top_k) {
        cnt = *reinterpret_cast<uint32_t*>(h + (size_t)kk * 12);
        memcpy(out_ids, h, (size_t)cnt * 8);
        memcpy(out_scores, h + (size_t)kk * 8, (size_t)cnt * 4);
    }
    *out_n = cnt;
    return ORAMA_OK;
}

// fold(0.0, f32::max/min) over the (small) vector map on the host — token_score.rs:398-401
void vec_min_max(const float* v, uint32_t n, float* mn, float* mx) {
    float a = 0.0f, b = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        if (std::isnan(v[i])) continue;
        if (v[i] > a) a = v[i];
        if (v[i] < b) b = v[i];
    }
    *mx = a;
    *mn = b;
}

int check_params(const orama_bm25_params* p) {
    ORAMA_REQUIRE(p, "null params");
    ORAMA_REQUIRE(p->n_tokens >= 1 && p->n_tokens <= kMaxTokens, "n_tokens %u outside [1, %u]", p->n_tokens,
                  kMaxTokens);
    ORAMA_REQUIRE(p->top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", p->top_k, kSelectMaxK);
    return ORAMA_OK;
}

}  // namespace

// ================================================================= resident postings (seam ii)
struct orama_post {
    orama_ctx* ctx = nullptr;
    std::shared_mutex mu;
    uint64_t n_docs = 0;
    uint32_t n_fields = 0;
    uint32_t n_lists = 0;
    uint64_t n_postings = 0;
    std::vector<uint64_t> h_docs;
    std::vector<float> avg_len;
    std::vector<uint32_t> field_of_list;
    std::vector<uint64_t> list_off;
    DevBuf d_docs, d_post_doc, d_post_val, d_omc, d_idf;
    bool has_omc = false;
    float idf_total_docs = -1.0f;  // the N the idf table was built for
    std::mutex idf_mu;
};

namespace {

// idf[df] = ln_1p((N - df + 0.5) / (df + 0.5)) via the host libm (bm25.rs:78-82), df in [0, n_docs]
int ensure_idf_table(orama_post* p, float total_documents, hipStream_t s) {
    std::lock_guard<std::mutex> g(p->idf_mu);
    if (p->idf_total_docs == total_documents && p->d_idf.p) return ORAMA_OK;
    const uint64_t n = p->n_docs + 1;
    std::vector<float> t((size_t)n);
    for (uint64_t df = 0; df < n; ++df) {
        const float d = (float)df;
        const float ratio = (total_documents - d + 0.5f) / (d + 0.5f);
        t[(size_t)df] = log1pf(ratio);
    }
    ORAMA_TRY(p->d_idf.reserve((size_t)n * 4));
    ORAMA_HIP_TRY(hipMemcpyAsync(p->d_idf.p, t.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    p->idf_total_docs = total_documents;
    return ORAMA_OK;
}

int post_search_impl(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                     const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                     const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, bool hybrid,
                     int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                     uint64_t* out_count) {
    ORAMA_REQUIRE(p && out_n, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_TRY(check_params(params));
    ORAMA_REQUIRE(n_refs == 0 || refs, "null refs");
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_REQUIRE(!hybrid || n_vec == 0 || (vec_doc && vec_score), "null vector map");
    ORAMA_HIP_TRY(hipSetDevice(p->ctx->device));
    std::shared_lock<std::shared_mutex> lk(p->mu);
    ORAMA_REQUIRE(p->n_docs > 0 || n_refs == 0, "postings store is empty (orama_post_build not called)");

    // group references by entry rank (position inside their token); rank r of every token forms one launch
    std::vector<uint32_t> rank(n_refs, 0), per_token(kMaxTokens, 0);
    uint32_t max_rank = 0;
    uint64_t total_postings = 0;
    for (uint32_t i = 0; i < n_refs; ++i) {
        ORAMA_REQUIRE(refs[i].token < params->n_tokens, "ref %u: token %u >= n_tokens %u", i, refs[i].token,
                      params->n_tokens);
        ORAMA_REQUIRE(refs[i].list < p->n_lists, "ref %u: list %u out of range", i, refs[i].list);
        rank[i] = per_token[refs[i].token]++;
        max_rank = std::max(max_rank, rank[i] + 1);
        total_postings += p->list_off[refs[i].list + 1] - p->list_off[refs[i].list];
    }
    ORAMA_REQUIRE(total_postings < 0xffffffffull, "query touches too many postings");

    // map the vector map to local doc indices (host, <= limit entries)
    std::vector<uint32_t> vidx(n_vec);
    for (uint32_t j = 0; j < n_vec; ++j) {
        auto it = std::lower_bound(p->h_docs.begin(), p->h_docs.end(), vec_doc[j]);
        ORAMA_REQUIRE(it != p->h_docs.end() && *it == vec_doc[j],
                      "hybrid: vector hit doc %llu is not a document of this index", (unsigned long long)vec_doc[j]);
        vidx[j] = (uint32_t)(it - p->h_docs.begin());
    }

    ScratchLease sc(p->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t touched_cap = std::min<uint64_t>(total_postings, p->n_docs);
    const uint64_t cand_cap = touched_cap + n_vec;
    QueryBuffers qb;
    ORAMA_TRY(prepare_query(sc.s.get(), p->n_docs ? p->n_docs : 1, params->n_tokens, cand_cap, &qb));
    ORAMA_TRY(ensure_idf_table(p, params->total_documents, s));

    // segments, grouped by rank
    std::vector<Bm25Seg> segs;
    std::vector<uint32_t> rank_begin(max_rank + 1, 0);
    std::vector<uint64_t> rank_total(max_rank, 0);
    segs.reserve(n_refs);
    for (uint32_t r = 0; r < max_rank; ++r) {
        rank_begin[r] = (uint32_t)segs.size();
        uint64_t virt = 0;
        for (uint32_t i = 0; i < n_refs; ++i) {
            if (rank[i] != r) continue;
            const uint32_t l = refs[i].list;
            Bm25Seg g{};
            g.post_begin = p->list_off[l];
            g.virt_begin = virt;
            g.len = (uint32_t)(p->list_off[l + 1] - p->list_off[l]);
            g.token = refs[i].token;
            g.boost = refs[i].boost;
            g.avg_len = p->avg_len[p->field_of_list[l]];
            if (g.len == 0) continue;
            virt += g.len;
            segs.push_back(g);
        }
        rank_total[r] = virt;
    }
    rank_begin[max_rank] = (uint32_t)segs.size();

    const uint64_t* d_allow = nullptr;
    if (allow_bitmap) {
        const size_t words = (size_t)((bitmap_bits + 63) / 64);
        ORAMA_TRY(sc->bitmap.reserve(std::max<size_t>(8, words * 8)));
        if (words) ORAMA_HIP_TRY(hipMemcpyAsync(sc->bitmap.p, allow_bitmap, words * 8, hipMemcpyHostToDevice, s));
        d_allow = sc->bitmap.as<uint64_t>();
    }
    // one H2D for the query descriptor: segments + vector entries
    const size_t seg_bytes = segs.size() * sizeof(Bm25Seg);
    const size_t vec_off = (seg_bytes + 15) & ~(size_t)15;
    const size_t desc_bytes = vec_off + (size_t)n_vec * 8;
    ORAMA_TRY(sc->h_in.reserve(desc_bytes + 16));
    ORAMA_TRY(sc->misc0.reserve(desc_bytes + 16));
    char* hd = sc->h_in.as<char>();
    if (seg_bytes) memcpy(hd, segs.data(), seg_bytes);
    if (n_vec) {
        memcpy(hd + vec_off, vidx.data(), (size_t)n_vec * 4);
        memcpy(hd + vec_off + (size_t)n_vec * 4, vec_score, (size_t)n_vec * 4);
    }
    if (desc_bytes) ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, hd, desc_bytes, hipMemcpyHostToDevice, s));

    for (uint32_t r = 0; r < max_rank; ++r) {
        Bm25Accum a;
        a.post_doc = p->d_post_doc.as<uint32_t>();
        a.post_val = p->d_post_val.as<uint32_t>();
        a.segs = sc->misc0.as<Bm25Seg>() + rank_begin[r];
        a.n_segs = rank_begin[r + 1] - rank_begin[r];
        a.total = rank_total[r];
        a.precomputed = false;
        a.b = b;
        a.docs = p->d_docs.as<uint64_t>();
        a.allow = d_allow;
        a.allow_bits = bitmap_bits;
        a.epoch = qb.epoch;
        a.n_docs = p->n_docs;
        a.acc = qb.acc;
        a.seen = qb.seen;
        a.touched = qb.touched;
        a.state = qb.state;
        ORAMA_TRY(launch_bm25_accumulate(p->ctx, a, s));
    }
    const float* omc = (apply_omc && p->has_omc) ? p->d_omc.as<float>() : nullptr;
    Bm25Finalize f;
    f.n_tokens = params->n_tokens;
    f.k = params->k;
    f.idf_table = p->d_idf.as<float>();
    f.use_threshold = params->use_threshold != 0;
    f.threshold = params->threshold;
    f.track_minmax = hybrid;
    f.omc_dense = hybrid ? nullptr : omc;  // hybrid: OMC after the combine
    f.epoch = qb.epoch;
    f.n_docs = p->n_docs;
    f.acc = qb.acc;
    f.touched = qb.touched;
    f.touched_cap = (uint32_t)touched_cap;
    f.state = qb.state;
    f.cand_score = qb.cand_score;
    f.cand_idx = qb.cand_idx;
    f.emit = qb.emit;
    ORAMA_TRY(launch_bm25_finalize(p->ctx, f, s));
    if (hybrid) {
        HybridCombine h;
        vec_min_max(vec_score, n_vec, &h.vec_min, &h.vec_max);
        h.vec_idx = reinterpret_cast<const uint32_t*>(sc->misc0.as<char>() + vec_off);
        h.vec_score = reinterpret_cast<const float*>(sc->misc0.as<char>() + vec_off + (size_t)n_vec * 4);
        h.n_vec = n_vec;
        h.omc_dense = omc;
        h.epoch = qb.epoch;
        h.cand_cap = (uint32_t)touched_cap;
        h.state = qb.state;
        h.cand_score = qb.cand_score;
        h.cand_idx = qb.cand_idx;
        h.emit = qb.emit;
        ORAMA_TRY(launch_hybrid_combine(p->ctx, h, s));
    }
    return select_and_download(p->ctx, sc.s.get(), qb, p->d_docs.as<uint64_t>(), (uint32_t)cand_cap, params->top_k,
                               out_ids, out_scores, out_n, out_count);
}

}  // namespace

extern "C" {

int orama_post_create(orama_ctx* ctx, orama_post** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    orama_post* p = new (std::nothrow) orama_post();
    if (!p) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    p->ctx = ctx;
    *out = p;
    return ORAMA_OK;
}

void orama_post_destroy(orama_post* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
}

int orama_post_build(orama_post* p, const uint64_t* docs, uint64_t n_docs, uint32_t n_fields,
                     const float* avg_field_len, uint32_t n_lists, const uint32_t* field_of_list,
                     const uint64_t* list_off, const uint64_t* post_doc, const uint32_t* post_tf,
                     const uint32_t* post_len) {
    ORAMA_REQUIRE(p, "null handle");
    ORAMA_REQUIRE(n_docs == 0 || docs, "null docs");
    ORAMA_REQUIRE(n_docs < 0xffffffffull, "postings store limited to 2^32-1 documents");
    ORAMA_REQUIRE(n_fields == 0 || avg_field_len, "null avg_field_len");
    ORAMA_REQUIRE(n_lists == 0 || (field_of_list && list_off), "null list table");
    for (uint64_t i = 1; i < n_docs; ++i)
        ORAMA_REQUIRE(docs[i - 1] < docs[i], "docs must be strictly ascending (position %llu)", (unsigned long long)i);
    const uint64_t n_post = n_lists ? list_off[n_lists] : 0;
    ORAMA_REQUIRE(n_post == 0 || (post_doc && post_tf && post_len), "null postings");
    ORAMA_HIP_TRY(hipSetDevice(p->ctx->device));
    std::unique_lock<std::shared_mutex> lk(p->mu);
    const bool dense = n_docs > 0 && docs[n_docs - 1] - docs[0] == n_docs - 1;
    std::vector<uint32_t> pd((size_t)n_post), pv((size_t)n_post);
    for (uint32_t l = 0; l < n_lists; ++l) {
        ORAMA_REQUIRE(field_of_list[l] < n_fields, "list %u: field %u out of range", l, field_of_list[l]);
        ORAMA_REQUIRE(list_off[l] <= list_off[l + 1], "list offsets must be non-decreasing");
        uint64_t prev = 0;
        for (uint64_t i = list_off[l]; i < list_off[l + 1]; ++i) {
            const uint64_t d = post_doc[i];
            ORAMA_REQUIRE(i == list_off[l] || d > prev, "list %u: docs must be strictly ascending", l);
            prev = d;
            uint64_t local;
            if (dense) {
                ORAMA_REQUIRE(d >= docs[0] && d <= docs[n_docs - 1], "list %u: doc %llu not in docs", l,
                              (unsigned long long)d);
                local = d - docs[0];
            } else {
                const uint64_t* it = std::lower_bound(docs, docs + n_docs, d);
                ORAMA_REQUIRE(it != docs + n_docs && *it == d, "list %u: doc %llu not in docs", l,
                              (unsigned long long)d);
                local = (uint64_t)(it - docs);
            }
            ORAMA_REQUIRE(post_tf[i] <= 0xffffu && post_len[i] <= 0xffffu,
                          "tf / field_length must fit u16 (IndexedValue::new(field_length: u16, ..))");
            pd[(size_t)i] = (uint32_t)local;
            pv[(size_t)i] = (post_tf[i] << 16) | post_len[i];
        }
    }
    ORAMA_TRY(p->d_docs.reserve(std::max<size_t>(8, (size_t)n_docs * 8)));
    ORAMA_TRY(p->d_post_doc.reserve(std::max<size_t>(4, (size_t)n_post * 4)));
    ORAMA_TRY(p->d_post_val.reserve(std::max<size_t>(4, (size_t)n_post * 4)));
    if (n_docs) ORAMA_HIP_TRY(hipMemcpy(p->d_docs.p, docs, (size_t)n_docs * 8, hipMemcpyHostToDevice));
    if (n_post) {
        ORAMA_HIP_TRY(hipMemcpy(p->d_post_doc.p, pd.data(), (size_t)n_post * 4, hipMemcpyHostToDevice));
        ORAMA_HIP_TRY(hipMemcpy(p->d_post_val.p, pv.data(), (size_t)n_post * 4, hipMemcpyHostToDevice));
    }
    p->n_docs = n_docs;
    p->n_fields = n_fields;
    p->n_lists = n_lists;
    p->n_postings = n_post;
    p->h_docs.assign(docs, docs + n_docs);
    p->avg_len.assign(avg_field_len, avg_field_len + n_fields);
    p->field_of_list.assign(field_of_list, field_of_list + n_lists);
    p->list_off.assign(list_off, list_off + (n_lists ? n_lists + 1 : 0));
    if (!n_lists) p->list_off.assign(1, 0);
    p->has_omc = false;
    p->idf_total_docs = -1.0f;
    return ORAMA_OK;
}

int orama_post_set_omc(orama_post* p, const uint64_t* omc_doc, const float* omc_mul, uint64_t n) {
    ORAMA_REQUIRE(p, "null handle");
    ORAMA_REQUIRE(n == 0 || (omc_doc && omc_mul), "null argument");
    ORAMA_HIP_TRY(hipSetDevice(p->ctx->device));
    std::unique_lock<std::shared_mutex> lk(p->mu);
    if (n == 0) {
        p->has_omc = false;
        return ORAMA_OK;
    }
    std::vector<float> dense((size_t)p->n_docs, 1.0f);  // x * 1.0 == x bit-for-bit: absent docs untouched
    for (uint64_t i = 0; i < n; ++i) {
        auto it = std::lower_bound(p->h_docs.begin(), p->h_docs.end(), omc_doc[i]);
        if (it == p->h_docs.end() || *it != omc_doc[i]) continue;  // multiplier of a doc not in this index
        dense[(size_t)(it - p->h_docs.begin())] = omc_mul[i];
    }
    ORAMA_TRY(p->d_omc.reserve(std::max<size_t>(4, dense.size() * 4)));
    if (!dense.empty())
        ORAMA_HIP_TRY(hipMemcpy(p->d_omc.p, dense.data(), dense.size() * 4, hipMemcpyHostToDevice));
    p->has_omc = true;
    return ORAMA_OK;
}

int orama_post_search(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                      const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                      int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                      uint64_t* out_count) {
    return post_search_impl(p, refs, n_refs, b, params, allow_bitmap, bitmap_bits, nullptr, nullptr, 0, false,
                            apply_omc, out_ids, out_scores, out_n, out_count);
}

int orama_post_search_hybrid(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                             const orama_bm25_params* params, const uint64_t* allow_bitmap,
                             uint64_t bitmap_bits, const uint64_t* vec_doc, const float* vec_score,
                             uint32_t n_vec, int apply_omc, uint64_t* out_ids, float* out_scores,
                             uint32_t* out_n, uint64_t* out_count) {
    return post_search_impl(p, refs, n_refs, b, params, allow_bitmap, bitmap_bits, vec_doc, vec_score, n_vec, true,
                            apply_omc, out_ids, out_scores, out_n, out_count);
}

// ================================================================= seam (i): host-provided contributions
int orama_bm25_score(orama_ctx* ctx, const orama_ntf_entry* entries, uint32_t n_entries,
                     const orama_bm25_params* params, const uint64_t* omc_doc, const float* omc_mul,
                     uint64_t n_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                     uint64_t* out_count) {
    ORAMA_REQUIRE(ctx && out_n, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_TRY(check_params(params));
    ORAMA_REQUIRE(n_entries == 0 || entries, "null entries");
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    uint64_t total = 0, max_id = 0;
    for (uint32_t e = 0; e < n_entries; ++e) {
        ORAMA_REQUIRE(entries[e].token < params->n_tokens, "entry %u: token out of range", e);
        ORAMA_REQUIRE(entries[e].len == 0 || (entries[e].doc && entries[e].ntf), "entry %u: null arrays", e);
        total += entries[e].len;
        for (uint64_t i = 0; i < entries[e].len; ++i) max_id = std::max(max_id, entries[e].doc[i]);
    }
    ORAMA_REQUIRE(total < 0xffffffffull, "too many postings");
    if (total == 0) return ORAMA_OK;
    // local doc space: identity when ids are reasonably dense (the reference assigns sequential u64
    // ids, write/collection_document_storage.rs:73-77), otherwise the sorted set of ids that occur.
    std::vector<uint64_t> docs;
    const bool identity = max_id < 0xfffffff0ull && max_id <= 8 * total + (1u << 20);
    if (!identity) {
        docs.reserve((size_t)total);
        for (uint32_t e = 0; e < n_entries; ++e) docs.insert(docs.end(), entries[e].doc, entries[e].doc + entries[e].len);
        std::sort(docs.begin(), docs.end());
        docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
    }
    const uint64_t n_docs = identity ? max_id + 1 : docs.size();
    ScratchLease sc(ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t touched_cap = std::min<uint64_t>(total, n_docs);
    QueryBuffers qb;
    ORAMA_TRY(prepare_query(sc.s.get(), n_docs, params->n_tokens, touched_cap, &qb));

    // pack postings (local doc, ntf bits) + segments grouped by rank + doc table
    std::vector<uint32_t> rank(n_entries, 0), per_token(kMaxTokens, 0);
    uint32_t max_rank = 0;
    for (uint32_t e = 0; e < n_entries; ++e) {
        rank[e] = per_token[entries[e].token]++;
        max_rank = std::max(max_rank, rank[e] + 1);
    }
    const size_t post_bytes = (size_t)total * 4;
    const size_t seg_off = ((2 * post_bytes) + 15) & ~(size_t)15;
    const size_t doc_off = (seg_off + (size_t)n_entries * sizeof(Bm25Seg) + 15) & ~(size_t)15;
    const size_t omc_off = doc_off + (size_t)n_docs * 8;
    const size_t all_bytes = omc_off + (size_t)n_omc * 8;
    ORAMA_TRY(sc->h_in.reserve(all_bytes + 64));
    ORAMA_TRY(sc->misc0.reserve(all_bytes + 64));
    char* hb = sc->h_in.as<char>();
    uint32_t* h_pd = reinterpret_cast<uint32_t*>(hb);
    uint32_t* h_pv = reinterpret_cast<uint32_t*>(hb + post_bytes);
    Bm25Seg* h_seg = reinterpret_cast<Bm25Seg*>(hb + seg_off);
    uint64_t* h_docs = reinterpret_cast<uint64_t*>(hb + doc_off);
    std::vector<uint32_t> rank_begin(max_rank + 1, 0);
    std::vector<uint64_t> rank_total(max_rank, 0);
    uint64_t cursor = 0;
    uint32_t nseg = 0;
    for (uint32_t r = 0; r < max_rank; ++r) {
        rank_begin[r] = nseg;
        uint64_t virt = 0;
        for (uint32_t e = 0; e < n_entries; ++e) {
            if (rank[e] != r || entries[e].len == 0) continue;
            Bm25Seg g{};
            g.post_begin = cursor;
            g.virt_begin = virt;
            g.len = (uint32_t)entries[e].len;
            g.token = entries[e].token;
            g.boost = 1.0f;
            g.avg_len = 1.0f;
            uint64_t prev = 0;
            for (uint64_t i = 0; i < entries[e].len; ++i) {
                const uint64_t d = entries[e].doc[i];
                uint32_t local;
                if (identity) {
                    local = (uint32_t)d;
                } else {
                    local = (uint32_t)(std::lower_bound(docs.begin(), docs.end(), d) - docs.begin());
                }
                (void)prev;
                h_pd[cursor + i] = local;
                memcpy(&h_pv[cursor + i], &entries[e].ntf[i], 4);
            }
            cursor += entries[e].len;
            virt += entries[e].len;
            h_seg[nseg++] = g;
        }
        rank_total[r] = virt;
    }
    rank_begin[max_rank] = nseg;
    if (identity) {
        for (uint64_t i = 0; i < n_docs; ++i) h_docs[i] = i;
    } else {
        memcpy(h_docs, docs.data(), (size_t)n_docs * 8);
    }
    // sparse OMC list → (local idx, multiplier), entries of unknown docs dropped
    uint32_t* h_oidx = reinterpret_cast<uint32_t*>(hb + omc_off);
    float* h_omul = reinterpret_cast<float*>(hb + omc_off + (size_t)n_omc * 4);
    uint32_t n_o = 0;
    for (uint64_t i = 0; i < n_omc; ++i) {
        uint64_t local;
        if (identity) {
            if (omc_doc[i] >= n_docs) continue;
            local = omc_doc[i];
        } else {
            auto it = std::lower_bound(docs.begin(), docs.end(), omc_doc[i]);
            if (it == docs.end() || *it != omc_doc[i]) continue;
            local = (uint64_t)(it - docs.begin());
        }
        h_oidx[n_o] = (uint32_t)local;
        h_omul[n_o] = omc_mul[i];
        ++n_o;
    }
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, hb, all_bytes, hipMemcpyHostToDevice, s));
    char* db = sc->misc0.as<char>();
    for (uint32_t r = 0; r < max_rank; ++r) {
        Bm25Accum a;
        a.post_doc = reinterpret_cast<const uint32_t*>(db);
        a.post_val = reinterpret_cast<const uint32_t*>(db + post_bytes);
        a.segs = reinterpret_cast<const Bm25Seg*>(db + seg_off) + rank_begin[r];
        a.n_segs = rank_begin[r + 1] - rank_begin[r];
        a.total = rank_total[r];
        a.precomputed = true;
        a.epoch = qb.epoch;
        a.n_docs = n_docs;
        a.acc = qb.acc;
        a.seen = qb.seen;
        a.touched = qb.touched;
        a.state = qb.state;
        ORAMA_TRY(launch_bm25_accumulate(ctx, a, s));
    }
    // df → host → idf (libm log1pf, bm25.rs:78-82) → device
    ORAMA_TRY(sc->h_out.reserve(sizeof(Bm25State) + kMaxTokens * 4));
    Bm25State* hst = sc->h_out.as<Bm25State>();
    ORAMA_HIP_TRY(hipMemcpyAsync(hst, qb.state, sizeof(Bm25State), hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    float* h_idf = reinterpret_cast<float*>(sc->h_out.as<char>() + sizeof(Bm25State));
    for (uint32_t t = 0; t < params->n_tokens; ++t) {
        const float df = (float)std::max<uint32_t>(hst->df[t], 1u);
        h_idf[t] = log1pf((params->total_documents - df + 0.5f) / (df + 0.5f));
    }
    ORAMA_TRY(sc->misc1.reserve(kMaxTokens * 4));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc1.p, h_idf, (size_t)params->n_tokens * 4, hipMemcpyHostToDevice, s));
    Bm25Finalize f;
    f.n_tokens = params->n_tokens;
    f.k = params->k;
    f.idf_vals = sc->misc1.as<float>();
    f.use_threshold = params->use_threshold != 0;
    f.threshold = params->threshold;
    f.epoch = qb.epoch;
    f.n_docs = n_docs;
    f.acc = qb.acc;
    f.touched = qb.touched;
    f.touched_cap = (uint32_t)touched_cap;
    f.state = qb.state;
    f.cand_score = qb.cand_score;
    f.cand_idx = qb.cand_idx;
    f.emit = qb.emit;
    ORAMA_TRY(launch_bm25_finalize(ctx, f, s));
    if (n_o)
        ORAMA_TRY(launch_omc_sparse(reinterpret_cast<const uint32_t*>(db + omc_off),
                                    reinterpret_cast<const float*>(db + omc_off + (size_t)n_omc * 4), n_o, qb.epoch,
                                    qb.emit, qb.cand_score, s));
    return select_and_download(ctx, sc.s.get(), qb, reinterpret_cast<const uint64_t*>(db + doc_off),
                               (uint32_t)touched_cap, params->top_k, out_ids, out_scores, out_n, out_count);
}

int orama_hybrid_combine(orama_ctx* ctx, const uint64_t* vec_doc, const float* vec_score, uint64_t n_vec,
                         const uint64_t* ft_doc, const float* ft_score, uint64_t n_ft, uint32_t top_k,
                         uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    ORAMA_REQUIRE(ctx && out_n, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_REQUIRE(top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", top_k, kSelectMaxK);
    ORAMA_REQUIRE(n_vec == 0 || (vec_doc && vec_score), "null vector map");
    ORAMA_REQUIRE(n_ft == 0 || (ft_doc && ft_score), "null fulltext map");
    ORAMA_REQUIRE(top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_REQUIRE(n_vec + n_ft < 0xffffffffull, "maps too large");
    if (n_vec + n_ft == 0) return ORAMA_OK;
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    // local doc space = sorted union of both key sets
    std::vector<uint64_t> docs;
    docs.reserve((size_t)(n_vec + n_ft));
    docs.insert(docs.end(), vec_doc, vec_doc + n_vec);
    docs.insert(docs.end(), ft_doc, ft_doc + n_ft);
    std::sort(docs.begin(), docs.end());
    docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
    const uint64_t n_docs = docs.size();
    ScratchLease sc(ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t cand_cap = n_ft + n_vec;
    QueryBuffers qb;
    ORAMA_TRY(prepare_query(sc.s.get(), n_docs, 1, cand_cap, &qb));
    const size_t ft_off = 0, vec_off = (size_t)n_ft * 8, doc_off = (vec_off + (size_t)n_vec * 8 + 15) & ~(size_t)15;
    const size_t all_bytes = doc_off + (size_t)n_docs * 8;
    ORAMA_TRY(sc->h_in.reserve(all_bytes + 16));
    ORAMA_TRY(sc->misc0.reserve(all_bytes + 16));
    char* hb = sc->h_in.as<char>();
    uint32_t* h_fidx = reinterpret_cast<uint32_t*>(hb + ft_off);
    float* h_fsc = reinterpret_cast<float*>(hb + ft_off + (size_t)n_ft * 4);
    uint32_t* h_vidx = reinterpret_cast<uint32_t*>(hb + vec_off);
    float* h_vsc = reinterpret_cast<float*>(hb + vec_off + (size_t)n_vec * 4);
    for (uint64_t i = 0; i < n_ft; ++i) {
        h_fidx[i] = (uint32_t)(std::lower_bound(docs.begin(), docs.end(), ft_doc[i]) - docs.begin());
        h_fsc[i] = ft_score[i];
    }
    for (uint64_t i = 0; i < n_vec; ++i) {
        h_vidx[i] = (uint32_t)(std::lower_bound(docs.begin(), docs.end(), vec_doc[i]) - docs.begin());
        h_vsc[i] = vec_score[i];
    }
    memcpy(hb + doc_off, docs.data(), (size_t)n_docs * 8);
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, hb, all_bytes, hipMemcpyHostToDevice, s));
    char* db = sc->misc0.as<char>();
    if (n_ft) {
        ORAMA_HIP_TRY(hipMemcpyAsync(qb.cand_idx, db + ft_off, (size_t)n_ft * 4, hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(qb.cand_score, db + ft_off + (size_t)n_ft * 4, (size_t)n_ft * 4,
                                     hipMemcpyDeviceToDevice, s));
    }
    ORAMA_TRY(launch_hybrid_ingest(ctx, (uint32_t)n_ft, qb.epoch, qb.cand_idx, qb.cand_score, qb.emit, qb.state, s));
    HybridCombine h;
    vec_min_max(vec_score, (uint32_t)n_vec, &h.vec_min, &h.vec_max);
    h.vec_idx = reinterpret_cast<const uint32_t*>(db + vec_off);
    h.vec_score = reinterpret_cast<const float*>(db + vec_off + (size_t)n_vec * 4);
    h.n_vec = (uint32_t)n_vec;
    h.epoch = qb.epoch;
    h.cand_cap = (uint32_t)n_ft;
    h.state = qb.state;
    h.cand_score = qb.cand_score;
    h.cand_idx = qb.cand_idx;
    h.emit = qb.emit;
    ORAMA_TRY(launch_hybrid_combine(ctx, h, s));
    return select_and_download(ctx, sc.s.get(), qb, reinterpret_cast<const uint64_t*>(db + doc_off),
                               (uint32_t)cand_cap, top_k, out_ids, out_scores, out_n, out_count);
}

// ================================================================= top_n
int orama_top_n(orama_ctx* ctx, const uint64_t* doc, const float* score, uint64_t n, uint32_t top_k,
                uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
    ORAMA_REQUIRE(ctx && out_n, "null argument");
    *out_n = 0;
    if (top_k == 0 || n == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc && score && out_ids && out_scores, "null argument");
    ORAMA_REQUIRE(top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", top_k, kSelectMaxK);
    ORAMA_REQUIRE(n < 0xffffffffull, "top_n limited to 2^32-1 entries");
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    ScratchLease sc(ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    // The device selection orders equal scores by list position; the declared rule is "doc id asc",
    // so present the list in doc order (host argsort — this standalone entry takes an arbitrary
    // HashMap-ordered list; the fused BM25/hybrid pipelines index docs in id order and skip this).
    std::vector<uint32_t> order((size_t)n);
    for (uint64_t i = 0; i < n; ++i) order[(size_t)i] = (uint32_t)i;
    bool sorted = true;
    for (uint64_t i = 1; i < n && sorted; ++i) sorted = doc[i - 1] <= doc[i];
    if (!sorted)
        std::sort(order.begin(), order.end(),
                  [&](uint32_t a, uint32_t b) { return doc[a] != doc[b] ? doc[a] < doc[b] : a < b; });
    ORAMA_TRY(sc->h_in.reserve((size_t)n * 12));
    uint64_t* hd = sc->h_in.as<uint64_t>();
    float* hs = reinterpret_cast<float*>(hd + n);
    for (uint64_t i = 0; i < n; ++i) {
        hd[i] = doc[order[(size_t)i]];
        hs[i] = score[order[(size_t)i]];
    }
    ORAMA_TRY(sc->misc0.reserve((size_t)n * 8));
    ORAMA_TRY(sc->dist.reserve((size_t)n * 4));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, hd, (size_t)n * 8, hipMemcpyHostToDevice, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->dist.p, hs, (size_t)n * 4, hipMemcpyHostToDevice, s));
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState)));
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * top_k));
    ORAMA_TRY(sc->out_ids.reserve((size_t)top_k * 8));
    ORAMA_TRY(sc->out_val.reserve((size_t)top_k * 4));
    ORAMA_TRY(sc->out_n.reserve(4));
    SelectPlan p;
    p.vals = sc->dist.as<float>();
    p.stride = n;
    p.n = (uint32_t)n;
    p.q = 1;
    p.k = top_k;
    p.descending = true;
    p.id_map = sc->misc0.as<uint64_t>();
    p.state = sc->sel_state.as<SelectState>();
    p.keys = sc->sel_keys.as<unsigned long long>();
    p.out_ids = sc->out_ids.as<uint64_t>();
    p.out_val = sc->out_val.as<float>();
    p.out_n = sc->out_n.as<uint32_t>();
    ORAMA_TRY(launch_select(ctx, p, s));
    ORAMA_TRY(sc->h_out.reserve((size_t)top_k * 12 + 4));
    char* h = sc->h_out.as<char>();
    ORAMA_HIP_TRY(hipMemcpyAsync(h, sc->out_ids.p, (size_t)top_k * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)top_k * 8, sc->out_val.p, (size_t)top_k * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)top_k * 12, sc->out_n.p, 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    const uint32_t cnt = *reinterpret_cast<uint32_t*>(h + (size_t)top_k * 12);
    memcpy(out_ids, h, (size_t)cnt * 8);
    memcpy(out_scores, h + (size_t)top_k * 8, (size_t)cnt * 4);
    *out_n = cnt;
    return ORAMA_OK;
}

}  // extern "C"
