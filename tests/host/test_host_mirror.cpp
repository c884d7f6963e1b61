// C++ parity tests of the host mirror (include/orama/host.hpp) against the oracle, written to read like the
// reference's own unit tests (src/collection_manager/bm25.rs:527-1043) and its search tests.
// Built and run by tests/test_host_cpp_gpu.py:  g++ -std=c++17 ... -lorama_hip -lorama_oracle
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "orama/host.hpp"
extern "C" {
#include "orama_oracle.h"
}

using namespace orama::host;

static int g_failed = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("  CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                    \
        }                                                                  \
    } while (0)
#define APPROX(a, b, tol) CHECK(std::fabs((double)(a) - (double)(b)) <= (tol))

static uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// bm25.rs:911-983 — test_canonical_bm25f_single_term_two_fields
static void test_canonical_bm25f_single_term_two_fields(Context& ctx) {
    auto scorer = BM25Scorer::plain(ctx);
    const float k = 1.2f, corpus_docs = 100.0f;
    const size_t term_docs = 10;
    const float title_ntf = orc_bm25f_normalized_tf(2, 10, 8.0f, 0.75f);
    const float content_ntf = orc_bm25f_normalized_tf(1, 200, 150.0f, 0.75f);
    scorer.reset_term();
    // the in-tree caller derives df from the postings: 10 documents hold the term in the title field
    for (DocumentId d = 1; d <= 10; ++d) scorer.add_precomputed_field(d, title_ntf, 2.0f);
    scorer.add_precomputed_field(1, content_ntf, 1.0f);
    scorer.finalize_term_plain(term_docs, corpus_docs, k, 1.0f);
    scorer.next_term();
    auto scores = scorer.get_scores();
    const float aggregated = 2.0f * title_ntf + 1.0f * content_ntf;
    const float ratio = (corpus_docs - (float)term_docs + 0.5f) / ((float)term_docs + 0.5f);
    const float idf = std::log1p(ratio);
    const float expected = idf * (k + 1.0f) * aggregated / (k + aggregated);
    CHECK(scores.size() == 10);
    APPROX(scores[1], expected, 1e-5);
    CHECK(bits(scores[1]) == bits(expected));  // same f32 expression, same libm idf → bit-identical
}

// bm25.rs:533-563 — test_bm25f_scorer_basic, through the contribution interface
static void test_bm25f_scorer_basic(Context& ctx) {
    auto scorer = BM25Scorer::plain(ctx);
    scorer.reset_term();
    const float ntf = orc_bm25f_normalized_tf(5, 100, 100.0f, 0.75f);  // == 5
    for (DocumentId d = 1; d <= 10; ++d) scorer.add_precomputed_field(d, ntf, 1.0f);
    scorer.finalize_term_plain(10, 100.0f, 1.2f, 1.0f);
    scorer.next_term();
    auto scores = scorer.get_scores();
    const float expected_idf = std::log1p((100.0f - 10.0f + 0.5f) / (10.0f + 0.5f));
    const float expected = expected_idf * (1.2f + 1.0f) * 5.0f / (1.2f + 5.0f);
    CHECK(scores.size() == 10);
    APPROX(scores[1], expected, 1e-6);
}

// threshold scorer — bm25.rs:325-429 + token_score.rs:211-218 (src/tests/fulltext_search.rs:478-600 in miniature)
static void test_threshold_scorer(Context& ctx) {
    // 3 docs; tokens: 0 in {1,2,3}, 1 in {1,2}, 2 in {1}
    for (uint32_t thr = 0; thr <= 3; ++thr) {
        auto scorer = BM25Scorer::with_threshold(ctx, thr);
        const DocumentId sets[3][3] = {{1, 2, 3}, {1, 2, 0}, {1, 0, 0}};
        for (int t = 0; t < 3; ++t) {
            scorer.reset_term();
            size_t df = 0;
            for (int i = 0; i < 3; ++i)
                if (sets[t][i]) {
                    scorer.add_precomputed_field(sets[t][i], 1.0f + 0.1f * t, 1.0f);
                    ++df;
                }
            scorer.finalize_term(df, 3.0f, 1.2f, 1.0f, 1u << t);
            scorer.next_term();
        }
        auto r = scorer.top_n(10);
        const size_t expect = thr <= 1 ? 3 : (thr == 2 ? 2 : 1);
        CHECK(r.count == expect);
        CHECK(r.hits.size() == expect);
        CHECK(r.hits[0].document_id == 1);
    }
}

// search.rs:39-48 + src/tests/omc_test.rs:485-553 — multiplicative ratios
static void test_omc_ratios(Context& ctx) {
    auto make = [&](const std::map<DocumentId, float>& omc) {
        auto scorer = BM25Scorer::plain(ctx);
        scorer.reset_term();
        for (DocumentId d = 1; d <= 6; ++d) scorer.add_precomputed_field(d, 1.0f, 1.0f);
        scorer.finalize_term_plain(6, 6.0f, 1.2f, 1.0f);
        scorer.next_term();
        return scorer.top_n(6, omc);
    };
    auto base = make({});
    auto boosted = make({{1, 0.25f}, {2, 0.5f}, {3, 2.0f}, {4, 5.0f}, {5, 10.0f}});
    CHECK(base.count == 6 && boosted.count == 6);
    const float s = base.hits[0].score;
    std::map<DocumentId, float> got;
    for (auto& h : boosted.hits) got[h.document_id] = h.score;
    APPROX(got[1] / s, 0.25, 1e-3);
    APPROX(got[2] / s, 0.5, 1e-3);
    APPROX(got[3] / s, 2.0, 1e-3);
    APPROX(got[4] / s, 5.0, 1e-3);
    APPROX(got[5] / s, 10.0, 1e-3);
    APPROX(got[6] / s, 1.0, 1e-6);
    CHECK(boosted.hits[0].document_id == 5 && boosted.hits[5].document_id == 1);
}

// embedding_field.rs:250-278 against the oracle (a1 + a2), incl. several vectors per document and the E5 rescale
static void test_embedding_field_search(Context& ctx) {
    for (Model model : {Model::BGESmall, Model::MultilingualE5Small}) {
        const size_t dim = dimensions(model), n_docs = 500;
        std::mt19937 rng(7);
        std::normal_distribution<float> nd;
        std::vector<float> q(dim);
        for (auto& x : q) x = nd(rng);
        EmbeddingFieldStorage field(ctx, model);
        std::vector<float> corpus;
        std::vector<uint64_t> row_doc;
        for (size_t d = 0; d < n_docs; ++d) {
            std::vector<std::vector<float>> vecs(1 + d % 3, std::vector<float>(dim));
            for (auto& v : vecs) {
                const float w = (d % 25 == 0) ? 0.9f : 0.1f;
                for (size_t i = 0; i < dim; ++i) v[i] = w * q[i] + (1.0f - w) * nd(rng) * 3.0f;
                corpus.insert(corpus.end(), v.begin(), v.end());
                row_doc.push_back(d + 100);
            }
            field.insert(d + 100, vecs);
        }
        field.insert(9999, {std::vector<float>(dim, 0.0f)});  // zero vector: rejected by the indexer
        CHECK(field.info().num_embeddings == row_doc.size());
        for (size_t limit : {size_t(5), size_t(60)}) {
            for (float sim : {0.0f, 0.7f}) {
                std::unordered_map<DocumentId, float> out;
                VectorSearchParams p;
                p.target = &q;
                p.similarity = sim;
                p.limit = limit;
                field.search(p, out);
                std::vector<uint64_t> od(limit), orow(limit);
                std::vector<float> odist(limit);
                const uint32_t m = orc_vector_search(corpus.data(), row_doc.size(), (uint32_t)dim, row_doc.data(), nullptr,
                                                     q.data(), 0, (uint32_t)limit, nullptr, 0, od.data(), odist.data(),
                                                     orow.data());
                std::vector<uint64_t> mdoc(limit + 1);
                std::vector<float> msc(limit + 1);
                uint64_t mn = 0;
                orc_embedding_epilogue(od.data(), odist.data(), m, is_e5(model), sim, mdoc.data(), msc.data(), &mn);
                CHECK(out.size() == mn);
                for (uint64_t i = 0; i < mn; ++i) {
                    CHECK(out.count(mdoc[i]) == 1);
                    APPROX(out[mdoc[i]], msc[i], 2e-3);  // 1e-4 per row; the E5 rescale divides by 0.3 and rows add up
                }
            }
        }
        // delete + compact keep the wrapper's contract (embedding_field.rs:240-299)
        field.remove(100);
        CHECK(field.has_pending_ops());
        field.compact(42);
        CHECK(!field.has_pending_ops() && field.current_version_number() == 42);
        std::unordered_map<DocumentId, float> out;
        VectorSearchParams p;
        p.target = &q;
        p.similarity = 0.0f;
        p.limit = 1000;
        field.search(p, out);
        CHECK(out.count(100) == 0 && out.count(125) == 1);
    }
}

// token_score.rs:393-422 + sort.rs:260-279 against the oracle, bit-exact
static void test_normalize_and_combine_and_top_n(Context& ctx) {
    std::unordered_map<DocumentId, float> vec{{1, 0.9f}, {2, 0.8f}, {7, -0.25f}}, ft{{1, 3.2f}, {3, 1.1f}, {4, 0.4f}, {7, 2.0f}};
    auto r = normalize_and_combine(ctx, vec, ft, 10);
    const uint64_t vd[] = {1, 2, 7}, fd[] = {1, 3, 4, 7};
    const float vs[] = {0.9f, 0.8f, -0.25f}, fs[] = {3.2f, 1.1f, 0.4f, 2.0f};
    uint64_t cd[8], td[8];
    float cs[8], ts[8];
    const uint64_t cn = orc_normalize_and_combine(vd, vs, 3, fd, fs, 4, cd, cs);
    const uint64_t tn = orc_top_n(cd, cs, cn, 10, td, ts);
    CHECK(r.count == cn && r.hits.size() == tn);
    for (uint64_t i = 0; i < tn && i < r.hits.size(); ++i) {
        CHECK(r.hits[i].document_id == td[i]);
        CHECK(bits(r.hits[i].score) == bits(ts[i]));
    }
    std::unordered_map<DocumentId, float> m{{9, 1.0f}, {3, 2.0f}, {7, NAN}, {1, 2.0f}, {5, 1.0f}};
    auto top = top_n(ctx, m, 4);
    CHECK(top.size() == 4 && top[0].document_id == 1 && top[1].document_id == 3 && top[2].document_id == 5 &&
          top[3].document_id == 9);
}

static void test_errors_surface_as_exceptions(Context& ctx) {
    EmbeddingFieldStorage field(ctx, Model::BGESmall);
    std::vector<float> wrong(10, 1.0f);
    VectorSearchParams p;
    p.target = &wrong;
    std::unordered_map<DocumentId, float> out;
    bool threw = false;
    try {
        field.search(p, out);
    } catch (const Error& e) {
        threw = e.status == ORAMA_ERR_INVALID;
    }
    CHECK(threw);
}

// PostingsStore (seam ii) against the oracle fed with host-computed ntf, then append == rebuild, filters (host
// words and resident), hybrid in one call vs the two-step form.
static void test_postings_store(Context& ctx) {
    const size_t n_docs = 3000, T = 4;
    std::mt19937 rng(11);
    std::vector<uint64_t> docs(n_docs);
    std::vector<uint32_t> len(n_docs);
    double sum_len = 0;
    for (size_t i = 0; i < n_docs; ++i) {
        docs[i] = 5 + 2 * i;
        len[i] = 5 + rng() % 200;
        sum_len += len[i];
    }
    const float avg = (float)(sum_len / (double)n_docs);
    auto make_lists = [&](size_t lo, size_t hi) {
        std::mt19937 r2(99);
        std::vector<PostingList> lists(T);
        for (size_t t = 0; t < T; ++t)
            for (size_t i = 0; i < n_docs; ++i) {
                const uint32_t roll = r2();  // same stream for every (lo, hi): list t is a fixed subset
                if (roll % (3 + t) != 0) continue;
                if (i < lo || i >= hi) continue;
                lists[t].docs.push_back(docs[i]);
                lists[t].tf.push_back(1 + (roll >> 8) % 4);
                lists[t].field_len.push_back(len[i]);
            }
        return lists;
    };
    const auto lists = make_lists(0, n_docs);
    PostingsStore store(ctx);
    store.build(docs, {avg}, lists);
    std::vector<TermRef> refs;
    for (uint32_t t = 0; t < T; ++t) refs.push_back(TermRef{t, t, 1.0f});
    FullTextParams p;
    p.n_tokens = T;
    p.total_documents = (float)n_docs;
    p.top_k = 50;
    // oracle: contributions with ntf = boost * tf / (1 - b + b * len / avglen)
    std::vector<std::vector<float>> ntf(T);
    std::vector<orc_entry> entries;
    for (size_t t = 0; t < T; ++t) {
        for (size_t i = 0; i < lists[t].docs.size(); ++i)
            ntf[t].push_back(1.0f * orc_bm25f_normalized_tf(lists[t].tf[i], lists[t].field_len[i], avg, 0.75f));
        entries.push_back(orc_entry{(uint32_t)t, lists[t].docs.data(), ntf[t].data(), (uint64_t)lists[t].docs.size()});
    }
    std::vector<uint64_t> od(n_docs), td(50);
    std::vector<float> os(n_docs), ts(50);
    for (int thr : {0, 2}) {
        p.use_threshold = thr != 0;
        p.threshold = (uint32_t)thr;
        const uint64_t on = orc_search_full_text(entries.data(), (uint32_t)entries.size(), T, (float)n_docs, 1.2f,
                                                 thr != 0, (uint32_t)thr, od.data(), os.data());
        const uint64_t tn = orc_top_n(od.data(), os.data(), on, 50, td.data(), ts.data());
        TopResult r = store.search(refs, p);
        CHECK(r.count == on && r.hits.size() == tn);
        for (uint64_t i = 0; i < tn && i < r.hits.size(); ++i)
            CHECK(r.hits[i].document_id == td[i] && bits(r.hits[i].score) == bits(ts[i]));
    }
    p.use_threshold = false;
    // filters: host words == resident handle
    DocBitmap bm(docs.back() + 1);
    for (size_t i = 0; i < n_docs; ++i)
        if (i % 4 != 1) bm.insert(docs[i]);
    ResidentBitmap rbm(ctx, bm);
    p.filter = bm;
    TopResult fa = store.search(refs, p);
    p.filter = rbm;
    TopResult fb = store.search(refs, p);
    CHECK(fa.count == fb.count && fa.hits.size() == fb.hits.size() && fa.count > 0);
    for (size_t i = 0; i < fa.hits.size() && i < fb.hits.size(); ++i)
        CHECK(fa.hits[i].document_id == fb.hits[i].document_id && bits(fa.hits[i].score) == bits(fb.hits[i].score) &&
              bm.contains(fa.hits[i].document_id));
    p.filter = FilterRef();
    // append == rebuild (average length passed index-wide both times)
    PostingsStore live(ctx);
    live.build(std::vector<uint64_t>(docs.begin(), docs.begin() + 2000), {avg}, make_lists(0, 2000));
    const uint32_t first = live.append(std::vector<uint64_t>(docs.begin() + 2000, docs.end()), {avg}, make_lists(2000, n_docs));
    CHECK(first == T);
    std::vector<TermRef> refs2;
    for (uint32_t t = 0; t < T; ++t) {
        refs2.push_back(TermRef{t, t, 1.0f});
        refs2.push_back(TermRef{t, first + t, 1.0f});
    }
    TopResult a = store.search(refs, p), b = live.search(refs2, p);
    CHECK(a.count == b.count && a.hits.size() == b.hits.size());
    for (size_t i = 0; i < a.hits.size() && i < b.hits.size(); ++i)
        CHECK(a.hits[i].document_id == b.hits[i].document_id && bits(a.hits[i].score) == bits(b.hits[i].score));
    // hybrid: one call == vector search + epilogue + search_hybrid
    EmbeddingFieldStorage field(ctx, Model::BGESmall);
    std::normal_distribution<float> nd;
    std::vector<float> q(384);
    for (auto& x : q) x = nd(rng);
    for (size_t i = 0; i < n_docs; ++i) {
        std::vector<float> v(384);
        const float w = (i % 40 == 0) ? 0.9f : 0.0f;
        for (size_t c = 0; c < 384; ++c) v[c] = w * q[c] + (1.0f - w) * nd(rng);
        field.insert(docs[i], {v});
    }
    VectorSearchParams vp;
    vp.target = &q;
    vp.similarity = 0.3f;
    vp.limit = 20;
    std::unordered_map<DocumentId, float> vmap;
    field.search(vp, vmap);
    CHECK(!vmap.empty());
    TopResult h2 = store.search_hybrid(refs, p, vmap), h1 = store.hybrid_search(field, vp, refs, p);
    CHECK(h1.count == h2.count && h1.hits.size() == h2.hits.size());
    for (size_t i = 0; i < h1.hits.size() && i < h2.hits.size(); ++i)
        CHECK(h1.hits[i].document_id == h2.hits[i].document_id && bits(h1.hits[i].score) == bits(h2.hits[i].score));
}

// SearchBatcher: 24 threads, one request each, answers identical to the direct wrapper call
static void test_search_batcher(Context& ctx) {
    EmbeddingFieldStorage field(ctx, Model::BGESmall, /*half_precision=*/true);
    std::mt19937 rng(5);
    std::normal_distribution<float> nd;
    for (DocumentId d = 0; d < 4000; ++d) {
        std::vector<float> v(384);
        for (auto& x : v) x = nd(rng);
        field.insert(d, {v});
    }
    const int T = 24;
    std::vector<std::vector<float>> qs(T, std::vector<float>(384));
    for (auto& q : qs)
        for (auto& x : q) x = nd(rng);
    std::vector<std::unordered_map<DocumentId, float>> direct(T), batched(T);
    for (int i = 0; i < T; ++i) {
        VectorSearchParams p;
        p.target = &qs[i];
        p.similarity = -1.0f;
        p.limit = 10 + i;
        field.search(p, direct[i]);
    }
    SearchBatcher batcher(field, 8, 500);
    std::vector<std::thread> th;
    for (int i = 0; i < T; ++i)
        th.emplace_back([&, i] {
            VectorSearchParams p;
            p.target = &qs[i];
            p.similarity = -1.0f;
            p.limit = 10 + i;
            batcher.search(p, batched[i]);
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < T; ++i) {
        CHECK(direct[i].size() == batched[i].size() && direct[i].size() == (size_t)(10 + i));
        for (auto& kv : direct[i]) CHECK(batched[i].count(kv.first) == 1 && bits(batched[i][kv.first]) == bits(kv.second));
    }
}

int main() {
    Context ctx(0);
#define RUN(t)                      \
    do {                            \
        std::printf("%s\n", #t);    \
        t(ctx);                     \
    } while (0)
    RUN(test_bm25f_scorer_basic);
    RUN(test_canonical_bm25f_single_term_two_fields);
    RUN(test_threshold_scorer);
    RUN(test_omc_ratios);
    RUN(test_embedding_field_search);
    RUN(test_normalize_and_combine_and_top_n);
    RUN(test_errors_surface_as_exceptions);
    RUN(test_postings_store);
    RUN(test_search_batcher);
    std::printf(g_failed ? "FAILED (%d checks)\n" : "ALL PASSED\n", g_failed);
    return g_failed ? 1 : 0;
}
