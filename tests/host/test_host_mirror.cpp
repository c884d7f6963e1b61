// C++ parity tests of the host mirror (include/orama/host.hpp) against the oracle, written to read like the
// reference's own unit tests (src/collection_manager/bm25.rs:527-1043) and its search tests.
// Built and run by tests/test_host_cpp_gpu.py:  g++ -std=c++17 ... -lorama_hip -lorama_oracle
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

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
    std::printf(g_failed ? "FAILED (%d checks)\n" : "ALL PASSED\n", g_failed);
    return g_failed ? 1 : 0;
}
