// orama/host.hpp — C++ host-side mirror of the reference's interfaces for the scoring hot path, header-only,
// on top of the C ABI (orama_hip.h).  The reference is Rust; this image has no Rust toolchain, so this mirror
// (same names, argument meaning and error behaviour) is what a C++ host — and the parity tests in
// tests/host/ — program against; INTEGRATION.md shows the equivalent Rust shim.
//
//   reference item                                                        here
//   --------------------------------------------------------------------  ---------------------------------
//   EmbeddingFieldStorage  (sides/read/index/embedding_field.rs:63-320)   orama::host::EmbeddingFieldStorage
//   VectorSearchParams     (index/committed_field/vector.rs:10-15)        orama::host::VectorSearchParams
//   Model::{dimensions,rescale_score} (src/python/embeddings.rs:52-92)    orama::host::Model
//   FilterResult<DocumentId> as a predicate (embedding_field.rs:54-61)    orama::host::DocBitmap
//   BM25Scorer<DocumentId> (src/collection_manager/bm25.rs:135-323)       orama::host::BM25Scorer
//   top_n                  (sides/read/sort.rs:260-279)                   orama::host::top_n
//   normalize_and_combine  (index/token_score.rs:393-422)                 orama::host::normalize_and_combine
//   apply_omc_multipliers  (sides/read/search.rs:39-48)                   folded into the scoring calls
//   StringFieldStorage postings on the GPU (index/string_field.rs:155-225) orama::host::PostingsStore
//   search_full_text / search_hybrid (index/token_score.rs:186-387)       PostingsStore::search / search_hybrid /
//                                                                         hybrid_search (both legs, one call)
//   cached filter (index/filter.rs:344-392), kept in HBM                  orama::host::ResidentBitmap
//   request coalescing at the wrapper (extension)      orama::host::SearchBatcher
//   anyhow::Error                                                         orama::host::Error (status + message)
#pragma once

#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../orama_hip.h"

namespace orama {
namespace host {

using DocumentId = uint64_t;  // types.rs:112

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error("liborama_hip status " + std::to_string(s) + ": " + m), status(s) {}
};
inline void check(int status) {
    if (status != ORAMA_OK) throw Error(status, orama_last_error());
}

struct TokenScore {  // types.rs:363-366
    DocumentId document_id;
    float score;
};

class Context {
   public:
    explicit Context(int device = 0) { check(orama_ctx_create(device, &h_)); }
    ~Context() { orama_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    orama_ctx* raw() const { return h_; }

   private:
    orama_ctx* h_ = nullptr;
};

enum class Model {  // src/python/embeddings.rs:13-63
    BGESmall, BGEBase, BGELarge, JinaEmbeddingsV2BaseCode, MultilingualE5Small, MultilingualE5Base,
    MultilingualE5Large, MultilingualMiniLML12V2
};
inline size_t dimensions(Model m) {
    switch (m) {
        case Model::BGESmall: case Model::MultilingualE5Small: case Model::MultilingualMiniLML12V2: return 384;
        case Model::BGEBase: case Model::JinaEmbeddingsV2BaseCode: case Model::MultilingualE5Base: return 768;
        default: return 1024;
    }
}
inline bool is_e5(Model m) {
    return m == Model::MultilingualE5Small || m == Model::MultilingualE5Base || m == Model::MultilingualE5Large;
}
inline float rescale_score(Model m, float score) {  // embeddings.rs:71-92
    if (!is_e5(m)) return score;
    const float MIN = 0.7f, MAX = 1.0f, DELTA = MAX - MIN;
    float c = score;
    if (c < MIN) c = MIN;
    if (c > MAX) c = MAX;
    return (c - MIN) / DELTA;
}

// FilterResult<DocumentId> materialised as a bitmap over document ids (bit d set <=> `contains(d)`).
struct DocBitmap {
    std::vector<uint64_t> words;
    uint64_t bits = 0;
    explicit DocBitmap(uint64_t n_bits = 0) : words((n_bits + 63) / 64, 0), bits(n_bits) {}
    void insert(DocumentId d) {
        if (d < bits) words[d >> 6] |= 1ull << (d & 63);
    }
    bool contains(DocumentId d) const { return d < bits && ((words[d >> 6] >> (d & 63)) & 1ull); }
};

// The same bitmap kept in HBM (orama_allow_*): created once per distinct filter, passed by token afterwards.
class ResidentBitmap {
   public:
    ResidentBitmap(Context& ctx, const DocBitmap& b) : bits_(b.bits) {
        check(orama_allow_create(ctx.raw(), b.words.data(), b.bits, &h_));
    }
    ~ResidentBitmap() { orama_allow_destroy(h_); }
    ResidentBitmap(const ResidentBitmap&) = delete;
    ResidentBitmap& operator=(const ResidentBitmap&) = delete;
    void set(DocumentId d, bool allowed) { check(orama_allow_set(h_, &d, 1, allowed ? 1 : 0)); }
    const uint64_t* token() const { return orama_allow_token(h_); }
    uint64_t bits() const { return bits_; }

   private:
    orama_allow* h_ = nullptr;
    uint64_t bits_ = 0;
};

// What a search takes as its filter: nothing, host words (uploaded for the call) or a resident bitmap.
struct FilterRef {
    const uint64_t* words = nullptr;
    uint64_t bits = 0;
    FilterRef() = default;
    FilterRef(const DocBitmap* b) : words(b ? b->words.data() : nullptr), bits(b ? b->bits : 0) {}  // NOLINT
    FilterRef(const DocBitmap& b) : words(b.words.data()), bits(b.bits) {}                            // NOLINT
    FilterRef(const ResidentBitmap& b) : words(b.token()), bits(b.bits()) {}                          // NOLINT
};

struct VectorSearchParams {  // committed_field/vector.rs:10-15
    const std::vector<float>* target = nullptr;
    float similarity = 0.7f;  // Similarity default, types.rs:879-885
    size_t limit = 10;        // Limit default, types.rs:748-754
    FilterRef filtered_doc_ids;
};

class EmbeddingFieldStorage {
   public:
    // :65-76.  half_precision: rows stored as f16 in the MFMA-tiled layout (K2; batched queries share a pass)
    EmbeddingFieldStorage(Context& ctx, Model model, bool half_precision = false)
        : model_(model), dim_(dimensions(model)) {
        check(orama_vec_create(ctx.raw(), (uint32_t)dim_, ORAMA_METRIC_COSINE,
                               half_precision ? ORAMA_DTYPE_F16 : ORAMA_DTYPE_F32, 0, &h_));
    }
    ~EmbeddingFieldStorage() { orama_vec_destroy(h_); }
    EmbeddingFieldStorage(const EmbeddingFieldStorage&) = delete;
    EmbeddingFieldStorage& operator=(const EmbeddingFieldStorage&) = delete;
    Model model() const { return model_; }

    // insert — :232-237 (N vectors per document; vectors the indexer rejects are dropped silently)
    void insert(DocumentId doc_id, const std::vector<std::vector<float>>& vectors) {
        std::vector<float> flat;
        std::vector<uint64_t> ids;
        for (const auto& v : vectors) {
            if (v.size() != dim_) continue;
            flat.insert(flat.end(), v.begin(), v.end());
            ids.push_back(doc_id);
        }
        if (!ids.empty()) check(orama_vec_insert(h_, ids.data(), flat.data(), ids.size(), nullptr));
    }
    void remove(DocumentId doc_id) { check(orama_vec_delete(h_, &doc_id, 1)); }  // delete — :240-242
    void compact(uint64_t version) { check(orama_vec_compact(h_, version)); }    // :286-290
    bool has_pending_ops() const { return info().pending_ops > 0; }              // :281-283
    uint64_t current_version_number() const { return info().version; }           // :298-300
    orama_vec_info_t info() const {
        orama_vec_info_t i{};
        check(orama_vec_info(h_, &i));
        return i;
    }

    // search — :250-278.  The storage call returns (doc, cosine distance) per row; the epilogue below is the
    // reference's own code: similarity = 1 - distance, rescale, cut-off, per-document sum.
    void search(const VectorSearchParams& params, std::unordered_map<DocumentId, float>& output) const {
        if (!params.target || params.target->size() != dim_) throw Error(ORAMA_ERR_INVALID, "target dimension mismatch");
        const size_t k = params.limit;
        std::vector<uint64_t> ids(k ? k : 1);
        std::vector<float> dist(k ? k : 1);
        uint32_t n = 0;
        const uint64_t* bm = params.filtered_doc_ids.words;
        const uint64_t bits = params.filtered_doc_ids.bits;
        check(orama_vec_search(h_, params.target->data(), 1, (uint32_t)k, bm, bits, ids.data(), dist.data(), &n));
        for (uint32_t i = 0; i < n; ++i) {
            const float similarity = 1.0f - dist[i];
            const float score = rescale_score(model_, similarity);
            if (score >= params.similarity) output[ids[i]] += score;
        }
    }
    orama_vec* raw() const { return h_; }

   private:
    Model model_;
    size_t dim_;
    orama_vec* h_ = nullptr;
};

struct TopResult {
    std::vector<TokenScore> hits;  // top-(limit+offset), score desc, DocumentId asc
    uint64_t count = 0;            // token_score_results.len() — search.rs:482
};

// BM25Scorer<DocumentId> restricted to the calls search_full_text makes (token_score.rs:211-302).  The
// reference scores while contributions arrive; here they are recorded and `top_n()/get_scores()` runs the
// GPU pass (orama_bm25_score): accumulate per (token, doc), finalise with the Lucene idf, threshold mask,
// OMC, count, top-k.
class BM25Scorer {
   public:
    static BM25Scorer plain(Context& ctx) { return BM25Scorer(ctx, false, 0); }
    static BM25Scorer with_threshold(Context& ctx, uint32_t threshold) { return BM25Scorer(ctx, true, threshold); }

    void reset_term() { cur_.clear(); }
    void add_precomputed_field(DocumentId key, float normalized_tf, float weight) {
        cur_.emplace_back(key, weight * normalized_tf);
    }
    size_t current_term_document_count() const {
        std::map<DocumentId, int> s;
        for (auto& c : cur_) s[c.first] = 1;
        return s.size();
    }
    // finalize_term / finalize_term_plain — bm25.rs:202-240.  corpus_term_frequency must be the number of
    // distinct documents of the term (what the in-tree caller passes, token_score.rs:262-275).
    void finalize_term(size_t corpus_term_frequency, float total_documents, float k, float phrase_boost = 1.0f,
                       uint32_t /*token_indexes*/ = 0) {
        if (phrase_boost != 1.0f) throw Error(ORAMA_ERR_UNSUPPORTED, "phrase_boost != 1.0");
        if (corpus_term_frequency != std::max<size_t>(current_term_document_count(), 1))
            throw Error(ORAMA_ERR_UNSUPPORTED, "corpus_term_frequency must equal the distinct documents of the term");
        total_documents_ = total_documents;
        k_ = k;
        // split the push stream into runs with unique documents = posting-list entries, in push order
        std::vector<uint64_t> d;
        std::vector<float> v;
        std::map<DocumentId, int> seen;
        auto flush = [&] {
            entries_.push_back(Entry{term_index_, d, v});
            d.clear();
            v.clear();
            seen.clear();
        };
        for (auto& c : cur_) {
            if (seen.count(c.first)) flush();
            seen[c.first] = 1;
            d.push_back(c.first);
            v.push_back(c.second);
        }
        flush();
    }
    void finalize_term_plain(size_t df, float total_documents, float k, float phrase_boost = 1.0f) {
        finalize_term(df, total_documents, k, phrase_boost, 0);
    }
    void next_term() {
        ++term_index_;
        cur_.clear();
    }

    // get_scores() + apply_omc_multipliers + count + top_n in one device pass.
    TopResult top_n(size_t n, const std::map<DocumentId, float>& omc = {}) const {
        std::vector<orama_ntf_entry> raw;
        for (const auto& e : entries_)
            raw.push_back(orama_ntf_entry{e.token, e.doc.data(), e.ntf.data(), (uint64_t)e.doc.size()});
        orama_bm25_params p{};
        p.total_documents = total_documents_;
        p.k = k_;
        p.n_tokens = std::max<uint32_t>(term_index_, 1);
        p.use_threshold = with_threshold_ ? 1 : 0;
        p.threshold = threshold_;
        p.top_k = (uint32_t)n;
        std::vector<uint64_t> od, ids(n ? n : 1);
        std::vector<float> om, sc(n ? n : 1);
        for (auto& kv : omc) {
            od.push_back(kv.first);
            om.push_back(kv.second);
        }
        uint32_t out_n = 0;
        TopResult r;
        check(orama_bm25_score(ctx_.raw(), raw.data(), (uint32_t)raw.size(), &p, od.data(), om.data(), od.size(),
                               ids.data(), sc.data(), &out_n, &r.count));
        for (uint32_t i = 0; i < out_n; ++i) r.hits.push_back(TokenScore{ids[i], sc[i]});
        return r;
    }
    // The whole HashMap<K, f32> (bm25.rs:416-428), any size: orama_bm25_score_map.
    std::unordered_map<DocumentId, float> get_scores() const {
        std::vector<orama_ntf_entry> raw;
        size_t total = 0;
        for (const auto& e : entries_) {
            raw.push_back(orama_ntf_entry{e.token, e.doc.data(), e.ntf.data(), (uint64_t)e.doc.size()});
            total += e.doc.size();
        }
        orama_bm25_params p{};
        p.total_documents = total_documents_;
        p.k = k_;
        p.n_tokens = std::max<uint32_t>(term_index_, 1);
        p.use_threshold = with_threshold_ ? 1 : 0;
        p.threshold = threshold_;
        p.top_k = 0;
        std::vector<uint64_t> ids(total ? total : 1);
        std::vector<float> sc(total ? total : 1);
        uint64_t n = 0;
        check(orama_bm25_score_map(ctx_.raw(), raw.data(), (uint32_t)raw.size(), &p, nullptr, nullptr, 0, total, ids.data(),
                                   sc.data(), &n));
        std::unordered_map<DocumentId, float> m;
        for (uint64_t i = 0; i < n; ++i) m[ids[i]] = sc[i];
        return m;
    }

   private:
    struct Entry {
        uint32_t token;
        std::vector<uint64_t> doc;
        std::vector<float> ntf;
    };
    BM25Scorer(Context& ctx, bool wt, uint32_t thr) : ctx_(ctx), with_threshold_(wt), threshold_(thr) {}
    Context& ctx_;
    bool with_threshold_;
    uint32_t threshold_;
    uint32_t term_index_ = 0;
    float total_documents_ = 1.0f, k_ = 1.2f;
    std::vector<std::pair<DocumentId, float>> cur_;
    std::vector<Entry> entries_;
};

// top_n — sort.rs:260-279
inline std::vector<TokenScore> top_n(Context& ctx, const std::unordered_map<DocumentId, float>& token_scores, size_t n) {
    std::vector<uint64_t> d, ids(n ? n : 1);
    std::vector<float> s, sc(n ? n : 1);
    for (auto& kv : token_scores) {
        d.push_back(kv.first);
        s.push_back(kv.second);
    }
    uint32_t out_n = 0;
    check(orama_top_n(ctx.raw(), d.data(), s.data(), d.size(), (uint32_t)n, ids.data(), sc.data(), &out_n));
    std::vector<TokenScore> r;
    for (uint32_t i = 0; i < out_n; ++i) r.push_back(TokenScore{ids[i], sc[i]});
    return r;
}

// normalize_and_combine — token_score.rs:393-422 (+ count + top_n)
inline TopResult normalize_and_combine(Context& ctx, const std::unordered_map<DocumentId, float>& vector,
                                       const std::unordered_map<DocumentId, float>& fulltext, size_t n) {
    std::vector<uint64_t> vd, fd, ids(n ? n : 1);
    std::vector<float> vs, fs, sc(n ? n : 1);
    for (auto& kv : vector) {
        vd.push_back(kv.first);
        vs.push_back(kv.second);
    }
    for (auto& kv : fulltext) {
        fd.push_back(kv.first);
        fs.push_back(kv.second);
    }
    uint32_t out_n = 0;
    TopResult r;
    check(orama_hybrid_combine(ctx.raw(), vd.data(), vs.data(), vd.size(), fd.data(), fs.data(), fd.size(),
                               (uint32_t)n, ids.data(), sc.data(), &out_n, &r.count));
    for (uint32_t i = 0; i < out_n; ++i) r.hits.push_back(TokenScore{ids[i], sc[i]});
    return r;
}

// Request coalescing in front of one EmbeddingFieldStorage (orama_batcher_*): concurrent single-target callers
// share corpus passes; `search` has the semantics of the storage call of EmbeddingFieldStorage::search without
// a filter and blocks the calling thread until its batch has been scanned.
class SearchBatcher {
   public:
    explicit SearchBatcher(EmbeddingFieldStorage& field, uint32_t max_batch = 64, uint32_t max_wait_us = 0)
        : field_(field) {
        check(orama_batcher_create(field.raw(), max_batch, max_wait_us, &h_));
    }
    ~SearchBatcher() { orama_batcher_destroy(h_); }
    SearchBatcher(const SearchBatcher&) = delete;
    SearchBatcher& operator=(const SearchBatcher&) = delete;

    // EmbeddingFieldStorage::search (:250-278) for an unfiltered request
    void search(const VectorSearchParams& params, std::unordered_map<DocumentId, float>& output) const {
        const size_t k = params.limit;
        std::vector<uint64_t> ids(k ? k : 1);
        std::vector<float> dist(k ? k : 1);
        uint32_t n = 0;
        check(orama_batcher_search(h_, params.target->data(), (uint32_t)k, ids.data(), dist.data(), &n));
        for (uint32_t i = 0; i < n; ++i) {
            const float similarity = 1.0f - dist[i];
            const float score = rescale_score(field_.model(), similarity);
            if (score >= params.similarity) output[ids[i]] += score;
        }
    }

   private:
    EmbeddingFieldStorage& field_;
    orama_batcher* h_ = nullptr;
};

// One posting list as StringFieldStorage holds it per (field, term): (doc, tf = positions.len(), field_length).
struct PostingList {
    uint32_t field = 0;
    std::vector<uint64_t> docs;  // ascending
    std::vector<uint32_t> tf, field_len;
};

// (token index, posting list, field boost) — SearchParams.boost, token_score.rs:234-238
using TermRef = orama_term_ref;

// The exact-match factor the third-party string store folds into ntf (token_score.rs:182-185, 226-228, 268-269).  Its value is
// not in the reference checkout (oramacore_fields 0.2.0 is un-vendored); boost_integration.rs:449-491 pins only that it is > 1.
// 1.5 is a PLACEHOLDER: the shim passes AcceleratorConfig.exact_match_boost, read from
// oramacore_fields::string::SearchParams' implementation when the crate is available.  1.0 = no factor.
constexpr float kDefaultExactMatchBoost = 1.5f;

// The reference of one (query token, posting list) pair as the kernels take it: the field boost times the exact-match factor
// when the list's dictionary term IS the query token (prefix / Levenshtein expansions keep 1.0) — one f32 product, as in
// oramacore_amd/token_score.py::TokenScoreContext._refs.
inline TermRef term_ref(uint32_t token, uint32_t list, float field_boost, bool term_is_token,
                        float exact_match_boost = kDefaultExactMatchBoost) {
    return TermRef{token, list, term_is_token && exact_match_boost != 1.0f ? field_boost * exact_match_boost : field_boost};
}

struct FullTextParams {            // what search_full_text derives (token_score.rs:186-302)
    uint32_t n_tokens = 1;
    float total_documents = 1.0f;  // the index's document_count (:221)
    size_t top_k = 10;             // limit + offset (sort.rs:24-34)
    bool use_threshold = false;    // Threshold -> floor(n_tokens * t) (:211-218)
    uint32_t threshold = 0;
    float k = 1.2f, b = 0.75f;     // Bm25Params::default()
    bool apply_omc = true;
    FilterRef filter;
};

// The committed postings of one index, resident in HBM (seam ii) + the scoring entry points over them.
class PostingsStore {
   public:
    explicit PostingsStore(Context& ctx) { check(orama_post_create(ctx.raw(), &h_)); }
    ~PostingsStore() { orama_post_destroy(h_); }
    PostingsStore(const PostingsStore&) = delete;
    PostingsStore& operator=(const PostingsStore&) = delete;

    // commit: all live documents (ascending ids), per-field average lengths, one list per (field, term)
    void build(const std::vector<uint64_t>& docs, const std::vector<float>& avg_field_len,
               const std::vector<PostingList>& lists) {
        Flat f(lists);
        check(orama_post_build(h_, docs.data(), docs.size(), (uint32_t)avg_field_len.size(), avg_field_len.data(),
                               (uint32_t)lists.size(), f.field.data(), f.off.data(), f.doc.data(), f.tf.data(),
                               f.len.data()));
    }
    // insert between commits: new documents + delta lists; returns the id of the first new list
    uint32_t append(const std::vector<uint64_t>& docs, const std::vector<float>& avg_field_len,
                    const std::vector<PostingList>& lists) {
        uint32_t first = 0;
        check(orama_post_info(h_, nullptr, &first, nullptr, nullptr));
        Flat f(lists);
        check(orama_post_append(h_, docs.data(), docs.size(), avg_field_len.data(), (uint32_t)lists.size(),
                                f.field.data(), f.off.data(), f.doc.data(), f.tf.data(), f.len.data()));
        return first;
    }
    void set_omc(const std::map<DocumentId, float>& omc) {  // Index::get_all_omc, index/mod.rs:1720-1739
        std::vector<uint64_t> d;
        std::vector<float> m;
        for (auto& kv : omc) {
            d.push_back(kv.first);
            m.push_back(kv.second);
        }
        check(orama_post_set_omc(h_, d.data(), m.data(), d.size()));
    }

    // search_full_text (+ apply_omc + count + top_n) — token_score.rs:186-302, search.rs:39-48,482, sort.rs:260-279
    TopResult search(const std::vector<TermRef>& refs, const FullTextParams& p) const {
        Out o(p.top_k);
        const orama_bm25_params bp = params(p);
        check(orama_post_search(h_, refs.data(), (uint32_t)refs.size(), p.b, &bp, p.filter.words, p.filter.bits,
                                p.apply_omc ? 1 : 0, o.ids.data(), o.sc.data(), &o.n, &o.r.count));
        return o.done();
    }
    // search_hybrid with the vector map already computed (token_score.rs:357-422)
    TopResult search_hybrid(const std::vector<TermRef>& refs, const FullTextParams& p,
                            const std::unordered_map<DocumentId, float>& vector) const {
        std::vector<uint64_t> vd;
        std::vector<float> vs;
        for (auto& kv : vector) {
            vd.push_back(kv.first);
            vs.push_back(kv.second);
        }
        Out o(p.top_k);
        const orama_bm25_params bp = params(p);
        check(orama_post_search_hybrid(h_, refs.data(), (uint32_t)refs.size(), p.b, &bp, p.filter.words,
                                       p.filter.bits, vd.data(), vs.data(), (uint32_t)vd.size(), p.apply_omc ? 1 : 0,
                                       o.ids.data(), o.sc.data(), &o.n, &o.r.count));
        return o.done();
    }
    // search_hybrid as ONE call: vector leg (field, target, limit, similarity) and full-text leg overlap on two
    // HIP streams; the EmbeddingFieldStorage::search epilogue runs inside the library
    TopResult hybrid_search(const EmbeddingFieldStorage& field, const VectorSearchParams& v,
                            const std::vector<TermRef>& refs, const FullTextParams& p) const {
        Out o(p.top_k);
        const orama_bm25_params bp = params(p);
        check(orama_hybrid_search(field.raw(), h_, v.target->data(), (uint32_t)v.limit, v.similarity,
                                  is_e5(field.model()) ? 1 : 0, refs.data(), (uint32_t)refs.size(), p.b, &bp,
                                  p.filter.words, p.filter.bits, p.apply_omc ? 1 : 0, o.ids.data(), o.sc.data(), &o.n,
                                  &o.r.count));
        return o.done();
    }
    orama_post* raw() const { return h_; }

   private:
    struct Flat {
        std::vector<uint32_t> field, tf, len;
        std::vector<uint64_t> off, doc;
        explicit Flat(const std::vector<PostingList>& lists) : off(1, 0) {
            for (const auto& l : lists) {
                field.push_back(l.field);
                doc.insert(doc.end(), l.docs.begin(), l.docs.end());
                tf.insert(tf.end(), l.tf.begin(), l.tf.end());
                len.insert(len.end(), l.field_len.begin(), l.field_len.end());
                off.push_back(doc.size());
            }
            if (doc.empty()) {  // keep data() non-null
                doc.reserve(1);
                tf.reserve(1);
                len.reserve(1);
            }
            if (field.empty()) field.reserve(1);
        }
    };
    struct Out {
        std::vector<uint64_t> ids;
        std::vector<float> sc;
        uint32_t n = 0;
        TopResult r;
        explicit Out(size_t k) : ids(k ? k : 1), sc(k ? k : 1) {}
        TopResult done() {
            for (uint32_t i = 0; i < n; ++i) r.hits.push_back(TokenScore{ids[i], sc[i]});
            return std::move(r);
        }
    };
    static orama_bm25_params params(const FullTextParams& p) {
        orama_bm25_params bp{};
        bp.total_documents = p.total_documents;
        bp.k = p.k;
        bp.n_tokens = p.n_tokens;
        bp.use_threshold = p.use_threshold ? 1 : 0;
        bp.threshold = p.threshold;
        bp.top_k = (uint32_t)p.top_k;
        return bp;
    }
    orama_post* h_ = nullptr;
};

}  // namespace host
}  // namespace orama
