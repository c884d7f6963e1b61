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
#include <functional>
#include <atomic>
#include <chrono>
#include <unordered_map>
#include <cmath>
#include <deque>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "bm25_kernels.hpp"
#include "bm25_ranges.hpp"
#include "hybrid_tail.hpp"
#include "shard_exchange.hpp"
#include "common.hpp"
#include "select.hpp"
#include "stage.hpp"
#include "vec_internal.hpp"

using namespace orama;

namespace {

// reserve + zero on (re)allocation: epoch-stamped arrays must never expose uninitialised epochs
int reserve_zeroed(DevBuf& b, size_t bytes, hipStream_t s) {
    if (bytes <= b.cap) return ORAMA_OK;
    ORAMA_TRY(b.reserve(bytes + bytes / 8));
    ORAMA_HIP_TRY(hipMemsetAsync(b.p, 0, b.cap, s));
    return ORAMA_OK;
}

constexpr size_t kDfStageOff = (sizeof(Bm25State) + 63) & ~(size_t)63;  // df read-back area inside h_misc

struct QueryBuffers {
    Bm25State* state = nullptr;
    uint32_t* touched = nullptr;
    float* cand_score = nullptr;
    uint32_t* cand_idx = nullptr;
    unsigned long long* acc = nullptr;
    uint32_t slots = 0;  // cells per document record: pow2 >= n_tokens + 1
    unsigned long long* emit = nullptr;
    uint32_t epoch = 0;
};

// Carve the per-query device buffers out of the scratch set and start a new epoch.
int prepare_query(Scratch* sc, uint64_t n_docs, uint32_t n_tokens, uint64_t cand_cap, uint32_t n_slots,
                  QueryBuffers* qb) {
    hipStream_t s = sc->stream;
    uint32_t slots = 2;
    while (slots < n_tokens + 1) slots <<= 1;
    ORAMA_TRY(reserve_zeroed(sc->bm25_acc, (size_t)slots * n_docs * 8, s));
    ORAMA_TRY(reserve_zeroed(sc->bm25_emit, (size_t)n_docs * 8, s));
    if (sc->bm25_epoch == 0xffffffffu) {  // wrap: forget every stamp
        ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_acc.p, 0, sc->bm25_acc.cap, s));
        ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_emit.p, 0, sc->bm25_emit.cap, s));
        sc->bm25_epoch = 0;
    }
    qb->epoch = ++sc->bm25_epoch;
    ORAMA_TRY(sc->misc2.reserve(sizeof(Bm25State)));
    ORAMA_TRY(sc->misc3.reserve((size_t)(cand_cap ? cand_cap : 1) * 4));      // touched
    ORAMA_TRY(sc->misc4.reserve((size_t)(cand_cap ? cand_cap : 1) * 4));      // cand_score
    ORAMA_TRY(sc->misc5.reserve((size_t)(cand_cap ? cand_cap : 1) * 4));      // cand_idx
    ORAMA_TRY(sc->h_misc.reserve(kDfStageOff + 4096));
    Bm25State* hs = sc->h_misc.as<Bm25State>();
    memset(hs, 0, sizeof(Bm25State));
    hs->min_key = 0xffffffffu;
    hs->list_len = n_slots;  // one candidate slot per referenced posting; hybrid appends behind them
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc2.p, hs, sizeof(Bm25State), hipMemcpyHostToDevice, s));
    qb->state = sc->misc2.as<Bm25State>();
    qb->touched = sc->misc3.as<uint32_t>();
    qb->cand_score = sc->misc4.as<float>();
    qb->cand_idx = sc->misc5.as<uint32_t>();
    qb->acc = sc->bm25_acc.as<unsigned long long>();
    qb->slots = slots;
    qb->emit = sc->bm25_emit.as<unsigned long long>();
    return ORAMA_OK;
}

// K4 over the candidate list into caller-named device outputs
int select_enqueue(orama_ctx* ctx, Scratch* sc, const QueryBuffers& qb, const uint64_t* d_docs, uint32_t cand_cap,
                   uint32_t top_k, uint64_t* d_out_ids, float* d_out_val, uint32_t* d_out_n) {
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState)));
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * top_k));
    SelectPlan p;
    p.vals = qb.cand_score;
    p.idx = qb.cand_idx;
    p.stride = 0;
    p.n_dev = &qb.state->list_len;
    p.n = cand_cap;
    p.q = 1;
    p.k = top_k;
    p.descending = true;
    p.id_map = d_docs;
    p.state = sc->sel_state.as<SelectState>();
    p.keys = sc->sel_keys.as<unsigned long long>();
    p.out_ids = d_out_ids;
    p.out_val = d_out_val;
    p.out_n = d_out_n;
    return launch_select(ctx, p, sc->stream);
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
    if (top_k)
        ORAMA_TRY(select_enqueue(ctx, sc, qb, d_docs, cand_cap, top_k, sc->out_ids.as<uint64_t>(),
                                 sc->out_val.as<float>(), sc->out_n.as<uint32_t>()));
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
    ORAMA_REQUIRE(p->n_tokens >= 1, "no query tokens");
    ORAMA_SUPPORT(p->n_tokens <= kMaxTokens, "n_tokens %u outside [1, %u]", p->n_tokens,
                  kMaxTokens);
    ORAMA_SUPPORT(p->top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", p->top_k, kSelectMaxK);
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
    std::vector<uint64_t> h_docs;  // empty when the ids are dense (docs[i] = dense_base + i)
    bool dense = false;
    uint64_t dense_base = 0;
    // DocumentId -> local index (rank of the id); false when the id is not a document of this index
    bool local_of(uint64_t id, uint32_t* out) const {
        if (dense) {
            if (id < dense_base || id - dense_base >= n_docs) return false;
            *out = (uint32_t)(id - dense_base);
            return true;
        }
        auto it = std::lower_bound(h_docs.begin(), h_docs.end(), id);
        if (it == h_docs.end() || *it != id) return false;
        *out = (uint32_t)(it - h_docs.begin());
        return true;
    }
    std::vector<float> avg_len;
    std::vector<uint32_t> field_of_list;
    std::vector<uint64_t> list_off;
    DevBuf d_docs, d_post_doc, d_post_val, d_omc;
    // tf / ((1 - b) + b * len / avg_len) of every posting for b = ntf_b and the CURRENT average lengths: the query-independent
    // part of the normalised tf, divided once per posting at build / append / set_avg_len instead of once per posting and
    // query (same f32 operations in the same order as the kernels' own division: same bits).  A search with another b
    // divides in the kernel.  +4 B per posting.
    DevBuf d_post_ntf;
    // Dense-list accelerators (round 6): for the longest lists (at least n_docs / 128 postings, as many as fit 6 bytes per posting
    // of the store) whose normalised tfs are all tame numbers, a bitmap of its documents + the exclusive popcount prefix of the
    // bitmap's words (n_docs / 4 bytes per list).  Derived data, rebuilt with post_ntf whenever the postings change; the range scorer reads a
    // frequent term's documents as WORDS where the query's published floor says none of them can reach the answer alone
    // (bm25_ranges_fast.hip).  acc_off_of_list[l] = RangeSeg::acc_off (0: none).
    DevBuf d_acc;
    std::vector<uint64_t> acc_off_of_list;
    uint32_t acc_words = 0;
    float ntf_b = 0.75f;  // Bm25Params::default().b
    bool ntf_valid = false;
    bool has_omc = false;
    uint64_t generation = 0;  // bumped by every rebuild of the doc table: facet fields resolved against an older one are stale
    // bumped (under the exclusive lock) by everything a search's answer depends on — postings, average lengths, multipliers: a
    // call that lets go of the shared lock between two phases (the sharded batch around its first exchange) compares it
    uint64_t mutations = 0;
    // corpus_docs.len() of a token that has SEVERAL lists (one per field, prefix / typo expansions) = the number of distinct
    // documents in the union of those lists (token_score.rs:262-275): a property of the INDEX, not of the query.  The first
    // query that brings a set of lists has it counted on the device (a second scoring-sized launch and a host round trip in
    // front of the real one); the count is remembered under the sorted list ids until the postings change.  Under a filter
    // the count — of one list or of several — depends on the filter: remembered as well when the filter is a RESIDENT bitmap
    // (orama_allow_*: the NOT-deleted bitmap every query of an index with pending deletes carries), under the version of its
    // content (0 = no filter); host words are anybody's guess and nothing is remembered for them.
    std::mutex df_union_mu;
    std::unordered_map<std::string, uint32_t> df_union;
    // entries; the map is emptied when it gets there (keys are at most 272 bytes: under 32 MB of host memory with the nodes)
    static constexpr size_t kDfUnionMax = 1u << 16;
    // (the count under a filter also depends on how many of the bitmap's bits the call declares valid — a posting is kept
    // only when id < bitmap_bits: the key carries both)
    static std::string df_union_key(const uint32_t* lists, uint32_t n, uint64_t filter_version, uint64_t filter_bits = 0) {
        uint32_t sorted[kRangeMaxRefs];
        std::copy(lists, lists + n, sorted);
        std::sort(sorted, sorted + n);
        std::string key(reinterpret_cast<const char*>(sorted), (size_t)n * 4);
        key.append(reinterpret_cast<const char*>(&filter_version), 8);
        if (filter_version) key.append(reinterpret_cast<const char*>(&filter_bits), 8);
        return key;
    }
    // Range width that worked for a set of lists: a query whose lists overlap heavily — the same term in
    // several fields, terms that occur together — overflows the scoring launch's cell tables at the default width, is scored
    // once for nothing and rerun with 8x narrower ranges; the NEXT query over the same lists starts at the width that held.
    std::unordered_map<std::string, uint32_t> shrink_hint;
};

namespace {

// orama_post::df_union.  `filtered` with filter_version == 0 (host words): nothing is known, nothing is kept.
// recall: true when every token of the query that needs a COUNT — several lists, or any list under a filter — has one
// remembered; df[t] is set for those tokens (the others keep the caller's value: the list length).
bool df_recall(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, uint32_t n_tokens, bool filtered, uint64_t filter_version,
               uint64_t filter_bits, uint32_t* df) {
    if (filtered && !filter_version) return false;
    const uint32_t need = filtered ? 1u : 2u;
    std::lock_guard<std::mutex> g(p->df_union_mu);
    for (uint32_t t = 0; t < n_tokens; ++t) {
        uint32_t lists[kRangeMaxRefs], nl = 0;
        for (uint32_t i = 0; i < n_refs; ++i) {
            if (refs[i].token != t) continue;
            const uint32_t l = refs[i].list;
            if (p->list_off[l + 1] == p->list_off[l]) continue;
            if (nl == kRangeMaxRefs) return false;
            lists[nl++] = l;
        }
        if (nl < need) continue;
        auto it = p->df_union.find(orama_post::df_union_key(lists, nl, filtered ? filter_version : 0, filter_bits));
        if (it == p->df_union.end()) return false;
        df[t] = it->second;
    }
    return true;
}
void df_remember(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, uint32_t n_tokens, bool filtered, uint64_t filter_version,
                 uint64_t filter_bits, const uint32_t* df) {
    if (filtered && !filter_version) return;
    const uint32_t need = filtered ? 1u : 2u;
    std::lock_guard<std::mutex> g(p->df_union_mu);
    if (p->df_union.size() >= orama_post::kDfUnionMax) p->df_union.clear();
    for (uint32_t t = 0; t < n_tokens; ++t) {
        uint32_t lists[kRangeMaxRefs], nl = 0;
        for (uint32_t i = 0; i < n_refs; ++i) {
            if (refs[i].token != t) continue;
            const uint32_t l = refs[i].list;
            if (p->list_off[l + 1] != p->list_off[l] && nl < kRangeMaxRefs) lists[nl++] = l;
        }
        if (nl >= need) p->df_union[orama_post::df_union_key(lists, nl, filtered ? filter_version : 0, filter_bits)] = df[t];
    }
}

// orama_post::shrink_hint: the non-empty lists of the whole query, sorted
std::string shrink_key(const orama_post* p, const orama_term_ref* refs, uint32_t n_refs) {
    uint32_t lists[kRangeMaxRefs], nl = 0;
    for (uint32_t i = 0; i < n_refs && nl < kRangeMaxRefs; ++i)
        if (p->list_off[refs[i].list + 1] != p->list_off[refs[i].list]) lists[nl++] = refs[i].list;
    return orama_post::df_union_key(lists, nl, 0);
}
uint32_t shrink_recall(orama_post* p, const orama_term_ref* refs, uint32_t n_refs) {
    std::lock_guard<std::mutex> g(p->df_union_mu);
    if (p->shrink_hint.empty()) return 0u;
    auto it = p->shrink_hint.find(shrink_key(p, refs, n_refs));
    return it == p->shrink_hint.end() ? 0u : it->second;
}
// The hint is the SMALLEST narrowing that held for the lists: a filtered query that needed 8x narrower ranges must not narrow the
// unfiltered queries over the same lists for good — a run that completes at a wider width overwrites a narrower hint (the hint is
// only where the retry loop STARTS: a width that overflows is narrowed again by the loop itself).
void shrink_remember(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, uint32_t shrink) {
    std::lock_guard<std::mutex> g(p->df_union_mu);
    if (p->shrink_hint.size() >= orama_post::kDfUnionMax) p->shrink_hint.clear();
    p->shrink_hint[shrink_key(p, refs, n_refs)] = shrink;
}

// orama_post::d_acc.  Best effort: without memory for them (or on any failure) the store simply has no accelerators.
// (Their only reader is the comparison unit bm25_ranges_fast.hip: built in the comparison flavour only.)
void build_dense_accelerators(orama_post* p, const uint64_t* d_list_off) {
    p->acc_off_of_list.assign(p->n_lists, 0);
    p->acc_words = 0;
#if !ORAMA_COMPARISON_KERNELS
    (void)d_list_off;
#else
    if (!p->ctx->bm25_dense_acc || p->n_docs < 32768 || p->n_docs > 0xffffffffull) return;
    // the longest lists first, as many as fit a budget of 6 bytes per posting of the store (+50 % at most), none under
    // n_docs / 128 postings (a word of such a list holds a posting every fourth time: reading it costs what gathering does)
    std::vector<uint32_t> lists;
    for (uint32_t l = 0; l < p->n_lists; ++l)
        if ((p->list_off[l + 1] - p->list_off[l]) * 128 >= p->n_docs) lists.push_back(l);
    if (lists.empty()) return;
    const uint32_t words = (uint32_t)((p->n_docs + 31) / 32);
    std::sort(lists.begin(), lists.end(), [&](uint32_t a, uint32_t b2) {
        const uint64_t la = p->list_off[a + 1] - p->list_off[a], lb = p->list_off[b2 + 1] - p->list_off[b2];
        return la != lb ? la > lb : a < b2;
    });
    const uint64_t fit = p->n_postings * 6 / ((uint64_t)words * 8);
    if (fit == 0) return;
    if (lists.size() > fit) lists.resize((size_t)fit);
    const uint32_t n = (uint32_t)lists.size();
    const size_t acc_bytes = (size_t)n * 2 * words * 4;
    DevBuf aux;  // [list indices u32 x n | min, max f32 x n]
    std::vector<float> minmax((size_t)n * 2);
    bool ok = p->d_acc.reserve(acc_bytes) == ORAMA_OK && aux.reserve((size_t)n * 12) == ORAMA_OK &&
              hipMemcpy(aux.p, lists.data(), (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
              launch_acc_build(p->d_post_doc.as<uint32_t>(), p->d_post_ntf.as<float>(), d_list_off, aux.as<uint32_t>(), n, words,
                               p->d_acc.as<uint32_t>(), reinterpret_cast<float*>(aux.as<char>() + (size_t)n * 4), nullptr) == ORAMA_OK &&
              hipDeviceSynchronize() == hipSuccess &&
              hipMemcpy(minmax.data(), aux.as<char>() + (size_t)n * 4, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
        if (const char* e = orama::dev_env("ORAMA_K3R_STATS"); e && std::atoi(e) != 0)
            fprintf(stderr, "[k3r] dense-list accelerators: build failed (%u lists, %.1f MB): %s\n", n, acc_bytes / 1e6, hipGetErrorString(hipGetLastError()));
        (void)hipGetLastError();
        clear_error();
        p->d_acc.release();
        return;
    }
    p->acc_words = words;
    if (const char* e = orama::dev_env("ORAMA_K3R_STATS"); e && std::atoi(e) != 0)  // (comparison flavour)
        fprintf(stderr, "[k3r] dense-list accelerators: %u of %u lists, %.1f MB; first list ntf in [%g, %g]\n", n, p->n_lists, acc_bytes / 1e6,
                minmax[0], minmax[1]);
    for (uint32_t i = 0; i < n; ++i) {
        // every normalised tf of the list within [2^-60, 2^60]: with a field boost within [2^-40, 2^39] (checked per query by
        // the kernel) every contribution S is a positive normal number under 2^100 — `applied` holds for every posting of the
        // list without looking at it (bm25.rs: S normal, the term a number)
        const float lo = minmax[2 * i], hi = minmax[2 * i + 1];
        if (lo >= 0x1p-60f && hi <= 0x1p60f) p->acc_off_of_list[lists[i]] = 1 + (uint64_t)i * 2 * words;
    }
#endif
}

// (Re)compute p->d_post_ntf for the store's current postings and average lengths.  Caller holds p->mu exclusively.
// The table is an OPTIMISATION (two IEEE divisions less per posting and query): the postings, documents and averages are
// committed before this runs, so a failure here — out of memory for the +4 B per posting, a HIP error — must not fail the
// build / append that called it (a caller that retried the append would append twice: ADVICE r04).  It leaves ntf_valid
// false — the kernels then divide themselves, same operations, same bits — and reports OK.
int refresh_post_ntf_try(orama_post* p) {
    ++p->mutations;
    {  // (every change of the postings comes through here: what was counted over the old lists is forgotten)
        std::lock_guard<std::mutex> g(p->df_union_mu);
        p->df_union.clear();
        p->shrink_hint.clear();
    }
    p->ntf_valid = false;
    p->acc_off_of_list.assign(p->n_lists, 0);
    if (p->n_postings == 0 || p->n_lists == 0) return ORAMA_OK;
    const size_t need = (size_t)p->n_postings * 4;
    if (need > p->d_post_ntf.cap) ORAMA_TRY(p->d_post_ntf.reserve(need + need / 4));  // (appends grow the arrays with the same slack)
    // list table on the device for the duration of the pass: [offsets u64 x (n_lists + 1) | field average f32 x n_lists]
    const size_t off_bytes = ((size_t)p->n_lists + 1) * 8;
    DevBuf table;
    ORAMA_TRY(table.reserve(off_bytes + (size_t)p->n_lists * 4));
    std::vector<float> list_avg(p->n_lists);
    for (uint32_t l = 0; l < p->n_lists; ++l) list_avg[l] = p->avg_len[p->field_of_list[l]];
    ORAMA_HIP_TRY(hipMemcpy(table.p, p->list_off.data(), off_bytes, hipMemcpyHostToDevice));
    ORAMA_HIP_TRY(hipMemcpy(table.as<char>() + off_bytes, list_avg.data(), (size_t)p->n_lists * 4, hipMemcpyHostToDevice));
    ORAMA_TRY(launch_ntf_precompute(p->d_post_val.as<uint32_t>(), p->d_post_ntf.as<float>(), table.as<uint64_t>(),
                                    reinterpret_cast<const float*>(table.as<char>() + off_bytes), p->n_lists, p->n_postings, p->ntf_b,
                                    nullptr));
    ORAMA_HIP_TRY(hipDeviceSynchronize());
    p->ntf_valid = true;
    build_dense_accelerators(p, table.as<uint64_t>());
    return ORAMA_OK;
}
int refresh_post_ntf(orama_post* p) {
    const int st = refresh_post_ntf_try(p);
    if (st == ORAMA_OK) return ORAMA_OK;
    p->ntf_valid = false;
    const hipError_t last = hipGetLastError();  // (read AND cleared: a failed allocation leaves its code behind, the store is intact)
    if (st == ORAMA_ERR_OOM || last == hipErrorOutOfMemory) {
        // only the +4 B per posting are missing: the kernels divide themselves — same operations, same bits — and the build /
        // append that called this has committed its lists (a caller that retried it would append twice)
        clear_error();
        return ORAMA_OK;
    }
    // anything else — a launch failure, an illegal address — is a real fault of the device: it would surface on some later,
    // unrelated call with a misleading context.  The lists ARE committed; the status says what happened.
    return st;
}

// State of one resident-postings query between its two stages.
struct PostQuery {
    QueryBuffers qb;
    uint64_t touched_cap = 0;
    uint64_t cand_cap = 0;
    size_t vec_stage_off = 0;  // byte offset of the stage-2 staging area inside sc->h_in (after the segments)
    uint32_t n_vec_cap = 0;
    size_t idf_stage_off = 0;  // pinned staging / device offsets of the idf values of the staged (sharded) form
    size_t idf_dev_off = 0;
    bool hybrid = false;
    const float* omc = nullptr;
};

int post_finalize(orama_post* p, Scratch* sc, const PostQuery& stq, const orama_bm25_params* params,
                  const float* d_idf_vals);

// Stage 1 (independent of the vector leg): descriptor upload, K3 accumulate (one launch per entry rank),
// K3 finalise (+ min/max of the full-text scores when hybrid).  Everything is enqueued on sc->stream.
int post_stage1(orama_post* p, Scratch* sc, const orama_term_ref* refs, uint32_t n_refs, float b,
                const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits, bool hybrid,
                int apply_omc, uint32_t n_vec_cap, PostQuery* st, bool finalize_now = true) {
    ORAMA_TRY(check_params(params));
    ORAMA_REQUIRE(n_refs == 0 || refs, "null refs");
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
    ORAMA_SUPPORT(total_postings < 0xffffffffull, "query touches too many postings");
    hipStream_t s = sc->stream;
    st->hybrid = hybrid;
    st->touched_cap = total_postings;  // one slot per posting (first touches hold the doc, the rest are empty)
    st->cand_cap = st->touched_cap + n_vec_cap;
    ORAMA_TRY(prepare_query(sc, p->n_docs ? p->n_docs : 1, params->n_tokens, st->cand_cap, (uint32_t)total_postings,
                            &st->qb));
    const QueryBuffers& qb = st->qb;

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
    ORAMA_TRY(resolve_allow(p->ctx, sc, allow_bitmap, bitmap_bits, s, &d_allow));
    const size_t seg_bytes = segs.size() * sizeof(Bm25Seg);
    // pinned staging: [segments | stage-2 vector entries] — stage 2 must not touch bytes an in-flight copy reads
    st->vec_stage_off = (seg_bytes + 63) & ~(size_t)63;
    st->n_vec_cap = n_vec_cap;
    st->idf_stage_off = (st->vec_stage_off + (size_t)n_vec_cap * 8 + 63) & ~(size_t)63;
    ORAMA_TRY(sc->h_in.reserve(st->idf_stage_off + kMaxTokens * 4 + 64));
    st->idf_dev_off = (seg_bytes + 63) & ~(size_t)63;  // idf values live behind the segments in misc0
    ORAMA_TRY(sc->misc0.reserve(st->idf_dev_off + kMaxTokens * 4 + 16));
    if (seg_bytes) {
        memcpy(sc->h_in.p, segs.data(), seg_bytes);
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, sc->h_in.p, seg_bytes, hipMemcpyHostToDevice, s));
    }
    uint64_t virt_base = 0;
    for (uint32_t r = 0; r < max_rank; ++r) {
        Bm25Accum a;
        a.virt_base = virt_base;
        virt_base += rank_total[r];
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
        a.slots = qb.slots;
        a.touched = qb.touched;
        a.state = qb.state;
        ORAMA_TRY(launch_bm25_accumulate(p->ctx, a, s));
    }
    st->omc = (apply_omc && p->has_omc) ? p->d_omc.as<float>() : nullptr;
    if (!finalize_now) return ORAMA_OK;
    // idf per token by the host libm (calculate_idf, bm25.rs:78-82; df.max(1), token_score.rs:275) — the device
    // never evaluates a logarithm.  df is known on the host when every token expands to at most one list and no
    // filter drops postings (docs are unique in a list: df = list length); otherwise it is what K3 counted and
    // costs one read-back of 4*n_tokens bytes.
    uint32_t df[kMaxTokens] = {0};
    bool df_known = d_allow == nullptr;
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (per_token[refs[i].token] > 1) df_known = false;
        df[refs[i].token] += (uint32_t)(p->list_off[refs[i].list + 1] - p->list_off[refs[i].list]);
    }
    if (!df_known) {
        uint32_t* h_df = reinterpret_cast<uint32_t*>(sc->h_misc.as<char>() + kDfStageOff);
        ORAMA_HIP_TRY(hipMemcpyAsync(h_df, qb.state->df, (size_t)params->n_tokens * 4, hipMemcpyDeviceToHost, s));
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
        memcpy(df, h_df, (size_t)params->n_tokens * 4);
    }
    float* h_idf = reinterpret_cast<float*>(sc->h_in.as<char>() + st->idf_stage_off);
    for (uint32_t t = 0; t < params->n_tokens; ++t) {
        const float d = (float)(df[t] < 1 ? 1u : df[t]);
        h_idf[t] = log1pf((params->total_documents - d + 0.5f) / (d + 0.5f));
    }
    float* d_idf = reinterpret_cast<float*>(sc->misc0.as<char>() + st->idf_dev_off);
    ORAMA_HIP_TRY(hipMemcpyAsync(d_idf, h_idf, (size_t)params->n_tokens * 4, hipMemcpyHostToDevice, s));
    return post_finalize(p, sc, *st, params, d_idf);
}

// K3 finalise (+ min/max of the full-text scores when hybrid) with idf[t] given (host libm; a sharded index
// computes it from the global df, §8e).
int post_finalize(orama_post* p, Scratch* sc, const PostQuery& stq, const orama_bm25_params* params,
                  const float* d_idf_vals) {
    const PostQuery* st = &stq;
    const QueryBuffers& qb = st->qb;
    const bool hybrid = st->hybrid;
    hipStream_t s = sc->stream;
    Bm25Finalize f;
    f.n_tokens = params->n_tokens;
    f.k = params->k;
    f.idf_vals = d_idf_vals;
    f.use_threshold = params->use_threshold != 0;
    f.threshold = params->threshold;
    f.track_minmax = hybrid;
    f.omc_dense = hybrid ? nullptr : st->omc;  // hybrid: OMC after the combine
    f.epoch = qb.epoch;
    f.n_docs = p->n_docs;
    f.acc = qb.acc;
    f.slots = qb.slots;
    f.touched = qb.touched;
    f.n_slots = (uint32_t)st->touched_cap;
    f.state = qb.state;
    f.cand_score = qb.cand_score;
    f.cand_idx = qb.cand_idx;
    f.emit = qb.emit;
    return launch_bm25_finalize(p->ctx, f, s);
}

// Stage 2a: [hybrid: K5 combine with the vector map +] OMC.  `skip_foreign`: vector hits whose doc is not a
// document of this store belong to another shard of the index and are combined there (sharded form, §8e).
int post_combine(orama_post* p, Scratch* sc, const PostQuery& st, const uint64_t* vec_doc, const float* vec_score,
                 uint32_t n_vec, bool skip_foreign) {
    if (!st.hybrid) return ORAMA_OK;
    hipStream_t s = sc->stream;
    const QueryBuffers& qb = st.qb;
    ORAMA_REQUIRE(n_vec <= st.n_vec_cap, "hybrid: vector map larger than announced");
    HybridCombine h;
    vec_min_max(vec_score, n_vec, &h.vec_min, &h.vec_max);  // over the WHOLE vector map, owned here or not
    // map the vector map to local doc indices (host, <= limit entries) and upload it
    uint32_t* h_idx = reinterpret_cast<uint32_t*>(sc->h_in.as<char>() + st.vec_stage_off);
    uint32_t n_own = 0;
    for (uint32_t j = 0; j < n_vec; ++j) {
        uint32_t local;
        if (p->local_of(vec_doc[j], &local)) {
            h_idx[n_own++] = local;
        } else {
            ORAMA_REQUIRE(skip_foreign, "hybrid: vector hit doc %llu is not a document of this index",
                          (unsigned long long)vec_doc[j]);
        }
    }
    float* h_sc = reinterpret_cast<float*>(h_idx + n_own);
    uint32_t local;
    for (uint32_t j = 0, o = 0; j < n_vec; ++j)
        if (p->local_of(vec_doc[j], &local)) h_sc[o++] = vec_score[j];
    ORAMA_TRY(sc->misc1.reserve((size_t)n_own * 8 + 16));
    if (n_own) ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc1.p, h_idx, (size_t)n_own * 8, hipMemcpyHostToDevice, s));
    h.vec_idx = sc->misc1.as<uint32_t>();
    h.vec_score = reinterpret_cast<const float*>(sc->misc1.as<uint32_t>() + n_own);
    h.n_vec = n_own;
    h.omc_dense = st.omc;
    h.epoch = qb.epoch;
    h.cand_cap = (uint32_t)st.touched_cap;
    h.state = qb.state;
    h.cand_score = qb.cand_score;
    h.cand_idx = qb.cand_idx;
    h.emit = qb.emit;
    return launch_hybrid_combine(p->ctx, h, s);
}

// Stage 2: combine, K4 top-k, download (synchronises sc->stream).
int post_stage2(orama_post* p, Scratch* sc, const PostQuery& st, const orama_bm25_params* params,
                const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, uint64_t* out_ids,
                float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    ORAMA_TRY(post_combine(p, sc, st, vec_doc, vec_score, n_vec, false));
    return select_and_download(p->ctx, sc, st.qb, p->d_docs.as<uint64_t>(), (uint32_t)st.cand_cap, params->top_k,
                               out_ids, out_scores, out_n, out_count);
}

// ---------------------------------------------------------------- K3r: range-partitioned scoring of query batches
struct RangeJob {
    const orama_term_ref* refs;
    uint32_t n_refs;
    const orama_bm25_params* params;
    uint64_t* out_ids;
    float* out_scores;
    uint32_t* out_n;
    uint64_t* out_count;
    // hybrid (normalize_and_combine with a vector map, token_score.rs:393-422): only as a batch of one
    bool hybrid = false;
    const uint64_t* vec_doc = nullptr;
    const float* vec_score = nullptr;
    uint32_t n_vec = 0;
    bool* fallback = nullptr;  // set when the answer could not be proven exact from the candidates: run K3 instead
    // the one-call hybrid search (orama_hybrid_search): the vector map does not exist yet when the full-text leg starts.
    // The range scorer enqueues bounds + scores + the raw top-(top_k + n_vec_max + 1) and only then asks for the map — the
    // provider joins the vector leg, which ran beside all of that — so that the work left after the scan is one tiny
    // per-document launch and a merge of <= top_k + 2 n_vec_max + 1 entries on the host (DESIGN.md K5 "hybrid tail").
    std::function<int(const uint64_t** doc, const float** score, uint32_t* n)> vec_provider;
    uint32_t n_vec_max = 0;
    // ... and, round 5, the DEVICE form of that tail (hybrid_tail.hip): the vector leg's answer [ids | distances | n] stays on the
    // device; behind `vec_ready` (recorded on the vector leg's stream) the range scorer's own stream runs the a2 epilogue, the
    // per-document scoring of the hits, normalize_and_combine and K4 — one read-back, one host wake-up.  The provider is
    // only asked when the device says the candidates cannot prove the answer (the K3 fallback wants the map on the host).
    bool device_tail = false;
    const uint64_t* d_vec_ids = nullptr;
    const float* d_vec_dist = nullptr;
    const uint32_t* d_vec_n = nullptr;
    hipEvent_t vec_ready = nullptr;
    uint32_t vec_limit = 0;
    float min_similarity = 0.0f;
    int rescale_e5 = 0;
    // sharded batches (orama_shard_post_search_batch): the index-wide document frequency of every token (kMaxTokens words) —
    // idf comes from it instead of from this shard's list lengths (corpus_docs.len() over the whole index, token_score.rs:262-275)
    const uint32_t* df_global = nullptr;
    // df pass of a sharded batch (post_search_ranges(..., df_pass = true)): this shard's document frequency of every token
    // (kMaxTokens words) — list lengths where that is exact, else counted by the counting launch — and nothing else
    uint32_t* df_out = nullptr;
    // score-map mode (a batch of one, not hybrid): leave the whole map behind in the scratch set — see RangeBatch::map_idx
    QueryBuffers* map = nullptr;
    uint32_t* map_list_len = nullptr;
};

constexpr uint64_t kRangeKeyBudget = 1ull << 28;    // key slots (8 B each) one set of launches may use

// The plain top-k search of the resident store can take the K3r path when its references fit the sort key.
bool ranges_eligible(const orama_post* p, const orama_term_ref* refs, uint32_t n_refs, const orama_bm25_params* params) {
    if (!p->ctx->bm25_ranges || !params || params->n_tokens < 1 || params->n_tokens > kMaxTokens) return false;
    uint32_t nonempty = 0;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (refs[i].list >= p->n_lists) return false;  // the ordinary path reports the error
        const uint64_t len = p->list_off[refs[i].list + 1] - p->list_off[refs[i].list];
        nonempty += len != 0;
        total += len;
    }
    return nonempty <= kRangeMaxRefs && total < 0x7fffffffull;
}

// Documents per range: `target` postings per range on average (option "k3r_target"; a workgroup holds at most kRangeCap = 2 048).
// The cost per posting falls with the postings per workgroup — per-workgroup work (tables, scans, barriers) is amortised over
// more of them: round 4's sort-free kernel takes 5.03 us per C4-shaped query at 1 280, 4.78 at 1 536, 4.57 at 1 792 and 4.36 at
// 2 000 (profiles/r04_k3r_target_sweep_v3.log) — but a range that exceeds the 2 048 reruns its whole query with 8x narrower
// ranges, so rounds 3-5 kept 25 % of headroom: 1 536 (a Poisson count of that mean is 13 standard deviations below the
// cap; documents of a term clustered in id space overflow at any target).  Round 6: 1 792 (6 sigma) — since round 5 an overflow
// costs a rerun "that much narrower plus a quarter" whose factor is remembered per list set, not 8 x for good; compact-list kernel,
// one lease: 231 K queries/s at 1 536, 239 K at 1 792, 242 K at 1 920 (profiles/r06_k3r_target_sweep.log).  Any width — a power of two would leave the average
// anywhere between target / 2 and target — `shrink` times 8x smaller after an overflow.
// `narrow16`: by how much the ranges are narrower than the target asks, in 1/16 (16 = not at all).  After an overflow the scoring
// launch says by how much its worst range was too large (RangeResult::pad1[0]: postings, documents with several postings or
// cells against what a workgroup takes) and the query is rerun that much narrower plus a quarter — 1.5 x at least; round 4
// went 8 x narrower whatever the excess: lists over the same documents (a term in two fields) ran at a third of the rate of
// independent lists ever after (scripts/bench_overlap_lists.py).  The factor that held is remembered per list set
// (orama_post::shrink_hint).
uint32_t choose_width(uint64_t n_docs, uint64_t total_postings, uint32_t narrow16, uint32_t target_opt) {
    const uint64_t target = target_opt >= 16 && target_opt <= kRangeCap ? target_opt : 7 * kRangeThreads;  // option "k3r_target" (0: 7/8 of what a workgroup holds)
    uint64_t w = total_postings ? n_docs * target / total_postings : n_docs;
    if (narrow16 > 16u) w = w * 16u / narrow16;
    if (w >= 64) w &= ~31ull;  // whole bitmap words per range: a dense list's documents are read as words (RangeSeg::acc_off)
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(w, 1), kRangeMaxWidth);
}
uint32_t narrower_after_overflow(uint32_t narrow16, uint32_t excess16) {
    const uint64_t cur = std::max(narrow16, 16u);
    // (the excess is measured at the current width: narrower by that factor again, a quarter on top for the ranges' spread)
    uint64_t next = cur * std::max(excess16, 17u) / 16u;
    next = next + next / 4;
    next = std::max<uint64_t>(next, cur + cur / 2);
    return (uint32_t)std::min<uint64_t>(next, 1u << 30);
}

// Hybrid answer from the range scorer's outputs (a batch of one, no OMC): normalize_and_combine + count + top_n
// (token_score.rs:393-422, search.rs:482, sort.rs:260-279) evaluated on the host over
//   cand      the best k_asked = top_k + n_vec + 1 full-text documents by (raw score desc, doc asc),
//   vft/vpr   the full-text score of every vector hit (same fold, range_score_docs_kernel) / whether it is in the map,
//   res       count of the full-text map and the extremes of its non-NaN scores.
// (s - min) / (max - min) is monotone in s, so the final top_k among full-text-only documents lies inside the raw
// top-(top_k + n_vec + 1) — unless rounding maps the last candidate's score onto the k-th final score (then an unseen
// document with a smaller id could tie its way in): that case is handed to the per-record scorer (*fallback).
// Same f32 operations as K5's kernels (this translation unit is compiled with -ffp-contract=off).
int hybrid_from_candidates(const RangeJob& jb, const RangeResult& res, const uint64_t* cand_id, const float* cand_score,
                           uint32_t n_cand, uint32_t k_asked, const float* vft, const uint32_t* vpr) {
    const uint32_t top_k = jb.params->top_k, nv = jb.n_vec;
    float vmn, vmx;
    vec_min_max(jb.vec_score, nv, &vmn, &vmx);
    float mx = 0.0f, mn = 0.0f;  // fold(0.0, f32::max / f32::min) over both maps
    if (vmx > mx) mx = vmx;
    if (vmn < mn) mn = vmn;
    auto ordered_to_f32 = [](uint32_t key) {
        const uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    if (res.max_key != 0u) {
        const float v = ordered_to_f32(res.max_key);
        if (v > mx) mx = v;
    }
    if (res.min_inv != 0u) {
        const float v = ordered_to_f32(~res.min_inv);
        if (v < mn) mn = v;
    }
    const float den = mx - mn;
    struct Entry {
        float score;
        uint64_t doc;
    };
    std::vector<Entry> out;
    out.reserve((size_t)n_cand + nv);
    // vector hits: ft' + v' when the document is in the full-text map, 0.0 + v' otherwise
    uint64_t count = res.count;
    for (uint32_t j = 0; j < nv; ++j) {
        const float v = (jb.vec_score[j] - mn) / den;
        float sc;
        if (vpr[j]) {
            sc = (vft[j] - mn) / den;
            sc = sc + v;
        } else {
            sc = 0.0f + v;
            ++count;
        }
        out.push_back(Entry{sc, jb.vec_doc[j]});
    }
    // full-text-only candidates
    float last_norm = 0.0f;
    bool have_last = false;
    for (uint32_t i = 0; i < n_cand; ++i) {
        const float sn = (cand_score[i] - mn) / den;
        if (i == n_cand - 1) {
            last_norm = sn;
            have_last = true;
        }
        bool is_vec = false;
        for (uint32_t j = 0; j < nv && !is_vec; ++j) is_vec = jb.vec_doc[j] == cand_id[i];
        if (!is_vec) out.push_back(Entry{sn, cand_id[i]});
    }
    // top_n: NaN never selected; score desc, DocumentId asc; -0.0 comes back as +0.0 (K4's key canonicalises the zero)
    std::vector<Entry> sel;
    sel.reserve(out.size());
    for (Entry e : out)
        if (e.score == e.score) {
            if (e.score == 0.0f) e.score = 0.0f;
            sel.push_back(e);
        }
    std::sort(sel.begin(), sel.end(), [](const Entry& a, const Entry& b) { return a.score > b.score || (a.score == b.score && a.doc < b.doc); });
    const uint32_t n_out = (uint32_t)std::min<size_t>(sel.size(), top_k);
    // exactness: were there full-text documents beyond the candidates that could still enter the top_k?
    if (n_cand >= k_asked && have_last && top_k > 0) {
        const bool full = sel.size() >= top_k;
        // an unseen document normalises to <= last_norm; it cannot displace anything when the k-th final score is strictly above
        if (!full || !(sel[top_k - 1].score > last_norm)) {
            if (last_norm == last_norm) {  // NaN: nothing unseen can be selected either
                *jb.fallback = true;
                return ORAMA_OK;
            }
        }
    }
    for (uint32_t i = 0; i < n_out; ++i) {
        jb.out_ids[i] = sel[i].doc;
        jb.out_scores[i] = sel[i].score;
    }
    *jb.out_n = n_out;
    if (jb.out_count) *jb.out_count = count;
    return ORAMA_OK;
}

// ORAMA_POST_CALL_TRACE=1: where the HOST spends a range-scorer call (mean microseconds per phase over every 2 000 chunks,
// on stderr) — table building, each enqueue, the wait, the hand-out.  Single-threaded use only (plain statics).
struct CallTrace {
    static constexpr int kMarks = 8;
    static bool on() {
        static const bool v = orama::dev_env("ORAMA_POST_CALL_TRACE") != nullptr;
        return v;
    }
    std::chrono::steady_clock::time_point t[kMarks];
    void mark(int i) {
        if (on()) t[i] = std::chrono::steady_clock::now();
    }
    void done() {
        if (!on()) return;
        static double sum[kMarks] = {0};
        static uint32_t n = 0;
        for (int i = 1; i < kMarks; ++i) sum[i] += std::chrono::duration<double, std::micro>(t[i] - t[i - 1]).count();
        if (++n == 2000) {
            fprintf(stderr, "post call trace (us, mean of %u): tables %.1f | upload enqueue %.1f | bounds launch %.1f | score launch %.1f | "
                    "top-k launches %.1f | read-back enqueue %.1f | wait %.1f\n", n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n,
                    sum[5] / n, sum[6] / n, sum[7] / n);
            for (double& x : sum) x = 0;
            n = 0;
        }
    }
};

// ORAMA_K3R_STATS=1: one stderr line per query scored with compact key lists (keys appended against postings).
static bool k3r_stats_enabled() {
    static const bool on = [] { const char* e = orama::dev_env("ORAMA_K3R_STATS"); return e && std::atoi(e) != 0; }();
    return on;
}

static std::atomic<uint32_t> k3r_stats_lines{0};  // (the first 24 queries of the process)

// Score `n_jobs` eligible queries (each validated by check_params and ranges_eligible) on sc->stream; synchronises.
// With a second set (`sc2`) the sets of launches are double-buffered: while the device scores one chunk of 32 queries, the
// host builds and uploads the tables of the next one on the other set's stream (the host side of a chunk — reference
// tables, idf by libm, the upload — costs about as much wall time as its launches take on the device).
int post_search_ranges(orama_post* p, Scratch* sc, const RangeJob* jobs, uint32_t n_jobs, float b,
                       const uint64_t* allow_bitmap, uint64_t bitmap_bits, int apply_omc, Scratch* sc2 = nullptr, bool df_pass = false) {
    struct Pending {
        uint32_t job;
        uint32_t shrink;
        uint64_t total;
        uint32_t recalled = 0;  // the hint the store held for these lists when the query was queued (0 = none)
    };
    // hybrid (a batch of one): the vector map as local document indices; a hit that is not a document of this index is the
    // per-record scorer's business (it reports the error)
    std::vector<uint32_t> vec_local;
    RangeJob hj;  // the hybrid job with its vector map filled in
    bool have_map = false;
    auto take_map = [&]() -> int {
        if (have_map) return ORAMA_OK;
        hj = jobs[0];
        if (hj.vec_provider) ORAMA_TRY(hj.vec_provider(&hj.vec_doc, &hj.vec_score, &hj.n_vec));
        have_map = true;
        vec_local.resize(hj.n_vec);
        for (uint32_t j = 0; j < hj.n_vec; ++j)
            if (!p->local_of(hj.vec_doc[j], &vec_local[j])) *hj.fallback = true;
        return ORAMA_OK;
    };
    const bool hybrid_job = n_jobs == 1 && jobs[0].hybrid;
    // the device form of the hybrid tail: hits and merged entries that fit its kernels (else the host form below)
    const bool device_tail = hybrid_job && jobs[0].device_tail && jobs[0].vec_provider && jobs[0].vec_limit >= 1 &&
                             jobs[0].vec_limit <= kHybridTailMaxVec &&
                             (uint64_t)jobs[0].params->top_k + 2ull * jobs[0].vec_limit + 1 <= kSelectMaxK;
    if (hybrid_job) {
        *jobs[0].fallback = false;
        if (!jobs[0].vec_provider) {
            ORAMA_TRY(take_map());
            if (*hj.fallback) return ORAMA_OK;
        }
    }
    std::deque<Pending> pending;
    for (uint32_t j = 0; j < n_jobs; ++j) {
        const RangeJob& jb = jobs[j];
        *jb.out_n = 0;
        if (jb.out_count) *jb.out_count = 0;
        ORAMA_REQUIRE(df_pass || jb.params->top_k == 0 || (jb.out_ids && jb.out_scores), "null output");
        ORAMA_REQUIRE(jb.n_refs == 0 || jb.refs, "null refs");
        uint64_t total = 0;
        for (uint32_t i = 0; i < jb.n_refs; ++i) {
            ORAMA_REQUIRE(jb.refs[i].token < jb.params->n_tokens, "ref %u: token %u >= n_tokens %u", i, jb.refs[i].token,
                          jb.params->n_tokens);
            ORAMA_REQUIRE(jb.refs[i].list < p->n_lists, "ref %u: list %u out of range", i, jb.refs[i].list);
            total += p->list_off[jb.refs[i].list + 1] - p->list_off[jb.refs[i].list];
        }
        if (!total && jb.hybrid) {  // no full-text side at all: the per-record path combines the vector map alone
            *jb.fallback = true;
            return ORAMA_OK;
        }
        if (df_pass) {
            // df is the list length when no filter drops postings and every token has one list; otherwise it is counted
            ORAMA_REQUIRE(jb.df_out, "internal: df pass without an output");
            uint32_t lists_of[kMaxTokens] = {0};
            bool counted = allow_bitmap != nullptr;
            for (uint32_t t = 0; t < kMaxTokens; ++t) jb.df_out[t] = 0;
            for (uint32_t i = 0; i < jb.n_refs; ++i) {
                const uint32_t len = (uint32_t)(p->list_off[jb.refs[i].list + 1] - p->list_off[jb.refs[i].list]);
                if (!len) continue;
                if (++lists_of[jb.refs[i].token] > 1) counted = true;
                jb.df_out[jb.refs[i].token] += len;
            }
            if (!counted || !total) continue;
            // (counted before, under this filter / for these sets of lists?  then the host has the answer)
            if (df_recall(p, jb.refs, jb.n_refs, jb.params->n_tokens, allow_bitmap != nullptr, allow_content_version(p->ctx, allow_bitmap),
                          bitmap_bits, jb.df_out))
                continue;
        }
        if (total) {
            const uint32_t hint = shrink_recall(p, jb.refs, jb.n_refs);
            pending.push_back({j, hint, total, hint});  // (shrink = narrow16: 0 / 16 = the target's width)
        }
    }
    ORAMA_REQUIRE(p->n_docs > 0 || pending.empty(), "postings store is empty (orama_post_build not called)");
    // Queries of similar length share a set of launches: the launches of a chunk are sized by its longest query (grid of the
    // scoring launch, stride of the key lists, chunks of the top-k reduction), and a workgroup that finds nothing to do still
    // has to be launched — and, in the reduction, to write its k empty outputs.
    if (pending.size() > kRangeBatchMax)
        std::stable_sort(pending.begin(), pending.end(), [](const Pending& a, const Pending& c) { return a.total > c.total; });

    // One set of launches: up to kRangeBatchMax queries whose padded key lists fit the budget.
    struct Chunk {
        Scratch* sc = nullptr;
        const uint64_t* d_allow = nullptr;
        uint64_t allow_version = 0;  // content version of a resident filter bitmap (0: none, or host words)
        bool allow_resolved = false;
        std::vector<Pending> members;
        std::vector<RangeSeg> segs;
        std::vector<RangeQuery> queries;
        std::vector<uint32_t> lens;
        CallTrace trace;
        uint32_t nq = 0, kmax = 0, kk = 1;
        uint64_t max_total = 0;
        size_t res_bytes = 0;
        RangeBatch rb;
        RangeResult* h_res = nullptr;
        const char* h_tail = nullptr;  // device tail: [flag u32 | n u32 | count u64 | ids top_k x u64 | scores top_k x f32] once the stream drained
    };
    Chunk slots[2];
    slots[0].sc = sc;
    slots[1].sc = sc2;
    const uint32_t n_slots = sc2 && !hybrid_job ? 2u : 1u;

    // build the chunk's tables, upload them, enqueue bounds + scores + top-k + the read-back on the chunk's stream
    auto enqueue = [&](Chunk& c) -> int {
        Scratch* sc = c.sc;
        hipStream_t s = sc->stream;
        c.trace.mark(0);
        if (!c.allow_resolved) {
            ORAMA_TRY(resolve_allow(p->ctx, sc, allow_bitmap, bitmap_bits, s, &c.d_allow, &c.allow_version));
            c.allow_resolved = true;
        }
        const uint64_t* d_allow = c.d_allow;
        c.members.clear();
        c.max_total = 0;
        c.kmax = 0;
        while (!pending.empty() && c.members.size() < kRangeBatchMax) {
            const Pending& pd = pending.front();
            const uint64_t mt = std::max(c.max_total, pd.total);
            if (!c.members.empty() && mt * (c.members.size() + 1) > kRangeKeyBudget) break;
            c.max_total = mt;
            const RangeJob& jb = jobs[pd.job];
            c.kmax = std::max(c.kmax, jb.hybrid ? jb.params->top_k + (jb.vec_provider ? jb.n_vec_max : jb.n_vec) + 1 : jb.params->top_k);
            c.members.push_back(pd);
            pending.pop_front();
        }
        const uint32_t nq = c.nq = (uint32_t)c.members.size();
        // stride of the key lists: even, so that every list starts 16-byte aligned (the top-k streams them with 16-byte loads)
        const uint64_t max_total = c.max_total = (c.max_total + 1) & ~1ull;
        std::vector<RangeSeg>& segs = c.segs;
        std::vector<RangeQuery>& queries = c.queries;
        segs.clear();
        queries.assign(nq, RangeQuery{});
        c.lens.assign(nq, 0);
        uint64_t virt = 0, bounds_entries = 0, max_bound_entries = 0;
        uint32_t max_ranges = 0;
        bool any_df = false;
        // (+ the hybrid job's vector hits: sized now so that nothing is re-allocated behind the scan)
        ORAMA_TRY(sc->h_misc.reserve((size_t)nq * kMaxTokens * 4 + 4096 + (size_t)(c.kmax + 1) * 12 + 64));
        float* h_idf = sc->h_misc.as<float>();
        for (uint32_t ci = 0; ci < nq; ++ci) {
            const Pending& pd = c.members[ci];
            const RangeJob& jb = jobs[pd.job];
            RangeQuery& q = queries[ci];
            q.key_off = (uint64_t)ci * max_total;
            q.bounds_base = bounds_entries;
            q.seg_begin = (uint32_t)segs.size();
            q.width = choose_width(p->n_docs, pd.total, pd.shrink, p->ctx->k3r_target);
            q.n_ranges = (uint32_t)((p->n_docs - 1) / q.width + 1);
            q.n_tokens = jb.params->n_tokens;
            q.use_threshold = jb.params->use_threshold != 0;
            q.threshold = jb.params->threshold;
            q.k = jb.params->k;
            q.track_minmax = jb.hybrid ? 1u : 0u;
            q.topk = jb.params->top_k;
            c.lens[ci] = (uint32_t)pd.total;
            max_ranges = std::max(max_ranges, q.n_ranges);
            uint32_t per_token[kMaxTokens] = {0}, df[kMaxTokens] = {0};
            bool df_known = d_allow == nullptr, multi_list = false;
            // references in (token, reference order): the rank of a list among its token's lists is its position
            for (uint32_t t = 0; t < q.n_tokens; ++t) {
                for (uint32_t i = 0; i < jb.n_refs; ++i) {
                    if (jb.refs[i].token != t) continue;
                    const uint32_t l = jb.refs[i].list;
                    const uint32_t len = (uint32_t)(p->list_off[l + 1] - p->list_off[l]);
                    if (len == 0) continue;
                    RangeSeg g{};
                    g.post_begin = p->list_off[l];
                    g.len = len;
                    g.tok_rank = (t << 10) | per_token[t];
                    g.boost = jb.refs[i].boost;
                    g.avg_len = p->avg_len[p->field_of_list[l]];
                    g.acc_off = l < p->acc_off_of_list.size() ? p->acc_off_of_list[l] : 0;
                    virt += len;
                    segs.push_back(g);
                    if (++per_token[t] > 1) {
                        df_known = false;
                        multi_list = true;
                    }
                    df[t] += len;
                }
            }
            if ((multi_list || d_allow != nullptr) && !jb.df_global && !df_pass)  // what an earlier query had counted
                df_known = df_recall(p, jb.refs, jb.n_refs, q.n_tokens, d_allow != nullptr, c.allow_version, bitmap_bits, df);
            q.seg_end = (uint32_t)segs.size();
            const uint32_t ns = q.seg_end - q.seg_begin;
            bounds_entries += ((uint64_t)q.n_ranges + 1) * ns;
            max_bound_entries = std::max<uint64_t>(max_bound_entries, ((uint64_t)q.n_ranges + 1) * ns);
            if (jb.df_global) {  // the index-wide df of a sharded index: nothing left to count here
                for (uint32_t t = 0; t < kMaxTokens; ++t) df[t] = jb.df_global[t];
                df_known = true;
            }
            q.want_df = df_known ? 0u : (multi_list ? 1u : 2u);
            any_df |= !df_known;
            // idf per token by the host libm (calculate_idf, bm25.rs:78-82; df.max(1), token_score.rs:275)
            for (uint32_t t = 0; t < kMaxTokens; ++t) {
                const float d = (float)(df[t] < 1 ? 1u : df[t]);
                h_idf[(size_t)ci * kMaxTokens + t] =
                    t < q.n_tokens ? log1pf((jb.params->total_documents - d + 0.5f) / (d + 0.5f)) : 0.0f;
            }
        }
        ORAMA_SUPPORT(virt < 0xffffffffull && bounds_entries < 0xffffffffull, "query batch references too many postings");
        // Compact key lists (round 5): a plain top-k batch — no score map, OMC, hybrid min / max, 64-bit masks, and the store's
        // pre-divided tf at hand (the conditions of the PLAIN scoring launch) — appends only the keys that can still reach the
        // answer; the list lengths the top-k reads are then the cursors the scoring launch counted up (RangeResult::n_keys).
        // (not for small batches: orama_ctx::bm25_compact_min)
        bool compact = !df_pass && c.kmax != 0 && p->ctx->bm25_compact_keys && nq >= p->ctx->bm25_compact_min && !(n_jobs == 1 && jobs[0].map) && !(apply_omc && p->has_omc) &&
                       p->ntf_valid && b == p->ntf_b;
        for (uint32_t ci = 0; ci < nq && compact; ++ci)
            compact = !queries[ci].track_minmax && queries[ci].seg_end - queries[ci].seg_begin <= 32u;
        // device tables: [segs | queries | idf | list lengths]
        const size_t seg_bytes = (segs.size() * sizeof(RangeSeg) + 63) & ~(size_t)63;
        const size_t q_bytes = ((size_t)nq * sizeof(RangeQuery) + 63) & ~(size_t)63;
        const size_t idf_bytes = (size_t)nq * kMaxTokens * 4;
        const size_t len_bytes = ((size_t)nq * 4 + 63) & ~(size_t)63;
        // (compact key lists: + the stripe table of the scoring launch's walk — it travels with the tables instead of riding in
        // every launch's kernel arguments: 1 KB that a lone query's launches carried for nothing)
        const size_t stripe_bytes = compact ? (((size_t)kRangeStripes * kRangeBatchMax + 1) * 4 + 63) & ~(size_t)63 : 0;
        const size_t tables_bytes = seg_bytes + q_bytes + idf_bytes + len_bytes + stripe_bytes;
        ORAMA_TRY(sc->h_in.reserve(tables_bytes));
        char* h = sc->h_in.as<char>();
        memcpy(h, segs.data(), segs.size() * sizeof(RangeSeg));
        memcpy(h + seg_bytes, queries.data(), (size_t)nq * sizeof(RangeQuery));
        memcpy(h + seg_bytes + q_bytes, h_idf, idf_bytes);
        memcpy(h + seg_bytes + q_bytes + idf_bytes, c.lens.data(), (size_t)nq * 4);
        uint32_t stripe_total = 0;
        if (compact) {  // the scoring launch's walk over the batch: kRangeStripes passes over the queries (RangeBatch::stripe_start)
            uint32_t* st_tab = reinterpret_cast<uint32_t*>(h + seg_bytes + q_bytes + idf_bytes + len_bytes);
            uint32_t w = 0;
            for (uint32_t st = 0; st < kRangeStripes; ++st)
                for (uint32_t ci = 0; ci < kRangeBatchMax; ++ci) {
                    st_tab[st * kRangeBatchMax + ci] = w;
                    if (ci < nq) {
                        const uint64_t n = queries[ci].n_ranges;
                        w += (uint32_t)((st + 1) * n / kRangeStripes - st * n / kRangeStripes);
                    }
                }
            st_tab[kRangeStripes * kRangeBatchMax] = stripe_total = w;
        }
        ORAMA_TRY(sc->misc0.reserve(tables_bytes));
        c.trace.mark(1);
        ORAMA_TRY(stage_block(p->ctx, sc->misc0.p, h, tables_bytes, hipMemcpyHostToDevice, s));
        c.trace.mark(2);
        ORAMA_TRY(sc->misc1.reserve((size_t)bounds_entries * 4));
        // device results in ONE block: [RangeResult x nq | ids | scores | n]  -> one read-back per chunk
        const uint32_t kk = c.kk = std::max(c.kmax, 1u);
        const size_t res_bytes = c.res_bytes = (size_t)nq * sizeof(RangeResult);
        const size_t out_bytes = res_bytes + (size_t)nq * kk * 12 + (size_t)nq * 4;
        const size_t pub_off = (out_bytes + 255) & ~(size_t)255;  // (behind what is read back: the published scores of compact lists)
        ORAMA_TRY(sc->misc2.reserve(pub_off + (compact ? (size_t)nq * kScorePubRanges * 4 : 0)));
        if (!df_pass) ORAMA_TRY(sc->misc3.reserve((size_t)nq * max_total * 8));
        char* d = sc->misc0.as<char>();
        float* d_idf = reinterpret_cast<float*>(d + seg_bytes + q_bytes);
        RangeBatch& rb = c.rb;
        rb = RangeBatch{};
        rb.segs = reinterpret_cast<const RangeSeg*>(d);
        rb.queries = reinterpret_cast<const RangeQuery*>(d + seg_bytes);
        rb.n_segs = (uint32_t)segs.size();
        rb.n_queries = nq;
        rb.total_postings = virt;
        rb.max_ranges = max_ranges;
        uint64_t pairs = 0;  // (query, range) pairs = workgroups of the scoring launch
        for (uint32_t ci = 0; ci < nq; ++ci) {
            pairs += queries[ci].n_ranges;
            ORAMA_SUPPORT(pairs < 0x7fffffffull, "query batch needs too many document ranges");
            rb.range_start[ci + 1] = (uint32_t)pairs;
            rb.max_refs = std::max(rb.max_refs, queries[ci].seg_end - queries[ci].seg_begin);
            rb.any_minmax |= queries[ci].track_minmax;
        }
        if (compact) {
            rb.stripe_start = reinterpret_cast<const uint32_t*>(sc->misc0.as<char>() + seg_bytes + q_bytes + idf_bytes + len_bytes);
            rb.stripe_total = stripe_total;
        }
        rb.max_bound_entries = max_bound_entries;
        rb.post_doc = p->d_post_doc.as<uint32_t>();
        rb.post_val = p->d_post_val.as<uint32_t>();
        rb.post_ntf = (p->ntf_valid && b == p->ntf_b) ? p->d_post_ntf.as<float>() : nullptr;
        rb.post_acc = (rb.post_ntf && p->acc_words) ? p->d_acc.as<uint32_t>() : nullptr;
        rb.acc_words = p->acc_words;
        rb.bounds = sc->misc1.as<uint32_t>();
        rb.docs = p->dense ? nullptr : p->d_docs.as<uint64_t>();
        rb.dense_base = p->dense_base;
        rb.allow = d_allow;
        rb.allow_bits = bitmap_bits;
        rb.b = b;
        rb.idf = d_idf;
        rb.omc_dense = (apply_omc && p->has_omc) ? p->d_omc.as<float>() : nullptr;
        rb.keys = sc->misc3.as<unsigned long long>();
        rb.results = sc->misc2.as<RangeResult>();
        rb.compact_keys = compact ? 1u : 0u;
        if (compact) rb.score_pub = reinterpret_cast<uint32_t*>(sc->misc2.as<char>() + pub_off);
        if (n_jobs == 1 && jobs[0].map) {
            // the per-document table is the set's epoch-stamped one (shared with the per-record scorer's use of the set)
            ORAMA_TRY(reserve_zeroed(sc->bm25_emit, (size_t)p->n_docs * 8, s));
            if (sc->bm25_epoch == 0xffffffffu) {  // wrap: forget every stamp
                if (sc->bm25_acc.p) ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_acc.p, 0, sc->bm25_acc.cap, s));
                ORAMA_HIP_TRY(hipMemsetAsync(sc->bm25_emit.p, 0, sc->bm25_emit.cap, s));
                sc->bm25_epoch = 0;
            }
            ORAMA_TRY(sc->misc5.reserve((size_t)max_total * 4));
            ORAMA_TRY(sc->dist.reserve((size_t)max_total * 4));
            rb.map_idx = sc->misc5.as<uint32_t>();
            rb.map_score = sc->dist.as<float>();
            rb.map_emit = sc->bm25_emit.as<unsigned long long>();
            rb.map_epoch = ++sc->bm25_epoch;
            QueryBuffers* m = jobs[0].map;
            *m = QueryBuffers{};
            m->cand_score = rb.map_score;
            m->cand_idx = rb.map_idx;
            m->emit = rb.map_emit;
            m->epoch = rb.map_epoch;
            *jobs[0].map_list_len = (uint32_t)c.members[0].total;  // (the stride may be one slot longer: that slot is never written)
        }
        if (const char* e = orama::dev_env("ORAMA_K3R_DBG")) rb.debug = (uint32_t)std::atoi(e);
        ORAMA_TRY(launch_range_bounds(p->ctx, rb, s));
        c.trace.mark(3);
        ORAMA_TRY(sc->h_out.reserve(out_bytes + 64));
        RangeResult* h_res = c.h_res = sc->h_out.as<RangeResult>();
        if (df_pass) {  // count, read the result words back, nothing else (complete() hands the df out)
            ORAMA_TRY(launch_range_score(p->ctx, rb, true, s));
            ORAMA_HIP_TRY(hipMemcpyAsync(h_res, sc->misc2.p, res_bytes, hipMemcpyDeviceToHost, s));
            return ORAMA_OK;
        }
        if (any_df) {
            // df counted on the device (filter, or tokens with several lists): one read-back, then idf by the host libm
            ORAMA_TRY(launch_range_score(p->ctx, rb, true, s));
            ORAMA_HIP_TRY(hipMemcpyAsync(h_res, sc->misc2.p, res_bytes, hipMemcpyDeviceToHost, s));
            ORAMA_HIP_TRY(hipStreamSynchronize(s));
            for (uint32_t ci = 0; ci < nq; ++ci) {
                if (!queries[ci].want_df) continue;
                const RangeJob& jb = jobs[c.members[ci].job];
                const orama_bm25_params* pr = jb.params;
                for (uint32_t t = 0; t < queries[ci].n_tokens; ++t) {
                    const float dd = (float)(h_res[ci].df[t] < 1 ? 1u : h_res[ci].df[t]);
                    h_idf[(size_t)ci * kMaxTokens + t] = log1pf((pr->total_documents - dd + 0.5f) / (dd + 0.5f));
                }
                if (!h_res[ci].overflow)  // what this query had counted, for the queries to come
                    df_remember(p, jb.refs, jb.n_refs, queries[ci].n_tokens, d_allow != nullptr, c.allow_version, bitmap_bits, h_res[ci].df);
            }
            ORAMA_HIP_TRY(hipMemcpyAsync(d_idf, h_idf, idf_bytes, hipMemcpyHostToDevice, s));
        }
        ORAMA_TRY(launch_range_score(p->ctx, rb, false, s));
        c.trace.mark(4);
        char* d_out = sc->misc2.as<char>();
        uint64_t* d_ids = reinterpret_cast<uint64_t*>(d_out + res_bytes);
        float* d_val = reinterpret_cast<float*>(d_out + res_bytes + (size_t)nq * kk * 8);
        uint32_t* d_n = reinterpret_cast<uint32_t*>(d_out + res_bytes + (size_t)nq * kk * 12);
        // The answers go straight to the pinned host block (round 5): the final launch of the top-k writes ids / scores / counts
        // there and carries the queries' result words along (KeysMirror) — one launch fewer at the end of every chunk's chain
        // (a lone query: 6 launches -> 5, 49 -> 45 us of chain).  Not where something on the device still reads the answers (the
        // device tail of a hybrid call), and not with copy commands asked for.
        const bool direct_out = c.kmax && !device_tail && p->ctx->stage_by_kernel && p->ctx->direct_out;
        if (c.kmax) {
            ORAMA_TRY(sc->misc4.reserve((size_t)keys_topk_scratch_keys((uint32_t)max_total, nq, c.kmax) * 8 + 8));
            char* o = direct_out ? reinterpret_cast<char*>(h_res) : d_out;
            KeysMirror mir;
            mir.src = reinterpret_cast<const uint32_t*>(d_out);
            mir.dst = reinterpret_cast<uint32_t*>(h_res);
            mir.words = (uint32_t)(sizeof(RangeResult) / 4);
            ORAMA_TRY(launch_keys_topk(p->ctx, rb.keys, (uint32_t)max_total, max_total, nq, c.kmax, true, p->d_docs.as<uint64_t>(),
                                       sc->misc4.as<unsigned long long>(), nullptr, reinterpret_cast<uint64_t*>(o + res_bytes),
                                       reinterpret_cast<float*>(o + res_bytes + (size_t)nq * kk * 8),
                                       reinterpret_cast<uint32_t*>(o + res_bytes + (size_t)nq * kk * 12), s,
                                       compact ? &rb.results[0].n_keys : reinterpret_cast<const uint32_t*>(d + seg_bytes + q_bytes + idf_bytes),
                                       &rb.results[0].topk_tau, (uint32_t)(sizeof(RangeResult) / 8), nullptr, compact,
                                       compact ? (uint32_t)(sizeof(RangeResult) / 4) : 1u, direct_out ? &mir : nullptr));
        }
        c.trace.mark(5);
        c.h_tail = nullptr;
        if (device_tail) {
            // ---- the hybrid tail on this stream (hybrid_tail.hip): waits for the vector leg's top-k, then four small launches
            const RangeJob& jb = jobs[0];
            const uint32_t L = jb.vec_limit, top_k = jb.params->top_k, cap = c.kmax + L;  // entries: <= k_asked + vector hits
            const size_t o_vdoc = 0, o_edoc = o_vdoc + (size_t)L * 8, o_vsc = o_edoc + (size_t)cap * 8, o_vloc = o_vsc + (size_t)L * 4,
                         o_vft = o_vloc + (size_t)L * 4, o_vpr = o_vft + (size_t)L * 4, o_esc = o_vpr + (size_t)L * 4,
                         o_state = o_esc + (size_t)cap * 4, o_out = (o_state + 16 + 15) & ~(size_t)15;
            const size_t out_tail = 16 + (size_t)top_k * 12;
            ORAMA_TRY(sc->misc5.reserve(o_out + out_tail + 16));
            char* t = sc->misc5.as<char>();
            HybridTailArgs ta;
            ta.v_ids = jb.d_vec_ids;
            ta.v_dist = jb.d_vec_dist;
            ta.v_n = jb.d_vec_n;
            ta.limit = L;
            ta.min_similarity = jb.min_similarity;
            ta.rescale_e5 = jb.rescale_e5;
            ta.docs = p->d_docs.as<uint64_t>();
            ta.n_docs = p->n_docs;
            ta.dense_base = p->dense_base;
            ta.dense = p->dense ? 1 : 0;
            ta.vdoc = reinterpret_cast<uint64_t*>(t + o_vdoc);
            ta.e_doc = reinterpret_cast<uint64_t*>(t + o_edoc);
            ta.vsc = reinterpret_cast<float*>(t + o_vsc);
            ta.vlocal = reinterpret_cast<uint32_t*>(t + o_vloc);
            ta.vft = reinterpret_cast<float*>(t + o_vft);
            ta.vpresent = reinterpret_cast<uint32_t*>(t + o_vpr);
            ta.e_score = reinterpret_cast<float*>(t + o_esc);
            ta.state = reinterpret_cast<uint32_t*>(t + o_state);
            ta.cand_id = d_ids;
            ta.cand_score = d_val;
            ta.cand_n = d_n;
            ta.k_asked = c.kmax;
            ta.top_k = top_k;
            ta.res_count = &rb.results[0].count;
            ta.res_max_key = &rb.results[0].max_key;
            ta.res_min_inv = &rb.results[0].min_inv;
            ta.res_overflow = &rb.results[0].overflow;
            char* d_tail = t + o_out;
            ta.out_flag = reinterpret_cast<uint32_t*>(d_tail);
            ta.out_count = reinterpret_cast<unsigned long long*>(d_tail + 8);
            ORAMA_HIP_TRY(hipStreamWaitEvent(s, jb.vec_ready, 0));
#if ORAMA_COMPARISON_KERNELS
            // (hybrid_tail.hip is a comparison unit since round 6: built, bit-identical, 40-70 us slower than the host tail —
            // profiles/r05_hybrid_device_tail_ab.log; the product library cannot switch it on: orama_ctx_set_option refuses)
            ORAMA_TRY(launch_hybrid_vec_epilogue(ta, s));
            ORAMA_TRY(launch_range_score_docs(p->ctx, rb, 0, ta.vlocal, L, ta.vft, ta.vpresent, s, ta.state));
            ORAMA_TRY(launch_hybrid_merge(ta, s));
#else
            ORAMA_REQUIRE(false, "internal: the device hybrid tail is not in this build");
#endif
            if (top_k) {
                SelectPlan sp;  // top_n: score desc, DocumentId asc, NaN already left out (sort.rs:260-279)
                sp.vals = ta.e_score;
                sp.stride = cap;
                sp.n = cap;
                sp.n_dev = ta.state + 2;
                sp.q = 1;
                sp.k = top_k;
                sp.descending = true;
                sp.id_map = ta.e_doc;
                sp.out_ids = reinterpret_cast<uint64_t*>(d_tail + 16);
                sp.out_val = reinterpret_cast<float*>(d_tail + 16 + (size_t)top_k * 8);
                sp.out_n = reinterpret_cast<uint32_t*>(d_tail + 4);
                ORAMA_TRY(launch_select(p->ctx, sp, s));
            }
            char* h_tail = sc->h_misc.as<char>() + (size_t)nq * kMaxTokens * 4 + 4096;  // (behind the idf staging: reserved above)
            ORAMA_TRY(stage_block(p->ctx, h_tail, d_tail, out_tail, hipMemcpyDeviceToHost, s));
            c.h_tail = h_tail;
        }
        if (!direct_out) ORAMA_TRY(stage_block(p->ctx, h_res, d_out, c.kmax ? out_bytes : res_bytes, hipMemcpyDeviceToHost, s));
        c.trace.mark(6);
        return ORAMA_OK;
    };

    // wait for the chunk; hand the answers out; a query whose ranges overflowed goes back to `pending` with smaller ranges
    auto complete = [&](Chunk& c) -> int {
        Scratch* sc = c.sc;
        hipStream_t s = sc->stream;
        const uint32_t nq = c.nq, kk = c.kk;
        const RangeResult* h_res = c.h_res;
        const char* h_ids = reinterpret_cast<const char*>(h_res) + c.res_bytes;
        const char* h_val = h_ids + (size_t)nq * kk * 8;
        const char* h_n = h_val + (size_t)nq * kk * 4;
        uint32_t nv = 0;
        float* h_vft = nullptr;          // full-text score of every vector hit ...
        uint32_t* h_vpresent = nullptr;  // ... and whether it is in the full-text map at all
        if (device_tail) {
            // one wake-up: the answer, or the word that says why not
            ORAMA_HIP_TRY(hipStreamSynchronize(s));
            const uint32_t flag = *reinterpret_cast<const uint32_t*>(c.h_tail);
            if (!(flag & 4u)) {  // (overflow: the loop below reruns the full-text leg with narrower ranges, tail included)
                const RangeJob& jb = jobs[0];
                if (flag & 3u) {  // the candidates cannot prove the answer / a hit outside the index: K3, with the map on the host
                    ORAMA_TRY(take_map());
                    *hj.fallback = true;
                    return ORAMA_OK;
                }
                const uint32_t top_k = jb.params->top_k;
                const uint32_t n = top_k ? std::min(*reinterpret_cast<const uint32_t*>(c.h_tail + 4), top_k) : 0u;
                memcpy(jb.out_ids, c.h_tail + 16, (size_t)n * 8);
                memcpy(jb.out_scores, c.h_tail + 16 + (size_t)top_k * 8, (size_t)n * 4);
                *jb.out_n = n;
                if (jb.out_count) *jb.out_count = *reinterpret_cast<const unsigned long long*>(c.h_tail + 8);
                return ORAMA_OK;
            }
        } else if (hybrid_job) {  // (one-call form: this is where the vector leg is joined — everything above ran beside it)
            ORAMA_TRY(take_map());
            if (*hj.fallback) {
                ORAMA_HIP_TRY(hipStreamSynchronize(s));
                return ORAMA_OK;
            }
            nv = hj.n_vec;
        }
        if (nv) {
            // the kernel reads the <= limit document indices from, and writes the scores to, pinned host memory directly:
            // two copy commands fewer on the critical path behind the scan (a few hundred bytes each way)
            ORAMA_TRY(sc->h_misc.reserve((size_t)nq * kMaxTokens * 4 + 4096 + (size_t)nv * 12));
            char* hv = sc->h_misc.as<char>() + (size_t)nq * kMaxTokens * 4 + 4096;  // behind the idf staging
            memcpy(hv, vec_local.data(), (size_t)nv * 4);
            h_vft = reinterpret_cast<float*>(hv + (size_t)nv * 4);
            h_vpresent = reinterpret_cast<uint32_t*>(hv + (size_t)nv * 8);
            ORAMA_TRY(launch_range_score_docs(p->ctx, c.rb, 0, reinterpret_cast<const uint32_t*>(hv), nv, h_vft, h_vpresent, s));
        }
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
        if (!df_pass && !hybrid_job) {
            c.trace.mark(7);
            c.trace.done();
        }
        for (uint32_t ci = 0; ci < nq; ++ci) {
            Pending pd = c.members[ci];
            const RangeJob& jb = jobs[pd.job];
            if (h_res[ci].overflow) {
                ORAMA_REQUIRE(c.queries[ci].width > 1, "internal: a one-document range overflowed");
                pd.shrink = narrower_after_overflow(pd.shrink, h_res[ci].pad1[0]);
                pending.push_back(pd);
                continue;
            }
            // the width the next query over these lists starts at — written only when it DIFFERS from what the store recalled for
            // this query (every narrowed query of a batch took df_union_mu and built a key string here, thousands of
            // times over the same lists)
            if (pd.shrink > 16u && !df_pass && pd.shrink != pd.recalled) shrink_remember(p, jb.refs, jb.n_refs, pd.shrink);
            if (df_pass) {
                memcpy(jb.df_out, h_res[ci].df, sizeof(uint32_t) * kMaxTokens);
                df_remember(p, jb.refs, jb.n_refs, jb.params->n_tokens, c.d_allow != nullptr, c.allow_version, bitmap_bits, h_res[ci].df);
                continue;
            }
            if (jb.hybrid) {
                ORAMA_TRY(hybrid_from_candidates(hj, h_res[ci], reinterpret_cast<const uint64_t*>(h_ids), reinterpret_cast<const float*>(h_val),
                                                 c.kmax ? reinterpret_cast<const uint32_t*>(h_n)[0] : 0u, c.kmax, h_vft, h_vpresent));
                continue;
            }
            if (jb.out_count) *jb.out_count = h_res[ci].count;
            if (c.rb.compact_keys && k3r_stats_enabled() && k3r_stats_lines.fetch_add(1) < 24)
                fprintf(stderr, "[k3r] query of %llu postings, %u ranges: %u keys appended (%.2f %%), top_k %u | dbg16: %u workgroups found a "
                        "published floor, %u above their own (fast body: wave iterations not scored); largest published %08x, largest local %08x | "
                        "fast body: %u wave iterations, %u kept postings of which %u of lists under the floor\n",
                        (unsigned long long)pd.total, c.queries[ci].n_ranges, h_res[ci].n_keys, 100.0 * h_res[ci].n_keys / (double)pd.total,
                        jb.params->top_k, h_res[ci].pad0[0], h_res[ci].pad0[1], h_res[ci].score_floor, h_res[ci].pad0[2], h_res[ci].pad0[3],
                        h_res[ci].pad0[5], h_res[ci].pad0[4]);
            if (jb.params->top_k) {
                const uint32_t n = std::min(reinterpret_cast<const uint32_t*>(h_n)[ci], jb.params->top_k);
                memcpy(jb.out_ids, h_ids + (size_t)ci * kk * 8, (size_t)n * 8);
                memcpy(jb.out_scores, h_val + (size_t)ci * kk * 4, (size_t)n * 4);
                *jb.out_n = n;
            }
        }
        return ORAMA_OK;
    };

    uint32_t head = 0, inflight = 0;
    int rc = ORAMA_OK;
    static const bool trace_flow = [] { const char* e = orama::dev_env("ORAMA_K3R_FLOW"); return e && std::atoi(e) != 0; }();
    const auto t_flow = std::chrono::steady_clock::now();
    auto us_flow = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_flow).count(); };
    while (rc == ORAMA_OK && (!pending.empty() || inflight)) {
        while (rc == ORAMA_OK && inflight < n_slots && !pending.empty()) {
            const double t0 = us_flow();
            rc = enqueue(slots[(head + inflight) % n_slots]);
            if (trace_flow && n_jobs > 64) fprintf(stderr, "[flow] %9.1f us: enqueued on slot %u in %.1f us (%u in flight before)\n", t0, (head + inflight) % n_slots, us_flow() - t0, inflight);
            ++inflight;  // (a chunk that failed half-way may still have launches on its stream: drained below)
        }
        if (rc != ORAMA_OK) break;
        const double t1 = us_flow();
        rc = complete(slots[head]);
        if (trace_flow && n_jobs > 64) fprintf(stderr, "[flow] %9.1f us: completed slot %u in %.1f us\n", t1, head, us_flow() - t1);
        head = (head + 1) % n_slots;
        --inflight;
    }
    if (rc != ORAMA_OK)  // nothing of the call stays in flight behind an error
        for (uint32_t i = 0; i < n_slots; ++i) (void)hipStreamSynchronize(slots[i].sc->stream);
    return rc;
}

int post_search_impl(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                     const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                     const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, bool hybrid,
                     int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                     uint64_t* out_count) {
    ORAMA_REQUIRE(p && out_n && params, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_REQUIRE(!hybrid || n_vec == 0 || (vec_doc && vec_score), "null vector map");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::shared_lock<std::shared_mutex> lk(p->mu);
    const bool eligible = ranges_eligible(p, refs, n_refs, params);
    const bool by_ranges = !hybrid && eligible;
    // hybrid on the range scorer: when no OMC applies (multipliers reorder documents arbitrarily: the candidate argument of
    // hybrid_from_candidates would not hold) and the candidates fit one selection
    const bool hybrid_by_ranges = hybrid && eligible && p->ctx->bm25_ranges_hybrid && !(apply_omc && p->has_omc) &&
                                  (uint64_t)params->top_k + n_vec + 1 <= kSelectMaxK;
    if (hybrid_by_ranges) {
        ORAMA_TRY(check_params(params));
        bool fallback = false;
        {
            ScratchLease scg(p->ctx, kScratchGeneral);
            ORAMA_TRY(scg.init());
            RangeJob job{refs, n_refs, params, out_ids, out_scores, out_n, out_count};
            job.hybrid = true;
            job.vec_doc = vec_doc;
            job.vec_score = vec_score;
            job.n_vec = n_vec;
            job.fallback = &fallback;
            ORAMA_TRY(post_search_ranges(p, scg.s.get(), &job, 1, b, allow_bitmap, bitmap_bits, 0));
        }
        if (!fallback) return ORAMA_OK;
        *out_n = 0;
        if (out_count) *out_count = 0;
    }
    ScratchLease sc(p->ctx, by_ranges ? kScratchGeneral : kScratchRecords);
    ORAMA_TRY(sc.init());
    if (by_ranges) {
        ORAMA_TRY(check_params(params));
        const RangeJob job{refs, n_refs, params, out_ids, out_scores, out_n, out_count};
        return post_search_ranges(p, sc.s.get(), &job, 1, b, allow_bitmap, bitmap_bits, apply_omc);
    }
    PostQuery st;
    ORAMA_TRY(post_stage1(p, sc.s.get(), refs, n_refs, b, params, allow_bitmap, bitmap_bits, hybrid, apply_omc, n_vec, &st));
    return post_stage2(p, sc.s.get(), st, params, vec_doc, vec_score, n_vec, out_ids, out_scores, out_n, out_count);
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
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
}

int orama_post_build(orama_post* p, const uint64_t* docs, uint64_t n_docs, uint32_t n_fields,
                     const float* avg_field_len, uint32_t n_lists, const uint32_t* field_of_list,
                     const uint64_t* list_off, const uint64_t* post_doc, const uint32_t* post_tf,
                     const uint32_t* post_len) {
    ORAMA_REQUIRE(p, "null handle");
    ORAMA_REQUIRE(n_docs == 0 || docs, "null docs");
    ORAMA_SUPPORT(n_docs < 0xffffffffull, "postings store limited to 2^32-1 documents");
    ORAMA_REQUIRE(n_fields == 0 || avg_field_len, "null avg_field_len");
    ORAMA_REQUIRE(n_lists == 0 || (field_of_list && list_off), "null list table");
    for (uint64_t i = 1; i < n_docs; ++i)
        ORAMA_REQUIRE(docs[i - 1] < docs[i], "docs must be strictly ascending (position %llu)", (unsigned long long)i);
    const uint64_t n_post = n_lists ? list_off[n_lists] : 0;
    ORAMA_REQUIRE(n_post == 0 || (post_doc && post_tf && post_len), "null postings");
    ORAMA_ON_DEVICE(p->ctx->device);
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
    p->dense = dense;
    p->dense_base = n_docs ? docs[0] : 0;
    if (dense) {
        p->h_docs.clear();
        p->h_docs.shrink_to_fit();
    } else {
        p->h_docs.assign(docs, docs + n_docs);
    }
    p->avg_len.assign(avg_field_len, avg_field_len + n_fields);
    p->field_of_list.assign(field_of_list, field_of_list + n_lists);
    p->list_off.assign(list_off, list_off + (n_lists ? n_lists + 1 : 0));
    if (!n_lists) p->list_off.assign(1, 0);
    // the multiplier array was sized and indexed for the previous doc table: call orama_post_set_omc again
    p->has_omc = false;
    p->d_omc.release();
    ++p->generation;
    return refresh_post_ntf(p);
}

// Live update between commits: new documents + delta posting lists appended in place.
int orama_post_append(orama_post* p, const uint64_t* docs, uint64_t n_new, const float* avg_field_len,
                      uint32_t n_lists_new, const uint32_t* field_of_list, const uint64_t* list_off,
                      const uint64_t* post_doc, const uint32_t* post_tf, const uint32_t* post_len) {
    ORAMA_REQUIRE(p, "null handle");
    ORAMA_REQUIRE(n_new == 0 || docs, "null docs");
    ORAMA_REQUIRE(n_lists_new == 0 || (field_of_list && list_off), "null list table");
    ORAMA_REQUIRE(p->n_fields > 0, "orama_post_append needs a built store (orama_post_build first)");
    ORAMA_REQUIRE(avg_field_len, "null avg_field_len");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::unique_lock<std::shared_mutex> lk(p->mu);
    const uint64_t n_old = p->n_docs, n_all = n_old + n_new;
    ORAMA_SUPPORT(n_all < 0xffffffffull, "postings store limited to 2^32-1 documents");
    const uint64_t last_old = n_old ? (p->dense ? p->dense_base + n_old - 1 : p->h_docs.back()) : 0;
    for (uint64_t i = 0; i < n_new; ++i) {
        ORAMA_REQUIRE((i == 0 && (n_old == 0 || docs[0] > last_old)) || (i > 0 && docs[i] > docs[i - 1]),
                      "appended docs must be ascending and greater than every stored id (position %llu)",
                      (unsigned long long)i);
    }
    // the id table after the append: stays implicit while the ids remain one dense run
    const bool dense_after = (n_old == 0 || p->dense) &&
                             (n_new == 0 || ((n_old == 0 || docs[0] == p->dense_base + n_old) &&
                                             docs[n_new - 1] - docs[0] == n_new - 1));
    std::vector<uint64_t> all_docs;
    if (!dense_after) {
        all_docs.reserve((size_t)n_all);
        if (p->dense) for (uint64_t i = 0; i < n_old; ++i) all_docs.push_back(p->dense_base + i);
        else all_docs = p->h_docs;
        all_docs.insert(all_docs.end(), docs, docs + n_new);
    }
    const uint64_t base_new = n_old ? p->dense_base : (n_new ? docs[0] : 0);
    auto local_after = [&](uint64_t id, uint32_t* out) -> bool {
        if (dense_after) {
            if (id < base_new || id - base_new >= n_all) return false;
            *out = (uint32_t)(id - base_new);
            return true;
        }
        auto it = std::lower_bound(all_docs.begin(), all_docs.end(), id);
        if (it == all_docs.end() || *it != id) return false;
        *out = (uint32_t)(it - all_docs.begin());
        return true;
    };
    const uint64_t n_post_new = n_lists_new ? list_off[n_lists_new] : 0;
    ORAMA_REQUIRE(n_post_new == 0 || (post_doc && post_tf && post_len), "null postings");
    std::vector<uint32_t> pd((size_t)n_post_new), pv((size_t)n_post_new);
    for (uint32_t l = 0; l < n_lists_new; ++l) {
        ORAMA_REQUIRE(field_of_list[l] < p->n_fields, "list %u: field %u out of range", l, field_of_list[l]);
        ORAMA_REQUIRE(list_off[l] <= list_off[l + 1], "list offsets must be non-decreasing");
        for (uint64_t i = list_off[l]; i < list_off[l + 1]; ++i) {
            ORAMA_REQUIRE(i == list_off[l] || post_doc[i] > post_doc[i - 1], "list %u: docs must be strictly ascending", l);
            uint32_t local;
            ORAMA_REQUIRE(local_after(post_doc[i], &local), "list %u: doc %llu not in the store", l,
                          (unsigned long long)post_doc[i]);
            ORAMA_REQUIRE(post_tf[i] <= 0xffffu && post_len[i] <= 0xffffu, "tf / field_length must fit u16");
            pd[(size_t)i] = local;
            pv[(size_t)i] = (post_tf[i] << 16) | post_len[i];
        }
    }
    // grow the device arrays (old contents copied device-to-device) and append
    auto grow = [](DevBuf& buf, size_t old_bytes, size_t new_bytes) -> int {
        if (new_bytes <= buf.cap) return ORAMA_OK;
        DevBuf bigger;
        ORAMA_TRY(bigger.reserve(new_bytes + new_bytes / 4));
        if (old_bytes) ORAMA_HIP_TRY(hipMemcpy(bigger.p, buf.p, old_bytes, hipMemcpyDeviceToDevice));
        std::swap(buf.p, bigger.p);
        std::swap(buf.cap, bigger.cap);
        return ORAMA_OK;
    };
    ORAMA_HIP_TRY(hipDeviceSynchronize());
    const uint64_t n_post_old = p->n_postings;
    ORAMA_TRY(grow(p->d_docs, (size_t)n_old * 8, std::max<size_t>(8, (size_t)n_all * 8)));
    ORAMA_TRY(grow(p->d_post_doc, (size_t)n_post_old * 4, std::max<size_t>(4, (size_t)(n_post_old + n_post_new) * 4)));
    ORAMA_TRY(grow(p->d_post_val, (size_t)n_post_old * 4, std::max<size_t>(4, (size_t)(n_post_old + n_post_new) * 4)));
    if (n_new) ORAMA_HIP_TRY(hipMemcpy(p->d_docs.as<uint64_t>() + n_old, docs, (size_t)n_new * 8, hipMemcpyHostToDevice));
    if (n_post_new) {
        ORAMA_HIP_TRY(hipMemcpy(p->d_post_doc.as<uint32_t>() + n_post_old, pd.data(), (size_t)n_post_new * 4, hipMemcpyHostToDevice));
        ORAMA_HIP_TRY(hipMemcpy(p->d_post_val.as<uint32_t>() + n_post_old, pv.data(), (size_t)n_post_new * 4, hipMemcpyHostToDevice));
    }
    if (p->has_omc && n_new) {  // new documents carry no multiplier: x * 1.0 == x
        ORAMA_TRY(grow(p->d_omc, (size_t)n_old * 4, (size_t)n_all * 4));
        std::vector<float> ones((size_t)n_new, 1.0f);
        ORAMA_HIP_TRY(hipMemcpy(p->d_omc.as<float>() + n_old, ones.data(), (size_t)n_new * 4, hipMemcpyHostToDevice));
    }
    if (p->list_off.empty()) p->list_off.assign(1, 0);
    for (uint32_t l = 0; l < n_lists_new; ++l) {
        p->field_of_list.push_back(field_of_list[l]);
        p->list_off.push_back(n_post_old + list_off[l + 1]);
    }
    p->n_lists += n_lists_new;
    p->n_postings = n_post_old + n_post_new;
    p->n_docs = n_all;
    p->dense = dense_after;
    if (dense_after) {
        p->dense_base = base_new;
        p->h_docs.clear();
    } else {
        p->h_docs.swap(all_docs);
    }
    p->avg_len.assign(avg_field_len, avg_field_len + p->n_fields);
    // the averages moved with the new documents: every stored quotient is stale (one streaming pass over the postings,
    // beside the device-to-device copies this call already makes when the arrays grow)
    return refresh_post_ntf(p);
}

int orama_post_fill_synthetic(orama_post* p, uint64_t n_docs, uint64_t first_doc_id, uint32_t n_lists,
                              const uint32_t* ranks, uint64_t seed, uint64_t* out_total_postings) {
    ORAMA_REQUIRE(p, "null handle");
    ORAMA_REQUIRE(n_docs >= 1 && n_docs < 0xffffffffull, "n_docs out of range");
    ORAMA_REQUIRE(n_lists >= 1 && ranks, "null ranks");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::unique_lock<std::shared_mutex> lk(p->mu);
    ScratchLease sc(p->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    // document lengths (device) → average field length (host, f64 mean → f32 like StringStorage::info())
    ORAMA_TRY(sc->misc1.reserve((size_t)n_docs * 2));
    ORAMA_TRY(launch_synth_doc_len(sc->misc1.as<uint16_t>(), n_docs, seed ^ 0xB25ull, s));
    std::vector<uint16_t> h_len((size_t)n_docs);
    ORAMA_HIP_TRY(hipMemcpyAsync(h_len.data(), sc->misc1.p, (size_t)n_docs * 2, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    double sum = 0.0;
    for (uint16_t l : h_len) sum += l;
    const double avg = sum / (double)n_docs;
    // Zipf(1.07) over V = 2^20 ranks: term frequency f_r = r^-1.07 / H; df_r = N * (1 - exp(-avg_len * f_r))
    double H = 0.0;
    for (uint32_t r = 1; r <= (1u << 20); ++r) H += std::pow((double)r, -1.07);
    std::vector<uint64_t> off((size_t)n_lists + 1, 0);
    for (uint32_t l = 0; l < n_lists; ++l) {
        ORAMA_REQUIRE(ranks[l] >= 1, "rank must be >= 1");
        const double f = std::pow((double)ranks[l], -1.07) / H;
        uint64_t df = (uint64_t)std::llround((double)n_docs * (1.0 - std::exp(-avg * f)));
        df = std::min<uint64_t>(std::max<uint64_t>(df, 1), n_docs);
        off[l + 1] = off[l] + df;
    }
    const uint64_t total = off[n_lists];
    ORAMA_TRY(p->d_docs.reserve((size_t)n_docs * 8));
    ORAMA_TRY(p->d_post_doc.reserve((size_t)total * 4));
    ORAMA_TRY(p->d_post_val.reserve((size_t)total * 4));
    ORAMA_TRY(sc->misc0.reserve(off.size() * 8));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, s));
    {   // docs[i] = first_doc_id + i
        std::vector<uint64_t> ids((size_t)std::min<uint64_t>(n_docs, 1u << 22));
        for (uint64_t i0 = 0; i0 < n_docs; i0 += ids.size()) {
            const uint64_t cnt = std::min<uint64_t>(ids.size(), n_docs - i0);
            for (uint64_t i = 0; i < cnt; ++i) ids[(size_t)i] = first_doc_id + i0 + i;
            ORAMA_HIP_TRY(hipMemcpy(p->d_docs.as<uint64_t>() + i0, ids.data(), (size_t)cnt * 8, hipMemcpyHostToDevice));
        }
    }
    ORAMA_TRY(launch_synth_postings(p->d_post_doc.as<uint32_t>(), p->d_post_val.as<uint32_t>(),
                                    sc->misc0.as<uint64_t>(), n_lists, n_docs, sc->misc1.as<uint16_t>(), seed, total, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    p->n_docs = n_docs;
    p->n_fields = 1;
    p->n_lists = n_lists;
    p->n_postings = total;
    p->dense = true;
    p->dense_base = first_doc_id;
    p->h_docs.clear();
    p->avg_len.assign(1, (float)avg);
    p->field_of_list.assign(n_lists, 0u);
    p->list_off = off;
    p->has_omc = false;
    p->d_omc.release();
    ++p->generation;
    if (out_total_postings) *out_total_postings = total;
    return refresh_post_ntf(p);
}

int orama_post_get_list(orama_post* p, uint32_t list, uint64_t capacity, uint64_t* out_doc, uint32_t* out_tf,
                        uint32_t* out_len, uint64_t* out_n) {
    ORAMA_REQUIRE(p && out_n, "null argument");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::shared_lock<std::shared_mutex> lk(p->mu);
    ORAMA_REQUIRE(list < p->n_lists, "list %u out of range", list);
    const uint64_t b = p->list_off[list], n = p->list_off[list + 1] - b;
    *out_n = n;
    if (n == 0 || capacity == 0) return ORAMA_OK;
    ORAMA_REQUIRE(capacity >= n && out_doc && out_tf && out_len, "capacity %llu < list length %llu",
                  (unsigned long long)capacity, (unsigned long long)n);
    std::vector<uint32_t> pd((size_t)n), pv((size_t)n);
    ORAMA_HIP_TRY(hipMemcpy(pd.data(), p->d_post_doc.as<uint32_t>() + b, (size_t)n * 4, hipMemcpyDeviceToHost));
    ORAMA_HIP_TRY(hipMemcpy(pv.data(), p->d_post_val.as<uint32_t>() + b, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        out_doc[i] = p->dense ? p->dense_base + pd[(size_t)i] : p->h_docs[pd[(size_t)i]];
        out_tf[i] = pv[(size_t)i] >> 16;
        out_len[i] = pv[(size_t)i] & 0xffffu;
    }
    return ORAMA_OK;
}

int orama_post_info(orama_post* p, uint64_t* n_docs, uint32_t* n_lists, uint64_t* n_postings, float* avg_len0) {
    ORAMA_REQUIRE(p, "null handle");
    std::shared_lock<std::shared_mutex> lk(p->mu);
    if (n_docs) *n_docs = p->n_docs;
    if (n_lists) *n_lists = p->n_lists;
    if (n_postings) *n_postings = p->n_postings;
    if (avg_len0) *avg_len0 = p->avg_len.empty() ? 0.0f : p->avg_len[0];
    return ORAMA_OK;
}

int orama_post_set_omc(orama_post* p, const uint64_t* omc_doc, const float* omc_mul, uint64_t n) {
    ORAMA_REQUIRE(p, "null handle");
    ORAMA_REQUIRE(n == 0 || (omc_doc && omc_mul), "null argument");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::unique_lock<std::shared_mutex> lk(p->mu);
    ++p->mutations;
    if (n == 0) {
        p->has_omc = false;
        return ORAMA_OK;
    }
    std::vector<float> dense((size_t)p->n_docs, 1.0f);  // x * 1.0 == x bit-for-bit: absent docs untouched
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t local;
        if (!p->local_of(omc_doc[i], &local)) continue;  // multiplier of a doc not in this index
        dense[local] = omc_mul[i];
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

// Many independent full-text queries from ONE caller.  Queries the range-partitioned scorer takes (K3r, bm25_ranges.hip:
// plain top-k over sorted resident lists — the normal case) are scored 32 at a time by ONE set of launches; the others
// are pulled off a shared counter by a handful of worker threads that run the per-query path, each on its own stream +
// scratch set, so that their launch-bound kernels overlap on the device.
// Queries of a batch are independent: a query that fails (malformed, outside the envelope, invalidated by a rebuild
// between validation and dispatch) gets its own status and out_n = 0; every other query is still answered.
static int post_search_batch_impl(orama_post* p, const orama_post_query_desc* queries, uint32_t n_queries, float b,
                                  const uint64_t* allow_bitmap, uint64_t bitmap_bits, int apply_omc, uint32_t max_parallel,
                                  uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                                  uint64_t* out_count, int* out_status) {
    ORAMA_REQUIRE(p && (n_queries == 0 || (queries && out_n)), "null argument");
    if (n_queries == 0) return ORAMA_OK;
    ORAMA_REQUIRE(stride_k == 0 || (out_ids && out_scores), "null output");
    std::vector<int> status(n_queries, ORAMA_OK);
    std::vector<std::string> errors(n_queries);
    auto fail = [&](uint32_t i, int st) {
        status[i] = st;
        errors[i] = orama_last_error();  // the calling thread's error slot
        out_n[i] = 0;
        if (out_count) out_count[i] = 0;
    };
    // queries that fit the range-partitioned scorer (K3r) are scored together by one set of launches per 32 queries;
    // the rest (or all of them when K3r is switched off) run the per-query path on worker threads
    std::vector<uint32_t> rest;
    {
        std::vector<RangeJob> jobs;
        std::vector<uint64_t> counts;
        std::vector<uint32_t> owner;
        // ONE shared lock from the eligibility test to the end of the range scorer: a rebuild or an append in between
        // could otherwise invalidate what ranges_eligible checked (list count, postings < 2^31 — the LDS tables of
        // range_score_kernel are sized by it)
        std::shared_lock<std::shared_mutex> lk(p->mu);
        for (uint32_t i = 0; i < n_queries; ++i) {
            out_n[i] = 0;
            if (out_count) out_count[i] = 0;
            if (queries[i].params.top_k > stride_k) {
                set_error("query %u: top_k %u exceeds the output stride %u", i, queries[i].params.top_k, stride_k);
                fail(i, ORAMA_ERR_INVALID);
                continue;
            }
            if (!ranges_eligible(p, queries[i].refs, queries[i].n_refs, &queries[i].params)) {
                rest.push_back(i);
                continue;
            }
            const int st = check_params(&queries[i].params);
            if (st != ORAMA_OK) fail(i, st);
            else owner.push_back(i);
        }
        if (!owner.empty()) {
            counts.assign(owner.size(), 0);
            for (size_t j = 0; j < owner.size(); ++j) {
                const uint32_t i = owner[j];
                jobs.push_back(RangeJob{queries[i].refs, queries[i].n_refs, &queries[i].params,
                                        out_ids ? out_ids + (size_t)i * stride_k : nullptr,
                                        out_scores ? out_scores + (size_t)i * stride_k : nullptr, &out_n[i], &counts[j]});
            }
            auto run = [&]() -> int {
                ORAMA_ON_DEVICE(p->ctx->device);
                ScratchLease sc(p->ctx), sc2(p->ctx);
                const bool two = jobs.size() > kRangeBatchMax;  // more than one set of launches: double-buffered
                if (two) ORAMA_TRY(ScratchLease::init_pair(sc, sc2));
                else ORAMA_TRY(sc.init());
                return post_search_ranges(p, sc.s.get(), jobs.data(), (uint32_t)jobs.size(), b, allow_bitmap, bitmap_bits, apply_omc,
                                          two ? sc2.s.get() : nullptr);
            };
            const int run_st = run();
            if (run_st == ORAMA_OK) {
                if (out_count)
                    for (size_t j = 0; j < owner.size(); ++j) out_count[owner[j]] = counts[j];
            } else if (run_st == ORAMA_ERR_INVALID || run_st == ORAMA_ERR_UNSUPPORTED) {
                // one query can be responsible for that: every query of the set is answered (or refused) on its own below
                for (uint32_t i : owner) {
                    out_n[i] = 0;
                    rest.push_back(i);
                }
            } else {
                // systemic (scratch pool saturated — BUSY after the acquire timeout —, the device, memory): the per-query
                // path would fail the same way, one timeout after the other; every query of the set gets the status now
                for (uint32_t i : owner) fail(i, run_st);
            }
        }
    }
    if (!rest.empty()) {
        const uint32_t n_rest = (uint32_t)rest.size();
        const uint32_t workers = std::max(1u, std::min(std::min(max_parallel ? max_parallel : 8u, n_rest), 64u));
        std::atomic<uint32_t> next{0};
        std::mutex err_mu;
        auto work = [&]() {
            for (;;) {
                const uint32_t r = next.fetch_add(1);
                if (r >= n_rest) return;
                const uint32_t i = rest[r];
                uint64_t cnt = 0;
                const int st = post_search_impl(p, queries[i].refs, queries[i].n_refs, b, &queries[i].params, allow_bitmap,
                                                bitmap_bits, nullptr, nullptr, 0, false, apply_omc,
                                                out_ids ? out_ids + (size_t)i * stride_k : nullptr,
                                                out_scores ? out_scores + (size_t)i * stride_k : nullptr, &out_n[i], &cnt);
                if (st == ORAMA_OK) {
                    if (out_count) out_count[i] = cnt;
                } else {
                    std::lock_guard<std::mutex> g(err_mu);
                    fail(i, st);  // this worker's error slot
                }
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t w = 1; w < workers; ++w) pool.emplace_back(work);
        work();  // the caller is worker 0
        for (auto& t : pool) t.join();
    }
    int first = ORAMA_OK;
    for (uint32_t i = 0; i < n_queries; ++i) {
        if (out_status) out_status[i] = status[i];
        if (first == ORAMA_OK && status[i] != ORAMA_OK) {
            first = status[i];
            set_error("query %u: %s", i, errors[i].c_str());
        }
    }
    return first;
}

int orama_post_search_batch(orama_post* p, const orama_post_query_desc* queries, uint32_t n_queries, float b,
                            const uint64_t* allow_bitmap, uint64_t bitmap_bits, int apply_omc, uint32_t max_parallel,
                            uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    return post_search_batch_impl(p, queries, n_queries, b, allow_bitmap, bitmap_bits, apply_omc, max_parallel, stride_k,
                                  out_ids, out_scores, out_n, out_count, nullptr);
}

int orama_post_search_batch_status(orama_post* p, const orama_post_query_desc* queries, uint32_t n_queries, float b,
                                   const uint64_t* allow_bitmap, uint64_t bitmap_bits, int apply_omc, uint32_t max_parallel,
                                   uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                                   uint64_t* out_count, int* out_status) {
    ORAMA_REQUIRE(n_queries == 0 || out_status, "null status array");
    return post_search_batch_impl(p, queries, n_queries, b, allow_bitmap, bitmap_bits, apply_omc, max_parallel, stride_k,
                                  out_ids, out_scores, out_n, out_count, out_status);
}

int orama_post_search_hybrid(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                             const orama_bm25_params* params, const uint64_t* allow_bitmap,
                             uint64_t bitmap_bits, const uint64_t* vec_doc, const float* vec_score,
                             uint32_t n_vec, int apply_omc, uint64_t* out_ids, float* out_scores,
                             uint32_t* out_n, uint64_t* out_count) {
    return post_search_impl(p, refs, n_refs, b, params, allow_bitmap, bitmap_bits, vec_doc, vec_score, n_vec, true,
                            apply_omc, out_ids, out_scores, out_n, out_count);
}

// ================================================================= score-map handle, facets, groups (§8f rank 4)
// The reference keeps the whole HashMap<DocumentId, f32> of a search around for facets and groups
// (src/collection_manager/sides/read/search.rs:355-400).  Here the map is the candidate list + position index the
// scorer left in its scratch set; an orama_scores handle keeps that scratch set leased (and the store read-locked)
// until it is destroyed, so facet / group passes run over it in HBM and nothing of size `count` crosses PCIe.
struct orama_scores {
    orama_post* p = nullptr;
    std::shared_lock<std::shared_mutex> lk;
    ScratchLease lease;
    PostQuery st;
    uint64_t generation = 0;
    uint64_t count = 0;
    uint32_t list_len = 0;
    std::mutex mu;  // facet / group / export calls on one handle are serialised (they share its stream)
    DevBuf tmp_a, tmp_b, tmp_c;
    explicit orama_scores(orama_post* post) : p(post), lk(post->mu), lease(post->ctx, kScratchRecords) {}
    ScoreMapDev dev() const {
        ScoreMapDev m;
        m.emit = st.qb.emit;
        m.cand_score = st.qb.cand_score;
        m.cand_idx = st.qb.cand_idx;
        m.docs = p->d_docs.as<uint64_t>();
        m.dense_base = p->dense_base;
        m.epoch = st.qb.epoch;
        return m;
    }
};

struct orama_facet_field {
    orama_post* p = nullptr;  // identity only (compared, never dereferenced after creation: the index may be gone)
    int device = 0;
    uint64_t generation = 0;
    bool numbers = false;
    uint32_t n_buckets = 0;
    uint64_t n_entries = 0;
    DevBuf entry_doc, entry_val, bucket_off;
};

namespace {
int check_field(orama_scores* sm, orama_facet_field* f, bool numbers) {
    ORAMA_REQUIRE(sm && f, "null argument");
    ORAMA_REQUIRE(f->p == sm->p, "facet field and score map belong to different indexes");
    ORAMA_REQUIRE(f->generation == sm->generation, "facet field is stale: the index was rebuilt since it was created");
    ORAMA_REQUIRE(f->numbers == numbers, numbers ? "this call needs a number field" : "this call needs a bucket field");
    return ORAMA_OK;
}
}  // namespace

extern "C" {

int orama_post_search_scores(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                             const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                             int hybrid, const uint64_t* vec_doc, const float* vec_score, uint32_t n_vec, int apply_omc,
                             uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count,
                             orama_scores** out_map) {
    ORAMA_REQUIRE(p && out_n && params && out_map, "null argument");
    *out_map = nullptr;
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_REQUIRE(!hybrid || n_vec == 0 || (vec_doc && vec_score), "null vector map");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::unique_ptr<orama_scores> h(new (std::nothrow) orama_scores(p));
    if (!h) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    ORAMA_TRY(h->lease.init());
    Scratch* sc = h->lease.s.get();
    uint64_t count = 0;
    // The full-text map on the range scorer (K3r): the candidate list + position index come out of the scoring launch itself
    // — 8 bytes per document of the index (the epoch-stamped position table) + 16 per referenced posting, instead of the
    // per-record scorer's 136 bytes per document (1.4 GB at 10 M documents).  Hybrid maps (every score is rewritten by
    // normalize_and_combine, vector-only documents join) and queries the range scorer does not take stay on K3.
    uint64_t total_postings = 0;
    for (uint32_t i = 0; i < n_refs && refs; ++i)
        if (refs[i].list < p->n_lists) total_postings += p->list_off[refs[i].list + 1] - p->list_off[refs[i].list];
    if (!hybrid && total_postings > 0 && ranges_eligible(p, refs, n_refs, params)) {
        ORAMA_TRY(check_params(params));
        RangeJob job{refs, n_refs, params, out_ids, out_scores, out_n, &count};
        job.map = &h->st.qb;
        job.map_list_len = &h->list_len;
        ORAMA_TRY(post_search_ranges(p, sc, &job, 1, b, allow_bitmap, bitmap_bits, apply_omc));
    } else {
        ORAMA_TRY(post_stage1(p, sc, refs, n_refs, b, params, allow_bitmap, bitmap_bits, hybrid != 0, apply_omc, hybrid ? n_vec : 0,
                              &h->st));
        ORAMA_TRY(post_stage2(p, sc, h->st, params, vec_doc, vec_score, hybrid ? n_vec : 0, out_ids, out_scores, out_n, &count));
        ORAMA_HIP_TRY(hipMemcpy(&h->list_len, &h->st.qb.state->list_len, 4, hipMemcpyDeviceToHost));
    }
    h->count = count;
    h->generation = p->generation;
    if (out_count) *out_count = count;
    h->lease.detach();  // the handle keeps the set until orama_scores_destroy: it no longer counts as an in-flight call
    *out_map = h.release();
    return ORAMA_OK;
}

void orama_scores_destroy(orama_scores* sm) {
    if (!sm) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(sm->p->ctx->device);
    if (sm->lease.s) (void)hipStreamSynchronize(sm->lease.s->stream);
    delete sm;
}

int orama_scores_count(orama_scores* sm, uint64_t* out) {
    ORAMA_REQUIRE(sm && out, "null argument");
    *out = sm->count;
    return ORAMA_OK;
}

int orama_scores_export(orama_scores* sm, uint64_t capacity, uint64_t* out_ids, float* out_scores, uint64_t* out_n) {
    ORAMA_REQUIRE(sm && out_n, "null argument");
    *out_n = sm->count;
    if (capacity == 0 || sm->count == 0) return ORAMA_OK;
    ORAMA_REQUIRE(capacity >= sm->count && out_ids && out_scores, "capacity %llu < map size %llu",
                  (unsigned long long)capacity, (unsigned long long)sm->count);
    ORAMA_ON_DEVICE(sm->p->ctx->device);
    std::lock_guard<std::mutex> g(sm->mu);
    hipStream_t s = sm->lease.s->stream;
    ORAMA_TRY(sm->tmp_a.reserve((size_t)sm->count * 8));
    ORAMA_TRY(sm->tmp_b.reserve((size_t)sm->count * 4));
    ORAMA_TRY(sm->tmp_c.reserve(16));
    ORAMA_TRY(launch_scores_export(sm->dev(), sm->list_len, sm->tmp_a.as<uint64_t>(), sm->tmp_b.as<float>(),
                                   sm->tmp_c.as<uint32_t>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_ids, sm->tmp_a.p, (size_t)sm->count * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_scores, sm->tmp_b.p, (size_t)sm->count * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    return ORAMA_OK;
}

int orama_scores_lookup(orama_scores* sm, const uint64_t* doc_ids, uint32_t n, float* out_scores, uint8_t* out_present) {
    ORAMA_REQUIRE(sm && (n == 0 || (doc_ids && out_scores && out_present)), "null argument");
    if (n == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(sm->p->ctx->device);
    std::lock_guard<std::mutex> g(sm->mu);
    hipStream_t s = sm->lease.s->stream;
    std::vector<uint32_t> local(n);
    for (uint32_t i = 0; i < n; ++i)
        if (!sm->p->local_of(doc_ids[i], &local[i])) local[i] = 0xffffffffu;
    ORAMA_TRY(sm->tmp_a.reserve((size_t)n * 4));
    ORAMA_TRY(sm->tmp_b.reserve((size_t)n * 4));
    ORAMA_TRY(sm->tmp_c.reserve((size_t)n + 16));
    ORAMA_HIP_TRY(hipMemcpyAsync(sm->tmp_a.p, local.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    ORAMA_TRY(launch_scores_lookup(sm->dev(), sm->tmp_a.as<uint32_t>(), n, sm->tmp_b.as<float>(), sm->tmp_c.as<uint8_t>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_scores, sm->tmp_b.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_present, sm->tmp_c.p, (size_t)n, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    return ORAMA_OK;
}

int orama_facet_field_create_buckets(orama_post* index, const uint64_t* bucket_off, const uint64_t* bucket_docs,
                                     uint32_t n_buckets, orama_facet_field** out) {
    ORAMA_REQUIRE(index && out && (n_buckets == 0 || bucket_off), "null argument");
    *out = nullptr;
    const uint64_t n = n_buckets ? bucket_off[n_buckets] : 0;
    ORAMA_REQUIRE(n == 0 || bucket_docs, "null bucket docs");
    for (uint32_t b = 0; b < n_buckets; ++b)
        ORAMA_REQUIRE(bucket_off[b] <= bucket_off[b + 1], "bucket offsets must be non-decreasing");
    ORAMA_ON_DEVICE(index->ctx->device);
    std::shared_lock<std::shared_mutex> lk(index->mu);
    std::unique_ptr<orama_facet_field> f(new (std::nothrow) orama_facet_field());
    if (!f) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    f->p = index;
    f->device = index->ctx->device;
    f->generation = index->generation;
    f->numbers = false;
    f->n_buckets = n_buckets;
    f->n_entries = n;
    // DocumentId -> local doc index once, here; ids the index does not hold can never be keys of a score map
    std::vector<uint32_t> local((size_t)n);
    for (uint64_t i = 0; i < n; ++i)
        if (!index->local_of(bucket_docs[i], &local[(size_t)i])) local[(size_t)i] = 0xffffffffu;
    ORAMA_TRY(f->entry_doc.reserve(std::max<size_t>(4, (size_t)n * 4)));
    ORAMA_TRY(f->bucket_off.reserve((size_t)(n_buckets + 1) * 8));
    if (n) ORAMA_HIP_TRY(hipMemcpy(f->entry_doc.p, local.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    const uint64_t zero = 0;
    ORAMA_HIP_TRY(hipMemcpy(f->bucket_off.p, n_buckets ? bucket_off : &zero, (size_t)(n_buckets + 1) * 8, hipMemcpyHostToDevice));
    *out = f.release();
    return ORAMA_OK;
}

int orama_facet_field_create_numbers(orama_post* index, const uint64_t* docs, const double* values, uint64_t n,
                                     orama_facet_field** out) {
    ORAMA_REQUIRE(index && out && (n == 0 || (docs && values)), "null argument");
    *out = nullptr;
    ORAMA_ON_DEVICE(index->ctx->device);
    std::shared_lock<std::shared_mutex> lk(index->mu);
    std::unique_ptr<orama_facet_field> f(new (std::nothrow) orama_facet_field());
    if (!f) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    f->p = index;
    f->device = index->ctx->device;
    f->generation = index->generation;
    f->numbers = true;
    f->n_entries = n;
    std::vector<uint32_t> local((size_t)n);
    for (uint64_t i = 0; i < n; ++i)
        if (!index->local_of(docs[i], &local[(size_t)i])) local[(size_t)i] = 0xffffffffu;
    ORAMA_TRY(f->entry_doc.reserve(std::max<size_t>(4, (size_t)n * 4)));
    ORAMA_TRY(f->entry_val.reserve(std::max<size_t>(8, (size_t)n * 8)));
    if (n) {
        ORAMA_HIP_TRY(hipMemcpy(f->entry_doc.p, local.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        ORAMA_HIP_TRY(hipMemcpy(f->entry_val.p, values, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    *out = f.release();
    return ORAMA_OK;
}

void orama_facet_field_destroy(orama_facet_field* f) {
    if (!f) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(f->device);
    (void)hipDeviceSynchronize();
    delete f;
}

int orama_facet_count(orama_scores* sm, orama_facet_field* f, uint64_t* out_counts) {
    ORAMA_TRY(check_field(sm, f, false));
    if (f->n_buckets == 0) return ORAMA_OK;
    ORAMA_REQUIRE(out_counts, "null output");
    ORAMA_ON_DEVICE(sm->p->ctx->device);
    std::lock_guard<std::mutex> g(sm->mu);
    hipStream_t s = sm->lease.s->stream;
    ORAMA_TRY(sm->tmp_a.reserve((size_t)f->n_buckets * 8));
    ORAMA_TRY(launch_facet_count_buckets(sm->dev(), f->entry_doc.as<uint32_t>(), f->bucket_off.as<uint64_t>(), f->n_buckets,
                                         f->n_entries, sm->tmp_a.as<unsigned long long>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_counts, sm->tmp_a.p, (size_t)f->n_buckets * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    return ORAMA_OK;
}

int orama_facet_count_ranges(orama_scores* sm, orama_facet_field* f, const double* from, const double* to,
                             uint32_t n_ranges, uint64_t* out_counts) {
    ORAMA_TRY(check_field(sm, f, true));
    if (n_ranges == 0) return ORAMA_OK;
    ORAMA_REQUIRE(from && to && out_counts, "null argument");
    ORAMA_ON_DEVICE(sm->p->ctx->device);
    std::lock_guard<std::mutex> g(sm->mu);
    hipStream_t s = sm->lease.s->stream;
    // ranges in chunks of 64 (the kernel keeps them in LDS)
    for (uint32_t r0 = 0; r0 < n_ranges; r0 += 64) {
        const uint32_t nr = std::min<uint32_t>(64, n_ranges - r0);
        ORAMA_TRY(sm->tmp_a.reserve(64 * 8));
        ORAMA_TRY(sm->tmp_b.reserve(2 * 64 * 8));
        ORAMA_HIP_TRY(hipMemcpyAsync(sm->tmp_b.p, from + r0, (size_t)nr * 8, hipMemcpyHostToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(sm->tmp_b.as<double>() + 64, to + r0, (size_t)nr * 8, hipMemcpyHostToDevice, s));
        ORAMA_TRY(launch_facet_count_ranges(sm->dev(), f->entry_doc.as<uint32_t>(), f->entry_val.as<double>(), f->n_entries,
                                            sm->tmp_b.as<double>(), sm->tmp_b.as<double>() + 64, nr,
                                            sm->tmp_a.as<unsigned long long>(), s));
        ORAMA_HIP_TRY(hipMemcpyAsync(out_counts + r0, sm->tmp_a.p, (size_t)nr * 8, hipMemcpyDeviceToHost, s));
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
    }
    return ORAMA_OK;
}

int orama_group_top(orama_scores* sm, orama_facet_field* f, uint32_t max_results, uint64_t* out_ids, float* out_scores,
                    uint32_t* out_n) {
    ORAMA_TRY(check_field(sm, f, false));
    if (f->n_buckets == 0) return ORAMA_OK;
    ORAMA_REQUIRE(out_ids && out_scores && out_n, "null output");
    ORAMA_REQUIRE(max_results >= 1, "max_results is 0");
    ORAMA_SUPPORT(max_results <= kGroupMaxK, "max_results %u outside [1, %u]", max_results, kGroupMaxK);
    ORAMA_ON_DEVICE(sm->p->ctx->device);
    std::lock_guard<std::mutex> g(sm->mu);
    hipStream_t s = sm->lease.s->stream;
    const size_t nk = (size_t)f->n_buckets * max_results;
    ORAMA_TRY(sm->tmp_a.reserve(nk * 8));
    ORAMA_TRY(sm->tmp_b.reserve(nk * 4));
    ORAMA_TRY(sm->tmp_c.reserve((size_t)f->n_buckets * 4));
    ORAMA_TRY(launch_group_top(sm->dev(), f->entry_doc.as<uint32_t>(), f->bucket_off.as<uint64_t>(), f->n_buckets, max_results,
                               sm->tmp_a.as<uint64_t>(), sm->tmp_b.as<float>(), sm->tmp_c.as<uint32_t>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_ids, sm->tmp_a.p, nk * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_scores, sm->tmp_b.p, nk * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_n, sm->tmp_c.p, (size_t)f->n_buckets * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    return ORAMA_OK;
}

}  // extern "C"

// ================================================================= staged query: one index sharded over GPUs
// SURVEY §8e.  Every stage enqueues on the caller's stream; the caller runs the collectives between them.
struct orama_post_query {
    orama_post* p = nullptr;
    std::shared_lock<std::shared_mutex> lk;
    ScratchLease lease;
    hipStream_t own_stream = nullptr;  // the lease's stream, put back when the query ends
    PostQuery st;
    orama_bm25_params params{};
    int stage = 0;  // 1 accumulated, 2 scored, 3 finished
    explicit orama_post_query(orama_post* post) : p(post), lk(post->mu), lease(post->ctx, kScratchRecords) {}
    ~orama_post_query() {
        if (lease.s && own_stream) lease.s->stream = own_stream;
    }
};

uint64_t orama_post_block_bytes(uint32_t top_k) { return packed_block_bytes(1, top_k) + 8; }

int orama_post_query_begin(orama_post* p, const orama_term_ref* refs, uint32_t n_refs, float b,
                           const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                           int hybrid, int apply_omc, uint32_t n_vec_cap, void* hip_stream, int32_t* d_df,
                           orama_post_query** out) {
    ORAMA_REQUIRE(p && params && d_df && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(params->top_k >= 1, "staged query: top_k must be >= 1");
    ORAMA_ON_DEVICE(p->ctx->device);
    std::unique_ptr<orama_post_query> q(new (std::nothrow) orama_post_query(p));
    if (!q) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    ORAMA_TRY(q->lease.init());
    Scratch* sc = q->lease.s.get();
    q->own_stream = sc->stream;
    sc->stream = static_cast<hipStream_t>(hip_stream);
    q->params = *params;
    ORAMA_TRY(post_stage1(p, sc, refs, n_refs, b, params, allow_bitmap, bitmap_bits, hybrid != 0, apply_omc, n_vec_cap,
                          &q->st, false));
    ORAMA_TRY(launch_df_export(q->st.qb.state, params->n_tokens, d_df, sc->stream));
    q->stage = 1;
    *out = q.release();
    return ORAMA_OK;
}

int orama_post_query_score(orama_post_query* q, const uint32_t* df_global, int64_t* d_minmax) {
    ORAMA_REQUIRE(q && df_global, "null argument");
    ORAMA_REQUIRE(q->stage == 1, "staged query: score called out of order");
    ORAMA_REQUIRE(!q->st.hybrid || d_minmax, "staged query: hybrid needs the min/max exchange buffer");
    orama_post* p = q->p;
    ORAMA_ON_DEVICE(p->ctx->device);
    Scratch* sc = q->lease.s.get();
    hipStream_t s = sc->stream;
    // idf by the host libm from the GLOBAL df and N (calculate_idf, bm25.rs:78-82; df.max(1), token_score.rs:275)
    float* h_idf = reinterpret_cast<float*>(sc->h_in.as<char>() + q->st.idf_stage_off);
    for (uint32_t t = 0; t < q->params.n_tokens; ++t) {
        const float d = (float)(df_global[t] < 1 ? 1u : df_global[t]);
        h_idf[t] = log1pf((q->params.total_documents - d + 0.5f) / (d + 0.5f));
    }
    float* d_idf = reinterpret_cast<float*>(sc->misc0.as<char>() + q->st.idf_dev_off);
    ORAMA_HIP_TRY(hipMemcpyAsync(d_idf, h_idf, (size_t)q->params.n_tokens * 4, hipMemcpyHostToDevice, s));
    ORAMA_TRY(post_finalize(p, sc, q->st, &q->params, d_idf));
    if (q->st.hybrid) ORAMA_TRY(launch_minmax_export(q->st.qb.state, reinterpret_cast<long long*>(d_minmax), s));
    q->stage = 2;
    return ORAMA_OK;
}

int orama_post_query_finish(orama_post_query* q, const int64_t* d_minmax_global, const uint64_t* vec_doc,
                            const float* vec_score, uint32_t n_vec, void* d_block) {
    ORAMA_REQUIRE(q && d_block, "null argument");
    ORAMA_REQUIRE(q->stage == 2, "staged query: finish called out of order");
    orama_post* p = q->p;
    ORAMA_ON_DEVICE(p->ctx->device);
    Scratch* sc = q->lease.s.get();
    hipStream_t s = sc->stream;
    if (q->st.hybrid) {
        ORAMA_REQUIRE(d_minmax_global, "staged query: hybrid needs the reduced min/max");
        ORAMA_REQUIRE(n_vec == 0 || (vec_doc && vec_score), "null vector map");
        ORAMA_TRY(launch_minmax_import(q->st.qb.state, reinterpret_cast<const long long*>(d_minmax_global), s));
        ORAMA_TRY(post_combine(p, sc, q->st, vec_doc, vec_score, n_vec, true));
    }
    const uint32_t k = q->params.top_k;
    char* blk = static_cast<char*>(d_block);
    ORAMA_TRY(sc->out_n.reserve(4));
    ORAMA_TRY(select_enqueue(p->ctx, sc, q->st.qb, p->d_docs.as<uint64_t>(), (uint32_t)q->st.cand_cap, k,
                             reinterpret_cast<uint64_t*>(blk), reinterpret_cast<float*>(blk + (size_t)k * 8),
                             sc->out_n.as<uint32_t>()));
    ORAMA_TRY(launch_count_export(q->st.qb.state, reinterpret_cast<unsigned long long*>(blk + packed_block_bytes(1, k)),
                                  s));
    q->stage = 3;
    return ORAMA_OK;
}

void orama_post_query_end(orama_post_query* q) {
    if (!q) return;
    // the scratch set goes back to the pool: everything enqueued on the caller's stream must have drained
    if (q->lease.s) (void)hipStreamSynchronize(q->lease.s->stream);
    delete q;
}

int orama_post_merge_blocks_device(orama_ctx* ctx, const void* d_blocks, uint32_t lists, uint32_t top_k,
                                   uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_n,
                                   uint64_t* d_out_count, void* hip_stream) {
    ORAMA_REQUIRE(ctx && d_blocks && d_out_ids && d_out_scores && d_out_n && d_out_count, "null argument");
    ORAMA_ON_DEVICE(ctx->device);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const uint64_t stride = orama_post_block_bytes(top_k);
    ORAMA_TRY(launch_merge_blocks(ctx, d_blocks, stride, lists, 1, top_k, true, d_out_ids, d_out_scores, d_out_n, s));
    return launch_count_sum(d_blocks, stride, packed_block_bytes(1, top_k), lists,
                            reinterpret_cast<unsigned long long*>(d_out_count), s);
}

int orama_post_set_avg_len(orama_post* p, const float* avg_field_len, uint32_t n_fields) {
    ORAMA_REQUIRE(p && avg_field_len, "null argument");
    std::unique_lock<std::shared_mutex> lk(p->mu);
    ORAMA_REQUIRE(n_fields == p->n_fields, "expected %u field averages, got %u", p->n_fields, n_fields);
    ORAMA_ON_DEVICE(p->ctx->device);
    p->avg_len.assign(avg_field_len, avg_field_len + n_fields);
    return refresh_post_ntf(p);
}

// search_hybrid (token_score.rs:357-387) as ONE call: the vector leg (K1/K2 + K4 on its own HIP stream) and
// the full-text leg (K3 on a second stream) run concurrently; the host applies the in-tree epilogue
// (embedding_field.rs:268-276) to the <= limit vector hits and stage 2 combines on the device.
int orama_hybrid_search(orama_vec* v, orama_post* p, const float* query, uint32_t limit, float min_similarity,
                        int rescale_e5, const orama_term_ref* refs, uint32_t n_refs, float b,
                        const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                        int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    ORAMA_REQUIRE(v && p && query && out_n && params, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_SUPPORT(limit <= kSelectMaxK, "limit %u exceeds the supported maximum %u", limit, kSelectMaxK);
    orama_ctx* ctx = p->ctx;
    ORAMA_REQUIRE(vec_ctx(v) == ctx, "vector store and postings store live on different contexts");
    ORAMA_ON_DEVICE(ctx->device);
    VecSharedLock vlk(v);
    std::shared_lock<std::shared_mutex> lk(p->mu);
    // a store with an fp16 shadow answers the vector leg with the two-stage plan (same answer, half the bytes scanned): it
    // needs a second vector set; its blocking half (VecTwoStage::finish) runs when the leg is joined
    const bool two_stage = vec_rows(v) > 0 && limit > 0 && vec_two_stage_usable(v, query, 1, limit);
    const uint32_t dim = vec_dim(v);
    const bool have_rows = vec_rows(v) > 0 && limit > 0;
    const uint32_t kk = limit ? limit : 1;
    bool two_stage_device = false;  // the vector leg of a shadow store in the plan's device form (set with the device tail below)
    bool hits_stay_on_device = false;  // the device tail reads the vector hits where the selection wrote them: no direct answers
    // ---- leg A: vector scan + top-`limit` rows on a's stream; the read-back lands in a->h_out
    auto vector_leg = [&](ScratchLease& a, ScratchLease& a2, VecTwoStage& ts) -> int {
        if (!have_rows) return ORAMA_OK;
        hipStream_t sa = a->stream;
        ORAMA_TRY(a->query.reserve((size_t)dim * 4));
        ORAMA_TRY(a->h_in.reserve((size_t)dim * 4));
        memcpy(a->h_in.p, query, (size_t)dim * 4);
        ORAMA_TRY(stage_block(ctx, a->query.p, a->h_in.p, (size_t)dim * 4, hipMemcpyHostToDevice, sa));
        const uint64_t* d_allow = nullptr;
        ORAMA_TRY(resolve_allow(ctx, a.s.get(), allow_bitmap, bitmap_bits, sa, &d_allow));
        // results in one block [ids | distances | n]: one read-back
        ORAMA_TRY(a->out_ids.reserve((size_t)kk * 12 + 8));
        uint64_t* d_ids = a->out_ids.as<uint64_t>();
        float* d_dist = reinterpret_cast<float*>(d_ids + kk);
        uint32_t* d_n = reinterpret_cast<uint32_t*>(d_dist + kk);
        ORAMA_TRY(a->h_out.reserve((size_t)kk * 12 + 8));
        if (two_stage && two_stage_device)  // (the device tail: nothing of the plan is left for the host to decide)
            return ts.begin_device(v, a, a2, a->query.as<float>(), 1, limit, d_allow, bitmap_bits, d_ids, d_dist, d_n);
        if (two_stage) {  // (the read-back behind the plan, before the host looks at the proof word: one wake-up — join_vector_leg)
            ORAMA_TRY(ts.begin(v, a, a2, a->query.as<float>(), 1, limit, d_allow, bitmap_bits, d_ids, d_dist, d_n));
            return stage_block(ctx, a->h_out.p, d_ids, (size_t)kk * 12 + 4, hipMemcpyDeviceToHost, sa);
        }
        if (vec_rows_are_f32(v) && ctx->stage_by_kernel && ctx->direct_out && !hits_stay_on_device) {
            // (fp32 rows: the selection's last launch only writes the answers — straight into the pinned block)
            char* h = a->h_out.as<char>();
            return vec_search_enqueue(v, a.s.get(), a->query.as<float>(), 1, limit, d_allow, bitmap_bits, reinterpret_cast<uint64_t*>(h),
                                      reinterpret_cast<float*>(h + (size_t)kk * 8), reinterpret_cast<uint32_t*>(h + (size_t)kk * 12), sa);
        }
        ORAMA_TRY(vec_search_enqueue(v, a.s.get(), a->query.as<float>(), 1, limit, d_allow, bitmap_bits, d_ids, d_dist, d_n, sa));
        ORAMA_TRY(stage_block(ctx, a->h_out.p, d_ids, (size_t)kk * 12 + 4, hipMemcpyDeviceToHost, sa));
        return ORAMA_OK;
    };
    // ---- join A; in-tree epilogue on the host: similarity, rescale, cut-off, per-doc sum (hit order)
    std::vector<uint64_t> vdoc;
    std::vector<float> vsc;
    auto join_vector_leg = [&](ScratchLease& a, VecTwoStage& ts) -> int {
        if (!have_rows) return ORAMA_OK;
        if (two_stage) {  // (an unproven candidate list is re-answered by the plain scan in here)
            bool reran = false;
            ORAMA_TRY(ts.finish(&reran));
            if (reran || two_stage_device)
                ORAMA_TRY(stage_block(ctx, a->h_out.p, a->out_ids.p, (size_t)kk * 12 + 4, hipMemcpyDeviceToHost, a->stream));
        }
        ORAMA_HIP_TRY(hipStreamSynchronize(a->stream));
        const char* ha = a->h_out.as<char>();
        const uint64_t* ids = reinterpret_cast<const uint64_t*>(ha);
        const float* dist = reinterpret_cast<const float*>(ha + (size_t)kk * 8);
        const uint32_t n = *reinterpret_cast<const uint32_t*>(ha + (size_t)kk * 12);
        for (uint32_t i = 0; i < n; ++i) {
            const float similarity = 1.0f - dist[i];
            float score = similarity;
            if (rescale_e5) {  // Model::rescale_score, src/python/embeddings.rs:71-92
                const float MIN = 0.7f, MAX = 1.0f, DELTA = MAX - MIN;
                float c = similarity;
                if (c < MIN) c = MIN;
                if (c > MAX) c = MAX;
                score = (c - MIN) / DELTA;
            }
            if (!(score >= min_similarity)) continue;
            size_t j = 0;
            for (; j < vdoc.size(); ++j)
                if (vdoc[j] == ids[i]) break;
            if (j == vdoc.size()) {
                vdoc.push_back(ids[i]);
                vsc.push_back(0.0f);
            }
            vsc[j] = vsc[j] + score;
        }
        return ORAMA_OK;
    };

    // ---- the full-text leg on the range scorer (K3r), the hybrid tail on <= top_k + 2 limit + 1 candidates: no OMC
    // (multipliers reorder documents arbitrarily), references that fit the sort key, candidates that fit one selection
    const bool by_ranges = ranges_eligible(p, refs, n_refs, params) && ctx->bm25_ranges_hybrid && !(apply_omc && p->has_omc) &&
                           (uint64_t)params->top_k + limit + 1 <= kSelectMaxK;
    if (by_ranges) {
        ORAMA_TRY(check_params(params));
        bool fallback = false, joined = false;
        {
            ScratchLease a(ctx, kScratchVector), a2(ctx, kScratchVector), g(ctx, kScratchGeneral);
            if (two_stage) ORAMA_TRY(ScratchLease::init_three(a, a2, g));
            else ORAMA_TRY(ScratchLease::init_pair(a, g));
            VecTwoStage ts;
            // the tail on the device (hybrid_tail.hip) where its kernels take the call: hits and merged entries within their
            // envelopes, and a vector leg that needs no host decision (a shadow store: the plan's device form)
            const bool dev_tail = ctx->hybrid_device_tail && have_rows && limit <= kHybridTailMaxVec &&
                                  (uint64_t)params->top_k + 2ull * limit + 1 <= kSelectMaxK &&
                                  (!two_stage || vec_two_stage_device_usable(v, 1, limit));
            two_stage_device = dev_tail && two_stage;
            hits_stay_on_device = dev_tail;
            ORAMA_TRY(vector_leg(a, a2, ts));
            RangeJob job{refs, n_refs, params, out_ids, out_scores, out_n, out_count};
            job.hybrid = true;
            job.fallback = &fallback;
            job.n_vec_max = have_rows ? limit : 0;
            if (dev_tail) {
                if (!a->ev_done) ORAMA_HIP_TRY(hipEventCreateWithFlags(&a->ev_done, hipEventDisableTiming));
                ORAMA_HIP_TRY(hipEventRecord(a->ev_done, a->stream));
                job.device_tail = true;
                job.d_vec_ids = a->out_ids.as<uint64_t>();
                job.d_vec_dist = reinterpret_cast<const float*>(job.d_vec_ids + kk);
                job.d_vec_n = reinterpret_cast<const uint32_t*>(job.d_vec_dist + kk);
                job.vec_ready = a->ev_done;
                job.vec_limit = limit;
                job.min_similarity = min_similarity;
                job.rescale_e5 = rescale_e5;
            }
            job.vec_provider = [&](const uint64_t** doc, const float** score, uint32_t* n) -> int {
                if (!joined) {
                    ORAMA_TRY(join_vector_leg(a, ts));
                    joined = true;
                }
                *doc = vdoc.data();
                *score = vsc.data();
                *n = (uint32_t)vdoc.size();
                return ORAMA_OK;
            };
            const int rc = post_search_ranges(p, g.s.get(), &job, 1, b, allow_bitmap, bitmap_bits, 0);
            if (rc != ORAMA_OK) {
                if (!joined && have_rows) (void)hipStreamSynchronize(a->stream);  // nothing of the call stays in flight
                return rc;
            }
            if (!fallback) {
                // (the device tail never joined the vector leg on the host: its stream has drained — the tail waited for it —
                // and a shadow store's read lock goes back here)
                if (two_stage && !joined) ORAMA_TRY(ts.finish());
                return ORAMA_OK;
            }
            if (!joined) {  // (no full-text side at all: the scorer never asked for the map)
                const uint64_t* d_;
                const float* s_;
                uint32_t n_;
                ORAMA_TRY(job.vec_provider(&d_, &s_, &n_));
            }
        }
        // the candidates could not prove the answer (or a vector hit is not a document of the index): the per-record scorer
        // combines the whole maps.  Every set of the first attempt is back in the pool before this one is drawn.
        *out_n = 0;
        if (out_count) *out_count = 0;
        ScratchLease rsc(ctx, kScratchRecords);
        ORAMA_TRY(rsc.init());
        PostQuery st;
        ORAMA_TRY(post_stage1(p, rsc.s.get(), refs, n_refs, b, params, allow_bitmap, bitmap_bits, true, apply_omc, (uint32_t)vdoc.size(), &st));
        return post_stage2(p, rsc.s.get(), st, params, vdoc.data(), vsc.data(), (uint32_t)vdoc.size(), out_ids, out_scores, out_n, out_count);
    }

    // ---- the full-text leg on the per-record scorer (K3): stage 1 beside the scan, combine + count + top-k after it
    ScratchLease a(ctx, kScratchVector), a2(ctx, kScratchVector), bsc(ctx, kScratchRecords);
    if (two_stage) ORAMA_TRY(ScratchLease::init_three(a, a2, bsc));
    else ORAMA_TRY(ScratchLease::init_pair(a, bsc));
    PostQuery st;
    VecTwoStage ts;
    ORAMA_TRY(vector_leg(a, a2, ts));
    // ---- leg B, stage 1: BM25F accumulate + finalise (overlaps leg A)
    ORAMA_TRY(post_stage1(p, bsc.s.get(), refs, n_refs, b, params, allow_bitmap, bitmap_bits, true, apply_omc, limit, &st));
    ORAMA_TRY(join_vector_leg(a, ts));
    // ---- leg B, stage 2: combine + OMC + count + top-k
    return post_stage2(p, bsc.s.get(), st, params, vdoc.data(), vsc.data(), (uint32_t)vdoc.size(), out_ids, out_scores,
                       out_n, out_count);
}

// ================================================================= seam (i): host-provided contributions
// export_capacity > 0: instead of the top-k, return EVERY (DocumentId, score) entry of the map (get_scores()).
static int bm25_score_impl(orama_ctx* ctx, const orama_ntf_entry* entries, uint32_t n_entries,
                           const orama_bm25_params* params_in, const uint64_t* omc_doc, const float* omc_mul,
                           uint64_t n_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                           uint64_t* out_count, uint64_t export_capacity, uint64_t* out_n64) {
    ORAMA_REQUIRE(ctx && (out_n || out_n64), "null argument");
    if (out_n) *out_n = 0;
    if (out_n64) *out_n64 = 0;
    if (out_count) *out_count = 0;
    ORAMA_TRY(check_params(params_in));
    orama_bm25_params params_copy = *params_in;
    const bool export_all = out_n64 != nullptr;
    if (export_all) params_copy.top_k = 0;
    const orama_bm25_params* params = &params_copy;
    ORAMA_REQUIRE(n_entries == 0 || entries, "null entries");
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_ON_DEVICE(ctx->device);
    // The reference scorer accepts the same (doc, ntf) key several times for one token — add_field pushes onto a
    // Vec per key (bm25.rs:355-366) and finalize sums them in push order — so a per_doc_ntf list may repeat a doc
    // (a prefix / fuzzy expansion inside the third-party store could emit one pair per matched term).  The
    // accumulate kernel gives every (token, doc) cell ONE writer per launch, so an entry whose docs repeat is split
    // here into pieces with unique docs: occurrence number o of a doc goes to piece o, the pieces take the place of
    // the entry in the token's rank order — the additions happen in exactly the reference's order.
    struct Piece {
        uint32_t token;
        const uint64_t* doc;
        const float* ntf;
        uint64_t len;
    };
    std::vector<Piece> pieces;
    std::vector<std::vector<uint64_t>> own_doc;  // storage of split entries (stable: reserved up front)
    std::vector<std::vector<float>> own_ntf;
    pieces.reserve(n_entries);
    uint64_t total = 0, max_id = 0;
    for (uint32_t e = 0; e < n_entries; ++e) {
        ORAMA_REQUIRE(entries[e].token < params->n_tokens, "entry %u: token out of range", e);
        ORAMA_REQUIRE(entries[e].len == 0 || (entries[e].doc && entries[e].ntf), "entry %u: null arrays", e);
        total += entries[e].len;
        bool ascending = true;  // strictly ascending docs are unique — the common case, one pass, no hashing
        for (uint64_t i = 0; i < entries[e].len; ++i) {
            max_id = std::max(max_id, entries[e].doc[i]);
            if (i && entries[e].doc[i] <= entries[e].doc[i - 1]) ascending = false;
        }
        if (ascending) {
            pieces.push_back(Piece{entries[e].token, entries[e].doc, entries[e].ntf, entries[e].len});
            continue;
        }
        std::unordered_map<uint64_t, uint32_t> seen;
        seen.reserve((size_t)entries[e].len);
        std::vector<std::vector<uint64_t>> pd;
        std::vector<std::vector<float>> pn;
        for (uint64_t i = 0; i < entries[e].len; ++i) {
            const uint32_t occ = seen[entries[e].doc[i]]++;
            if (occ >= pd.size()) {
                pd.emplace_back();
                pn.emplace_back();
            }
            pd[occ].push_back(entries[e].doc[i]);
            pn[occ].push_back(entries[e].ntf[i]);
        }
        for (size_t o = 0; o < pd.size(); ++o) {
            own_doc.push_back(std::move(pd[o]));
            own_ntf.push_back(std::move(pn[o]));
            // the inner heap buffers do not move when the outer vectors grow
            pieces.push_back(Piece{entries[e].token, own_doc.back().data(), own_ntf.back().data(),
                                   (uint64_t)own_doc.back().size()});
        }
    }
    const uint32_t n_pieces = (uint32_t)pieces.size();
    ORAMA_SUPPORT(total < 0xffffffffull, "too many postings");
    if (total == 0) return ORAMA_OK;
    // local doc space: identity when ids are reasonably dense (the reference assigns sequential u64
    // ids, write/collection_document_storage.rs:73-77), otherwise the sorted set of ids that occur.
    std::vector<uint64_t> docs;
    const bool identity = max_id < 0xfffffff0ull && max_id <= 8 * total + (1u << 20);
    if (!identity) {
        docs.reserve((size_t)total);
        for (uint32_t e = 0; e < n_pieces; ++e) docs.insert(docs.end(), pieces[e].doc, pieces[e].doc + pieces[e].len);
        std::sort(docs.begin(), docs.end());
        docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
    }
    const uint64_t n_docs = identity ? max_id + 1 : docs.size();
    ScratchLease sc(ctx, kScratchRecords);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t touched_cap = total;  // one slot per posting
    QueryBuffers qb;
    ORAMA_TRY(prepare_query(sc.s.get(), n_docs, params->n_tokens, touched_cap, (uint32_t)total, &qb));

    // pack postings (local doc, ntf bits) + segments grouped by rank + doc table
    std::vector<uint32_t> rank(n_pieces, 0), per_token(kMaxTokens, 0);
    uint32_t max_rank = 0;
    for (uint32_t e = 0; e < n_pieces; ++e) {
        rank[e] = per_token[pieces[e].token]++;
        max_rank = std::max(max_rank, rank[e] + 1);
    }
    const size_t post_bytes = (size_t)total * 4;
    const size_t seg_off = ((2 * post_bytes) + 15) & ~(size_t)15;
    const size_t doc_off = (seg_off + (size_t)n_pieces * sizeof(Bm25Seg) + 15) & ~(size_t)15;
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
        for (uint32_t e = 0; e < n_pieces; ++e) {
            if (rank[e] != r || pieces[e].len == 0) continue;
            Bm25Seg g{};
            g.post_begin = cursor;
            g.virt_begin = virt;
            g.len = (uint32_t)pieces[e].len;
            g.token = pieces[e].token;
            g.boost = 1.0f;
            g.avg_len = 1.0f;
            for (uint64_t i = 0; i < pieces[e].len; ++i) {
                const uint64_t d = pieces[e].doc[i];
                uint32_t local;
                if (identity) {
                    local = (uint32_t)d;
                } else {
                    local = (uint32_t)(std::lower_bound(docs.begin(), docs.end(), d) - docs.begin());
                }
                h_pd[cursor + i] = local;
                memcpy(&h_pv[cursor + i], &pieces[e].ntf[i], 4);
            }
            cursor += pieces[e].len;
            virt += pieces[e].len;
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
    uint64_t virt_base = 0;
    for (uint32_t r = 0; r < max_rank; ++r) {
        Bm25Accum a;
        a.virt_base = virt_base;
        virt_base += rank_total[r];
        a.post_doc = reinterpret_cast<const uint32_t*>(db);
        a.post_val = reinterpret_cast<const uint32_t*>(db + post_bytes);
        a.segs = reinterpret_cast<const Bm25Seg*>(db + seg_off) + rank_begin[r];
        a.n_segs = rank_begin[r + 1] - rank_begin[r];
        a.total = rank_total[r];
        a.precomputed = true;
        a.epoch = qb.epoch;
        a.n_docs = n_docs;
        a.acc = qb.acc;
        a.slots = qb.slots;
        a.touched = qb.touched;
        a.state = qb.state;
        ORAMA_TRY(launch_bm25_accumulate(ctx, a, s));
    }
    // df → host → idf (libm log1pf, bm25.rs:78-82) → device
    // sized for the later result download too: the buffer must not be re-allocated while the idf upload reads it
    ORAMA_TRY(sc->h_out.reserve((size_t)kSelectMaxK * 12 + 64 + 2 * sizeof(Bm25State) + kMaxTokens * 4));
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
    f.slots = qb.slots;
    f.touched = qb.touched;
    f.n_slots = (uint32_t)touched_cap;
    f.state = qb.state;
    f.cand_score = qb.cand_score;
    f.cand_idx = qb.cand_idx;
    f.emit = qb.emit;
    ORAMA_TRY(launch_bm25_finalize(ctx, f, s));
    if (n_o)
        ORAMA_TRY(launch_omc_sparse(reinterpret_cast<const uint32_t*>(db + omc_off),
                                    reinterpret_cast<const float*>(db + omc_off + (size_t)n_omc * 4), n_o, qb.epoch,
                                    qb.emit, qb.cand_score, s));
    uint32_t dummy_n = 0;
    uint64_t count = 0;
    ORAMA_TRY(select_and_download(ctx, sc.s.get(), qb, reinterpret_cast<const uint64_t*>(db + doc_off),
                                  (uint32_t)touched_cap, params->top_k, out_ids, out_scores, out_n ? out_n : &dummy_n, &count));
    if (out_count) *out_count = count;
    if (!export_all) return ORAMA_OK;
    *out_n64 = count;
    if (export_capacity == 0 || count == 0) return ORAMA_OK;
    ORAMA_REQUIRE(export_capacity >= count && out_ids && out_scores, "capacity %llu < map size %llu",
                  (unsigned long long)export_capacity, (unsigned long long)count);
    ScoreMapDev m;
    m.emit = qb.emit;
    m.cand_score = qb.cand_score;
    m.cand_idx = qb.cand_idx;
    m.docs = reinterpret_cast<const uint64_t*>(db + doc_off);
    m.epoch = qb.epoch;
    ORAMA_TRY(sc->out_ids.reserve((size_t)count * 8));
    ORAMA_TRY(sc->out_val.reserve((size_t)count * 4));
    ORAMA_TRY(sc->out_n.reserve(16));
    ORAMA_TRY(launch_scores_export(m, (uint32_t)touched_cap, sc->out_ids.as<uint64_t>(), sc->out_val.as<float>(),
                                   sc->out_n.as<uint32_t>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_ids, sc->out_ids.p, (size_t)count * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_scores, sc->out_val.p, (size_t)count * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    return ORAMA_OK;
}

int orama_bm25_score(orama_ctx* ctx, const orama_ntf_entry* entries, uint32_t n_entries,
                     const orama_bm25_params* params, const uint64_t* omc_doc, const float* omc_mul,
                     uint64_t n_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                     uint64_t* out_count) {
    ORAMA_REQUIRE(out_n, "null argument");
    return bm25_score_impl(ctx, entries, n_entries, params, omc_doc, omc_mul, n_omc, out_ids, out_scores, out_n, out_count, 0,
                           nullptr);
}

int orama_bm25_score_map(orama_ctx* ctx, const orama_ntf_entry* entries, uint32_t n_entries,
                         const orama_bm25_params* params, const uint64_t* omc_doc, const float* omc_mul, uint64_t n_omc,
                         uint64_t capacity, uint64_t* out_ids, float* out_scores, uint64_t* out_n) {
    ORAMA_REQUIRE(out_n, "null argument");
    return bm25_score_impl(ctx, entries, n_entries, params, omc_doc, omc_mul, n_omc, out_ids, out_scores, nullptr, nullptr,
                           capacity, out_n);
}

int orama_hybrid_combine(orama_ctx* ctx, const uint64_t* vec_doc, const float* vec_score, uint64_t n_vec,
                         const uint64_t* ft_doc, const float* ft_score, uint64_t n_ft, uint32_t top_k,
                         uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    ORAMA_REQUIRE(ctx && out_n, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_SUPPORT(top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", top_k, kSelectMaxK);
    ORAMA_REQUIRE(n_vec == 0 || (vec_doc && vec_score), "null vector map");
    ORAMA_REQUIRE(n_ft == 0 || (ft_doc && ft_score), "null fulltext map");
    ORAMA_REQUIRE(top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_SUPPORT(n_vec + n_ft < 0xffffffffull, "maps too large");
    if (n_vec + n_ft == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(ctx->device);
    // local doc space = sorted union of both key sets
    std::vector<uint64_t> docs;
    docs.reserve((size_t)(n_vec + n_ft));
    docs.insert(docs.end(), vec_doc, vec_doc + n_vec);
    docs.insert(docs.end(), ft_doc, ft_doc + n_ft);
    std::sort(docs.begin(), docs.end());
    docs.erase(std::unique(docs.begin(), docs.end()), docs.end());
    const uint64_t n_docs = docs.size();
    ScratchLease sc(ctx, kScratchRecords);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t cand_cap = n_ft + n_vec;
    QueryBuffers qb;
    ORAMA_TRY(prepare_query(sc.s.get(), n_docs, 1, cand_cap, (uint32_t)n_ft, &qb));
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
    ORAMA_SUPPORT(top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", top_k, kSelectMaxK);
    ORAMA_SUPPORT(n < 0xffffffffull, "top_n limited to 2^32-1 entries");
    ORAMA_ON_DEVICE(ctx->device);
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
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * kSelectMaxK));  // (lists of up to 16 chunks: the two-launch form)
    ORAMA_TRY(sc->out_ids.reserve((size_t)top_k * 8));
    ORAMA_TRY(sc->out_val.reserve((size_t)top_k * 4));
    ORAMA_TRY(sc->out_n.reserve(4));
    SelectPlan p;
    p.vals = sc->dist.as<float>();
    p.stride = n;
    p.n = (uint32_t)n;
    p.q = 1;
    p.k = top_k;
    p.keys_capacity = kSelectMaxK;
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

// Reciprocal-rank fusion — EXTRA, not the parity path: the reference merges by min-max + sum
// (normalize_and_combine); north_star names RRF, so it is offered next to it.  Both lists are cut to their best
// `depth` entries with K4 (score desc, DocumentId asc), ranks start at 1, score[doc] = sum over the lists holding
// doc of 1 / (rrf_k + rank) in f32 — the full-text term first, then the vector term, like the reference's combine
// adds vector' onto fulltext'.  count = documents in the union of the two cut lists.
int orama_hybrid_rrf(orama_ctx* ctx, const uint64_t* vec_doc, const float* vec_score, uint64_t n_vec,
                     const uint64_t* ft_doc, const float* ft_score, uint64_t n_ft, float rrf_k, uint32_t depth,
                     uint32_t top_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    ORAMA_REQUIRE(ctx && out_n, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    ORAMA_REQUIRE(depth >= 1 && depth <= kSelectMaxK, "rrf depth %u outside [1, %u]", depth, kSelectMaxK);
    ORAMA_REQUIRE(top_k == 0 || (out_ids && out_scores), "null output");
    ORAMA_REQUIRE(rrf_k >= 0.0f, "rrf_k must be >= 0");
    std::vector<uint64_t> fd(depth), vd(depth);
    std::vector<float> fs(depth), vs(depth);
    uint32_t nf = 0, nv = 0;
    ORAMA_TRY(orama_top_n(ctx, ft_doc, ft_score, n_ft, depth, fd.data(), fs.data(), &nf));
    ORAMA_TRY(orama_top_n(ctx, vec_doc, vec_score, n_vec, depth, vd.data(), vs.data(), &nv));
    std::vector<std::pair<uint64_t, float>> fused;  // (doc, score), full-text order first
    fused.reserve((size_t)nf + nv);
    for (uint32_t r = 0; r < nf; ++r) fused.emplace_back(fd[r], 0.0f + 1.0f / (rrf_k + (float)(r + 1)));
    std::vector<std::pair<uint64_t, uint32_t>> pos;  // doc -> index in fused
    pos.reserve(nf);
    for (uint32_t r = 0; r < nf; ++r) pos.emplace_back(fd[r], r);
    std::sort(pos.begin(), pos.end());
    for (uint32_t r = 0; r < nv; ++r) {
        const float term = 1.0f / (rrf_k + (float)(r + 1));
        auto it = std::lower_bound(pos.begin(), pos.end(), std::make_pair(vd[r], 0u));
        if (it != pos.end() && it->first == vd[r]) fused[it->second].second = fused[it->second].second + term;
        else fused.emplace_back(vd[r], 0.0f + term);
    }
    if (out_count) *out_count = fused.size();
    std::sort(fused.begin(), fused.end(), [](const std::pair<uint64_t, float>& a, const std::pair<uint64_t, float>& b) {
        return a.second != b.second ? a.second > b.second : a.first < b.first;
    });
    const uint32_t n = (uint32_t)std::min<size_t>(top_k, fused.size());
    for (uint32_t i = 0; i < n; ++i) {
        out_ids[i] = fused[i].first;
        out_scores[i] = fused[i].second;
    }
    *out_n = n;
    return ORAMA_OK;
}


// ------------------------------------------------------------------ full-text batches over a shard group
// orama_post_search_batch for an index sharded by document range, for EVERY shape of group: all shards in this
// process (co-located, or one process driving several GPUs) or one process per rank.  The reference reads one index-wide
// quantity before it scores — df per token = corpus_docs.len() (token_score.rs:262-275) — and merges what the indexes return
// (sort.rs:260-279, search.rs:482).  Per block of <= 512 queries, three phases on every shard and TWO collectives however many
// queries the block holds (round 3 had no rank form: such groups answered a batch one staged query at a time, 1.5 K/s):
//   A  this shard's df of every token of every query — the list length where that is exact; under a filter or with several
//      lists per token, the counting launch of the range scorer (K3r) — summed over the shards: ONE all-reduce of
//      n_queries x (64 + 2) words (the two extra words: "a shard cannot take this query on the range scorer" and "a
//      shard failed", so that every rank takes the same decision without a second exchange);
//   B  every shard scores the block with the range scorer, idf from the index-wide df: its own top-k lists and counts;
//   C  ONE all-gather of the shards' blocks [ids | scores | n | count]; every rank merges them by (score desc,
//      DocumentId asc) and sums the counts.
// Queries the range scorer does not take (more than 64 non-empty lists, ...) are answered afterwards, one by one, by
// orama_shard_post_search — in the same order on every rank.  Same answers as orama_post_search_batch over the union of the
// shards, bit for bit.  Every rank of the group must make the same call (same queries, same order).
int orama_shard_post_search_batch(orama_shard_group* g, orama_post* const* shards, const orama_post_query_desc* queries,
                                  uint32_t n_queries, float b, const uint64_t* const* allow_bitmaps, uint64_t bitmap_bits,
                                  int apply_omc, uint32_t stride_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                                  uint64_t* out_count, int* out_status) {
    ORAMA_REQUIRE(g && shards && (n_queries == 0 || (queries && out_n)), "null argument");
    if (n_queries == 0) return ORAMA_OK;
    ORAMA_REQUIRE(stride_k >= 1 && out_ids && out_scores, "null output");
    uint32_t world = 0, nl = 0;
    shard_group_shape(g, &world, &nl, nullptr);
    for (uint32_t i = 0; i < nl; ++i) ORAMA_REQUIRE(shards[i], "null shard %u", i);
    ShardCall call(g);  // one lane of the group from the first collective to the last
    ORAMA_TRY(call.init());
    static const bool trace = orama::dev_env("ORAMA_SHARD_BATCH_TRACE") != nullptr;  // phase times of every block on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    std::vector<int> status(n_queries, ORAMA_OK);
    std::vector<std::string> errors(n_queries);
    for (uint32_t j = 0; j < n_queries; ++j) {
        out_n[j] = 0;
        if (out_count) out_count[j] = 0;
    }
    std::vector<uint32_t> slow;
    constexpr uint32_t kBlockQueries = 512;  // queries per exchange (a block of 512 x top-100 lists is 620 KB per shard)
    constexpr uint32_t kWordsPerQuery = kMaxTokens + 2;
    int call_status = ORAMA_OK;  // a systemic failure (a shard's scratch pool, the device): the same on every rank, ends the call
    for (uint32_t q0 = 0; q0 < n_queries && call_status == ORAMA_OK; q0 += kBlockQueries) {
        const uint32_t nq = std::min(kBlockQueries, n_queries - q0);
        // ---- which queries of the block can take the three phases at all (rank-independent tests first)
        std::vector<uint32_t> cand;  // indices into `queries`
        for (uint32_t j = q0; j < q0 + nq; ++j) {
            const orama_post_query_desc& qd = queries[j];
            if (qd.params.top_k > stride_k || qd.params.top_k == 0) {
                set_error("query %u: top_k %u outside [1, stride %u]", j, qd.params.top_k, stride_k);
                status[j] = ORAMA_ERR_INVALID;
                errors[j] = orama_last_error();
                continue;
            }
            bool ok = check_params(&qd.params) == ORAMA_OK && qd.n_refs <= kRangeMaxRefs && (qd.n_refs == 0 || qd.refs);
            for (uint32_t r = 0; r < qd.n_refs && ok; ++r) ok = qd.refs[r].token < qd.params.n_tokens;
            if (ok) cand.push_back(j);
            else slow.push_back(j);  // (the one-by-one path reports what is wrong with it)
        }
        const uint32_t nc = (uint32_t)cand.size();
        if (nc == 0) continue;
        // every shard is read-locked during its df pass and again during its scoring pass — NOT across the exchange between them
        // (a writer on any rank waited for a network collective): a store that changed in between is noticed
        // (`mutations`) and the block's queries go to the staged one-by-one form, which holds its locks from df to answer
        std::vector<std::shared_lock<std::shared_mutex>> locks;
        for (uint32_t i = 0; i < nl; ++i) locks.emplace_back(shards[i]->mu);
        std::vector<uint64_t> mutations_a(nl);
        for (uint32_t i = 0; i < nl; ++i) mutations_a[i] = shards[i]->mutations;
        // a failure that belongs to a QUERY (INVALID / UNSUPPORTED out of a shard's pass) is not the call's: those queries take
        // the one-by-one form, which reports per query; BUSY / HIP / OOM are the call's
        auto query_specific = [](int st) { return st == ORAMA_ERR_INVALID || st == ORAMA_ERR_UNSUPPORTED; };
        std::vector<uint32_t> words((size_t)nc * kWordsPerQuery, 0);  // [df x 64 | cannot | failed] per query, summed over the shards
        auto df_of_query = [&](size_t c) { return &words[c * kWordsPerQuery]; };
        std::vector<std::vector<uint32_t>> shard_df(nl, std::vector<uint32_t>((size_t)nc * kMaxTokens, 0));
        std::vector<int> shard_status(nl, ORAMA_OK);
        std::vector<std::string> shard_error(nl);
        std::vector<char> can((size_t)nl * nc, 1);
        auto on_shards = [&](const std::function<int(uint32_t)>& body) {
            auto run = [&](uint32_t i) {
                shard_status[i] = body(i);
                if (shard_status[i] != ORAMA_OK) shard_error[i] = orama_last_error();  // this thread's slot
            };
            std::vector<std::thread> pool;
            for (uint32_t i = 1; i < nl; ++i) pool.emplace_back(run, i);
            run(0);
            for (auto& t : pool) t.join();
        };
        // ---- phase A: this process's df of every token (+ the two flags)
        const auto t_block = now();
        on_shards([&](uint32_t i) -> int {
            orama_post* p = shards[i];
            ORAMA_ON_DEVICE(p->ctx->device);
            std::vector<RangeJob> jobs;
            std::vector<uint32_t> dummy_n(nc, 0);
            for (uint32_t c = 0; c < nc; ++c) {
                const orama_post_query_desc& qd = queries[cand[c]];
                if (!ranges_eligible(p, qd.refs, qd.n_refs, &qd.params)) {
                    can[(size_t)i * nc + c] = 0;
                    continue;
                }
                RangeJob job{qd.refs, qd.n_refs, &qd.params, nullptr, nullptr, &dummy_n[c], nullptr};
                job.df_out = &shard_df[i][(size_t)c * kMaxTokens];
                jobs.push_back(job);
            }
            if (jobs.empty()) return ORAMA_OK;
            ScratchLease sc(p->ctx), sc2(p->ctx);
            const bool two = jobs.size() > kRangeBatchMax;
            if (two) ORAMA_TRY(ScratchLease::init_pair(sc, sc2));
            else ORAMA_TRY(sc.init());
            return post_search_ranges(p, sc.s.get(), jobs.data(), (uint32_t)jobs.size(), b, allow_bitmaps ? allow_bitmaps[i] : nullptr,
                                      allow_bitmaps ? bitmap_bits : 0, apply_omc, two ? sc2.s.get() : nullptr, /*df_pass=*/true);
        });
        for (uint32_t i = 0; i < nl; ++i)
            for (uint32_t c = 0; c < nc; ++c) {
                uint32_t* w = df_of_query(c);
                for (uint32_t t = 0; t < kMaxTokens; ++t) w[t] += shard_df[i][(size_t)c * kMaxTokens + t];
                if (!can[(size_t)i * nc + c] || query_specific(shard_status[i])) w[kMaxTokens] += 1;
                if (shard_status[i] != ORAMA_OK && !query_specific(shard_status[i])) w[kMaxTokens + 1] += 1;
            }
        int local_fail = ORAMA_OK;
        std::string local_error;
        for (uint32_t i = 0; i < nl && local_fail == ORAMA_OK; ++i)
            if (shard_status[i] != ORAMA_OK && !query_specific(shard_status[i])) local_fail = shard_status[i], local_error = shard_error[i];
        locks.clear();
        const double ms_a = ms_since(t_block);
        // ---- exchange 1: index-wide df and flags
        {
            const int st = shard_sum_u32(g, words.data(), words.size());
            if (st != ORAMA_OK) return st;  // the transport itself failed: nothing more can be agreed on
        }
        bool any_failed = false;
        for (uint32_t c = 0; c < nc; ++c) any_failed = any_failed || df_of_query(c)[kMaxTokens + 1] != 0;
        if (any_failed) {  // (seen by every rank: all of them stop here)
            call_status = local_fail != ORAMA_OK ? local_fail : ORAMA_ERR_HIP;
            if (local_fail != ORAMA_OK) set_error("sharded batch: %s", local_error.c_str());
            else set_error("sharded batch: the df pass failed on another rank");
            break;
        }
        std::vector<uint32_t> fast;  // positions in `cand`
        for (uint32_t c = 0; c < nc; ++c) {
            if (df_of_query(c)[kMaxTokens] == 0) fast.push_back(c);
            else slow.push_back(cand[c]);
        }
        const uint32_t nf = (uint32_t)fast.size();
        if (nf == 0) continue;
        const double ms_x1 = ms_since(t_block) - ms_a;
        const auto t_b = now();
        // ---- phase B: every local shard scores the fast queries with the index-wide df -> its block
        // block of one shard: [nf x stride_k ids u64][nf x stride_k scores f32][nf n u32][nf count u64][failed u32, pad]
        const size_t ids_bytes = (size_t)nf * stride_k * 8, sc_bytes = (size_t)nf * stride_k * 4, n_bytes = (((size_t)nf * 4) + 7) & ~(size_t)7;
        const size_t cnt_bytes = (size_t)nf * 8, block_bytes = ids_bytes + ((sc_bytes + 7) & ~(size_t)7) + n_bytes + cnt_bytes + 8;
        const size_t off_sc = ids_bytes, off_n = off_sc + ((sc_bytes + 7) & ~(size_t)7), off_cnt = off_n + n_bytes, off_fail = off_cnt + cnt_bytes;
        std::vector<char> local_blocks(block_bytes * nl, 0);
        std::vector<uint32_t> df_fast((size_t)nf * kMaxTokens);
        for (uint32_t x = 0; x < nf; ++x) memcpy(&df_fast[(size_t)x * kMaxTokens], df_of_query(fast[x]), sizeof(uint32_t) * kMaxTokens);
        for (uint32_t i = 0; i < nl; ++i) locks.emplace_back(shards[i]->mu);
        bool stale = false;  // a store of this process changed between the df pass and now: its df no longer describes it
        for (uint32_t i = 0; i < nl; ++i) stale = stale || shards[i]->mutations != mutations_a[i];
        if (!stale) on_shards([&](uint32_t i) -> int {
            orama_post* p = shards[i];
            ORAMA_ON_DEVICE(p->ctx->device);
            char* blk = local_blocks.data() + (size_t)i * block_bytes;
            std::vector<RangeJob> jobs;
            jobs.reserve(nf);
            for (uint32_t x = 0; x < nf; ++x) {
                const orama_post_query_desc& qd = queries[cand[fast[x]]];
                RangeJob job{qd.refs, qd.n_refs, &qd.params, reinterpret_cast<uint64_t*>(blk) + (size_t)x * stride_k,
                             reinterpret_cast<float*>(blk + off_sc) + (size_t)x * stride_k, reinterpret_cast<uint32_t*>(blk + off_n) + x,
                             reinterpret_cast<uint64_t*>(blk + off_cnt) + x};
                job.df_global = &df_fast[(size_t)x * kMaxTokens];
                jobs.push_back(job);
            }
            ScratchLease sc(p->ctx), sc2(p->ctx);
            const bool two = jobs.size() > kRangeBatchMax;
            if (two) ORAMA_TRY(ScratchLease::init_pair(sc, sc2));
            else ORAMA_TRY(sc.init());
            return post_search_ranges(p, sc.s.get(), jobs.data(), (uint32_t)jobs.size(), b, allow_bitmaps ? allow_bitmaps[i] : nullptr,
                                      allow_bitmaps ? bitmap_bits : 0, apply_omc, two ? sc2.s.get() : nullptr);
        });
        local_fail = ORAMA_OK;
        for (uint32_t i = 0; i < nl; ++i) {
            uint32_t* word = reinterpret_cast<uint32_t*>(local_blocks.data() + (size_t)i * block_bytes + off_fail);
            if (stale || query_specific(shard_status[i])) {
                *word = 2u;  // not the call's failure: the block's queries are answered one by one
            } else if (shard_status[i] != ORAMA_OK) {
                *word = 1u;
                if (local_fail == ORAMA_OK) local_fail = shard_status[i], local_error = shard_error[i];
            }
        }
        locks.clear();
        const double ms_b = ms_since(t_b);
        const auto t_x2 = now();
        // ---- exchange 2: every shard's block
        std::vector<char> all_blocks(block_bytes * world);
        {
            const int st = shard_gather_blocks(g, local_blocks.data(), block_bytes, all_blocks.data());
            if (st != ORAMA_OK) return st;
        }
        bool any_one_by_one = false;
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t word = *reinterpret_cast<const uint32_t*>(all_blocks.data() + (size_t)r * block_bytes + off_fail);
            any_failed = any_failed || word == 1u;
            any_one_by_one = any_one_by_one || word == 2u;
        }
        if (any_failed) {
            call_status = local_fail != ORAMA_OK ? local_fail : ORAMA_ERR_HIP;
            if (local_fail != ORAMA_OK) set_error("sharded batch: %s", local_error.c_str());
            else set_error("sharded batch: the scoring pass failed on another rank");
            break;
        }
        if (any_one_by_one) {  // (seen by every rank: the same queries join `slow` everywhere)
            for (uint32_t x = 0; x < nf; ++x) slow.push_back(cand[fast[x]]);
            continue;
        }
        const double ms_x2 = ms_since(t_x2);
        const auto t_c = now();
        // ---- phase C: merge by (score desc, DocumentId asc), sum the counts
        // (every shard's list is already in that order: a k-way merge of `world` sorted lists — a full sort of world x k hits per
        // query was 7 us per query on the host, a third of the call)
        std::vector<uint32_t> head(world);
        for (uint32_t x = 0; x < nf; ++x) {
            const uint32_t j = cand[fast[x]], k = queries[j].params.top_k;
            uint64_t count = 0;
            for (uint32_t r = 0; r < world; ++r) {
                head[r] = 0;
                count += reinterpret_cast<const uint64_t*>(all_blocks.data() + (size_t)r * block_bytes + off_cnt)[x];
            }
            uint32_t n = 0;
            for (; n < k; ++n) {
                int best = -1;
                float best_sc = 0.0f;
                uint64_t best_id = 0;
                for (uint32_t r = 0; r < world; ++r) {
                    const char* blk = all_blocks.data() + (size_t)r * block_bytes;
                    const uint32_t len = std::min(reinterpret_cast<const uint32_t*>(blk + off_n)[x], stride_k);
                    if (head[r] >= len) continue;
                    const float sc = (reinterpret_cast<const float*>(blk + off_sc) + (size_t)x * stride_k)[head[r]];
                    const uint64_t id = (reinterpret_cast<const uint64_t*>(blk) + (size_t)x * stride_k)[head[r]];
                    if (best < 0 || sc > best_sc || (sc == best_sc && id < best_id)) best = (int)r, best_sc = sc, best_id = id;
                }
                if (best < 0) break;
                ++head[best];
                out_ids[(size_t)j * stride_k + n] = best_id;
                out_scores[(size_t)j * stride_k + n] = best_sc;
            }
            out_n[j] = n;
            if (out_count) out_count[j] = count;
        }
        if (trace)
            fprintf(stderr, "[shard batch] %u queries, %u local shards of %u: df pass %.3f ms | exchange %.3f | scoring %.3f | gather %.3f | merge %.3f\n",
                    nf, nl, world, ms_a, ms_x1, ms_b, ms_x2, ms_since(t_c));
    }
    if (call_status != ORAMA_OK) {  // systemic: every query of the call carries it (nothing is retried one by one)
        const std::string err = orama_last_error();
        for (uint32_t j = 0; j < n_queries; ++j) {
            out_n[j] = 0;
            if (out_count) out_count[j] = 0;
            if (out_status) out_status[j] = call_status;
        }
        set_error("%s", err.c_str());
        return call_status;
    }
    // (the shard locks are released: the staged query below takes them itself; `slow` is the same list on every rank)
    std::sort(slow.begin(), slow.end());
    for (uint32_t j : slow) {
        const orama_post_query_desc& qd = queries[j];
        uint64_t cnt = 0;
        const int st = orama_shard_post_search(g, shards, qd.refs, qd.n_refs, b, &qd.params, allow_bitmaps, bitmap_bits, apply_omc, 0,
                                               nullptr, nullptr, 0, out_ids + (size_t)j * stride_k, out_scores + (size_t)j * stride_k,
                                               &out_n[j], &cnt);
        if (st == ORAMA_OK) {
            if (out_count) out_count[j] = cnt;
        } else {
            status[j] = st;
            errors[j] = orama_last_error();
            out_n[j] = 0;
        }
    }
    int first = ORAMA_OK;
    for (uint32_t j = 0; j < n_queries; ++j) {
        if (out_status) out_status[j] = status[j];
        if (first == ORAMA_OK && status[j] != ORAMA_OK) {
            first = status[j];
            set_error("query %u: %s", j, errors[j].c_str());
        }
    }
    return first;
}

}  // extern "C"
