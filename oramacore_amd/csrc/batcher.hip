// batcher.hip — request micro-batcher in front of orama_vec_search.
//
// The reference has no batch entry: every HTTP request reaches EmbeddingFieldStorage::search
// (embedding_field.rs:250-278) alone, on its own tokio worker.  One corpus pass costs the same HBM traffic for 1
// query as for 64 (K2 is bandwidth-bound), so concurrent single-query callers are coalesced here: callers block in
// orama_batcher_search, a dispatcher thread turns whatever is pending into ONE orama_vec_search(q = batch); two
// dispatchers alternate so that the host side of one batch overlaps the corpus pass of the other.
// No artificial delay by default — batches form naturally while the previous pass occupies the GPU; max_wait_us > 0
// additionally holds an under-full batch open for that long after its first request.
//
// Per-request k: the batch runs with the largest k; the total order (distance asc, doc asc, row asc) makes every
// smaller-k answer a prefix of it.
// Filters: a batch shares ONE allow bitmap, so requests are grouped by their (bitmap pointer, bits) pair — the
// dispatcher takes the oldest pending request and every later one carrying the same pair.  That is the common case:
// while deletes are pending EVERY search of an index carries the same NOT-deleted predicate
// (index/filter.rs:344-392), which the shim keeps as one resident bitmap (orama_allow_*), so single-query traffic
// stays on the MFMA path under live deletes; requests with other filters simply form their own batches.
#include <chrono>
#include <condition_variable>
#include <deque>
#include <string>
#include <thread>

#include "common.hpp"
#include "select.hpp"
#include "vec_internal.hpp"

using namespace orama;

namespace {
struct Request {
    const float* query = nullptr;
    uint32_t k = 0;
    const uint64_t* allow = nullptr;  // resident token or caller-owned host words (borrowed while the caller blocks)
    uint64_t allow_bits = 0;
    uint64_t* out_ids = nullptr;
    float* out_dist = nullptr;
    uint32_t* out_n = nullptr;
    int status = ORAMA_OK;
    std::string error;
    bool done = false;
    // The caller sleeps on its own request — own condition variable AND own mutex: a finished batch wakes exactly its
    // members, and neither the wake-ups nor the result copies touch the batcher's queue mutex (with one shared mutex
    // 256 woken callers and the other dispatcher's gathering queued up behind each other: the GPU idled a third of
    // the time at 512 callers).
    std::mutex m;
    std::condition_variable cv;
};
}  // namespace

struct orama_batcher {
    orama_vec* v = nullptr;
    // a batcher in front of a SHARD GROUP (orama_batcher_create_group): passes go through orama_shard_vec_search over
    // `shards`; a request's filter is then the caller's array of resident per-shard tokens (grouped by its address)
    orama_shard_group* group = nullptr;
    std::vector<orama_vec*> shards;
    uint32_t dim = 0;
    uint32_t max_batch = 64;
    uint32_t max_wait_us = 0;
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<Request*> pending;
    bool stop = false;
    uint64_t n_requests = 0, n_batches = 0;
    uint32_t largest = 0;
    // Two dispatchers alternate.  A corpus pass costs the same for 1 query as for 256, so a batch must be gathered as
    // LATE as possible: a dispatcher takes `pass_mu` first and only then collects what is pending, runs the pass and
    // releases it; handing the results out happens outside — beside the other dispatcher's pass.
    std::mutex pass_mu;
    std::vector<std::thread> workers;

    void run() {
        std::vector<Request*> batch;
        std::vector<float> queries;
        std::vector<uint64_t> ids;
        std::vector<float> dist;
        std::vector<uint32_t> cnt;
        for (;;) {
            batch.clear();
            std::unique_lock<std::mutex> pass(pass_mu);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !pending.empty(); });
                if (pending.empty() && stop) return;
                if (max_wait_us && pending.size() < max_batch) {
                    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(max_wait_us);
                    cv_work.wait_until(lk, deadline, [&] { return stop || pending.size() >= max_batch; });
                    if (pending.empty()) continue;  // the other dispatcher took them
                }
                // the oldest request decides the filter of this batch; later requests with the same filter join it
                const uint64_t* allow = pending.front()->allow;
                const uint64_t bits = pending.front()->allow_bits;
                for (auto it = pending.begin(); it != pending.end() && batch.size() < max_batch;) {
                    if ((*it)->allow == allow && (*it)->allow_bits == bits) {
                        batch.push_back(*it);
                        it = pending.erase(it);
                    } else {
                        ++it;
                    }
                }
                n_requests += batch.size();
                n_batches += 1;
                largest = std::max<uint32_t>(largest, (uint32_t)batch.size());
            }
            const uint32_t q = (uint32_t)batch.size();
            uint32_t kmax = 0;
            for (Request* r : batch) kmax = std::max(kmax, r->k);
            queries.resize((size_t)q * dim);
            for (uint32_t i = 0; i < q; ++i) memcpy(&queries[(size_t)i * dim], batch[i]->query, (size_t)dim * 4);
            ids.assign((size_t)q * std::max(kmax, 1u), 0);
            dist.assign((size_t)q * std::max(kmax, 1u), 0.f);
            cnt.assign(q, 0);
            int st = ORAMA_OK;
            std::string err;
            auto pass_search = [&](const float* qs, uint32_t nq, uint32_t k, uint64_t* o_ids, float* o_dist, uint32_t* o_n) -> int {
                if (group)
                    return orama_shard_vec_search(group, shards.data(), qs, nq, k, reinterpret_cast<const uint64_t* const*>(batch[0]->allow),
                                                  batch[0]->allow_bits, o_ids, o_dist, o_n);
                return orama_vec_search(v, qs, nq, k, batch[0]->allow, batch[0]->allow_bits, o_ids, o_dist, o_n);
            };
            if (kmax > 0) {
                st = pass_search(queries.data(), q, kmax, ids.data(), dist.data(), cnt.data());
                if (st != ORAMA_OK) err = orama_last_error();  // this thread's error slot
            }
            // a pass that failed for a reason ONE request can be responsible for (a bad argument, a shape outside the
            // envelope) is repeated one request at a time, so that only the request that cannot be served sees the error.
            // A systemic failure — the scratch pool saturated (BUSY after the acquire timeout), the device, memory — would
            // fail every retry the same way, each waiting out the timeout again while the pass lock is held: it goes to
            // every caller of the batch at once.
            std::vector<int> sts(q, st);
            std::vector<std::string> errs(q, err);
            const bool query_specific = st == ORAMA_ERR_INVALID || st == ORAMA_ERR_UNSUPPORTED;
            if (st != ORAMA_OK && q > 1 && query_specific) {
                for (uint32_t i = 0; i < q; ++i) {
                    sts[i] = ORAMA_OK;
                    if (batch[i]->k == 0) continue;
                    sts[i] = pass_search(&queries[(size_t)i * dim], 1, batch[i]->k, &ids[(size_t)i * kmax], &dist[(size_t)i * kmax], &cnt[i]);
                    if (sts[i] != ORAMA_OK) errs[i] = orama_last_error();
                }
            }
            pass.unlock();
            for (uint32_t i = 0; i < q; ++i) {
                Request* r = batch[i];
                const int st = sts[i];
                r->status = st;
                r->error = errs[i];
                if (st == ORAMA_OK) {
                    const uint32_t n = std::min(cnt[i], r->k);
                    memcpy(r->out_ids, &ids[(size_t)i * kmax], (size_t)n * 8);
                    memcpy(r->out_dist, &dist[(size_t)i * kmax], (size_t)n * 4);
                    *r->out_n = n;
                }
                std::lock_guard<std::mutex> rl(r->m);  // notified under the lock: the request lives on the caller's stack
                r->done = true;
                r->cv.notify_one();
            }
        }
    }
};

extern "C" {

int orama_batcher_create(orama_vec* v, uint32_t max_batch, uint32_t max_wait_us, orama_batcher** out) {
    ORAMA_REQUIRE(v && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(max_batch >= 1 && max_batch <= 1024, "max_batch %u outside [1, 1024]", max_batch);
    orama_batcher* b = new (std::nothrow) orama_batcher();
    if (!b) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    b->v = v;
    b->dim = vec_dim(v);
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    for (int w = 0; w < 2; ++w) b->workers.emplace_back([b] { b->run(); });
    *out = b;
    return ORAMA_OK;
}

int orama_batcher_create_group(orama_shard_group* g, orama_vec* const* shards, uint32_t max_batch, uint32_t max_wait_us,
                               orama_batcher** out) {
    ORAMA_REQUIRE(g && shards && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(max_batch >= 1 && max_batch <= 1024, "max_batch %u outside [1, 1024]", max_batch);
    uint32_t world = 0, n_local = 0;
    ORAMA_TRY(orama_shard_group_info(g, &world, &n_local, nullptr, nullptr));
    // batches form from whatever requests have arrived: with ranks in other processes every process would form different
    // batches and issue different collectives
    ORAMA_SUPPORT(world == n_local, "a request batcher needs every shard of the group in this process (%u of %u are local)", n_local, world);
    for (uint32_t i = 0; i < n_local; ++i) ORAMA_REQUIRE(shards[i], "null shard %u", i);
    orama_batcher* b = new (std::nothrow) orama_batcher();
    if (!b) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    b->group = g;
    b->shards.assign(shards, shards + n_local);
    b->v = shards[0];
    b->dim = vec_dim(shards[0]);
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    for (int w = 0; w < 2; ++w) b->workers.emplace_back([b] { b->run(); });
    *out = b;
    return ORAMA_OK;
}

void orama_batcher_destroy(orama_batcher* b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;  // pending requests are still served before the dispatcher leaves
    }
    b->cv_work.notify_all();
    for (auto& t : b->workers)
        if (t.joinable()) t.join();
    delete b;
}

int orama_batcher_search(orama_batcher* b, const float* query, uint32_t k, uint64_t* out_ids, float* out_dist,
                         uint32_t* out_n) {
    return orama_batcher_search_filtered(b, query, k, nullptr, 0, out_ids, out_dist, out_n);
}

int orama_batcher_search_filtered(orama_batcher* b, const float* query, uint32_t k, const uint64_t* allow_bitmap,
                                  uint64_t bitmap_bits, uint64_t* out_ids, float* out_dist, uint32_t* out_n) {
    ORAMA_REQUIRE(b && query && out_n, "null argument");
    *out_n = 0;
    if (k == 0) return ORAMA_OK;
    ORAMA_REQUIRE(out_ids && out_dist, "null output");
    ORAMA_SUPPORT(k <= kSelectMaxK, "limit %u exceeds the supported maximum %u", k, kSelectMaxK);
    Request r;
    r.query = query;
    r.k = k;
    r.allow = allow_bitmap;
    r.allow_bits = allow_bitmap ? bitmap_bits : 0;
    r.out_ids = out_ids;
    r.out_dist = out_dist;
    r.out_n = out_n;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        ORAMA_REQUIRE(!b->stop, "batcher is shutting down");
        b->pending.push_back(&r);
        b->cv_work.notify_one();
    }
    {
        std::unique_lock<std::mutex> rl(r.m);
        r.cv.wait(rl, [&] { return r.done; });
    }
    if (r.status != ORAMA_OK) set_error("%s", r.error.c_str());
    return r.status;
}

int orama_batcher_stats(orama_batcher* b, uint64_t* requests, uint64_t* batches, uint32_t* largest_batch) {
    ORAMA_REQUIRE(b, "null handle");
    std::lock_guard<std::mutex> lk(b->mu);
    if (requests) *requests = b->n_requests;
    if (batches) *batches = b->n_batches;
    if (largest_batch) *largest_batch = b->largest;
    return ORAMA_OK;
}

}  // extern "C"

// ================================================================= full-text request batcher
// Same idea for BM25 searches: the reference's tokio workers each call search_full_text alone; K3r (bm25_ranges.hip)
// scores 32 queries per set of launches, so concurrent single-query callers are coalesced into orama_post_search_batch
// calls.  Requests are grouped by (filter bitmap, b, apply_omc) — the arguments a batch shares.  Two dispatcher
// threads alternate, so that the host part of one batch (tables, read-back) overlaps the device part of the other.
namespace {
struct PostRequest {
    orama_post_query_desc desc;
    float b = 0.75f;
    const uint64_t* allow = nullptr;
    uint64_t allow_bits = 0;
    int apply_omc = 0;
    uint64_t* out_ids = nullptr;
    float* out_scores = nullptr;
    uint32_t* out_n = nullptr;
    uint64_t* out_count = nullptr;
    int status = ORAMA_OK;
    std::string error;
    bool done = false;
    std::mutex m;  // own mutex and condition variable: see Request
    std::condition_variable cv;
};
}  // namespace

struct orama_post_batcher {
    orama_post* p = nullptr;
    // in front of a shard group (orama_post_batcher_create_group): dispatches go through orama_shard_post_search_batch
    orama_shard_group* group = nullptr;
    std::vector<orama_post*> shards;
    uint32_t max_batch = 64;
    uint32_t max_wait_us = 0;
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<PostRequest*> pending;
    bool stop = false;
    uint64_t n_requests = 0, n_batches = 0;
    uint32_t largest = 0;
    std::vector<std::thread> workers;

    void run() {
        std::vector<PostRequest*> batch;
        std::vector<orama_post_query_desc> descs;
        std::vector<uint64_t> ids, counts;
        std::vector<float> scores;
        std::vector<uint32_t> ns;
        std::vector<int> sts;
        for (;;) {
            batch.clear();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !pending.empty(); });
                if (pending.empty() && stop) return;
                if (max_wait_us && pending.size() < max_batch) {
                    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(max_wait_us);
                    cv_work.wait_until(lk, deadline, [&] { return stop || pending.size() >= max_batch; });
                    if (pending.empty()) continue;  // the other dispatcher took them
                }
                const PostRequest* first = pending.front();
                for (auto it = pending.begin(); it != pending.end() && batch.size() < max_batch;) {
                    PostRequest* r = *it;
                    if (r->allow == first->allow && r->allow_bits == first->allow_bits && r->b == first->b &&
                        r->apply_omc == first->apply_omc) {
                        batch.push_back(r);
                        it = pending.erase(it);
                    } else {
                        ++it;
                    }
                }
                n_requests += batch.size();
                n_batches += 1;
                largest = std::max<uint32_t>(largest, (uint32_t)batch.size());
            }
            const uint32_t q = (uint32_t)batch.size();
            uint32_t kmax = 0;
            descs.resize(q);
            for (uint32_t i = 0; i < q; ++i) {
                descs[i] = batch[i]->desc;
                kmax = std::max(kmax, descs[i].params.top_k);
            }
            const uint32_t stride = std::max(kmax, 1u);
            ids.assign((size_t)q * stride, 0);
            scores.assign((size_t)q * stride, 0.f);
            ns.assign(q, 0);
            counts.assign(q, 0);
            // one status per request: a request that is outside the envelope, or invalidated by a rebuild between its
            // validation and this dispatch, fails alone — the callers coalesced with it get their answers
            constexpr int kNoStatus = -1;  // (never a status of the ABI: "the dispatch did not reach this request")
            sts.assign(q, kNoStatus);
            const int bst =
                group ? orama_shard_post_search_batch(group, shards.data(), descs.data(), q, batch[0]->b,
                                                      reinterpret_cast<const uint64_t* const*>(batch[0]->allow), batch[0]->allow_bits,
                                                      batch[0]->apply_omc, stride, ids.data(), scores.data(), ns.data(), counts.data(),
                                                      sts.data())
                      : orama_post_search_batch_status(p, descs.data(), q, batch[0]->b, batch[0]->allow, batch[0]->allow_bits,
                                                       batch[0]->apply_omc, 8, stride, ids.data(), scores.data(), ns.data(),
                                                       counts.data(), sts.data());
            std::string err;
            if (bst != ORAMA_OK) err = orama_last_error();  // this thread's error slot: the first failing query's message
            for (uint32_t i = 0; i < q; ++i) {
                PostRequest* r = batch[i];
                // a request the dispatch never wrote a status for: the call failed before it got there (bad shared
                // arguments, the group's info call) — it shares the call's status; a call that succeeded wrote every status
                const int st = sts[i] == kNoStatus ? (bst != ORAMA_OK ? bst : ORAMA_OK) : sts[i];
                r->status = st;
                if (st != ORAMA_OK) r->error = err;
                if (st == ORAMA_OK) {
                    const uint32_t n = std::min(ns[i], r->desc.params.top_k);
                    if (n) {
                        memcpy(r->out_ids, &ids[(size_t)i * stride], (size_t)n * 8);
                        memcpy(r->out_scores, &scores[(size_t)i * stride], (size_t)n * 4);
                    }
                    *r->out_n = n;
                    if (r->out_count) *r->out_count = counts[i];
                }
                std::lock_guard<std::mutex> rl(r->m);  // notified under the lock: the request lives on the caller's stack
                r->done = true;
                r->cv.notify_one();
            }
        }
    }
};

extern "C" {

int orama_post_batcher_create(orama_post* p, uint32_t max_batch, uint32_t max_wait_us, orama_post_batcher** out) {
    ORAMA_REQUIRE(p && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(max_batch >= 1 && max_batch <= 4096, "max_batch %u outside [1, 4096]", max_batch);
    orama_post_batcher* b = new (std::nothrow) orama_post_batcher();
    if (!b) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    b->p = p;
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    for (int w = 0; w < 2; ++w) b->workers.emplace_back([b] { b->run(); });
    *out = b;
    return ORAMA_OK;
}

int orama_post_batcher_create_group(orama_shard_group* g, orama_post* const* shards, uint32_t max_batch, uint32_t max_wait_us,
                                    orama_post_batcher** out) {
    ORAMA_REQUIRE(g && shards && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(max_batch >= 1 && max_batch <= 4096, "max_batch %u outside [1, 4096]", max_batch);
    uint32_t world = 0, n_local = 0;
    ORAMA_TRY(orama_shard_group_info(g, &world, &n_local, nullptr, nullptr));
    ORAMA_SUPPORT(world == n_local, "a request batcher needs every shard of the group in this process (%u of %u are local)", n_local, world);
    for (uint32_t i = 0; i < n_local; ++i) ORAMA_REQUIRE(shards[i], "null shard %u", i);
    orama_post_batcher* b = new (std::nothrow) orama_post_batcher();
    if (!b) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    b->group = g;
    b->shards.assign(shards, shards + n_local);
    b->p = shards[0];
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    for (int w = 0; w < 2; ++w) b->workers.emplace_back([b] { b->run(); });
    *out = b;
    return ORAMA_OK;
}

void orama_post_batcher_destroy(orama_post_batcher* b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;  // pending requests are still served before the dispatchers leave
    }
    b->cv_work.notify_all();
    for (auto& t : b->workers)
        if (t.joinable()) t.join();
    delete b;
}

int orama_post_batcher_search(orama_post_batcher* b, const orama_term_ref* refs, uint32_t n_refs, float bm25_b,
                              const orama_bm25_params* params, const uint64_t* allow_bitmap, uint64_t bitmap_bits,
                              int apply_omc, uint64_t* out_ids, float* out_scores, uint32_t* out_n, uint64_t* out_count) {
    ORAMA_REQUIRE(b && params && out_n, "null argument");
    *out_n = 0;
    if (out_count) *out_count = 0;
    // a malformed request must fail alone, not with the batch it would have joined
    ORAMA_REQUIRE(n_refs == 0 || refs, "null refs");
    ORAMA_REQUIRE(params->n_tokens >= 1, "no query tokens");
    ORAMA_SUPPORT(params->n_tokens <= 64, "n_tokens %u outside [1, 64]", params->n_tokens);
    ORAMA_SUPPORT(params->top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", params->top_k, kSelectMaxK);
    ORAMA_REQUIRE(params->top_k == 0 || (out_ids && out_scores), "null output");
    uint32_t n_lists = 0;
    ORAMA_TRY(orama_post_info(b->p, nullptr, &n_lists, nullptr, nullptr));
    for (uint32_t i = 0; i < n_refs; ++i) {
        ORAMA_REQUIRE(refs[i].token < params->n_tokens, "ref %u: token %u >= n_tokens %u", i, refs[i].token, params->n_tokens);
        ORAMA_REQUIRE(refs[i].list < n_lists, "ref %u: list %u out of range", i, refs[i].list);
    }
    PostRequest r;
    r.desc.refs = refs;
    r.desc.n_refs = n_refs;
    r.desc.params = *params;
    r.b = bm25_b;
    r.allow = allow_bitmap;
    r.allow_bits = allow_bitmap ? bitmap_bits : 0;
    r.apply_omc = apply_omc ? 1 : 0;
    r.out_ids = out_ids;
    r.out_scores = out_scores;
    r.out_n = out_n;
    r.out_count = out_count;
    {
        std::unique_lock<std::mutex> lk(b->mu);
        ORAMA_REQUIRE(!b->stop, "batcher is shutting down");
        b->pending.push_back(&r);
        b->cv_work.notify_one();
    }
    {
        std::unique_lock<std::mutex> rl(r.m);
        r.cv.wait(rl, [&] { return r.done; });
    }
    if (r.status != ORAMA_OK) set_error("%s", r.error.c_str());
    return r.status;
}

int orama_post_batcher_stats(orama_post_batcher* b, uint64_t* requests, uint64_t* batches, uint32_t* largest_batch) {
    ORAMA_REQUIRE(b, "null handle");
    std::lock_guard<std::mutex> lk(b->mu);
    if (requests) *requests = b->n_requests;
    if (batches) *batches = b->n_batches;
    if (largest_batch) *largest_batch = b->largest;
    return ORAMA_OK;
}

}  // extern "C"
