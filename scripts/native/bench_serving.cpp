// bench_serving.cpp — native load generator for the request batchers: T caller threads issue single-query requests
// through the C ABI, the way the reference's tokio workers call search_full_text / EmbeddingFieldStorage::search
// (src/collection_manager/sides/read/collection.rs:846-884).  Python threads cannot generate this load (the GIL
// caps them near 20 K requests/s), hence a C++ driver.
//   bench_serving bm25 [docs=10000000] [requests_per_thread=400] [threads=1,8,32,128]
//   bench_serving hybrid [docs=10000000] [requests_per_thread=30] [threads=1,8,32,64] [f32|shadow]   (fp32 vectors + BM25F,
//                      min-max merge; shadow = the vector store also keeps an fp16 copy: two-stage exact vector leg)
//   bench_serving vec  [rows=10000000] [requests_per_thread=100] [threads=8,64,256,512] [f16|f32|shadow] [shards=1]   (768 dims, top-100;
//                      shards > 1: the rows split over a co-located shard GROUP on this GPU — orama_shard_vec_search from
//                      every caller thread / orama_batcher_create_group — against the single store of shards = 1;
//                      shadow = fp32 rows + fp16 shadow, exact fp32 answers by the two-stage plan)
// Build: g++ -O2 -std=c++17 -I include scripts/native/bench_serving.cpp -L oramacore_amd/csrc -lorama_hip -pthread
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "orama_hip.h"

#define CHECK(x)                                                                     \
    do {                                                                             \
        if ((x) != ORAMA_OK) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #x, orama_last_error());              \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

static std::vector<int> parse_list(const char* s) {
    std::vector<int> v;
    for (const char* p = s; *p;) {
        v.push_back(atoi(p));
        while (*p && *p != ',') ++p;
        if (*p) ++p;
    }
    return v;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "bm25";
    const uint64_t n_docs = argc > 2 ? strtoull(argv[2], nullptr, 10) : 10000000ull;
    const int per_thread = argc > 3 ? atoi(argv[3]) : 400;
    const std::vector<int> thread_counts = parse_list(argc > 4 ? argv[4] : "1,8,32,128");
    orama_ctx* ctx = nullptr;
    CHECK(orama_ctx_create(0, &ctx));
    if (mode == "vec") {
        const uint32_t dim = 768, K = 100, NQ = 512;
        orama_vec* vec = nullptr;
        const std::string dts = argc > 5 ? argv[5] : "f16";
        const int dt = dts == "f32" ? ORAMA_DTYPE_F32 : dts == "shadow" ? ORAMA_DTYPE_F32_SHADOW16 : ORAMA_DTYPE_F16;
        const int n_shards = argc > 6 ? std::max(1, atoi(argv[6])) : 1;
        orama_shard_group* group = nullptr;
        std::vector<orama_vec*> shards;
        if (n_shards > 1) {
            std::vector<int> devs((size_t)n_shards, 0);
            CHECK(orama_shard_group_create(devs.data(), (uint32_t)n_shards, 0, &group));
            const uint64_t per = n_docs / (uint64_t)n_shards;
            for (int s = 0; s < n_shards; ++s) {
                orama_vec* sv = nullptr;
                const uint64_t rows = s == n_shards - 1 ? n_docs - per * (uint64_t)(n_shards - 1) : per;
                CHECK(orama_vec_create(orama_shard_group_ctx(group, (uint32_t)s), dim, ORAMA_METRIC_COSINE, dt, rows, &sv));
                CHECK(orama_vec_fill_synthetic(sv, rows, 0x5EED + (uint64_t)s, per * (uint64_t)s));
                shards.push_back(sv);
            }
            vec = shards[0];
        } else {
            CHECK(orama_vec_create(ctx, dim, ORAMA_METRIC_COSINE, dt, n_docs, &vec));
            CHECK(orama_vec_fill_synthetic(vec, n_docs, 0x5EED, 0));
        }
        std::mt19937_64 rng(0xBEEF);
        std::normal_distribution<float> g(0.f, 1.f);
        std::vector<float> qv((size_t)NQ * dim);
        for (auto& x : qv) x = g(rng);
        printf("vec: %llu x %u rows (%s store%s), top-%u, single-query requests\n", (unsigned long long)n_docs, dim, dts.c_str(),
               n_shards > 1 ? (", " + std::to_string(n_shards) + " co-located shards in one group").c_str() : "", K);
        for (int batched = 0; batched < 2; ++batched) {
            for (int nt : thread_counts) {
                if (!batched && nt > 64) continue;  // direct calls: one corpus pass per request
                orama_batcher* batcher = nullptr;
                if (batched && group) CHECK(orama_batcher_create_group(group, shards.data(), dt == ORAMA_DTYPE_F32 ? 64 : 256, 0, &batcher));
                else if (batched) CHECK(orama_batcher_create(vec, dt == ORAMA_DTYPE_F32 ? 64 : 256, 0, &batcher));
                const int count = batched ? per_thread : std::max(4, per_thread / 8);
                std::atomic<uint64_t> checksum{0};
                auto worker = [&](int tid, int cnt) {
                    std::vector<uint64_t> ids(K);
                    std::vector<float> dist(K);
                    uint64_t acc = 0;
                    for (int i = 0; i < cnt; ++i) {
                        const float* q = &qv[(size_t)((tid * 131 + i) % NQ) * dim];
                        uint32_t n = 0;
                        if (batched) CHECK(orama_batcher_search(batcher, q, K, ids.data(), dist.data(), &n));
                        else if (group) CHECK(orama_shard_vec_search(group, shards.data(), q, 1, K, nullptr, 0, ids.data(), dist.data(), &n));
                        else CHECK(orama_vec_search(vec, q, 1, K, nullptr, 0, ids.data(), dist.data(), &n));
                        acc += ids[0];
                    }
                    checksum += acc;
                };
                worker(0, 3);
                checksum = 0;
                std::vector<std::thread> ths;
                const auto t0 = std::chrono::steady_clock::now();
                for (int t = 0; t < nt; ++t) ths.emplace_back(worker, t, count);
                for (auto& t : ths) t.join();
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                uint64_t req = 0, bat = 0;
                uint32_t largest = 0;
                if (batched) {
                    CHECK(orama_batcher_stats(batcher, &req, &bat, &largest));
                    orama_batcher_destroy(batcher);
                }
                printf("%-28s %4d caller threads: %9.0f requests/s",
                       batched ? "orama_batcher_search" : (group ? "orama_shard_vec_search" : "orama_vec_search (direct)"), nt,
                       (double)nt * count / el);
                if (batched) printf("   (mean batch %.1f, largest %u)", bat ? (double)req / (double)bat : 0.0, largest);
                printf("   checksum %llu\n", (unsigned long long)checksum.load());
                fflush(stdout);
            }
        }
        if (group) {
            uint32_t max_lanes = 0, lanes = 0;
            CHECK(orama_shard_group_lanes(group, &max_lanes, &lanes));
            printf("group: %u of at most %u lanes made\n", lanes, max_lanes);
            for (orama_vec* sv : shards) orama_vec_destroy(sv);
            orama_shard_group_destroy(group);
        } else {
            orama_vec_destroy(vec);
        }
        orama_ctx_destroy(ctx);
        return 0;
    }
    const bool hybrid = mode == "hybrid";
    if (mode != "bm25" && !hybrid) {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    orama_post* post = nullptr;
    CHECK(orama_post_create(ctx, &post));
    // the C4 full-text shape: 2048 posting lists with Zipf ranks log-uniform in [100, 100000], 12 tokens per query
    std::mt19937_64 rng(0xB26);
    std::uniform_real_distribution<double> u(std::log(100.0), std::log(100000.0));
    std::vector<uint32_t> ranks;
    for (int i = 0; i < 2048; ++i) ranks.push_back((uint32_t)std::exp(u(rng)));
    std::sort(ranks.begin(), ranks.end());
    ranks.erase(std::unique(ranks.begin(), ranks.end()), ranks.end());
    uint64_t total = 0;
    CHECK(orama_post_fill_synthetic(post, n_docs, 0, (uint32_t)ranks.size(), ranks.data(), 0xB25, &total));
    const uint32_t T = 12, K = 100, NQ = 512;
    std::vector<std::vector<orama_term_ref>> queries(NQ);
    for (auto& q : queries) {
        std::vector<uint32_t> pick;
        while (pick.size() < T) {
            const uint32_t l = (uint32_t)(rng() % ranks.size());
            if (std::find(pick.begin(), pick.end(), l) == pick.end()) pick.push_back(l);
        }
        for (uint32_t t = 0; t < T; ++t) q.push_back(orama_term_ref{t, pick[t], 1.0f});
    }
    orama_bm25_params params;
    memset(&params, 0, sizeof(params));
    params.total_documents = (float)n_docs;
    params.n_tokens = T;
    params.top_k = K;
    params.k = 1.2f;
    printf("bm25: %llu docs, %zu lists, %llu postings resident, %u tokens per query, top-%u\n", (unsigned long long)n_docs,
           ranks.size(), (unsigned long long)total, T, K);

    if (hybrid) {
        const uint32_t dim = 768;
        orama_vec* vec = nullptr;
        const bool shadow = argc > 5 && std::string(argv[5]) == "shadow";
        CHECK(orama_vec_create(ctx, dim, ORAMA_METRIC_COSINE, shadow ? ORAMA_DTYPE_F32_SHADOW16 : ORAMA_DTYPE_F32, n_docs, &vec));
        CHECK(orama_vec_fill_synthetic(vec, n_docs, 0x5EED, 0));
        std::normal_distribution<float> g(0.f, 1.f);
        std::vector<float> qv((size_t)NQ * dim);
        for (auto& x : qv) x = g(rng);
        printf("%s", shadow ? "vector store: fp32 rows + fp16 shadow (two-stage exact)\n" : "vector store: fp32 rows\n");
        printf("hybrid: + %llu x %u fp32 rows; a request = vector top-%u + BM25F + min-max merge + top-%u\n", (unsigned long long)n_docs, dim, K, K);
        for (int batched = 0; batched < 2; ++batched) {
            for (int nt : thread_counts) {
                orama_batcher* vb = nullptr;
                // plain fp32: a corpus pass proposes for <= 64 queries (K1x); with a shadow the fp16 scan takes 256
                if (batched) CHECK(orama_batcher_create(vec, shadow ? 256 : 64, 0, &vb));
                std::atomic<uint64_t> checksum{0};
                auto worker = [&](int tid, int cnt) {
                    std::vector<uint64_t> ids(K), vid(K);
                    std::vector<float> sc(K), vd(K);
                    uint64_t acc = 0;
                    for (int i = 0; i < cnt; ++i) {
                        const size_t qi = (size_t)(tid * 131 + i) % NQ;
                        const auto& q = queries[qi];
                        uint32_t n = 0;
                        uint64_t c = 0;
                        if (!batched) {
                            CHECK(orama_hybrid_search(vec, post, &qv[qi * dim], K, 0.0f, 0, q.data(), T, 0.75f, &params, nullptr, 0, 1,
                                                      ids.data(), sc.data(), &n, &c));
                        } else {
                            uint32_t nv = 0;
                            CHECK(orama_batcher_search(vb, &qv[qi * dim], K, vid.data(), vd.data(), &nv));
                            for (uint32_t j = 0; j < nv; ++j) vd[j] = 1.0f - vd[j];  // a2 epilogue: one row per document here
                            CHECK(orama_post_search_hybrid(post, q.data(), T, 0.75f, &params, nullptr, 0, vid.data(), vd.data(), nv, 1,
                                                           ids.data(), sc.data(), &n, &c));
                        }
                        acc += ids[0] + c;
                    }
                    checksum += acc;
                };
                worker(0, 3);
                checksum = 0;
                std::vector<std::thread> ths;
                const auto t0 = std::chrono::steady_clock::now();
                for (int t = 0; t < nt; ++t) ths.emplace_back(worker, t, per_thread);
                for (auto& t : ths) t.join();
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                uint64_t req = 0, bat = 0;
                uint32_t largest = 0;
                if (batched) {
                    CHECK(orama_batcher_stats(vb, &req, &bat, &largest));
                    orama_batcher_destroy(vb);
                }
                printf("%-46s %4d caller threads: %8.0f requests/s", batched ? "vector batcher + orama_post_search_hybrid" : "orama_hybrid_search (one call per request)",
                       nt, (double)nt * per_thread / el);
                if (batched) printf("   (mean vector batch %.1f)", bat ? (double)req / (double)bat : 0.0);
                printf("   checksum %llu\n", (unsigned long long)checksum.load());
                fflush(stdout);
            }
        }
        orama_vec_destroy(vec);
        orama_post_destroy(post);
        orama_ctx_destroy(ctx);
        return 0;
    }
    for (int batched = 0; batched < 2; ++batched) {
        for (int nt : thread_counts) {
            orama_post_batcher* batcher = nullptr;
            if (batched) CHECK(orama_post_batcher_create(post, 256, 0, &batcher));
            std::atomic<uint64_t> checksum{0};
            auto worker = [&](int tid, int count) {
                std::vector<uint64_t> ids(K);
                std::vector<float> sc(K);
                uint64_t acc = 0;
                for (int i = 0; i < count; ++i) {
                    const auto& q = queries[(size_t)(tid * 131 + i) % NQ];
                    uint32_t n = 0;
                    uint64_t cnt = 0;
                    if (batched) CHECK(orama_post_batcher_search(batcher, q.data(), T, 0.75f, &params, nullptr, 0, 1, ids.data(), sc.data(), &n, &cnt));
                    else CHECK(orama_post_search(post, q.data(), T, 0.75f, &params, nullptr, 0, 1, ids.data(), sc.data(), &n, &cnt));
                    acc += ids[0] + cnt;
                }
                checksum += acc;
            };
            worker(0, 20);  // warm-up
            const uint64_t warm = checksum.exchange(0);
            (void)warm;
            std::vector<std::thread> ths;
            const auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < nt; ++t) ths.emplace_back(worker, t, per_thread);
            for (auto& t : ths) t.join();
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t req = 0, bat = 0;
            uint32_t largest = 0;
            if (batched) {
                CHECK(orama_post_batcher_stats(batcher, &req, &bat, &largest));
                orama_post_batcher_destroy(batcher);
            }
            printf("%-28s %4d caller threads: %9.0f requests/s", batched ? "orama_post_batcher_search" : "orama_post_search (direct)", nt,
                   (double)nt * per_thread / el);
            if (batched) printf("   (mean batch %.1f, largest %u)", bat ? (double)req / (double)bat : 0.0, largest);
            printf("   checksum %llu\n", (unsigned long long)checksum.load());
            fflush(stdout);
        }
    }
    orama_post_destroy(post);
    orama_ctx_destroy(ctx);
    return 0;
}
