// context.hip — error channel, per-GPU context, scratch pool, HIP-event profiler.
#include <algorithm>
#include <cstdlib>
#include <memory>

#include <chrono>

#include "common.hpp"
#include "vec_kernels.hpp"

namespace orama {

static thread_local char g_err[1024] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = 0; }

// ---------------------------------------------------------------- profiler
hipEvent_t Profiler::get_event() {
    if (!free_events.empty()) {
        hipEvent_t e = free_events.back();
        free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void Profiler::begin(const char*, hipStream_t s, hipEvent_t* start) {
    std::lock_guard<std::mutex> g(mu);
    hipEvent_t e = get_event();
    if (e && hipEventRecord(e, s) == hipSuccess) *start = e;
}

void Profiler::end(const char* name, hipStream_t s, hipEvent_t start) {
    std::lock_guard<std::mutex> g(mu);
    hipEvent_t e = get_event();
    if (!e) return;
    (void)hipEventRecord(e, s);
    acc[name].pending.emplace_back(start, e);
}

void Profiler::pair(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> g(mu);
    hipEvent_t a = get_event(), b = get_event();
    if (!a || !b) {
        if (a) free_events.push_back(a);
        if (b) free_events.push_back(b);
        return;
    }
    *start = a;
    *stop = b;
    acc[name].pending.emplace_back(a, b);
}

int Profiler::resolve() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& kv : acc) {
        for (auto& pr : kv.second.pending) {
            (void)hipEventSynchronize(pr.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.ms += (double)ms;
                kv.second.n += 1;
                if (kv.second.samples.size() >= 65536) kv.second.samples.erase(kv.second.samples.begin(), kv.second.samples.begin() + 32768);
                kv.second.samples.push_back(ms);
            }
            free_events.push_back(pr.first);
            free_events.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
    return ORAMA_OK;
}

void Profiler::reset() {
    resolve();
    std::lock_guard<std::mutex> g(mu);
    for (auto& kv : acc) {
        kv.second.ms = 0.0;
        kv.second.n = 0;
        kv.second.samples.clear();
    }
}

Profiler::~Profiler() {
    for (auto& kv : acc)
        for (auto& pr : kv.second.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (hipEvent_t e : free_events) (void)hipEventDestroy(e);
}

}  // namespace orama

namespace {
int take_one(orama_ctx* c, std::unique_ptr<orama::Scratch>* out, int kind) {  // pool_mu held
    for (size_t i = c->pool.size(); i-- > 0;) {  // most recently returned set of this kind: its buffers already fit
        if (c->pool[i]->kind != kind) continue;
        *out = std::move(c->pool[i]);
        c->pool.erase(c->pool.begin() + (long)i);
        return ORAMA_OK;
    }
    std::unique_ptr<orama::Scratch> s(new orama::Scratch());
    s->kind = kind;
    ORAMA_HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    *out = std::move(s);
    return ORAMA_OK;
}
}  // namespace

// The bound on sets in flight is served FIRST COME, FIRST SERVED.  Every waiter sleeps on its own condition variable in a
// queue; whoever returns capacity (release, detach_one, a waiter that timed out) grants it to the head of the queue for as
// long as the head's request fits, and wakes exactly the granted callers.  (Rounds 1-2 woke "one waiter" of a shared
// condition variable and let multi-set callers go before single-set ones: with 512 callers of the one-call hybrid search
// — three sets each, ten calls in flight — the wake-up order starved some callers until their 30 s timeout while the pool
// was turning over 450 times a second, and a single-set caller could not run at all while any multi-set caller waited.)
// A request for several sets is granted as a whole, so callers never hold some sets while waiting for others.
void orama_ctx::grant_waiters() {
    while (!pool_waiters.empty() && leased + pool_waiters.front()->need <= max_inflight) {
        PoolWaiter* w = pool_waiters.front();
        pool_waiters.pop_front();
        leased += w->need;  // reserved for it: it takes its sets when it wakes
        w->granted = true;
        w->cv.notify_one();
    }
}

int orama_ctx::acquire(std::unique_ptr<orama::Scratch>* out, int kind) {
    std::unique_ptr<orama::Scratch>* outs[1] = {out};
    return acquire_n(1, outs, &kind);
}

int orama_ctx::acquire_n(uint32_t n, std::unique_ptr<orama::Scratch>** outs, const int* kinds) {
    std::unique_lock<std::mutex> g(pool_mu);
    const uint32_t need = std::min(n, max_inflight);  // a pool smaller than the request would wait for ever
    if (pool_waiters.empty() && leased + need <= max_inflight) {
        leased += need;
    } else {
        PoolWaiter w;
        w.need = need;
        pool_waiters.push_back(&w);
        const bool ok = w.cv.wait_for(g, std::chrono::milliseconds(acquire_timeout_ms), [&] { return w.granted; });
        if (!ok) {
            for (auto it = pool_waiters.begin(); it != pool_waiters.end(); ++it)
                if (*it == &w) {
                    pool_waiters.erase(it);
                    break;
                }
            grant_waiters();  // (the queue's head may have been this caller: what is free may fit the next one)
            orama::set_error("%u scratch set(s) did not come free within %u ms (%u of %u leased, %zu callers waiting; "
                             "ORAMA_MAX_INFLIGHT / ORAMA_ACQUIRE_TIMEOUT_MS): the call was not started", n, acquire_timeout_ms,
                             leased, max_inflight, pool_waiters.size());
            return ORAMA_ERR_BUSY;
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        const int st = take_one(this, outs[i], kinds ? kinds[i] : orama::kScratchGeneral);
        if (st != ORAMA_OK) {
            for (uint32_t j = 0; j < i; ++j) pool.push_back(std::move(*outs[j]));
            leased -= need;
            grant_waiters();
            return st;
        }
    }
    leased += n - need;
    return ORAMA_OK;
}

int orama_ctx::acquire2(std::unique_ptr<orama::Scratch>* a, std::unique_ptr<orama::Scratch>* b) {
    std::unique_ptr<orama::Scratch>* outs[2] = {a, b};
    return acquire_n(2, outs);
}

namespace {
// HBM held by one scratch set (the buffers that scale with the corpus / the index)
size_t scratch_bytes(const orama::Scratch& s) {
    size_t b = 0;
    for (const orama::DevBuf* d : {&s.query, &s.dist, &s.sel_state, &s.sel_keys, &s.out_idx, &s.out_val, &s.out_n, &s.out_ids,
                                   &s.bitmap, &s.f16_bfrag, &s.misc0, &s.misc1, &s.misc2, &s.misc3, &s.misc4, &s.misc5,
                                   &s.bm25_acc, &s.bm25_emit})
        b += d->cap;
    return b;
}
size_t scratch_pool_budget() {
    static const size_t v = [] {
        const char* e = std::getenv("ORAMA_SCRATCH_POOL_MIB");
        return (e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)65536) << 20;
    }();
    return v;
}
}  // namespace

// Scratch sets are pooled so that a search allocates nothing in steady state — but a set sized by a large index
// (the BM25 accumulator alone is slots x n_docs x 8 B: 1.3 GB at 10 M docs and 12 tokens) stays that large.  Many
// concurrent callers (the reference searches from every tokio worker) would otherwise pin peak-concurrency x that
// much HBM forever.  Two bounds: at most `max_inflight` sets are out at any time (ORAMA_MAX_INFLIGHT, default 32 — further
// callers wait their turn, which also bounds the peak), and when the idle pool exceeds its byte budget
// (ORAMA_SCRATCH_POOL_MIB, default 64 GiB of the 288) the set being returned gives its large buffers back to the driver
// and keeps only its stream and small buffers.  (A budget below what the steady concurrency needs makes every query
// re-allocate and re-zero its accumulator: 10 K -> 0.5 K queries/s at 8 threads with an 8 GiB budget.)
void orama_ctx::detach_one() {
    std::lock_guard<std::mutex> g(pool_mu);
    if (leased) --leased;
    ++held;
    grant_waiters();
}

void orama_ctx::release(std::unique_ptr<orama::Scratch> s, bool detached) {
    std::lock_guard<std::mutex> g(pool_mu);
    size_t pooled = 0;
    for (const auto& q : pool) pooled += scratch_bytes(*q);
    if (pooled + scratch_bytes(*s) > scratch_pool_budget()) {
        ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        for (orama::DevBuf* d : {&s->dist, &s->sel_keys, &s->out_ids, &s->bitmap, &s->misc0, &s->misc1, &s->misc2, &s->misc3,
                                 &s->misc4, &s->misc5, &s->bm25_acc, &s->bm25_emit})
            if (d->cap > ((size_t)16 << 20)) d->release();
    }
    pool.push_back(std::move(s));
    if (detached) {
        if (held) --held;
    } else if (leased) {
        --leased;
    }
    grant_waiters();
}

namespace orama {
uint64_t allow_content_version(orama_ctx* ctx, const uint64_t* allow_bitmap) {
    if (!allow_bitmap) return 0;
    std::lock_guard<std::mutex> g(ctx->allow_mu);
    auto v = ctx->allow_version.find(allow_bitmap);
    return v != ctx->allow_version.end() ? v->second : 0;
}

int resolve_allow(orama_ctx* ctx, Scratch* sc, const uint64_t* allow_bitmap, uint64_t bitmap_bits, hipStream_t s,
                  const uint64_t** d_allow, uint64_t* version) {
    *d_allow = nullptr;
    if (version) *version = 0;
    if (!allow_bitmap) return ORAMA_OK;
    {
        std::lock_guard<std::mutex> g(ctx->allow_mu);
        auto it = ctx->allow_reg.find(allow_bitmap);
        if (it != ctx->allow_reg.end()) {
            ORAMA_REQUIRE(bitmap_bits <= it->second, "resident bitmap holds %llu bits, %llu requested",
                          (unsigned long long)it->second, (unsigned long long)bitmap_bits);
            *d_allow = allow_bitmap;
            if (version) {
                auto v = ctx->allow_version.find(allow_bitmap);
                *version = v != ctx->allow_version.end() ? v->second : 0;
            }
            return ORAMA_OK;
        }
    }
    const size_t words = (size_t)((bitmap_bits + 63) / 64);
    ORAMA_TRY(sc->bitmap.reserve(std::max<size_t>(8, words * 8)));
    if (words) ORAMA_HIP_TRY(hipMemcpyAsync(sc->bitmap.p, allow_bitmap, words * 8, hipMemcpyHostToDevice, s));
    *d_allow = sc->bitmap.as<uint64_t>();
    return ORAMA_OK;
}
}  // namespace orama

namespace {
__global__ void allow_set_kernel(unsigned long long* __restrict__ words, const uint64_t* __restrict__ ids, uint64_t n, bool allowed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t id = ids[i];
        const unsigned long long bit = 1ull << (id & 63);
        if (allowed) atomicOr(&words[id >> 6], bit);
        else atomicAnd(&words[id >> 6], ~bit);
    }
}
}  // namespace

// Resident allow-bitmap: the materialised FilterResult<DocumentId> kept in HBM.
struct orama_allow {
    orama_ctx* ctx = nullptr;
    orama::DevBuf words;
    uint64_t bits = 0;
};

namespace {
// Options of orama_ctx_set_option: what tests and tuning scripts switch on a context (none is needed by a deployment).
struct CtxOption {
    const char* name;
    long long lo, hi;
    int (*set)(orama_ctx*, long long);
};
#define ORAMA_OPT(NAME, LO, HI, STMT) \
    CtxOption { NAME, LO, HI, [](orama_ctx* c, long long v) -> int { STMT; return ORAMA_OK; } }
const CtxOption kCtxOptions[] = {
    ORAMA_OPT("fused_topk", 0, 2, c->fused_topk = (int)v),                  // K1 per-wave top-k: 0 never, 1 always, 2 = the rule
    ORAMA_OPT("f32_multi", 0, 1, c->f32_multi = (int)v),                    // K1b for 2..8 fp32 queries
    ORAMA_OPT("f32_batch_cvt", 0, 1, c->f32_batch_cvt = v != 0),            // fp32 batches: K1x proposes (1) or K1m (0)
    ORAMA_OPT("f16_solo", 0, 2, c->f16_solo = (int)v),                      // K1h for shadow scans of <= 4 queries
    ORAMA_OPT("f16_wide", 0, 5, c->f16_wide = (int)v),                      // (orama_ctx_set_f16_wide validates against the build)
    ORAMA_OPT("f16_kc", 8, 16, c->f16_kc = (int)v),
    ORAMA_OPT("f16_nbuf", 2, 4, c->f16_nbuf = (int)v),
    ORAMA_OPT("bm25_ranges", 0, 1, c->bm25_ranges = (int)v),
    ORAMA_OPT("f16_head_rows", 0, 1ll << 24, c->f16_head_rows = (uint64_t)v),
    ORAMA_OPT("f16_cand_mib", 0, 1ll << 20, c->f16_cand_mib = (uint64_t)v),
    ORAMA_OPT("f16_chunk_grow", -1, 1, c->f16_chunk_grow = (int)v),
    ORAMA_OPT("f16_grow_factor", 2, 64, c->f16_grow_factor = (int)v),
    ORAMA_OPT("two_stage_spare", 1, 4096, c->two_stage_spare = (uint32_t)v),
    ORAMA_OPT("k3r_target", 0, 4096, c->k3r_target = (uint32_t)v),  // (0 = default; above what a workgroup holds: the default)
    ORAMA_OPT("bm25_ranges_hybrid", 0, 1, c->bm25_ranges_hybrid = v != 0),
    // K3r key lists: 0 = one slot per posting (round 4), 1 = compact lists for batches of >= 8 queries (default), 2 = always
    ORAMA_OPT("k3r_compact", 0, 2, (c->bm25_compact_keys = v != 0, c->bm25_compact_min = v == 2 ? 1u : 8u)),
    ORAMA_OPT("select_wide", 0, 3, c->select_wide = (int)v),                // K4 over a lone query's distances (select.hip)
    ORAMA_OPT("select_pairs", 0, 1, c->select_pairs = v != 0),
#if ORAMA_COMPARISON_KERNELS
    ORAMA_OPT("bm25_dense_acc", 0, 1, c->bm25_dense_acc = v != 0),           // dense-list bitmaps for postings stores built from now on (<= +6 B / posting; read by the comparison unit only)
    ORAMA_OPT("k3r_fast", 0, 1, c->k3r_fast = v != 0),                      // plain top-k batches scored by bm25_ranges_fast.hip (the r06 experiment)
    ORAMA_OPT("hybrid_device_tail", 0, 1, c->hybrid_device_tail = v != 0),  // orama_hybrid_search finishes on the device (hybrid_tail.hip)
#else
    CtxOption{"hybrid_device_tail", 0, 1, [](orama_ctx*, long long v) -> int {  // a comparison unit: not in this library
                  if (v == 0) return ORAMA_OK;
                  orama::set_error("hybrid_device_tail: hybrid_tail.hip is built into liborama_hip_cmp.so only (ORAMA_COMPARISON_KERNELS=1)");
                  return ORAMA_ERR_UNSUPPORTED;
              }},
#endif
    ORAMA_OPT("direct_out", 0, 1, c->direct_out = v != 0),
    ORAMA_OPT("stage_copy", 0, 1, c->stage_by_kernel = v != 0),             // 1 = small blocks by kernel, 0 = SDMA copies
    ORAMA_OPT("scan_done_event", 0, 1, c->scan_done_on_dispatch = v != 0),  // 1 = the event rides on the scan's dispatch
};
#undef ORAMA_OPT
}  // namespace

extern "C" {

int orama_allow_create(orama_ctx* ctx, const uint64_t* words, uint64_t bitmap_bits, orama_allow** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(bitmap_bits == 0 || words, "null words");
    ORAMA_ON_DEVICE(ctx->device);
    std::unique_ptr<orama_allow> a(new (std::nothrow) orama_allow());
    if (!a) {
        orama::set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    a->ctx = ctx;
    a->bits = bitmap_bits;
    const size_t n = (size_t)((bitmap_bits + 63) / 64);
    ORAMA_TRY(a->words.reserve(std::max<size_t>(8, n * 8)));
    if (n) ORAMA_HIP_TRY(hipMemcpy(a->words.p, words, n * 8, hipMemcpyHostToDevice));
    {
        std::lock_guard<std::mutex> g(ctx->allow_mu);
        ctx->allow_reg[a->words.p] = bitmap_bits;
        ctx->allow_version[a->words.p] = ctx->allow_next_version++;
    }
    *out = a.release();
    return ORAMA_OK;
}

void orama_allow_destroy(orama_allow* a) {
    if (!a) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(a->ctx->device);
    (void)hipDeviceSynchronize();  // searches still reading the bitmap
    {
        std::lock_guard<std::mutex> g(a->ctx->allow_mu);
        a->ctx->allow_reg.erase(a->words.p);
        a->ctx->allow_version.erase(a->words.p);
    }
    delete a;
}

const uint64_t* orama_allow_token(const orama_allow* a) {
    return a ? static_cast<const uint64_t*>(a->words.p) : nullptr;
}

int orama_allow_set(orama_allow* a, const uint64_t* doc_ids, uint64_t n, int allowed) {
    ORAMA_REQUIRE(a && (n == 0 || doc_ids), "null argument");
    if (n == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(a->ctx->device);
    for (uint64_t i = 0; i < n; ++i)
        ORAMA_REQUIRE(doc_ids[i] < a->bits, "doc id %llu outside the bitmap (%llu bits)",
                      (unsigned long long)doc_ids[i], (unsigned long long)a->bits);
    // one upload + one kernel of atomic bit updates: searches reading the bitmap meanwhile see each document either
    // way, as a reader of the reference's filter sees a delete before or after it (index/filter.rs:344-392)
    orama::ScratchLease sc(a->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    ORAMA_TRY(sc->h_in.reserve((size_t)n * 8));
    ORAMA_TRY(sc->misc0.reserve((size_t)n * 8));
    memcpy(sc->h_in.p, doc_ids, (size_t)n * 8);
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, sc->h_in.p, (size_t)n * 8, hipMemcpyHostToDevice, s));
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(allow_set_kernel, dim3(blocks), dim3(256), 0, s, a->words.as<unsigned long long>(),
                       sc->misc0.as<uint64_t>(), n, allowed != 0);
    ORAMA_HIP_TRY(hipGetLastError());
    {  // a new content: nothing counted under the old one applies (before the bits move AND after: a search that resolved the
       // bitmap in between remembers its counts under a version nobody will ask for again)
        std::lock_guard<std::mutex> g(a->ctx->allow_mu);
        a->ctx->allow_version[a->words.p] = a->ctx->allow_next_version++;
    }
    const hipError_t e = hipStreamSynchronize(s);
    {
        std::lock_guard<std::mutex> g(a->ctx->allow_mu);
        a->ctx->allow_version[a->words.p] = a->ctx->allow_next_version++;
    }
    ORAMA_HIP_TRY(e);
    return ORAMA_OK;
}

int orama_abi_version(void) { return ORAMA_ABI_VERSION; }

const char* orama_last_error(void) { return orama::g_err; }

int orama_device_count(int* out) {
    ORAMA_REQUIRE(out != nullptr, "orama_device_count: out is NULL");
    *out = 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        orama::set_error("no HIP device available (%s) — liborama_hip has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return ORAMA_ERR_HIP;
    }
    *out = count;
    return ORAMA_OK;
}

int orama_ctx_create(int device_ordinal, orama_ctx** out) {
    ORAMA_REQUIRE(out != nullptr, "orama_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        orama::set_error("no HIP device available (%s) — liborama_hip has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return ORAMA_ERR_HIP;
    }
    ORAMA_REQUIRE(device_ordinal >= 0 && device_ordinal < count, "device %d out of range [0,%d)",
                  device_ordinal, count);
    ORAMA_ON_DEVICE(device_ordinal);
    hipDeviceProp_t prop;
    ORAMA_HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    orama_ctx* c = new (std::nothrow) orama_ctx();
    if (!c) {
        orama::set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    c->scan_tuning = orama::default_scan_tuning();
    // deployment knobs (the product library's whole environment, with ORAMA_RCCL_LIB / ORAMA_SHARD_LANES in shard_group.hip,
    // ORAMA_SCRATCH_POOL_MIB above and ORAMA_VMM in vec_store.hip)
    if (const char* e = std::getenv("ORAMA_TWO_STAGE")) c->two_stage = std::max(0, std::min(2, std::atoi(e)));
    if (const char* e = std::getenv("ORAMA_MAX_INFLIGHT")) c->max_inflight = (uint32_t)std::max(2, std::atoi(e));
    if (const char* e = std::getenv("ORAMA_ACQUIRE_TIMEOUT_MS")) c->acquire_timeout_ms = (uint32_t)std::max(1, std::atoi(e));
#if ORAMA_COMPARISON_KERNELS
    // comparison builds: every option of orama_ctx_set_option can also come from the environment as ORAMA_<NAME> (the A/B
    // scripts of rounds 1-5 keep working against liborama_hip_cmp.so)
    for (const CtxOption& o : kCtxOptions) {
        std::string env = "ORAMA_";
        for (const char* p = o.name; *p; ++p) env.push_back((char)std::toupper((unsigned char)*p));
        if (const char* e = orama::dev_env(env.c_str())) {
            long long v = 0;
            if (std::strcmp(o.name, "scan_done_event") == 0) v = std::strcmp(e, "record") != 0;
            else if (std::strcmp(o.name, "stage_copy") == 0) v = std::strcmp(e, "dma") != 0;
            else v = std::atoll(e);
            (void)o.set(c, v);
        }
    }
    if (const char* e = orama::dev_env("ORAMA_K3R_COMPACT")) c->bm25_compact_keys = std::atoi(e) != 0;
    if (const char* e = orama::dev_env("ORAMA_K3R_MERGE")) c->k3r_merge = std::atoi(e) != 0;
#else
    if (c->f16_wide == 1 || c->f16_wide == 5) c->f16_wide = 4;  // K2c / K2h are not in this build
#endif
    c->device = device_ordinal;
    c->compute_units = prop.multiProcessorCount;
    c->hbm_bytes = (uint64_t)prop.totalGlobalMem;
    // (some driver stacks leave the marketing name empty: say what IS known rather than print " (gfx950…)")
    snprintf(c->name, sizeof(c->name), "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD GPU, marketing name not reported",
             prop.gcnArchName, prop.multiProcessorCount);
    *out = c;
    return ORAMA_OK;
}

void orama_ctx_destroy(orama_ctx* ctx) {
    if (!ctx) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

int orama_ctx_synchronize(orama_ctx* ctx) {
    ORAMA_REQUIRE(ctx, "null ctx");
    ORAMA_ON_DEVICE(ctx->device);
    ORAMA_HIP_TRY(hipDeviceSynchronize());
    return ORAMA_OK;
}

int orama_ctx_device_info(orama_ctx* ctx, char* name256, int* compute_units, uint64_t* hbm_bytes) {
    ORAMA_REQUIRE(ctx, "null ctx");
    if (name256) {
        strncpy(name256, ctx->name, 255);
        name256[255] = 0;
    }
    if (compute_units) *compute_units = ctx->compute_units;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return ORAMA_OK;
}

int orama_ctx_pci_bus_id(orama_ctx* ctx, char* out, int capacity) {
    ORAMA_REQUIRE(ctx && out && capacity >= 13, "null argument or a buffer under 13 bytes");
    ORAMA_HIP_TRY(hipDeviceGetPCIBusId(out, capacity, ctx->device));
    return ORAMA_OK;
}

int orama_ctx_set_scan_tuning(orama_ctx* ctx, int rows_per_wave, int blocks_per_cu, int nontemporal) {
    ORAMA_REQUIRE(ctx, "null ctx");
    orama::ScanTuning t;
    t.rows_per_wave = rows_per_wave;
    t.blocks_per_cu = blocks_per_cu;
    t.nontemporal = nontemporal;
    ORAMA_REQUIRE(orama::scan_tuning_valid(t), "scan tuning out of range (rows in {1,2,4,8}, blocks/CU in [1,32])");
    ctx->scan_tuning = t;
    return ORAMA_OK;
}

int orama_ctx_set_f16_tuning(orama_ctx* ctx, int ksteps_per_chunk, int ring_chunks) {
    ORAMA_REQUIRE(ctx, "null ctx");
    ORAMA_REQUIRE((ksteps_per_chunk == 8 || ksteps_per_chunk == 12 || ksteps_per_chunk == 16) && ring_chunks >= 2 &&
                      ring_chunks <= 4,
                  "f16 tuning out of range (k-steps per chunk in {8,12,16}, ring chunks in [2,4])");
    ctx->f16_kc = ksteps_per_chunk;
    ctx->f16_nbuf = ring_chunks;
    return ORAMA_OK;
}

int orama_dev_malloc(orama_ctx* ctx, uint64_t bytes, void** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    ORAMA_ON_DEVICE(ctx->device);
    ORAMA_HIP_TRY(hipMalloc(out, bytes ? (size_t)bytes : 8));
    return ORAMA_OK;
}

void orama_dev_free(orama_ctx* ctx, void* d_ptr) {
    if (!ctx || !d_ptr) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(ctx->device);
    (void)hipFree(d_ptr);
}

int orama_dev_upload(orama_ctx* ctx, void* d_dst, uint64_t offset, const void* src, uint64_t bytes) {
    ORAMA_REQUIRE(ctx && (bytes == 0 || (d_dst && src)), "null argument");
    if (bytes == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(ctx->device);
    ORAMA_HIP_TRY(hipMemcpy(static_cast<char*>(d_dst) + offset, src, (size_t)bytes, hipMemcpyHostToDevice));
    return ORAMA_OK;
}

int orama_dev_download(orama_ctx* ctx, const void* d_src, uint64_t offset, void* dst, uint64_t bytes) {
    ORAMA_REQUIRE(ctx && (bytes == 0 || (d_src && dst)), "null argument");
    if (bytes == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(ctx->device);
    ORAMA_HIP_TRY(hipMemcpy(dst, static_cast<const char*>(d_src) + offset, (size_t)bytes, hipMemcpyDeviceToHost));
    return ORAMA_OK;
}

int orama_stream_create(orama_ctx* ctx, int high_priority, void** out_stream) {
    ORAMA_REQUIRE(ctx && out_stream, "null argument");
    *out_stream = nullptr;
    ORAMA_ON_DEVICE(ctx->device);
    int lo = 0, hi = 0;  // numerically lower = higher priority
    ORAMA_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s = nullptr;
    ORAMA_HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo));
    *out_stream = s;
    return ORAMA_OK;
}

void orama_stream_destroy(orama_ctx* ctx, void* stream) {
    if (!ctx || !stream) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(ctx->device);
    (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
}

int orama_stream_synchronize(orama_ctx* ctx, void* stream) {
    ORAMA_REQUIRE(ctx, "null ctx");
    ORAMA_ON_DEVICE(ctx->device);
    ORAMA_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return ORAMA_OK;
}

int orama_ctx_set_option(orama_ctx* ctx, const char* name, long long value) {
    ORAMA_REQUIRE(ctx && name, "null argument");
    for (const CtxOption& o : kCtxOptions) {
        if (std::strcmp(o.name, name) != 0) continue;
        ORAMA_REQUIRE(value >= o.lo && value <= o.hi, "option %s: %lld outside [%lld, %lld]", name, value, o.lo, o.hi);
        return o.set(ctx, value);
    }
    orama::set_error("unknown option %s", name);
    return ORAMA_ERR_INVALID;
}

int orama_ctx_set_f32_batch(orama_ctx* ctx, int min_queries) {
    ORAMA_REQUIRE(ctx, "null context");
    ORAMA_REQUIRE(min_queries >= 0 && min_queries <= 4096, "f32 batch threshold %d outside [0, 4096]", min_queries);
    ctx->f32_mfma_min_q = (uint32_t)min_queries;
    return ORAMA_OK;
}

int orama_ctx_set_two_stage(orama_ctx* ctx, int on) {
    ORAMA_REQUIRE(ctx, "null context");
    ORAMA_REQUIRE(on >= 0 && on <= 2, "two-stage mode %d outside [0, 2]", on);
    ctx->two_stage = on;
    return ORAMA_OK;
}

int orama_ctx_set_bm25_ranges(orama_ctx* ctx, int on) {
    ORAMA_REQUIRE(ctx, "null context");
    ORAMA_REQUIRE(on >= 0 && on <= 4, "bm25 ranges mode %d outside [0, 4]", on);
    // 0 / 1 / 2 choose the scorer and nothing else: the key-list form is its own option ("k3r_compact") and survives this call —
    // a test's restoring set_bm25_ranges(1) used to switch compact lists back on behind an A/B run's back
    ctx->bm25_ranges = on != 0;
    ctx->bm25_ranges_hybrid = on == 1 || on >= 3;
    if (on == 3) ctx->bm25_compact_keys = false;                          // (kept: = mode 1 + option k3r_compact 0)
    if (on == 4) ctx->bm25_compact_keys = true, ctx->bm25_compact_min = 1u;  // (kept: = mode 1 + option k3r_compact 2)
    return ORAMA_OK;
}

int orama_ctx_set_f16_wide(orama_ctx* ctx, int mode) {
    ORAMA_REQUIRE(ctx, "null ctx");
    ORAMA_REQUIRE(mode >= 0 && mode <= 5, "f16 wide mode %d outside [0, 5]", mode);
#if !ORAMA_COMPARISON_KERNELS
    ORAMA_SUPPORT(mode != 1 && mode != 5, "f16 wide mode %d (K2c / K2h) exists in comparison builds only (ORAMA_COMPARISON_KERNELS=1)", mode);
#endif
    ctx->f16_wide = mode;
    return ORAMA_OK;
}

int orama_prof_enable(orama_ctx* ctx, int on) {
    ORAMA_REQUIRE(ctx, "null ctx");
    ctx->prof.on = on != 0;
    return ORAMA_OK;
}

int orama_prof_reset(orama_ctx* ctx) {
    ORAMA_REQUIRE(ctx, "null ctx");
    ORAMA_ON_DEVICE(ctx->device);
    ctx->prof.reset();
    return ORAMA_OK;
}

int orama_prof_get(orama_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches) {
    ORAMA_REQUIRE(ctx && kernel, "null argument");
    ORAMA_ON_DEVICE(ctx->device);
    ctx->prof.resolve();
    std::lock_guard<std::mutex> g(ctx->prof.mu);
    auto it = ctx->prof.acc.find(kernel);
    double ms = 0.0;
    uint64_t n = 0;
    if (it != ctx->prof.acc.end()) {
        ms = it->second.ms;
        n = it->second.n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return ORAMA_OK;
}

int orama_prof_samples(orama_ctx* ctx, const char* kernel, float* out_ms, uint64_t capacity, uint64_t* n) {
    ORAMA_REQUIRE(ctx && kernel && n, "null argument");
    ORAMA_ON_DEVICE(ctx->device);
    ctx->prof.resolve();
    std::lock_guard<std::mutex> g(ctx->prof.mu);
    auto it = ctx->prof.acc.find(kernel);
    uint64_t have = it == ctx->prof.acc.end() ? 0 : (uint64_t)it->second.samples.size();
    if (out_ms)
        for (uint64_t i = 0; i < have && i < capacity; ++i) out_ms[i] = it->second.samples[i];
    *n = have;
    return ORAMA_OK;
}

}  // extern "C"
