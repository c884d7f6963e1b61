// common.hpp — shared host-side plumbing of liborama_hip.so (errors, device buffers, the per-GPU
// context with its stream/scratch pools and the HIP-event profiler).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <unordered_map>
#include <vector>

#include "orama_hip.h"

namespace orama {

// The environment of the PRODUCT library is the deployment allow-list (ORAMA_RCCL_LIB, ORAMA_SCRATCH_POOL_MIB,
// ORAMA_MAX_INFLIGHT, ORAMA_ACQUIRE_TIMEOUT_MS, ORAMA_SHARD_LANES, ORAMA_TWO_STAGE, ORAMA_VMM — read with std::getenv where they
// apply; tests/test_abi.py greps for any other).  Sweeps, A/B switches, timing ablations and traces are read through dev_env:
// nullptr unless the library was built as the comparison flavour (ORAMA_COMPARISON_KERNELS=1, liborama_hip_cmp.so) — a variable
// that can change an answer or a code path has no business in liborama_hip.so.  What tests and tuning
// scripts legitimately switch goes through orama_ctx_set_option.
inline const char* dev_env(const char* name) {
#if ORAMA_COMPARISON_KERNELS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

#define ORAMA_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            ::orama::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,    \
                               __LINE__);                                                    \
            return e__ == hipErrorOutOfMemory ? ORAMA_ERR_OOM : ORAMA_ERR_HIP;               \
        }                                                                                    \
    } while (0)

#define ORAMA_TRY(expr)                  \
    do {                                 \
        int s__ = (expr);                \
        if (s__ != ORAMA_OK) return s__; \
    } while (0)

#define ORAMA_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            ::orama::set_error(__VA_ARGS__); \
            return ORAMA_ERR_INVALID;       \
        }                                   \
    } while (0)

// A valid request outside the implemented envelope (the limits listed at the top of orama_hip.h).
#define ORAMA_SUPPORT(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            ::orama::set_error(__VA_ARGS__); \
            return ORAMA_ERR_UNSUPPORTED;   \
        }                                   \
    } while (0)

// ---------------------------------------------------------------- current device
// The HIP "current device" is per-thread state that belongs to the CALLER (a Rust host may run its own HIP code on the
// same thread): every entry point selects its context's device for the duration of the call and puts the caller's
// device back when it returns.
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    DeviceScope() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
        else prev = -1;  // nothing to put back
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define ORAMA_CAT2_(a, b) a##b
#define ORAMA_CAT_(a, b) ORAMA_CAT2_(a, b)
// select `device` until the end of the enclosing scope (fails the call like ORAMA_HIP_TRY)
#define ORAMA_ON_DEVICE(device)                                        \
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(device);    \
    ORAMA_HIP_TRY(ORAMA_CAT_(dev_scope__, __LINE__).err)

// ---------------------------------------------------------------- device / pinned buffers
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // Ensure capacity (contents NOT preserved when it grows).
    int reserve(size_t bytes) {
        if (bytes <= cap) return ORAMA_OK;
        release();
        size_t want = bytes < 256 ? 256 : bytes;
        ORAMA_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return ORAMA_OK;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    int reserve(size_t bytes) {
        if (bytes <= cap) return ORAMA_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        // (portable + mapped: kernels of EVERY device of the process read and write it — the staging kernel, stage.hip)
        ORAMA_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped));
        cap = want;
        return ORAMA_OK;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// ---------------------------------------------------------------- profiler
// Brackets launches of the named hot kernels with hipEvents on the launching stream; the pairs
// are resolved lazily in orama_prof_get (after the caller synchronised).
struct Profiler {
    std::mutex mu;
    bool on = false;
    struct Acc {
        double ms = 0.0;
        uint64_t n = 0;
        std::vector<float> samples;  // per-launch durations (orama_prof_samples), bounded
        std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    };
    std::map<std::string, Acc> acc;
    std::vector<hipEvent_t> free_events;

    hipEvent_t get_event();
    void begin(const char* name, hipStream_t s, hipEvent_t* start);
    void end(const char* name, hipStream_t s, hipEvent_t start);
    void pair(const char* name, hipEvent_t* start, hipEvent_t* stop);  // for ONE dispatch (hipExtLaunchKernelGGL): nothing is recorded here
    int resolve();
    void reset();
    ~Profiler();
};

struct ProfScope {
    Profiler* prof;
    const char* name;
    hipStream_t stream;
    hipEvent_t start = nullptr;
    ProfScope(Profiler* p, const char* n, hipStream_t s) : prof(p), name(n), stream(s) {
        if (prof && prof->on) prof->begin(name, stream, &start);
    }
    ~ProfScope() {
        if (start) prof->end(name, stream, start);
    }
};

// The event pair of ONE kernel dispatch: handed to hipExtLaunchKernelGGL, which stamps them from the dispatch's own completion
// signal — no event-record (barrier) packets before and after the kernel.  Both null when the profiler is off.
struct ProfLaunch {
    hipEvent_t start = nullptr, stop = nullptr;
    ProfLaunch(Profiler* p, const char* name) {
        if (p && p->on) p->pair(name, &start, &stop);
    }
};

// ---------------------------------------------------------------- per-call scratch
// One set per in-flight search: its own stream, device scratch and pinned staging, so that
// concurrent orama_*_search calls never share mutable state (re-entrancy contract of the ABI).
// Three kinds of set, never mixed in use: kScratchRecords sets carry K3's per-document records (1.4 GB at 10 M documents),
// kScratchVector sets the buffers of vector searches (a wide fp16 batch sizes its candidate lists in GB), kScratchGeneral
// sets everything else (the range scorer needs a few MB).  A set that served several of these would end up holding all
// of their buffers — 32 such sets overrun the pool's byte budget — or, worse, keep being re-sized: a vector batch that
// draws a set last used by a small full-text query re-allocates GBs (hipFree synchronises the device).
constexpr int kScratchGeneral = 0, kScratchRecords = 1, kScratchVector = 2;

struct Scratch {
    int kind = kScratchGeneral;
    hipStream_t stream = nullptr;
    DevBuf query;      // q x dim f32
    DevBuf dist;       // per-row distances / dense scores
    DevBuf sel_state;  // SelectState[q]
    DevBuf sel_keys;   // collected composite keys
    DevBuf out_idx, out_val, out_n, out_ids;
    DevBuf bitmap;     // uploaded allow bitmap
    DevBuf f16_bfrag;  // K2c: fp16 query fragments + 1/|q| of the current wide batch
    DevBuf f16_tau_cap;  // ORAMA_F16_TAU_ORACLE=1 (experiment): the previous call's final k-th distances
    uint32_t f16_tau_cap_q = 0, f16_tau_cap_k = 0;
    DevBuf misc0, misc1, misc2, misc3, misc4, misc5;
    // fp32 batches on the matrix cores (vec_store.hip search_enqueue_f32_batch): the candidate stage's answers + the exact
    // distances of the candidates, and the list sets / answers of the device-side fallback for unproven queries
    DevBuf mfma_cand, mfma_fb_lists, mfma_fb_tmp, mfma_fb_out;
    PinnedBuf h_in, h_out, h_misc;
    // BM25 accumulators: epoch-stamped, zeroed only when (re)allocated (see bm25_kernels.hip)
    DevBuf bm25_acc, bm25_emit;
    uint32_t bm25_epoch = 0;
    // two-stream search (scan on one stream, top-k tail on another): scan → tail and tail → next-scan ordering
    hipEvent_t ev_scan = nullptr, ev_tail = nullptr;
    bool tail_recorded = false;
    // "everything enqueued on this set's stream so far is done", for ANOTHER set's stream to wait on (the one-call hybrid
    // search: the range scorer's stream runs the tail behind the vector leg's top-k); created on first use
    hipEvent_t ev_done = nullptr;
    ~Scratch() {
        if (stream) (void)hipStreamDestroy(stream);
        if (ev_scan) (void)hipEventDestroy(ev_scan);
        if (ev_tail) (void)hipEventDestroy(ev_tail);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
};

}  // namespace orama

namespace orama {
struct ScanTuning {
    int rows_per_wave = 8;  // rows each wave keeps in flight per iteration (1, 2, 4, 8); capped so that
                            // rows x ceil(dim/256) <= 16 loads of 16 B per lane (768 dims: 4 rows, 384 dims: 8 rows)
    int blocks_per_cu = 2;  // persistent grid = CUs x this (measured best on MI355X: profiles/r01_sweep_ns_v1.json)
    int nontemporal = 1;    // stream the corpus with `nt` loads
};
}  // namespace orama

struct orama_ctx {
    int device = 0;
    orama::ScanTuning scan_tuning;  // defaults from ORAMA_SCAN_* env, see orama_ctx_set_scan_tuning
    // K1 keeps per-wave top-k lists (k <= 128) in registers instead of writing dense distances: 0 never, 1 always, 2 =
    // for a single query over >= 3 GB of rows (ORAMA_FUSED_TOPK).  Exact (same answer as the dense path, tested).  Round 1
    // measured it slower everywhere (profiles/r01_fused_topk_experiment.md: the LDS bitonic reduction of the wave lists);
    // with the radix-select reduction of round 2 it wins at NS size: 4.51 -> 4.32 ms per scan.
    int fused_topk = 2;
    int f32_multi = 1;   // K1b: fp32 batches of 2..8 queries share one corpus pass (ORAMA_F32_MULTI=0 disables)
    // K1m (vec_f32_mfma.hip): fp32 batches of at least this many queries share corpus passes of <= 32 queries on the matrix
    // cores (v_mfma_f32_32x32x2_f32) with K2's threshold filter behind them; 0 = never (orama_ctx_set_f32_batch)
    uint32_t f32_mfma_min_q = 9;
    // ... and the candidate scan of such a batch is K1x (vec_f32_cvt.hip: the fp32 rows rounded to fp16 in registers, fp16 MFMA,
    // <= 64 queries per pass, HBM-bound) where every row has a sound fp16 image; false = always K1m (option "f32_batch_cvt")
    bool f32_batch_cvt = true;
    // fp16 batches of 65..256 queries share one corpus pass: 4 = K2q (queries stationary in registers, default; batches of
    // <= 128 and rows wider than 768 dimensions take K2d), 5 = K2h (K loop split over a wave pair), 2 = K2d (dedicated
    // loader waves), 3 = K2d second geometry, 1 = K2c (round 1: MFMA waves issue the DMA), 0 = K2 in passes of 64
    // (ORAMA_F16_WIDE)
    int f16_wide = 4;
    // plain BM25 top-k searches of a resident store use the range-partitioned scorer (K3r, bm25_ranges.hip);
    // 0 = always the per-document-record scorer K3 (ORAMA_BM25_RANGES, orama_ctx_set_bm25_ranges)
    int bm25_ranges = 1;
    int bm25_ranges_hybrid = 1;  // orama_post_search_hybrid on the range scorer where it applies (ORAMA_BM25_RANGES_HYBRID=0: K3)
    // K3r's plain top-k batches append only the keys that can still reach the answer (round 5, bm25_ranges.hip "COMPACT");
    // false = one key slot per posting as in round 4 (ORAMA_K3R_COMPACT=0, orama_ctx_set_bm25_ranges(ctx, 3): A/B runs)
    // orama_hybrid_search on the range scorer can finish on the device (hybrid_tail.hip: one read-back, one host wake-up) —
    // built in round 5, measured, NOT the default: four small dependent launches behind the scan's top-k cost more than the
    // wake-up they save (C4 p50 4.454 against 4.403 ms, shadow store 2.447 against 2.360: profiles/r05_hybrid_device_tail_ab.log).
    // ORAMA_HYBRID_DEVICE_TAIL=1 enables it (the parity test runs both forms).
    bool hybrid_device_tail = false;
    bool bm25_compact_keys = true;
    // ... for batches of at least this many queries: the lists' cursors are bumped by returning global atomics (~50 ns each on
    // one address) and a lone query's 400 workgroups run together — 14.7 -> 11.7 K single calls per second with compact lists
    // (orama_ctx_set_bm25_ranges(ctx, 4): every batch size, for the parity tests)
    uint32_t bm25_compact_min = 8;
    // small host<->device blocks (the range scorer's chunk tables and answers, a lone query and its hits) move by a kernel of
    // the caller's stream instead of an SDMA copy (stage.hip); ORAMA_STAGE_COPY=dma puts the copy commands back
    bool stage_by_kernel = true;
    // the selection's final launch writes the answers to the pinned host block itself — K3r's chunks, a lone fp32 vector search,
    // the vector leg of a hybrid call (ORAMA_DIRECT_OUT=0: a staging launch behind it)
    bool direct_out = true;
    // two-stream searches: the scan's completion event rides on its dispatch instead of a record packet behind it (vec_store.hip
    // scan_end; ORAMA_SCAN_DONE_EVENT=record restores the packet)
    bool scan_done_on_dispatch = true;
    int select_wide = 1;  // K4: a few long dense lists take one round of 32 values per thread (select.hip, pairs_reduce_wide_kernel)
    int k3r_merge = 0;  // comparison builds only (ORAMA_COMPARISON_KERNELS=1): ORAMA_K3R_MERGE=1 scores ranges with the round-3 merge tree
    // stores created as ORAMA_DTYPE_F32_SHADOW16 answer orama_vec_search in two stages (fp16 candidates, fp32 decision);
    // 0 = always the plain fp32 scan, 2 = two stages also where the plain scan is expected to be faster (small stores
    // with few queries) (ORAMA_TWO_STAGE, orama_ctx_set_two_stage)
    int two_stage = 1;
    int f16_solo = 2;    // shadow scans of <= 4 queries use K1h (dot products, no MFMA): 2 = K1-shaped loop where it applies, 1 = register ring; ORAMA_F16_SOLO=0: K2
    // K2 register ring: k-steps per chunk, chunks (ORAMA_F16_KC / ORAMA_F16_NBUF).  8 x 2 since round 3: with the kernel free
    // of scratch the shallow ring fits three waves per SIMD (130-158 VGPRs) and that is worth more than a third chunk in
    // flight (16 queries: 2.29 vs 2.37 ms per pass, 64 queries: equal; profiles/r03_k2_ring_sweep.log)
    int f16_kc = 8, f16_nbuf = 2;
    // fp16 filter scans (vec_store.hip search_enqueue_f16; orama_ctx_set_option "f16_head_rows" / "f16_cand_mib" / "f16_chunk_grow" /
    // "f16_grow_factor"): rows of the dense head (0 = 131 072), candidate budget in MiB (0 = 6 GiB), super-chunk growth (-1 = the
    // rule: on for wide passes, 0 off, 1 on) and its factor
    uint64_t f16_head_rows = 0, f16_cand_mib = 0;
    int f16_chunk_grow = -1, f16_grow_factor = 2;
    uint32_t two_stage_spare = 256;  // spare candidates of the two-stage plan ("two_stage_spare")
    // postings per document range of K3r ("k3r_target").  1 792 since round 6 (1 536 before): the scoring launch's cost is mostly
    // per-WORKGROUP fixed work, so fuller ranges are cheaper per posting — 231 K -> 239 K queries/s in one lease, 242 K at 1 920
    // (profiles/r06_k3r_target_sweep.log); 1 920 leaves the 2 048-posting cap ~3 sigma of a Poisson range away, 1 792 six
    uint32_t k3r_target = 0;  // (0: 7/8 of what a scoring workgroup holds — 1 792 with 256 threads)
    // comparison builds: the plain top-k batch's scoring launch by bm25_ranges_fast.hip (1) instead of bm25_ranges.hip's body (0)
    bool k3r_fast = false;
    // comparison builds: dense-list accelerators of the postings store (orama_post::d_acc), built at build / append time when set.  Their only reader is
    // the comparison unit bm25_ranges_fast.hip (not faster: profiles/r06_k3r_fast_body.md), so the default is off
    bool bm25_dense_acc = false;
    bool select_pairs = true;        // K4: (value, index) lists in two launches ("select_pairs" 0 = histogram passes)
    int compute_units = 0;
    uint64_t hbm_bytes = 0;
    char name[256] = {0};
    orama::Profiler prof;
    std::mutex pool_mu;
    // callers waiting for their sets, first come first served — each on its own condition variable (a release wakes exactly
    // the caller whose turn it is; see orama_ctx::acquire_n)
    struct PoolWaiter {
        uint32_t need = 0;
        bool granted = false;
        std::condition_variable cv;
    };
    std::deque<PoolWaiter*> pool_waiters;
    void grant_waiters();  // pool_mu held
    std::vector<std::unique_ptr<orama::Scratch>> pool;
    uint32_t leased = 0;  // scratch sets out on lease; bounded by max_inflight (callers beyond it wait their turn)
    uint32_t held = 0;    // sets kept by long-lived handles (orama_scores): NOT counted against max_inflight — a handle is
                          // released by its owner, not by the end of a call, so counting it could starve every search
    uint32_t max_inflight = 32;
    uint32_t acquire_timeout_ms = 30000;  // a caller that cannot get its sets within this fails with ORAMA_ERR_BUSY
    // resident allow-bitmaps (orama_allow_*): device pointer -> bits; a search whose `allow_bitmap` argument is one
    // of these pointers uses it in place instead of uploading host words
    std::mutex allow_mu;
    std::unordered_map<const void*, uint64_t> allow_reg;
    // ... and the version of its CONTENT: a number no other content of any resident bitmap of this context ever had (a new one
    // at creation and after every orama_allow_set).  What was counted under a filter can be remembered under it
    // (orama_post::df_union).
    std::unordered_map<const void*, uint64_t> allow_version;
    uint64_t allow_next_version = 1;

    // Borrow a scratch set (creates one when the pool is empty); blocks while max_inflight sets are out.
    int acquire(std::unique_ptr<orama::Scratch>* out, int kind = orama::kScratchGeneral);
    // Two sets at once (the fused hybrid search runs its legs on two streams): taken together, so that callers holding
    // one set each can never wait for each other.
    int acquire2(std::unique_ptr<orama::Scratch>* a, std::unique_ptr<orama::Scratch>* b);
    int acquire_n(uint32_t n, std::unique_ptr<orama::Scratch>** outs, const int* kinds = nullptr);
    void release(std::unique_ptr<orama::Scratch> s, bool detached = false);
    // A leased set becomes handle-held: it stops counting against max_inflight (waiters are woken).
    void detach_one();
};

namespace orama {
struct ScratchLease {
    orama_ctx* ctx;
    std::unique_ptr<Scratch> s;
    int kind;
    bool detached = false;
    explicit ScratchLease(orama_ctx* c, int k = kScratchGeneral) : ctx(c), kind(k) {}
    // the set outlives the call (a handle keeps it): stop counting it against the in-flight bound
    void detach() {
        if (s && !detached) {
            ctx->detach_one();
            detached = true;
        }
    }
    int init() { return ctx->acquire(&s, kind); }
    static int init_pair(ScratchLease& a, ScratchLease& b) {
        std::unique_ptr<Scratch>* outs[2] = {&a.s, &b.s};
        const int kinds[2] = {a.kind, b.kind};
        return a.ctx->acquire_n(2, outs, kinds);
    }
    static int init_three(ScratchLease& a, ScratchLease& b, ScratchLease& c) {
        std::unique_ptr<Scratch>* outs[3] = {&a.s, &b.s, &c.s};
        const int kinds[3] = {a.kind, b.kind, c.kind};
        return a.ctx->acquire_n(3, outs, kinds);
    }
    ~ScratchLease() {
        if (s) ctx->release(std::move(s), detached);
    }
    Scratch* operator->() { return s.get(); }
};

// The device bitmap a search reads: `allow_bitmap` itself when it is the token of a resident bitmap
// (orama_allow_token), else a per-call upload of the host words into sc->bitmap on `s`.  nullptr stays nullptr.
// The content version of a resident bitmap (orama_ctx::allow_version); 0 for host words and for nullptr.
uint64_t allow_content_version(orama_ctx* ctx, const uint64_t* allow_bitmap);
// `*version` (optional): the content version of a resident bitmap, 0 for host words (nobody knows what they hold).
int resolve_allow(orama_ctx* ctx, Scratch* sc, const uint64_t* allow_bitmap, uint64_t bitmap_bits, hipStream_t s,
                  const uint64_t** d_allow, uint64_t* version = nullptr);

inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
}  // namespace orama
