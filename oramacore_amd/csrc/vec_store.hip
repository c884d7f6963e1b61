// vec_store.hip — HBM-resident vector store behind the orama_vec_* entry points.
// Mirrors what oramacore_fields::embedding::EmbeddingStorage provides to
// EmbeddingFieldStorage (src/collection_manager/sides/read/index/embedding_field.rs:63-320):
// insert (N rows per doc) / delete (tombstone) / compact / info / search.
//
// HBM layout
//   f32 (the reference's element type): one contiguous row-major matrix [rows][dim], rows 16-B aligned
//        when dim % 4 == 0 — scanned by K1 (vec_kernels.hip);
//   f16 (extension): MFMA-fragment tiles [row/32][k/16][lane][8 halves] — scanned by K2 (vec_f16.hip);
//   per row: 1/|x| (f32), DocumentId (u64), tombstone bit.
// Sized for 288 GB: the matrix is ONE allocation (30.7 GB for 10 M x 768 f32) that grows geometrically.
#include <algorithm>
#include <cmath>
#include <shared_mutex>
#include <unordered_map>

#include <cstdlib>

#include "common.hpp"
#include "select.hpp"
#include "vec_f16.hpp"
#include "vec_internal.hpp"
#include "vec_kernels.hpp"

using namespace orama;

struct orama_vec {
    orama_ctx* ctx = nullptr;
    uint32_t dim = 0;
    int metric = ORAMA_METRIC_COSINE;
    int dtype = ORAMA_DTYPE_F32;
    std::shared_mutex mu;  // searches: shared; insert/delete/compact: exclusive

    DevBuf rows;      // f32: cap_rows x dim; f16: tiles(cap_rows) x tile_bytes
    DevBuf inv_norm;  // cap_rows f32
    DevBuf row_doc;   // cap_rows u64
    DevBuf dead;      // cap_rows bits (u32 words)
    uint64_t n_rows = 0;
    uint64_t cap_rows = 0;
    uint64_t n_dead = 0;
    uint64_t version = 0;

    std::vector<uint64_t> h_row_doc;  // host mirror of row_doc
    std::vector<uint32_t> h_dead;     // host mirror of the tombstone bitmap
    bool doc_rows_built = false;
    std::unordered_map<uint64_t, std::vector<uint32_t>> doc_rows;  // built lazily for delete

    // scratch for the device-pointer entry point, one per caller stream
    std::mutex dev_mu;
    std::map<hipStream_t, std::unique_ptr<Scratch>> dev_scratch;

    bool f16() const { return dtype == ORAMA_DTYPE_F16; }
    size_t row_bytes() const { return (size_t)dim * sizeof(float); }  // f32 row (host side / f32 store)
    size_t matrix_bytes(uint64_t rows_) const {
        return f16() ? (size_t)(f16_tiles(rows_) * f16_tile_bytes(dim)) : (size_t)rows_ * row_bytes();
    }
};

namespace {

struct Arrays {
    void *rows = nullptr, *norm = nullptr, *doc = nullptr, *dead = nullptr;
    uint64_t cap = 0;
};

int alloc_arrays(orama_vec* v, uint64_t cap, Arrays* a, hipStream_t s) {
    if (v->f16()) cap = (cap + 31) & ~31ull;
    const size_t dead_words = (size_t)((cap + 31) / 32);
    ORAMA_HIP_TRY(hipMalloc(&a->rows, std::max<size_t>(256, v->matrix_bytes(cap))));
    // f16: K2c fetches the 256 inverse norms of a block tile with one 1-KiB DMA — pad to whole block tiles
    const size_t norm_bytes = (size_t)(v->f16() ? ((cap + 255) & ~255ull) + 256 : cap) * sizeof(float);
    ORAMA_HIP_TRY(hipMalloc(&a->norm, std::max<size_t>(norm_bytes, 4)));
    ORAMA_HIP_TRY(hipMalloc(&a->doc, (size_t)cap * sizeof(uint64_t)));
    ORAMA_HIP_TRY(hipMalloc(&a->dead, dead_words * sizeof(uint32_t)));
    ORAMA_HIP_TRY(hipMemsetAsync(a->dead, 0, dead_words * sizeof(uint32_t), s));
    if (v->f16()) {  // padding rows of partial tiles must read as zeros
        ORAMA_HIP_TRY(hipMemsetAsync(a->rows, 0, v->matrix_bytes(cap), s));
        ORAMA_HIP_TRY(hipMemsetAsync(a->norm, 0, norm_bytes, s));
    }
    a->cap = cap;
    return ORAMA_OK;
}

void adopt_arrays(orama_vec* v, const Arrays& a) {
    v->rows.release();
    v->inv_norm.release();
    v->row_doc.release();
    v->dead.release();
    v->rows.p = a.rows;
    v->rows.cap = std::max<size_t>(256, v->matrix_bytes(a.cap));
    v->inv_norm.p = a.norm;
    v->inv_norm.cap = (size_t)a.cap * sizeof(float);
    v->row_doc.p = a.doc;
    v->row_doc.cap = (size_t)a.cap * sizeof(uint64_t);
    v->dead.p = a.dead;
    v->dead.cap = (size_t)((a.cap + 31) / 32) * sizeof(uint32_t);
    v->cap_rows = a.cap;
}

int grow(orama_vec* v, uint64_t need_rows, hipStream_t s) {
    if (need_rows <= v->cap_rows) return ORAMA_OK;
    ORAMA_REQUIRE(need_rows < 0xfffffff0ull, "vector store limited to 2^32-16 rows");
    uint64_t cap = v->cap_rows ? v->cap_rows : 1024;
    while (cap < need_rows) cap += cap / 2 + 1024;
    Arrays a;
    ORAMA_TRY(alloc_arrays(v, cap, &a, s));
    if (v->n_rows) {
        ORAMA_HIP_TRY(hipMemcpyAsync(a.rows, v->rows.p, v->matrix_bytes(v->n_rows), hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(a.norm, v->inv_norm.p, (size_t)v->n_rows * sizeof(float),
                                     hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(a.doc, v->row_doc.p, (size_t)v->n_rows * sizeof(uint64_t),
                                     hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(a.dead, v->dead.p, (size_t)((v->n_rows + 31) / 32) * 4,
                                     hipMemcpyDeviceToDevice, s));
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    adopt_arrays(v, a);
    return ORAMA_OK;
}

bool row_valid(const float* x, uint32_t d) {  // EmbeddingIndexer::index_vec_vec -> None (assumption)
    float n2 = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        if (!std::isfinite(x[i])) return false;
        n2 += x[i] * x[i];
    }
    return std::isfinite(n2) && n2 > 0.0f;
}

// Write `n` f32 rows that already sit in HBM at `d_src` into the store at rows [first, first+n).
int store_rows_from_device(orama_vec* v, void* rows_base, float* norm_base, const float* d_src, uint64_t first,
                           uint64_t n, hipStream_t s) {
    if (v->f16()) {
        ORAMA_TRY(launch_f16_store_rows(rows_base, d_src, first, n, v->dim, s));
        ORAMA_TRY(launch_f16_inv_norm(rows_base, first, n, v->dim, norm_base, s, v->metric));
    } else {
        ORAMA_HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(rows_base) + (size_t)first * v->row_bytes(), d_src,
                                     (size_t)n * v->row_bytes(), hipMemcpyDeviceToDevice, s));
        if (v->metric == ORAMA_METRIC_COSINE)
            ORAMA_TRY(launch_row_inv_norm_f32(reinterpret_cast<const float*>(rows_base), first, n, v->dim,
                                              norm_base, s));
    }
    return ORAMA_OK;
}

// ---------------------------------------------------------------- f32: K1 + dense K4
// Stream ordering of the two-stream mode: scans run on `s_scan`, the top-k tail on `s`; the tail waits for its
// scan, and a scan that reuses this scratch set waits for the previous tail (events in the scratch set).
int scan_begin(Scratch* sc, hipStream_t s_scan, hipStream_t s) {
    if (s_scan == s) return ORAMA_OK;
    if (!sc->ev_scan) ORAMA_HIP_TRY(hipEventCreateWithFlags(&sc->ev_scan, hipEventDisableTiming));
    if (!sc->ev_tail) ORAMA_HIP_TRY(hipEventCreateWithFlags(&sc->ev_tail, hipEventDisableTiming));
    if (sc->tail_recorded) ORAMA_HIP_TRY(hipStreamWaitEvent(s_scan, sc->ev_tail, 0));
    return ORAMA_OK;
}
int scan_end(Scratch* sc, hipStream_t s_scan, hipStream_t s) {
    if (s_scan == s) return ORAMA_OK;
    ORAMA_HIP_TRY(hipEventRecord(sc->ev_scan, s_scan));
    ORAMA_HIP_TRY(hipStreamWaitEvent(s, sc->ev_scan, 0));
    return ORAMA_OK;
}
int tail_end(Scratch* sc, hipStream_t s_scan, hipStream_t s) {
    if (s_scan == s) return ORAMA_OK;
    ORAMA_HIP_TRY(hipEventRecord(sc->ev_tail, s));
    sc->tail_recorded = true;
    return ORAMA_OK;
}

int search_enqueue_f32(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s, hipStream_t s_scan) {
    const uint64_t n = v->n_rows;
    if (k <= kWaveListKeys && v->ctx->fused_topk) {
        // fused path: each K1 wave keeps its own best-k in registers; no dense distance array at all.
        ScanArgs a;
        a.corpus = v->rows.as<float>();
        a.inv_norm = v->inv_norm.as<float>();
        a.n = n;
        a.dim = v->dim;
        a.metric = v->metric;
        a.row_doc = v->row_doc.as<uint64_t>();
        a.dead = v->n_dead ? v->dead.as<uint32_t>() : nullptr;
        a.allow = d_allow;
        a.allow_bits = allow_bits;
        a.topk = k;
        const uint32_t waves = vec_scan_f32_waves(v->ctx, a);
        const uint32_t n_keys = waves * kWaveListKeys;
        const uint32_t chunks = (n_keys + kKeysChunk - 1) / kKeysChunk;
        ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)q * n_keys));
        ORAMA_TRY(sc->dist.reserve(sizeof(unsigned long long) * (size_t)q * (size_t)(chunks + 1) * k * 2 + 64));
        ORAMA_TRY(scan_begin(sc, s_scan, s));
        for (uint32_t j = 0; j < q; ++j) {
            a.query = d_queries + (size_t)j * v->dim;
            a.wave_lists = sc->sel_keys.as<unsigned long long>() + (size_t)j * n_keys;
            ORAMA_TRY(launch_vec_scan_f32(v->ctx, a, s_scan));
        }
        ORAMA_TRY(scan_end(sc, s_scan, s));
        ORAMA_TRY(launch_keys_topk(v->ctx, sc->sel_keys.as<unsigned long long>(), n_keys, n_keys, q, k, false,
                                v->row_doc.as<uint64_t>(), sc->dist.as<unsigned long long>(), nullptr, d_out_ids,
                                d_out_dist, d_out_n, s));
        return tail_end(sc, s_scan, s);
    }
    // dense path (k > 128): distances for every row, then K4 radix select.
    // queries are processed in groups so that the dense distance buffer stays <= ~1 GiB
    uint32_t group = q;
    if (n > 0) {
        const uint64_t g = (1ull << 28) / n;  // 2^28 floats
        group = (uint32_t)(g < 1 ? 1 : (g > q ? q : g));
    }
    ORAMA_TRY(sc->dist.reserve((size_t)group * (size_t)(n ? n : 1) * sizeof(float)));
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState) * (size_t)group));
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)group * k));
    for (uint32_t q0 = 0; q0 < q; q0 += group) {
        const uint32_t gq = (q - q0) < group ? (q - q0) : group;
        ORAMA_TRY(scan_begin(sc, s_scan, s));
        for (uint32_t j = 0; j < gq;) {
            ScanArgs a;
            a.corpus = v->rows.as<float>();
            a.inv_norm = v->inv_norm.as<float>();
            a.query = d_queries + (size_t)(q0 + j) * v->dim;
            a.n = n;
            a.dim = v->dim;
            a.metric = v->metric;
            a.row_doc = v->row_doc.as<uint64_t>();
            a.dead = v->n_dead ? v->dead.as<uint32_t>() : nullptr;
            a.allow = d_allow;
            a.allow_bits = allow_bits;
            a.out_dist = sc->dist.as<float>() + (size_t)j * n;
            // K1b: up to 8 queries share one corpus pass (a micro-batch of concurrent requests)
            const uint32_t nq = std::min<uint32_t>(gq - j, v->ctx->f32_multi ? kScanMultiMaxQ : 1u);
            if (nq >= 2 && vec_scan_f32_multi_supported(a)) {
                ORAMA_TRY(launch_vec_scan_f32_multi(v->ctx, a, nq, n, s_scan));
                j += nq;
            } else {
                ORAMA_TRY(launch_vec_scan_f32(v->ctx, a, s_scan));
                j += 1;
            }
        }
        ORAMA_TRY(scan_end(sc, s_scan, s));
        SelectPlan p;
        p.vals = sc->dist.as<float>();
        p.stride = n;
        p.n = (uint32_t)n;
        p.q = gq;
        p.k = k;
        p.descending = false;
        p.id_map = v->row_doc.as<uint64_t>();
        p.state = sc->sel_state.as<SelectState>();
        p.keys = sc->sel_keys.as<unsigned long long>();
        p.out_ids = d_out_ids + (size_t)q0 * k;
        p.out_val = d_out_dist + (size_t)q0 * k;
        p.out_n = d_out_n + q0;
        ORAMA_TRY(launch_select(v->ctx, p, s));
        ORAMA_TRY(tail_end(sc, s_scan, s));
    }
    return ORAMA_OK;
}

// ---------------------------------------------------------------- f16: K2 with fused threshold filter
// Per pass of <= 64 queries:
//   1. dense scan of the first S1 rows → per-query top-k (K4, batched) → tau_j = k-th best distance;
//   2. filter scan of the remaining rows in super-chunks: rows with distance < tau_j are appended to the
//      query's candidate list (which starts with the current best k); capacity = every row of the
//      super-chunk, so the result is exact for ANY data order; after each super-chunk the list is reduced
//      to the new best k and tau tightens;
//   3. the last reduction also maps rows → DocumentIds and applies the final tie order.
// Scores are never materialised for more than S1 rows; HBM traffic beyond the corpus pass is the
// candidate appends (expected k·ln(N/S1) per query on unordered data).
int search_enqueue_f16(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s) {
    const uint64_t n = v->n_rows;
    constexpr uint64_t kS1 = 131072;                 // dense head (rows), multiple of 32
    constexpr uint64_t kCandBudget = 6ull << 30;     // bytes of candidate lists per pass
    const uint32_t kpad_k = f16_kpad(v->dim);
    for (uint32_t q0 = 0; q0 < q;) {
        // <= 64 queries: K2 (whole batch as LDS-resident B fragments); more: K2c (GEMM-tiled, <= 256 per pass)
        const bool wide = v->ctx->f16_wide && (q - q0) > kF16MaxQ && (kpad_k / 16) % 2 == 0;
        const uint32_t gq = std::min<uint32_t>(wide ? kF16WideMaxQ : kF16MaxQ, q - q0);
        if (wide) ORAMA_TRY(sc->f16_bfrag.reserve(f16_wide_query_bytes(v->dim)));
        bool wide_prepared = false;
        auto scan = [&](const F16ScanArgs& args) -> int {
            if (!wide) return launch_vec_scan_f16(v->ctx, args, s);
            if (!wide_prepared)
                ORAMA_TRY(launch_f16_prepare_queries(args.queries, args.q, args.dim, args.metric, sc->f16_bfrag.p, s));
            wide_prepared = true;
            // f16_wide: 1 = K2c (MFMA waves also issue the DMA), 2.. = K2d (dedicated loader waves), geometry f16_wide-1
            if (v->ctx->f16_wide >= 2) return launch_vec_scan_f16_pc(v->ctx, args, sc->f16_bfrag.p, s, v->ctx->f16_wide - 1);
            return launch_vec_scan_f16_wide(v->ctx, args, sc->f16_bfrag.p, false, s);
        };
        const uint64_t s1 = std::min<uint64_t>(n, kS1);
        // super-chunk size: gq * (rows + k) * 8 B <= budget
        static const uint64_t budget = [] {
            const char* e = std::getenv("ORAMA_F16_CAND_MIB");
            return e ? (uint64_t)std::strtoull(e, nullptr, 10) << 20 : kCandBudget;
        }();
        uint64_t chunk_rows = budget / ((uint64_t)gq * 8);
        chunk_rows = std::max<uint64_t>(chunk_rows & ~255ull, 1u << 20);  // whole K2c block tiles (256 rows)
        const uint64_t rest = n > s1 ? n - s1 : 0;
        const uint64_t cand_stride = std::min<uint64_t>(rest, chunk_rows) + k;
        ORAMA_TRY(sc->dist.reserve((size_t)gq * (size_t)std::max<uint64_t>(s1, 1) * 4));
        ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState) * (size_t)gq));
        ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)gq * k));
        ORAMA_TRY(sc->misc1.reserve((size_t)gq * cand_stride * 4));  // cand_dist
        ORAMA_TRY(sc->misc2.reserve((size_t)gq * cand_stride * 4));  // cand_row
        ORAMA_TRY(sc->misc3.reserve((size_t)gq * 4 * 2));            // cand_count | tau
        ORAMA_TRY(sc->misc4.reserve((size_t)gq * k * 4));            // best rows
        ORAMA_TRY(sc->misc5.reserve((size_t)gq * k * 4));            // best dist
        float* cand_dist = sc->misc1.as<float>();
        uint32_t* cand_row = sc->misc2.as<uint32_t>();
        uint32_t* cand_count = sc->misc3.as<uint32_t>();
        float* tau = reinterpret_cast<float*>(cand_count + gq);
        uint32_t* best_row = sc->misc4.as<uint32_t>();
        float* best_dist = sc->misc5.as<float>();
        uint64_t* out_ids = d_out_ids + (size_t)q0 * k;
        float* out_dist = d_out_dist + (size_t)q0 * k;
        uint32_t* out_n = d_out_n + q0;

        F16ScanArgs a;
        a.tiled = v->rows.p;
        a.inv_norm = v->inv_norm.as<float>();
        a.queries = d_queries + (size_t)q0 * v->dim;
        a.q = gq;
        a.dim = v->dim;
        a.metric = v->metric;
        a.n_rows = n;
        a.row_doc = v->row_doc.as<uint64_t>();
        a.dead = v->n_dead ? v->dead.as<uint32_t>() : nullptr;
        a.allow = d_allow;
        a.allow_bits = allow_bits;
        // 1. dense head
        a.row_begin = 0;
        a.row_end = s1;
        a.out_dense = sc->dist.as<float>();
        a.dense_stride = s1;
        ORAMA_TRY(scan(a));
        SelectPlan p;
        p.vals = sc->dist.as<float>();
        p.stride = s1;
        p.n = (uint32_t)s1;
        p.q = gq;
        p.k = k;
        p.descending = false;
        p.state = sc->sel_state.as<SelectState>();
        p.keys = sc->sel_keys.as<unsigned long long>();
        const bool only_head = rest == 0;
        if (only_head) {
            p.id_map = v->row_doc.as<uint64_t>();
            p.out_ids = out_ids;
            p.out_val = out_dist;
            p.out_n = out_n;
            ORAMA_TRY(launch_select(v->ctx, p, s));
            q0 += gq;
            continue;
        }
        p.out_idx = best_row;
        p.out_val = best_dist;
        p.out_n = out_n;  // borrowed as the running count
        ORAMA_TRY(launch_select(v->ctx, p, s));
        // 2. filter scan of the rest in super-chunks
        for (uint64_t r0 = s1; r0 < n; r0 += chunk_rows) {
            const uint64_t r1 = std::min<uint64_t>(n, r0 + chunk_rows);
            ORAMA_TRY(launch_f16_seed_candidates(best_dist, best_row, out_n, gq, k, tau, cand_dist, cand_row,
                                                 cand_count, cand_stride, s));
            a.row_begin = r0;
            a.row_end = r1;
            a.out_dense = nullptr;
            a.tau = tau;
            a.cand_dist = cand_dist;
            a.cand_row = cand_row;
            a.cand_count = cand_count;
            a.cand_stride = cand_stride;
            ORAMA_TRY(scan(a));
            SelectPlan c;
            c.vals = cand_dist;
            c.idx = cand_row;
            c.stride = cand_stride;
            c.n_dev = cand_count;
            c.n = (uint32_t)cand_stride;
            c.n_hint = 1u << 16;  // lists hold ~k·ln(N/S1) entries on unordered data; the grid-stride loops cover more
            c.q = gq;
            c.k = k;
            c.descending = false;
            c.state = sc->sel_state.as<SelectState>();
            c.keys = sc->sel_keys.as<unsigned long long>();
            c.out_n = out_n;
            if (r1 == n) {  // 3. final reduction: ids + final tie order
                c.id_map = v->row_doc.as<uint64_t>();
                c.out_ids = out_ids;
                c.out_val = out_dist;
            } else {
                c.out_idx = best_row;
                c.out_val = best_dist;
            }
            ORAMA_TRY(launch_select(v->ctx, c, s));
        }
        q0 += gq;
    }
    return ORAMA_OK;
}

// `s_scan` (nullable = same as s): stream the corpus scans run on.  The f16 pipeline interleaves scans and
// selections with dependencies in both directions, so it stays on `s`.
int search_enqueue(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                   const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                   uint32_t* d_out_n, hipStream_t s, hipStream_t s_scan = nullptr, bool two_streams = false) {
    return v->f16() ? search_enqueue_f16(v, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist,
                                         d_out_n, s)
                    : search_enqueue_f32(v, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist,
                                         d_out_n, s, two_streams ? s_scan : s);
}

}  // namespace

namespace orama {
VecSharedLock::VecSharedLock(orama_vec* v) : v_(v) { v_->mu.lock_shared(); }
VecSharedLock::~VecSharedLock() { v_->mu.unlock_shared(); }
orama_ctx* vec_ctx(orama_vec* v) { return v->ctx; }
uint32_t vec_dim(orama_vec* v) { return v->dim; }
uint64_t vec_rows(orama_vec* v) { return v->n_rows; }
int vec_search_enqueue(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s) {
    return search_enqueue(v, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, s);
}
}  // namespace orama

extern "C" {

int orama_vec_create(orama_ctx* ctx, uint32_t dim, int metric, int dtype, uint64_t reserve_rows,
                     orama_vec** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(dim >= 1 && dim <= 65536, "dimensions %u outside [1, 65536]", dim);
    ORAMA_REQUIRE(metric == ORAMA_METRIC_COSINE || metric == ORAMA_METRIC_L2SQ, "unknown metric %d", metric);
    ORAMA_REQUIRE(dtype == ORAMA_DTYPE_F32 || dtype == ORAMA_DTYPE_F16, "unknown dtype %d", dtype);
    if (dtype == ORAMA_DTYPE_F16 && dim > 2048) {
        set_error("f16 storage: dimensions %u > 2048 exceed the LDS query tile", dim);
        return ORAMA_ERR_UNSUPPORTED;
    }
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    orama_vec* v = new (std::nothrow) orama_vec();
    if (!v) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    v->ctx = ctx;
    v->dim = dim;
    v->metric = metric;
    v->dtype = dtype;
    if (reserve_rows) {
        ScratchLease sc(ctx);
        int st = sc.init();
        if (st == ORAMA_OK) st = grow(v, reserve_rows, sc->stream);
        if (st != ORAMA_OK) {
            delete v;
            return st;
        }
    }
    *out = v;
    return ORAMA_OK;
}

void orama_vec_destroy(orama_vec* v) {
    if (!v) return;
    (void)hipSetDevice(v->ctx->device);
    (void)hipDeviceSynchronize();
    delete v;
}

int orama_vec_insert(orama_vec* v, const uint64_t* doc_ids, const float* rows, uint64_t n_rows,
                     uint64_t* accepted) {
    ORAMA_REQUIRE(v, "null handle");
    if (accepted) *accepted = 0;
    if (n_rows == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc_ids && rows, "null input");
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_lock<std::shared_mutex> lk(v->mu);
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    // validate + pack accepted rows into pinned staging, in slabs
    const size_t rb = v->row_bytes();
    const uint64_t slab_rows = std::max<uint64_t>(1, (64ull << 20) / rb);
    uint64_t total_ok = 0;
    for (uint64_t r0 = 0; r0 < n_rows; r0 += slab_rows) {
        const uint64_t cnt = std::min(slab_rows, n_rows - r0);
        ORAMA_TRY(sc->h_in.reserve((size_t)cnt * rb));
        float* stage = sc->h_in.as<float>();
        uint64_t ok = 0;
        const uint64_t first = v->n_rows;
        for (uint64_t i = 0; i < cnt; ++i) {
            const float* x = rows + (r0 + i) * (uint64_t)v->dim;
            if (!row_valid(x, v->dim)) continue;
            memcpy(stage + ok * (uint64_t)v->dim, x, rb);
            v->h_row_doc.push_back(doc_ids[r0 + i]);
            if (v->doc_rows_built) v->doc_rows[doc_ids[r0 + i]].push_back((uint32_t)(first + ok));
            ++ok;
        }
        if (!ok) continue;
        ORAMA_TRY(grow(v, first + ok, s));
        ORAMA_TRY(sc->misc1.reserve((size_t)ok * rb));
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc1.p, stage, (size_t)ok * rb, hipMemcpyHostToDevice, s));
        ORAMA_TRY(store_rows_from_device(v, v->rows.p, v->inv_norm.as<float>(), sc->misc1.as<float>(), first, ok, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(v->row_doc.as<uint64_t>() + first, v->h_row_doc.data() + first,
                                     (size_t)ok * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
        v->n_rows = first + ok;
        v->h_dead.resize((size_t)((v->n_rows + 31) / 32), 0u);
        total_ok += ok;
    }
    if (accepted) *accepted = total_ok;
    return ORAMA_OK;
}

int orama_vec_delete(orama_vec* v, const uint64_t* doc_ids, uint64_t n) {
    ORAMA_REQUIRE(v, "null handle");
    if (n == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc_ids, "null input");
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_lock<std::shared_mutex> lk(v->mu);
    if (!v->doc_rows_built) {
        v->doc_rows.clear();
        v->doc_rows.reserve((size_t)v->n_rows);
        for (uint64_t r = 0; r < v->n_rows; ++r) v->doc_rows[v->h_row_doc[r]].push_back((uint32_t)r);
        v->doc_rows_built = true;
    }
    v->h_dead.resize((size_t)((v->n_rows + 31) / 32), 0u);
    bool changed = false;
    for (uint64_t i = 0; i < n; ++i) {
        auto it = v->doc_rows.find(doc_ids[i]);
        if (it == v->doc_rows.end()) continue;
        for (uint32_t r : it->second) {
            uint32_t& w = v->h_dead[r >> 5];
            const uint32_t bit = 1u << (r & 31);
            if (!(w & bit)) {
                w |= bit;
                ++v->n_dead;
                changed = true;
            }
        }
        v->doc_rows.erase(it);
    }
    if (changed) {
        ORAMA_HIP_TRY(hipMemcpy(v->dead.p, v->h_dead.data(), v->h_dead.size() * sizeof(uint32_t),
                                hipMemcpyHostToDevice));
    }
    return ORAMA_OK;
}

int orama_vec_compact(orama_vec* v, uint64_t version) {
    ORAMA_REQUIRE(v, "null handle");
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_lock<std::shared_mutex> lk(v->mu);
    v->version = version;
    if (v->n_dead == 0) return ORAMA_OK;
    // Re-pack live rows (device gather through a row-index list, in slabs), rebuild the side arrays.
    std::vector<uint64_t> live;
    live.reserve((size_t)(v->n_rows - v->n_dead));
    for (uint64_t r = 0; r < v->n_rows; ++r)
        if (!((v->h_dead[r >> 5] >> (r & 31)) & 1u)) live.push_back(r);
    const uint64_t m = live.size();
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    Arrays a;
    ORAMA_TRY(alloc_arrays(v, m ? m : 1, &a, s));
    std::vector<uint64_t> ndoc_h(m);
    for (uint64_t i = 0; i < m; ++i) ndoc_h[i] = v->h_row_doc[live[i]];
    if (m) {
        const uint64_t slab = std::max<uint64_t>(1, (256ull << 20) / v->row_bytes());
        ORAMA_TRY(sc->misc0.reserve((size_t)m * sizeof(uint64_t)));
        ORAMA_TRY(sc->misc1.reserve((size_t)std::min(slab, m) * v->row_bytes()));
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, live.data(), (size_t)m * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        for (uint64_t i0 = 0; i0 < m; i0 += slab) {
            const uint64_t cnt = std::min(slab, m - i0);
            if (v->f16())
                ORAMA_TRY(launch_f16_gather_rows(v->rows.p, sc->misc0.as<uint64_t>() + i0, cnt, v->dim,
                                                 sc->misc1.as<float>(), s));
            else
                ORAMA_TRY(launch_gather_rows_f32(v->rows.as<float>(), sc->misc0.as<uint64_t>() + i0, cnt, v->dim,
                                                 sc->misc1.as<float>(), s));
            ORAMA_TRY(store_rows_from_device(v, a.rows, reinterpret_cast<float*>(a.norm), sc->misc1.as<float>(),
                                             i0, cnt, s));
        }
        ORAMA_HIP_TRY(hipMemcpyAsync(a.doc, ndoc_h.data(), (size_t)m * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    adopt_arrays(v, a);
    v->n_rows = m;
    v->n_dead = 0;
    v->h_row_doc.swap(ndoc_h);
    v->h_dead.assign((size_t)((m + 31) / 32), 0u);
    v->doc_rows.clear();
    v->doc_rows_built = false;
    return ORAMA_OK;
}

int orama_vec_info(orama_vec* v, orama_vec_info_t* out) {
    ORAMA_REQUIRE(v && out, "null argument");
    std::shared_lock<std::shared_mutex> lk(v->mu);
    out->dimensions = v->dim;
    out->num_rows = v->n_rows;
    out->num_embeddings = v->n_rows - v->n_dead;
    out->pending_ops = v->n_dead;
    out->version = v->version;
    out->hbm_bytes = (uint64_t)(v->rows.cap + v->inv_norm.cap + v->row_doc.cap + v->dead.cap);
    return ORAMA_OK;
}

int orama_vec_search(orama_vec* v, const float* queries, uint32_t q, uint32_t k,
                     const uint64_t* allow_bitmap, uint64_t bitmap_bits, uint64_t* out_ids,
                     float* out_dist, uint32_t* out_n) {
    ORAMA_REQUIRE(v, "null handle");
    ORAMA_REQUIRE(q >= 1 && queries && out_ids && out_dist && out_n, "null argument");
    for (uint32_t i = 0; i < q; ++i) out_n[i] = 0;
    if (k == 0) return ORAMA_OK;  // limit 0: empty result, like the reference's CappedHeap(0)
    ORAMA_REQUIRE(k <= kSelectMaxK, "limit %u exceeds the supported maximum %u", k, kSelectMaxK);
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    std::shared_lock<std::shared_mutex> lk(v->mu);
    if (v->n_rows == 0) return ORAMA_OK;
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const size_t qbytes = (size_t)q * v->dim * sizeof(float);
    ORAMA_TRY(sc->query.reserve(qbytes));
    ORAMA_TRY(sc->h_in.reserve(qbytes));
    memcpy(sc->h_in.p, queries, qbytes);
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->query.p, sc->h_in.p, qbytes, hipMemcpyHostToDevice, s));
    const uint64_t* d_allow = nullptr;
    ORAMA_TRY(resolve_allow(v->ctx, sc.s.get(), allow_bitmap, bitmap_bits, s, &d_allow));
    const size_t nk = (size_t)q * k;
    ORAMA_TRY(sc->out_ids.reserve(nk * 8));
    ORAMA_TRY(sc->out_val.reserve(nk * 4));
    ORAMA_TRY(sc->out_n.reserve((size_t)q * 4));
    ORAMA_TRY(search_enqueue(v, sc.s.get(), sc->query.as<float>(), q, k, d_allow, bitmap_bits,
                             sc->out_ids.as<uint64_t>(), sc->out_val.as<float>(), sc->out_n.as<uint32_t>(), s));
    ORAMA_TRY(sc->h_out.reserve(nk * 12 + (size_t)q * 4));
    char* h = sc->h_out.as<char>();
    ORAMA_HIP_TRY(hipMemcpyAsync(h, sc->out_ids.p, nk * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h + nk * 8, sc->out_val.p, nk * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h + nk * 12, sc->out_n.p, (size_t)q * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    memcpy(out_ids, h, nk * 8);
    memcpy(out_dist, h + nk * 8, nk * 4);
    memcpy(out_n, h + nk * 12, (size_t)q * 4);
    return ORAMA_OK;
}

int orama_vec_search_device(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                            const uint64_t* d_allow_bitmap, uint64_t bitmap_bits,
                            uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                            void* hip_stream) {
    ORAMA_REQUIRE(v, "null handle");
    ORAMA_REQUIRE(q >= 1 && d_queries && d_out_ids && d_out_dist && d_out_n, "null argument");
    ORAMA_REQUIRE(k >= 1 && k <= kSelectMaxK, "limit %u outside [1, %u]", k, kSelectMaxK);
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    std::shared_lock<std::shared_mutex> lk(v->mu);
    Scratch* sc = nullptr;
    {
        std::lock_guard<std::mutex> g(v->dev_mu);
        auto& slot = v->dev_scratch[s];
        if (!slot) slot.reset(new Scratch());
        sc = slot.get();
    }
    return search_enqueue(v, sc, d_queries, q, k, d_allow_bitmap, bitmap_bits, d_out_ids, d_out_dist, d_out_n, s);
}

int orama_vec_search_packed_device2(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                                    const uint64_t* d_allow_bitmap, uint64_t bitmap_bits, void* d_packed_block,
                                    uint32_t* d_out_n, void* scan_stream, void* tail_stream) {
    ORAMA_REQUIRE(v && d_packed_block, "null argument");
    ORAMA_REQUIRE(q >= 1 && d_queries && d_out_n, "null argument");
    ORAMA_REQUIRE(k >= 1 && k <= kSelectMaxK, "limit %u outside [1, %u]", k, kSelectMaxK);
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    hipStream_t s = (hipStream_t)tail_stream, ss = (hipStream_t)scan_stream;
    std::shared_lock<std::shared_mutex> lk(v->mu);
    Scratch* sc = nullptr;
    {
        std::lock_guard<std::mutex> g(v->dev_mu);
        auto& slot = v->dev_scratch[s];
        if (!slot) slot.reset(new Scratch());
        sc = slot.get();
    }
    char* base = reinterpret_cast<char*>(d_packed_block);
    return search_enqueue(v, sc, d_queries, q, k, d_allow_bitmap, bitmap_bits, reinterpret_cast<uint64_t*>(base),
                          reinterpret_cast<float*>(base + (uint64_t)q * k * 8), d_out_n, s, ss, true);
}

int orama_merge_candidates_device(orama_ctx* ctx, const uint64_t* d_ids, const float* d_dist,
                                  uint32_t lists, uint32_t q, uint32_t k, uint64_t* d_out_ids,
                                  float* d_out_dist, uint32_t* d_out_n, void* hip_stream) {
    ORAMA_REQUIRE(ctx && d_ids && d_dist && d_out_ids && d_out_dist, "null argument");
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    return launch_merge_candidates(ctx, d_ids, d_dist, lists, q, k, d_out_ids, d_out_dist, d_out_n,
                                   (hipStream_t)hip_stream);
}

uint64_t orama_packed_block_bytes(uint32_t q, uint32_t k) { return packed_block_bytes(q, k); }

int orama_vec_search_packed_device(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                                   const uint64_t* d_allow_bitmap, uint64_t bitmap_bits,
                                   void* d_packed_block, uint32_t* d_out_n, void* hip_stream) {
    ORAMA_REQUIRE(d_packed_block, "null argument");
    char* base = reinterpret_cast<char*>(d_packed_block);
    return orama_vec_search_device(v, d_queries, q, k, d_allow_bitmap, bitmap_bits,
                                   reinterpret_cast<uint64_t*>(base),
                                   reinterpret_cast<float*>(base + (uint64_t)q * k * 8), d_out_n, hip_stream);
}

int orama_merge_packed_device(orama_ctx* ctx, const void* d_packed_blocks, uint32_t lists, uint32_t q,
                              uint32_t k, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                              void* hip_stream) {
    ORAMA_REQUIRE(ctx && d_packed_blocks && d_out_ids && d_out_dist, "null argument");
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    return launch_merge_packed(ctx, d_packed_blocks, lists, q, k, d_out_ids, d_out_dist, d_out_n,
                               (hipStream_t)hip_stream);
}

int orama_vec_fill_synthetic(orama_vec* v, uint64_t n_rows, uint64_t seed, uint64_t first_doc_id) {
    ORAMA_REQUIRE(v, "null handle");
    if (n_rows == 0) return ORAMA_OK;
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    std::unique_lock<std::shared_mutex> lk(v->mu);
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t first = v->n_rows;
    ORAMA_TRY(grow(v, first + n_rows, s));
    if (v->f16()) {
        // generate f32 slabs in scratch with the SAME generator (row index = store row), then quantise
        const uint64_t slab = std::max<uint64_t>(32, ((512ull << 20) / v->row_bytes()) & ~31ull);
        ORAMA_TRY(sc->misc1.reserve((size_t)std::min(slab, n_rows) * v->row_bytes()));
        for (uint64_t r0 = 0; r0 < n_rows; r0 += slab) {
            const uint64_t cnt = std::min(slab, n_rows - r0);
            // rows are generated at virtual positions first+r0.. by offsetting the base pointer
            float* virt = sc->misc1.as<float>() - (first + r0) * (uint64_t)v->dim;
            ORAMA_TRY(launch_synth_fill_f32(virt, first + r0, cnt, v->dim, seed, s));
            ORAMA_TRY(store_rows_from_device(v, v->rows.p, v->inv_norm.as<float>(), sc->misc1.as<float>(),
                                             first + r0, cnt, s));
        }
    } else {
        ORAMA_TRY(launch_synth_fill_f32(v->rows.as<float>(), first, n_rows, v->dim, seed, s));
        if (v->metric == ORAMA_METRIC_COSINE)
            ORAMA_TRY(launch_row_inv_norm_f32(v->rows.as<float>(), first, n_rows, v->dim, v->inv_norm.as<float>(), s));
    }
    ORAMA_TRY(launch_iota_u64(v->row_doc.as<uint64_t>() + first, n_rows, first_doc_id, s));
    v->h_row_doc.resize((size_t)(first + n_rows));
    for (uint64_t i = 0; i < n_rows; ++i) v->h_row_doc[first + i] = first_doc_id + i;
    if (v->doc_rows_built)
        for (uint64_t i = 0; i < n_rows; ++i) v->doc_rows[first_doc_id + i].push_back((uint32_t)(first + i));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    v->n_rows = first + n_rows;
    v->h_dead.resize((size_t)((v->n_rows + 31) / 32), 0u);
    return ORAMA_OK;
}

int orama_vec_get_rows(orama_vec* v, const uint64_t* row_idx, uint64_t n, float* out_rows,
                       uint64_t* out_doc_ids) {
    ORAMA_REQUIRE(v, "null handle");
    if (n == 0) return ORAMA_OK;
    ORAMA_REQUIRE(row_idx && out_rows, "null argument");
    ORAMA_HIP_TRY(hipSetDevice(v->ctx->device));
    std::shared_lock<std::shared_mutex> lk(v->mu);
    for (uint64_t i = 0; i < n; ++i)
        ORAMA_REQUIRE(row_idx[i] < v->n_rows, "row %llu out of range", (unsigned long long)row_idx[i]);
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    ORAMA_TRY(sc->misc0.reserve((size_t)n * 8));
    ORAMA_TRY(sc->misc1.reserve((size_t)n * v->row_bytes()));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, row_idx, (size_t)n * 8, hipMemcpyHostToDevice, s));
    if (v->f16())
        ORAMA_TRY(launch_f16_gather_rows(v->rows.p, sc->misc0.as<uint64_t>(), n, v->dim, sc->misc1.as<float>(), s));
    else
        ORAMA_TRY(launch_gather_rows_f32(v->rows.as<float>(), sc->misc0.as<uint64_t>(), n, v->dim,
                                         sc->misc1.as<float>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_rows, sc->misc1.p, (size_t)n * v->row_bytes(), hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    if (out_doc_ids)
        for (uint64_t i = 0; i < n; ++i) out_doc_ids[i] = v->h_row_doc[row_idx[i]];
    return ORAMA_OK;
}

}  // extern "C"
