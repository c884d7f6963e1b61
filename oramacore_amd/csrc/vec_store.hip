// vec_store.hip — HBM-resident vector store behind the orama_vec_* entry points.
// Mirrors what oramacore_fields::embedding::EmbeddingStorage provides to
// EmbeddingFieldStorage (src/collection_manager/sides/read/index/embedding_field.rs:63-320):
// insert (N rows per doc) / delete (tombstone) / compact / info / search.
//
// HBM layout: one contiguous row-major f32 matrix [rows][dim] (rows 16-B aligned when dim % 4 == 0),
// plus per row: 1/|x| (f32, cosine), DocumentId (u64) and a tombstone bit.  Sized for 288 GB: the
// matrix is one allocation (30.7 GB for 10 M x 768) that grows geometrically.
#include <cmath>
#include <shared_mutex>
#include <unordered_map>

#include "common.hpp"
#include "select.hpp"
#include "vec_kernels.hpp"

using namespace orama;

struct orama_vec {
    orama_ctx* ctx = nullptr;
    uint32_t dim = 0;
    int metric = ORAMA_METRIC_COSINE;
    int dtype = ORAMA_DTYPE_F32;
    std::shared_mutex mu;  // searches: shared; insert/delete/compact: exclusive

    DevBuf rows;      // cap_rows x dim f32
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

    size_t row_bytes() const { return (size_t)dim * sizeof(float); }
};

namespace {

int grow(orama_vec* v, uint64_t need_rows, hipStream_t s) {
    if (need_rows <= v->cap_rows) return ORAMA_OK;
    ORAMA_REQUIRE(need_rows < 0xffffffffull, "vector store limited to 2^32-1 rows");
    uint64_t cap = v->cap_rows ? v->cap_rows : 1024;
    while (cap < need_rows) cap += cap / 2 + 1024;
    void *nrows = nullptr, *nnorm = nullptr, *ndoc = nullptr, *ndead = nullptr;
    const size_t dead_words = (size_t)((cap + 31) / 32);
    ORAMA_HIP_TRY(hipMalloc(&nrows, (size_t)cap * v->row_bytes()));
    ORAMA_HIP_TRY(hipMalloc(&nnorm, (size_t)cap * sizeof(float)));
    ORAMA_HIP_TRY(hipMalloc(&ndoc, (size_t)cap * sizeof(uint64_t)));
    ORAMA_HIP_TRY(hipMalloc(&ndead, dead_words * sizeof(uint32_t)));
    ORAMA_HIP_TRY(hipMemsetAsync(ndead, 0, dead_words * sizeof(uint32_t), s));
    if (v->n_rows) {
        ORAMA_HIP_TRY(hipMemcpyAsync(nrows, v->rows.p, (size_t)v->n_rows * v->row_bytes(),
                                     hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(nnorm, v->inv_norm.p, (size_t)v->n_rows * sizeof(float),
                                     hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(ndoc, v->row_doc.p, (size_t)v->n_rows * sizeof(uint64_t),
                                     hipMemcpyDeviceToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(ndead, v->dead.p, (size_t)((v->n_rows + 31) / 32) * 4,
                                     hipMemcpyDeviceToDevice, s));
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    v->rows.release();
    v->inv_norm.release();
    v->row_doc.release();
    v->dead.release();
    v->rows.p = nrows;
    v->rows.cap = (size_t)cap * v->row_bytes();
    v->inv_norm.p = nnorm;
    v->inv_norm.cap = (size_t)cap * sizeof(float);
    v->row_doc.p = ndoc;
    v->row_doc.cap = (size_t)cap * sizeof(uint64_t);
    v->dead.p = ndead;
    v->dead.cap = dead_words * sizeof(uint32_t);
    v->cap_rows = cap;
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

// Enqueue scan + top-k for `q` queries already resident at d_queries; results into device buffers.
int search_enqueue(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                   const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids,
                   uint32_t* d_out_rows, float* d_out_dist, uint32_t* d_out_n, hipStream_t s) {
    const uint64_t n = v->n_rows;
    // queries are processed in groups so that the dense distance buffer stays <= ~1 GiB
    uint32_t group = 1;
    if (n > 0) {
        uint64_t g = (1ull << 28) / n;  // 2^28 floats
        group = (uint32_t)(g < 1 ? 1 : (g > q ? q : g));
    } else {
        group = q;
    }
    ORAMA_TRY(sc->dist.reserve((size_t)group * (size_t)(n ? n : 1) * sizeof(float)));
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState) * (size_t)group));
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)group * k));
    for (uint32_t q0 = 0; q0 < q; q0 += group) {
        const uint32_t gq = (q - q0) < group ? (q - q0) : group;
        for (uint32_t j = 0; j < gq; ++j) {
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
            ORAMA_TRY(launch_vec_scan_f32(v->ctx, a, s));
        }
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
        p.out_idx = d_out_rows ? d_out_rows + (size_t)q0 * k : nullptr;
        p.out_ids = d_out_ids + (size_t)q0 * k;
        p.out_val = d_out_dist + (size_t)q0 * k;
        p.out_n = d_out_n + q0;
        ORAMA_TRY(launch_select(v->ctx, p, s));
    }
    return ORAMA_OK;
}

}  // namespace

extern "C" {

int orama_vec_create(orama_ctx* ctx, uint32_t dim, int metric, int dtype, uint64_t reserve_rows,
                     orama_vec** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(dim >= 1 && dim <= 65536, "dimensions %u outside [1, 65536]", dim);
    ORAMA_REQUIRE(metric == ORAMA_METRIC_COSINE || metric == ORAMA_METRIC_L2SQ, "unknown metric %d",
                  metric);
    if (dtype != ORAMA_DTYPE_F32) {
        set_error("dtype %d: only f32 storage is implemented in this build", dtype);
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
        ORAMA_HIP_TRY(hipMemcpyAsync(v->rows.as<char>() + (size_t)first * rb, stage, (size_t)ok * rb,
                                     hipMemcpyHostToDevice, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(v->row_doc.as<uint64_t>() + first, v->h_row_doc.data() + first,
                                     (size_t)ok * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        if (v->metric == ORAMA_METRIC_COSINE)
            ORAMA_TRY(launch_row_inv_norm_f32(v->rows.as<float>(), first, ok, v->dim,
                                              v->inv_norm.as<float>(), s));
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
    // Re-pack live rows (device gather through a row-index list), rebuild the side arrays.
    std::vector<uint64_t> live;
    live.reserve((size_t)(v->n_rows - v->n_dead));
    for (uint64_t r = 0; r < v->n_rows; ++r)
        if (!((v->h_dead[r >> 5] >> (r & 31)) & 1u)) live.push_back(r);
    const uint64_t m = live.size();
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    void *nrows = nullptr, *nnorm = nullptr, *ndoc = nullptr, *ndead = nullptr;
    const uint64_t cap = m ? m : 1;
    const size_t dead_words = (size_t)((cap + 31) / 32);
    ORAMA_HIP_TRY(hipMalloc(&nrows, (size_t)cap * v->row_bytes()));
    ORAMA_HIP_TRY(hipMalloc(&nnorm, (size_t)cap * sizeof(float)));
    ORAMA_HIP_TRY(hipMalloc(&ndoc, (size_t)cap * sizeof(uint64_t)));
    ORAMA_HIP_TRY(hipMalloc(&ndead, dead_words * sizeof(uint32_t)));
    ORAMA_HIP_TRY(hipMemsetAsync(ndead, 0, dead_words * sizeof(uint32_t), s));
    std::vector<uint64_t> ndoc_h(m);
    for (uint64_t i = 0; i < m; ++i) ndoc_h[i] = v->h_row_doc[live[i]];
    if (m) {
        ORAMA_TRY(sc->misc0.reserve((size_t)m * sizeof(uint64_t)));
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, live.data(), (size_t)m * sizeof(uint64_t),
                                     hipMemcpyHostToDevice, s));
        ORAMA_TRY(launch_gather_rows_f32(v->rows.as<float>(), sc->misc0.as<uint64_t>(), m, v->dim,
                                         (float*)nrows, s));
        ORAMA_HIP_TRY(hipMemcpyAsync(ndoc, ndoc_h.data(), (size_t)m * sizeof(uint64_t),
                                     hipMemcpyHostToDevice, s));
        if (v->metric == ORAMA_METRIC_COSINE)
            ORAMA_TRY(launch_row_inv_norm_f32((const float*)nrows, 0, m, v->dim, (float*)nnorm, s));
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    v->rows.release();
    v->inv_norm.release();
    v->row_doc.release();
    v->dead.release();
    v->rows.p = nrows;
    v->rows.cap = (size_t)cap * v->row_bytes();
    v->inv_norm.p = nnorm;
    v->inv_norm.cap = (size_t)cap * sizeof(float);
    v->row_doc.p = ndoc;
    v->row_doc.cap = (size_t)cap * sizeof(uint64_t);
    v->dead.p = ndead;
    v->dead.cap = dead_words * sizeof(uint32_t);
    v->cap_rows = cap;
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
    if (allow_bitmap) {
        const size_t words = (size_t)((bitmap_bits + 63) / 64);
        ORAMA_TRY(sc->bitmap.reserve(std::max<size_t>(8, words * 8)));
        if (words)
            ORAMA_HIP_TRY(hipMemcpyAsync(sc->bitmap.p, allow_bitmap, words * 8, hipMemcpyHostToDevice, s));
        d_allow = sc->bitmap.as<uint64_t>();
    }
    const size_t nk = (size_t)q * k;
    ORAMA_TRY(sc->out_ids.reserve(nk * 8));
    ORAMA_TRY(sc->out_val.reserve(nk * 4));
    ORAMA_TRY(sc->out_n.reserve((size_t)q * 4));
    ORAMA_TRY(search_enqueue(v, sc.s.get(), sc->query.as<float>(), q, k, d_allow, bitmap_bits,
                             sc->out_ids.as<uint64_t>(), nullptr, sc->out_val.as<float>(),
                             sc->out_n.as<uint32_t>(), s));
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
    return search_enqueue(v, sc, d_queries, q, k, d_allow_bitmap, bitmap_bits, d_out_ids, nullptr,
                          d_out_dist, d_out_n, s);
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
                                   reinterpret_cast<float*>(base + (uint64_t)q * k * 8), d_out_n,
                                   hip_stream);
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
    ORAMA_TRY(launch_synth_fill_f32(v->rows.as<float>(), first, n_rows, v->dim, seed, s));
    ORAMA_TRY(launch_iota_u64(v->row_doc.as<uint64_t>() + first, n_rows, first_doc_id, s));
    if (v->metric == ORAMA_METRIC_COSINE)
        ORAMA_TRY(launch_row_inv_norm_f32(v->rows.as<float>(), first, n_rows, v->dim,
                                          v->inv_norm.as<float>(), s));
    v->h_row_doc.resize((size_t)(first + n_rows));
    for (uint64_t i = 0; i < n_rows; ++i) v->h_row_doc[first + i] = first_doc_id + i;
    if (v->doc_rows_built)
        for (uint64_t i = 0; i < n_rows; ++i)
            v->doc_rows[first_doc_id + i].push_back((uint32_t)(first + i));
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
    ORAMA_TRY(launch_gather_rows_f32(v->rows.as<float>(), sc->misc0.as<uint64_t>(), n, v->dim,
                                     sc->misc1.as<float>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_rows, sc->misc1.p, (size_t)n * v->row_bytes(),
                                 hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    if (out_doc_ids)
        for (uint64_t i = 0; i < n; ++i) out_doc_ids[i] = v->h_row_doc[row_idx[i]];
    return ORAMA_OK;
}

}  // extern "C"
