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
#include <atomic>
#include <cmath>
#include <shared_mutex>

#include <cstdlib>

#include "common.hpp"
#include "device_utils.hpp"
#include "select.hpp"
#include "vec_f16.hpp"
#include "vec_f32_mfma.hpp"
#include "vec_internal.hpp"
#include "stage.hpp"
#include "vec_kernels.hpp"

using namespace orama;

// Plain device-side copy / zero-fill.  Everything that WRITES into (or reads out of) the in-place-growing arrays below
// goes through these kernels, never through hipMemcpyAsync / hipMemsetAsync: on this stack the copy engines do not
// reliably see memory that was mapped with hipMemMap moments ago (rows and DocumentIds arrived corrupted in ~1 of 50
// runs of the GPU suite), while kernels — which use the shader page tables the mapping call updates — always do.
namespace {
__global__ void copy_bytes_kernel(char* __restrict__ dst, const char* __restrict__ src, size_t bytes) {
    const size_t n16 = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0 ? bytes / 16 : 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (size_t i = n16 * 16 + t; i < bytes; i += stride) dst[i] = src[i];
}
__global__ void zero_bytes_kernel(char* __restrict__ dst, size_t bytes) {
    const size_t n16 = ((uintptr_t)dst & 15) == 0 ? bytes / 16 : 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 z = {0u, 0u, 0u, 0u};
    for (size_t i = t; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = z;
    for (size_t i = n16 * 16 + t; i < bytes; i += stride) dst[i] = 0;
}
int launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return ORAMA_OK;
    const uint32_t blocks = (uint32_t)std::min<size_t>(4096, std::max<size_t>(1, bytes / (256 * 16)));
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(blocks), dim3(256), 0, s, static_cast<char*>(dst), static_cast<const char*>(src), bytes);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
int launch_zero_bytes(void* dst, size_t bytes, hipStream_t s) {
    if (!bytes) return ORAMA_OK;
    const uint32_t blocks = (uint32_t)std::min<size_t>(4096, std::max<size_t>(1, bytes / (256 * 16)));
    hipLaunchKernelGGL(zero_bytes_kernel, dim3(blocks), dim3(256), 0, s, static_cast<char*>(dst), bytes);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
}  // namespace

// A device array that grows IN PLACE: one virtual range reserved up front (hipMemAddressReserve), physical chunks
// mapped behind it on demand (hipMemCreate / hipMemMap / hipMemSetAccess — profiles/r02_vmm_probe.log: 1 GiB mapped in
// 0.02-0.1 ms, same streaming bandwidth as hipMalloc).  The base address never changes while the array fits its
// reservation, so a scan that started before an insert keeps reading valid memory and appends need no copy and no
// reader exclusion.  Outgrowing the reservation (4x the initial need) re-maps the SAME physical chunks into a bigger
// range — still no copy, but the base moves, so that step runs under the store's exclusive lock.
struct GrowBuf {
    char* base = nullptr;
    size_t va_bytes = 0;   // reserved virtual range
    size_t mapped = 0;     // bytes backed by physical memory (multiple of chunk)
    size_t chunk = 0;
    bool vmm = false;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> handle_bytes;
    char* old_base = nullptr;  // non-VMM growth: the previous allocation, alive until drop_old()

    void drop_old() {
        if (old_base) (void)hipFree(old_base);
        old_base = nullptr;
    }

    GrowBuf() = default;
    GrowBuf(const GrowBuf&) = delete;
    GrowBuf& operator=(const GrowBuf&) = delete;
    ~GrowBuf() { release(); }

    static bool vmm_supported(int device) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeVirtualMemoryManagementSupported, device) != hipSuccess) return false;
        // OPT-IN (ORAMA_VMM=1).  The mechanism works (scripts/micro/vmm_probe*.hip) and removes every copy from growth,
        // but with many stores created and destroyed in one process — virtual ranges freed and re-reserved over different
        // physical memory — kernels on this stack (ROCm 7.2 / 7.0 runtimes alike) intermittently read and wrote through
        // stale translations: 8 of 61 vector tests failed at random with zeroed / foreign DocumentIds
        // (profiles/r02_vmm_store_failures.md).  Until that is understood the default is allocate + copy BESIDE the
        // running searches (see reserve_rows_locked), which blocks readers only for the pointer swap.
        static const bool on = [] {
            const char* e = std::getenv("ORAMA_VMM");
            return e && std::atoi(e) != 0;
        }();
        return v != 0 && on;
    }
    hipMemAllocationProp prop() const {
        hipMemAllocationProp p = {};
        p.type = hipMemAllocationTypePinned;
        p.location.type = hipMemLocationTypeDevice;
        p.location.id = device;
        return p;
    }
    // piece by piece, exactly as mapped (one hipMemUnmap over a range that spans several allocations is not relied upon)
    void unmap_all() {
        size_t off = 0;
        for (size_t i = 0; i < handles.size(); ++i) {
            (void)hipMemUnmap(base + off, handle_bytes[i]);
            off += handle_bytes[i];
        }
    }
    void release() {
        drop_old();
        if (!base) return;
        if (vmm) {
            unmap_all();
            for (auto h : handles) (void)hipMemRelease(h);
            (void)hipMemAddressFree(base, va_bytes);
        } else {
            (void)hipFree(base);
        }
        handles.clear();
        base = nullptr;
        va_bytes = mapped = 0;
    }
    // the last GiB of a reservation is slack for the piece rounding below
    static constexpr size_t kSlack = (size_t)1 << 30;
    bool needs_move(size_t bytes) const { return vmm ? bytes + kSlack > va_bytes && bytes > mapped : bytes > mapped; }
    // Make [0, bytes) usable.  New memory is zeroed on `s` when `zero_new`.  When needs_move(bytes) the base changes:
    // the caller must hold the store's exclusive lock (no scan may be reading the old range).
    int ensure(int dev, size_t bytes, size_t va_hint, bool zero_new, hipStream_t s) {
        device = dev;
        if (bytes <= mapped) return ORAMA_OK;
        if (!base) {
            vmm = vmm_supported(dev);
            chunk = 2u << 20;  // 2 MiB granules; big arrays take whole multiples per step below
        }
        if (!vmm) {  // fallback: allocate-and-copy (the pre-VMM behaviour), geometric growth
            size_t cap = mapped ? mapped : 4096;
            while (cap < bytes) cap += cap / 2 + 4096;
            char* nb = nullptr;
            ORAMA_HIP_TRY(hipMalloc(&nb, cap));
            if (mapped) ORAMA_HIP_TRY(hipMemcpyAsync(nb, base, mapped, hipMemcpyDeviceToDevice, s));
            if (zero_new) ORAMA_HIP_TRY(hipMemsetAsync(nb + mapped, 0, cap - mapped, s));
            ORAMA_HIP_TRY(hipStreamSynchronize(s));
            old_base = base;  // released by the caller once no reader can hold it (drop_old)
            base = nb;
            mapped = va_bytes = cap;
            return ORAMA_OK;
        }
        auto round_up = [&](size_t x) { return (x + chunk - 1) / chunk * chunk; };
        if (bytes + kSlack > va_bytes) {  // (re-)reserve: 4x head room, the existing physical chunks move to the new range
            size_t want = round_up(std::max(std::max(bytes * 4, va_hint), (size_t)64 << 20)) + kSlack;
            void* nb = nullptr;
            ORAMA_HIP_TRY(hipMemAddressReserve(&nb, want, chunk, nullptr, 0));
            if (base) {
                unmap_all();
                size_t off = 0;
                hipMemAccessDesc acc = {};
                acc.location = prop().location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                for (size_t i = 0; i < handles.size(); ++i) {
                    const size_t sz = handle_bytes[i];
                    ORAMA_HIP_TRY(hipMemMap(static_cast<char*>(nb) + off, sz, 0, handles[i], 0));
                    off += sz;
                }
                if (off) ORAMA_HIP_TRY(hipMemSetAccess(nb, off, &acc, 1));
                (void)hipMemAddressFree(base, va_bytes);
            }
            base = static_cast<char*>(nb);
            va_bytes = want;
        }
        // map what is missing in physical pieces of exactly 1 GiB, 64 MiB or 2 MiB: hipMemSetAccess on this stack
        // rejects some other sizes ("invalid argument" for e.g. 313 x 2 MiB), these three are measured good
        // (scripts/micro/vmm_probe2.hip).  Over-allocation is below one piece of the size class.
        size_t need = round_up(bytes) - mapped;
        hipMemAllocationProp p = prop();
        hipMemAccessDesc acc = {};
        acc.location = p.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        size_t zero_from = ~(size_t)0;
        while (need) {
            const size_t piece = need >= ((size_t)1 << 30) ? (size_t)1 << 30 : (need > ((size_t)32 << 20) ? (size_t)64 << 20 : chunk);
            hipMemGenericAllocationHandle_t h;
            ORAMA_HIP_TRY(hipMemCreate(&h, piece, &p, 0));
            hipError_t e = hipMemMap(base + mapped, piece, 0, h, 0);
            if (e != hipSuccess) {
                (void)hipMemRelease(h);
                ORAMA_HIP_TRY(e);
            }
            handles.push_back(h);
            handle_bytes.push_back(piece);
            if (zero_new) zero_from = std::min(zero_from, mapped);
            mapped += piece;
            need -= std::min(need, piece);
        }
        // ONE hipMemSetAccess over everything mapped so far: per-piece calls fail on this stack for a smaller piece that
        // follows >= ~14 one-GiB pieces ("invalid argument"; scripts/micro/vmm_probe2.hip reproduces it), the whole-range
        // form does not.  Re-granting access to the part that was already accessible is a no-op for running kernels.
        hipError_t ea = hipMemSetAccess(base, mapped, &acc, 1);
        if (ea != hipSuccess) {
            set_error("hipMemSetAccess(%p, %zu bytes of a %zu-byte reservation, %zu pieces): %s", (void*)base, mapped, va_bytes,
                      handles.size(), hipGetErrorString(ea));
            return ORAMA_ERR_HIP;
        }
        if (zero_from != ~(size_t)0) ORAMA_TRY(launch_zero_bytes(base + zero_from, mapped - zero_from, s));
        return ORAMA_OK;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(base); }
};

struct orama_vec {
    orama_ctx* ctx = nullptr;
    uint32_t dim = 0;
    int metric = ORAMA_METRIC_COSINE;
    int dtype = ORAMA_DTYPE_F32;
    // Locking (the reference's insert(&self) runs beside searches, index/mod.rs:1436,1688-1698; deletes and compaction are
    // exclusive there, index/mod.rs:1346-1424):
    //   mu        shared: searches, inserts, deletes      exclusive: compaction, re-reservation of an array
    //   write_mu  one writer (insert / delete / compact) at a time
    // Inserts write rows BEYOND the published row count and publish it last, so a search works on the snapshot it read at
    // its start (arc-swap semantics) and is never blocked by an insert.
    std::shared_mutex mu;
    std::mutex write_mu;

    GrowBuf rows;      // f32: rows x dim; f16: tiles(rows) x tile_bytes
    GrowBuf inv_norm;  // f32 per row (+ one block tile of padding for the f16 kernels)
    GrowBuf row_doc;   // u64 per row
    GrowBuf dead;      // bit per row (u32 words), zero = live
    std::atomic<uint64_t> n_rows{0};  // published rows
    std::atomic<uint64_t> n_dead{0};
    uint64_t cap_rows = 0;            // rows the arrays can hold without growing
    uint64_t version = 0;

    // Two-stage exact search (ORAMA_DTYPE_F32_SHADOW16): an fp16 copy of the same rows in the same order (an ordinary
    // fp16 store owned by this one) proposes candidates, the fp32 rows decide.  Every mutation goes to the shadow first,
    // then here, under composite_mu; `shadow_ok` drops to false for good when a row is inserted whose fp16 image
    // carries no error bound (tiny norm / element beyond the fp16 range): searches then scan the fp32 rows.
    std::unique_ptr<orama_vec> shadow;
    std::mutex composite_mu;
    std::atomic<bool> shadow_ok{true};
    std::atomic<uint64_t> two_stage_queries{0}, two_stage_fallbacks{0};
    std::atomic<uint64_t> mfma_batch_queries{0};  // queries answered through K1m / K1x + rerank (search_enqueue_f32_batch)
    // plain fp32 stores: every row accepted so far has an fp16 image with a usable error bound (row_shadow_safe: |x_i| < 6e4,
    // |x|^2 >= 1e-4) — what K1x, the convert-in-registers candidate scan, needs; drops to false for good at the first row that
    // does not (batches then take K1m, f32 on the matrix cores)
    std::atomic<bool> f16_safe{true};

    // scratch for the device-pointer entry point, one per caller stream
    std::mutex dev_mu;
    std::map<hipStream_t, std::unique_ptr<Scratch>> dev_scratch;
    std::map<hipStream_t, std::unique_ptr<Scratch>> dev_scratch2;  // shadow stores: the candidate stage's set (two_stage_search)

    bool f16() const { return dtype == ORAMA_DTYPE_F16; }
    size_t row_bytes() const { return (size_t)dim * sizeof(float); }  // f32 row (host side / f32 store)
    // (f32: one 32-row tile of slack behind the rows and their norms — K1m reads a partial last tile whole and masks the
    // rows beyond the published count in its epilogue, vec_f32_mfma.hpp)
    size_t matrix_bytes(uint64_t rows_) const {
        return f16() ? (size_t)(f16_tiles(rows_) * f16_tile_bytes(dim)) : (size_t)(rows_ + kF32MfmaSlackRows) * row_bytes();
    }
    size_t norm_bytes(uint64_t rows_) const {
        // f16: K2c / K2d fetch the inverse norms of a block tile with one 1-KiB DMA — pad to whole block tiles
        return (size_t)(f16() ? ((rows_ + 255) & ~255ull) + 256 : rows_ + kF32MfmaSlackRows) * sizeof(float);
    }
};

namespace {

// What one search reads: pointers + the row count published when it started.
struct View {
    void* rows = nullptr;
    float* inv_norm = nullptr;
    uint64_t* row_doc = nullptr;
    uint32_t* dead = nullptr;  // nullptr when no row was deleted at snapshot time
    uint64_t n_rows = 0;
};
View snapshot(orama_vec* v) {
    View w;
    w.n_rows = v->n_rows.load(std::memory_order_acquire);
    w.rows = v->rows.base;
    w.inv_norm = v->inv_norm.as<float>();
    w.row_doc = v->row_doc.as<uint64_t>();
    w.dead = v->n_dead.load(std::memory_order_acquire) ? v->dead.as<uint32_t>() : nullptr;
    return w;
}

bool grow_needs_move(orama_vec* v, uint64_t need_rows) {
    uint64_t cap = v->f16() ? ((need_rows + 31) & ~31ull) : need_rows;
    return v->rows.needs_move(std::max<size_t>(256, v->matrix_bytes(cap))) || v->inv_norm.needs_move(v->norm_bytes(cap)) ||
           v->row_doc.needs_move((size_t)cap * 8) || v->dead.needs_move((size_t)((cap + 31) / 32) * 4);
}

// Capacity for `need_rows` rows.  New memory of the f16 matrix / norms / tombstones is zeroed (padding rows of
// partial tiles must read as zeros; zero = live).  Call with the exclusive lock held iff grow_needs_move().
int grow(orama_vec* v, uint64_t need_rows, hipStream_t s) {
    if (need_rows <= v->cap_rows) return ORAMA_OK;
    ORAMA_SUPPORT(need_rows < 0xfffffff0ull, "vector store limited to 2^32-16 rows");
    uint64_t cap = need_rows;
    if (v->f16()) cap = (cap + 31) & ~31ull;
    const int dev = v->ctx->device;
    const uint64_t hint_rows = std::max<uint64_t>(cap * 4, 1u << 20);
    ORAMA_TRY(v->rows.ensure(dev, std::max<size_t>(256, v->matrix_bytes(cap)), v->matrix_bytes(hint_rows), v->f16(), s));
    ORAMA_TRY(v->inv_norm.ensure(dev, v->norm_bytes(cap), v->norm_bytes(hint_rows), v->f16(), s));
    ORAMA_TRY(v->row_doc.ensure(dev, (size_t)cap * 8, (size_t)hint_rows * 8, false, s));
    ORAMA_TRY(v->dead.ensure(dev, (size_t)((cap + 31) / 32) * 4, (size_t)((hint_rows + 31) / 32) * 4, true, s));
    // capacity = what the smallest array now holds
    uint64_t c = v->row_doc.mapped / 8;
    c = std::min<uint64_t>(c, v->dead.mapped / 4 * 32);
    if (v->f16()) {
        c = std::min<uint64_t>(c, (uint64_t)(v->rows.mapped / f16_tile_bytes(v->dim)) * 32);
        const uint64_t nf = v->inv_norm.mapped / 4;
        c = std::min<uint64_t>(c, nf > 512 ? (nf - 256) & ~255ull : 0);
    } else {
        const uint64_t rm = v->rows.mapped / v->row_bytes(), nm = v->inv_norm.mapped / 4;
        c = std::min<uint64_t>(c, rm > kF32MfmaSlackRows ? rm - kF32MfmaSlackRows : 0);
        c = std::min<uint64_t>(c, nm > kF32MfmaSlackRows ? nm - kF32MfmaSlackRows : 0);
    }
    ORAMA_REQUIRE(c >= need_rows, "internal: capacity %llu below the %llu rows asked for", (unsigned long long)c,
                  (unsigned long long)need_rows);
    v->cap_rows = c;
    return ORAMA_OK;
}

// ---- small device helpers of the mutation path
// tombstone every row whose DocumentId is in the (sorted) list; *newly counts rows that were live before
__global__ void mark_deleted_kernel(const uint64_t* __restrict__ row_doc, uint64_t n_rows, const uint64_t* __restrict__ docs,
                                    uint32_t n_docs, uint32_t* __restrict__ dead, unsigned long long* __restrict__ newly) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t d = row_doc[r];
        uint32_t lo = 0, hi = n_docs;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (docs[mid] < d) lo = mid + 1; else hi = mid;
        }
        if (lo < n_docs && docs[lo] == d) {
            const uint32_t bit = 1u << (r & 31);
            const uint32_t old = atomicOr(&dead[r >> 5], bit);
            if (!(old & bit)) atomicAdd(newly, 1ull);
        }
    }
}
__global__ void gather_u64_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, uint64_t n,
                                  uint64_t* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = src[idx[i]];
}
// ---- the two-stage plan's fallback without the host (device path: orama_vec_search_*device*, sessions)
// A query the fp16 shadow cannot serve (the test its rows pass at insert: |q|^2 >= 1e-4, every |q_i| < 6e4 — NaN fails both)
// flags itself "not proven"; one wave per query.
__global__ void query_shadow_unsafe_kernel(const float* __restrict__ queries, uint32_t q, uint32_t dim, uint32_t* __restrict__ flag) {
    const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= q) return;
    const int lane = threadIdx.x & 63;
    const float* x = queries + (size_t)j * dim;
    float n2 = 0.0f, mx = 0.0f;
    bool nan = false;
    for (uint32_t i = lane; i < dim; i += 64) {
        const float v = x[i];
        n2 = fmaf(v, v, n2);
        mx = fmaxf(mx, fabsf(v));
        nan |= v != v;
    }
    n2 = wave_sum(n2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const bool any_nan = __ballot(nan) != 0ull;
    if (lane == 0 && (any_nan || !(n2 >= 1e-4f && mx < 6.0e4f))) flag[j] = 1u;
}
// pick[0 .. *n_pick) = the flagged queries in ascending order (one workgroup of 256 threads)
// (`round_n[r]` = how many of the picked queries round r answers: the fallback keeps list sets for `sets` queries and is
// enqueued ceil(q / sets) times — a round with nothing to do ends at once)
__global__ void pick_flagged_kernel(const uint32_t* __restrict__ flag, uint32_t q, uint32_t* __restrict__ pick,
                                    uint32_t* __restrict__ n_pick, uint32_t sets, uint32_t rounds, uint32_t* __restrict__ round_n) {
    __shared__ uint32_t cnt[256];
    const uint32_t per = (q + 255u) / 256u;
    const uint32_t lo = min(q, threadIdx.x * per), hi = min(q, lo + per);
    uint32_t c = 0;
    for (uint32_t j = lo; j < hi; ++j) c += flag[j] != 0u;
    cnt[threadIdx.x] = c;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t t = 0; t < threadIdx.x; ++t) base += cnt[t];
    for (uint32_t j = lo; j < hi; ++j)
        if (flag[j] != 0u) pick[base++] = j;
    if (threadIdx.x == 255) {
        *n_pick = base;
        for (uint32_t r = 0; r < rounds; ++r) round_n[r] = base > r * sets ? min(sets, base - r * sets) : 0u;
    }
}
// the answers of the picked queries go to their own slots (workgroup f = the f-th picked query)
__global__ void scatter_picked_kernel(const uint32_t* __restrict__ pick, const uint32_t* __restrict__ n_pick, uint32_t k,
                                      const uint64_t* __restrict__ ids, const float* __restrict__ dist, const uint32_t* __restrict__ n,
                                      uint64_t* __restrict__ out_ids, float* __restrict__ out_dist, uint32_t* __restrict__ out_n) {
    const uint32_t f = blockIdx.x;
    if (f >= *n_pick) return;
    const uint32_t j = pick[f];
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        out_ids[(size_t)j * k + i] = ids[(size_t)f * k + i];
        out_dist[(size_t)j * k + i] = dist[(size_t)f * k + i];
    }
    if (threadIdx.x == 0) out_n[j] = n[f];
}

constexpr uint32_t kTwoStageFallbackSets = 64;  // list sets of the device form's fp32 fallback (two_stage_search)

uint32_t blocks_for_rows(uint64_t n) { return (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(1, (n + 255) / 256)); }

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
        ORAMA_TRY(launch_copy_bytes(reinterpret_cast<char*>(rows_base) + (size_t)first * v->row_bytes(), d_src,
                                    (size_t)n * v->row_bytes(), s));
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
    // (a tail that has already finished needs no wait packet in front of the scan: a caller with one step in flight never has one)
    if (sc->tail_recorded && hipEventQuery(sc->ev_tail) != hipSuccess) ORAMA_HIP_TRY(hipStreamWaitEvent(s_scan, sc->ev_tail, 0));
    (void)hipGetLastError();  // (hipErrorNotReady of the query is not an error of this call)
    return ORAMA_OK;
}
// `attached`: an event that completes with the LAST scan launch (it rode on the dispatch: launch_vec_scan_f32) — then no record
// packet goes behind the scan (round 5: every packet between two scans of a pipelined session costs microseconds of an idle chip)
int scan_end(Scratch* sc, hipStream_t s_scan, hipStream_t s, hipEvent_t attached = nullptr) {
    if (s_scan == s) return ORAMA_OK;
    if (attached) {
        ORAMA_HIP_TRY(hipStreamWaitEvent(s, attached, 0));
        return ORAMA_OK;
    }
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

int search_enqueue_f16(orama_vec* v, const View& w, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s, uint32_t* d_out_rows = nullptr, uint32_t* d_inexact = nullptr,
                       int f32_rows = 0);

// ---------------------------------------------------------------- f32 batches: K1m proposes, K1 decides
// A batch of >= ctx->f32_mfma_min_q queries over a plain fp32 store shares corpus passes of <= 32 queries on the matrix cores
// (K1m, vec_f32_mfma.hip) — and still answers with K1's bits, whatever the batch:
//   1. K1m + K2's filter pipeline return the k1 = k + spare best rows of every query by K1m's distance.  K1m and K1 evaluate
//      the same f32 dot product in different orders: |d_K1m - d_K1| <= gamma_n sum|x_i q_i| / (|x||q|) <= 768 x 2^-24 = 4.6e-5
//      for the sequential chain (K1's tree is tighter), plus a few ulp for 1/|q|: eps = 6e-5 is a bound, the typical
//      difference is 1e-7.  Every row of K1's top-k therefore has a K1m distance <= tau + 2 eps, tau = the k-th best K1m
//      distance: the candidate list is COMPLETE when it is not full or its last entry lies beyond tau + 2 eps
//      (shadow_band_kernel — the proof of the fp16 two-stage plan with a band 40 x narrower);
//   2. K1's own arithmetic on the candidates (rerank_f32_kernel: bit-identical distances), K4 with the one-stage tie rule;
//   3. a query whose list is not proven (duplicates / ties around the k-th place) is answered again ON THE DEVICE by K1's own
//      kernel over the flagged queries only (the two-stage plan's device form: pick -> picked scan -> key lists -> scatter);
//      with nothing flagged those launches end at once.
// Everything is enqueued on `s`; no host round trip.  Envelope: cosine, dim % 32 == 0 and <= 864 (the query tile lives in LDS),
// k <= 128 (the fallback's per-wave lists).
// Round 6, second half — K1x (vec_f32_cvt.hip): when every row of the store has a sound fp16 image (orama_vec::f16_safe) the
// PROPOSAL is the fp16 two-stage plan's, computed from the fp32 rows themselves: rows rounded to fp16 in registers,
// v_mfma_f32_32x32x16_f16, <= 64 queries per pass, HBM-bound instead of bound by the f32 matrix pipe; eps = kShadowEps (2.5e-3,
// the plan's proven bound), k1 = max(2k, k + 256) candidates, a query without a sound fp16 image flags itself
// (query_shadow_unsafe_kernel) and takes the fallback.  The answer is K1's either way.
constexpr float kF32MfmaEps = 6.0e-5f;
constexpr float kF32CvtEps = 2.5e-3f;  // = kShadowEps (defined with the two-stage plan below)
constexpr uint32_t kF32MfmaFallbackSets = 32;  // list sets of the device-side K1 fallback (4 MB each at the north-star shape; more flagged queries take further rounds)
// 0 = K1 / K1b, 1 = K1m proposes, 2 = K1x proposes
int f32_batch_plan(const orama_vec* v, uint32_t q, uint32_t k) {
    if (!(v->ctx->f32_mfma_min_q && q >= 2 && k >= 1 && k <= kWaveListKeys && !v->f16() && vec_rerank_f32_supported(v->dim))) return 0;
    const bool cvt = v->ctx->f32_batch_cvt && v->f16_safe.load(std::memory_order_acquire) && vec_scan_f32_cvt_supports(v->dim, v->metric) &&
                     2ull * k <= kSelectMaxK;
    // K1x serves 2..8 queries too where the pass, not the launches, is the cost: one 4.5 ms pass + ~0.25 ms of candidate stage
    // against K1b's 5.6 ms + dense selection at 10 M x 768; under 4 GB of rows K1b's single launch wins (the two-stage plan's rule)
    const uint64_t bytes = v->n_rows.load(std::memory_order_relaxed) * (uint64_t)v->row_bytes();
    const uint32_t min_q = cvt && bytes >= (4ull << 30) ? 2u : v->ctx->f32_mfma_min_q;
    if (q < min_q) return 0;
    if (cvt) return 2;
    return q >= v->ctx->f32_mfma_min_q && vec_scan_f32_mfma_supports(v->dim, v->metric) ? 1 : 0;
}
bool f32_batch_on_mfma(const orama_vec* v, uint32_t q, uint32_t k) { return f32_batch_plan(v, q, k) != 0; }

int search_enqueue_f32_batch(orama_vec* v, const View& w, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                             const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                             uint32_t* d_out_n, hipStream_t s) {
    const int plan = f32_batch_plan(v, q, k);
    ORAMA_REQUIRE(plan != 0, "internal: fp32 batch outside its envelope");
    // spare candidates decide how often the proof fails.  K1m: on the north-star rows the (k + 32)-th distance lies ~2e-3 behind
    // the k-th, 15 bands of 1.2e-4.  K1x: the two-stage plan's max(2k, k + 256) against its band of 5e-3 (0.1 % of the
    // north-star queries fall back).  Candidates are cheap (k1 rows of 3 KiB per query against a 30 GB pass).
    const float eps = plan == 2 ? kF32CvtEps : kF32MfmaEps;
    const uint32_t k1 = plan == 2 ? (uint32_t)std::min<uint64_t>(kSelectMaxK, std::max<uint64_t>(2ull * k, (uint64_t)k + std::max<uint32_t>(1u, v->ctx->two_stage_spare)))
                                  : std::min<uint32_t>(kSelectMaxK, k + std::max<uint32_t>(32u, k / 2));
    const size_t n1 = (size_t)q * k1;
    // every buffer first: nothing is (re)allocated behind launches that are already enqueued (the candidate stage reserves its
    // own — sel_state / sel_keys among them, larger than the final selection needs — before ITS launches)
    ORAMA_TRY(sc->mfma_cand.reserve(n1 * (8 + 4 + 4 + 4) + (size_t)q * 4 * 2 + 64));
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState) * (size_t)q));  // the final selection runs over all q lists at once
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)q * kSelectMaxK));
    uint64_t* c_ids = sc->mfma_cand.as<uint64_t>();
    float* c_dm = reinterpret_cast<float*>(c_ids + n1);
    uint32_t* c_rows = reinterpret_cast<uint32_t*>(c_dm + n1);
    float* c_exact = reinterpret_cast<float*>(c_rows + n1);
    uint32_t* d_n1 = reinterpret_cast<uint32_t*>(c_exact + n1);
    uint32_t* d_flag = d_n1 + q;
    ScanArgs fa;  // the fallback's scan (K1, fused mode)
    fa.corpus = static_cast<const float*>(w.rows);
    fa.inv_norm = w.inv_norm;
    fa.query = d_queries;
    fa.n = w.n_rows;
    fa.dim = v->dim;
    fa.metric = v->metric;
    fa.row_doc = w.row_doc;
    fa.dead = w.dead;
    fa.allow = d_allow;
    fa.allow_bits = allow_bits;
    fa.topk = k;
    ORAMA_REQUIRE(vec_scan_f32_picked_supported(fa), "internal: fp32 batch outside the picked scan's envelope");
    const uint32_t fb_keys = vec_scan_f32_picked_waves(v->ctx, fa) * kWaveListKeys;
    const uint32_t fb_sets = std::min<uint32_t>(q, kF32MfmaFallbackSets);
    const uint32_t fb_rounds = (q + fb_sets - 1) / fb_sets;
    ORAMA_TRY(sc->mfma_fb_lists.reserve((size_t)fb_sets * fb_keys * 8));
    ORAMA_TRY(sc->mfma_fb_tmp.reserve((size_t)keys_topk_scratch_keys(fb_keys, fb_sets, k) * 8 + 8));
    const size_t nk = (size_t)fb_sets * k;
    ORAMA_TRY(sc->mfma_fb_out.reserve(nk * 12 + (size_t)fb_sets * 4 + (size_t)q * 4 + 4 + (size_t)fb_rounds * 4 + 64));
    uint64_t* fb_ids = sc->mfma_fb_out.as<uint64_t>();
    float* fb_dist = reinterpret_cast<float*>(fb_ids + nk);
    uint32_t* fb_n = reinterpret_cast<uint32_t*>(fb_dist + nk);
    uint32_t* d_pick = fb_n + fb_sets;
    uint32_t* d_n_pick = d_pick + q;
    uint32_t* d_round_n = d_n_pick + 1;
    fa.wave_lists = sc->mfma_fb_lists.as<unsigned long long>();

    // 1. candidates by K1m's distance (ids are not needed yet: the final selection maps rows to DocumentIds)
    ORAMA_TRY(search_enqueue_f16(v, w, sc, d_queries, q, k1, d_allow, allow_bits, c_ids, c_dm, d_n1, s, c_rows, nullptr, /*f32_rows=*/plan));
    ORAMA_TRY(launch_shadow_band(c_dm, d_n1, q, k, k1, 2.0f * eps, d_flag, s));
    if (plan == 2) {  // a query the fp16 image cannot serve (tiny norm, an element beyond the fp16 range, NaN) flags itself
        hipLaunchKernelGGL(query_shadow_unsafe_kernel, dim3((q + 3) / 4), dim3(256), 0, s, d_queries, q, v->dim, d_flag);
        ORAMA_HIP_TRY(hipGetLastError());
    }
    // 2. K1's distances of the candidates, then the final order
    ORAMA_TRY(launch_rerank_f32(static_cast<const float*>(w.rows), w.inv_norm, v->dim, d_queries, q, c_rows, d_n1, k1, c_exact, s));
    SelectPlan p;
    p.vals = c_exact;
    p.idx = c_rows;
    p.stride = k1;
    p.n_dev = d_n1;
    p.n = k1;
    p.q = q;
    p.k = k;
    p.descending = false;
    p.id_map = w.row_doc;
    p.state = sc->sel_state.as<SelectState>();
    p.keys = sc->sel_keys.as<unsigned long long>();
    p.keys_capacity = (uint64_t)q * kSelectMaxK;
    p.out_ids = d_out_ids;
    p.out_val = d_out_dist;
    p.out_n = d_out_n;
    ORAMA_TRY(launch_select(v->ctx, p, s));
    // 3. unproven queries: K1 itself, on the device
    hipLaunchKernelGGL(pick_flagged_kernel, dim3(1), dim3(256), 0, s, d_flag, q, d_pick, d_n_pick, fb_sets, fb_rounds, d_round_n);
    ORAMA_HIP_TRY(hipGetLastError());
    for (uint32_t r = 0; r < fb_rounds; ++r) {
        const uint32_t* picks = d_pick + (size_t)r * fb_sets;
        ORAMA_TRY(launch_vec_scan_f32_picked(v->ctx, fa, picks, d_round_n + r, fb_keys, s));
        ORAMA_TRY(launch_keys_topk(v->ctx, fa.wave_lists, fb_keys, fb_keys, fb_sets, k, false, w.row_doc, sc->mfma_fb_tmp.as<unsigned long long>(),
                                   nullptr, fb_ids, fb_dist, fb_n, s, nullptr, nullptr, 0, d_round_n + r));
        hipLaunchKernelGGL(scatter_picked_kernel, dim3(fb_sets), dim3(128), 0, s, picks, d_round_n + r, k, fb_ids, fb_dist, fb_n, d_out_ids,
                           d_out_dist, d_out_n);
        ORAMA_HIP_TRY(hipGetLastError());
    }
    v->mfma_batch_queries.fetch_add(q, std::memory_order_relaxed);
    return ORAMA_OK;
}

int search_enqueue_f32(orama_vec* v, const View& w, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s, hipStream_t s_scan) {
    const uint64_t n = w.n_rows;
    if (f32_batch_on_mfma(v, q, k)) {
        // the batch shares corpus passes of <= 32 queries on the matrix cores, K2's pipeline behind them (dense head ->
        // thresholds -> filter scan -> candidate lists): scans and selections depend on each other both ways, one stream
        return search_enqueue_f32_batch(v, w, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, s);
    }
    // fused_topk: 1 = always, 0 = never, 2 (default) = for ONE query over a large corpus: the dense path writes and
    // re-reads 4 B per row for K4 (40 MB at 10 M rows) and K1b does not apply to a single query — measured at NS
    // 4.51 -> 4.32 ms per scan (85 -> 89 % of the HBM roofline); on a 1 M x 384 corpus the per-wave lists and their
    // reduction cost more than the 4 MB they save (0.24 -> 0.28 ms); 1.25 M x 768 (3.8 GB, the 8-GPU shard) already gains 1.5 %, 2.5 M
    // x 768 4.6 %: the rule asks for >= 3 GB of rows
    const bool fuse = v->ctx->fused_topk == 1 ||
                      (v->ctx->fused_topk == 2 && q == 1 && n * (uint64_t)v->row_bytes() >= (3ull << 30));
    if (k <= kWaveListKeys && fuse) {
        // fused path: each K1 wave keeps its own best-k in registers; no dense distance array at all.
        ScanArgs a;
        a.corpus = static_cast<const float*>(w.rows);
        a.inv_norm = w.inv_norm;
        a.n = n;
        a.dim = v->dim;
        a.metric = v->metric;
        a.row_doc = w.row_doc;
        a.dead = w.dead;
        a.allow = d_allow;
        a.allow_bits = allow_bits;
        a.topk = k;
        const uint32_t waves = vec_scan_f32_waves(v->ctx, a);
        const uint32_t n_keys = waves * kWaveListKeys;
        const uint32_t chunks = (n_keys + kKeysChunk - 1) / kKeysChunk;
        ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)q * n_keys));
        ORAMA_TRY(sc->dist.reserve(sizeof(unsigned long long) * (size_t)q * (size_t)(chunks + 1) * k * 2 + 64));
        ORAMA_TRY(scan_begin(sc, s_scan, s));
        hipEvent_t rode = nullptr;
        for (uint32_t j = 0; j < q; ++j) {
            a.query = d_queries + (size_t)j * v->dim;
            a.wave_lists = sc->sel_keys.as<unsigned long long>() + (size_t)j * n_keys;
            const bool last = j + 1 == q && s_scan != s;
            ORAMA_TRY(launch_vec_scan_f32(v->ctx, a, s_scan, last ? sc->ev_scan : nullptr, last ? &rode : nullptr));
        }
        ORAMA_TRY(scan_end(sc, s_scan, s, rode));
        ORAMA_TRY(launch_keys_topk(v->ctx, sc->sel_keys.as<unsigned long long>(), n_keys, n_keys, q, k, false,
                                w.row_doc, sc->dist.as<unsigned long long>(), nullptr, d_out_ids,
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
    // (kSelectMaxK keys per list: what the two-launch (value, index) selection wants — a lone query over 1 M rows takes it)
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)group * std::max<uint32_t>(k, kSelectMaxK)));
    for (uint32_t q0 = 0; q0 < q; q0 += group) {
        const uint32_t gq = (q - q0) < group ? (q - q0) : group;
        ORAMA_TRY(scan_begin(sc, s_scan, s));
        hipEvent_t rode = nullptr;
        for (uint32_t j = 0; j < gq;) {
            ScanArgs a;
            a.corpus = static_cast<const float*>(w.rows);
            a.inv_norm = w.inv_norm;
            a.query = d_queries + (size_t)(q0 + j) * v->dim;
            a.n = n;
            a.dim = v->dim;
            a.metric = v->metric;
            a.row_doc = w.row_doc;
            a.dead = w.dead;
            a.allow = d_allow;
            a.allow_bits = allow_bits;
            a.out_dist = sc->dist.as<float>() + (size_t)j * n;
            // K1b: up to 8 queries share one corpus pass (a micro-batch of concurrent requests)
            const uint32_t nq = std::min<uint32_t>(gq - j, v->ctx->f32_multi ? kScanMultiMaxQ : 1u);
            if (nq >= 2 && vec_scan_f32_multi_supported(a)) {
                ORAMA_TRY(launch_vec_scan_f32_multi(v->ctx, a, nq, n, s_scan));
                j += nq;
            } else {
                const bool last = j + 1 == gq && s_scan != s;
                ORAMA_TRY(launch_vec_scan_f32(v->ctx, a, s_scan, last ? sc->ev_scan : nullptr, last ? &rode : nullptr));
                j += 1;
            }
        }
        ORAMA_TRY(scan_end(sc, s_scan, s, rode));
        SelectPlan p;
        p.vals = sc->dist.as<float>();
        p.stride = n;
        p.n = (uint32_t)n;
        p.q = gq;
        p.k = k;
        p.descending = false;
        p.id_map = w.row_doc;
        p.state = sc->sel_state.as<SelectState>();
        p.keys = sc->sel_keys.as<unsigned long long>();
        p.keys_capacity = (uint64_t)group * std::max<uint32_t>(k, kSelectMaxK);
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
// `f32_rows` != 0: the rows are the plain fp32 store's (row-major f32) and the scan is K1m (1: f32 on the matrix cores, <= 32
// queries per pass) or K1x (2: rows rounded to fp16 in registers, <= 64 queries per pass).
int search_enqueue_f16(orama_vec* v, const View& w, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s, uint32_t* d_out_rows, uint32_t* d_inexact, int f32_rows) {
    const uint64_t n = w.n_rows;
    if (d_inexact) ORAMA_HIP_TRY(hipMemsetAsync(d_inexact, 0, (size_t)q * 4, s));
    if (!f32_rows && d_out_rows && q == 1 && v->ctx->f16_solo == 2 && (d_inexact || k <= kF16WaveListKeys)) {
        // candidate stage of the two-stage plan for ONE query: K1h keeps every wave's best k rows in registers — one
        // launch over the store + the key reduction, instead of dense head, selection, filter scan, selection
        F16ScanArgs fa;
        fa.tiled = w.rows;
        fa.inv_norm = w.inv_norm;
        fa.queries = d_queries;
        fa.q = 1;
        fa.dim = v->dim;
        fa.metric = v->metric;
        fa.n_rows = n;
        fa.row_begin = 0;
        fa.row_end = n;
        fa.row_doc = w.row_doc;
        fa.dead = w.dead;
        fa.allow = d_allow;
        fa.allow_bits = allow_bits;
        fa.topk = k;
        const uint32_t waves = vec_scan_f16_fused_waves(v->ctx, fa);
        if (waves) {
            const uint32_t n_keys = waves * kF16WaveListKeys;
            ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)n_keys));
            ORAMA_TRY(sc->dist.reserve(keys_topk_scratch_keys(n_keys, 1, k) * 8 + 64));
            ORAMA_TRY(sc->misc0.reserve((size_t)waves * 8));
            fa.wave_lists = sc->sel_keys.as<unsigned long long>();
            fa.wave_thr = sc->misc0.as<unsigned long long>();
            ORAMA_TRY(launch_vec_scan_f16(v->ctx, fa, s));
            ORAMA_TRY(launch_keys_topk(v->ctx, fa.wave_lists, n_keys, n_keys, 1, k, false, w.row_doc, sc->dist.as<unsigned long long>(),
                                       d_out_rows, d_out_ids, d_out_dist, d_out_n, s));
            // a wave keeps 64 rows: did one of them evict a row that belongs to the k best?  (never with k <= 64)
            if (k > kF16WaveListKeys) ORAMA_TRY(launch_shadow_wave_check(fa.wave_thr, waves, d_out_dist, d_out_n, k, d_inexact, s));
            return ORAMA_OK;
        }
    }
    constexpr uint64_t kS1 = 131072;                 // dense head (rows), multiple of 32
    constexpr uint64_t kCandBudget = 6ull << 30;     // bytes of candidate lists per pass
    const uint32_t kpad_k = f16_kpad(v->dim);
    for (uint32_t q0 = 0; q0 < q;) {
        // <= 64 queries: K2 (whole batch as LDS-resident B fragments); more: K2c (GEMM-tiled, <= 256 per pass)
        const bool wide = !f32_rows && v->ctx->f16_wide && (q - q0) > kF16MaxQ && (kpad_k / 16) % 2 == 0;
        const uint32_t gq = std::min<uint32_t>(f32_rows == 2 ? vec_scan_f32_cvt_max_q(v->dim) : f32_rows ? kF32MfmaMaxQ
                                               : wide     ? kF16WideMaxQ : vec_scan_f16_max_q(v->dim), q - q0);
        if (wide) ORAMA_TRY(sc->f16_bfrag.reserve(f16_wide_query_bytes(v->dim)));
        bool wide_prepared = false;
        auto scan = [&](const F16ScanArgs& args) -> int {
            if (f32_rows == 2) return launch_vec_scan_f32_cvt(v->ctx, args, s);
            if (f32_rows) return launch_vec_scan_f32_mfma(v->ctx, args, s);
            if (!wide) return launch_vec_scan_f16(v->ctx, args, s);
            if (!wide_prepared)
                ORAMA_TRY(launch_f16_prepare_queries(args.queries, args.q, args.dim, args.metric, sc->f16_bfrag.p, s));
            wide_prepared = true;
            // f16_wide: 1 = K2c (MFMA waves also issue the DMA), 2 / 3 = K2d (dedicated loader waves, geometry 1 / 2),
            // 4 = K2q where it applies (queries stationary in registers, one query tile per wave: 129..256 queries, rows of
            // <= 768 dimensions) and K2d for everything else; 5 = K2h first (two query tiles per wave, the K loop split over a
            // wave pair — same energy per pass as K2q and K2d, a little slower: kept as the experiment it is,
            // profiles/r03_power_energy.md)
            // (modes 1 and 5 exist in comparison builds only — ORAMA_COMPARISON_KERNELS=1; orama_ctx_set_f16_wide refuses them otherwise)
            const int fw = v->ctx->f16_wide;
#if ORAMA_COMPARISON_KERNELS
            if (fw == 5 && vec_scan_f16_kh_supports(args.dim, args.q)) return launch_vec_scan_f16_kh(v->ctx, args, sc->f16_bfrag.p, s);
            if (fw == 1) return launch_vec_scan_f16_wide(v->ctx, args, sc->f16_bfrag.p, false, s);
#endif
            if (fw >= 4 && vec_scan_f16_qs_supports(args.dim, args.q)) return launch_vec_scan_f16_qs(v->ctx, args, sc->f16_bfrag.p, s);
            return launch_vec_scan_f16_pc(v->ctx, args, sc->f16_bfrag.p, s, fw == 3 ? 2 : 1);
        };
        // (option "f16_head_rows": the dense head — sweeps; a multiple of 256)
        const uint64_t hr = v->ctx->f16_head_rows;
        const uint64_t head_rows = hr >= 4096 && hr % 256 == 0 ? hr : kS1;
        const uint64_t s1 = std::min<uint64_t>(n, head_rows);
        // super-chunk size: gq * (rows + k) * 8 B <= budget (option "f16_cand_mib")
        const uint64_t budget = v->ctx->f16_cand_mib ? v->ctx->f16_cand_mib << 20 : kCandBudget;
        uint64_t chunk_rows = budget / ((uint64_t)gq * 8);
        chunk_rows = std::max<uint64_t>(chunk_rows & ~255ull, 1u << 20);  // whole K2c block tiles (256 rows)
        const uint64_t rest = n > s1 ? n - s1 : 0;
        const uint64_t cand_stride = std::min<uint64_t>(rest, chunk_rows) + k;
        ORAMA_TRY(sc->dist.reserve((size_t)gq * (size_t)std::max<uint64_t>(s1, 1) * 4));
        ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState) * (size_t)gq));
        ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)gq * kSelectMaxK));  // two-launch selections
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
        a.tiled = w.rows;
        a.inv_norm = w.inv_norm;
        a.queries = d_queries + (size_t)q0 * v->dim;
        a.q = gq;
        a.dim = v->dim;
        a.metric = v->metric;
        a.n_rows = n;
        a.row_doc = w.row_doc;
        a.dead = w.dead;
        a.allow = d_allow;
        a.allow_bits = allow_bits;
        a.solo = !f32_rows && d_out_rows != nullptr && v->ctx->f16_solo;  // the shadow stage of the two-stage plan
        // ORAMA_F16_TAU_ORACLE=1, the CEILING of any threshold exchange between shards (scripts/
        // f16_tau_ceiling_probe.py): the filter passes run under the final k-th distances the PREVIOUS call of this scratch set
        // left behind (one ulp up) — exact only when that call asked the same queries, which the probe does
        static const bool tau_oracle = [] { const char* e = orama::dev_env("ORAMA_F16_TAU_ORACLE"); return e && std::atoi(e) != 0; }();
        const float* tau_cap = nullptr;
        if (tau_oracle && q0 == 0 && gq == q) {
            const bool have = sc->f16_tau_cap_q == gq && sc->f16_tau_cap_k == k;
            ORAMA_TRY(sc->f16_tau_cap.reserve((size_t)gq * 4));
            if (have) tau_cap = sc->f16_tau_cap.as<float>();
        }
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
        p.keys_capacity = (uint64_t)gq * kSelectMaxK;
        const bool only_head = rest == 0;
        if (only_head) {
            p.id_map = w.row_doc;
            p.out_ids = out_ids;
            p.out_idx = d_out_rows ? d_out_rows + (size_t)q0 * k : nullptr;
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
        // Super-chunks may GROW geometrically (ORAMA_F16_CHUNK_GROW=1; the next one (ORAMA_F16_GROW_FACTOR - 1) times
        // everything scanned before it): the threshold tau_j only tightens between super-chunks, so growth keeps the share
        // of passing rows near k / rows-so-far instead of the head's k / 131072, for the price of ~log(N / S1) more scan +
        // select launches.  That paid while a passing row drained its wave's prefetch ring (64 queries with 228
        // candidates each: 5.4 -> 3.2 ms of scan).  Since K2 and K2d stage passing rows in LDS and append them in bulk it
        // costs more than it saves at every batch size measured (two-stage: 100 queries 4.87 -> 3.88 ms without growth,
        // 128: 4.41 -> 3.94, 200: 6.10 -> 5.67, 64: 3.35 -> 2.97) and was OFF by default through round 3.
        const int grow_env = v->ctx->f16_chunk_grow;  // options "f16_chunk_grow" / "f16_grow_factor"
        const uint64_t grow_factor_env = (uint64_t)std::max(2, v->ctx->f16_grow_factor);
        // Round 4: ON for the wide passes (more than 64 queries: K2q / K2d) with factor 8 — there the candidate budget cuts the rest
        // into ~3 M-row super-chunks anyway, and ONE extra small chunk behind the head (7 x 131 072 rows) hands the first big
        // one a threshold from 1 M rows instead of 131 072: 2 400 -> ~1 000 candidates per query, C5 shard 4.80 -> 4.69 ms per
        // step now that a selection costs 45 instead of 75 us.  64 queries (one super-chunk covers the whole rest) lose 5 %
        // with it (2.51 -> 2.64 ms): off there.
        const bool grow = grow_env >= 0 ? grow_env > 0 : wide;
        const uint64_t grow_factor = grow_env > 0 ? grow_factor_env : 8ull;
        uint64_t this_chunk =
            grow ? std::min<uint64_t>(chunk_rows, (grow_factor - 1) * std::max<uint64_t>(s1 & ~255ull, 1u << 12)) : chunk_rows;
        for (uint64_t r0 = s1; r0 < n;) {
            const uint64_t r1 = std::min<uint64_t>(n, r0 + this_chunk);
            ORAMA_TRY(launch_f16_seed_candidates(best_dist, best_row, out_n, gq, k, tau, cand_dist, cand_row,
                                                 cand_count, cand_stride, s, tau_cap));
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
            // lists hold k + ~k x (rows of the super-chunk / rows before it) entries on unordered data — a few hundred to a
            // thousand: ONE workgroup reduces and finishes such a list; longer ones (sorted data) are split over the two the
            // hint asks for and take more rounds
            c.n_hint = 2u * kKeysChunk;
            c.q = gq;
            c.k = k;
            c.descending = false;
            c.state = sc->sel_state.as<SelectState>();
            c.keys = sc->sel_keys.as<unsigned long long>();
            c.keys_capacity = (uint64_t)gq * kSelectMaxK;
            c.out_n = out_n;
            if (r1 == n) {  // 3. final reduction: ids + final tie order
                c.id_map = w.row_doc;
                c.out_ids = out_ids;
                c.out_idx = d_out_rows ? d_out_rows + (size_t)q0 * k : nullptr;
                c.out_val = out_dist;
            } else {
                c.out_idx = best_row;
                c.out_val = best_dist;
            }
            ORAMA_TRY(launch_select(v->ctx, c, s));
            r0 = r1;
            if (grow) this_chunk = std::min<uint64_t>(chunk_rows, (grow_factor - 1) * (r1 & ~255ull));
        }
        if (tau_oracle && q0 == 0 && gq == q) {
            ORAMA_TRY(launch_f16_remember_kth(out_dist, out_n, k, gq, sc->f16_tau_cap.as<float>(), s));
            sc->f16_tau_cap_q = gq;
            sc->f16_tau_cap_k = k;
        }
        q0 += gq;
    }
    return ORAMA_OK;
}

// `s_scan` (nullable = same as s): stream the corpus scans run on.  The f16 pipeline interleaves scans and
// selections with dependencies in both directions, so it stays on `s`.
// The caller holds the store's shared lock; the search works on the snapshot taken here: rows published later are
// not seen, and nothing it reads can move (arrays grow in place; compaction needs the exclusive lock).
int search_enqueue(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                   const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                   uint32_t* d_out_n, hipStream_t s, hipStream_t s_scan = nullptr, bool two_streams = false) {
    const View w = snapshot(v);
    if (w.n_rows == 0) {  // empty store: no hits
        ORAMA_HIP_TRY(hipMemsetAsync(d_out_n, 0, (size_t)q * 4, s));
        return ORAMA_OK;
    }
    return v->f16() ? search_enqueue_f16(v, w, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist,
                                         d_out_n, s)
                    : search_enqueue_f32(v, w, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist,
                                         d_out_n, s, two_streams ? s_scan : s);
}

// ---------------------------------------------------------------- two-stage exact search (fp32 rows + fp16 shadow)
// Stage 1: the fp16 shadow (same rows, same order, half the bytes, MFMA for batches) returns the k1 = max(2k, k + 256)
// best rows by its approximate distance.  |shadow - exact| <= eps for every row (both operands rounded to fp16:
// elementwise relative error 2^-11, and the shadow's norms come from the rounded rows: <= 2^-9 on the cosine, plus the
// f32 accumulation), so every row of the exact top-k has a shadow distance <= tau + 2 eps, tau = the k-th best shadow
// distance: the candidate list is COMPLETE when it is not full or its last entry lies beyond tau + 2 eps
// (shadow_band_kernel); otherwise the query is flagged and answered by the plain fp32 scan.
// Stage 2: K1's own arithmetic on the candidate rows (rerank_f32_kernel: bit-identical distances), then K4 with the
// same tie rule as the one-stage path.  The answer equals the fp32 scan's bit for bit.
constexpr float kShadowEps = 2.5e-3f;

// `flags_pending` == nullptr selects the DEVICE form (orama_vec_search_*device*, sessions — no host between the stages and the
// consumer of the answers): the queries that are not proven are re-answered on the device, behind the selection, by K1's own
// kernel run over a list of queries that is made on the device (pick_flagged_kernel -> launch_vec_scan_f32_picked -> the key
// lists' top-k restricted to the picked lists -> scatter_picked_kernel).  With nothing flagged — 99.9 % of the queries — these
// launches end at once (~40 us per call); every flagged query costs one fp32 pass, as on the host form.  A query the shadow
// cannot serve at all (query_shadow_unsafe_kernel: the test vec_two_stage_usable applies on the host) is flagged like an
// unproven one.
int two_stage_search(orama_vec* v, Scratch* sc, Scratch* sc2, const float* d_queries, uint32_t q, uint32_t k,
                     const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                     hipStream_t s, bool* flags_pending) {
    // (the caller holds the shadow's mu shared — and v->mu shared, which also excludes a compaction of the pair — until
    // the launches below have completed; on the device form: until they are enqueued, like every *_device call)
    orama_vec* sh = v->shadow.get();
    const bool device_form = flags_pending == nullptr;
    if (flags_pending) *flags_pending = false;
    const View w = snapshot(v);
    View ws = snapshot(sh);
    ws.n_rows = std::min(ws.n_rows, w.n_rows);  // the shadow is written first: it may already hold unpublished rows
    if (ws.n_rows == 0) {
        ORAMA_HIP_TRY(hipMemsetAsync(d_out_n, 0, (size_t)q * 4, s));
        return ORAMA_OK;
    }
    // k1 - k spare candidates decide how often the completeness proof fails: on the north-star corpus the 228th best
    // shadow distance lies ~7e-3 behind the 100th, barely more than 2 eps, and 0.1 % of the queries fell back to the
    // fp32 scan (4.3 ms each: 0.8 ms per batch of 256 on average); the 356th lies ~1e-2 behind.  Candidates are cheap.
    const uint64_t spare = std::max<uint32_t>(1u, v->ctx->two_stage_spare);  // option "two_stage_spare"
    const uint32_t k1 = (uint32_t)std::min<uint64_t>(kSelectMaxK, std::max<uint64_t>(2ull * k, (uint64_t)k + spare));
    // stage 1 works in the second scratch set (the fp16 pipeline uses most buffers of one)
    const size_t n1 = (size_t)q * k1;
    ORAMA_TRY(sc2->out_ids.reserve(n1 * 8));
    ORAMA_TRY(sc2->out_val.reserve(n1 * 4));
    ORAMA_TRY(sc2->out_idx.reserve(n1 * 4));
    ORAMA_TRY(sc2->out_n.reserve((size_t)q * 4 * 3));
    uint32_t* d_n1 = sc2->out_n.as<uint32_t>();
    uint32_t* d_flag = d_n1 + q;
    uint32_t* d_inexact = d_n1 + 2 * (size_t)q;  // stage 1 itself could not prove its list (K1h's 64 rows per wave)
    // the device form's buffers first: nothing is (re)allocated behind launches that are already enqueued
    ScanArgs fa;  // the fallback's scan (K1, fused mode)
    uint32_t fb_keys = 0;
    uint32_t fb_sets = 0, fb_rounds = 0;
    uint32_t *d_pick = nullptr, *d_n_pick = nullptr, *d_round_n = nullptr, *fb_n = nullptr;
    uint64_t* fb_ids = nullptr;
    float* fb_dist = nullptr;
    if (device_form) {
        fa.corpus = static_cast<const float*>(w.rows);
        fa.inv_norm = w.inv_norm;
        fa.query = d_queries;
        fa.n = w.n_rows;
        fa.dim = v->dim;
        fa.metric = v->metric;
        fa.row_doc = w.row_doc;
        fa.dead = w.dead;
        fa.allow = d_allow;
        fa.allow_bits = allow_bits;
        fa.topk = k;
        ORAMA_REQUIRE(k <= kWaveListKeys && vec_scan_f32_picked_supported(fa), "internal: two-stage device form outside its envelope");
        fb_keys = vec_scan_f32_picked_waves(v->ctx, fa) * kWaveListKeys;
        // list sets for kTwoStageFallbackSets queries (4 MB each at the north-star shape), not for q: the path answers ~0.1 % of
        // the queries, and a batch of 256 pinned 1 GB per tail stream and shard for it.  More flagged queries than
        // sets take further rounds of the same three launches
        fb_sets = std::min<uint32_t>(q, kTwoStageFallbackSets);
        fb_rounds = (q + fb_sets - 1) / fb_sets;
        ORAMA_TRY(sc->misc3.reserve((size_t)fb_sets * fb_keys * 8));
        ORAMA_TRY(sc->misc4.reserve((size_t)keys_topk_scratch_keys(fb_keys, fb_sets, k) * 8 + 8));
        const size_t nk = (size_t)fb_sets * k;
        ORAMA_TRY(sc->misc5.reserve(nk * 12 + (size_t)fb_sets * 4 + (size_t)q * 4 + 4 + (size_t)fb_rounds * 4 + 64));
        fb_ids = sc->misc5.as<uint64_t>();
        fb_dist = reinterpret_cast<float*>(fb_ids + nk);
        fb_n = reinterpret_cast<uint32_t*>(fb_dist + nk);
        d_pick = fb_n + fb_sets;
        d_n_pick = d_pick + q;
        d_round_n = d_n_pick + 1;
        fa.wave_lists = sc->misc3.as<unsigned long long>();
    }
    ORAMA_TRY(search_enqueue_f16(sh, ws, sc2, d_queries, q, k1, d_allow, allow_bits, sc2->out_ids.as<uint64_t>(),
                                 sc2->out_val.as<float>(), d_n1, s, sc2->out_idx.as<uint32_t>(), d_inexact));
    ORAMA_TRY(launch_shadow_band(sc2->out_val.as<float>(), d_n1, q, k, k1, 2.0f * kShadowEps, d_flag, s, d_inexact));
    // stage 2: exact distances of the candidates, then the final order
    ORAMA_TRY(sc->dist.reserve(n1 * 4));
    ORAMA_TRY(launch_rerank_f32(static_cast<const float*>(w.rows), w.inv_norm, v->dim, d_queries, q, sc2->out_idx.as<uint32_t>(),
                                d_n1, k1, sc->dist.as<float>(), s));
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState) * (size_t)q));
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * (size_t)q * k));
    SelectPlan p;
    p.vals = sc->dist.as<float>();
    p.idx = sc2->out_idx.as<uint32_t>();
    p.stride = k1;
    p.n_dev = d_n1;
    p.n = k1;
    p.q = q;
    p.k = k;
    p.descending = false;
    p.id_map = w.row_doc;
    p.state = sc->sel_state.as<SelectState>();
    p.keys = sc->sel_keys.as<unsigned long long>();
    p.out_ids = d_out_ids;
    p.out_val = d_out_dist;
    p.out_n = d_out_n;
    ORAMA_TRY(launch_select(v->ctx, p, s));
    if (device_form) {
        hipLaunchKernelGGL(query_shadow_unsafe_kernel, dim3((q + 3) / 4), dim3(256), 0, s, d_queries, q, v->dim, d_flag);
        hipLaunchKernelGGL(pick_flagged_kernel, dim3(1), dim3(256), 0, s, d_flag, q, d_pick, d_n_pick, fb_sets, fb_rounds, d_round_n);
        ORAMA_HIP_TRY(hipGetLastError());
        for (uint32_t r = 0; r < fb_rounds; ++r) {
            const uint32_t* picks = d_pick + (size_t)r * fb_sets;
            ORAMA_TRY(launch_vec_scan_f32_picked(v->ctx, fa, picks, d_round_n + r, fb_keys, s));
            ORAMA_TRY(launch_keys_topk(v->ctx, fa.wave_lists, fb_keys, fb_keys, fb_sets, k, false, w.row_doc, sc->misc4.as<unsigned long long>(),
                                       nullptr, fb_ids, fb_dist, fb_n, s, nullptr, nullptr, 0, d_round_n + r));
            hipLaunchKernelGGL(scatter_picked_kernel, dim3(fb_sets), dim3(128), 0, s, picks, d_round_n + r, k, fb_ids, fb_dist, fb_n, d_out_ids,
                               d_out_dist, d_out_n);
            ORAMA_HIP_TRY(hipGetLastError());
        }
        v->two_stage_queries.fetch_add(q, std::memory_order_relaxed);
        return ORAMA_OK;
    }
    ORAMA_TRY(sc2->h_out.reserve((size_t)q * 4));
    ORAMA_TRY(stage_block(v->ctx, sc2->h_out.p, d_flag, (size_t)q * 4, hipMemcpyDeviceToHost, s));
    *flags_pending = true;  // sc2->h_out holds one word per query once `s` has drained: non-zero = not proven
    return ORAMA_OK;
}

// The device entry points take the two-stage plan when the store keeps a shadow and the call lies inside the plan's envelope
// (the host form's conditions without the look at the queries — they are on the device: query_shadow_unsafe_kernel).
bool two_stage_device_usable(orama_vec* v, uint32_t q, uint32_t k) {
    if (!(v->shadow && v->ctx->two_stage && v->shadow_ok.load(std::memory_order_acquire) && k >= 1 && k <= kWaveListKeys &&
          v->metric == ORAMA_METRIC_COSINE && (v->dim & 3) == 0 && v->dim <= 1024))
        return false;
    const uint64_t rows = v->n_rows.load(std::memory_order_relaxed);
    if (v->ctx->two_stage != 2 && q <= 8 && rows * v->row_bytes() < (4ull << 30)) return false;  // (as vec_two_stage_usable)
    // (wave lists of the fallback: one set per query up to kTwoStageFallbackSets, 4 MB each at the default grid — two_stage_search)
    return q <= 4096;
}

}  // namespace

namespace orama {
VecSharedLock::VecSharedLock(orama_vec* v) : v_(v) { v_->mu.lock_shared(); }
VecSharedLock::~VecSharedLock() { v_->mu.unlock_shared(); }
orama_ctx* vec_ctx(orama_vec* v) { return v->ctx; }
uint32_t vec_dim(orama_vec* v) { return v->dim; }
uint64_t vec_rows(orama_vec* v) { return v->n_rows.load(std::memory_order_acquire); }
bool vec_rows_are_f32(orama_vec* v) { return !v->f16(); }
int vec_search_enqueue(orama_vec* v, Scratch* sc, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist,
                       uint32_t* d_out_n, hipStream_t s) {
    return search_enqueue(v, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, s);
}
bool vec_two_stage_usable(orama_vec* v, const float* queries, uint32_t q, uint32_t k) {
    if (!(v->shadow && v->ctx->two_stage && v->shadow_ok.load(std::memory_order_acquire) && k >= 1 &&
          2 * (uint64_t)k <= kSelectMaxK))
        return false;
    // a small store is scanned in fp32 faster than the two stages' extra launches take (1 M x 384: 0.25 ms vs 0.29 ms);
    // batches beyond K1b's 8 queries per pass always gain
    if (v->ctx->two_stage != 2 && q <= 8 && (uint64_t)v->n_rows.load(std::memory_order_relaxed) * v->row_bytes() < (4ull << 30))
        return false;
    for (uint32_t j = 0; j < q; ++j) {
        // the same test the rows pass at insert (row_shadow_safe)
        const float* x = queries + (size_t)j * v->dim;
        float n2 = 0.0f, mx = 0.0f;
        for (uint32_t i = 0; i < v->dim; ++i) {
            n2 += x[i] * x[i];
            mx = std::max(mx, std::fabs(x[i]));
        }
        if (!(n2 >= 1e-4f && mx < 6.0e4f)) return false;
    }
    return true;
}
VecTwoStage::~VecTwoStage() {
    if (locked_) {  // begin() without finish() (an error in between): nothing of the call may stay in flight
        (void)hipStreamSynchronize(sc_->stream);
        unlock();
    }
}
void VecTwoStage::unlock() {
    if (locked_) static_cast<std::shared_mutex*>(locked_)->unlock_shared();
    locked_ = nullptr;
}
int VecTwoStage::begin(orama_vec* v, ScratchLease& sc, ScratchLease& sc2, const float* d_queries, uint32_t q, uint32_t k,
                       const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n) {
    ORAMA_REQUIRE(!locked_, "internal: two-stage search begun twice");
    v_ = v, sc_ = sc.s.get(), sc2_ = sc2.s.get(), d_queries_ = d_queries, q_ = q, k_ = k, d_allow_ = d_allow, allow_bits_ = allow_bits;
    d_out_ids_ = d_out_ids, d_out_dist_ = d_out_dist, d_out_n_ = d_out_n;
    v->shadow->mu.lock_shared();
    locked_ = &v->shadow->mu;
    return two_stage_search(v, sc.s.get(), sc2.s.get(), d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, sc->stream, &flags_pending_);
}
bool vec_two_stage_device_usable(orama_vec* v, uint32_t q, uint32_t k) { return two_stage_device_usable(v, q, k); }
int VecTwoStage::begin_device(orama_vec* v, ScratchLease& sc, ScratchLease& sc2, const float* d_queries, uint32_t q, uint32_t k,
                              const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n) {
    ORAMA_REQUIRE(!locked_, "internal: two-stage search begun twice");
    ORAMA_REQUIRE(two_stage_device_usable(v, q, k), "internal: two-stage device form outside its envelope");
    v_ = v, sc_ = sc.s.get(), sc2_ = sc2.s.get(), d_queries_ = d_queries, q_ = q, k_ = k, d_allow_ = d_allow, allow_bits_ = allow_bits;
    d_out_ids_ = d_out_ids, d_out_dist_ = d_out_dist, d_out_n_ = d_out_n;
    v->shadow->mu.lock_shared();
    locked_ = &v->shadow->mu;
    flags_pending_ = false;
    device_form_ = true;
    return two_stage_search(v, sc.s.get(), sc2.s.get(), d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, sc->stream, nullptr);
}
int VecTwoStage::finish(bool* reran) {
    if (reran) *reran = false;
    ORAMA_REQUIRE(locked_, "internal: two-stage search not begun");
    hipStream_t s = sc_->stream;
    const hipError_t e = hipStreamSynchronize(s);
    unlock();
    ORAMA_HIP_TRY(e);
    if (device_form_) return ORAMA_OK;  // (the device form counts its queries itself, two_stage_search)
    v_->two_stage_queries.fetch_add(q_, std::memory_order_relaxed);
    if (!flags_pending_) return ORAMA_OK;
    const uint32_t* hf = sc2_->h_out.as<uint32_t>();
    bool any = false;
    for (uint32_t j = 0; j < q_; ++j) {
        if (!hf[j]) continue;
        any = true;
        v_->two_stage_fallbacks.fetch_add(1, std::memory_order_relaxed);
        ORAMA_TRY(search_enqueue(v_, sc_, d_queries_ + (size_t)j * v_->dim, 1, k_, d_allow_, allow_bits_, d_out_ids_ + (size_t)j * k_,
                                 d_out_dist_ + (size_t)j * k_, d_out_n_ + j, s));
    }
    if (any) ORAMA_HIP_TRY(hipStreamSynchronize(s));
    if (reran) *reran = any;
    return ORAMA_OK;
}
int vec_two_stage_search(orama_vec* v, ScratchLease& sc, ScratchLease& sc2, const float* d_queries, uint32_t q, uint32_t k,
                         const uint64_t* d_allow, uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n) {
    VecTwoStage call;
    ORAMA_TRY(call.begin(v, sc, sc2, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n));
    return call.finish();
}
}  // namespace orama

extern "C" {

int orama_vec_create(orama_ctx* ctx, uint32_t dim, int metric, int dtype, uint64_t reserve_rows,
                     orama_vec** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(dim >= 1, "dimensions is 0");
    ORAMA_SUPPORT(dim <= 65536, "dimensions %u outside [1, 65536]", dim);
    ORAMA_REQUIRE(metric == ORAMA_METRIC_COSINE || metric == ORAMA_METRIC_L2SQ, "unknown metric %d", metric);
    ORAMA_REQUIRE(dtype == ORAMA_DTYPE_F32 || dtype == ORAMA_DTYPE_F16 || dtype == ORAMA_DTYPE_F32_SHADOW16, "unknown dtype %d",
                  dtype);
    if (dtype == ORAMA_DTYPE_F32_SHADOW16) {
        // fp32 rows decide every answer; an fp16 copy of the same rows proposes the candidates (two-stage exact search)
        ORAMA_SUPPORT(metric == ORAMA_METRIC_COSINE && vec_rerank_f32_supported(dim) && dim <= 2048,
                      "the fp16 shadow supports the cosine metric and dimensions that are a multiple of 4 up to 1024");
        orama_vec* primary = nullptr;
        ORAMA_TRY(orama_vec_create(ctx, dim, metric, ORAMA_DTYPE_F32, reserve_rows, &primary));
        orama_vec* sh = nullptr;
        const int st = orama_vec_create(ctx, dim, metric, ORAMA_DTYPE_F16, reserve_rows, &sh);
        if (st != ORAMA_OK) {
            orama_vec_destroy(primary);
            return st;
        }
        primary->shadow.reset(sh);
        *out = primary;
        return ORAMA_OK;
    }
    if (dtype == ORAMA_DTYPE_F16 && dim > 2048) {
        set_error("f16 storage: dimensions %u > 2048 exceed the LDS query tile", dim);
        return ORAMA_ERR_UNSUPPORTED;
    }
    ORAMA_ON_DEVICE(ctx->device);
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
        if (st == ORAMA_OK && hipStreamSynchronize(sc->stream) != hipSuccess) st = ORAMA_ERR_HIP;
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
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(v->ctx->device);
    (void)hipDeviceSynchronize();
    delete v;
}

// Writers reserve capacity first (write_mu held by the caller).
//   default  : allocate bigger arrays and copy the old contents BESIDE the running searches (shared lock: a search that
//              snapshots a mix of old and new base pointers still reads identical published rows), then take the
//              exclusive lock for an instant — a barrier that outlives every search which may still hold an old base —
//              and free the old arrays.  Readers are never blocked for the duration of a copy.
//   ORAMA_VMM: grow in place (no copy at all); only outgrowing the virtual reservation re-maps under the exclusive lock.
static int reserve_rows_locked(orama_vec* v, uint64_t need_rows, hipStream_t s) {
    if (need_rows <= v->cap_rows) return ORAMA_OK;
    const bool vmm = GrowBuf::vmm_supported(v->ctx->device);
    if (vmm && grow_needs_move(v, need_rows)) {
        std::unique_lock<std::shared_mutex> x(v->mu);  // waits for running scans, blocks new ones for the re-map
        ORAMA_TRY(grow(v, need_rows, s));
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
        return ORAMA_OK;
    }
    {
        std::shared_lock<std::shared_mutex> r(v->mu);  // vs compaction only
        // geometric growth: the copy is amortised over the rows that fit the new capacity
        const uint64_t target = vmm ? need_rows : std::max<uint64_t>(need_rows, v->cap_rows + v->cap_rows / 2 + 1024);
        ORAMA_TRY(grow(v, target, s));
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
    }
    if (!vmm) {
        { std::unique_lock<std::shared_mutex> barrier(v->mu); }
        v->rows.drop_old();
        v->inv_norm.drop_old();
        v->row_doc.drop_old();
        v->dead.drop_old();
    }
    return ORAMA_OK;
}

static int vec_insert_one(orama_vec* v, const uint64_t* doc_ids, const float* rows, uint64_t n_rows, uint64_t* accepted);
static int vec_delete_one(orama_vec* v, const uint64_t* doc_ids, uint64_t n);
static int vec_compact_one(orama_vec* v, uint64_t version, bool mu_held = false);
static int vec_fill_synthetic_one(orama_vec* v, uint64_t n_rows, uint64_t seed, uint64_t first_doc_id);

// fp16 image of a row with a usable error bound: elementwise relative error 2^-11 needs the elements inside the fp16
// normal range — |x_i| < 6e4, and a norm large enough that the elements below 6.1e-5 (absolute error 2^-25 each) do not
// matter
static bool row_shadow_safe(const float* x, uint32_t d) {
    float n2 = 0.0f, mx = 0.0f;
    for (uint32_t i = 0; i < d; ++i) {
        n2 += x[i] * x[i];
        mx = std::max(mx, std::fabs(x[i]));
    }
    return n2 >= 1e-4f && mx < 6.0e4f;
}

int orama_vec_insert(orama_vec* v, const uint64_t* doc_ids, const float* rows, uint64_t n_rows,
                     uint64_t* accepted) {
    ORAMA_REQUIRE(v, "null handle");
    if (!v->shadow) return vec_insert_one(v, doc_ids, rows, n_rows, accepted);
    if (accepted) *accepted = 0;
    if (n_rows == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc_ids && rows, "null input");
    std::lock_guard<std::mutex> cl(v->composite_mu);
    if (v->shadow_ok.load(std::memory_order_relaxed)) {
        for (uint64_t i = 0; i < n_rows; ++i) {
            const float* x = rows + i * (uint64_t)v->dim;
            if (row_valid(x, v->dim) && !row_shadow_safe(x, v->dim)) {
                v->shadow_ok.store(false, std::memory_order_release);
                break;
            }
        }
    }
    // shadow first: a search clamps the shadow's row count to this store's published count
    uint64_t a0 = 0, a1 = 0;
    const int st0 = vec_insert_one(v->shadow.get(), doc_ids, rows, n_rows, &a0);
    const int st1 = st0 == ORAMA_OK ? vec_insert_one(v, doc_ids, rows, n_rows, &a1) : ORAMA_OK;
    // The two-stage plan reranks shadow row i from fp32 row i: the two copies must hold the same rows at the same
    // indices.  A failed insert (OOM, a HIP error) may have published some slabs in one copy only; from then on the
    // indices disagree, so the plan is switched off for good and every search runs the plain fp32 scan.
    if (st0 != ORAMA_OK || st1 != ORAMA_OK || a0 != a1 ||
        v->shadow->n_rows.load(std::memory_order_acquire) != v->n_rows.load(std::memory_order_acquire))
        v->shadow_ok.store(false, std::memory_order_release);
    ORAMA_TRY(st0);
    ORAMA_TRY(st1);
    ORAMA_REQUIRE(a0 == a1, "internal: the fp16 shadow accepted %llu rows, the store %llu (two-stage plan switched off)",
                  (unsigned long long)a0, (unsigned long long)a1);
    if (accepted) *accepted = a1;
    return ORAMA_OK;
}

static int vec_insert_one(orama_vec* v, const uint64_t* doc_ids, const float* rows, uint64_t n_rows,
                          uint64_t* accepted) {
    ORAMA_REQUIRE(v, "null handle");
    if (accepted) *accepted = 0;
    if (n_rows == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc_ids && rows, "null input");
    ORAMA_ON_DEVICE(v->ctx->device);
    std::lock_guard<std::mutex> wl(v->write_mu);  // one writer; searches keep running on the published snapshot
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    // validate + pack accepted rows into pinned staging, in slabs
    const size_t rb = v->row_bytes();
    const uint64_t slab_rows = std::max<uint64_t>(1, (64ull << 20) / rb);
    uint64_t total_ok = 0;
    for (uint64_t r0 = 0; r0 < n_rows; r0 += slab_rows) {
        const uint64_t cnt = std::min(slab_rows, n_rows - r0);
        ORAMA_TRY(sc->h_in.reserve((size_t)cnt * (rb + 8)));
        float* stage = sc->h_in.as<float>();
        uint64_t* stage_doc = reinterpret_cast<uint64_t*>(sc->h_in.as<char>() + (size_t)cnt * rb);
        uint64_t ok = 0;
        for (uint64_t i = 0; i < cnt; ++i) {
            const float* x = rows + (r0 + i) * (uint64_t)v->dim;
            if (!row_valid(x, v->dim)) continue;
            if (!v->f16() && v->f16_safe.load(std::memory_order_relaxed) && !row_shadow_safe(x, v->dim))
                v->f16_safe.store(false, std::memory_order_release);
            memcpy(stage + ok * (uint64_t)v->dim, x, rb);
            stage_doc[ok] = doc_ids[r0 + i];
            ++ok;
        }
        if (!ok) continue;
        const uint64_t first = v->n_rows.load(std::memory_order_relaxed);
        ORAMA_TRY(reserve_rows_locked(v, first + ok, s));
        std::shared_lock<std::shared_mutex> r(v->mu);  // vs compaction
        ORAMA_TRY(sc->misc1.reserve((size_t)ok * rb));
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc1.p, stage, (size_t)ok * rb, hipMemcpyHostToDevice, s));
        // rows [first, first + ok) are beyond the published count: no search reads them yet
        ORAMA_TRY(store_rows_from_device(v, v->rows.base, v->inv_norm.as<float>(), sc->misc1.as<float>(), first, ok, s));
        ORAMA_TRY(sc->misc2.reserve((size_t)ok * sizeof(uint64_t)));
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc2.p, stage_doc, (size_t)ok * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        ORAMA_TRY(launch_copy_bytes(v->row_doc.as<uint64_t>() + first, sc->misc2.p, (size_t)ok * sizeof(uint64_t), s));
        ORAMA_HIP_TRY(hipStreamSynchronize(s));
        v->n_rows.store(first + ok, std::memory_order_release);  // publish: later searches see the slab
        total_ok += ok;
    }
    if (accepted) *accepted = total_ok;
    return ORAMA_OK;
}

int orama_vec_delete(orama_vec* v, const uint64_t* doc_ids, uint64_t n) {
    ORAMA_REQUIRE(v, "null handle");
    if (!v->shadow) return vec_delete_one(v, doc_ids, n);
    std::lock_guard<std::mutex> cl(v->composite_mu);
    ORAMA_TRY(vec_delete_one(v->shadow.get(), doc_ids, n));
    return vec_delete_one(v, doc_ids, n);
}

static int vec_delete_one(orama_vec* v, const uint64_t* doc_ids, uint64_t n) {
    ORAMA_REQUIRE(v, "null handle");
    if (n == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc_ids, "null input");
    ORAMA_REQUIRE(n < 0xffffffffull, "too many ids in one delete");
    ORAMA_ON_DEVICE(v->ctx->device);
    std::lock_guard<std::mutex> wl(v->write_mu);
    std::shared_lock<std::shared_mutex> r(v->mu);  // searches keep running; each sees the tombstones or not
    const uint64_t rows = v->n_rows.load(std::memory_order_acquire);
    if (rows == 0) return ORAMA_OK;
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    // one pass over row_doc on the device (8 B per row: 80 MB at 10 M rows) instead of a host hash map of every row
    std::vector<uint64_t> sorted(doc_ids, doc_ids + n);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    ORAMA_TRY(sc->misc0.reserve(sorted.size() * 8 + 16));
    ORAMA_TRY(sc->h_in.reserve(sorted.size() * 8));
    memcpy(sc->h_in.p, sorted.data(), sorted.size() * 8);
    unsigned long long* d_newly = reinterpret_cast<unsigned long long*>(sc->misc0.as<char>() + sorted.size() * 8);
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, sc->h_in.p, sorted.size() * 8, hipMemcpyHostToDevice, s));
    ORAMA_HIP_TRY(hipMemsetAsync(d_newly, 0, 8, s));
    hipLaunchKernelGGL(mark_deleted_kernel, dim3(blocks_for_rows(rows)), dim3(256), 0, s, v->row_doc.as<uint64_t>(), rows,
                       sc->misc0.as<uint64_t>(), (uint32_t)sorted.size(), v->dead.as<uint32_t>(), d_newly);
    ORAMA_HIP_TRY(hipGetLastError());
    ORAMA_TRY(sc->h_out.reserve(8));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->h_out.p, d_newly, 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    const uint64_t newly = *sc->h_out.as<uint64_t>();
    if (newly) v->n_dead.fetch_add(newly, std::memory_order_release);
    return ORAMA_OK;
}

int orama_vec_compact(orama_vec* v, uint64_t version) {
    ORAMA_REQUIRE(v, "null handle");
    if (!v->shadow) return vec_compact_one(v, version);
    // both copies drop the same rows (the same deletes reached both), so the row indices stay aligned.  Every search of
    // this store — two-stage or not — holds v->mu shared for its whole duration: taking it exclusively HERE, around both
    // re-packs, means no search ever sees one copy compacted and the other not (and the lock order of a search,
    // v->mu then shadow->mu, is the order used here).
    std::lock_guard<std::mutex> cl(v->composite_mu);
    ORAMA_ON_DEVICE(v->ctx->device);
    std::unique_lock<std::shared_mutex> lk(v->mu);
    ORAMA_TRY(vec_compact_one(v->shadow.get(), version));
    return vec_compact_one(v, version, true);
}

static int vec_compact_one(orama_vec* v, uint64_t version, bool mu_held) {
    ORAMA_REQUIRE(v, "null handle");
    ORAMA_ON_DEVICE(v->ctx->device);
    std::lock_guard<std::mutex> wl(v->write_mu);
    std::unique_lock<std::shared_mutex> lk(v->mu, std::defer_lock);  // rows move: no scan may run (the reference's compact is exclusive too)
    if (!mu_held) lk.lock();
    v->version = version;
    const uint64_t n_old = v->n_rows.load(std::memory_order_acquire);
    if (v->n_dead.load(std::memory_order_acquire) == 0) return ORAMA_OK;
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    // live row list from the tombstone bitmap (n/8 bytes over PCIe; compaction is rare)
    std::vector<uint32_t> h_dead((size_t)((n_old + 31) / 32));
    ORAMA_TRY(sc->misc2.reserve(h_dead.size() * 4));
    ORAMA_TRY(launch_copy_bytes(sc->misc2.p, v->dead.base, h_dead.size() * 4, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h_dead.data(), sc->misc2.p, h_dead.size() * 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    std::vector<uint64_t> live;
    live.reserve((size_t)n_old);
    for (uint64_t r = 0; r < n_old; ++r)
        if (!((h_dead[r >> 5] >> (r & 31)) & 1u)) live.push_back(r);
    const uint64_t m = live.size();
    // Re-pack into FRESH arrays (device gather through the row-index list, in slabs), then swap.
    std::unique_ptr<orama_vec> nv(new (std::nothrow) orama_vec());
    if (!nv) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    nv->ctx = v->ctx;
    nv->dim = v->dim;
    nv->metric = v->metric;
    nv->dtype = v->dtype;
    ORAMA_TRY(grow(nv.get(), m ? m : 1, s));
    if (m) {
        const uint64_t slab = std::max<uint64_t>(1, (256ull << 20) / v->row_bytes());
        ORAMA_TRY(sc->misc0.reserve((size_t)m * sizeof(uint64_t)));
        ORAMA_TRY(sc->misc1.reserve((size_t)std::min(slab, m) * v->row_bytes()));
        ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, live.data(), (size_t)m * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        for (uint64_t i0 = 0; i0 < m; i0 += slab) {
            const uint64_t cnt = std::min(slab, m - i0);
            if (v->f16())
                ORAMA_TRY(launch_f16_gather_rows(v->rows.base, sc->misc0.as<uint64_t>() + i0, cnt, v->dim,
                                                 sc->misc1.as<float>(), s));
            else
                ORAMA_TRY(launch_gather_rows_f32(v->rows.as<float>(), sc->misc0.as<uint64_t>() + i0, cnt, v->dim,
                                                 sc->misc1.as<float>(), s));
            ORAMA_TRY(store_rows_from_device(v, nv->rows.base, nv->inv_norm.as<float>(), sc->misc1.as<float>(), i0, cnt, s));
        }
        hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks_for_rows(m)), dim3(256), 0, s, v->row_doc.as<uint64_t>(),
                           sc->misc0.as<uint64_t>(), m, nv->row_doc.as<uint64_t>());
        ORAMA_HIP_TRY(hipGetLastError());
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    auto swap_buf = [](GrowBuf& a, GrowBuf& b) {
        std::swap(a.base, b.base);
        std::swap(a.va_bytes, b.va_bytes);
        std::swap(a.mapped, b.mapped);
        std::swap(a.chunk, b.chunk);
        std::swap(a.vmm, b.vmm);
        std::swap(a.device, b.device);
        a.handles.swap(b.handles);
        a.handle_bytes.swap(b.handle_bytes);
    };
    swap_buf(v->rows, nv->rows);
    swap_buf(v->inv_norm, nv->inv_norm);
    swap_buf(v->row_doc, nv->row_doc);
    swap_buf(v->dead, nv->dead);
    v->cap_rows = nv->cap_rows;
    v->n_rows.store(m, std::memory_order_release);
    v->n_dead.store(0, std::memory_order_release);
    return ORAMA_OK;  // nv (holding the old arrays) is released here
}

int orama_vec_info(orama_vec* v, orama_vec_info_t* out) {
    ORAMA_REQUIRE(v && out, "null argument");
    std::shared_lock<std::shared_mutex> lk(v->mu);
    const uint64_t n = v->n_rows.load(std::memory_order_acquire), d = v->n_dead.load(std::memory_order_acquire);
    out->dimensions = v->dim;
    out->num_rows = n;
    out->num_embeddings = n - d;
    out->pending_ops = d;
    out->version = v->version;
    out->hbm_bytes = (uint64_t)(v->rows.mapped + v->inv_norm.mapped + v->row_doc.mapped + v->dead.mapped);
    out->two_stage_queries = v->two_stage_queries.load(std::memory_order_relaxed);
    out->two_stage_fallbacks = v->two_stage_fallbacks.load(std::memory_order_relaxed);
    if (v->shadow) {
        const orama_vec* sh = v->shadow.get();
        out->hbm_bytes += (uint64_t)(sh->rows.mapped + sh->inv_norm.mapped + sh->row_doc.mapped + sh->dead.mapped);
    }
    return ORAMA_OK;
}

int orama_vec_search(orama_vec* v, const float* queries, uint32_t q, uint32_t k,
                     const uint64_t* allow_bitmap, uint64_t bitmap_bits, uint64_t* out_ids,
                     float* out_dist, uint32_t* out_n) {
    ORAMA_REQUIRE(v, "null handle");
    ORAMA_REQUIRE(q >= 1 && queries && out_ids && out_dist && out_n, "null argument");
    for (uint32_t i = 0; i < q; ++i) out_n[i] = 0;
    if (k == 0) return ORAMA_OK;  // limit 0: empty result, like the reference's CappedHeap(0)
    ORAMA_SUPPORT(k <= kSelectMaxK, "limit %u exceeds the supported maximum %u", k, kSelectMaxK);
    ORAMA_ON_DEVICE(v->ctx->device);
    std::shared_lock<std::shared_mutex> lk(v->mu);
    if (v->n_rows.load(std::memory_order_acquire) == 0) return ORAMA_OK;
    // fp32 rows + fp16 shadow: candidates from the shadow scan, exact distances from the fp32 rows (two_stage_search)
    const bool two_stage = vec_two_stage_usable(v, queries, q, k);
    ScratchLease sc(v->ctx, kScratchVector), sc2(v->ctx, kScratchVector);
    if (two_stage) ORAMA_TRY(ScratchLease::init_pair(sc, sc2));  // both at once: callers holding one set each cannot wait for each other
    else ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const size_t qbytes = (size_t)q * v->dim * sizeof(float);
    ORAMA_TRY(sc->query.reserve(qbytes));
    ORAMA_TRY(sc->h_in.reserve(qbytes));
    memcpy(sc->h_in.p, queries, qbytes);
    ORAMA_TRY(stage_block(v->ctx, sc->query.p, sc->h_in.p, qbytes, hipMemcpyHostToDevice, s));
    const uint64_t* d_allow = nullptr;
    ORAMA_TRY(resolve_allow(v->ctx, sc.s.get(), allow_bitmap, bitmap_bits, s, &d_allow));
    const size_t nk = (size_t)q * k;
    ORAMA_TRY(sc->out_ids.reserve(nk * 8));
    ORAMA_TRY(sc->out_val.reserve(nk * 4));
    ORAMA_TRY(sc->out_n.reserve((size_t)q * 4));
    ORAMA_TRY(sc->h_out.reserve(nk * 12 + (size_t)q * 4));
    char* h = sc->h_out.as<char>();
    const StagePart back[3] = {{h, sc->out_ids.p, nk * 8}, {h + nk * 8, sc->out_val.p, nk * 4}, {h + nk * 12, sc->out_n.p, (size_t)q * 4}};
    if (two_stage) {
        // the read-back is enqueued behind the plan BEFORE the host looks at the proof words: one wake-up for the call (round 5:
        // flags, wake-up, read-back, wake-up cost a lone query 35 us); a query that was not proven — rare — is answered again
        // by the plain scan inside finish(), and only then the read-back is repeated
        VecTwoStage call;
        ORAMA_TRY(call.begin(v, sc, sc2, sc->query.as<float>(), q, k, d_allow, bitmap_bits, sc->out_ids.as<uint64_t>(),
                             sc->out_val.as<float>(), sc->out_n.as<uint32_t>()));
        ORAMA_TRY(stage_blocks(v->ctx, back, 3, hipMemcpyDeviceToHost, s));
        bool reran = false;
        ORAMA_TRY(call.finish(&reran));
        if (reran) ORAMA_TRY(stage_blocks(v->ctx, back, 3, hipMemcpyDeviceToHost, s));
    } else if (!v->f16() && v->ctx->stage_by_kernel && v->ctx->direct_out) {
        // fp32 rows: the selection's last launch only WRITES the answers — straight into the pinned block (no read-back launch)
        ORAMA_TRY(search_enqueue(v, sc.s.get(), sc->query.as<float>(), q, k, d_allow, bitmap_bits, reinterpret_cast<uint64_t*>(h),
                                 reinterpret_cast<float*>(h + nk * 8), reinterpret_cast<uint32_t*>(h + nk * 12), s));
    } else {
        ORAMA_TRY(search_enqueue(v, sc.s.get(), sc->query.as<float>(), q, k, d_allow, bitmap_bits,
                                 sc->out_ids.as<uint64_t>(), sc->out_val.as<float>(), sc->out_n.as<uint32_t>(), s));
        ORAMA_TRY(stage_blocks(v->ctx, back, 3, hipMemcpyDeviceToHost, s));  // one launch, no copy engine (stage.hip)
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    memcpy(out_ids, h, nk * 8);
    memcpy(out_dist, h + nk * 8, nk * 4);
    memcpy(out_n, h + nk * 12, (size_t)q * 4);
    return ORAMA_OK;
}

// Device-resident search on the caller's stream(s); v->mu is held shared by the caller.  A store with an fp16 shadow takes the
// two-stage plan in its device form (two_stage_search) — everything on the tail stream `s`: the fp16 pipeline interleaves
// scans and selections with dependencies in both directions.
static int device_search_enqueue(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k, const uint64_t* d_allow,
                                 uint64_t allow_bits, uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n, hipStream_t s,
                                 hipStream_t s_scan, bool two_streams) {
    const bool two_stage = two_stage_device_usable(v, q, k);
    Scratch *sc = nullptr, *sc2 = nullptr;
    {
        std::lock_guard<std::mutex> g(v->dev_mu);
        auto& slot = v->dev_scratch[s];
        if (!slot) slot.reset(new Scratch());
        sc = slot.get();
        if (two_stage) {
            auto& slot2 = v->dev_scratch2[s];
            if (!slot2) slot2.reset(new Scratch());
            sc2 = slot2.get();
        }
    }
    if (two_stage) {
        std::shared_lock<std::shared_mutex> sl(v->shadow->mu);
        return two_stage_search(v, sc, sc2, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, s, nullptr);
    }
    return search_enqueue(v, sc, d_queries, q, k, d_allow, allow_bits, d_out_ids, d_out_dist, d_out_n, s, s_scan, two_streams);
}

int orama_vec_search_device(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                            const uint64_t* d_allow_bitmap, uint64_t bitmap_bits,
                            uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                            void* hip_stream) {
    ORAMA_REQUIRE(v, "null handle");
    ORAMA_REQUIRE(q >= 1 && d_queries && d_out_ids && d_out_dist && d_out_n, "null argument");
    ORAMA_REQUIRE(k >= 1, "limit is 0");
    ORAMA_SUPPORT(k <= kSelectMaxK, "limit %u outside [1, %u]", k, kSelectMaxK);
    ORAMA_ON_DEVICE(v->ctx->device);
    hipStream_t s = (hipStream_t)hip_stream;
    std::shared_lock<std::shared_mutex> lk(v->mu);
    return device_search_enqueue(v, d_queries, q, k, d_allow_bitmap, bitmap_bits, d_out_ids, d_out_dist, d_out_n, s, nullptr, false);
}

int orama_vec_search_packed_device2(orama_vec* v, const float* d_queries, uint32_t q, uint32_t k,
                                    const uint64_t* d_allow_bitmap, uint64_t bitmap_bits, void* d_packed_block,
                                    uint32_t* d_out_n, void* scan_stream, void* tail_stream) {
    ORAMA_REQUIRE(v && d_packed_block, "null argument");
    ORAMA_REQUIRE(q >= 1 && d_queries && d_out_n, "null argument");
    ORAMA_REQUIRE(k >= 1, "limit is 0");
    ORAMA_SUPPORT(k <= kSelectMaxK, "limit %u outside [1, %u]", k, kSelectMaxK);
    ORAMA_ON_DEVICE(v->ctx->device);
    hipStream_t s = (hipStream_t)tail_stream, ss = (hipStream_t)scan_stream;
    std::shared_lock<std::shared_mutex> lk(v->mu);
    char* base = reinterpret_cast<char*>(d_packed_block);
    return device_search_enqueue(v, d_queries, q, k, d_allow_bitmap, bitmap_bits, reinterpret_cast<uint64_t*>(base),
                                 reinterpret_cast<float*>(base + (uint64_t)q * k * 8), d_out_n, s, ss, true);
}

int orama_merge_candidates_device(orama_ctx* ctx, const uint64_t* d_ids, const float* d_dist,
                                  uint32_t lists, uint32_t q, uint32_t k, uint64_t* d_out_ids,
                                  float* d_out_dist, uint32_t* d_out_n, void* hip_stream) {
    ORAMA_REQUIRE(ctx && d_ids && d_dist && d_out_ids && d_out_dist, "null argument");
    ORAMA_ON_DEVICE(ctx->device);
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
    ORAMA_ON_DEVICE(ctx->device);
    return launch_merge_packed(ctx, d_packed_blocks, lists, q, k, d_out_ids, d_out_dist, d_out_n,
                               (hipStream_t)hip_stream);
}

int orama_vec_fill_synthetic(orama_vec* v, uint64_t n_rows, uint64_t seed, uint64_t first_doc_id) {
    ORAMA_REQUIRE(v, "null handle");
    if (!v->shadow) return vec_fill_synthetic_one(v, n_rows, seed, first_doc_id);
    std::lock_guard<std::mutex> cl(v->composite_mu);
    ORAMA_TRY(vec_fill_synthetic_one(v->shadow.get(), n_rows, seed, first_doc_id));  // same generator, same row order
    return vec_fill_synthetic_one(v, n_rows, seed, first_doc_id);
}

static int vec_fill_synthetic_one(orama_vec* v, uint64_t n_rows, uint64_t seed, uint64_t first_doc_id) {
    ORAMA_REQUIRE(v, "null handle");
    if (n_rows == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(v->ctx->device);
    std::lock_guard<std::mutex> wl(v->write_mu);
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    const uint64_t first = v->n_rows.load(std::memory_order_relaxed);
    ORAMA_TRY(reserve_rows_locked(v, first + n_rows, s));
    std::shared_lock<std::shared_mutex> r(v->mu);
    if (v->f16()) {
        // generate f32 slabs in scratch with the SAME generator (row index = store row), then quantise
        const uint64_t slab = std::max<uint64_t>(32, ((512ull << 20) / v->row_bytes()) & ~31ull);
        ORAMA_TRY(sc->misc1.reserve((size_t)std::min(slab, n_rows) * v->row_bytes()));
        for (uint64_t r0 = 0; r0 < n_rows; r0 += slab) {
            const uint64_t cnt = std::min(slab, n_rows - r0);
            // rows are generated at virtual positions first+r0.. by offsetting the base pointer
            float* virt = sc->misc1.as<float>() - (first + r0) * (uint64_t)v->dim;
            ORAMA_TRY(launch_synth_fill_f32(virt, first + r0, cnt, v->dim, seed, s));
            ORAMA_TRY(store_rows_from_device(v, v->rows.base, v->inv_norm.as<float>(), sc->misc1.as<float>(),
                                             first + r0, cnt, s));
        }
    } else {
        ORAMA_TRY(launch_synth_fill_f32(v->rows.as<float>(), first, n_rows, v->dim, seed, s));
        if (v->metric == ORAMA_METRIC_COSINE)
            ORAMA_TRY(launch_row_inv_norm_f32(v->rows.as<float>(), first, n_rows, v->dim, v->inv_norm.as<float>(), s));
    }
    ORAMA_TRY(launch_iota_u64(v->row_doc.as<uint64_t>() + first, n_rows, first_doc_id, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    v->n_rows.store(first + n_rows, std::memory_order_release);
    return ORAMA_OK;
}

int orama_vec_get_rows(orama_vec* v, const uint64_t* row_idx, uint64_t n, float* out_rows,
                       uint64_t* out_doc_ids) {
    ORAMA_REQUIRE(v, "null handle");
    if (n == 0) return ORAMA_OK;
    ORAMA_REQUIRE(row_idx && out_rows, "null argument");
    ORAMA_ON_DEVICE(v->ctx->device);
    std::shared_lock<std::shared_mutex> lk(v->mu);
    const uint64_t rows = v->n_rows.load(std::memory_order_acquire);
    for (uint64_t i = 0; i < n; ++i)
        ORAMA_REQUIRE(row_idx[i] < rows, "row %llu out of range", (unsigned long long)row_idx[i]);
    ScratchLease sc(v->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    ORAMA_TRY(sc->misc0.reserve((size_t)n * 8));
    ORAMA_TRY(sc->misc1.reserve((size_t)n * v->row_bytes()));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, row_idx, (size_t)n * 8, hipMemcpyHostToDevice, s));
    if (v->f16())
        ORAMA_TRY(launch_f16_gather_rows(v->rows.base, sc->misc0.as<uint64_t>(), n, v->dim, sc->misc1.as<float>(), s));
    else
        ORAMA_TRY(launch_gather_rows_f32(v->rows.as<float>(), sc->misc0.as<uint64_t>(), n, v->dim,
                                         sc->misc1.as<float>(), s));
    ORAMA_HIP_TRY(hipMemcpyAsync(out_rows, sc->misc1.p, (size_t)n * v->row_bytes(), hipMemcpyDeviceToHost, s));
    if (out_doc_ids) {
        ORAMA_TRY(sc->misc2.reserve((size_t)n * 8));
        hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks_for_rows(n)), dim3(256), 0, s, v->row_doc.as<uint64_t>(),
                           sc->misc0.as<uint64_t>(), n, sc->misc2.as<uint64_t>());
        ORAMA_HIP_TRY(hipGetLastError());
        ORAMA_HIP_TRY(hipMemcpyAsync(out_doc_ids, sc->misc2.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    }
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    return ORAMA_OK;
}

}  // extern "C"
