// facets.hip — facet counting and group-by top-k over the HBM-resident score map.
//
// The reference hands the whole HashMap<DocumentId, f32> of a search to FacetContext / GroupContext
// (src/collection_manager/sides/read/search.rs:355-400, index/facet.rs:35-209, index/group.rs:107-170, sort.rs:203-213):
//   facet  : per field value (bool / string key) or number range, |{docs with that value} ∩ keys(token_scores)|
//   group  : per value combination, the best `max_results` docs of the group by score (NaN skipped)
// Here the map never leaves the GPU: it is the candidate list the scorer left behind —
//   emit[doc] = {epoch:32 | position:32} for docs in the map, cand_idx[pos] = doc, cand_score[pos] = score
// — and the field values are resident too (one u32 local doc index per (value, doc) entry, buckets contiguous), so a
// facet is one streaming pass over the field's entries with a gather into `emit` (HBM-bound: 4 B per entry + one
// 8-B gather per entry), and a group-by is one workgroup per group streaming its entries through an LDS-resident
// running top-k of 64-bit keys  ordered(score) << 32 | ~doc  (score desc, DocumentId asc — local order = id order).
#include "bm25_kernels.hpp"

#include "device_utils.hpp"

namespace orama {

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kNoDoc = 0xffffffffu;

__device__ __forceinline__ bool in_map(const ScoreMapDev& m, uint32_t doc, uint32_t* pos_out) {
    const unsigned long long e = m.emit[doc];
    if ((uint32_t)(e >> 32) != m.epoch) return false;
    const uint32_t pos = (uint32_t)e;
    if (m.cand_idx[pos] != doc) return false;  // emitted, then dropped (never happens today; cheap to keep exact)
    *pos_out = pos;
    return true;
}

// bucket of entry i: largest b with off[b] <= i
__device__ __forceinline__ uint32_t bucket_of(const uint64_t* __restrict__ off, uint32_t n_buckets, uint64_t i) {
    uint32_t lo = 0, hi = n_buckets;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads) void facet_count_buckets_kernel(ScoreMapDev m, const uint32_t* __restrict__ entry_doc,
                                                                       const uint64_t* __restrict__ bucket_off,
                                                                       uint32_t n_buckets, uint64_t n_entries,
                                                                       unsigned long long* __restrict__ counts) {
    for (uint64_t i0 = (uint64_t)blockIdx.x * kThreads; i0 < n_entries; i0 += (uint64_t)gridDim.x * kThreads) {
        const uint64_t i = i0 + threadIdx.x;
        bool hit = false;
        uint32_t b = 0;
        if (i < n_entries) {
            const uint32_t doc = entry_doc[i];
            uint32_t pos;
            hit = doc != kNoDoc && in_map(m, doc, &pos);
            if (hit) b = bucket_of(bucket_off, n_buckets, i);
        }
        // entries are bucket-contiguous: the hits of a wave fall into a handful of buckets — one atomic per bucket
        unsigned long long live = __ballot(hit);
        while (live) {
            const int leader = __ffsll((long long)live) - 1;
            const uint32_t lb = (uint32_t)__shfl((int)b, leader, 64);
            const unsigned long long same = __ballot(hit && b == lb);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&counts[lb], (unsigned long long)__popcll(same));
            live &= ~same;
            if (hit && b == lb) hit = false;
        }
    }
}

constexpr uint32_t kMaxRanges = 64;

__global__ __launch_bounds__(kThreads) void facet_count_ranges_kernel(ScoreMapDev m, const uint32_t* __restrict__ entry_doc,
                                                                      const double* __restrict__ entry_val, uint64_t n_entries,
                                                                      const double* __restrict__ from,
                                                                      const double* __restrict__ to, uint32_t n_ranges,
                                                                      unsigned long long* __restrict__ counts) {
    __shared__ double s_from[kMaxRanges], s_to[kMaxRanges];
    __shared__ uint32_t s_cnt[kMaxRanges];
    for (uint32_t r = threadIdx.x; r < n_ranges; r += kThreads) {
        s_from[r] = from[r];
        s_to[r] = to[r];
        s_cnt[r] = 0;
    }
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n_entries; i += (uint64_t)gridDim.x * kThreads) {
        const uint32_t doc = entry_doc[i];
        uint32_t pos;
        if (doc == kNoDoc || !in_map(m, doc, &pos)) continue;
        const double v = entry_val[i];
        for (uint32_t r = 0; r < n_ranges; ++r)
            if (s_from[r] <= v && v <= s_to[r]) atomicAdd(&s_cnt[r], 1u);  // BetweenInclusive, number_field.rs:604-631
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < n_ranges; r += kThreads)
        if (s_cnt[r]) atomicAdd(&counts[r], (unsigned long long)s_cnt[r]);
}

// ---------------------------------------------------------------- group-by: per-bucket top-k
constexpr int kGroupThreads = 1024;
constexpr uint32_t kGroupCap = 4096;  // LDS keys: running best k + staged candidates

__device__ void lds_sort_keys_desc(unsigned long long* keys, uint32_t p2) {
    for (uint32_t size = 2; size <= p2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < (p2 >> 1); t += blockDim.x) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;  // "up" block: larger key first
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((b > a) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kGroupThreads) void group_top_kernel(ScoreMapDev m, const uint32_t* __restrict__ entry_doc,
                                                                  const uint64_t* __restrict__ bucket_off, uint32_t k,
                                                                  uint64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                                  uint32_t* __restrict__ out_n) {
    __shared__ unsigned long long keys[kGroupCap];
    __shared__ uint32_t fill;
    const uint32_t g = blockIdx.x;
    const uint64_t begin = bucket_off[g], end = bucket_off[g + 1];
    if (threadIdx.x == 0) fill = 0;
    __syncthreads();
    // round: stage up to kGroupCap - fill new candidates behind the current best, sort, keep the best k
    const uint32_t batch = kGroupCap - k;  // entries examined per round never overflow the staging area
    for (uint64_t base = begin; base < end; base += batch) {
        const uint64_t stop = base + batch < end ? base + batch : end;
        for (uint64_t i = base + threadIdx.x; i < stop; i += kGroupThreads) {
            const uint32_t doc = entry_doc[i];
            uint32_t pos;
            if (doc == kNoDoc || !in_map(m, doc, &pos)) continue;
            const float s = m.cand_score[pos];
            if (s != s) continue;  // NotNan::new(..) Err -> skipped (sort.rs:207)
            const uint32_t slot = atomicAdd(&fill, 1u);
            keys[slot] = ((unsigned long long)f32_to_ordered(s) << 32) | (unsigned long long)(uint32_t)(~doc);
        }
        __syncthreads();
        const uint32_t n = fill;
        if (n > k || stop == end) {
            uint32_t p2 = 2;
            while (p2 < n) p2 <<= 1;
            for (uint32_t i = n + threadIdx.x; i < p2; i += kGroupThreads) keys[i] = 0ull;  // below every real key
            __syncthreads();
            lds_sort_keys_desc(keys, p2);
            if (threadIdx.x == 0) fill = n < k ? n : k;
            __syncthreads();
        }
    }
    const uint32_t n = fill < k ? fill : k;
    for (uint32_t i = threadIdx.x; i < k; i += kGroupThreads) {
        if (i < n) {
            const unsigned long long key = keys[i];
            const uint32_t doc = ~(uint32_t)key;
            out_ids[(uint64_t)g * k + i] = m.docs ? m.docs[doc] : m.dense_base + doc;
            out_scores[(uint64_t)g * k + i] = ordered_to_f32((uint32_t)(key >> 32));
        } else {
            out_ids[(uint64_t)g * k + i] = ~0ull;
            out_scores[(uint64_t)g * k + i] = -__builtin_huge_valf();
        }
    }
    if (threadIdx.x == 0) out_n[g] = n;
}

// the whole map as (DocumentId, score) pairs.  The ORDER of the pairs is unspecified (slots are handed out by an atomic
// cursor) — like iterating the reference's HashMap<DocumentId, f32>; callers that need an order sort by id.
__global__ __launch_bounds__(kThreads) void scores_export_kernel(ScoreMapDev m, uint32_t list_len, uint64_t* __restrict__ out_ids,
                                                                 float* __restrict__ out_scores, uint32_t* __restrict__ cursor) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < list_len; i += gridDim.x * kThreads) {
        const uint32_t doc = m.cand_idx[i];
        if (doc == kNoDoc) continue;
        const uint32_t slot = atomicAdd(cursor, 1u);
        out_ids[slot] = m.docs ? m.docs[doc] : m.dense_base + doc;
        out_scores[slot] = m.cand_score[i];
    }
}

__global__ __launch_bounds__(kThreads) void scores_lookup_kernel(ScoreMapDev m, const uint32_t* __restrict__ doc, uint32_t n,
                                                                 float* __restrict__ out, uint8_t* __restrict__ present) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    uint32_t pos;
    const bool ok = doc[i] != kNoDoc && in_map(m, doc[i], &pos);
    present[i] = ok ? 1 : 0;
    out[i] = ok ? m.cand_score[pos] : 0.0f;
}

uint32_t grid_for(uint64_t n, uint32_t max_blocks) {
    uint64_t b = (n + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    return (uint32_t)(b > max_blocks ? max_blocks : b);
}

}  // namespace

int launch_facet_count_buckets(const ScoreMapDev& m, const uint32_t* d_entry_doc, const uint64_t* d_bucket_off,
                               uint32_t n_buckets, uint64_t n_entries, unsigned long long* d_counts, hipStream_t s) {
    ORAMA_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)n_buckets * 8, s));
    if (n_entries == 0 || n_buckets == 0) return ORAMA_OK;
    hipLaunchKernelGGL(facet_count_buckets_kernel, dim3(grid_for(n_entries, 4096)), dim3(kThreads), 0, s, m, d_entry_doc,
                       d_bucket_off, n_buckets, n_entries, d_counts);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_facet_count_ranges(const ScoreMapDev& m, const uint32_t* d_entry_doc, const double* d_entry_val, uint64_t n_entries,
                              const double* d_from, const double* d_to, uint32_t n_ranges, unsigned long long* d_counts,
                              hipStream_t s) {
    ORAMA_SUPPORT(n_ranges <= kMaxRanges, "at most %u ranges per facet call", kMaxRanges);
    ORAMA_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)(n_ranges ? n_ranges : 1) * 8, s));
    if (n_entries == 0 || n_ranges == 0) return ORAMA_OK;
    hipLaunchKernelGGL(facet_count_ranges_kernel, dim3(grid_for(n_entries, 2048)), dim3(kThreads), 0, s, m, d_entry_doc,
                       d_entry_val, n_entries, d_from, d_to, n_ranges, d_counts);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_group_top(const ScoreMapDev& m, const uint32_t* d_entry_doc, const uint64_t* d_bucket_off, uint32_t n_buckets,
                     uint32_t k, uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_n, hipStream_t s) {
    ORAMA_REQUIRE(k >= 1, "group max_results is 0");
    ORAMA_SUPPORT(k <= kGroupMaxK, "group max_results %u outside [1, %u]", k, kGroupMaxK);
    if (n_buckets == 0) return ORAMA_OK;
    hipLaunchKernelGGL(group_top_kernel, dim3(n_buckets), dim3(kGroupThreads), 0, s, m, d_entry_doc, d_bucket_off, k,
                       d_out_ids, d_out_scores, d_out_n);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_scores_export(const ScoreMapDev& m, uint32_t list_len, uint64_t* d_out_ids, float* d_out_scores,
                         uint32_t* d_cursor, hipStream_t s) {
    ORAMA_HIP_TRY(hipMemsetAsync(d_cursor, 0, 4, s));
    if (list_len == 0) return ORAMA_OK;
    hipLaunchKernelGGL(scores_export_kernel, dim3(grid_for(list_len, 2048)), dim3(kThreads), 0, s, m, list_len, d_out_ids,
                       d_out_scores, d_cursor);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_scores_lookup(const ScoreMapDev& m, const uint32_t* d_doc, uint32_t n, float* d_out, uint8_t* d_present,
                         hipStream_t s) {
    if (n == 0) return ORAMA_OK;
    hipLaunchKernelGGL(scores_lookup_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, m, d_doc, n, d_out,
                       d_present);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
