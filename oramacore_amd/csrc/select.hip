// select.hip — K4 / K6: exact, deterministic top-k on gfx950.
//
// Algorithm: MSB-first radix select on a 64-bit composite key
//     key = ordered(value) << 32 | ~idx          ("larger key wins")
// so that equal values are ordered by ascending idx and every key is unique.  Three histogram
// passes (11/11/10 bits) fix the value threshold; three more passes over the idx bits run only
// when several equal values straddle the k boundary (they early-exit on `done` otherwise).  A
// collect pass appends exactly k' = min(k, #non-NaN) keys, and one workgroup sorts them in LDS
// with the full tie order (value, 64-bit id asc, idx asc).
//
// Roofline: HBM — each pass streams the value array once (4 B/element); the final sort is
// LDS-resident.  For the dense single-query vector path this is 4 x N x 4 B on top of the
// N x D x 4 B corpus pass (0.5 % at D = 768).
//
// That multi-pass form serves long dense lists (millions of values).  The lists the hot paths produce take one or two
// launches of the forms further down, all built on one cut: the candidates of a round stay in registers, a floor (the k-th
// best so far, a list's shared running bound, or a bound from the lanes' best keys) decides what reaches LDS, and a
// couple of hundred survivors are cut by counting ranks —
//   keys_reduce_kernel + keys_final_kernel    lists of 64-bit keys (K1's per-wave lists, K3r's one slot per posting),
//   pairs_reduce_kernel (+ keys_final_kernel)  (value, index) lists: the candidate lists and the dense heads of the fp16
//                                              scans; a list of one round is finished by the workgroup that reduces it.
#include "select.hpp"

// This file is compiled twice: as the unit "select" (ORAMA_SELECT_UNIT 1, everything but the two kernels of the second half of
// round 5) and, included by select_wide.hip, as a unit that holds ONLY those two (select_tiny_kernel, pairs_reduce_wide_kernel)
// and the device helpers they share with the rest.  Why: with the two kernels in THIS code object the key-list kernels of a lone
// full-text call — same source, same instruction counts, same registers — ran 6 us longer per call (event span around
// keys_reduce + keys_final 27.6 -> 33.8 us, 14.5 K -> 13.4 K single calls per second; linking the unit without them gave the
// 27.6 us back: profiles/r05_select_unit_split.log).  The helpers are internal to each unit (anonymous namespace).
#ifndef ORAMA_SELECT_UNIT
#define ORAMA_SELECT_UNIT 1
#endif

#include <algorithm>
#include <cstdlib>

#include "device_utils.hpp"

namespace orama {

namespace {

__constant__ const uint32_t kShift[6] = {53, 42, 32, 21, 10, 0};
__constant__ const uint32_t kBits[6] = {11, 11, 10, 11, 11, 10};

constexpr int kHistThreads = 256;
constexpr int kSortThreads = 1024;

// phase stamps of the key-list kernels: defined by scripts/micro/keys_reduce_probe.hip (which includes this file), nothing here
#ifndef ORAMA_KEYS_STAMP
#define ORAMA_KEYS_STAMP(i) ((void)0)
#define ORAMA_KEYS_NOTE(i, v) ((void)0)
#endif

__device__ __forceinline__ unsigned long long make_key(float v, uint32_t idx, bool descending) {
    uint32_t o = f32_to_ordered(v);
    uint32_t hi = descending ? o : ~o;
    return ((unsigned long long)hi << 32) | (unsigned long long)(uint32_t)(~idx);
}

#if ORAMA_SELECT_UNIT == 1
// ---------------------------------------------------------------- histogram pass
__global__ __launch_bounds__(kHistThreads) void select_hist_kernel(
    const float* __restrict__ vals, const uint32_t* __restrict__ idx, uint64_t stride,
    const uint32_t* __restrict__ n_dev, uint32_t n_max, bool descending, SelectState* state,
    int pass) {
    const uint32_t qi = blockIdx.y;
    SelectState* st = state + qi;
    if (st->done) return;
    const uint32_t n = n_dev ? min(n_dev[qi], n_max) : n_max;
    const float* v = vals + (uint64_t)qi * stride;
    const uint32_t* ix = idx ? idx + (uint64_t)qi * stride : nullptr;

    __shared__ uint32_t hist[2048];
    for (int i = threadIdx.x; i < 2048; i += kHistThreads) hist[i] = 0;
    __syncthreads();

    const uint32_t shift = kShift[pass];
    const uint32_t mask = (1u << kBits[pass]) - 1u;
    const unsigned long long prefix = st->prefix;
    const uint32_t pshift = shift + kBits[pass];

    auto visit = [&](float x, uint32_t id) {
        if (x != x) return;
        unsigned long long key = make_key(x, id, descending);
        if (pass > 0 && (key >> pshift) != prefix) return;
        atomicAdd(&hist[(uint32_t)(key >> shift) & mask], 1u);
    };

    const uint32_t tid = blockIdx.x * kHistThreads + threadIdx.x;
    const uint32_t nthreads = gridDim.x * kHistThreads;
    const bool vec = (ix == nullptr) && ((stride & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(vals) & 15) == 0);
    if (vec) {
        const uint32_t n4 = n >> 2;
        const float4* v4 = reinterpret_cast<const float4*>(v);
        for (uint32_t i = tid; i < n4; i += nthreads) {
            float4 x = v4[i];
            visit(x.x, 4 * i + 0);
            visit(x.y, 4 * i + 1);
            visit(x.z, 4 * i + 2);
            visit(x.w, 4 * i + 3);
        }
        for (uint32_t i = (n4 << 2) + tid; i < n; i += nthreads) visit(v[i], i);
    } else {
        for (uint32_t i = tid; i < n; i += nthreads) visit(v[i], ix ? ix[i] : i);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += kHistThreads) {
        uint32_t c = hist[i];
        if (c) atomicAdd(&st->hist[pass][i], c);
    }
}

// ---------------------------------------------------------------- pick the bin
__global__ __launch_bounds__(256) void select_scan_kernel(SelectState* state, uint32_t k, int pass) {
    SelectState* st = state + blockIdx.x;
    if (st->done) return;
    const uint32_t* h = st->hist[pass];
    __shared__ uint32_t suffix[257];
    const int t = threadIdx.x;
    uint32_t local[8];
    uint32_t s = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        local[b] = h[t * 8 + b];
        s += local[b];
    }
    suffix[t] = s;
    if (t == 0) suffix[256] = 0;
    __syncthreads();
    // inclusive suffix sum over threads (Hillis–Steele)
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t add = (t + off < 256) ? suffix[t + off] : 0;
        __syncthreads();
        suffix[t] += add;
        __syncthreads();
    }
    __shared__ uint32_t rem_s;
    if (t == 0) {
        if (pass == 0) {
            uint32_t valid = suffix[0];
            st->valid = valid;
            st->kprime = valid < k ? valid : k;
            st->remaining = st->kprime;
        }
        rem_s = st->remaining;
    }
    __syncthreads();
    const uint32_t rem = rem_s;
    if (rem == 0) {  // nothing to select (k' == 0)
        if (t == 0) {
            st->done = 1;
            st->sel_shift = 0;
            st->prefix = ~0ull;
        }
        return;
    }
    const uint32_t above = suffix[t + 1];  // elements in bins owned by higher threads
    if (suffix[t] >= rem && above < rem) {
        uint32_t cum = above;
#pragma unroll
        for (int b = 7; b >= 0; --b) {
            if (cum + local[b] >= rem) {
                const uint32_t bin = (uint32_t)(t * 8 + b);
                const uint32_t r2 = rem - cum;
                st->prefix = (st->prefix << kBits[pass]) | (unsigned long long)bin;
                st->remaining = r2;
                st->sel_shift = kShift[pass];
                if (local[b] == r2 || pass == 5) st->done = 1;
                break;
            }
            cum += local[b];
        }
    }
}

// ---------------------------------------------------------------- collect the k' winners
__global__ __launch_bounds__(kHistThreads) void select_collect_kernel(
    const float* __restrict__ vals, const uint32_t* __restrict__ idx, uint64_t stride,
    const uint32_t* __restrict__ n_dev, uint32_t n_max, bool descending, SelectState* state,
    unsigned long long* __restrict__ keys, uint32_t kcap) {
    const uint32_t qi = blockIdx.y;
    SelectState* st = state + qi;
    const uint32_t kprime = st->kprime;
    if (kprime == 0) return;
    const uint32_t n = n_dev ? min(n_dev[qi], n_max) : n_max;
    const float* v = vals + (uint64_t)qi * stride;
    const uint32_t* ix = idx ? idx + (uint64_t)qi * stride : nullptr;
    unsigned long long* out = keys + (uint64_t)qi * kcap;
    const unsigned long long prefix = st->prefix;
    const uint32_t shift = st->sel_shift;

    auto visit = [&](float x, uint32_t id) {
        if (x != x) return;
        unsigned long long key = make_key(x, id, descending);
        if ((key >> shift) >= prefix) {
            uint32_t pos = atomicAdd(&st->out_count, 1u);
            if (pos < kcap) out[pos] = key;
        }
    };
    const uint32_t tid = blockIdx.x * kHistThreads + threadIdx.x;
    const uint32_t nthreads = gridDim.x * kHistThreads;
    const bool vec = (ix == nullptr) && ((stride & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(vals) & 15) == 0);
    if (vec) {
        const uint32_t n4 = n >> 2;
        const float4* v4 = reinterpret_cast<const float4*>(v);
        for (uint32_t i = tid; i < n4; i += nthreads) {
            float4 x = v4[i];
            visit(x.x, 4 * i + 0);
            visit(x.y, 4 * i + 1);
            visit(x.z, 4 * i + 2);
            visit(x.w, 4 * i + 3);
        }
        for (uint32_t i = (n4 << 2) + tid; i < n; i += nthreads) visit(v[i], i);
    } else {
        for (uint32_t i = tid; i < n; i += nthreads) visit(v[i], ix ? ix[i] : i);
    }
}

#endif
// ---------------------------------------------------------------- LDS bitonic sort
// Sort record: hi (ordered value key, larger first), id (64-bit, smaller first), idx (smaller first).
struct SortLds {
    uint32_t hi[kSelectMaxK];
    uint32_t idx[kSelectMaxK];
    uint64_t id[kSelectMaxK];
};

__device__ __forceinline__ bool rec_before(uint32_t ha, uint64_t ia, uint32_t xa, uint32_t hb,
                                           uint64_t ib, uint32_t xb) {
    if (ha != hb) return ha > hb;
    if (ia != ib) return ia < ib;
    return xa < xb;
}

// USE_ID = false orders by (value, index) only — the order in which the k-th boundary is cut (DESIGN §3 rule 4)
template <bool USE_ID = true>
__device__ void lds_bitonic_sort(SortLds& s, uint32_t p2) {
    for (uint32_t size = 2; size <= p2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < (p2 >> 1); t += blockDim.x) {
                uint32_t lo = 2 * t - (t & (stride - 1));
                uint32_t hi = lo + stride;
                bool up = ((lo & size) == 0);  // "up" block: best first
                uint32_t ha = s.hi[lo], hb = s.hi[hi];
                uint64_t ia = s.id[lo], ib = s.id[hi];
                uint32_t xa = s.idx[lo], xb = s.idx[hi];
                bool b_first = USE_ID ? rec_before(hb, ib, xb, ha, ia, xa) : rec_before(hb, 0, xb, ha, 0, xa);
                if (b_first == up) {
                    s.hi[lo] = hb; s.hi[hi] = ha;
                    s.id[lo] = ib; s.id[hi] = ia;
                    s.idx[lo] = xb; s.idx[hi] = xa;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v) {
    uint32_t p = 2;
    while (p < v) p <<= 1;
    return p;
}

__device__ void write_sorted(const SortLds& s, uint32_t count, uint32_t k, bool descending,
                             uint32_t* out_idx, uint64_t* out_ids, float* out_val, uint32_t* out_n) {
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        if (i < count) {
            if (out_idx) out_idx[i] = s.idx[i];
            if (out_ids) out_ids[i] = s.id[i];
            out_val[i] = ordered_to_f32(descending ? s.hi[i] : ~s.hi[i]);
        } else {
            if (out_idx) out_idx[i] = 0xffffffffu;
            if (out_ids) out_ids[i] = ~0ull;
            out_val[i] = descending ? -__builtin_huge_valf() : __builtin_huge_valf();
        }
    }
    if (threadIdx.x == 0 && out_n) *out_n = count;
}

#if ORAMA_SELECT_UNIT == 1
// Final ordering of the k' collected keys.
__global__ __launch_bounds__(kSortThreads) void select_sort_kernel(
    const SelectState* __restrict__ state, const unsigned long long* __restrict__ keys,
    uint32_t kcap, uint32_t k, bool descending, const uint64_t* __restrict__ id_map,
    uint32_t* out_idx, uint64_t* out_ids, float* out_val, uint32_t* out_n) {
    __shared__ SortLds s;
    const uint32_t qi = blockIdx.x;
    const uint32_t count = min(state[qi].kprime, kcap);
    const uint32_t p2 = next_pow2(count);
    const unsigned long long* in = keys + (uint64_t)qi * kcap;
    for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) {
        if (i < count) {
            unsigned long long key = in[i];
            uint32_t ix = ~(uint32_t)key;
            s.hi[i] = (uint32_t)(key >> 32);
            s.idx[i] = ix;
            s.id[i] = id_map ? id_map[ix] : (uint64_t)ix;
        } else {
            s.hi[i] = 0;
            s.idx[i] = 0xffffffffu;
            s.id[i] = ~0ull;
        }
    }
    __syncthreads();
    lds_bitonic_sort(s, p2);
    write_sorted(s, count, k, descending, out_idx ? out_idx + (uint64_t)qi * k : nullptr,
                 out_ids ? out_ids + (uint64_t)qi * k : nullptr, out_val + (uint64_t)qi * k,
                 out_n ? out_n + qi : nullptr);
}

// Whole selection in one workgroup when the list fits LDS (n <= kSelectMaxK).
__global__ __launch_bounds__(kSortThreads) void select_small_kernel(
    const float* __restrict__ vals, const uint32_t* __restrict__ idx, uint64_t stride,
    const uint32_t* __restrict__ n_dev, uint32_t n_max, uint32_t k, bool descending,
    const uint64_t* __restrict__ id_map, uint32_t* out_idx, uint64_t* out_ids, float* out_val,
    uint32_t* out_n) {
    __shared__ SortLds s;
    __shared__ uint32_t valid_s;
    const uint32_t qi = blockIdx.x;
    const uint32_t n = n_dev ? min(n_dev[qi], n_max) : n_max;
    const float* v = vals + (uint64_t)qi * stride;
    const uint32_t* ix = idx ? idx + (uint64_t)qi * stride : nullptr;
    const uint32_t p2 = next_pow2(n);
    if (threadIdx.x == 0) valid_s = 0;
    __syncthreads();
    uint32_t my_valid = 0;
    for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) {
        float x = i < n ? v[i] : __builtin_nanf("");
        if (x == x) {
            uint32_t id = ix ? ix[i] : i;
            uint32_t o = f32_to_ordered(x);
            s.hi[i] = descending ? o : ~o;
            s.idx[i] = id;
            s.id[i] = id_map ? id_map[id] : (uint64_t)id;
            ++my_valid;
        } else {
            s.hi[i] = 0;
            s.idx[i] = 0xffffffffu;
            s.id[i] = ~0ull;
        }
    }
    if (my_valid) atomicAdd(&valid_s, my_valid);
    __syncthreads();
    const uint32_t count = min(valid_s, k);
    if (id_map && valid_s > k) {
        // more candidates than answers and ids of their own: the k-th boundary is cut by (value, INDEX) — as the radix
        // selection of the longer lists, the reduction levels of the key lists and the CPU restatement of the path cut it — and only the survivors
        // are ordered by (value, id, index).  (One sort by (value, id, index) let the lowest ids win the cut: with ids that do
        // not grow with the index, which tied entries came back depended on the list's length — ADVICE r03.)
        lds_bitonic_sort<false>(s, p2);
        const uint32_t pk = next_pow2(max(count, 1u));
        for (uint32_t i = count + threadIdx.x; i < pk; i += blockDim.x) {
            s.hi[i] = 0;
            s.idx[i] = 0xffffffffu;
            s.id[i] = ~0ull;
        }
        __syncthreads();
        lds_bitonic_sort(s, pk);
    } else {
        lds_bitonic_sort(s, p2);
    }
    write_sorted(s, count, k, descending, out_idx ? out_idx + (uint64_t)qi * k : nullptr,
                 out_ids ? out_ids + (uint64_t)qi * k : nullptr, out_val + (uint64_t)qi * k,
                 out_n ? out_n + qi : nullptr);
}

// K6: merge `lists` x k candidates per query (ids are already 64-bit DocumentIds).
// List l of the ids lives at ids + l*ids_stride bytes, of the distances at dist + l*dist_stride bytes
// (separate arrays: stride = q*k elements; packed all-gather blocks: stride = block size).
__global__ __launch_bounds__(kSortThreads) void merge_candidates_kernel(
    const char* __restrict__ ids, uint64_t ids_stride, const char* __restrict__ dist,
    uint64_t dist_stride, uint32_t lists, uint32_t q, uint32_t k, bool descending, uint64_t* out_ids,
    float* out_dist, uint32_t* out_n) {
    __shared__ SortLds s;
    __shared__ uint32_t valid_s;
    const uint32_t qi = blockIdx.x;
    const uint32_t n = lists * k;
    const uint32_t p2 = next_pow2(n);
    if (threadIdx.x == 0) valid_s = 0;
    __syncthreads();
    uint32_t my_valid = 0;
    for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) {
        bool ok = false;
        if (i < n) {
            uint32_t l = i / k, j = i - l * k;
            const uint64_t src = (uint64_t)qi * k + j;
            const uint64_t id = reinterpret_cast<const uint64_t*>(ids + (uint64_t)l * ids_stride)[src];
            const float x = reinterpret_cast<const float*>(dist + (uint64_t)l * dist_stride)[src];
            if (id != ~0ull && x == x) {
                s.hi[i] = descending ? f32_to_ordered(x) : ~f32_to_ordered(x);
                s.idx[i] = i;
                s.id[i] = id;
                ok = true;
                ++my_valid;
            }
        }
        if (!ok) {
            s.hi[i] = 0;
            s.idx[i] = 0xffffffffu;
            s.id[i] = ~0ull;
        }
    }
    if (my_valid) atomicAdd(&valid_s, my_valid);
    __syncthreads();
    lds_bitonic_sort(s, p2);
    const uint32_t count = min(valid_s, k);
    write_sorted(s, count, k, descending, nullptr, out_ids + (uint64_t)qi * k, out_dist + (uint64_t)qi * k,
                 out_n ? out_n + qi : nullptr);
}

#endif
// The per-wave extremes of a workgroup's 16 waves -> the workgroup's, in every lane: lanes 0..15 read one entry each and the
// wave reduces by shuffles (a loop over the 16 entries had the compiler fetch all of them at once: 64 registers that cost
// pairs_reduce_kernel its second resident workgroup per CU).
__device__ __forceinline__ void workgroup_extremes(const unsigned long long* red_max, const unsigned long long* red_min,
                                                   unsigned long long* mx, unsigned long long* mn) {
    static_assert(kSortThreads / 64 == 16, "one entry per lane of the first 16");
    const int lane = threadIdx.x & 63;
    unsigned long long a = lane < 16 ? red_max[lane] : 0ull, b = lane < 16 ? red_min[lane] : ~0ull;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const unsigned long long x = __shfl_xor(a, off, 64), y = __shfl_xor(b, off, 64);
        a = x > a ? x : a;
        b = y < b ? y : b;
    }
    *mx = uniform_u64(a);  // (lane 0 holds the result; scalar registers from here on)
    *mn = uniform_u64(b);
}

// ---------------------------------------------------------------- exact cuts without histogram rounds
// The radix selection below costs up to seven barrier-separated rounds of LDS atomics that pile onto a few bins (the
// scores of one query share their leading bits): 12-15 us for the 8 192 keys of a chunk that starts without a bound — all
// chunks of a lone query (profiles/r04_k3r_single_query_timeline.log) and the first wave of workgroups of a batch.  With
// 16 waves on a CU every VALU instruction of a workgroup costs 16 issue cycles, so the replacements count instructions:
//   wave_kth_largest_hi   a LOWER BOUND of the workgroup's k-th best key from registers alone: every lane gives the best of
//                         its keys, every wave the j-th best of its 64 lane bests (j = ceil(k / waves)) by a 32-step binary
//                         search on the value word — one compare per step, the counting is scalar (ballot + s_bcnt1) —
//                         and the smallest of the waves' answers has waves * j >= k keys at or above it.  About 2k of
//                         8 192 keys survive it.  (First form: every lane counted the lane bests above its own, 64 x
//                         (2 v_readlane + v_cmp_u64 + add): 2.45 us of a 6.4 us workgroup, scripts/micro/keys_reduce_probe.)
//   compact_to_lds        one LDS atomic per WAVE for all eight keys of its lanes (eight dependent atomics before: 1.6 us).
//   rank_by_counting      the exact rank of each of a few hundred keys: T lanes per key count the keys above it (broadcast
//                         LDS reads, no atomics, no barrier).  Keys are unique, so ranks are a permutation: rank r < k IS
//                         the output position, in key order — the final kernel's records need no sort.  Work grows with
//                         the square of the count: up to 320 keys (k = 100 leaves ~200).
//                         (One wave's binary search for the exact k-th key — value word, then index word among its ties —
//                         was tried in its place: 3.0 us against 1.8 at 200 keys, and its registers cost the reduction
//                         kernels their second resident workgroup per CU; profiles/r04_keys_topk_phases.log.)
constexpr uint32_t kWaveBoundMaxK = 512;   // j <= 32 of 64 lane bests per wave (round 5: 256 -> 512 for the two-stage plan's k + 256 candidates)
constexpr uint32_t kRankCountMax = 512;
constexpr uint32_t kRankCountSmall = 320;  // the reduction kernels count ranks up to here (k = 100 leaves ~200), the final kernel up to 512

// the j-th largest of the wave's 64 values (0 when fewer than j of them are non-zero)
__device__ __forceinline__ uint32_t wave_kth_largest_hi(uint32_t v, uint32_t j) {
    uint32_t p = 0;
#pragma unroll
    for (int b = 31; b >= 0; --b) {
        const uint32_t t = p | (1u << b);
        p = (uint32_t)__popcll(__ballot(v >= t)) >= j ? t : p;  // (uniform: the compare is the only vector instruction)
    }
    return p;
}

// Workgroup-wide lower bound of the k-th best of the keys the lanes hold (`lane_best` = the best key of this lane), or 1
// ("every non-empty key") when a wave cannot vouch for its share.  `red` holds one entry per wave; the caller's next
// barrier-separated use of it must come after a barrier of its own.  Contains one barrier.
// `n_keys` = how many of the workgroup's threads hold at least one key, or any larger number (keys laid out key i -> thread
// i % 1024: the key count will do): with fewer than 1 024 only the first n_keys / 64 waves are full — they alone vouch, each
// for a larger share (a list of 800 keys left the last waves empty and the bound at 0: all 800 went through the histogram
// rounds).
__device__ __forceinline__ unsigned long long workgroup_kth_lower_bound(unsigned long long lane_best, uint32_t k,
                                                                        unsigned long long* red, uint32_t n_keys = kKeysChunk) {
    constexpr uint32_t kWaves = kSortThreads / 64;
    const uint32_t vouching = min(kWaves, n_keys / 64u);                   // (uniform)
    const uint32_t j = vouching ? (k + vouching - 1u) / vouching : 65u;    // each vouching wave's share
    const uint32_t wave = uniform_u32(threadIdx.x >> 6);  // (scalar: the LDS address below is not kept in a vector register)
    uint32_t wb = ~0u;                                    // a wave that does not vouch does not lower the bound either
    if (wave < vouching) wb = j <= 64u ? wave_kth_largest_hi((uint32_t)(lane_best >> 32), j) : 0u;
    if (!vouching) wb = 0u;
    if ((threadIdx.x & 63) == 0) red[wave] = wb;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t b = lane < (int)kWaves ? (uint32_t)red[lane] : ~0u;
#pragma unroll
    for (int off = (int)kWaves / 2; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(b, off, 64);
        b = o < b ? o : b;
    }
    b = uniform_u32(b);
    return b ? (unsigned long long)b << 32 : 1ull;  // every key whose value word reaches the bound: at least j per wave
}

// The lanes' keys at or above `floor_key` (>= 1: empties never) go to s[cursor ...]: one LDS atomic per wave.
template <int N>
__device__ __forceinline__ void compact_to_lds(const unsigned long long (&kr)[N], unsigned long long floor_key,
                                               unsigned long long* s, uint32_t* cursor) {
    const int lane = threadIdx.x & 63;
    unsigned long long m[N];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        m[j] = __ballot(kr[j] >= floor_key);
        tot += (uint32_t)__popcll(m[j]);
    }
    uint32_t base = 0;
    if (lane == 0 && tot) base = atomicAdd(cursor, tot);
    base = __shfl(base, 0, 64);
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (kr[j] >= floor_key) s[base + (uint32_t)__popcll(m[j] & below)] = kr[j];
        base += (uint32_t)__popcll(m[j]);
    }
}

// s[0 .. cnt): unique non-empty keys, cnt <= kRankCountMax.  Returns the number of keys above the key this thread answers
// for, s[threadIdx.x / T] (T = lanes per key: 8 up to 128 keys, 4 up to 256, 2 above); `key` receives that key, 0 when the
// thread has none.
__device__ __forceinline__ uint32_t rank_by_counting(const unsigned long long* s, uint32_t cnt, unsigned long long* key) {
    const uint32_t T = cnt <= kSortThreads / 8 ? 8u : cnt <= kSortThreads / 4 ? 4u : 2u;
    const uint32_t i = threadIdx.x / T, part = threadIdx.x & (T - 1u);
    const unsigned long long mine = i < cnt ? s[i] : ~0ull;
    uint32_t above = 0;
#pragma unroll 4
    for (uint32_t j = part; j < cnt; j += T) above += s[j] > mine ? 1u : 0u;
    above += __shfl_xor(above, 1, 64);
    if (T >= 4u) above += __shfl_xor(above, 2, 64);
    if (T >= 8u) above += __shfl_xor(above, 4, 64);
    *key = (i < cnt && part == 0u) ? mine : 0ull;
    return above;
}

#if ORAMA_SELECT_UNIT == 1
// ---------------------------------------------------------------- key lists (fused per-wave top-k of K1, K3r)
// One workgroup per (chunk, list): the best k of up to 8192 u64 keys, as a SET (the final kernel orders the survivors).
// Keys are unique (the low word is ~index), 0 = empty.  The chunk passes through registers; what reaches the floor (the
// list's running bound, or the bound from the lanes' best keys) goes to LDS; up to 320 survivors are cut by counting ranks
// (see above).  Beyond that — large k, lists of equal values — an MSB-first radix select in LDS: the bits above the first
// one in which the survivors' largest and smallest key differ are skipped (BM25 scores of one query share sign, exponent
// and often leading mantissa bits), then 8-bit digits: per-digit histogram with LDS atomics, one wave finds the bin that
// holds the k-th key, everything above it is taken; stops as soon as a bin is taken whole.
// (Walking several chunks per workgroup, carrying the best k from round to round, was measured and dropped: 172 -> 168 K
// BM25 queries/s — profiles/r04_keys_topk_phases.log (d).)
__global__ __launch_bounds__(kSortThreads, 8) void keys_reduce_kernel(const unsigned long long* __restrict__ keys,
                                                                   uint32_t n_keys, uint64_t in_stride,
                                                                   const uint32_t* __restrict__ n_per_list,
                                                                   uint32_t k, unsigned long long* __restrict__ out,
                                                                   uint64_t out_stride, unsigned long long* tau = nullptr,
                                                                   uint32_t tau_stride = 0, const uint32_t* __restrict__ n_active = nullptr,
                                                                   uint32_t direct_cap = 0, uint32_t n_stride = 1,
                                                                   uint32_t chunk_step = 0, uint32_t derive_k = 0) {
    __shared__ unsigned long long s[kKeysChunk];
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long red_max[kSortThreads / 64], red_min[kSortThreads / 64];
    __shared__ uint32_t sel_bin, sel_above, sel_cnt, cursor;
    // with a bound the grid is (lists, chunks): workgroups are dispatched list-fastest, so the first wave of workgroups
    // holds the first chunks of EVERY list and the later chunks of each list find a bound (chunk-fastest, a list's chunks
    // would all start together and none would)
    const uint32_t qi = tau ? blockIdx.x : blockIdx.y, chunk0 = tau ? blockIdx.y : blockIdx.x;
    if (n_active && qi >= *n_active) return;  // (uniform) only the first *n_active lists exist: nothing of the others is read or written
    if (n_per_list) {
        const uint32_t len = n_per_list[(uint64_t)qi * n_stride];
        // (derive_k != 0: this is the SECOND level over a counted list — its input holds derive_k survivors of every chunk of
        // the caller's list that exists; a list whose first level the final kernel can take whole is not reduced again)
        n_keys = min(n_keys, derive_k ? ((len + kKeysChunk - 1u) / kKeysChunk) * derive_k : len);  // lists shorter than the stride: the tail is not read
        if (derive_k && len <= direct_cap) return;
    }
    // counted lists: a chunk past the end writes nothing, and (first level) a list the final kernel takes whole is not reduced
    const bool sparse = direct_cap != 0;
    const uint32_t whole_cap = derive_k ? 0u : direct_cap;
    const int lane = threadIdx.x & 63, wave = (int)uniform_u32(threadIdx.x >> 6);
    // chunk_step != 0 (counted lists: the grid holds a few workgroups per list, not one per chunk of the worst case): the
    // workgroup walks chunks chunk0, chunk0 + chunk_step, ... of its list — each one the same procedure, under the list's
    // running bound as the earlier ones left it
    auto process = [&](const uint32_t chunk) {
    // A list's chunks share a running bound (tau, zero at launch): every workgroup that had to select publishes the k-th best
    // key of its chunk — a lower bound of the list's k-th best — and a chunk treats what lies below the bound it finds as
    // empty: about k of its 8 192 keys are left, the histogram rounds (LDS atomics that pile onto a few bins, since the
    // scores of one query share their leading bits) run over those, or not at all when at most k are left.  Workgroups of a
    // list start in waves; all but the first wave find a bound.
    const unsigned long long tau0 = tau ? __hip_atomic_load(tau + (uint64_t)qi * tau_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    const unsigned long long* in = keys + (uint64_t)qi * in_stride;
    const uint32_t begin = chunk * kKeysChunk;
    unsigned long long* o = out + (uint64_t)qi * out_stride + (uint64_t)chunk * k;
    // counted lists (direct_cap != 0: the length was produced on the device — K3r's compact key lists): a list the final kernel
    // can take whole is not reduced at all, and a chunk past the end writes nothing — the final kernel reads
    // ceil(length / chunk) * k survivors, not the grid's worth
    if (sparse && (n_keys <= whole_cap || begin >= n_keys)) return;
    if (begin >= n_keys) {
        for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) o[i] = 0ull;
        return;
    }
    const uint32_t n_in = min(kKeysChunk, n_keys - begin);
    ORAMA_KEYS_STAMP(0);
    // Stream the chunk through REGISTERS: every lane issues all of its loads first (16 bytes each where the chunk is
    // 16-byte aligned: the whole 64 KB chunk is in flight at once), drops what lies below the running bound, and only the
    // survivors — about k of 8 192 once a bound exists — are compacted into LDS (one LDS atomic per wave and round).  The
    // first form parked all 8 192 keys in LDS before it looked at them: one 8-byte load in flight per lane and 64 KB of LDS
    // stores per chunk, 1.2 TB/s over the key lists of a BM25 batch.
    static_assert(kKeysChunk == 8 * kSortThreads, "a lane carries 8 keys of the chunk");
    unsigned long long kr[8];
    const unsigned long long* src = in + begin;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = 2u * (threadIdx.x + (uint32_t)j * kSortThreads);
            ull2 v = {0ull, 0ull};
            if (i + 1 < n_in) v = *reinterpret_cast<const ull2*>(src + i);
            else if (i < n_in) v.x = src[i];
            kr[2 * j] = v.x;
            kr[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = threadIdx.x + (uint32_t)j * kSortThreads;
            kr[j] = i < n_in ? src[i] : 0ull;
        }
    }
    ORAMA_KEYS_STAMP(1);
    if (threadIdx.x == 0) {
        cursor = 0;
        sel_cnt = 0;
    }
    unsigned long long floor_key = tau0 > 1ull ? tau0 : 1ull;  // below the bound: cannot be among the list's best k; 0: empty
    if (tau0 <= 1ull && k <= kWaveBoundMaxK) {  // (workgroup-uniform) no bound yet: one from the lanes' best keys
        unsigned long long best = kr[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) best = kr[j] > best ? kr[j] : best;
        floor_key = workgroup_kth_lower_bound(best, k, red_min);  // (a short last chunk leaves waves empty: no bound, the histogram rounds)
    } else {
        __syncthreads();
    }
    ORAMA_KEYS_STAMP(2);
    compact_to_lds(kr, floor_key, s, &cursor);
    __syncthreads();
    const uint32_t cnt = cursor;  // non-empty keys at or above the bound, s[0 .. cnt)
    ORAMA_KEYS_STAMP(3);
    if (cnt > k && cnt <= kRankCountSmall) {  // (workgroup-uniform) a couple of hundred left: their exact ranks
        unsigned long long key;
        const uint32_t r = rank_by_counting(s, cnt, &key);
        if (key && r < k) {
            o[r] = key;  // cnt > k unique keys: every position below k is written
            if (tau && r == k - 1u) atomicMax(tau + (uint64_t)qi * tau_stride, key);
        }
        ORAMA_KEYS_STAMP(4);
        return;
    }
    __syncthreads();              // (everybody has read it: the final compaction counts with it again)
    if (threadIdx.x == 0) cursor = 0;
    uint32_t nz = cnt;
    unsigned long long mx = 0ull, mn = ~0ull;
    if (nz > k) {  // (workgroup-uniform) extremes of the survivors: the selection below skips their common leading bits
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const unsigned long long key = s[i];
            mx = key > mx ? key : mx;
            mn = key < mn ? key : mn;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long a = __shfl_xor(mx, off, 64), b = __shfl_xor(mn, off, 64);
            mx = a > mx ? a : mx;
            mn = b < mn ? b : mn;
        }
        if (lane == 0) {
            red_max[wave] = mx;
            red_min[wave] = mn;
        }
        __syncthreads();
        workgroup_extremes(red_max, red_min, &mx, &mn);
    }
    __syncthreads();
    unsigned long long thr = 1ull;  // take every non-empty key
    if (nz > k) {
        // bits above `low` are common to all non-empty keys; the k-th largest is searched below them
        uint32_t low = 64u - (uint32_t)__builtin_clzll(mx ^ mn);  // mx != mn: nz > k >= 1 unique keys
        unsigned long long prefix = low >= 64u ? 0ull : (mx >> low) << low;
        uint32_t need = k;
        for (;;) {
            const uint32_t d = low < 8u ? low : 8u;
            const uint32_t shift = low - d;
            const unsigned long long above_mask = low >= 64u ? 0ull : ~0ull << low;
            for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
                const unsigned long long key = s[i];
                if (key && (key & above_mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & ((1u << d) - 1u)], 1u);
            }
            __syncthreads();
            if (wave == 0) {
                uint32_t c[4], tot = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c[j] = hist[lane * 4 + j];
                    tot += c[j];
                }
                uint32_t incl = tot;  // keys in the bins of this lane and of the lanes above it
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t y = __shfl_down(incl, off, 64);
                    if (lane + off < 64) incl += y;
                }
                uint32_t above = incl - tot;
                if (above < need && need <= incl) {  // exactly one lane: the k-th key lies in its bins
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        if (above + c[j] >= need) {
                            sel_bin = (uint32_t)lane * 4 + j;
                            sel_above = above;
                            sel_cnt = c[j];
                            break;
                        }
                        above += c[j];
                    }
                }
            }
            __syncthreads();
            prefix |= (unsigned long long)uniform_u32(sel_bin) << shift;
            need -= uniform_u32(sel_above);
            low = shift;
            const bool whole = uniform_u32(sel_cnt) == need;
            __syncthreads();  // sel_* are rewritten by the next round
            if (whole || low == 0) break;
        }
        thr = prefix;  // keys >= thr: exactly k of them (unique keys)
        if (tau && threadIdx.x == 0) atomicMax(tau + (uint64_t)qi * tau_stride, thr);
    }
    // compact the selected keys (order is irrelevant here): one LDS atomic per wave and pass
    for (uint32_t i0 = 0; i0 < cnt; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const unsigned long long key = i < cnt ? s[i] : 0ull;
        const bool take = key >= thr;  // thr >= 1: empties never
        const unsigned long long m = __ballot(take);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&cursor, (uint32_t)__popcll(m));
        base = __shfl(base, 0, 64);
        if (take) {
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < k) o[pos] = key;
        }
    }
    __syncthreads();
    const uint32_t taken = min(cursor, k);
    for (uint32_t i = taken + threadIdx.x; i < k; i += blockDim.x) o[i] = 0ull;
    ORAMA_KEYS_STAMP(5);
    };
    if (!chunk_step) {
        process(chunk0);
        return;
    }
    for (uint32_t c = chunk0; (uint64_t)c * kKeysChunk < n_keys; c += chunk_step) {
        process(c);
        __syncthreads();  // (the next chunk reuses the key buffer, the cursors and the selection words)
    }
}

#endif
// ---------------------------------------------------------------- (value, index) lists in two launches
// The candidate lists of the fp16 scans (a few thousand entries per query, length known on the device only) went
// through the general selection: memset + 6 x (histogram + scan) + collect + sort = 15 launches of 2-15 us with a
// 6-11 us gap before each (~0.2 ms), 5-8 times per wide batch; this form took 36 + 39 us in round 2.  This kernel is
// keys_reduce_kernel's selection applied to keys built on the fly from (value, index) pairs: list `qi` is split
// into gridDim.x contiguous ranges, a workgroup walks its range in rounds of up to 8192 keys, carrying its best k along,
// and writes its best k; keys_final_kernel then orders gridDim.x * k <= 4096 survivors.  Exact for any list length (a list
// of millions of entries just takes more rounds).  NaN values become empty keys.  Round 4: the keys of a round stay in
// registers, only what reaches the floor (the k-th best so far, or the bound of workgroup_kth_lower_bound) goes to LDS, and
// the cut is made by counting ranks — which is also what lets the dense heads of the fp16 scans (131 072 values per query:
// one round for each of 16 workgroups) take this form instead of six histogram passes.
__device__ unsigned long long lds_keys_threshold(unsigned long long* s, uint32_t cnt, uint32_t k, uint32_t* hist,
                                                 unsigned long long* red_max, unsigned long long* red_min,
                                                 uint32_t* red_nz, uint32_t* sel /* bin, above, cnt */) {
    const int lane = threadIdx.x & 63, wave = (int)uniform_u32(threadIdx.x >> 6);
    unsigned long long mx = 0ull, mn = ~0ull;
    uint32_t nz = 0;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const unsigned long long key = s[i];
        if (key) {
            ++nz;
            mx = key > mx ? key : mx;
            mn = key < mn ? key : mn;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(mx, off, 64), b = __shfl_xor(mn, off, 64);
        mx = a > mx ? a : mx;
        mn = b < mn ? b : mn;
        nz += __shfl_xor(nz, off, 64);
    }
    __syncthreads();  // the reduction arrays may still be read by a previous call
    if (lane == 0) {
        red_max[wave] = mx;
        red_min[wave] = mn;
        red_nz[wave] = nz;
    }
    __syncthreads();
    workgroup_extremes(red_max, red_min, &mx, &mn);
    nz = lane < 16 ? red_nz[lane] : 0u;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) nz += __shfl_xor(nz, off, 64);
    nz = uniform_u32(nz);
    if (nz <= k) return 1ull;  // take every non-empty key
    uint32_t low = 64u - (uint32_t)__builtin_clzll(mx ^ mn);  // mx != mn: nz > k >= 1 unique keys
    unsigned long long prefix = low >= 64u ? 0ull : (mx >> low) << low;
    uint32_t need = k;
    for (;;) {
        const uint32_t d = low < 8u ? low : 8u;
        const uint32_t shift = low - d;
        const unsigned long long above_mask = low >= 64u ? 0ull : ~0ull << low;
        for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const unsigned long long key = s[i];
            if (key && (key & above_mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & ((1u << d) - 1u)], 1u);
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t c[4], tot = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = hist[lane * 4 + j];
                tot += c[j];
            }
            uint32_t incl = tot;  // keys in the bins of this lane and of the lanes above it
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_down(incl, off, 64);
                if (lane + off < 64) incl += y;
            }
            uint32_t above = incl - tot;
            if (above < need && need <= incl) {  // exactly one lane: the k-th key lies in its bins
#pragma unroll
                for (int j = 3; j >= 0; --j) {
                    if (above + c[j] >= need) {
                        sel[0] = (uint32_t)lane * 4 + j;
                        sel[1] = above;
                        sel[2] = c[j];
                        break;
                    }
                    above += c[j];
                }
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)uniform_u32(sel[0]) << shift;
        need -= uniform_u32(sel[1]);
        low = shift;
        const bool whole = uniform_u32(sel[2]) == need;
        __syncthreads();  // sel is rewritten by the next round
        if (whole || low == 0) break;
    }
    return prefix;  // keys >= prefix: exactly k of them (unique keys)
}

// Where a list of ONE round — the workgroup that reduces it has seen all of it — gets its final order right away (ids,
// (value, id, index) order, padding, count: what keys_final_kernel does), and the word that tells keys_final_kernel so.
struct PairsFinal {
    const uint64_t* id_map = nullptr;
    uint32_t* out_idx = nullptr;
    uint64_t* out_ids = nullptr;
    float* out_val = nullptr;
    uint32_t* out_n = nullptr;
    uint32_t* done = nullptr;  // one word per list: 1 = the outputs are written (nullptr: never finish here)
};

// The final order of a list whose best `kept` (<= 1 024: one per thread) keys sit at s[0 .. kept): ids, (value, id, index) order,
// padding, count — what keys_final_kernel does, in the workgroup that reduced the list.
__device__ __forceinline__ void finish_whole_list(unsigned long long* s, uint32_t kept, bool in_key_order, uint32_t k, bool descending,
                                               uint32_t qi, const PairsFinal& fin) {
    // the final order here, without a second launch: the records alias the key buffer (both 64 KB)
    static_assert(sizeof(SortLds) <= kKeysChunk * 8, "the records alias the key buffer");
    const uint32_t i = threadIdx.x;
    const unsigned long long mine = i < kept ? s[i] : 0ull;
    __syncthreads();  // every key is in a register: the records may be written
    SortLds& rec = *reinterpret_cast<SortLds*>(s);
    const uint32_t p2 = next_pow2(max(kept, 1u));
    if (i < kept) {
        const uint32_t ri = ~(uint32_t)mine;
        rec.hi[i] = (uint32_t)(mine >> 32);
        rec.idx[i] = ri;
        rec.id[i] = fin.id_map ? fin.id_map[ri] : (uint64_t)ri;
    } else if (i < p2) {
        rec.hi[i] = 0;
        rec.idx[i] = 0xffffffffu;
        rec.id[i] = ~0ull;
    }
    bool unordered = !in_key_order;
    if (in_key_order) {  // final order unless two neighbours of equal value carry their ids the other way round
        __syncthreads();
        bool swapped = false;
        for (uint32_t i = threadIdx.x; i + 1 < kept; i += blockDim.x)
            swapped |= rec.hi[i] == rec.hi[i + 1] && rec.id[i] > rec.id[i + 1];
        unordered = __syncthreads_or(swapped) != 0;
    } else {
        __syncthreads();
    }
    if (unordered) lds_bitonic_sort(rec, p2);
    write_sorted(rec, kept, k, descending, fin.out_idx ? fin.out_idx + (uint64_t)qi * k : nullptr,
                 fin.out_ids ? fin.out_ids + (uint64_t)qi * k : nullptr, fin.out_val + (uint64_t)qi * k,
                 fin.out_n ? fin.out_n + qi : nullptr);
}

#if ORAMA_SELECT_UNIT == 2
// A list of <= 512 values (the reranked candidates of the two-stage plan: 356 for k = 100; the merged entries of a hybrid call)
// in ONE workgroup without a sort (round 5): select_small_kernel's two bitonic sorts of 512 and 128 records are 73 barrier-
// separated steps — 23-29 us behind a lone query (profiles/r05_lone_call_timelines.log).  The keys are unique, so the rank of a
// key — counted, rank_by_counting — IS its position: the best k land in key order, which is the final order unless two
// neighbours of equal value carry their DocumentIds the other way round (finish_whole_list sorts only then).  Same keys, same
// cut (value, then index), same final order (value, id, index) as every other form.
__global__ __launch_bounds__(kSortThreads) void select_tiny_kernel(const float* __restrict__ vals, const uint32_t* __restrict__ idx,
                                                                  uint64_t stride, const uint32_t* __restrict__ n_dev, uint32_t n_max,
                                                                  uint32_t k, bool descending, PairsFinal fin) {
    __shared__ unsigned long long s[kKeysChunk];  // (finish_whole_list lays its records over the key buffer)
    __shared__ uint32_t cursor;
    const uint32_t qi = blockIdx.x;
    const uint32_t n = uniform_u32(n_dev ? min(n_dev[qi], n_max) : n_max);  // <= kRankCountMax: the launcher's rule
    const float* v = vals + (uint64_t)qi * stride;
    const uint32_t* ix = idx ? idx + (uint64_t)qi * stride : nullptr;
    if (threadIdx.x == 0) cursor = 0;
    __syncthreads();
    unsigned long long key = 0ull;
    if (threadIdx.x < n) {
        const float x = v[threadIdx.x];
        if (x == x) key = make_key(x, ix ? ix[threadIdx.x] : threadIdx.x, descending);
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(key != 0ull);
    uint32_t base = 0;
    if (lane == 0 && m) base = atomicAdd(&cursor, (uint32_t)__popcll(m));
    base = __shfl(base, 0, 64);
    if (key) s[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
    __syncthreads();
    const uint32_t valid = uniform_u32(cursor);
    const uint32_t kept = min(valid, k);
    unsigned long long mine;
    const uint32_t r = rank_by_counting(s, valid, &mine);
    __syncthreads();  // every key is in a register
    if (mine && r < kept) s[r] = mine;
    __syncthreads();
    finish_whole_list(s, kept, true, k, descending, qi, fin);
}

#endif
#if ORAMA_SELECT_UNIT == 1
template <bool HAS_IDX>
__global__ __launch_bounds__(kSortThreads, 8) void pairs_reduce_kernel(const float* __restrict__ vals,
                                                                    const uint32_t* __restrict__ idx, uint64_t stride,
                                                                    const uint32_t* __restrict__ n_dev, uint32_t n_max,
                                                                    bool descending, uint32_t k,
                                                                    unsigned long long* __restrict__ out, PairsFinal fin) {
    __shared__ unsigned long long s[kKeysChunk];
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long red_max[kSortThreads / 64], red_min[kSortThreads / 64];
    __shared__ uint32_t red_nz[kSortThreads / 64];
    __shared__ uint32_t sel[3], cursor, cursor2;  // appended keys of a round / keys kept by its cut
    __shared__ unsigned long long kth_s;
    // grid (lists, parts), LIST-fastest: consecutive workgroups go to the 8 XCDs in turn, and of a short list only part 0 works —
    // part-fastest, with 8 parts, EVERY working workgroup landed on XCD 0 (256 of them on 32 CUs: 66 us instead of 12)
    const uint32_t qi = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const uint32_t n = uniform_u32(n_dev ? min(n_max, n_dev[qi]) : n_max);  // (scalar registers for everything derived from it)
    unsigned long long* o = out + ((uint64_t)qi * parts + part) * k;
    uint32_t pos, end;
    if (n <= kKeysChunk) {
        // a list of one round (the candidate lists of the fp16 scans hold a few hundred entries; the grid was sized for the
        // worst case): ONE workgroup takes all of it, the others only write their k empty outputs
        if (part != 0) {
            if (!(fin.done && k <= kSortThreads))  // (finished by workgroup 0: the final kernel will not look)
                for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) o[i] = 0ull;
            return;
        }
        pos = 0;
        end = n;
    } else {
        const uint64_t per = ((uint64_t)n + parts - 1) / parts;  // (64-bit: n may be close to 2^32)
        pos = (uint32_t)min((uint64_t)n, part * per);
        end = (uint32_t)min((uint64_t)n, (uint64_t)pos + per);
    }
    const float* v = vals + (uint64_t)qi * stride;
    const uint32_t* ix = HAS_IDX ? idx + (uint64_t)qi * stride : nullptr;
    constexpr uint32_t kPerThread = kKeysChunk / kSortThreads;
    // Rounds of up to 8 192 values.  The best k so far sit at s[0 .. kept); once a round had to cut, the k-th best key is known
    // (`kth`) and is the floor of every later round: a value below it never reaches LDS.  A round that starts without a floor
    // takes a lower bound from the lanes' best keys (workgroup_kth_lower_bound); what survives — a couple of hundred keys — is
    // cut by counting ranks, as in keys_reduce_kernel; the histogram rounds remain for the rest (k beyond ~150, lists of equal
    // values).  (One wave's search for 321..512 survivors is left out here: its registers cost this kernel its second
    // resident workgroup per CU.)
    uint32_t kept = 0;
    unsigned long long kth = 0ull;  // (uniform) 0: not known
    bool in_key_order = true;       // (uniform) s[0 .. kept) is in key order
    for (;;) {
        const uint32_t take = min(kKeysChunk - kept, end - pos);
        ORAMA_KEYS_STAMP(0);
        unsigned long long kr[kPerThread];
        unsigned long long best = 0ull;
#pragma unroll
        for (uint32_t t = 0; t < kPerThread; ++t) {
            const uint32_t i = t * blockDim.x + threadIdx.x;
            unsigned long long key = 0ull;
            if (i < take) {
                const float x = v[pos + i];
                if (x == x) key = make_key(x, HAS_IDX ? ix[pos + i] : pos + i, descending);
            }
            kr[t] = key;
            best = key > best ? key : best;
        }
        pos += take;
        ORAMA_KEYS_STAMP(1);
        if (threadIdx.x == 0) {
            cursor = kept;
            cursor2 = 0;
        }
        unsigned long long floor_key = kth > 1ull ? kth : 1ull;
        if (kth == 0ull && k <= kWaveBoundMaxK && kept + take > kRankCountSmall) floor_key = workgroup_kth_lower_bound(best, k, red_min, take);
        else __syncthreads();
        ORAMA_KEYS_STAMP(2);
        compact_to_lds(kr, floor_key, s, &cursor);  // behind the kept keys
        __syncthreads();
        const uint32_t cnt = uniform_u32(cursor);
        ORAMA_KEYS_STAMP(3);
        if (cnt <= k) {
            in_key_order = in_key_order && cnt == kept;
            kept = cnt;  // nothing to cut (the k-th best stays unknown until a round cuts)
            __syncthreads();  // (the cursor is rewritten by the next round)
        } else if (cnt <= kRankCountSmall) {
            unsigned long long key;
            const uint32_t r = rank_by_counting(s, cnt, &key);
            __syncthreads();  // every key is in a register
            if (key && r < k) {
                s[r] = key;  // in key order
                if (r == k - 1u) kth_s = key;
            }
            __syncthreads();
            kept = k;
            kth = uniform_u64(kth_s);
            in_key_order = true;
        } else {
            const unsigned long long thr = lds_keys_threshold(s, cnt, k, hist, red_max, red_min, red_nz, sel);  // (barriers inside)
            // keep the keys at or above it, in place: the first 1 024 are read before anything is written (what is kept lands
            // below k <= 4 096... at most `cnt` positions, all of them already read by then); exactly k keys reach thr
            const int lane = threadIdx.x & 63;
            unsigned long long key = threadIdx.x < cnt ? s[threadIdx.x] : 0ull;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < cnt; i0 += blockDim.x) {
                if (i0) {
                    __syncthreads();  // (writes of the previous pass land below the positions read now only if k <= i0: make it so)
                    key = i0 + threadIdx.x < cnt ? s[i0 + threadIdx.x] : 0ull;
                    __syncthreads();
                }
                const bool tk = key >= thr;  // thr >= 1: empties never
                const unsigned long long m = __ballot(tk);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(&cursor2, (uint32_t)__popcll(m));
                base = __shfl(base, 0, 64);
                if (tk) s[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key;
            }
            __syncthreads();
            kept = min(uniform_u32(cursor2), k);
            kth = uniform_u64(thr);
            in_key_order = false;
            __syncthreads();
        }
        if (pos >= end) break;
    }
    ORAMA_KEYS_STAMP(4);
    if (fin.done && part == 0) {
        // (uniform) this workgroup has seen the whole list, and its best k are one key per thread
        const bool whole = n <= kKeysChunk && k <= kSortThreads;
        if (threadIdx.x == 0) fin.done[qi] = whole ? 1u : 0u;
        if (whole) {
            finish_whole_list(s, kept, in_key_order, k, descending, qi, fin);
            ORAMA_KEYS_STAMP(5);
            if (threadIdx.x == 0) ORAMA_KEYS_NOTE(6, ((unsigned long long)in_key_order << 32) | kept);
            return;
        }
    }
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) o[i] = i < kept ? s[i] : 0ull;
}

#endif
#if ORAMA_SELECT_UNIT == 2
// ---- a FEW LONG dense lists (a lone query's distance array: 1 M values), round 5 -------------------------------------------------
// pairs_reduce_kernel walks such a list in rounds of 8 192 values: 40 parts of 1 M values take 3-4 rounds each, every round a
// chain of load -> bound -> compact -> cut: 39 us behind a 230 us scan of 1 M x 384 rows (profiles/r05_c2_kernel_stats.md).
// Here a part's <= 32 768 values are loaded ONCE, 32 per thread, all loads in flight together; the bound from the lanes' best
// keys (workgroup_kth_lower_bound) leaves a few hundred of them for LDS and ONE cut finishes the part.  The bound is a
// heuristic for the survivor count, not a guarantee (one wave of small values lowers it for everybody): when more than
// 8 192 keys reach it, nothing has been lost — the values are still in registers — and the part is redone in rounds of 7 per
// thread (7 168 + k <= 8 192) exactly as pairs_reduce_kernel would.  One workgroup per CU (the registers), which 40 parts do
// not mind.
constexpr int kWidePer = 32;

struct WideState {
    uint32_t kept;
    unsigned long long kth;  // 0: not known
};

// One round over x[T0 .. T0 + TN) (value t of thread i is element t * 1024 + i of the part).  Returns false — and leaves
// s[0 .. st.kept) and `st` as they were — when more keys reached the floor than LDS holds behind the kept ones.
template <int T0, int TN>
__device__ __forceinline__ bool wide_round(const float (&x)[kWidePer], uint32_t base_index, uint32_t cnt_in, bool descending,
                                           uint32_t k, unsigned long long* s, uint32_t* hist, unsigned long long* red_max,
                                           unsigned long long* red_min, uint32_t* red_nz, uint32_t* sel, uint32_t* cursor,
                                           uint32_t* cursor2, unsigned long long* kth_s, WideState& st) {
    if ((uint32_t)T0 * kSortThreads >= cnt_in) return true;  // (uniform) nothing of the part lies here
    const uint32_t here = min((uint32_t)TN * kSortThreads, cnt_in - (uint32_t)T0 * kSortThreads);
    const int lane = threadIdx.x & 63;
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float v = x[T0 + j];
        const unsigned long long key = v == v ? make_key(v, base_index + (uint32_t)(T0 + j) * kSortThreads + threadIdx.x, descending) : 0ull;
        best = key > best ? key : best;
    }
    if (threadIdx.x == 0) {
        *cursor = st.kept;
        *cursor2 = 0;
    }
    unsigned long long floor_key = st.kth > 1ull ? st.kth : 1ull;
    {
        const unsigned long long lb = workgroup_kth_lower_bound(best, k, red_min, here);  // (one barrier inside)
        floor_key = lb > floor_key ? lb : floor_key;
    }
    // the keys at or above the floor, behind the kept ones; one LDS atomic per wave and 8 values; nothing written past the end
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j0 = 0; j0 < TN; j0 += 8) {
        unsigned long long kr[8], m[8];
        uint32_t tot = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kr[j] = 0ull;
            if (j0 + j < TN) {
                const float v = x[T0 + j0 + j];
                if (v == v) kr[j] = make_key(v, base_index + (uint32_t)(T0 + j0 + j) * kSortThreads + threadIdx.x, descending);
            }
            m[j] = __ballot(kr[j] >= floor_key);
            tot += (uint32_t)__popcll(m[j]);
        }
        uint32_t base = 0;
        if (lane == 0 && tot) base = atomicAdd(cursor, tot);
        base = __shfl(base, 0, 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t at = base + (uint32_t)__popcll(m[j] & below);
            if (kr[j] >= floor_key && at < kKeysChunk) s[at] = kr[j];
            base += (uint32_t)__popcll(m[j]);
        }
    }
    __syncthreads();
    const uint32_t cnt = uniform_u32(*cursor);
    if (cnt > kKeysChunk) {  // (uniform) more than LDS holds: the caller takes smaller rounds
        __syncthreads();     // (the cursor is rewritten by the next round)
        return false;
    }
    if (cnt <= k) {
        st.kept = cnt;
        __syncthreads();
    } else if (cnt <= kRankCountSmall) {
        unsigned long long key;
        const uint32_t r = rank_by_counting(s, cnt, &key);
        __syncthreads();  // every key is in a register
        if (key && r < k) {
            s[r] = key;
            if (r == k - 1u) *kth_s = key;
        }
        __syncthreads();
        st.kept = k;
        st.kth = uniform_u64(*kth_s);
    } else {
        const unsigned long long thr = lds_keys_threshold(s, cnt, k, hist, red_max, red_min, red_nz, sel);  // (barriers inside)
        unsigned long long key = threadIdx.x < cnt ? s[threadIdx.x] : 0ull;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < cnt; i0 += blockDim.x) {
            if (i0) {
                __syncthreads();
                key = i0 + threadIdx.x < cnt ? s[i0 + threadIdx.x] : 0ull;
                __syncthreads();
            }
            const bool tk = key >= thr;
            const unsigned long long m = __ballot(tk);
            uint32_t base = 0;
            if (lane == 0 && m) base = atomicAdd(cursor2, (uint32_t)__popcll(m));
            base = __shfl(base, 0, 64);
            if (tk) s[base + (uint32_t)__popcll(m & below)] = key;
        }
        __syncthreads();
        st.kept = min(uniform_u32(*cursor2), k);
        st.kth = uniform_u64(thr);
        __syncthreads();
    }
    return true;
}

__global__ __launch_bounds__(kSortThreads) void pairs_reduce_wide_kernel(const float* __restrict__ vals, uint64_t stride,
                                                                        const uint32_t* __restrict__ n_dev, uint32_t n_max,
                                                                        bool descending, uint32_t k,
                                                                        unsigned long long* __restrict__ out, bool force_narrow) {
    __shared__ unsigned long long s[kKeysChunk];
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long red_max[kSortThreads / 64], red_min[kSortThreads / 64];
    __shared__ uint32_t red_nz[kSortThreads / 64];
    __shared__ uint32_t sel[3], cursor, cursor2;
    __shared__ unsigned long long kth_s;
    const uint32_t qi = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
    const uint32_t n = uniform_u32(n_dev ? min(n_max, n_dev[qi]) : n_max);
    const uint64_t per = ((uint64_t)n + parts - 1) / parts;  // <= kWidePer * 1024: the launcher's rule
    const uint32_t pos = (uint32_t)min((uint64_t)n, part * per);
    const uint32_t end = (uint32_t)min((uint64_t)n, (uint64_t)pos + per);
    const uint32_t cnt_in = end - pos;
    const float* v = vals + (uint64_t)qi * stride + pos;
    float x[kWidePer];
#pragma unroll
    for (int t = 0; t < kWidePer; ++t) {
        const uint32_t i = (uint32_t)t * kSortThreads + threadIdx.x;
        x[t] = i < cnt_in ? v[i] : __builtin_nanf("");
    }
    WideState st{0u, 0ull};
#define ORAMA_WIDE_ROUND(T0, TN) \
    wide_round<T0, TN>(x, pos, cnt_in, descending, k, s, hist, red_max, red_min, red_nz, sel, &cursor, &cursor2, &kth_s, st)
    if (force_narrow || !ORAMA_WIDE_ROUND(0, kWidePer)) {
        // (uniform) rounds that always fit: 7 x 1 024 values + the kept keys <= 8 192 — the wide path is gated on
        // k <= kWaveBoundMaxK (launch_select), so the invariant is the constants': asserted, and a round that does not fit
        // (it cannot) stops the kernel loudly instead of dropping keys
        static_assert(7 * kSortThreads + kWaveBoundMaxK <= kKeysChunk, "the narrow fallback rounds of pairs_reduce_wide_kernel must fit the key chunk");
        bool fits = ORAMA_WIDE_ROUND(0, 7);
        fits = ORAMA_WIDE_ROUND(7, 7) && fits;
        fits = ORAMA_WIDE_ROUND(14, 7) && fits;
        fits = ORAMA_WIDE_ROUND(21, 7) && fits;
        fits = ORAMA_WIDE_ROUND(28, 4) && fits;
        if (!fits) __builtin_trap();
    }
#undef ORAMA_WIDE_ROUND
    unsigned long long* o = out + ((uint64_t)qi * parts + part) * k;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) o[i] = i < st.kept ? s[i] : 0ull;
}

#endif
#if ORAMA_SELECT_UNIT == 1
// Final ordering of <= 4096 keys: (value, 64-bit id asc, idx asc), empties (0) dropped.
__global__ __launch_bounds__(kSortThreads) void keys_final_kernel(const unsigned long long* __restrict__ keys,
                                                                  uint32_t n_keys, uint64_t in_stride,
                                                                  const uint32_t* __restrict__ n_per_list, uint32_t k,
                                                                  bool descending,
                                                                  const uint64_t* __restrict__ id_map,
                                                                  uint32_t* out_idx, uint64_t* out_ids, float* out_val,
                                                                  uint32_t* out_n, const uint32_t* __restrict__ n_active = nullptr,
                                                                  const uint32_t* __restrict__ done = nullptr,
                                                                  const unsigned long long* __restrict__ direct_keys = nullptr,
                                                                  uint64_t direct_stride = 0, uint32_t direct_cap = 0,
                                                                  uint32_t n_stride = 1,
                                                                  const unsigned long long* __restrict__ l2_keys = nullptr,
                                                                  uint64_t l2_stride = 0, KeysMirror mirror = KeysMirror()) {
    __shared__ SortLds s;
    __shared__ uint32_t valid_s;
    __shared__ uint32_t hist[256];
    if (mirror.src)  // (the words are final: everything that writes them was launched before this kernel)
        for (uint32_t i = threadIdx.x; i < mirror.words; i += blockDim.x)
            mirror.dst[(size_t)blockIdx.x * mirror.words + i] = mirror.src[(size_t)blockIdx.x * mirror.words + i];
    if (n_active && blockIdx.x >= *n_active) return;  // (uniform) see keys_reduce_kernel
    if (done && done[blockIdx.x]) return;              // (uniform) pairs_reduce_kernel has finished this list itself
    __shared__ unsigned long long red_max[kSortThreads / 64], red_min[kSortThreads / 64];
    __shared__ uint32_t red_nz[kSortThreads / 64];
    __shared__ uint32_t sel[3], cursor;
    const uint32_t qi = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long* in = keys + (uint64_t)qi * in_stride;
    if (direct_cap) {
        // counted lists (see keys_reduce_kernel): n_per_list is the length of the CALLER's list; `keys` is the reduced level
        const uint32_t len = min(n_per_list[(uint64_t)qi * n_stride], (uint32_t)direct_stride);
        if (len <= direct_cap) {  // (uniform) nothing was reduced: the list itself
            in = direct_keys + (uint64_t)qi * direct_stride;
            n_keys = len;
        } else {
            const uint32_t n1 = ((len + kKeysChunk - 1u) / kKeysChunk) * k;  // what the first level left of the list
            if (n1 <= direct_cap || !l2_keys) {
                n_keys = min(n_keys, n1);
            } else {  // (uniform) more than this kernel takes: the second level's output
                in = l2_keys + (uint64_t)qi * l2_stride;
                n_keys = ((n1 + kKeysChunk - 1u) / kKeysChunk) * k;
            }
        }
    } else if (n_per_list) {
        n_keys = min(n_keys, n_per_list[(uint64_t)qi * n_stride]);
    }
    if (n_keys > k) {
        // More candidates than answers (e.g. 32 chunks x 100 survivors of a scan's wave lists: 3 200 keys for 100 results):
        // ordering all of them is a 4 096-element bitonic sort with a 64-bit id gather per element, 80 us of a 4.4 ms query.
        // The k best BY KEY are cut out first — the radix threshold of the reduction levels, so that the cut among equal
        // values keeps the lowest indices exactly as every level before this one does (DESIGN.md §3 rule 4) — and only
        // those are ordered by (value, id, index).  For EVERY list longer than k (round 3 sorted lists of up to
        // 2 x pow2(k) keys whole and let the lowest DocumentIds win the cut there: which tied rows came back depended on the
        // list's length wherever DocumentIds are not monotonic in the row index — ADVICE r03).
        // (the key buffer aliases the whole record array: 8 192 keys — two reduction chunks' worth, see keys_final_capacity)
        unsigned long long* kb = reinterpret_cast<unsigned long long*>(&s);
        constexpr uint32_t kPerThread = kKeysChunk / kSortThreads;
        static_assert(sizeof(SortLds) >= kKeysChunk * 8, "the key buffer must fit the record array");
        // The candidates pass through REGISTERS first (all loads of a lane in flight at once); a lower bound of the k-th best
        // from the lanes' best keys (workgroup_kth_lower_bound) leaves about 2k of them for LDS, whose exact ranks are counted
        // (rank_by_counting) — round 3 parked all of them in LDS and ran the histogram rounds over them, 17-22 us for the
        // 7 200 candidates of a lone BM25 query.  Lists the bound cannot thin (large k, mostly empty lanes) take that path still.
        unsigned long long mine[kPerThread];
        unsigned long long best = 0ull;
        ORAMA_KEYS_STAMP(0);
#pragma unroll
        for (uint32_t t = 0; t < kPerThread; ++t) {
            const uint32_t i = t * blockDim.x + threadIdx.x;
            mine[t] = i < n_keys ? in[i] : 0ull;
            best = mine[t] > best ? mine[t] : best;
        }
        ORAMA_KEYS_STAMP(1);
        if (threadIdx.x == 0) cursor = 0;
        unsigned long long floor_key = 1ull;
        if (k <= kWaveBoundMaxK) floor_key = workgroup_kth_lower_bound(best, k, red_min, n_keys);
        else __syncthreads();
        ORAMA_KEYS_STAMP(2);
        compact_to_lds(mine, floor_key, kb, &cursor);
        __syncthreads();
        const uint32_t cnt = cursor;  // candidates at or above the bound, kb[0 .. cnt): at least min(k, non-empty keys)
        uint32_t count;
        bool in_key_order = false;
        ORAMA_KEYS_STAMP(3);
        if (cnt > k && cnt <= kRankCountMax) {  // (workgroup-uniform)
            unsigned long long key;
            const uint32_t r = rank_by_counting(kb, cnt, &key);
            __syncthreads();  // every key is in a register: the records may be written
            if (key && r < k) {
                const uint32_t ix = ~(uint32_t)key;
                s.hi[r] = (uint32_t)(key >> 32);
                s.idx[r] = ix;
                s.id[r] = id_map ? id_map[ix] : (uint64_t)ix;
            }
            count = k;
            in_key_order = true;  // records sit in (value, index) order
        } else {
            __syncthreads();  // (everybody has read the cursor: the second compaction counts with it again)
            if (threadIdx.x == 0) cursor = 0;
            __syncthreads();
            const unsigned long long thr = lds_keys_threshold(kb, cnt, k, hist, red_max, red_min, red_nz, sel);
#pragma unroll
            for (uint32_t t = 0; t < kPerThread; ++t) {
                const uint32_t i = t * blockDim.x + threadIdx.x;
                mine[t] = i < cnt ? kb[i] : 0ull;
            }
            __syncthreads();  // every key is in a register: the records may be written
#pragma unroll
            for (uint32_t t = 0; t < kPerThread; ++t) {
                const bool tk = mine[t] >= thr;  // thr >= 1: empties never
                const unsigned long long m = __ballot(tk);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(&cursor, (uint32_t)__popcll(m));
                base = __shfl(base, 0, 64);
                if (tk) {
                    const uint32_t p = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (p < k) {
                        const uint32_t ix = ~(uint32_t)mine[t];
                        s.hi[p] = (uint32_t)(mine[t] >> 32);
                        s.idx[p] = ix;
                        s.id[p] = id_map ? id_map[ix] : (uint64_t)ix;
                    }
                }
            }
            __syncthreads();
            count = min(cursor, k);
        }
        ORAMA_KEYS_STAMP(4);
        const uint32_t p2 = next_pow2(max(count, 1u));
        for (uint32_t i = count + threadIdx.x; i < p2; i += blockDim.x) {
            s.hi[i] = 0;
            s.idx[i] = 0xffffffffu;
            s.id[i] = ~0ull;
        }
        // Records in (value, index) order are in FINAL order unless two neighbours of equal value carry their ids the other
        // way round (ids need not grow with the index): only then are they sorted — 28 barrier-separated passes for 128.
        bool unordered = !in_key_order;
        if (in_key_order) {
            __syncthreads();
            bool swapped = false;
            for (uint32_t i = threadIdx.x; i + 1 < count; i += blockDim.x)
                swapped |= s.hi[i] == s.hi[i + 1] && s.id[i] > s.id[i + 1];
            unordered = __syncthreads_or(swapped) != 0;
        } else {
            __syncthreads();
        }
        if (unordered) lds_bitonic_sort(s, p2);
        ORAMA_KEYS_STAMP(5);
        write_sorted(s, count, k, descending, out_idx ? out_idx + (uint64_t)qi * k : nullptr,
                     out_ids ? out_ids + (uint64_t)qi * k : nullptr, out_val + (uint64_t)qi * k, out_n ? out_n + qi : nullptr);
        ORAMA_KEYS_STAMP(6);
        return;
    }
    const uint32_t p2 = next_pow2(max(n_keys, 1u));
    if (threadIdx.x == 0) valid_s = 0;
    __syncthreads();
    uint32_t my_valid = 0;
    for (uint32_t i = threadIdx.x; i < p2; i += blockDim.x) {
        const unsigned long long key = i < n_keys ? in[i] : 0ull;
        if (key != 0ull) {
            const uint32_t ix = ~(uint32_t)key;
            s.hi[i] = (uint32_t)(key >> 32);
            s.idx[i] = ix;
            s.id[i] = id_map ? id_map[ix] : (uint64_t)ix;
            ++my_valid;
        } else {
            s.hi[i] = 0;
            s.idx[i] = 0xffffffffu;
            s.id[i] = ~0ull;
        }
    }
    if (my_valid) atomicAdd(&valid_s, my_valid);
    __syncthreads();
    lds_bitonic_sort(s, p2);
    const uint32_t count = min(valid_s, k);
    write_sorted(s, count, k, descending, out_idx ? out_idx + (uint64_t)qi * k : nullptr,
                 out_ids ? out_ids + (uint64_t)qi * k : nullptr, out_val + (uint64_t)qi * k,
                 out_n ? out_n + qi : nullptr);
}

#endif
}  // namespace

#if ORAMA_SELECT_UNIT == 1
// Keys the final kernel takes: a full sort handles 4 096 records; when the k best are cut out first (more than two sorts'
// worth of candidates, see keys_final_kernel) the candidates only pass through the key buffer, which holds 8 192.
static uint32_t keys_final_capacity(uint32_t k) {
    uint32_t p2 = 2;
    while (p2 < k) p2 <<= 1;
    return 2 * p2 < kSelectMaxK ? kKeysChunk : kSelectMaxK;
}

uint64_t keys_topk_scratch_keys(uint32_t n_keys, uint32_t q, uint32_t k) {
    uint64_t total = 0;
    uint32_t n = n_keys;
    while (n > keys_final_capacity(k)) {
        const uint32_t chunks = (n + kKeysChunk - 1) / kKeysChunk;
        total += (uint64_t)q * chunks * k;
        n = chunks * k;
    }
    return total;
}

int launch_keys_topk(orama_ctx* ctx, const unsigned long long* d_keys, uint32_t n_keys, uint64_t stride,
                     uint32_t q, uint32_t k, bool descending, const uint64_t* id_map,
                     unsigned long long* d_tmp, uint32_t* out_idx, uint64_t* out_ids, float* out_val,
                     uint32_t* out_n, hipStream_t stream, const uint32_t* d_n_per_list, unsigned long long* d_tau,
                     uint32_t tau_stride, const uint32_t* d_n_active, bool counted, uint32_t n_per_list_stride,
                     const KeysMirror* mirror) {
    ORAMA_REQUIRE(k >= 1 && k <= kSelectMaxK && q >= 1 && d_keys && out_val, "keys top-k: bad arguments");
    const KeysMirror mir = mirror ? *mirror : KeysMirror();
    ProfScope prof(&ctx->prof, "topk_select", stream);
    if (counted && d_n_per_list) {
        // Lists whose lengths were produced on the device and are expected to be MUCH shorter than n_keys (K3r's compact key
        // lists: a few percent of one slot per posting).  One reduction level sized for the worst case, of which only the
        // chunks that exist do anything, and none at all for a list the final kernel can take whole.
        const uint32_t cap = keys_final_capacity(k);
        const uint32_t chunks = (n_keys + kKeysChunk - 1) / kKeysChunk;
        const uint64_t n1 = (uint64_t)chunks * k;                              // first-level survivors of a list in the worst case
        const uint32_t chunks2 = (uint32_t)((n1 + kKeysChunk - 1) / kKeysChunk);
        if ((uint64_t)chunks2 * k <= cap) {
            unsigned long long* l2 = nullptr;
            if (n_keys > cap) {
                ORAMA_REQUIRE(d_tmp, "keys top-k: scratch missing");
                // (a few workgroups per list, each walking every walkers-th chunk that exists: the worst case is one slot per
                // posting — 72 to 200 chunks per list of which 3 to 8 exist, thousands of workgroups that end at once)
                const uint32_t walkers = std::min<uint32_t>(chunks, 8u);
                hipLaunchKernelGGL(keys_reduce_kernel, d_tau ? dim3(q, walkers) : dim3(walkers, q), dim3(kSortThreads), 0, stream, d_keys,
                                   n_keys, stride, d_n_per_list, k, d_tmp, n1, d_tau, tau_stride, d_n_active, cap,
                                   n_per_list_stride, walkers);
                if (n1 > cap) {
                    // the worst case leaves more first-level survivors than the final kernel takes (lists of more than
                    // cap / k chunks: 660 K keys at k = 100): a second level for the lists that really are that long — its
                    // workgroups end at once for every other list, and the final kernel reads the level that applies
                    l2 = d_tmp + (uint64_t)q * n1;
                    hipLaunchKernelGGL(keys_reduce_kernel, dim3(chunks2, q), dim3(kSortThreads), 0, stream, d_tmp, (uint32_t)n1, n1,
                                       d_n_per_list, k, l2, (uint64_t)chunks2 * k, (unsigned long long*)nullptr, 0u, d_n_active,
                                       (uint32_t)(((uint64_t)cap / k) * kKeysChunk), n_per_list_stride, 0u, k);
                }
            }
            hipLaunchKernelGGL(keys_final_kernel, dim3(q), dim3(kSortThreads), 0, stream, d_tmp ? d_tmp : d_keys, (uint32_t)n1, n1,
                               d_n_per_list, k, descending, id_map, out_idx, out_ids, out_val, out_n, d_n_active,
                               (const uint32_t*)nullptr, d_keys, stride, cap, n_per_list_stride, l2, (uint64_t)chunks2 * k, mir);
            ORAMA_HIP_TRY(hipGetLastError());
            return ORAMA_OK;
        }
        // (longer lists: the general levels below, reading the counted lengths at the first one)
    }

    const unsigned long long* cur = d_keys;
    uint64_t cur_stride = stride;
    uint32_t n = n_keys;
    unsigned long long* tmp = d_tmp;
    const uint32_t* n_per_list = d_n_per_list;  // applies to the caller's lists only; reduced levels are full
    uint32_t n_stride = n_per_list_stride;
    while (n > keys_final_capacity(k)) {
        ORAMA_REQUIRE(tmp, "keys top-k: scratch missing");
        const uint32_t chunks = (n + kKeysChunk - 1) / kKeysChunk;
        const uint64_t out_stride = (uint64_t)chunks * k;
        hipLaunchKernelGGL(keys_reduce_kernel, d_tau ? dim3(q, chunks) : dim3(chunks, q), dim3(kSortThreads), 0, stream, cur, n,
                           cur_stride, n_per_list, k, tmp, out_stride, d_tau, tau_stride, d_n_active, 0u, n_stride);
        d_tau = nullptr;  // (a bound belongs to the caller's lists: the next level starts without one)
        cur = tmp;
        cur_stride = out_stride;
        n = chunks * k;
        n_per_list = nullptr;
        n_stride = 1;
        tmp = tmp + (uint64_t)q * out_stride;  // next level (if any) writes behind this one
    }
    hipLaunchKernelGGL(keys_final_kernel, dim3(q), dim3(kSortThreads), 0, stream, cur, n, cur_stride, n_per_list, k,
                       descending, id_map, out_idx, out_ids, out_val, out_n, d_n_active, (const uint32_t*)nullptr,
                       (const unsigned long long*)nullptr, (uint64_t)0, 0u, n_stride, (const unsigned long long*)nullptr, (uint64_t)0, mir);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}


int launch_select(orama_ctx* ctx, const SelectPlan& p, hipStream_t stream) {
    ORAMA_REQUIRE(p.k >= 1, "top-k: k is 0");
    ORAMA_SUPPORT(p.k <= kSelectMaxK, "top-k: k=%u outside [1, %u]", p.k, kSelectMaxK);
    ORAMA_REQUIRE(p.q >= 1 && p.vals && p.out_val, "top-k: bad plan");
    ProfScope prof(&ctx->prof, "topk_select", stream);
    if (p.n <= kSelectMaxK) {
        if (p.n == 0) {
            // empty list: emit padding + zero counts through the small kernel with n = 0
        }
        if (p.n >= 1 && p.n <= kRankCountMax && p.k <= kSortThreads && ctx->select_wide != 0) return launch_select_tiny(p, stream);
        hipLaunchKernelGGL(select_small_kernel, dim3(p.q), dim3(kSortThreads), 0, stream, p.vals,
                           p.idx, p.stride, p.n_dev, p.n, p.k, p.descending, p.id_map, p.out_idx,
                           p.out_ids, p.out_val, p.out_n);
        ORAMA_HIP_TRY(hipGetLastError());
        return ORAMA_OK;
    }
    ORAMA_REQUIRE(p.state && p.keys, "top-k: scratch missing");
    const uint64_t expect = p.n_hint ? p.n_hint : p.n;
    const uint64_t list_chunks = std::max<uint64_t>(1, (expect + kKeysChunk - 1) / kKeysChunk);
    // A FEW long lists (a lone query's dense distance array: 1 M values = 122 chunks) take the two launches as well, with as
    // many parts as the final kernel can order (kSelectMaxK / k: 40 at k = 100) walking at most 8 rounds each — the
    // histogram form is 16 launches with a gap before each: 98 us behind a 240 us scan of 1 M x 384 rows
    // (profiles/r04_c2_kernel_stats.md), the two launches ~30.
    const uint32_t parts_cap = p.q <= 4 ? std::max<uint32_t>(16u, kSelectMaxK / p.k) : 16u;
    const bool few_long = p.q <= 4 && list_chunks <= 8ull * std::min<uint32_t>(parts_cap, std::max<uint32_t>(1u, kSelectMaxK / p.k));
    if (p.keys_capacity >= (uint64_t)p.q * kSelectMaxK && (list_chunks <= 16 || few_long) && ctx->select_pairs) {
        // (value, index) lists in two launches: `parts` workgroups per list keep their best k, one orders parts * k keys
        // (lists expected to be longer than 16 chunks keep the histogram passes, which spread one list over the chip).
        // Lists of a length known here (the dense heads of the fp16 scans: 131 072 distances per query) take it too since
        // round 4: 16 workgroups per list, one round each.
        uint32_t parts = kSelectMaxK / p.k;
        parts = std::min<uint32_t>(parts, (uint32_t)list_chunks);
        parts = std::min<uint32_t>(parts, parts_cap);
        // lists of a known length (dense heads): two resident waves of workgroups (2 per CU) — a workgroup's first round costs
        // ~10 us (bound, ~2k survivors, their ranks), every later one runs under the floor of the k-th best so far: 256 lists
        // x 16 parts of one round each took 108 us, x 4 parts of four rounds 88-97
        if (!p.n_dev) parts = std::min<uint32_t>(parts, std::max<uint32_t>(1u, (4u * (uint32_t)ctx->compute_units) / p.q));
        // a list of one round is finished by the workgroup that reduces it (the candidate lists of the fp16 scans: a few
        // hundred entries); the final kernel then finds its word set and ends at once.  The words sit in the histogram
        // state this form does not use.
        PairsFinal fin;
        fin.id_map = p.id_map;
        fin.out_idx = p.out_idx;
        fin.out_ids = p.out_ids;
        fin.out_val = p.out_val;
        fin.out_n = p.out_n;
        fin.done = reinterpret_cast<uint32_t*>(p.state);
        static_assert(sizeof(SelectState) >= sizeof(uint32_t), "one word per list");
        // a few long dense lists whose parts fit one round of 32 values per thread (1 M values in 40 parts at k = 100)
        const int wide_mode = ctx->select_wide;  // ORAMA_SELECT_WIDE: 0 = rounds of 8 192 (round 4), 2 = the fallback rounds only (tests)
        bool wide = wide_mode != 0 && !p.idx && list_chunks > 16 && p.k <= kWaveBoundMaxK &&
                    ((uint64_t)p.n + parts - 1) / parts <= (uint64_t)kSelectWidePartValues;
        // many dense lists of a known length (the dense heads of the fp16 scans: 131 072 distances for each of 64 / 256 queries)
        // as well: 4 parts of one round each per list instead of 16 parts of one round (C3) or 4 parts of four rounds (C5)
        if (!wide && wide_mode != 0 && wide_mode != 3 && !p.idx && !p.n_dev && list_chunks >= 4 && list_chunks <= 16 && p.k <= kWaveBoundMaxK) {
            const uint32_t pw = (uint32_t)(((uint64_t)p.n + kSelectWidePartValues - 1) / kSelectWidePartValues);
            if ((uint64_t)pw * p.k <= kSelectMaxK) {
                parts = pw;
                wide = true;
            }
        }
        if (wide) {
            ORAMA_TRY(launch_pairs_reduce_wide(p, parts, wide_mode == 2, stream));
            fin.done = nullptr;  // (never a whole list: the final kernel always orders)
        } else if (p.idx)
            hipLaunchKernelGGL(pairs_reduce_kernel<true>, dim3(p.q, parts), dim3(kSortThreads), 0, stream, p.vals, p.idx, p.stride,
                               p.n_dev, p.n, p.descending, p.k, p.keys, fin);
        else
            hipLaunchKernelGGL(pairs_reduce_kernel<false>, dim3(p.q, parts), dim3(kSortThreads), 0, stream, p.vals, p.idx, p.stride,
                               p.n_dev, p.n, p.descending, p.k, p.keys, fin);
        hipLaunchKernelGGL(keys_final_kernel, dim3(p.q), dim3(kSortThreads), 0, stream, p.keys, parts * p.k,
                           (uint64_t)parts * p.k, nullptr, p.k, p.descending, p.id_map, p.out_idx, p.out_ids, p.out_val,
                           p.out_n, nullptr, fin.done);
        ORAMA_HIP_TRY(hipGetLastError());
        return ORAMA_OK;
    }
    ORAMA_HIP_TRY(hipMemsetAsync(p.state, 0, sizeof(SelectState) * (size_t)p.q, stream));
    uint32_t blocks = ceil_div_u32(p.n_hint ? p.n_hint : p.n, kHistThreads * 16);
    uint32_t max_blocks = (uint32_t)ctx->compute_units * 8u;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    for (int pass = 0; pass < 6; ++pass) {
        hipLaunchKernelGGL(select_hist_kernel, dim3(blocks, p.q), dim3(kHistThreads), 0, stream,
                           p.vals, p.idx, p.stride, p.n_dev, p.n, p.descending, p.state, pass);
        hipLaunchKernelGGL(select_scan_kernel, dim3(p.q), dim3(256), 0, stream, p.state, p.k, pass);
    }
    hipLaunchKernelGGL(select_collect_kernel, dim3(blocks, p.q), dim3(kHistThreads), 0, stream,
                       p.vals, p.idx, p.stride, p.n_dev, p.n, p.descending, p.state, p.keys, p.k);
    hipLaunchKernelGGL(select_sort_kernel, dim3(p.q), dim3(kSortThreads), 0, stream, p.state, p.keys,
                       p.k, p.k, p.descending, p.id_map, p.out_idx, p.out_ids, p.out_val, p.out_n);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_merge_candidates(orama_ctx* ctx, const uint64_t* d_ids, const float* d_dist,
                            uint32_t lists, uint32_t q, uint32_t k, uint64_t* d_out_ids,
                            float* d_out_dist, uint32_t* d_out_n, hipStream_t stream) {
    (void)ctx;
    ORAMA_REQUIRE(lists >= 1 && q >= 1 && k >= 1, "merge: empty shape");
    ORAMA_SUPPORT((uint64_t)lists * k <= kSelectMaxK, "merge: lists*k=%llu exceeds %u",
                  (unsigned long long)lists * k, kSelectMaxK);
    hipLaunchKernelGGL(merge_candidates_kernel, dim3(q), dim3(kSortThreads), 0, stream,
                       reinterpret_cast<const char*>(d_ids), (uint64_t)q * k * 8,
                       reinterpret_cast<const char*>(d_dist), (uint64_t)q * k * 4, lists, q, k, false, d_out_ids,
                       d_out_dist, d_out_n);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

uint64_t packed_block_bytes(uint32_t q, uint32_t k) {
    return (((uint64_t)q * k * 12) + 7) & ~7ull;
}

int launch_merge_packed(orama_ctx* ctx, const void* d_packed, uint32_t lists, uint32_t q, uint32_t k,
                        uint64_t* d_out_ids, float* d_out_dist, uint32_t* d_out_n,
                        hipStream_t stream) {
    return launch_merge_blocks(ctx, d_packed, packed_block_bytes(q, k), lists, q, k, false, d_out_ids, d_out_dist,
                               d_out_n, stream);
}

int launch_merge_blocks(orama_ctx* ctx, const void* d_blocks, uint64_t block_stride, uint32_t lists, uint32_t q,
                        uint32_t k, bool descending, uint64_t* d_out_ids, float* d_out_val, uint32_t* d_out_n,
                        hipStream_t stream) {
    (void)ctx;
    ORAMA_REQUIRE(lists >= 1 && q >= 1 && k >= 1, "merge: empty shape");
    ORAMA_SUPPORT((uint64_t)lists * k <= kSelectMaxK, "merge: lists*k=%llu exceeds %u",
                  (unsigned long long)lists * k, kSelectMaxK);
    ORAMA_REQUIRE(block_stride >= packed_block_bytes(q, k) && block_stride % 8 == 0, "merge: bad block stride");
    const char* base = reinterpret_cast<const char*>(d_blocks);
    hipLaunchKernelGGL(merge_candidates_kernel, dim3(q), dim3(kSortThreads), 0, stream, base, block_stride,
                       base + (uint64_t)q * k * 8, block_stride, lists, q, k, descending, d_out_ids, d_out_val,
                       d_out_n);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

#else
// ---- the launchers of this unit's two kernels (select.hpp; called by launch_select of the unit "select")
static_assert((uint32_t)kWidePer * kSortThreads == kSelectWidePartValues, "select.hpp states what one part of the wide form takes");

int launch_select_tiny(const SelectPlan& p, hipStream_t stream) {
    PairsFinal fin;
    fin.id_map = p.id_map;
    fin.out_idx = p.out_idx;
    fin.out_ids = p.out_ids;
    fin.out_val = p.out_val;
    fin.out_n = p.out_n;
    hipLaunchKernelGGL(select_tiny_kernel, dim3(p.q), dim3(kSortThreads), 0, stream, p.vals, p.idx, p.stride, p.n_dev, p.n, p.k,
                       p.descending, fin);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_pairs_reduce_wide(const SelectPlan& p, uint32_t parts, bool force_narrow, hipStream_t stream) {
    hipLaunchKernelGGL(pairs_reduce_wide_kernel, dim3(p.q, parts), dim3(kSortThreads), 0, stream, p.vals, p.stride, p.n_dev, p.n,
                       p.descending, p.k, p.keys, force_narrow);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
#endif
}  // namespace orama
