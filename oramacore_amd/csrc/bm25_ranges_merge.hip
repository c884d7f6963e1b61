// bm25_ranges_merge.hip — COMPARISON BUILD ONLY (ORAMA_COMPARISON_KERNELS=1): the round-2/3 form of K3r's scoring launch.
//
// One workgroup per (range, query) gathers the <= 2048 postings of its range, packs each into a 64-bit key
// [local doc:15 | token:6 | rank:10 | dropped:1 | ntf bits:32], merges the per-list runs with a merge tree in LDS (merge
// path, log2(lists) levels) and lets the first thread of every document walk its run in (token, reference) order.
// 8.6 VALU wave-instructions per posting (profiles/r03_k3r_sq_counters.md); replaced in round 4 by the sort-free kernel
// of bm25_ranges.hip, which performs the same additions in the same order.  Kept so that one build can time both
// (orama_ctx_set_k3r_merge / ORAMA_K3R_MERGE=1) and check them bit for bit against each other; not part of the product
// library.
//
// Compiled with -ffp-contract=off (see bm25_kernels.hip).
#include "bm25_ranges.hpp"

#include "bm25_ranges_dev.hpp"

#if ORAMA_COMPARISON_KERNELS

namespace orama {

namespace {

constexpr int kThreads = 256;

// key of the in-range merge: [local doc:15 | token:6 | rank:10 | dropped:1 | ntf bits:32] — the order of the upper
// 31 bits is (document, token, list rank); a posting dropped by the filter keeps its place in its run.
__device__ __forceinline__ uint32_t key_doc(unsigned long long k) { return (uint32_t)(k >> 49); }
__device__ __forceinline__ uint32_t key_tok(unsigned long long k) { return (uint32_t)(k >> 43) & 63u; }
__device__ __forceinline__ uint32_t key_doc_tok(unsigned long long k) { return (uint32_t)(k >> 43); }
__device__ __forceinline__ bool key_dropped(unsigned long long k) { return (k >> 32) & 1ull; }

constexpr int kPerThread = kRangeCap / kThreads;  // merge elements a thread carries in registers

template <bool DF_ONLY>
__global__ __launch_bounds__(kThreads) void range_score_merge_kernel(RangeBatch b) {
    __shared__ unsigned long long s[kRangeCap];
    __shared__ unsigned long long seg_pos[kRangeMaxRefs];  // first posting of each reference inside this range
    __shared__ uint32_t seg_off[kRangeMaxRefs + 1];        // start of each reference's run among the gathered postings
    __shared__ uint32_t seg_key[kRangeMaxRefs];            // token << 11 | rank << 1
    __shared__ float seg_boost[kRangeMaxRefs], seg_avg[kRangeMaxRefs];
    __shared__ float idf[kMaxTokens];
    __shared__ uint32_t df_lds[kMaxTokens];
    __shared__ uint32_t red[4];

    const uint32_t qi = blockIdx.y;
    const RangeQuery q = b.queries[qi];
    const uint32_t r = blockIdx.x;
    if (r >= q.n_ranges) return;
    if (DF_ONLY && !q.want_df) return;
    const uint32_t ns = q.seg_end - q.seg_begin;
    const RangeSeg* segs = b.segs + q.seg_begin;

    if (threadIdx.x < 4) red[threadIdx.x] = 0;
    for (uint32_t t = threadIdx.x; t < kMaxTokens; t += kThreads) {
        df_lds[t] = 0;
        if (!DF_ONLY) idf[t] = t < q.n_tokens ? b.idf[(size_t)qi * kMaxTokens + t] : 0.0f;
    }
    __syncthreads();
    // this range's run of every reference (the bounds of a range are contiguous over the references, and their
    // address does not depend on the reference table: both loads are issued together);
    // slot base = postings of the query in earlier ranges
    uint32_t base_part = 0;
    for (uint32_t i = threadIdx.x; i < ns; i += kThreads) {
        const uint32_t* row = b.bounds + q.bounds_base + (uint64_t)r * ns;
        const uint32_t b0 = row[i], b1 = row[ns + i];
        const RangeSeg sg = segs[i];
        seg_pos[i] = sg.post_begin + b0;
        seg_off[i + 1] = b1 - b0;
        seg_key[i] = sg.tok_rank << 1;
        seg_boost[i] = sg.boost;
        seg_avg[i] = sg.avg_len;
        base_part += b0;
    }
    base_part = wave_sum_u32(base_part);
    if ((threadIdx.x & 63) == 0 && base_part) atomicAdd(&red[0], base_part);
    __syncthreads();
    if (threadIdx.x < 64) {  // inclusive scan of the run lengths by one wave, 64 references at a time
        uint32_t carry = 0;
        for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
            const uint32_t i = i0 + threadIdx.x;
            uint32_t x = i < ns ? seg_off[i + 1] : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off, 64);
                if ((int)threadIdx.x >= off) x += y;
            }
            if (i < ns) seg_off[i + 1] = carry + x;
            carry += __shfl(x, 63, 64);
        }
        if (threadIdx.x == 0) seg_off[0] = 0;
    }
    __syncthreads();
    const uint32_t cap = seg_off[ns];
    if (cap == 0 || (b.debug & 4u)) return;
    const uint32_t slot_base = red[0];
    if (cap > kRangeCap) {
        // the query is rerun with smaller ranges; its slots still reach the batch's top-k, so they must be empty
        if (!DF_ONLY)
            for (uint32_t e = threadIdx.x; e < cap; e += kThreads) {
                b.keys[q.key_off + slot_base + e] = 0ull;
                if (b.map_idx) b.map_idx[slot_base + e] = 0xffffffffu;
            }
        if (threadIdx.x == 0) b.results[qi].overflow = 1;
        return;
    }

    // gather: element e belongs to the reference whose [seg_off[i], seg_off[i+1]) holds it; runs are sorted by
    // document, so by key
    const float one_minus_b = 1.0f - b.b;
    const uint32_t doc0 = r * q.width;
    for (uint32_t e = threadIdx.x; e < cap; e += kThreads) {
        uint32_t lo = 0, hi = ns;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (seg_off[mid] <= e) lo = mid; else hi = mid;
        }
        const uint64_t p = seg_pos[lo] + (e - seg_off[lo]);
        const uint32_t doc = b.post_doc[p];
        uint32_t dropped = 0;
        if (b.allow) {  // collect_contributions_with_filter: filtered docs never reach the scorer
            const uint64_t id = b.docs ? b.docs[doc] : b.dense_base + doc;  // dense ids: no table lookup
            dropped = !(id < b.allow_bits && ((b.allow[id >> 6] >> (id & 63)) & 1ull));
        }
        uint32_t ntf_bits = 0;
        if (!DF_ONLY) {
            const uint32_t val = b.post_val[p];
            const float tf = (float)(val >> 16);
            const float len = (float)(val & 0xffffu);
            const float ntf = seg_boost[lo] * (tf / (one_minus_b + b.b * (len / seg_avg[lo])));
            ntf_bits = __builtin_bit_cast(uint32_t, ntf);
        }
        s[e] = ((unsigned long long)(((doc - doc0) << 17) | seg_key[lo] | dropped) << 32) | ntf_bits;
    }
    __syncthreads();
    // merge tree over the runs: at level l the sorted groups are 2^l consecutive references; neighbouring groups are
    // merged pairwise (merge path): a thread produces K = ceil(cap / 256) CONSECUTIVE outputs — one binary search
    // along its diagonal finds how many elements of each group precede its first output, then it merges sequentially
    // (one LDS read per output).  The spans of the groups never change, only the order inside them, so a thread's
    // first output stays in the pair of the run it started in.  The upper 32 bits of a key are unique: no ties.
    // Outputs travel through registers: produce everything, barrier, write in place, barrier.
    {
        const uint32_t K = (cap + kThreads - 1) / kThreads;  // 1..kPerThread
        const uint32_t o_begin = threadIdx.x * K;
        const uint32_t o_end = min(cap, o_begin + K);
        uint32_t run0 = 0;
        if (o_begin < cap) {
            uint32_t lo = 0, hi = ns;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (seg_off[mid] <= o_begin) lo = mid; else hi = mid;
            }
            run0 = lo;
        }
        constexpr unsigned long long kEnd = ~0ull;  // above every key (local documents use 15 bits)
        const bool skip_merge = (b.debug & 1u) || (DF_ONLY && q.want_df == 2u);
        for (uint32_t lvl = 0; (1u << lvl) < ns && !skip_merge; ++lvl) {
            unsigned long long outv[kPerThread];
            if (o_begin < cap) {
                uint32_t pair = run0 >> (lvl + 1);
                uint32_t sa = seg_off[min(ns, (2u * pair) << lvl)];
                uint32_t sm = seg_off[min(ns, (2u * pair + 1u) << lvl)];
                uint32_t sb = seg_off[min(ns, (2u * pair + 2u) << lvl)];
                // merge path: i elements of the left group and diag - i of the right one precede output o_begin
                const uint32_t diag = o_begin - sa, len_a = sm - sa, len_b = sb - sm;
                uint32_t lo = diag > len_b ? diag - len_b : 0u, hi = min(diag, len_a);
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s[sa + mid] < s[sm + (diag - 1u - mid)]) lo = mid + 1u; else hi = mid;
                }
                uint32_t i = sa + lo, j = sm + (diag - lo);
                unsigned long long ka = i < sm ? s[i] : kEnd, kb = j < sb ? s[j] : kEnd;
                uint32_t o = o_begin;
#pragma unroll
                for (int n = 0; n < kPerThread; ++n) {
                    outv[n] = 0ull;
                    if ((uint32_t)n >= K) break;  // (workgroup-uniform: a range of <= 1 024 postings skips half the unrolled steps)
                    if (o < o_end) {
                        while (o >= sb) {  // the outputs continue in the next pair, from its beginning
                            ++pair;
                            sa = sb;
                            sm = seg_off[min(ns, (2u * pair + 1u) << lvl)];
                            sb = seg_off[min(ns, (2u * pair + 2u) << lvl)];
                            i = sa;
                            j = sm;
                            ka = i < sm ? s[i] : kEnd;
                            kb = j < sb ? s[j] : kEnd;
                        }
                        if (kb < ka) {
                            outv[n] = kb;
                            ++j;
                            kb = j < sb ? s[j] : kEnd;
                        } else {
                            outv[n] = ka;
                            ++i;
                            ka = i < sm ? s[i] : kEnd;
                        }
                        ++o;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < kPerThread; ++n)
                if ((uint32_t)n < K && o_begin + (uint32_t)n < o_end) s[o_begin + n] = outv[n];
            __syncthreads();
        }
    }

    if (DF_ONLY) {
        // corpus_docs.len(): distinct (token, document) pairs among the kept postings (token_score.rs:262-275).
        // want_df == 2: every token has ONE list, so every kept posting is its own pair — counted as gathered, unmerged.
        for (uint32_t e = threadIdx.x; e < cap; e += kThreads) {
            const unsigned long long key = s[e];
            if (key_dropped(key)) continue;
            bool first = true;
            if (q.want_df == 2u) {
                atomicAdd(&df_lds[key_tok(key)], 1u);
                continue;
            }
            for (uint32_t j = e; j > 0; --j) {
                const unsigned long long kj = s[j - 1];
                if (key_doc_tok(kj) != key_doc_tok(key)) break;
                if (!key_dropped(kj)) {
                    first = false;
                    break;
                }
            }
            if (first) atomicAdd(&df_lds[key_tok(key)], 1u);
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < q.n_tokens; t += kThreads)
            if (df_lds[t]) atomicAdd(&b.results[qi].df[t], df_lds[t]);
        return;
    }

    const float k1 = q.k + 1.0f;
    unsigned long long* out = b.keys + q.key_off + slot_base;
    uint32_t my_count = 0, my_max = 0u, my_min_inv = 0u;
    for (uint32_t e = threadIdx.x; e < cap; e += kThreads) {
        const unsigned long long key = s[e];
        unsigned long long out_key = 0ull;
        uint32_t map_doc = 0xffffffffu;  // the document whose map entry this slot holds (score-map mode)
        float map_score = 0.0f;
        if (!(b.debug & 2u) && (e == 0 || key_doc(s[e - 1]) != key_doc(key))) {
            // first posting of a document: fold its run (lists of a token in reference order, tokens ascending)
            const uint32_t dl = key_doc(key);
            DocFold f;
            unsigned long long kj = key;
            for (uint32_t j = e;;) {
                if (!key_dropped(kj)) f.add(key_tok(kj), __builtin_bit_cast(float, (uint32_t)kj), idf, q.k, k1);
                if (++j >= cap) break;
                kj = s[j];
                if (key_doc(kj) != dl) break;
            }
            if (f.finish(idf, q.k, k1, q.use_threshold, q.threshold)) {
                const uint32_t doc = doc0 + dl;
                float score = f.score;
                if (q.track_minmax && score == score) {  // hybrid: min / max of the full-text scores (before any OMC)
                    const uint32_t ord = f32_to_ordered(score);
                    my_max = max(my_max, ord);
                    my_min_inv = max(my_min_inv, ~ord);
                }
                if (b.omc_dense) score = score * b.omc_dense[doc];
                ++my_count;
                map_doc = doc;
                map_score = score;
                if (score == score)  // a NaN score stays in the map (count) and is never selected
                    out_key = ((unsigned long long)f32_to_ordered(score) << 32) | (unsigned long long)(~doc);
            }
        }
        out[e] = out_key;
        if (b.map_idx) {
            // score-map mode (a batch of ONE query): slot = position of the entry in the map's candidate list, the
            // per-document table points back at it (ScoreMapDev, facets.hip) — NaN scores included, they count
            const uint32_t pos = slot_base + e;
            b.map_idx[pos] = map_doc;
            if (map_doc != 0xffffffffu) {
                b.map_score[pos] = map_score;
                b.map_emit[map_doc] = ((unsigned long long)b.map_epoch << 32) | pos;
            }
        }
    }
    my_count = wave_sum_u32(my_count);
    if ((threadIdx.x & 63) == 0 && my_count) atomicAdd(&red[1], my_count);
    if (q.track_minmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            my_max = max(my_max, (uint32_t)__shfl_xor((int)my_max, off, 64));
            my_min_inv = max(my_min_inv, (uint32_t)__shfl_xor((int)my_min_inv, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            if (my_max) atomicMax(&red[2], my_max);
            if (my_min_inv) atomicMax(&red[3], my_min_inv);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (red[1]) atomicAdd(&b.results[qi].count, red[1]);
        if (red[2]) atomicMax(&b.results[qi].max_key, red[2]);
        if (red[3]) atomicMax(&b.results[qi].min_inv, red[3]);
    }
}

}  // namespace

int launch_range_score_merge(orama_ctx* ctx, const RangeBatch& b, bool df_only, hipStream_t stream) {
    ProfScope prof(&ctx->prof, df_only ? "bm25_range_df" : "bm25_range_score", stream);
    const dim3 grid(b.max_ranges, b.n_queries);
    if (df_only) hipLaunchKernelGGL(range_score_merge_kernel<true>, grid, dim3(kThreads), 0, stream, b);
    else hipLaunchKernelGGL(range_score_merge_kernel<false>, grid, dim3(kThreads), 0, stream, b);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama

#endif  // ORAMA_COMPARISON_KERNELS
