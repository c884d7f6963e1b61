// hybrid_tail.hip — the tail of the one-call hybrid search on the DEVICE (round 5).
//
// orama_hybrid_search runs the vector scan and the full-text leg (K3r) side by side; round 3 left behind the scan: the
// scan's top-k, a read-back, a HOST wake-up (a2 epilogue, the ids' local indices), one tiny per-document launch, a second
// wake-up, the host merge.  Here everything between the scan's top-k and the answer stays on the stream:
//
//   hybrid_vec_epilogue_kernel   EmbeddingFieldStorage::search's epilogue (embedding_field.rs:264-276) over the <= limit rows of
//                                the vector block: similarity = 1 - distance, Model::rescale_score (python/embeddings.rs:71-92),
//                                the cut-off, `out[doc] += score` in hit order — the entries in order of their first passing
//                                hit, as the host loop of orama_hybrid_search builds them — and each document's local index;
//   range_score_docs_kernel      (bm25_ranges.hip) the full-text score of those documents, same fold as the range kernel;
//   hybrid_merge_kernel          normalize_and_combine + count (token_score.rs:393-422, search.rs:482) over the vector hits
//                                and the raw full-text candidates, and the exactness rule of hybrid_from_candidates
//                                (fulltext.hip): `flag` = the candidates cannot prove the answer -> the host falls back;
//   K4 (select.hip)              top_n over the <= top_k + 2 limit + 1 entries: score desc, DocumentId asc.
//
// One read-back, one wake-up.  Same f32 operations in the same order as the host forms (this unit is compiled with
// -ffp-contract=off like every unit that must round like the scalar reference): bit-identical answers, tested against the
// host tail, K3 and the CPU restatement (tests/test_bm25_ranges_gpu.py::test_one_call_hybrid_search_on_the_range_scorer,
// tests/test_full_size_gpu.py::test_c4_full_size_hybrid_bit_exact).
#include "hybrid_tail.hpp"

#include "device_utils.hpp"

namespace orama {

namespace {

constexpr int kTailThreads = 256;

// `n` <= kHybridTailMaxVec rows; one workgroup.
__global__ __launch_bounds__(kTailThreads) void hybrid_vec_epilogue_kernel(HybridTailArgs a) {
    __shared__ unsigned long long ids[kHybridTailMaxVec];
    __shared__ float score[kHybridTailMaxVec];
    __shared__ uint32_t first[kHybridTailMaxVec];  // 1: the first passing hit of its document
    __shared__ uint32_t n_first, foreign;
    const uint32_t n = min(*a.v_n, a.limit);
    if (threadIdx.x == 0) {
        n_first = 0;
        foreign = 0;
    }
    for (uint32_t i = threadIdx.x; i < n; i += kTailThreads) {
        const float similarity = 1.0f - a.v_dist[i];
        float sc = similarity;
        if (a.rescale_e5) {  // Model::rescale_score, src/python/embeddings.rs:71-92
            const float MIN = 0.7f, MAX = 1.0f, DELTA = MAX - MIN;
            float c = similarity;
            if (c < MIN) c = MIN;
            if (c > MAX) c = MAX;
            sc = (c - MIN) / DELTA;
        }
        ids[i] = a.v_ids[i];
        // a hit under the cut-off (or NaN) takes no part: marked by a score the loops below skip
        score[i] = sc;
        first[i] = (sc >= a.min_similarity) ? 1u : 2u;  // 2: dropped
    }
    __syncthreads();
    // the first PASSING hit of every document owns its entry; its thread adds the document's passing hits in hit order
    // (`*entry += score` from 0.0, embedding_field.rs:276)
    for (uint32_t i = threadIdx.x; i < n; i += kTailThreads) {
        if (first[i] == 2u) continue;
        bool is_first = true;
        for (uint32_t j = 0; j < i && is_first; ++j) is_first = !(first[j] != 2u && ids[j] == ids[i]);
        if (!is_first) {
            first[i] = 0u;  // (only ever read by its own thread again, and by the count below after the barrier)
            continue;
        }
    }
    __syncthreads();
    // positions in order of the first passing hit: a serial count is <= 512 steps of one lane; the wave does it by ballots
    if (threadIdx.x < 64) {
        uint32_t base = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + threadIdx.x;
            const bool f = i < n && first[i] == 1u;
            const unsigned long long m = __ballot(f);
            if (f) first[i] = 0x80000000u | (base + (uint32_t)__popcll(m & ((1ull << threadIdx.x) - 1ull)));
            base += (uint32_t)__popcll(m);
        }
        if (threadIdx.x == 0) n_first = base;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += kTailThreads) {
        if (!(first[i] & 0x80000000u)) continue;
        const uint32_t pos = first[i] & 0x7fffffffu;
        float sum = 0.0f;
        for (uint32_t j = i; j < n; ++j)
            if (ids[j] == ids[i] && score[j] >= a.min_similarity) sum = sum + score[j];
        const uint64_t doc = ids[i];
        uint32_t local = 0;
        bool found;
        if (a.dense) {
            found = doc >= a.dense_base && doc - a.dense_base < a.n_docs;
            local = found ? (uint32_t)(doc - a.dense_base) : 0u;
        } else {
            uint64_t lo = 0, hi = a.n_docs;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (a.docs[mid] < doc) lo = mid + 1; else hi = mid;
            }
            found = lo < a.n_docs && a.docs[lo] == doc;
            local = (uint32_t)lo;
        }
        if (!found) atomicOr(&foreign, 1u);
        a.vdoc[pos] = doc;
        a.vsc[pos] = sum;
        a.vlocal[pos] = found ? local : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.state[0] = n_first;   // entries of the vector map
        a.state[1] = foreign;   // a hit is not a document of this index: the per-record scorer reports it
    }
}

// One workgroup.  Entries: [vector hits | full-text-only candidates], NaN scores left out (never selected), count as the
// host form computes it.
__global__ __launch_bounds__(kTailThreads) void hybrid_merge_kernel(HybridTailArgs a) {
    __shared__ float red_mx[kTailThreads / 64], red_mn[kTailThreads / 64];
    __shared__ uint32_t cursor, above, absent;
    const uint32_t nv = a.state[0];
    const uint32_t n_cand = min(*a.cand_n, a.k_asked);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        cursor = 0;
        above = 0;
        absent = 0;
    }
    // fold(0.0, f32::max / f32::min) over both maps (token_score.rs:398-407); NaN ignored like f32::max
    float mx = 0.0f, mn = 0.0f;
    for (uint32_t j = threadIdx.x; j < nv; j += kTailThreads) {
        const float v = a.vsc[j];
        if (v != v) continue;
        if (v > mx) mx = v;
        if (v < mn) mn = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        mn = fminf(mn, __shfl_xor(mn, off, 64));
    }
    if (lane == 0) {
        red_mx[wave] = mx;
        red_mn[wave] = mn;
    }
    __syncthreads();
    mx = red_mx[0];
    mn = red_mn[0];
#pragma unroll
    for (int w = 1; w < kTailThreads / 64; ++w) {
        mx = fmaxf(mx, red_mx[w]);
        mn = fminf(mn, red_mn[w]);
    }
    const uint32_t max_key = *a.res_max_key, min_inv = *a.res_min_inv;
    if (max_key != 0u) {
        const float v = ordered_to_f32(max_key);
        if (v > mx) mx = v;
    }
    if (min_inv != 0u) {
        const float v = ordered_to_f32(~min_inv);
        if (v < mn) mn = v;
    }
    const float den = mx - mn;
    // the last raw candidate normalised: an unseen full-text document normalises to <= it
    float last_norm = 0.0f;
    const bool have_last = n_cand > 0;
    if (have_last) last_norm = (a.cand_score[n_cand - 1] - mn) / den;
    const uint32_t total = nv + n_cand;
    for (uint32_t e0 = 0; e0 < total; e0 += kTailThreads) {
        const uint32_t e = e0 + threadIdx.x;
        bool keep = false;
        float sc = 0.0f;
        uint64_t doc = 0;
        if (e < nv) {  // vector hits: ft' + v' when the document is in the full-text map, 0.0 + v' otherwise
            const float v = (a.vsc[e] - mn) / den;
            if (a.vpresent[e]) {
                sc = (a.vft[e] - mn) / den;
                sc = sc + v;
            } else {
                sc = 0.0f + v;
                atomicAdd(&absent, 1u);
            }
            doc = a.vdoc[e];
            keep = true;
        } else if (e < total) {  // full-text-only candidates
            const uint32_t i = e - nv;
            doc = a.cand_id[i];
            bool is_vec = false;
            for (uint32_t j = 0; j < nv && !is_vec; ++j) is_vec = a.vdoc[j] == doc;
            sc = (a.cand_score[i] - mn) / den;
            keep = !is_vec;
        }
        keep = keep && sc == sc;  // NaN never selected (sort.rs:262-266)
        const unsigned long long m = __ballot(keep);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&cursor, (uint32_t)__popcll(m));
        base = __shfl(base, 0, 64);
        if (keep) {
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            a.e_score[pos] = sc;
            a.e_doc[pos] = doc;
            if (sc > last_norm) atomicAdd(&above, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n_entries = cursor;
        a.state[2] = n_entries;
        // exactness (hybrid_from_candidates): were there full-text documents beyond the candidates that could still enter the
        // top_k?  Only when the candidate list is full; then the k-th final score must lie strictly above what an unseen
        // document can reach — i.e. at least top_k entries score above last_norm — unless last_norm is NaN (nothing unseen
        // can be selected either)
        uint32_t flag = a.state[1] ? 2u : 0u;
        if (n_cand >= a.k_asked && have_last && a.top_k > 0 && last_norm == last_norm && above < a.top_k) flag |= 1u;
        if (*a.res_overflow) flag |= 4u;  // the ranges overflowed: the host reruns the full-text leg (nothing here is used)
        a.out_flag[0] = flag;
        a.out_count[0] = (unsigned long long)*a.res_count + absent;
    }
}

}  // namespace

int launch_hybrid_vec_epilogue(const HybridTailArgs& a, hipStream_t stream) {
    ORAMA_REQUIRE(a.limit >= 1 && a.limit <= kHybridTailMaxVec, "hybrid tail: limit outside the device form's envelope");
    hipLaunchKernelGGL(hybrid_vec_epilogue_kernel, dim3(1), dim3(kTailThreads), 0, stream, a);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_hybrid_merge(const HybridTailArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(hybrid_merge_kernel, dim3(1), dim3(kTailThreads), 0, stream, a);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
