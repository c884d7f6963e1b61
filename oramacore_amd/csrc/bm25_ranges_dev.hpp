// bm25_ranges_dev.hpp — device helpers shared by the range scorer's kernels (bm25_ranges.hip and the comparison build of the
// round-2/3 merge-tree kernel, bm25_ranges_merge.hip).  Compiled with -ffp-contract=off (see bm25_kernels.hip).
#pragma once

#include "bm25_ranges.hpp"
#include "device_utils.hpp"

namespace orama {

__device__ __forceinline__ bool f32_is_normal(float x) {
    const uint32_t e = (__builtin_bit_cast(uint32_t, x) >> 23) & 0xffu;
    return e != 0u && e != 0xffu;
}

// The additions BM25Scorer performs for ONE document, fed its (token, ntf) contributions in (token, reference) order:
// S_t = sum of the token's ntf (Iterator::sum from 0.0, weight 1.0), score += idf_t (k+1) S_t / (k + S_t) unless S_t is
// not normal or the term is NaN, token mask for the threshold (bm25.rs:369-428, 484-524).  Shared by the range kernels
// and the per-document kernel of the hybrid path so that all of them produce the same bits.
struct DocFold {
    float score = 0.0f;  // entry(key).or_insert(0.0)
    uint32_t mask = 0u;
    bool applied = false, have = false;
    uint32_t tok = 0;
    float sum = 0.0f;
    __device__ __forceinline__ void close_token(const float* idf, float k, float k1) {
        if (f32_is_normal(sum)) {
            const float term = idf[tok] * k1 * sum / (k + sum);  // bm25f_score, bm25.rs:124-126
            if (term == term) {
                score = score + term * 1.0f;  // phrase boost 1.0
                mask |= 1u << (tok & 31u);    // 1 << term_index on u32 (wrapping shift)
                applied = true;
            }
        }
    }
    __device__ __forceinline__ void add(uint32_t t, float ntf, const float* idf, float k, float k1) {
        if (have && t != tok) {
            close_token(idf, k, k1);
            sum = 0.0f;
        }
        tok = t;
        have = true;
        sum = sum + 1.0f * ntf;  // Iterator::sum() from 0.0, weight 1.0
    }
    // a token whose contributions were already summed in reference order (S = 0.0 + ntf_0 + ntf_1 ...): same bits as add() x n
    __device__ __forceinline__ void add_summed(uint32_t t, float s, const float* idf, float k, float k1) {
        tok = t;
        have = true;
        sum = s;
        close_token(idf, k, k1);
        have = false;
    }
    // true: the document is in the score map (threshold passed); `score` is final (before OMC)
    __device__ __forceinline__ bool finish(const float* idf, float k, float k1, uint32_t use_threshold, uint32_t threshold) {
        if (have) close_token(idf, k, k1);
        return applied && !(use_threshold && (uint32_t)__popc(mask) < threshold);
    }
};

// Normalised term frequency of one posting, in the operation order K3 and the CPU restatement use:
//   boost * (tf / ((1 - b) + b * (len / avg_len)))     — bm25.rs:99-110 with the field boost folded in
// `pre` is the part that does not depend on the query (stored per posting by the store, see ntf_precompute_kernel).
__device__ __forceinline__ float ntf_pre_of(uint32_t val, float one_minus_b, float b, float avg_len) {
    const float tf = (float)(val >> 16);
    const float len = (float)(val & 0xffffu);
    return tf / (one_minus_b + b * (len / avg_len));
}

}  // namespace orama
