// bm25_kernels.hip — K3: postings accumulate → BM25F finalise; K5: hybrid min-max combine.
//
// Compiled with -ffp-contract=off: every f32 operation below rounds once, in the order the
// reference's scalar Rust evaluates it, so scores are bit-identical to the CPU restatement
// (idf comes from the host's libm log1pf — the function Rust's f32::ln_1p lowers to — through a
// table or a per-token array, never from the device math library).
//
// Data flow per query (all HBM-resident, no host round trip in the resident mode):
//   accumulate (one launch per entry rank; rank-0 covers every token's first posting list):
//       for each posting (doc, tf|len) of each (token, list) reference, coalesced 2 x 4-byte loads,
//       S[doc][token] (+)= boost * tf / (1 - b + b*len/avglen); first touch of (token, doc) bumps
//       df[token]; first touch of doc appends it to the touched list.  Inside one launch a
//       (token, doc) cell is written by at most one thread (docs are unique inside a list), so the
//       f32 accumulation order across a token's lists is the list order — deterministic, no atomics
//       on the accumulators.
//   finalise: one thread per touched doc walks the tokens IN ORDER:
//       score += idf_t * (k+1) * S / (k+S)  (skipped unless S.is_normal(), bm25.rs:387/501),
//       token mask, threshold filter (bm25.rs:416-428), OMC multiply (search.rs:39-48), and emits
//       the doc into the candidate list = the score map (its length is `count`, search.rs:482).
//   [hybrid] normalise / add vector scores / OMC on the candidate list (token_score.rs:393-422).
//   K4 top-k over the candidate list.
// Roofline: HBM (gather/scatter): algorithmic bytes = 8 B per posting + 4 B per touched doc.
#include "bm25_kernels.hpp"

#include "device_utils.hpp"

namespace orama {

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kNoDoc = 0xffffffffu;  // empty slot of the touched / candidate lists

__device__ __forceinline__ bool f32_is_normal(float x) {
    const uint32_t e = (__builtin_bit_cast(uint32_t, x) >> 23) & 0xffu;
    return e != 0u && e != 0xffu;
}

template <bool PRE>
__global__ __launch_bounds__(kThreads) void bm25_accumulate_kernel(Bm25Accum a) {
    __shared__ Bm25Seg segs[kMaxTokens];
    __shared__ uint32_t df_lds[kMaxTokens];
    for (uint32_t i = threadIdx.x; i < a.n_segs; i += kThreads) segs[i] = a.segs[i];
    for (uint32_t i = threadIdx.x; i < kMaxTokens; i += kThreads) df_lds[i] = 0;
    __syncthreads();
    const float one_minus_b = 1.0f - a.b;
    for (uint64_t v = (uint64_t)blockIdx.x * kThreads + threadIdx.x; v < a.total;
         v += (uint64_t)gridDim.x * kThreads) {
        // segment lookup: segs are sorted by virt_begin (<= 64 entries)
        uint32_t lo = 0, hi = a.n_segs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (segs[mid].virt_begin <= v) lo = mid; else hi = mid;
        }
        const Bm25Seg& s = segs[lo];
        const uint64_t p = s.post_begin + (v - s.virt_begin);
        const uint32_t doc = a.post_doc[p];
        const uint32_t val = a.post_val[p];
        uint32_t first_touch_doc = kNoDoc;
        if (a.allow) {  // collect_contributions_with_filter: filtered docs never reach the scorer
            const uint64_t id = a.docs[doc];
            if (id >= a.allow_bits || !((a.allow[id >> 6] >> (id & 63)) & 1ull)) {
                a.touched[a.virt_base + v] = kNoDoc;
                continue;
            }
        }
        float ntf;
        if (PRE) {
            ntf = __builtin_bit_cast(float, val);
        } else {
            const float tf = (float)(val >> 16);
            const float len = (float)(val & 0xffffu);
            ntf = s.boost * (tf / (one_minus_b + a.b * (len / s.avg_len)));
        }
        unsigned long long* rec = a.acc + (uint64_t)doc * a.slots;
        unsigned long long* w = rec + s.token;
        const unsigned long long old = *w;
        float sum;
        if ((uint32_t)(old >> 32) != a.epoch) {
            sum = 0.0f + 1.0f * ntf;  // Iterator::sum() from 0.0, weight = 1.0
            atomicAdd(&df_lds[s.token], 1u);
            // first touch of the doc in this query: the stamp lives in the HIGH word of the record's last cell,
            // the same position accumulator cells keep their epoch, so a scratch set shared by queries with
            // different record sizes can never mistake stale float bits for a stamp
            if (atomicExch(reinterpret_cast<uint32_t*>(rec + (a.slots - 1)) + 1, a.epoch) != a.epoch) {
                first_touch_doc = doc;
            }
        } else {
            sum = __builtin_bit_cast(float, (uint32_t)old) + 1.0f * ntf;
        }
        *w = ((unsigned long long)a.epoch << 32) | (unsigned long long)__builtin_bit_cast(uint32_t, sum);
        a.touched[a.virt_base + v] = first_touch_doc;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kMaxTokens; i += kThreads)
        if (df_lds[i]) atomicAdd(&a.state->df[i], df_lds[i]);
}

template <bool TRACK_MINMAX>
__global__ __launch_bounds__(kThreads) void bm25_finalize_kernel(Bm25Finalize f) {
    __shared__ float idf[kMaxTokens];
    for (uint32_t t = threadIdx.x; t < f.n_tokens; t += kThreads) idf[t] = f.idf_vals[t];
    __syncthreads();
    const uint32_t n = f.n_slots;
    const float k1 = f.k + 1.0f;
    uint32_t my_max = 0u, my_min = 0xffffffffu, my_count = 0u;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const uint32_t doc = f.touched[i];
        // slot i of the candidate list mirrors slot i of the touched list; empty = (NaN, kNoDoc)
        f.cand_idx[i] = kNoDoc;
        f.cand_score[i] = __builtin_nanf("");
        if (doc == kNoDoc) continue;
        float score = 0.0f;  // entry(key).or_insert(0.0)
        uint32_t mask = 0u;
        bool applied = false;
        const unsigned long long* rec = f.acc + (uint64_t)doc * f.slots;
        // the record is read in groups of 16 cells issued together (one round trip for up to 15 tokens instead of
        // one dependent load per token); the tokens are still folded in ascending order
        for (uint32_t t0 = 0; t0 < f.n_tokens; t0 += 16) {
            unsigned long long w[16];
            if (f.slots >= 4) {  // records of >= 4 cells are 32-byte aligned: 16-byte vector loads
                typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ull2 v = {0ull, 0ull};
                    if (t0 + 2 * j < f.slots) v = *reinterpret_cast<const ull2*>(rec + t0 + 2 * j);
                    w[2 * j] = v.x;
                    w[2 * j + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = (t0 + j < f.slots) ? rec[t0 + j] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t t = t0 + j;
                if (t >= f.n_tokens) break;
                if ((uint32_t)(w[j] >> 32) != f.epoch) continue;
                const float s = __builtin_bit_cast(float, (uint32_t)w[j]);
                if (!f32_is_normal(s)) continue;
                const float term = idf[t] * k1 * s / (f.k + s);  // bm25f_score, bm25.rs:124-126
                if (term != term) continue;
                score = score + term * 1.0f;  // phrase boost 1.0
                mask |= 1u << (t & 31u);      // 1 << term_index on u32 (wrapping shift)
                applied = true;
            }
        }
        if (!applied) continue;
        if (f.use_threshold && (uint32_t)__popc(mask) < f.threshold) continue;
        if (f.omc_dense) score = score * f.omc_dense[doc];
        f.cand_score[i] = score;
        f.cand_idx[i] = doc;
        f.emit[doc] = ((unsigned long long)f.epoch << 32) | (unsigned long long)i;
        ++my_count;
        if (TRACK_MINMAX && score == score) {
            const uint32_t key = f32_to_ordered(score);
            my_max = max(my_max, key);
            my_min = min(my_min, key);
        }
    }
    // wave reduce → LDS block reduce → ONE atomic per block and quantity (a hot word retires ~88 atomics/µs,
    // so per-wave atomics on `cand_count` alone used to cost ~0.1 ms at 600K slots)
    __shared__ uint32_t blk_max, blk_min, blk_count;
    if (threadIdx.x == 0) {
        blk_max = 0u;
        blk_min = 0xffffffffu;
        blk_count = 0u;
    }
    __syncthreads();
    my_count = wave_sum_u32(my_count);
    if (TRACK_MINMAX) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            my_max = max(my_max, (uint32_t)__shfl_xor((int)my_max, off, 64));
            my_min = min(my_min, (uint32_t)__shfl_xor((int)my_min, off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        if (my_count) atomicAdd(&blk_count, my_count);
        if (TRACK_MINMAX) {
            atomicMax(&blk_max, my_max);
            atomicMin(&blk_min, my_min);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blk_count) atomicAdd(&f.state->cand_count, blk_count);
        if (TRACK_MINMAX) {
            if (blk_max != 0u) atomicMax(&f.state->max_key, blk_max);
            if (blk_min != 0xffffffffu) atomicMin(&f.state->min_key, blk_min);
        }
    }
}

// fold(0.0, f32::max) / fold(0.0, f32::min) over both maps — token_score.rs:398-401
__device__ __forceinline__ void hybrid_min_max(const HybridCombine& h, float& mn, float& mx) {
    mx = 0.0f;
    mn = 0.0f;
    if (h.vec_max > mx) mx = h.vec_max;
    if (h.vec_min < mn) mn = h.vec_min;
    const uint32_t kmax = h.state->max_key, kmin = h.state->min_key;
    if (kmax != 0u) {
        const float v = ordered_to_f32(kmax);
        if (v > mx) mx = v;
    }
    if (kmin != 0xffffffffu) {
        const float v = ordered_to_f32(kmin);
        if (v < mn) mn = v;
    }
}

__global__ __launch_bounds__(kThreads) void hybrid_normalize_kernel(HybridCombine h) {
    float mn, mx;
    hybrid_min_max(h, mn, mx);
    const float den = mx - mn;
    const uint32_t n = h.state->list_len;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        if (h.cand_idx[i] != kNoDoc) h.cand_score[i] = (h.cand_score[i] - mn) / den;
}

__global__ __launch_bounds__(kThreads) void hybrid_add_vector_kernel(HybridCombine h) {
    float mn, mx;
    hybrid_min_max(h, mn, mx);
    const float den = mx - mn;
    for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < h.n_vec; j += gridDim.x * kThreads) {
        const uint32_t doc = h.vec_idx[j];
        const float v = (h.vec_score[j] - mn) / den;
        const unsigned long long e = h.emit[doc];
        if ((uint32_t)(e >> 32) == h.epoch) {
            const uint32_t pos = (uint32_t)e;
            h.cand_score[pos] = h.cand_score[pos] + v;  // *e += v
        } else {
            const uint32_t pos = atomicAdd(&h.state->list_len, 1u);  // <= limit appends per query
            atomicAdd(&h.state->cand_count, 1u);
            h.cand_score[pos] = 0.0f + v;  // entry(k).or_default() += v
            h.cand_idx[pos] = doc;
            h.emit[doc] = ((unsigned long long)h.epoch << 32) | (unsigned long long)pos;
        }
    }
}

__global__ __launch_bounds__(kThreads) void hybrid_ingest_kernel(uint32_t n, uint32_t epoch,
                                                                 const uint32_t* __restrict__ cand_idx,
                                                                 const float* __restrict__ cand_score,
                                                                 unsigned long long* __restrict__ emit,
                                                                 Bm25State* state) {
    uint32_t my_max = 0u, my_min = 0xffffffffu;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        emit[cand_idx[i]] = ((unsigned long long)epoch << 32) | (unsigned long long)i;
        const float s = cand_score[i];
        if (s == s) {
            const uint32_t key = f32_to_ordered(s);
            my_max = max(my_max, key);
            my_min = min(my_min, key);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        my_max = max(my_max, (uint32_t)__shfl_xor((int)my_max, off, 64));
        my_min = min(my_min, (uint32_t)__shfl_xor((int)my_min, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (my_max != 0u) atomicMax(&state->max_key, my_max);
        if (my_min != 0xffffffffu) atomicMin(&state->min_key, my_min);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state->cand_count = n;
        state->list_len = n;
    }
}

__global__ __launch_bounds__(kThreads) void omc_dense_kernel(const float* __restrict__ omc,
                                                             const Bm25State* __restrict__ state,
                                                             const uint32_t* __restrict__ cand_idx,
                                                             float* __restrict__ cand_score) {
    const uint32_t n = state->list_len;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        if (cand_idx[i] != kNoDoc) cand_score[i] = cand_score[i] * omc[cand_idx[i]];
}

__global__ __launch_bounds__(kThreads) void omc_sparse_kernel(const uint32_t* __restrict__ idx,
                                                              const float* __restrict__ mul, uint32_t n,
                                                              uint32_t epoch,
                                                              const unsigned long long* __restrict__ emit,
                                                              float* __restrict__ cand_score) {
    for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < n; j += gridDim.x * kThreads) {
        const unsigned long long e = emit[idx[j]];
        if ((uint32_t)(e >> 32) == epoch) {
            const uint32_t pos = (uint32_t)e;
            cand_score[pos] = cand_score[pos] * mul[j];
        }
    }
}

// ---- sharded-index exchange words.  The all-reduce MAX runs on i64 lanes:
//   lane 0 = ordered key of the largest full-text score (0 = none), lane 1 = 0xffffffff - smallest key.
__global__ void minmax_export_kernel(const Bm25State* __restrict__ st, long long* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = (long long)st->max_key;
        out[1] = (long long)(0xffffffffu - st->min_key);
    }
}
__global__ void minmax_import_kernel(Bm25State* __restrict__ st, const long long* __restrict__ in) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->max_key = (uint32_t)in[0];
        st->min_key = 0xffffffffu - (uint32_t)in[1];
    }
}
__global__ void df_export_kernel(const Bm25State* __restrict__ st, uint32_t n_tokens, int* __restrict__ out) {
    for (uint32_t t = threadIdx.x; t < n_tokens; t += blockDim.x) out[t] = (int)st->df[t];
}
__global__ void count_export_kernel(const Bm25State* __restrict__ st, unsigned long long* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)st->cand_count;
}
__global__ void count_sum_kernel(const char* __restrict__ blocks, uint64_t stride, uint64_t off, uint32_t lists,
                                 unsigned long long* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long c = 0;
        for (uint32_t l = 0; l < lists; ++l)
            c += *reinterpret_cast<const unsigned long long*>(blocks + (uint64_t)l * stride + off);
        out[0] = c;
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(kThreads) void synth_doc_len_kernel(uint16_t* __restrict__ len, uint64_t n_docs,
                                                                 uint64_t seed) {
    for (uint64_t d = (uint64_t)blockIdx.x * kThreads + threadIdx.x; d < n_docs;
         d += (uint64_t)gridDim.x * kThreads) {
        const uint64_t h = mix64(seed ^ (d * 0xD1342543DE82EF95ull));
        const float u1 = ((float)((uint32_t)h >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float u2 = ((float)((uint32_t)(h >> 32) >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
        float l = __expf(4.0f + 0.6f * z);
        l = fminf(fmaxf(rintf(l), 4.0f), 2000.0f);
        len[d] = (uint16_t)l;
    }
}

__global__ __launch_bounds__(kThreads) void synth_postings_kernel(uint32_t* __restrict__ post_doc,
                                                                  uint32_t* __restrict__ post_val,
                                                                  const uint64_t* __restrict__ list_off,
                                                                  uint32_t n_lists, uint64_t n_docs,
                                                                  const uint16_t* __restrict__ len, uint64_t seed,
                                                                  uint64_t total) {
    for (uint64_t p = (uint64_t)blockIdx.x * kThreads + threadIdx.x; p < total;
         p += (uint64_t)gridDim.x * kThreads) {
        uint32_t lo = 0, hi = n_lists;  // list_off[lo] <= p < list_off[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (list_off[mid] <= p) lo = mid; else hi = mid;
        }
        const uint64_t i = p - list_off[lo];
        const uint64_t df = list_off[lo + 1] - list_off[lo];
        // integer strata: doc in [i*N/df, (i+1)*N/df) — ascending and unique because df <= N
        const uint64_t base = (i * n_docs) / df, next = ((i + 1) * n_docs) / df;
        const uint64_t h = mix64(seed ^ ((uint64_t)lo * 0x2545F4914F6CDD1Dull) ^ (i * 0x9E3779B97F4A7C15ull));
        const uint64_t doc = base + (h >> 11) % (next - base);
        const uint32_t tf = ((h & 3u) == 0u) ? 2u + (uint32_t)((h >> 2) & 1u) : 1u;
        post_doc[p] = (uint32_t)doc;
        post_val[p] = (tf << 16) | (uint32_t)len[doc];
    }
}

uint32_t grid_for(uint64_t items, orama_ctx* ctx, uint32_t per_thread = 4) {
    uint64_t blocks = (items + (uint64_t)kThreads * per_thread - 1) / ((uint64_t)kThreads * per_thread);
    const uint64_t cap = (uint64_t)ctx->compute_units * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (uint32_t)blocks;
}

}  // namespace

int launch_bm25_accumulate(orama_ctx* ctx, const Bm25Accum& a, hipStream_t stream) {
    if (a.total == 0 || a.n_segs == 0) return ORAMA_OK;
    ORAMA_REQUIRE(a.n_segs <= kMaxTokens, "bm25: too many segments in one launch");
    ProfScope prof(&ctx->prof, "bm25_accumulate", stream);
    const dim3 grid(grid_for(a.total, ctx));
    if (a.precomputed)
        hipLaunchKernelGGL(bm25_accumulate_kernel<true>, grid, dim3(kThreads), 0, stream, a);
    else
        hipLaunchKernelGGL(bm25_accumulate_kernel<false>, grid, dim3(kThreads), 0, stream, a);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_bm25_finalize(orama_ctx* ctx, const Bm25Finalize& f, hipStream_t stream) {
    ORAMA_REQUIRE(f.n_tokens >= 1, "bm25: no tokens");
    ORAMA_SUPPORT(f.n_tokens <= kMaxTokens, "bm25: n_tokens %u outside [1, %u]",
                  f.n_tokens, kMaxTokens);
    ORAMA_REQUIRE(f.idf_vals, "bm25: idf values missing");
    ProfScope prof(&ctx->prof, "bm25_finalize", stream);
    dim3 grid(grid_for(f.n_slots ? f.n_slots : 1, ctx, 1));
    if (grid.x > (uint32_t)ctx->compute_units * 4u) grid.x = (uint32_t)ctx->compute_units * 4u;
    if (f.track_minmax)
        hipLaunchKernelGGL(bm25_finalize_kernel<true>, grid, dim3(kThreads), 0, stream, f);
    else
        hipLaunchKernelGGL(bm25_finalize_kernel<false>, grid, dim3(kThreads), 0, stream, f);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_hybrid_ingest(orama_ctx* ctx, uint32_t n, uint32_t epoch, const uint32_t* cand_idx,
                         const float* cand_score, unsigned long long* emit, Bm25State* state,
                         hipStream_t stream) {
    hipLaunchKernelGGL(hybrid_ingest_kernel, dim3(grid_for(n ? n : 1, ctx, 1)), dim3(kThreads), 0, stream, n,
                       epoch, cand_idx, cand_score, emit, state);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_hybrid_combine(orama_ctx* ctx, const HybridCombine& h, hipStream_t stream) {
    hipLaunchKernelGGL(hybrid_normalize_kernel, dim3(grid_for(h.cand_cap ? h.cand_cap : 1, ctx, 1)),
                       dim3(kThreads), 0, stream, h);
    if (h.n_vec)
        hipLaunchKernelGGL(hybrid_add_vector_kernel, dim3(grid_for(h.n_vec, ctx, 1)), dim3(kThreads), 0,
                           stream, h);
    if (h.omc_dense)
        hipLaunchKernelGGL(omc_dense_kernel, dim3(grid_for(h.cand_cap + h.n_vec, ctx, 1)), dim3(kThreads),
                           0, stream, h.omc_dense, h.state, h.cand_idx, h.cand_score);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_minmax_export(const Bm25State* st, long long* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(minmax_export_kernel, dim3(1), dim3(64), 0, stream, st, d_out);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
int launch_minmax_import(Bm25State* st, const long long* d_in, hipStream_t stream) {
    hipLaunchKernelGGL(minmax_import_kernel, dim3(1), dim3(64), 0, stream, st, d_in);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
int launch_df_export(const Bm25State* st, uint32_t n_tokens, int* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(df_export_kernel, dim3(1), dim3(64), 0, stream, st, n_tokens, d_out);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
int launch_count_export(const Bm25State* st, unsigned long long* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(count_export_kernel, dim3(1), dim3(64), 0, stream, st, d_out);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}
int launch_count_sum(const void* d_blocks, uint64_t stride, uint64_t off, uint32_t lists, unsigned long long* d_out,
                     hipStream_t stream) {
    hipLaunchKernelGGL(count_sum_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const char*>(d_blocks), stride,
                       off, lists, d_out);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_synth_doc_len(uint16_t* d_len, uint64_t n_docs, uint64_t seed, hipStream_t stream) {
    if (n_docs == 0) return ORAMA_OK;
    uint64_t blocks = (n_docs + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(synth_doc_len_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, stream, d_len, n_docs, seed);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_synth_postings(uint32_t* post_doc, uint32_t* post_val, const uint64_t* d_list_off, uint32_t n_lists,
                          uint64_t n_docs, const uint16_t* d_len, uint64_t seed, uint64_t total,
                          hipStream_t stream) {
    if (total == 0) return ORAMA_OK;
    uint64_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(synth_postings_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, stream, post_doc, post_val,
                       d_list_off, n_lists, n_docs, d_len, seed, total);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_omc_sparse(const uint32_t* d_idx, const float* d_mul, uint32_t n, uint32_t epoch,
                      const unsigned long long* emit, float* cand_score, hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    uint32_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(omc_sparse_kernel, dim3(blocks), dim3(kThreads), 0, stream, d_idx, d_mul, n, epoch,
                       emit, cand_score);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
