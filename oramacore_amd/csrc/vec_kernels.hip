// vec_kernels.hip — K1: brute-force distance scan over the HBM-resident embedding matrix.
//
// Mapping (gfx950): one 64-lane wave owns a row at a time; lane l loads the 16-byte pieces
// l, l+64, l+128 … of the row, so every wave-wide load instruction fetches one contiguous 1 KiB
// run of the row (full 128-B lines, no re-reads).  The query never goes through LDS: with this
// mapping lane l only ever needs query elements 4l…4l+3 (+256·c), so they sit in 4·NCHUNK VGPRs
// for the whole kernel.  Each wave keeps ROWS rows (ROWS·NCHUNK independent 16-B loads per lane)
// in flight; rows are dealt to waves round-robin so that the chip reads one contiguous window of
// the matrix at any moment.  Per row: 4·NCHUNK FMAs/lane, a DPP wave reduction, one scalar
// epilogue → VALU load is a few % of the HBM time; the kernel is bound by HBM bandwidth
// (algorithmic bytes = n · dim · 4 per pass; distances written back are 4 B/row, < 0.2 %).
#include <hip/hip_ext.h>

#include "vec_kernels.hpp"

#include <cstdlib>

#include "device_utils.hpp"

namespace orama {

typedef float f32x4 __attribute__((ext_vector_type(4)));

bool scan_tuning_valid(const ScanTuning& t) {
    const int r = t.rows_per_wave;
    return (r == 1 || r == 2 || r == 4 || r == 8) && t.blocks_per_cu >= 1 && t.blocks_per_cu <= 32;
}

ScanTuning default_scan_tuning() {
    ScanTuning x;
    if (const char* e = orama::dev_env("ORAMA_SCAN_ROWS")) x.rows_per_wave = std::atoi(e);
    if (const char* e = orama::dev_env("ORAMA_SCAN_BLOCKS_PER_CU")) x.blocks_per_cu = std::atoi(e);
    if (const char* e = orama::dev_env("ORAMA_SCAN_NT")) x.nontemporal = std::atoi(e);
    if (!scan_tuning_valid(x)) x = ScanTuning();
    return x;
}

namespace {

constexpr int kScanThreads = 256;
constexpr int kWavesPerBlock = kScanThreads / kWave;

template <bool NT>
__device__ __forceinline__ f32x4 load16(const f32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    return *p;
}

__device__ __forceinline__ bool row_excluded(uint64_t row, const uint32_t* dead,
                                             const uint64_t* row_doc, const uint64_t* allow,
                                             uint64_t allow_bits) {
    if (dead && ((dead[row >> 5] >> (row & 31)) & 1u)) return true;
    if (allow) {
        uint64_t doc = row_doc[row];
        if (doc >= allow_bits) return true;
        if (!((allow[doc >> 6] >> (doc & 63)) & 1ull)) return true;
    }
    return false;
}

// NCHUNK = ceil(dim/4 / 64) 16-byte pieces per lane per row; EXACT: dim/4 == 64*NCHUNK.
template <int NCHUNK, bool EXACT, int ROWS, bool NT, int METRIC, bool FUSED>
__global__ __launch_bounds__(kScanThreads) void vec_scan_f32_kernel(ScanArgs a) {
#include "vec_scan_f32_body.inc"
}

// The same pass for a set of queries chosen ON THE DEVICE (the two-stage plan's fallback on the device path, vec_store.hip):
// queries pick[0 .. *n_pick) of a.query, one after the other, each with its own wave lists; *n_pick == 0 — the usual case —
// ends the launch at once.  Same arithmetic per row as the kernel above (the body is the same text), cosine and fused mode only.
template <int NCHUNK, bool EXACT, int ROWS>
__global__ __launch_bounds__(kScanThreads) void vec_scan_f32_picked_kernel(ScanArgs a, const uint32_t* __restrict__ pick,
                                                                            const uint32_t* __restrict__ n_pick, uint64_t list_stride) {
    const uint32_t np = uniform_u32(*n_pick);
    const float* queries = a.query;
    unsigned long long* lists = a.wave_lists;
    constexpr bool NT = false, FUSED = true;
    constexpr int METRIC = ORAMA_METRIC_COSINE;
    for (uint32_t f = 0; f < np; ++f) {
        a.query = queries + (size_t)uniform_u32(pick[f]) * a.dim;
        a.wave_lists = lists + (uint64_t)f * list_stride;
        {
#include "vec_scan_f32_body.inc"
        }
    }
}

// Any dim (also dim % 4 != 0): scalar loads, one row per wave iteration.
template <int METRIC>
__global__ __launch_bounds__(kScanThreads) void vec_scan_f32_generic_kernel(ScanArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform_u32(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    float qq = 0.0f;
    for (uint32_t c = lane; c < a.dim; c += kWave) qq = fmaf(a.query[c], a.query[c], qq);
    float qscale = 0.0f;
    if (METRIC == ORAMA_METRIC_COSINE) {
        qq = wave_sum(qq);
        qscale = qq > 0.0f ? 1.0f / sqrtf(qq) : 0.0f;
    }
    const bool fused = a.wave_lists != nullptr;
    WaveTopK best;
    for (uint64_t row = wave; row < a.n; row += nwaves) {
        const bool live = !row_excluded(row, a.dead, a.row_doc, a.allow, a.allow_bits);
        float acc = 0.0f;
        if (live) {
            const float* p = a.corpus + row * (uint64_t)a.dim;
            for (uint32_t c = lane; c < a.dim; c += kWave) {
                if (METRIC == ORAMA_METRIC_COSINE) {
                    acc = fmaf(p[c], a.query[c], acc);
                } else {
                    float t = p[c] - a.query[c];
                    acc = fmaf(t, t, acc);
                }
            }
        }
        const float tot = wave_sum(acc);
        float dist = METRIC == ORAMA_METRIC_COSINE ? 1.0f - tot * (a.inv_norm[row] * qscale) : tot;
        if (!live) dist = __builtin_nanf("");
        if (fused) {
            if (live && dist == dist) {
                const unsigned long long key =
                    ((unsigned long long)(~f32_to_ordered(dist)) << 32) | (unsigned long long)(uint32_t)(~(uint32_t)row);
                if (best.count < a.topk || key > best.thr) best.insert(key, a.topk, lane);
            }
        } else if (lane == 0) {
            a.out_dist[row] = dist;
        }
    }
    if (fused) {
        unsigned long long* out = a.wave_lists + (uint64_t)wave * kWaveListKeys;
        out[lane] = best.s0;
        out[lane + 64] = best.s1;
    }
}

// `ev`: the profiler's event pair for THIS dispatch (ProfLaunch; both null when the profiler is off).  The events ride on the
// kernel's own dispatch packet (hipExtLaunchKernelGGL) instead of bracketing it with two recorded events: no barrier packets
// between consecutive scans of a pipelined session — 13 us of a 4.33 ms step with the profiler on (scripts/ns_step_gap_probe.py).
template <int NCHUNK, bool EXACT, int ROWS, bool NT>
void launch_scan_metric(const ScanArgs& a, dim3 grid, hipStream_t s, const ProfLaunch& ev) {
    const bool fused = a.wave_lists != nullptr;
    if (a.metric == ORAMA_METRIC_COSINE) {
        if (fused)
            hipExtLaunchKernelGGL((vec_scan_f32_kernel<NCHUNK, EXACT, ROWS, NT, ORAMA_METRIC_COSINE, true>), grid,
                                  dim3(kScanThreads), 0, s, ev.start, ev.stop, 0, a);
        else
            hipExtLaunchKernelGGL((vec_scan_f32_kernel<NCHUNK, EXACT, ROWS, NT, ORAMA_METRIC_COSINE, false>), grid,
                                  dim3(kScanThreads), 0, s, ev.start, ev.stop, 0, a);
    } else {
        if (fused)
            hipExtLaunchKernelGGL((vec_scan_f32_kernel<NCHUNK, EXACT, ROWS, NT, ORAMA_METRIC_L2SQ, true>), grid,
                                  dim3(kScanThreads), 0, s, ev.start, ev.stop, 0, a);
        else
            hipExtLaunchKernelGGL((vec_scan_f32_kernel<NCHUNK, EXACT, ROWS, NT, ORAMA_METRIC_L2SQ, false>), grid,
                                  dim3(kScanThreads), 0, s, ev.start, ev.stop, 0, a);
    }
}

template <int NCHUNK, bool EXACT>
void launch_scan_rows(const ScanArgs& a, const ScanTuning& t, dim3 grid, hipStream_t s, const ProfLaunch& ev) {
    const bool nt = t.nontemporal != 0;
    // register budget: ROWS * NCHUNK * 4 data VGPRs — cap ROWS for wide rows
    int rows = t.rows_per_wave;
    while (rows * NCHUNK > 16) rows >>= 1;
    switch (rows) {
        case 8:
            nt ? launch_scan_metric<NCHUNK, EXACT, 8, true>(a, grid, s, ev)
               : launch_scan_metric<NCHUNK, EXACT, 8, false>(a, grid, s, ev);
            break;
        case 4:
            nt ? launch_scan_metric<NCHUNK, EXACT, 4, true>(a, grid, s, ev)
               : launch_scan_metric<NCHUNK, EXACT, 4, false>(a, grid, s, ev);
            break;
        case 2:
            nt ? launch_scan_metric<NCHUNK, EXACT, 2, true>(a, grid, s, ev)
               : launch_scan_metric<NCHUNK, EXACT, 2, false>(a, grid, s, ev);
            break;
        default:
            nt ? launch_scan_metric<NCHUNK, EXACT, 1, true>(a, grid, s, ev)
               : launch_scan_metric<NCHUNK, EXACT, 1, false>(a, grid, s, ev);
            break;
    }
}

// ---------------------------------------------------------------- row norms
__global__ __launch_bounds__(kScanThreads) void row_inv_norm_kernel(const float* __restrict__ corpus,
                                                                    uint64_t first, uint64_t n,
                                                                    uint32_t dim,
                                                                    float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const float* p = corpus + (first + i) * (uint64_t)dim;
        float ss = 0.0f;
        for (uint32_t c = lane; c < dim; c += kWave) ss = fmaf(p[c], p[c], ss);
        ss = wave_sum(ss);
        if (lane == 0) inv_norm[first + i] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
    }
}

// ---------------------------------------------------------------- synthetic data
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint32_t bits) {  // (0, 1]
    return ((float)(bits >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(kScanThreads) void synth_fill_kernel(float* __restrict__ corpus,
                                                                  uint64_t first, uint64_t n,
                                                                  uint32_t dim, uint64_t seed, bool unit_norm) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const uint64_t row = first + i;
        float* p = corpus + row * (uint64_t)dim;
        float ss = 0.0f;
        // pass 1: norm of the gaussian row (values are regenerated in pass 2 — pure ALU, no HBM)
        for (uint32_t c = lane * 2; c < dim; c += 2 * kWave) {
            uint64_t h = splitmix64(seed ^ (row * 0xD1342543DE82EF95ull + (uint64_t)(c >> 1)));
            float r = sqrtf(-2.0f * __logf(u01((uint32_t)h)));
            float th = 6.28318530718f * u01((uint32_t)(h >> 32));
            float g0 = r * __cosf(th), g1 = r * __sinf(th);
            if (seed == ~0ull) g0 = g1 = 1.0f;  // diagnostic corpus: constant rows (minimal operand toggling)
            ss = fmaf(g0, g0, ss);
            if (c + 1 < dim) ss = fmaf(g1, g1, ss);
        }
        ss = wave_sum(ss);
        uint64_t hs = splitmix64(seed ^ 0xA5A5A5A55A5A5A5Aull ^ (row * 0x2545F4914F6CDD1Dull));
        // (ORAMA_SYNTH_UNIT_NORM=1: rows of norm 1 — the corpus of the unit-norm store experiment, scripts/k2q_unit_norm_probe.py;
        // SURVEY §8d's workloads re-scale every row by u ~ U(0.5, 2) so that the norms matter)
        const float u = unit_norm ? 1.0f : 0.5f + 1.5f * u01((uint32_t)hs);
        const float scale = ss > 0.0f ? u / sqrtf(ss) : 0.0f;
        for (uint32_t c = lane * 2; c < dim; c += 2 * kWave) {
            uint64_t h = splitmix64(seed ^ (row * 0xD1342543DE82EF95ull + (uint64_t)(c >> 1)));
            float r = sqrtf(-2.0f * __logf(u01((uint32_t)h)));
            float th = 6.28318530718f * u01((uint32_t)(h >> 32));
            float g0 = r * __cosf(th), g1 = r * __sinf(th);
            if (seed == ~0ull) g0 = g1 = 1.0f;
            p[c] = g0 * scale;
            if (c + 1 < dim) p[c + 1] = g1 * scale;
        }
    }
}

__global__ __launch_bounds__(kScanThreads) void gather_rows_kernel(const float* __restrict__ corpus,
                                                                   const uint64_t* __restrict__ idx,
                                                                   uint64_t n, uint32_t dim,
                                                                   float* __restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const float* p = corpus + idx[i] * (uint64_t)dim;
        float* o = out + i * (uint64_t)dim;
        for (uint32_t c = lane; c < dim; c += kWave) o[c] = p[c];
    }
}

__global__ void iota_u64_kernel(uint64_t* ids, uint64_t n, uint64_t first) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        ids[i] = first + i;
}

uint32_t grid_for_rows(uint64_t n_rows_per_wave_units, uint32_t cap) {
    uint64_t blocks = (n_rows_per_wave_units + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    return (uint32_t)blocks;
}

}  // namespace

namespace {
// ---------------------------------------------------------------- exact re-scoring of candidate rows (two-stage search)
// One wave per (query, candidate): the distance K1's vectorised kernel computes for that (row, query) — the same
// lane -> column mapping (lane l holds the 16-byte pieces l, l + 64, ...), the same per-lane fma chain, the same wave_sum
// tree, the same final expression — so the value is bit-identical to a K1 scan of that row (tests compare them).
// Cosine only, dim % 4 == 0 and dim <= 1024 (the K1 geometry this mirrors).
__global__ __launch_bounds__(kScanThreads) void rerank_f32_kernel(const float* __restrict__ corpus,
                                                                  const float* __restrict__ inv_norm, uint32_t dim,
                                                                  const float* __restrict__ queries, uint32_t q,
                                                                  const uint32_t* __restrict__ cand_rows,
                                                                  const uint32_t* __restrict__ cand_n, uint32_t stride,
                                                                  float* __restrict__ out_dist) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const uint32_t qi = (uint32_t)(wave / stride), ci = (uint32_t)(wave % stride);
    if (qi >= q) return;
    const uint32_t d4 = dim >> 2;
    const uint32_t nchunk = (d4 + kWave - 1) / kWave;
    if (ci >= cand_n[qi]) {
        if (lane == 0) out_dist[(uint64_t)qi * stride + ci] = __builtin_nanf("");
        return;
    }
    const uint64_t row = cand_rows[(uint64_t)qi * stride + ci];
    const f32x4* __restrict__ qp = reinterpret_cast<const f32x4*>(queries + (uint64_t)qi * dim);
    const f32x4* __restrict__ xp = reinterpret_cast<const f32x4*>(corpus) + row * d4;
    float qq = 0.0f, acc = 0.0f;
    f32x4 qv[4], xv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t f = (uint32_t)c * kWave + lane;
        const bool ok = (uint32_t)c < nchunk && f < d4;
        qv[c] = ok ? qp[f] : f32x4{0.f, 0.f, 0.f, 0.f};
        xv[c] = ok ? xp[f] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if ((uint32_t)c >= nchunk) break;
        qq = fmaf(qv[c].x, qv[c].x, qq);
        qq = fmaf(qv[c].y, qv[c].y, qq);
        qq = fmaf(qv[c].z, qv[c].z, qq);
        qq = fmaf(qv[c].w, qv[c].w, qq);
    }
    qq = wave_sum(qq);
    const float qscale = qq > 0.0f ? 1.0f / sqrtf(qq) : 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if ((uint32_t)c >= nchunk) break;
        acc = fmaf(xv[c].x, qv[c].x, acc);
        acc = fmaf(xv[c].y, qv[c].y, acc);
        acc = fmaf(xv[c].z, qv[c].z, acc);
        acc = fmaf(xv[c].w, qv[c].w, acc);
    }
    const float tot = wave_sum(acc);
    const float inv = inv_norm[row];
    const float dist = 1.0f - tot * (inv * qscale);
    if (lane == 0) out_dist[(uint64_t)qi * stride + ci] = dist;
}

// flag[j] = 1 when the candidate list of query j may miss a row of the exact top-k: the list is full (n == k1) and its
// last shadow distance lies within `band` of the k-th one.  shadow_dist: ascending per query.
__global__ void shadow_band_kernel(const float* __restrict__ shadow_dist, const uint32_t* __restrict__ n, uint32_t q,
                                   uint32_t k, uint32_t k1, float band, uint32_t* __restrict__ flag,
                                   const uint32_t* __restrict__ also) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= q) return;
    uint32_t f = 0;
    if (n[j] >= k1 && k1 > 0) {
        const float tau = shadow_dist[(uint64_t)j * k1 + (k - 1)];
        const float last = shadow_dist[(uint64_t)j * k1 + (k1 - 1)];
        f = !(last > tau + band);  // also when either is NaN
    }
    if (also && also[j]) f = 1;  // the candidate stage flagged its own list
    flag[j] = f;
}
}  // namespace

bool vec_rerank_f32_supported(uint32_t dim) { return (dim & 3) == 0 && (dim >> 2) <= 4 * kWave; }

int launch_rerank_f32(const float* corpus, const float* inv_norm, uint32_t dim, const float* d_queries, uint32_t q,
                      const uint32_t* d_cand_rows, const uint32_t* d_cand_n, uint32_t stride, float* d_out_dist,
                      hipStream_t stream) {
    ORAMA_REQUIRE(vec_rerank_f32_supported(dim), "rerank: dim %u not on the vectorised K1 path", dim);
    if (q == 0 || stride == 0) return ORAMA_OK;
    const uint64_t waves = (uint64_t)q * stride;
    const uint64_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(rerank_f32_kernel, dim3((uint32_t)blocks), dim3(kScanThreads), 0, stream, corpus, inv_norm, dim, d_queries,
                       q, d_cand_rows, d_cand_n, stride, d_out_dist);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_shadow_band(const float* d_shadow_dist, const uint32_t* d_n, uint32_t q, uint32_t k, uint32_t k1, float band,
                       uint32_t* d_flag, hipStream_t stream, const uint32_t* d_also) {
    hipLaunchKernelGGL(shadow_band_kernel, dim3((q + 63) / 64), dim3(64), 0, stream, d_shadow_dist, d_n, q, k, k1, band, d_flag,
                       d_also);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

namespace {
// geometry shared by the launcher and vec_scan_f32_waves()
struct ScanGeom {
    bool vec4;
    int nchunk;
    bool exact;
    int rows;
    uint32_t blocks;
};
ScanGeom scan_geom(orama_ctx* ctx, const ScanArgs& a) {
    const ScanTuning& t = ctx->scan_tuning;
    const uint32_t cap = (uint32_t)ctx->compute_units * (uint32_t)t.blocks_per_cu;
    const uint32_t d4 = a.dim >> 2;
    ScanGeom g{};
    g.vec4 = (a.dim & 3) == 0 && d4 <= 4 * kWave;
    if (g.vec4) {
        g.nchunk = (int)((d4 + kWave - 1) / kWave);
        g.exact = (d4 == (uint32_t)g.nchunk * kWave);
        g.rows = t.rows_per_wave;
        while (g.rows * g.nchunk > 16) g.rows >>= 1;
        g.blocks = grid_for_rows((a.n + g.rows - 1) / g.rows, cap);
    } else {
        g.rows = 1;
        g.blocks = grid_for_rows(a.n, cap);
    }
    return g;
}
}  // namespace

uint32_t vec_scan_f32_waves(orama_ctx* ctx, const ScanArgs& a) {
    return a.n ? scan_geom(ctx, a).blocks * kWavesPerBlock : 0;
}

// `done` / `attached`: the two-stream mode's "this scan has finished" event without a record packet behind the kernel —
// `done` (or the profiler's stop event when the profiler is on) rides on the dispatch, and *attached tells the caller which
// event to wait for (nullptr: none rode along — the generic kernel, or ORAMA_SCAN_DONE_EVENT=record — record one yourself).
int launch_vec_scan_f32(orama_ctx* ctx, const ScanArgs& a_in, hipStream_t stream, hipEvent_t done, hipEvent_t* attached) {
    if (attached) *attached = nullptr;
    const ScanArgs& a = a_in;
    ORAMA_REQUIRE(a.corpus && a.query && a.dim > 0, "vec_scan: bad arguments");
    ORAMA_REQUIRE(a.out_dist || (a.wave_lists && a.topk >= 1 && a.topk <= kWaveListKeys), "vec_scan: no output mode");
    ORAMA_REQUIRE(a.metric != ORAMA_METRIC_COSINE || a.inv_norm, "vec_scan: cosine needs inv_norm");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan: filter needs row_doc");
    ORAMA_REQUIRE(a.n < 0xffffffffull, "vec_scan: too many rows");
    if (a.n == 0) return ORAMA_OK;
    const ScanTuning& t = ctx->scan_tuning;
    const ScanGeom g = scan_geom(ctx, a);
    const dim3 grid(g.blocks);
    ProfLaunch ev(g.vec4 ? &ctx->prof : nullptr, "vec_scan_f32");            // events on the dispatch itself
    if (g.vec4 && attached && ctx->scan_done_on_dispatch) {
        if (!ev.stop) ev.stop = done;
        *attached = ev.stop;
    }
    ProfScope prof(g.vec4 ? nullptr : &ctx->prof, "vec_scan_f32", stream);  // (the generic kernel: bracketed as before)
    if (g.vec4) {
        switch (g.nchunk) {
            case 1:
                g.exact ? launch_scan_rows<1, true>(a, t, grid, stream, ev)
                        : launch_scan_rows<1, false>(a, t, grid, stream, ev);
                break;
            case 2:
                g.exact ? launch_scan_rows<2, true>(a, t, grid, stream, ev)
                        : launch_scan_rows<2, false>(a, t, grid, stream, ev);
                break;
            case 3:
                g.exact ? launch_scan_rows<3, true>(a, t, grid, stream, ev)
                        : launch_scan_rows<3, false>(a, t, grid, stream, ev);
                break;
            default:
                g.exact ? launch_scan_rows<4, true>(a, t, grid, stream, ev)
                        : launch_scan_rows<4, false>(a, t, grid, stream, ev);
                break;
        }
    } else {
        if (a.metric == ORAMA_METRIC_COSINE)
            hipLaunchKernelGGL((vec_scan_f32_generic_kernel<ORAMA_METRIC_COSINE>), grid, dim3(kScanThreads), 0, stream, a);
        else
            hipLaunchKernelGGL((vec_scan_f32_generic_kernel<ORAMA_METRIC_L2SQ>), grid, dim3(kScanThreads), 0, stream, a);
    }
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

namespace {
// picked scans: 4 rows in flight per wave whatever the tuning says (the fallback's speed matters little, its instantiations
// do; 4 rows x 4 chunks x 4 registers is the widest the register budget of K1 takes)
constexpr int kPickedRows = 4;
template <int NCHUNK, bool EXACT>
void launch_picked(const ScanArgs& a, dim3 grid, const uint32_t* pick, const uint32_t* n_pick, uint64_t stride, hipStream_t s) {
    static_assert(kPickedRows * NCHUNK <= 16, "register budget of the scan body");
    hipLaunchKernelGGL((vec_scan_f32_picked_kernel<NCHUNK, EXACT, kPickedRows>), grid, dim3(kScanThreads), 0, s, a, pick, n_pick, stride);
}
}  // namespace

bool vec_scan_f32_picked_supported(const ScanArgs& a) {
    const uint32_t d4 = a.dim >> 2;
    return (a.dim & 3) == 0 && d4 <= 4 * kWave && a.metric == ORAMA_METRIC_COSINE;
}

uint32_t vec_scan_f32_picked_waves(orama_ctx* ctx, const ScanArgs& a) {
    if (!a.n) return 0;
    const uint32_t cap = (uint32_t)ctx->compute_units * (uint32_t)ctx->scan_tuning.blocks_per_cu;
    return grid_for_rows((a.n + kPickedRows - 1) / kPickedRows, cap) * kWavesPerBlock;
}

int launch_vec_scan_f32_picked(orama_ctx* ctx, const ScanArgs& a, const uint32_t* d_pick, const uint32_t* d_n_pick,
                               uint64_t list_stride, hipStream_t stream) {
    ORAMA_REQUIRE(a.corpus && a.query && a.inv_norm && a.wave_lists && d_pick && d_n_pick && a.topk >= 1 && a.topk <= kWaveListKeys,
                  "picked scan: bad arguments");
    ORAMA_REQUIRE(vec_scan_f32_picked_supported(a), "picked scan: cosine over dimensions that are a multiple of 4 up to 1024");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan: filter needs row_doc");
    ORAMA_REQUIRE(a.n < 0xffffffffull, "vec_scan: too many rows");
    if (a.n == 0) return ORAMA_OK;
    const uint32_t d4 = a.dim >> 2;
    const int nchunk = (int)((d4 + kWave - 1) / kWave);
    const bool exact = d4 == (uint32_t)nchunk * kWave;
    const dim3 grid(vec_scan_f32_picked_waves(ctx, a) / kWavesPerBlock);
    switch (nchunk) {
        case 1: exact ? launch_picked<1, true>(a, grid, d_pick, d_n_pick, list_stride, stream) : launch_picked<1, false>(a, grid, d_pick, d_n_pick, list_stride, stream); break;
        case 2: exact ? launch_picked<2, true>(a, grid, d_pick, d_n_pick, list_stride, stream) : launch_picked<2, false>(a, grid, d_pick, d_n_pick, list_stride, stream); break;
        case 3: exact ? launch_picked<3, true>(a, grid, d_pick, d_n_pick, list_stride, stream) : launch_picked<3, false>(a, grid, d_pick, d_n_pick, list_stride, stream); break;
        default: exact ? launch_picked<4, true>(a, grid, d_pick, d_n_pick, list_stride, stream) : launch_picked<4, false>(a, grid, d_pick, d_n_pick, list_stride, stream); break;
    }
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_row_inv_norm_f32(const float* corpus, uint64_t first, uint64_t n, uint32_t dim,
                            float* inv_norm, hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3(grid_for_rows(n, 4096)), dim3(kScanThreads), 0,
                       stream, corpus, first, n, dim, inv_norm);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_synth_fill_f32(float* corpus, uint64_t first, uint64_t n, uint32_t dim, uint64_t seed,
                          hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    static const bool unit_norm = [] { const char* e = orama::dev_env("ORAMA_SYNTH_UNIT_NORM"); return e && std::atoi(e) != 0; }();
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid_for_rows(n, 8192)), dim3(kScanThreads), 0, stream,
                       corpus, first, n, dim, seed, unit_norm);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_gather_rows_f32(const float* corpus, const uint64_t* d_row_idx, uint64_t n, uint32_t dim,
                           float* d_out, hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for_rows(n, 4096)), dim3(kScanThreads), 0,
                       stream, corpus, d_row_idx, n, dim, d_out);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_iota_u64(uint64_t* d_ids, uint64_t n, uint64_t first, hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iota_u64_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_ids, n, first);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
