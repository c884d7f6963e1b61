// vec_multi.hip — K1b: the fp32 scan for a small query batch (2..8 queries per corpus pass).
//
// Same row mapping as K1 (vec_kernels.hip): a wave owns ROWS rows at a time, lane l holds the 16-byte pieces
// l, l+64, … of each row.  The difference is only that QB queries sit in registers (4·NCHUNK·QB VGPRs) and every
// loaded row is multiplied against all of them before the next rows arrive — HBM traffic per pass is unchanged
// (n · dim · 4 B), so a batch of QB concurrent requests (orama_batcher) costs one pass instead of QB.
// Per (row, query) the arithmetic is K1's (same FMA order per lane, the same summation tree over the lanes, the
// same epilogue), so a batched answer is bit-identical to the solo answer.  The wave reduction is the VALU cost
// that matters (K1's wave_sum per (row, query) made 8 queries VALU-bound: 6.2 ms vs 4.6 ms for the bytes); the
// transposed reduction below brings 8 queries down to ~130 instructions per row.
#include "vec_kernels.hpp"

#include "device_utils.hpp"

namespace orama {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kScanThreads = 256;
constexpr int kWavesPerBlock = kScanThreads / kWave;

__device__ __forceinline__ bool row_excluded(uint64_t row, const uint32_t* dead, const uint64_t* row_doc,
                                             const uint64_t* allow, uint64_t allow_bits) {
    if (dead && ((dead[row >> 5] >> (row & 31)) & 1u)) return true;
    if (allow) {
        uint64_t doc = row_doc[row];
        if (doc >= allow_bits) return true;
        if (!((allow[doc >> 6] >> (doc & 63)) & 1ull)) return true;
    }
    return false;
}

template <int NCHUNK, bool EXACT, int ROWS, int METRIC, int QB>
__global__ __launch_bounds__(kScanThreads) void vec_scan_f32_multi_kernel(ScanArgs a, uint32_t nq,
                                                                          uint64_t out_stride) {
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uniform_u32(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * kWavesPerBlock;
    const uint32_t d4 = a.dim >> 2;
    const f32x4* __restrict__ base = reinterpret_cast<const f32x4*>(a.corpus);

    f32x4 qv[QB][NCHUNK];
    float qscale[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        float qq = 0.0f;
        const f32x4* qp = reinterpret_cast<const f32x4*>(a.query + (size_t)((uint32_t)j < nq ? j : 0) * a.dim);
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            const uint32_t f = c * kWave + lane;
            if (EXACT || f < d4) {
                qv[j][c] = qp[f];
            } else {
                qv[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            qq = fmaf(qv[j][c].x, qv[j][c].x, qq);
            qq = fmaf(qv[j][c].y, qv[j][c].y, qq);
            qq = fmaf(qv[j][c].z, qv[j][c].z, qq);
            qq = fmaf(qv[j][c].w, qv[j][c].w, qq);
        }
        qscale[j] = 0.0f;
        if (METRIC == ORAMA_METRIC_COSINE) {
            qq = wave_sum(qq);
            qscale[j] = qq > 0.0f ? 1.0f / sqrtf(qq) : 0.0f;
        }
    }
    const bool filtered = (a.dead != nullptr) || (a.allow != nullptr);
    // the query whose totals land in this lane after the transposed reduction, and its 1/|q|
    const int my_q = lane_query<QB>(lane);
    float my_qscale = 0.0f;
#pragma unroll
    for (int j = 0; j < QB; ++j)
        if (my_q == j) my_qscale = qscale[j];

    for (uint64_t r0 = (uint64_t)wave * ROWS; r0 < a.n; r0 += (uint64_t)nwaves * ROWS) {
        f32x4 x[ROWS][NCHUNK];
        bool live[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint64_t row = r0 + r;
            live[r] = row < a.n;
            if (filtered && live[r]) live[r] = !row_excluded(row, a.dead, a.row_doc, a.allow, a.allow_bits);
            const f32x4* p = base + row * d4 + lane;
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                const bool ok = live[r] && (EXACT || (uint32_t)(c * kWave + lane) < d4);
                x[r][c] = ok ? __builtin_nontemporal_load(p + c * kWave) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        float inv[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            inv[r] = (METRIC == ORAMA_METRIC_COSINE && live[r]) ? a.inv_norm[r0 + r] : 0.0f;
        // per lane: partial sums of every (row, query); then ONE transposed reduction per row
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float acc[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                acc[j] = 0.0f;
                if ((uint32_t)j < nq) {  // wave-uniform
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c) {
                        if (METRIC == ORAMA_METRIC_COSINE) {
                            acc[j] = fmaf(x[r][c].x, qv[j][c].x, acc[j]);
                            acc[j] = fmaf(x[r][c].y, qv[j][c].y, acc[j]);
                            acc[j] = fmaf(x[r][c].z, qv[j][c].z, acc[j]);
                            acc[j] = fmaf(x[r][c].w, qv[j][c].w, acc[j]);
                        } else {
                            float t0 = x[r][c].x - qv[j][c].x, t1 = x[r][c].y - qv[j][c].y;
                            float t2 = x[r][c].z - qv[j][c].z, t3 = x[r][c].w - qv[j][c].w;
                            acc[j] = fmaf(t0, t0, acc[j]);
                            acc[j] = fmaf(t1, t1, acc[j]);
                            acc[j] = fmaf(t2, t2, acc[j]);
                            acc[j] = fmaf(t3, t3, acc[j]);
                        }
                    }
                }
            }
            const float tot = wave_sum_scatter<QB>(acc, lane);  // lane l: total of query my_q
            float dist;
            if (METRIC == ORAMA_METRIC_COSINE) {
                dist = 1.0f - tot * (inv[r] * my_qscale);
            } else {
                dist = tot;
            }
            if (!live[r]) dist = __builtin_nanf("");
            if (lane < QB && (uint32_t)my_q < nq && r0 + r < a.n)
                a.out_dist[(uint64_t)my_q * out_stride + r0 + r] = dist;
        }
    }
}

template <int NCHUNK, bool EXACT, int QB>
void launch_multi(const ScanArgs& a, uint32_t nq, uint64_t out_stride, dim3 grid, hipStream_t s) {
    constexpr int ROWS = NCHUNK <= 2 ? 8 : (NCHUNK == 3 ? 4 : 2);
    if (a.metric == ORAMA_METRIC_COSINE)
        hipLaunchKernelGGL((vec_scan_f32_multi_kernel<NCHUNK, EXACT, ROWS, ORAMA_METRIC_COSINE, QB>), grid,
                           dim3(kScanThreads), 0, s, a, nq, out_stride);
    else
        hipLaunchKernelGGL((vec_scan_f32_multi_kernel<NCHUNK, EXACT, ROWS, ORAMA_METRIC_L2SQ, QB>), grid,
                           dim3(kScanThreads), 0, s, a, nq, out_stride);
}

template <int NCHUNK, bool EXACT>
void launch_multi_q(const ScanArgs& a, uint32_t nq, uint64_t out_stride, dim3 grid, hipStream_t s) {
    if (nq <= 4)
        launch_multi<NCHUNK, EXACT, 4>(a, nq, out_stride, grid, s);
    else
        launch_multi<NCHUNK, EXACT, 8>(a, nq, out_stride, grid, s);
}

}  // namespace

bool vec_scan_f32_multi_supported(const ScanArgs& a) {
    const uint32_t d4 = a.dim >> 2;
    return (a.dim & 3) == 0 && d4 <= 4 * kWave;
}

int launch_vec_scan_f32_multi(orama_ctx* ctx, const ScanArgs& a, uint32_t nq, uint64_t out_stride,
                              hipStream_t stream) {
    ORAMA_REQUIRE(a.corpus && a.query && a.dim > 0 && a.out_dist, "vec_scan_multi: bad arguments");
    ORAMA_REQUIRE(nq >= 2 && nq <= kScanMultiMaxQ, "vec_scan_multi: nq=%u outside [2, %u]", nq, kScanMultiMaxQ);
    ORAMA_REQUIRE(vec_scan_f32_multi_supported(a), "vec_scan_multi: dim %u not supported", a.dim);
    ORAMA_REQUIRE(a.metric != ORAMA_METRIC_COSINE || a.inv_norm, "vec_scan_multi: cosine needs inv_norm");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_multi: filter needs row_doc");
    ORAMA_REQUIRE(a.n < 0xffffffffull, "vec_scan_multi: too many rows");
    if (a.n == 0) return ORAMA_OK;
    ProfScope prof(&ctx->prof, "vec_scan_f32_multi", stream);
    const uint32_t d4 = a.dim >> 2;
    const int nchunk = (int)((d4 + kWave - 1) / kWave);
    const bool exact = d4 == (uint32_t)nchunk * kWave;
    const int rows = nchunk <= 2 ? 8 : (nchunk == 3 ? 4 : 2);
    uint64_t blocks = ((a.n + rows - 1) / rows + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint64_t cap = (uint64_t)ctx->compute_units * (uint64_t)ctx->scan_tuning.blocks_per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 grid((uint32_t)blocks);
    switch (nchunk) {
        case 1:
            exact ? launch_multi_q<1, true>(a, nq, out_stride, grid, stream)
                  : launch_multi_q<1, false>(a, nq, out_stride, grid, stream);
            break;
        case 2:
            exact ? launch_multi_q<2, true>(a, nq, out_stride, grid, stream)
                  : launch_multi_q<2, false>(a, nq, out_stride, grid, stream);
            break;
        case 3:
            exact ? launch_multi_q<3, true>(a, nq, out_stride, grid, stream)
                  : launch_multi_q<3, false>(a, nq, out_stride, grid, stream);
            break;
        default:
            exact ? launch_multi_q<4, true>(a, nq, out_stride, grid, stream)
                  : launch_multi_q<4, false>(a, nq, out_stride, grid, stream);
            break;
    }
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
