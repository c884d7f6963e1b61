// vec_f16_prep.hip — the query side of the wide fp16 scans (K2d vec_f16_pc.hip, K2q vec_f16_qs.hip and the comparison
// kernels K2c / K2h): the f32 queries of a batch rounded to fp16 and laid out as MFMA B fragments, + 1/|q| of the
// rounded query.  (Lived in vec_f16_wide.hip until round 4; K2c itself is no longer part of the product build.)
#include "vec_f16.hpp"

#include "device_utils.hpp"

namespace orama {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// queries (f32) -> fp16 B fragments [query tile 0..7][k-step][lane][8 halves] + 1/|q| of the rounded query
__global__ __launch_bounds__(256) void f16_prepare_queries_kernel(const float* __restrict__ queries, uint32_t q,
                                                                  uint32_t dim, uint32_t ksteps, bool l2,
                                                                  char* __restrict__ bfrag, float* __restrict__ qinv) {
    const uint32_t frag_total = 8u * ksteps * 64u;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < frag_total; idx += gridDim.x * blockDim.x) {
        const uint32_t qt = idx / (ksteps * 64u);
        const uint32_t rem = idx - qt * (ksteps * 64u);
        const uint32_t ks = rem >> 6, l = rem & 63;
        const uint32_t j = qt * 32 + (l & 31);
        const uint32_t k0 = ks * 16 + (l >> 5) * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            const float x = (j < q && k < dim) ? queries[(size_t)j * dim + k] : 0.0f;
            v[e] = (_Float16)x;
        }
        *reinterpret_cast<h8*>(bfrag + (size_t)idx * 16) = v;
    }
    // |q| of the fp16-rounded query, f32 accumulation in k order (the order K2 uses): ONE sequential fmaf chain per query —
    // the order is part of the result's bits — but the chain's inputs need not arrive one dependent global load at a time
    // (the first form: thread j of block 0 walked query j's row alone, 768 loads in a row = 0.15 ms in front of every
    // 256-query scan).  Wave w of block b owns query 4 b + w: its lanes stage the rounded row in LDS 1 024 values at a
    // time (coalesced), lane 0 runs the chain from LDS.
    __shared__ float row[4][1024];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t j = blockIdx.x * 4u + w;
    if (j < 256u) {
        float ss = 0.0f;
        if (j < q) {
            for (uint32_t k0 = 0; k0 < dim; k0 += 1024u) {
                const uint32_t n = min(1024u, dim - k0);
                for (uint32_t k = lane; k < n; k += 64u) row[w][k] = (float)(_Float16)queries[(size_t)j * dim + k0 + k];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed
                if (lane == 0)
                    for (uint32_t k = 0; k < n; ++k) ss = fmaf(row[w][k], row[w][k], ss);  // (x = 0 beyond dim: fmaf(0, 0, ss) == ss)
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) qinv[j] = l2 ? ss : (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f);
    }
}

}  // namespace

int launch_f16_prepare_queries(const float* d_queries, uint32_t q, uint32_t dim, int metric, void* d_query_frags,
                               hipStream_t stream) {
    ORAMA_REQUIRE(d_queries && d_query_frags && q >= 1 && q <= kF16WideMaxQ, "f16_prepare_queries: bad arguments");
    const uint32_t ksteps = f16_kpad(dim) / 16;
    char* bfrag = reinterpret_cast<char*>(d_query_frags);
    float* qinv = reinterpret_cast<float*>(bfrag + (size_t)8 * ksteps * 1024);
    // (geometry fixed: 64 blocks x 4 waves = one wave per query slot of the 256)
    hipLaunchKernelGGL(f16_prepare_queries_kernel, dim3(64), dim3(256), 0, stream, d_queries, q, dim, ksteps,
                       metric == ORAMA_METRIC_L2SQ, bfrag, qinv);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

size_t f16_wide_query_bytes(uint32_t dim) { return (size_t)8 * (f16_kpad(dim) / 16) * 1024 + 256 * sizeof(float); }

}  // namespace orama
