// vec_f16_pc.hip — K2d: the fp16 scan for wide query batches (65..256 queries per corpus pass; config C5) as a
// PRODUCER / CONSUMER kernel: dedicated loader waves feed an LDS ring with global->LDS DMA, consumer waves do nothing
// but LDS fragment reads and MFMAs.
//
// Why (profiles/r02_k2c_ablation.log, 10 M x 768 fp16, 256 queries, one MI355X): K2c — the round-1 kernel, where the
// MFMA waves also issue the DMA — takes 5.18 ms per pass although its two halves are short on their own: DMA only
// 2.44 ms (corpus + query fragments; the corpus alone: 2.46 ms, i.e. the L2-resident query fragments ride for free),
// LDS reads + MFMA only 2.70 ms.  Adding the corpus DMA to the compute costs +0.87 ms, the query-fragment DMA another
// +1.6 ms: every global_load_lds issued by a wave that should be issuing MFMAs stalls that wave for 125-230 cycles
// (VMEM issue blocks while the CU's miss queue is full), and the per-stage barrier hands the stall to everybody.
// A wave that ONLY issues DMA absorbs that back-pressure without touching the matrix pipes.
//
//   workgroup = NC consumer waves + NL loader waves, one workgroup per CU, persistent over block tiles
//   block tile = (32·RT·WR) rows x (32·CT·WQ) queries; consumer (wr, wq) accumulates RT x CT MFMA tiles
//   K loop     = stages of KS k-steps through an LDS ring of NBUF stages; loaders run D = NBUF-1 stages ahead
//   loaders    : fragment t of a stage (1 KiB = one global_load_lds_dwordx4 per wave) is moved by loader t % NL;
//                after issuing stage g+D a loader waits with a COUNTED vmcnt until stage g+1 has landed — LDS-DMA
//                completes in issue order under counted waits (profiles/r02_ldsdma_order_probe.log: 0 violations in
//                2.1 G checked words, cold/hot and nt/plain mixes) — then meets the consumers at the stage barrier
//   consumers  : per k-step RT + CT ds_read_b128 (lane-linear, conflict-free) feed RT·CT v_mfma_f32_32x32x16_f16
//   one s_barrier per stage: "stage g+1 is in LDS" and "stage g has been read" in one rendezvous
//
// HBM traffic: the corpus once per launch (rows·kpad·2 bytes); the prepared query fragments come from L2.
// Accumulation order over k is K2's (ascending k-steps into one accumulator chain), so a wide batch returns
// distances bit-identical to solo queries.  Epilogue = K2's on the accumulator registers (1 - s/|x||q| or the L2
// form, tombstones, allow bit, threshold filter / dense store).
#include "vec_f16.hpp"

#include <cstdlib>
#include <type_traits>

#include "device_utils.hpp"

namespace orama {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// 16 bytes per lane, global -> LDS, asynchronous (counted by vmcnt); lane l's data lands at LDS address m0 + 16 l.
// m0 is an INPUT operand of the statement ("{m0}"): the compiler materialises it and knows it is live; the leading
// s_nop is the wait state gfx9 wants between a write of m0 and an LDS-DMA instruction reading it.
__device__ __forceinline__ void pc_dma16(uint64_t saddr_uniform, uint32_t voff, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(saddr_uniform), "{m0}"(lds_addr_uniform)
                 : "memory");
}
__device__ __forceinline__ void pc_dma16_nt(uint64_t saddr_uniform, uint32_t voff, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                 :
                 : "v"(voff), "s"(saddr_uniform), "{m0}"(lds_addr_uniform)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Geometry of one instantiation.
template <int RT_, int CT_, int WR_, int WQ_, int NL_, int KS_, int NBUF_, int PF_ = 0>
struct PcCfg {
    static constexpr int RT = RT_, CT = CT_, WR = WR_, WQ = WQ_, NL = NL_, KS = KS_, NBUF = NBUF_;
    static constexpr bool PF = PF_ != 0;               // consumers prefetch the next k-step's fragments (2x fragment registers)
    static constexpr int NC = WR * WQ;                 // consumer waves
    static constexpr int kThreads = (NC + NL) * 64;
    static constexpr int kRowTiles = RT * WR;          // 32-row tiles per block tile
    static constexpr int kQTiles = CT * WQ;            // 32-query tiles per block tile
    static constexpr int FA = kRowTiles * KS;          // corpus fragments per stage
    static constexpr int FB = kQTiles * KS;            // query fragments per stage
    static constexpr int F = FA + FB;
    static constexpr int IPS = (F + NL - 1) / NL;      // DMA instructions per loader and stage (the last one only for l < F % NL)
    static constexpr int IPS_MIN = F / NL;             // what every loader issues at least: the counted waits use this
    static constexpr int D = NBUF - 1;                 // stages a loader runs ahead
    static constexpr int kStageBytes = F * 1024;
    static constexpr int kMetaOff = NBUF * kStageBytes;
    static constexpr int kMetaBytes = 2 * 1024;        // 1/|x| (or |x|^2) of the block tile's rows, double-buffered
    static constexpr int kLdsBytes = kMetaOff + kMetaBytes;
    static_assert(IPS_MIN * (D > 1 ? D - 1 : 1) < 64, "counted vmcnt wait must fit 6 bits");
    static_assert(D >= 1 && KS >= 1, "ring of at least two stages");
    static_assert(kRowTiles * 32 <= 256, "tile metadata is one 1-KiB DMA");
    static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
    static_assert(kThreads <= 1024, "workgroup size");
};

template <class C, int DBG>
__global__ __launch_bounds__(C::kThreads) void vec_scan_f16_pc_kernel(F16ScanArgs a, const char* __restrict__ bfrag,
                                                                      const float* __restrict__ qinv,
                                                                      uint32_t ksteps, uint64_t tile_bytes,
                                                                      unsigned long long* __restrict__ trace) {
    // DBG bit 4 (16): block 0 records s_memtime stamps per stage (loader 0: after the barrier / after issuing /
    // after the counted wait; consumer 0: after the barrier / after the stage's MFMAs were issued) — scripts/k2d_trace.py
    constexpr bool TRACE = (DBG & 16) != 0;
    constexpr uint64_t kTraceStages = 1024;
    constexpr int RT = C::RT, CT = C::CT, KS = C::KS, NBUF = C::NBUF, NL = C::NL, NC = C::NC;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = uniform_u32(tid >> 6);
    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;                                   // row tiles [t_first, t_end)
    const uint64_t n_bt = (t_end - t_first + C::kRowTiles - 1) / C::kRowTiles;      // block tiles
    if (blockIdx.x >= n_bt) return;
    const uint64_t my_bt = (n_bt - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t S = ksteps / KS;                                                 // stages per block tile
    const uint64_t total = my_bt * S;
    const float* inv_lds = reinterpret_cast<const float*>(lds + C::kMetaOff);      // [2][256]

    if (w >= NC) {
        // ================================================================= loader wave
        const int l = w - NC;
        const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds;
        const uint32_t vlane = (uint32_t)lane * 16;
        const char* base = reinterpret_cast<const char*>(a.tiled);
        // fragment t = l + NL i of a stage: t < FA -> corpus (row tile t / KS, k-step t % KS), else query fragment
        uint64_t f_off[C::IPS];   // A: byte offset inside the block tile at k-step 0; B: absolute address at k-step 0
        uint32_t f_tile[C::IPS];  // A: row tile inside the block tile; B: 0xffffffff
#pragma unroll
        for (int i = 0; i < C::IPS; ++i) {
            const uint32_t t = (uint32_t)l + (uint32_t)NL * i;
            if (t >= (uint32_t)C::F) {
                f_tile[i] = 0xfffffffeu;  // no such fragment for this loader
                f_off[i] = 0;
            } else if (t < (uint32_t)C::FA) {
                f_tile[i] = t / KS;
                f_off[i] = (uint64_t)(t / KS) * tile_bytes + (uint64_t)(t % KS) * 1024;
            } else {
                const uint32_t u = t - C::FA;
                f_tile[i] = 0xffffffffu;
                f_off[i] = (uint64_t)(uintptr_t)bfrag + ((uint64_t)(u / KS) * ksteps + (u % KS)) * 1024;
            }
        }
        const uint64_t bt_stride = (uint64_t)gridDim.x * C::kRowTiles * tile_bytes;
        uint64_t ld_bt = blockIdx.x;        // block tile under the load cursor
        uint32_t ld_s = 0, ld_par = 0;      // stage inside it, metadata buffer parity
        uint64_t ld_koff = 0;
        uint64_t ld_base = (uint64_t)(uintptr_t)base + (t_first + (uint64_t)blockIdx.x * C::kRowTiles) * tile_bytes;
        const bool partial_last = ((t_end - t_first) % C::kRowTiles) != 0;
        auto issue_stage = [&](int buf) {
            const uint32_t lbuf = lds_base + (uint32_t)buf * C::kStageBytes;
            if (!(DBG & 2) && ld_s == 0 && l == 0) {
                // first stage of a block tile: also its row metadata (1 KiB, contiguous; rows past the end of the
                // store read the zero-initialised padding of the array)
                pc_dma16((uint64_t)(uintptr_t)a.inv_norm + (t_first + ld_bt * C::kRowTiles) * 128, vlane,
                         lds_base + C::kMetaOff + ld_par * 1024);
            }
            const bool clamp = partial_last && ld_bt == n_bt - 1;  // the last block tile may hold fewer row tiles
#pragma unroll
            for (int i = 0; i < C::IPS; ++i) {
                if (DBG & 2) continue;
                const uint32_t t = (uint32_t)l + (uint32_t)NL * i;
                if (i == C::IPS - 1 && t >= (uint32_t)C::F) continue;  // wave-uniform
                if (f_tile[i] != 0xffffffffu) {
                    uint64_t sa = ld_base + f_off[i] + ld_koff;
                    if (clamp && t_first + ld_bt * C::kRowTiles + f_tile[i] >= t_end)  // re-read a valid tile (masked later)
                        sa = (uint64_t)(uintptr_t)base + (t_end - 1) * tile_bytes +
                             (f_off[i] - (uint64_t)f_tile[i] * tile_bytes) + ld_koff;
                    pc_dma16_nt(sa, vlane, lbuf + t * 1024);
                } else if (!(DBG & 4)) {
                    pc_dma16(f_off[i] + ld_koff, vlane, lbuf + t * 1024);
                }
            }
            ld_koff += (uint64_t)KS * 1024;
            if (++ld_s == S) {
                ld_s = 0;
                ld_koff = 0;
                ld_bt += gridDim.x;
                ld_base += bt_stride;
                ld_par ^= 1;
            }
        };
        // prologue: stages 0 .. D-1, then wait for stage 0
        uint64_t issued = 0;
        int ibuf = 0;
        for (; issued < (uint64_t)C::D && issued < total; ++issued) {
            issue_stage(ibuf);
            ibuf = ibuf == NBUF - 1 ? 0 : ibuf + 1;
        }
        // counted waits: everything but the newest D-1 stages must have landed.  The constant uses IPS_MIN, which is
        // exact for loaders issuing IPS_MIN per stage and conservative (waits for a little more) for the others and
        // for stages that carry the extra metadata fragment.
        if (total > (uint64_t)C::D) wait_vmcnt<C::IPS_MIN * (C::D - 1)>();
        else wait_vmcnt<0>();
        for (uint64_t g = 0; g < total; ++g) {
            __syncthreads();  // B_g: stage g is in LDS for everybody; stage g-1 has been read
            const bool tr = TRACE && blockIdx.x == 0 && l == 0 && g < kTraceStages && lane == 0;
            if (tr) trace[g * 8 + 0] = __builtin_amdgcn_s_memtime();
            if (issued < total) {
                issue_stage(ibuf);  // stage g + D into the buffer stage g-1 just left
                ibuf = ibuf == NBUF - 1 ? 0 : ibuf + 1;
                ++issued;
            }
            if (tr) trace[g * 8 + 1] = __builtin_amdgcn_s_memtime();
            // stage g+1 must have landed before B_{g+1}: everything except the newest D-1 stages
            if (issued == g + 1 + (uint64_t)C::D) wait_vmcnt<C::IPS_MIN * (C::D - 1)>();
            else wait_vmcnt<0>();  // tail of the launch: fewer stages in flight than the constant assumes
            if (tr) trace[g * 8 + 2] = __builtin_amdgcn_s_memtime();
        }
        return;
    }

    // ===================================================================== consumer wave
    const int wr = w % C::WR;  // row group: row tiles RT wr .. RT wr + RT - 1 of the block tile
    const int wq = w / C::WR;  // query group: query tiles CT wq .. CT wq + CT - 1
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;

    f16v acc[RT][CT];
    typedef const uint32_t __attribute__((address_space(4))) cu32;
    auto epilogue = [&](uint64_t bt, uint32_t par) {
        const uint32_t hi = (lane >> 5) ? 4u : 0u;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const uint32_t tl = (uint32_t)(wr * RT + i);
            const uint64_t tile = t_first + bt * C::kRowTiles + tl;
            if (tile >= t_end) continue;  // wave-uniform
            const uint32_t dead_word = a.dead ? ((cu32*)(uintptr_t)a.dead)[tile] : 0u;  // scalar load
            float nrm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[r] = inv_lds[par * 256 + tl * 32 + (r & 3) + 8 * (r >> 2) + hi];
            const bool full = tile * 32 + 32 <= a.row_end;
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const uint32_t col = (uint32_t)(wq * CT + j) * 32 + (lane & 31);
                if (col >= a.q) continue;
                // per-lane constants of the query column, re-read per block tile (L2 hits) instead of living in
                // registers across the K loop: the loop runs at the edge of the register budget
                const float qi = qinv[col];
                const float tau = a.tau ? a.tau[col] : 0.0f;
                // cosine: 1 - s (1/|x|)(1/|q|);  L2: (|q|^2 + |x|^2) - 2 s   (nrm / qi hold the squared norms then)
                auto distance = [&](int r) -> float {
                    return l2 ? (qi + nrm[r]) - 2.0f * acc[i][j][r] : 1.0f - acc[i][j][r] * (nrm[r] * qi);
                };
                if (!a.out_dense) {
                    // filter mode, fast reject: almost no row beats the running k-th best distance, so take the
                    // minimum of the 16 distances first and look closer only when it passes
                    float best = __builtin_huge_valf();
#pragma unroll
                    for (int r = 0; r < 16; ++r) best = fminf(best, distance(r));
                    if (!(best < tau)) continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t ri = (uint32_t)((r & 3) + 8 * (r >> 2)) + hi;
                    const uint64_t row = tile * 32 + ri;
                    if (!full && row >= a.row_end) continue;
                    const float dist = distance(r);
                    bool excluded = (dead_word >> ri) & 1u;
                    if (a.out_dense) {
                        if (!excluded && a.allow) {
                            const uint64_t doc = a.row_doc[row];
                            excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                        }
                        a.out_dense[(uint64_t)col * a.dense_stride + (row - a.row_begin)] =
                            excluded ? __builtin_nanf("") : dist;
                    } else if (!excluded && dist < tau && !(a.dbg & 2u)) {
                        if (a.allow) {  // only rows that pass the threshold pay for the filter lookup
                            const uint64_t doc = a.row_doc[row];
                            excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                        }
                        if (!excluded) {
                            const uint32_t pos = atomicAdd(&a.cand_count[col], 1u);
                            a.cand_dist[(uint64_t)col * a.cand_stride + pos] = dist;
                            a.cand_row[(uint64_t)col * a.cand_stride + pos] = (uint32_t)row;
                        }
                    }
                }
            }
        }
    };

    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (TRACE && w == 0 && lane == 0) trace[8192 + 2 * blockIdx.x] = __builtin_amdgcn_s_memtime();
    uint64_t cp_bt = blockIdx.x;
    uint32_t cp_s = 0, cp_par = 0;
    int buf = 0;
    // one stage of the K loop; FIRST (a new block tile) starts every accumulator chain from C = 0 instead of
    // clearing RT*CT*16 registers — decided once per stage, not per MFMA
    auto stage = [&](const char* la, const char* lb, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int NS = C::PF ? 2 : 1;
        h8 fa[NS][RT], fb[NS][CT];
        auto load_frags = [&](int ks, int slot) {
            if (DBG & 8) return;
#pragma unroll
            for (int i = 0; i < RT; ++i)
                fa[slot][i] = *reinterpret_cast<const h8*>(la + (size_t)((wr * RT + i) * KS + ks) * 1024);
#pragma unroll
            for (int j = 0; j < CT; ++j)
                fb[slot][j] = *reinterpret_cast<const h8*>(lb + (size_t)((wq * CT + j) * KS + ks) * 1024);
        };
        if (C::PF) load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int sl = C::PF ? (ks & 1) : 0;
            if (C::PF) {
                if (ks + 1 < KS) load_frags(ks + 1, sl ^ 1);
            } else {
                load_frags(ks, 0);
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (DBG & 1) {
                        if (ks == 0 && FIRST) acc[i][j] = zero;
                        asm volatile("" ::"v"(fa[sl][i]), "v"(fb[sl][j]));
                    } else if (ks == 0 && FIRST) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], zero, 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
                    }
                }
            // without prefetch: keep the fragment reads of the next k-step behind these MFMAs — hoisted above them
            // they cost RT + CT more live fragments, which a tight register budget pays for by spilling accumulators
            if (!C::PF) __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (uint64_t g = 0; g < total; ++g) {
        __syncthreads();  // B_g
        const bool tr = TRACE && blockIdx.x == 0 && w == 0 && g < kTraceStages && lane == 0;
        if (tr) trace[g * 8 + 4] = __builtin_amdgcn_s_memtime();
        const char* la = lds + (size_t)buf * C::kStageBytes + (size_t)lane * 16;
        const char* lb = la + (size_t)C::FA * 1024;
        if (cp_s == 0) stage(la, lb, std::true_type{});
        else stage(la, lb, std::false_type{});
        if (tr) trace[g * 8 + 5] = __builtin_amdgcn_s_memtime();
        if (++cp_s == S) {
            if (DBG == 0 || DBG == 16) {
                epilogue(cp_bt, cp_par);
                if (tr) trace[g * 8 + 6] = __builtin_amdgcn_s_memtime();
            } else {  // ablation builds: keep every accumulator alive
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
                if (sum == 12345.678f) a.cand_count[0] = 1;
            }
            cp_s = 0;
            cp_par ^= 1;
            cp_bt += gridDim.x;
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    if (TRACE && w == 0 && lane == 0) trace[8192 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
}

template <class C, int DBG>
int pc_launch(orama_ctx* ctx, const F16ScanArgs& a, const char* bfrag, const float* qinv, uint32_t ksteps,
              hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_pc_kernel<C, DBG>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    uint64_t blocks = (tiles + C::kRowTiles - 1) / C::kRowTiles;
    if (blocks > (uint64_t)ctx->compute_units) blocks = (uint64_t)ctx->compute_units;
    unsigned long long* trace = nullptr;
    if (DBG & 16) {
        const char* e = std::getenv("ORAMA_K2D_TRACE");  // device pointer of >= 64 KiB (hex), set by the probe script
        if (e) trace = reinterpret_cast<unsigned long long*>(std::strtoull(e, nullptr, 16));
        ORAMA_REQUIRE(trace, "trace build needs ORAMA_K2D_TRACE");
    }
    hipLaunchKernelGGL((vec_scan_f16_pc_kernel<C, DBG>), dim3((uint32_t)blocks), dim3(C::kThreads), C::kLdsBytes, stream,
                       a, bfrag, qinv, ksteps, f16_tile_bytes(a.dim), trace);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace

// Geometries (ctx->f16_wide - 1; DESIGN.md §4 K2d has the measurements):
//   1: 12 consumers of 2 x 2 tiles (fragment prefetch) + 4 loaders = 16 waves (<= 128 VGPRs), block tile 192 rows x
//      256 queries, stages of 2 k-steps (28 KiB), ring of 5                                 — the default
//   2: 8 consumers of 3 x 2 tiles (fragment prefetch) + 4 loaders = 12 waves (<= 168 VGPRs), block tile 192 x 256
// Tried and dropped (profiles/r02_k2d_geometries.md): 8 consumers of 2 x 4 tiles at <= 168 VGPRs (the allocator
// spills accumulators inside the K loop: 12.2 ms), 4 consumers of 2 x 4 tiles + 4 loaders (128 x 256 block tile,
// twice the query-fragment traffic: 6.4 ms), 3 x 2 tiles without fragment prefetch (6.2 ms).
using PcB = PcCfg<2, 2, 3, 4, 4, 2, 5, 1>;
using PcD = PcCfg<3, 2, 2, 4, 4, 2, 5, 1>;
// <= 128 queries: half the query tiles — 8 consumers of 2 x 2 tiles, block tile 256 rows x 128 queries, ring of 5
using PcA2 = PcCfg<2, 2, 4, 2, 4, 2, 5, 1>;

int launch_vec_scan_f16_pc(orama_ctx* ctx, const F16ScanArgs& a_in, void* d_query_frags, hipStream_t stream, int geometry) {
    static const uint32_t k2dbg = [] { const char* e = std::getenv("ORAMA_K2_DBG"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    F16ScanArgs a = a_in;
    if (!a.out_dense) a.dbg = k2dbg & 2u;  // timing ablation: no candidate appends
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries && d_query_frags, "vec_scan_f16_pc: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF16WideMaxQ, "vec_scan_f16_pc: q=%u outside [1, %u]", a.q, kF16WideMaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f16_pc: bad row range");
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count), "vec_scan_f16_pc: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f16_pc: filter needs row_doc");
    const uint32_t ksteps = f16_kpad(a.dim) / 16;
    ORAMA_REQUIRE(ksteps % 2 == 0, "vec_scan_f16_pc: kpad not a multiple of 32");
    if (a.row_begin == a.row_end) return ORAMA_OK;
    const char* bfrag = reinterpret_cast<const char*>(d_query_frags);
    const float* qinv = reinterpret_cast<const float*>(bfrag + (size_t)8 * ksteps * 1024);
    ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
    int dbg = 0;
    if (const char* e = std::getenv("ORAMA_K2C_DBG")) dbg = std::atoi(e);
    if (dbg && !a.out_dense && geometry == 1 && a.q > 128) {  // ablation builds of the default geometry (timing only)
        switch (dbg) {
            case 9: return pc_launch<PcB, 9>(ctx, a, bfrag, qinv, ksteps, stream);    // DMA only
            case 13: return pc_launch<PcB, 13>(ctx, a, bfrag, qinv, ksteps, stream);  // corpus DMA only
            case 2: return pc_launch<PcB, 2>(ctx, a, bfrag, qinv, ksteps, stream);    // LDS reads + MFMA + barriers
            case 1: return pc_launch<PcB, 1>(ctx, a, bfrag, qinv, ksteps, stream);    // DMA + LDS reads, no MFMA
            case 4: return pc_launch<PcB, 4>(ctx, a, bfrag, qinv, ksteps, stream);    // corpus DMA + compute
            case 16: return pc_launch<PcB, 16>(ctx, a, bfrag, qinv, ksteps, stream);  // full kernel + timeline stamps
            case 25: return pc_launch<PcB, 25>(ctx, a, bfrag, qinv, ksteps, stream);  // DMA only + stamps
            default: break;
        }
    }
    if (a.q <= 128) return pc_launch<PcA2, 0>(ctx, a, bfrag, qinv, ksteps, stream);
    if (geometry == 2) return pc_launch<PcD, 0>(ctx, a, bfrag, qinv, ksteps, stream);
    return pc_launch<PcB, 0>(ctx, a, bfrag, qinv, ksteps, stream);
}

}  // namespace orama
