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
#include "vec_f16_async.hpp"

namespace orama {

namespace {

using f16async::f16v;
using f16async::h8;
using f16async::wave_or_u32;

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
    static constexpr int kDeadOff = kMetaOff + kMetaBytes;  // tombstone words of the block tile's row tiles, double-buffered
    static constexpr int kDeadBytes = 2 * 256;             // (a 64-lane dword DMA writes 256 bytes)
    static constexpr int kQOff = kDeadOff + kDeadBytes;     // 1/|q| (or |q|^2) and the threshold of every query column
    static constexpr int kQBytes = 2 * 256 * 4;
    // per consumer wave: a 64-bin histogram and kStageCap staged rows (distance, row: 4 bytes each; column: 1 byte)
    static constexpr int kStageOff = kQOff + kQBytes;
    // what LDS is left after the ring, per consumer wave: 256 bytes of histogram + 9 bytes per staged row, in steps of
    // 32 rows, 96 (64 to take + a flush threshold of 32) .. 1024
    static constexpr int kStageFit = ((160 * 1024 - kStageOff) / NC - 256) / 9 / 32 * 32;
    static constexpr int kStageCap = kStageFit > 1024 ? 1024 : kStageFit;
    static_assert(kStageCap >= 96, "no room for the staging area");
    static constexpr int kWaveStage = 256 + kStageCap * 9 + (16 - (kStageCap * 9) % 16) % 16;
    static constexpr int kLdsBytes = kStageOff + NC * kWaveStage;
    static_assert(kQTiles * 32 <= 256, "query metadata is two 1-KiB arrays");
    static_assert(IPS_MIN * (D > 1 ? D - 1 : 1) < 64, "counted vmcnt wait must fit 6 bits");
    static_assert(D >= 1 && KS >= 1, "ring of at least two stages");
    static_assert(kRowTiles * 32 <= 256, "tile metadata is one 1-KiB DMA");
    static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
    static_assert(kThreads <= 1024, "workgroup size");
};

// 4 bytes per lane, global -> LDS (lane l's dword lands at m0 + 4 l); per-lane 64-bit addresses
__device__ __forceinline__ void pc_dma4(const void* src, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, off"
                 :
                 : "v"(src), "{m0}"(lds_addr_uniform)
                 : "memory");
}

// DENSE: the launch writes every distance (the head of the store) instead of keeping the rows under the thresholds.
template <class C, int DBG, bool DENSE>
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
    const uint32_t* dead_lds = reinterpret_cast<const uint32_t*>(lds + C::kDeadOff);  // [2][64], the first kRowTiles used
    float* q_lds = reinterpret_cast<float*>(lds + C::kQOff);                        // [256] 1/|q|, [256] threshold
    // per-column constants of the epilogue: read from LDS there (lgkmcnt) — as global loads they cost every consumer
    // a memory round trip per block tile.  Visible to everybody after the first stage barrier.
    for (uint32_t c = tid; c < 256u; c += C::kThreads) {
        q_lds[c] = c < a.q ? qinv[c] : 0.0f;
        q_lds[256 + c] = (a.tau && c < a.q) ? a.tau[c] : 0.0f;
    }

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
                if (a.dead) {  // and the tombstone words of its row tiles (lanes past the last tile re-read it)
                    uint64_t tl = t_first + ld_bt * C::kRowTiles + ((uint32_t)lane < (uint32_t)C::kRowTiles ? lane : C::kRowTiles - 1);
                    if (tl >= t_end) tl = t_end - 1;
                    pc_dma4(a.dead + tl, lds_base + C::kDeadOff + ld_par * 256);
                }
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
    typedef float f4 __attribute__((ext_vector_type(4)));
    // ---- filter mode: rows under the thresholds are staged per consumer wave in LDS and appended in bulk with ONE
    // global atomic instruction per flush (K2's scheme, vec_f16.hip: an LDS histogram ranks the staged rows within
    // their query column; lane c reserves column c's slots).  A consumer owns CT x 32 <= 64 columns.
    constexpr uint32_t kCap = C::kStageCap;
    const uint32_t stage_off = uniform_u32((uint32_t)C::kStageOff + (uint32_t)w * (uint32_t)C::kWaveStage);
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds + stage_off);
    uint32_t* st_dist = hist + 64;
    uint32_t* st_row = st_dist + kCap;
    uint8_t* st_col = reinterpret_cast<uint8_t*>(st_row + kCap);
    const uint32_t col0 = (uint32_t)(wq * CT) * 32;  // first query column of this consumer
    uint32_t staged = 0;                              // wave-uniform
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    auto bin_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto bin_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto flush = [&]() {
        wave_fence();
        bin_store(&hist[lane], 0u);
        wave_fence();
        uint32_t rank[(kCap + 63) / 64];
#pragma unroll
        for (uint32_t t = 0; t < (kCap + 63) / 64; ++t) {
            const uint32_t i = t * 64 + lane;
            rank[t] = ~0u;
            if (i < staged) {
                bool keep = !(a.dbg & 2u);
                if (a.allow) {
                    const uint64_t doc = a.row_doc[st_row[i]];
                    keep = keep && doc < a.allow_bits && ((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                }
                if (keep) rank[t] = __hip_atomic_fetch_add(&hist[st_col[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        wave_fence();
        const uint32_t mine = bin_load(&hist[lane]);
        bin_store(&hist[lane], mine ? atomicAdd(&a.cand_count[col0 + lane], mine) : 0u);
        wave_fence();
#pragma unroll
        for (uint32_t t = 0; t < (kCap + 63) / 64; ++t) {
            const uint32_t i = t * 64 + lane;
            if (i < staged && rank[t] != ~0u) {
                const uint32_t c = st_col[i];
                const uint64_t pos = (uint64_t)(col0 + c) * a.cand_stride + bin_load(&hist[c]) + rank[t];
                a.cand_dist[pos] = __uint_as_float(st_dist[i]);
                a.cand_row[pos] = st_row[i];
            }
        }
        wave_fence();
        staged = 0;
    };

    // Filter mode: `start` = 16 (i CT + j) + r of the first accumulator row still to be looked at; returns 16 RT CT when the
    // block tile is done, else the position at which the staging area ran full — the caller flushes (at a point where nothing
    // of the epilogue is live: inlined into the row loop the flush cost the default geometry 8 bytes of scratch per lane,
    // and a spilled value's reload waits on vmcnt) and calls again.  Dense mode: always done.
    auto epilogue = [&](uint64_t bt, uint32_t par, uint32_t start) -> uint32_t {
        const uint32_t hi = (lane >> 5) ? 4u : 0u;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const uint32_t tl = (uint32_t)(wr * RT + i);
            const uint64_t tile = t_first + bt * C::kRowTiles + tl;
            if (tile >= t_end) continue;  // wave-uniform
            if (!DENSE && (uint32_t)(i + 1) * CT * 16u <= start) continue;  // done before the flush
            const uint32_t dead_word = a.dead ? dead_lds[par * 64 + tl] : 0u;
            // this lane's 16 accumulator rows are (r & 3) + 8 (r >> 2) + hi: four 16-byte reads of the tile's norms
            const float* nrm = inv_lds + par * 256 + tl * 32 + hi;
            f16v nrmv;
            {
                const f4* np = reinterpret_cast<const f4*>(nrm);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f4 v = np[2 * g4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) nrmv[4 * g4 + e] = v[e];
                }
            }
            const bool full = tile * 32 + 32 <= a.row_end;
            const uint32_t left = full ? 32u : (uint32_t)(a.row_end - tile * 32);  // rows of the tile inside the store
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                if (!DENSE && (uint32_t)(i * CT + j + 1) * 16u <= start) continue;  // done before the flush
                const uint32_t cl = (uint32_t)j * 32 + (lane & 31);  // column inside this consumer's group
                const uint32_t col = col0 + cl;
                const bool live = col < a.q;
                if (DENSE && !live) continue;
                const float qi = q_lds[col];
                const float tau = q_lds[256 + col];
                // cosine: 1 - s (1/|x|)(1/|q|);  L2: (|q|^2 + |x|^2) - 2 s   (nrm / qi hold the squared norms then) —
                // as the fused operations the compiler contracts the plain expressions to (vec_f16.hip)
                auto dist_of = [&](float dot, float n, float qv) -> float {
                    return l2 ? __builtin_fmaf(-2.0f, dot, qv + n) : __builtin_fmaf(-dot, n * qv, 1.0f);
                };
                if constexpr (DENSE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t ri = (uint32_t)((r & 3) + 8 * (r >> 2)) + hi;
                        const uint64_t row = tile * 32 + ri;
                        if (!full && row >= a.row_end) continue;
                        bool excluded = (dead_word >> ri) & 1u;
                        if (!excluded && a.allow) {
                            const uint64_t doc = a.row_doc[row];
                            excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                        }
                        a.out_dense[(uint64_t)col * a.dense_stride + (row - a.row_begin)] =
                            excluded ? __builtin_nanf("") : dist_of(acc[i][j][r], nrmv[r], qi);
                    }
                    continue;
                }
                // fast reject: the minimum of the 16 distances against the threshold, one ballot
                float best = __builtin_huge_valf();
                if (l2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) best = fminf(best, __builtin_fmaf(-2.0f, acc[i][j][r], qi + nrmv[r]));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) best = fminf(best, __builtin_fmaf(-acc[i][j][r], nrmv[r] * qi, 1.0f));
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && best < tau) == 0, 1)) continue;
                float qi_s = qi;  // (an operand the optimiser cannot see through: nothing is kept from the fast path)
                asm volatile("" : "+v"(qi_s));
                const uint32_t alive = (~dead_word & (left >= 32u ? ~0u : ((1u << left) - 1u))) >> hi;
                uint32_t m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    m |= ((live && dist_of(acc[i][j][r], nrmv[r], qi_s) < tau ? 1u : 0u) & (alive >> ((r & 3) + 8 * (r >> 2)))) << r;
                // the accumulator rows somebody passes (usually one or two of the 16): OR over the wave, then only those, with
                // a wave-uniform index into the accumulators
                uint32_t any = wave_or_u32(m);
                const uint32_t base = (uint32_t)(i * CT + j) * 16u;
                if (start > base) any &= ~0u << (start - base);  // resuming after a flush
#pragma unroll 1
                while (any) {
                    const uint32_t r = (uint32_t)__builtin_ctz(any);
                    const bool mine = (m >> r) & 1u;
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
                    const uint32_t n_pass = (uint32_t)__popcll(bal);
                    if (staged + n_pass > kCap) return base + r;  // no room: flush, then resume here
                    any &= any - 1u;
                    if (mine) {
                        const uint32_t pos =
                            staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        st_dist[pos] = __float_as_uint(dist_of(acc[i][j][r], nrm[(r & 3u) + 8u * (r >> 2)], qi_s));
                        st_row[pos] = (uint32_t)(tile * 32) + ((r & 3u) + 8u * (r >> 2)) + hi;
                        st_col[pos] = (uint8_t)cl;
                    }
                    staged = uniform_u32(staged + n_pass);
                }
            }
        }
        return 16u * RT * CT;
    };
    auto finish_tile = [&](uint64_t bt, uint32_t par) {
        uint32_t at = 0;
        while ((at = epilogue(bt, par, at)) < 16u * RT * CT) flush();
        if (!DENSE && staged > kCap - 64) flush();  // room for a block tile's usual few rows
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
                finish_tile(cp_bt, cp_par);
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
    if (!DENSE && staged) flush();
    if (TRACE && w == 0 && lane == 0) trace[8192 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
}

template <class C, int DBG, bool WITH_DENSE = true>
int pc_launch(orama_ctx* ctx, const F16ScanArgs& a, const char* bfrag, const float* qinv, uint32_t ksteps,
              hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_pc_kernel<C, DBG, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if constexpr (WITH_DENSE)
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_pc_kernel<C, DBG, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    uint64_t blocks = (tiles + C::kRowTiles - 1) / C::kRowTiles;
    if (blocks > (uint64_t)ctx->compute_units) blocks = (uint64_t)ctx->compute_units;
    unsigned long long* trace = nullptr;
    if (DBG & 16) {
        const char* e = orama::dev_env("ORAMA_K2D_TRACE");  // device pointer of >= 64 KiB (hex), set by the probe script
        if (e) trace = reinterpret_cast<unsigned long long*>(std::strtoull(e, nullptr, 16));
        ORAMA_REQUIRE(trace, "trace build needs ORAMA_K2D_TRACE");
    }
    if (a.out_dense) {
        if constexpr (WITH_DENSE)
            hipLaunchKernelGGL((vec_scan_f16_pc_kernel<C, DBG, true>), dim3((uint32_t)blocks), dim3(C::kThreads), C::kLdsBytes,
                               stream, a, bfrag, qinv, ksteps, f16_tile_bytes(a.dim), trace);
        else
            ORAMA_REQUIRE(false, "this geometry has no dense form");
    } else
        hipLaunchKernelGGL((vec_scan_f16_pc_kernel<C, DBG, false>), dim3((uint32_t)blocks), dim3(C::kThreads), C::kLdsBytes,
                           stream, a, bfrag, qinv, ksteps, f16_tile_bytes(a.dim), trace);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace

// Geometries (ctx->f16_wide - 1; DESIGN.md §4 K2d has the measurements):
//   1: 12 consumers of 2 x 2 tiles (fragment prefetch) + 4 loaders = 16 waves (<= 128 VGPRs), block tile 192 rows x
//      256 queries, stages of 2 k-steps (28 KiB), ring of 5                                 — the default
//   2: 8 consumers of 3 x 2 tiles (fragment prefetch) + 4 loaders = 12 waves (<= 168 VGPRs), block tile 192 x 256
// Tried and dropped (profiles/r02_k2d_geometries.md): PcB with a ring of 4 and 384 staged rows per wave instead of 96
// (no difference: 4.55 vs 4.52 ms), 8 consumers of 2 x 4 tiles at <= 168 VGPRs (the allocator
// spills accumulators inside the K loop: 12.2 ms), 4 consumers of 2 x 4 tiles + 4 loaders (128 x 256 block tile,
// twice the query-fragment traffic: 6.4 ms), 3 x 2 tiles without fragment prefetch (6.2 ms).
using PcB = PcCfg<2, 2, 3, 4, 4, 2, 5, 1>;
using PcD = PcCfg<3, 2, 2, 4, 4, 2, 5, 1>;
// <= 128 queries: half the query tiles — 8 consumers of 2 x 2 tiles, block tile 256 rows x 128 queries, ring of 5
using PcA2 = PcCfg<2, 2, 4, 2, 4, 2, 5, 1>;

int launch_vec_scan_f16_pc(orama_ctx* ctx, const F16ScanArgs& a_in, void* d_query_frags, hipStream_t stream, int geometry) {
    static const uint32_t k2dbg = [] { const char* e = orama::dev_env("ORAMA_K2_DBG"); return e ? (uint32_t)std::atoi(e) : 0u; }();
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
    if (const char* e = orama::dev_env("ORAMA_K2C_DBG")) dbg = std::atoi(e);
    if (dbg && !a.out_dense && geometry == 1 && a.q > 128) {  // ablation builds of the default geometry (timing only)
        switch (dbg) {
            case 9: return pc_launch<PcB, 9>(ctx, a, bfrag, qinv, ksteps, stream);    // DMA only
            case 13: return pc_launch<PcB, 13>(ctx, a, bfrag, qinv, ksteps, stream);  // corpus DMA only
            case 2: return pc_launch<PcB, 2>(ctx, a, bfrag, qinv, ksteps, stream);    // LDS reads + MFMA + barriers
            case 1: return pc_launch<PcB, 1>(ctx, a, bfrag, qinv, ksteps, stream);    // DMA + LDS reads, no MFMA
            case 4: return pc_launch<PcB, 4>(ctx, a, bfrag, qinv, ksteps, stream);    // corpus DMA + compute
            case 16: return pc_launch<PcB, 16>(ctx, a, bfrag, qinv, ksteps, stream);  // full kernel + timeline stamps
            case 25: return pc_launch<PcB, 25>(ctx, a, bfrag, qinv, ksteps, stream);  // DMA only + stamps
            case 32: return pc_launch<PcB, 32>(ctx, a, bfrag, qinv, ksteps, stream);  // everything but the epilogue
            default: break;
        }
    }
    if (a.q <= 128) return pc_launch<PcA2, 0>(ctx, a, bfrag, qinv, ksteps, stream);
    // (the 3x2 ablation geometry's dense form needs 16 bytes of scratch per lane: dense output always takes the default one)
    if (geometry == 2 && !a.out_dense) return pc_launch<PcD, 0, false>(ctx, a, bfrag, qinv, ksteps, stream);
    return pc_launch<PcB, 0>(ctx, a, bfrag, qinv, ksteps, stream);
}

}  // namespace orama
