// vec_f16_qs.hip — K2q: the fp16 scan for wide query batches (65..256 queries per corpus pass; config C5) with the
// QUERIES STATIONARY IN REGISTERS.
//
// Why (profiles/r02_k2c_ablation.log, r02_k2d_stage_timeline.log): K2d streams both MFMA operands through LDS — per
// block tile of 192 rows x 256 queries every k-step moves 6 corpus fragments (HBM) AND 8 query fragments (L2) into the
// ring, so the global->LDS traffic is 2.33x the corpus bytes, its loader waves are the long leg of every stage
// (1 016 of 1 380 ticks) and the kernel sits at 43-45 % of the HBM roofline.  The query operand never changes during a
// launch: 256 queries x 768 dims of fp16 are 384 KiB, and a compute unit has 512 KiB of vector registers.
//
//   workgroup = 8 waves (2 per SIMD, <= 256 registers each), one workgroup per CU, persistent over block tiles
//   wave w    : owns ONE 32-query tile c = w % NQT for the whole launch — its B fragments for all kpad/16 k-steps
//               (kpad/4 registers: 192 at 768 dims) are loaded once from the prepared fragment buffer and stay in VGPRs —
//               and accumulates 2 row tiles x that query tile (2 x 16 accumulator registers)
//   block tile: 64 rows x 256 queries (NQT = 8; <= 128 queries: NQT = 4, two row groups, 128 rows)
//   LDS       : ONLY the corpus ring — stages of KS k-steps x the block tile's row tiles, NBUF stages, filled by
//               global->LDS DMA (1 KiB per instruction, `nt`); every wave issues its 1/8 of a stage right after the
//               stage barrier and waits for it with a COUNTED vmcnt (LDS-DMA of one wave completes in issue order:
//               profiles/r02_ldsdma_order_probe.log), D = NBUF - 1 stages (112 KiB at 768 dims) in flight per CU
//   per k-step: 2 ds_read_b128 (the A fragments, lane-linear) + 2 v_mfma_f32_32x32x16_f16 per wave; the B operand comes
//               from the register file.  One s_barrier per stage of KS k-steps.
//
// global->LDS traffic = the corpus, once (1.0x instead of 2.33x); LDS reads per MFMA as K2d's 2 x 2 tiles (1 KiB).
// Accumulation order over k is K2's (ascending k-steps into one accumulator chain starting from C = 0; corpus = A
// operand, query = B operand), so a wide batch returns distances bit-identical to solo queries and to K2c / K2d
// (tests/test_vector_f16_gpu.py::test_wide_batches_equal_solo_queries runs every form).
// Limit: kpad <= 768 (the B fragments must fit the register budget); wider rows take K2d.
#include "vec_f16.hpp"

#include <cstdlib>
#include <type_traits>

#include "device_utils.hpp"
#include "vec_f16_async.hpp"

namespace orama {

namespace {

using namespace f16async;

// KSTEPS = kpad / 16; NQT = 32-query tiles per block tile (8 or 4); KS k-steps per stage; NBUF stages in the ring
template <int KSTEPS_, int NQT_, int KS_, int NBUF_, bool LAGGED_ = true, int RTW_ = 1, int P_ = 2>
struct QsCfg {
    static constexpr int KSTEPS = KSTEPS_, NQT = NQT_, KS = KS_, NBUF = NBUF_;
    static constexpr int kWaves = 8, kThreads = kWaves * 64;
    static constexpr int NG = kWaves / NQT;            // row groups (waves sharing a query tile work on different rows)
    static constexpr int RTW = RTW_;                   // row tiles per wave (accumulators: 16 RTW registers)
    static constexpr int RTB = RTW * NG;               // row tiles per block tile
    static constexpr int S = KSTEPS / KS;              // stages per block tile
    static constexpr int F = RTB * KS;                 // fragments (1 KiB) per stage
    static constexpr int IPS = F / kWaves;             // DMA instructions per wave and stage
    // The two waves of a SIMD (w and w + 4: the second four waves land on the same four SIMDs) run LAG stages apart, so
    // that the one's epilogue, barrier wait and DMA issue fall into the other's MFMA stream instead of idling the
    // matrix pipe of that SIMD (with all eight waves in lock step the epilogue alone cost 1.3 of 4.9 ms: 64 rows per
    // block tile mean an epilogue every 6 stages).  A ring buffer is re-filled once the lagging group has read it.
    static constexpr int LAG = (LAGGED_ && NQT == 8) ? (KSTEPS / KS + 1) / 2 : 0;  // half a block tile
    static constexpr int D = NBUF - 1 - LAG;           // stages the DMA issue runs ahead of the leading group
    static constexpr int P = P_;                       // k-steps the fragment reads run ahead of the MFMAs (4 RTW (P + 1) registers)
    static constexpr int kStageBytes = F * 1024;
    // metadata buffers: a block tile's norms are issued with its first stage — D stages before the leading group multiplies
    // that stage — and read last by the lagging group's epilogue S + LAG stages later
    static constexpr int MB = (D + S + LAG - 1) / S + 2;
    static constexpr int kMetaSlot = (RTB * 32 + 63) / 64 * 64;    // floats per metadata buffer: whole 64-lane dword DMAs
    static constexpr int kMetaOff = NBUF * kStageBytes;            // [MB][kMetaSlot] floats: 1/|x| (or |x|^2)
    static constexpr int kMetaBytes = MB * kMetaSlot * 4;
    static constexpr int kDeadOff = kMetaOff + kMetaBytes;         // [MB][64] tombstone words (a 64-lane dword DMA)
    static constexpr int kDeadBytes = MB * 256;
    static constexpr int kQOff = kDeadOff + kDeadBytes;            // [256] 1/|q| (or |q|^2), [256] threshold
    static constexpr int kQBytes = 3 * 256 * 4;                  // + the fast-reject bound of the cosine form
    static constexpr int kFlushOff = kQOff + kQBytes;              // FlushArgs: what only a flush needs of the kernel arguments
    static constexpr int kStageOff = kFlushOff + 64;               // per wave: 64-bin histogram + kStageCap staged rows
    static constexpr int kStageFit = ((160 * 1024 - kStageOff) / kWaves - 256) / 12 / 16 * 16;  // 12 bytes per staged row
    static constexpr int kStageCap = kStageFit > 1024 ? 1024 : kStageFit;
    static constexpr int kWaveStage = 256 + kStageCap * 12;
    static constexpr int kLdsBytes = kStageOff + kWaves * kWaveStage;
    static_assert(KSTEPS % KS == 0 && KS % 8 == 0, "whole stages per block tile; fragment t = w + 8 j of a stage");
    static_assert(F % kWaves == 0, "every wave issues the same number of DMA instructions per stage");
    static_assert(IPS * (D - 1) < 64, "counted vmcnt wait must fit 6 bits");
    static_assert(D >= 3, "stage g + 1 must have landed and at least one more be in flight while stage g is multiplied");
    static_assert(NQT == 8 || LAG == 0, "the lag pairs wave w with wave w + 4: one query tile per wave");
    static_assert(P >= 1 && P <= KS, "fragment reads run ahead into the NEXT stage at most");
    static_assert(KSTEPS * 4 <= 192, "the B fragments must fit the register budget");
    static_assert(kStageCap >= 64, "no room for the staging area (one accumulator row of a wave may pass 64 rows)");
    static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
};

// DBG bits (timing ablations, ORAMA_K2C_DBG): 1 no MFMA, 2 no DMA, 8 no LDS fragment reads, 32 no epilogue, 64 = the fast reject
// of a UNIT-NORM store (no norm reads from LDS, no multiplies: 16 maxima against the column's bound; right answers on rows of
// norm 1 only — the ceiling experiment of VERDICT r04 next #5b, scripts/k2q_unit_norm_probe.py), 16 = block 0 records
// s_memtime stamps per step (waves 0 and 4: arrival at the barrier, release, DMA issued, stage multiplied, epilogue done —
// scripts/k2q_trace.py)
template <class C, int DBG, bool DENSE, bool L2>
__global__ __launch_bounds__(C::kThreads) void vec_scan_f16_qs_kernel(F16ScanArgs a, const char* __restrict__ bfrag,
                                                                      const float* __restrict__ qinv, uint64_t tile_bytes,
                                                                      unsigned long long* __restrict__ trace) {
    constexpr bool TRACE = (DBG & 16) != 0;
    constexpr int KSTEPS = C::KSTEPS, KS = C::KS, NBUF = C::NBUF, S = C::S, RTB = C::RTB, RTW = C::RTW, IPS = C::IPS, D = C::D;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = (int)uniform_u32((uint32_t)(tid >> 6));
    const int c = w % C::NQT;   // query tile of this wave
    const int grp = w / C::NQT;  // row group
    // (tile indices and stage counters are 32-bit — a store holds < 2^32 rows — and only byte addresses 64-bit: the kernel
    // lives at the edge of the scalar register file as well)
    const uint32_t t_first = (uint32_t)(a.row_begin >> 5);
    const uint32_t t_end = (uint32_t)((a.row_end + 31) >> 5);            // row tiles [t_first, t_end)
    const uint32_t n_bt = (t_end - t_first + RTB - 1) / RTB;            // block tiles
    if (blockIdx.x >= n_bt) return;
    const uint32_t my_bt = (n_bt - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t total = my_bt * S;                                   // stages this workgroup runs
    const float* inv_lds = reinterpret_cast<const float*>(lds + C::kMetaOff);
    const uint32_t* dead_lds = reinterpret_cast<const uint32_t*>(lds + C::kDeadOff);
    float* q_lds = reinterpret_cast<float*>(lds + C::kQOff);
    for (uint32_t i = tid; i < 256u; i += C::kThreads) {
        q_lds[i] = i < a.q ? qinv[i] : 0.0f;
        q_lds[256 + i] = (a.tau && i < a.q) ? a.tau[i] : -__builtin_huge_valf();
        // Fast-reject bound of the cosine form.  A row passes iff fma(-s, fl(n qi), 1) < tau (s = the dot product, n = 1/|x|,
        // qi = 1/|q|).  That implies s n qi > 1 - tau - 4e-7 (one rounding of the fma, |tau| <= 2), hence — two more
        // roundings, fl(n qi) and fl(s n) — fl(s n) > (1 - tau - 1e-6) / qi for qi > 0: a tile none of whose rows reaches
        // that bound (lowered by another 1e-6 relative for the division's own rounding) has no passing row.  One multiply
        // and a running maximum per element instead of multiply, fma and minimum: the epilogue's arithmetic is VALU work
        // beside a matrix pipe that already takes the package to its power limit (profiles/r03_power_energy.md).
        float bound = __builtin_huge_valf();  // nothing passes: a column beyond the batch
        if (a.tau && i < a.q) {
            const float qv = qinv[i], tv = a.tau[i];
            if (qv > 0.0f) {
                bound = (1.0f - tv - 1e-6f) / qv;
                bound -= fabsf(bound) * 1e-6f;
            } else {
                bound = 1.0f < tv ? -__builtin_huge_valf() : __builtin_huge_valf();  // a zero query is at distance 1 from every row
            }
        }
        q_lds[512 + i] = bound;
    }
    QsFlushArgs* fl = reinterpret_cast<QsFlushArgs*>(lds + C::kFlushOff);
    if (!DENSE && tid == 0) {
        fl->cand_dist = a.cand_dist;
        fl->cand_row = a.cand_row;
        fl->cand_count = a.cand_count;
        fl->cand_stride = a.cand_stride;
        fl->row_doc = a.row_doc;
        fl->allow = a.allow;
        fl->allow_bits = a.allow_bits;
        fl->no_appends = a.dbg & 2u;
    }

    // ---- the stationary operand: this wave's query tile, every k-step, straight into registers
    h8 bq[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
        bq[ks] = *reinterpret_cast<const h8*>(bfrag + ((size_t)c * KSTEPS + ks) * 1024 + (size_t)lane * 16);
    // The fragments are HERE before anything else is issued, and the compiler is told so: otherwise it keeps counting these
    // loads as pending and may plant `s_waitcnt vmcnt(N)` in front of their first uses inside the K loop, where vmcnt also
    // counts the DMA of the prefetch ring it knows nothing about: each of those waits would drain the ring.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(bq[ks]));

    // ---- loader state: fragment t = w + 8 j of a stage is row tile t / KS of the block tile, k-step t % KS of the stage
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t vlane = (uint32_t)lane * 16;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    // fragment t = w + 8 j of a stage: row tile t / KS = j / (KS / 8) of the block tile, k-step t % KS = w + 8 (j % (KS / 8))
    constexpr int JK = KS / 8;
    const bool partial_last = ((t_end - t_first) % RTB) != 0;
    uint32_t ld_tile0 = t_first + blockIdx.x * RTB;  // first row tile of the block tile under the load cursor
    // address of (that tile, k-step w of the cursor's stage); row tile j of the block tile is j * tile_bytes further
    uint64_t ld_ptr = (uint64_t)(uintptr_t)base + (uint64_t)ld_tile0 * tile_bytes + (uint64_t)w * 1024;
    const uint32_t bt_tiles = gridDim.x * RTB;
    const uint64_t bt_skip = (uint64_t)bt_tiles * tile_bytes - (uint64_t)S * KS * 1024;  // from the end of a block tile's row to the next one
    uint32_t ld_s = 0, ld_mb = 0;  // stage inside the block tile; its metadata buffer
    auto issue_stage = [&](uint32_t buf) {
        if (DBG & 2) return;
        const uint32_t lbuf = lds_base + buf * (uint32_t)C::kStageBytes + (uint32_t)w * 1024;
        if (ld_s == 0 && w < C::NG) {
            // first stage of a block tile: also the 1/|x| of its rows (64 per row group, one dword per lane; rows past
            // the end of the store read the zero-initialised padding of the array) and the tombstone words of its tiles
            qs_dma4((uint64_t)(uintptr_t)(a.inv_norm + (uint64_t)(ld_tile0 + (uint32_t)w * RTW) * 32), vlane >> 2,
                    lds_base + C::kMetaOff + (ld_mb * (uint32_t)C::kMetaSlot + (uint32_t)w * RTW * 32u) * 4u);
            if (a.dead && w == 0) {
                // lane l reads the word of row tile min(l, RTB - 1, last tile of the store)
                const uint32_t last = t_end - 1 - ld_tile0 < (uint32_t)(RTB - 1) ? t_end - 1 - ld_tile0 : (uint32_t)(RTB - 1);
                uint32_t tl = (uint32_t)lane;
                tl = tl < last ? tl : last;
                qs_dma4((uint64_t)(uintptr_t)(a.dead + ld_tile0), tl * 4u, lds_base + C::kDeadOff + ld_mb * 256);
            }
        }
        if (RTB > 1 && partial_last && ld_tile0 + RTB > t_end) {
            // the last block tile holds fewer row tiles: the missing ones re-read its last valid tile (masked in the epilogue)
            const uint32_t last = t_end - 1 - ld_tile0;
#pragma unroll
            for (int j = 0; j < IPS; ++j)
                qs_dma16_nt(ld_ptr + (uint64_t)((uint32_t)(j / JK) < last ? (uint32_t)(j / JK) : last) * tile_bytes + (uint64_t)(j % JK) * 8192,
                            vlane, lbuf + (uint32_t)j * 8192);
        } else {
#pragma unroll
            for (int j = 0; j < IPS; ++j)
                qs_dma16_nt(ld_ptr + (uint64_t)(j / JK) * tile_bytes + (uint64_t)(j % JK) * 8192, vlane, lbuf + (uint32_t)j * 8192);
        }
        ld_ptr += (uint64_t)KS * 1024;
        if (++ld_s == (uint32_t)S) {
            ld_s = 0;
            ld_ptr += bt_skip;
            ld_tile0 += bt_tiles;
            ld_mb = ld_mb == C::MB - 1 ? 0 : ld_mb + 1;
        }
    };

    // ---- filter mode: rows under the thresholds are staged per wave in LDS and appended in bulk with ONE global atomic
    // instruction per flush (K2's scheme, vec_f16.hip).  A wave owns 32 columns.  Everything here is written to need few
    // registers: the 192 registers of query fragments stay live through the epilogue.
    constexpr uint32_t kCap = C::kStageCap;
    const uint32_t stage_off = uniform_u32((uint32_t)C::kStageOff + (uint32_t)w * (uint32_t)C::kWaveStage);
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds + stage_off);
    uint32_t* st_dist = hist + 64;
    uint32_t* st_row = st_dist + kCap;
    uint32_t* st_meta = st_row + kCap;  // column (bits 0..7) | 1 + rank among the kept rows of its column (bits 8..), 0 = dropped
    const uint32_t col0 = (uint32_t)c * 32;
    uint32_t staged = 0;  // wave-uniform
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    auto bin_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto bin_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto flush = [&]() {
        wave_fence();
        bin_store(&hist[lane], 0u);
        wave_fence();
        const uint64_t* row_doc = reinterpret_cast<const uint64_t*>(qs_uniform_u64((uint64_t)(uintptr_t)fl->row_doc));
        const uint64_t* allow = reinterpret_cast<const uint64_t*>(qs_uniform_u64((uint64_t)(uintptr_t)fl->allow));
        const uint64_t allow_bits = qs_uniform_u64(fl->allow_bits);
        const bool appends = uniform_u32(fl->no_appends) == 0;
#pragma unroll 1
        for (uint32_t i = (uint32_t)lane; i < staged; i += 64) {
            bool keep = appends;
            if (allow) {
                const uint64_t doc = row_doc[st_row[i]];
                keep = keep && doc < allow_bits && ((allow[doc >> 6] >> (doc & 63)) & 1ull);
            }
            const uint32_t cl = st_meta[i] & 0xffu;
            st_meta[i] = keep ? cl | ((1u + __hip_atomic_fetch_add(&hist[cl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)) << 8) : cl;
        }
        wave_fence();
        const uint32_t mine = bin_load(&hist[lane]);
        // (the returned value is waited for with vmcnt, in order: this also drains the wave's share of the prefetch ring —
        // a flush happens a few times per launch and wave)
        uint32_t* cand_count = reinterpret_cast<uint32_t*>(qs_uniform_u64((uint64_t)(uintptr_t)fl->cand_count));
        bin_store(&hist[lane], (mine && lane < 32) ? atomicAdd(&cand_count[col0 + lane], mine) : 0u);
        wave_fence();
        float* cand_dist = reinterpret_cast<float*>(qs_uniform_u64((uint64_t)(uintptr_t)fl->cand_dist));
        uint32_t* cand_row = reinterpret_cast<uint32_t*>(qs_uniform_u64((uint64_t)(uintptr_t)fl->cand_row));
        const uint64_t cand_stride = qs_uniform_u64(fl->cand_stride);
#pragma unroll 1
        for (uint32_t i = (uint32_t)lane; i < staged; i += 64) {
            const uint32_t meta = st_meta[i];
            if (meta >> 8) {
                const uint32_t cl = meta & 0xffu;
                const uint64_t pos = (uint64_t)(col0 + cl) * cand_stride + bin_load(&hist[cl]) + ((meta >> 8) - 1u);
                cand_dist[pos] = __uint_as_float(st_dist[i]);
                cand_row[pos] = st_row[i];
            }
        }
        wave_fence();
        staged = 0;
    };

    f16v acc[RTW];
    // Filter mode: `start` = 16 i + r of the first accumulator row still to be looked at; returns 16 RTW when the block tile is
    // done, else the position at which the staging area ran full — the caller flushes (at a point where nothing of the
    // epilogue is live: a flush inlined here needs more registers than the 192 fragment registers leave) and calls again.
    auto epilogue = [&](uint32_t tile0, uint32_t mb, uint32_t start) -> uint32_t {
        const uint32_t hi = (lane >> 5) ? 4u : 0u;
        const uint32_t cl = (uint32_t)(lane & 31);  // column inside this wave's query tile
        const uint32_t col = col0 + cl;
        const bool live = col < a.q;
        // this lane's column: 1/|q| (or |q|^2) and its threshold (-inf beyond the batch: such a column passes nothing) —
        // LDS reads in flight together with the first tile's norms: ONE exposed round trip
        const float qi = q_lds[col];
        const float tau = q_lds[256 + col];
        // cosine: 1 - s (1/|x|)(1/|q|);  L2: (|q|^2 + |x|^2) - 2 s   (nrm / qi hold the squared norms then) — as the fused
        // operations the compiler contracts the plain expressions to (vec_f16.hip).  L2 is a template parameter: as a run
        // time flag the compiler evaluated BOTH forms per element and selected.
        auto dist_of = [&](float dot, float n, float qv) -> float {
            if constexpr (L2) return __builtin_fmaf(-2.0f, dot, qv + n);
            else return __builtin_fmaf(-dot, n * qv, 1.0f);
        };
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
            const uint32_t tl = (uint32_t)(grp * RTW + i);
            const uint32_t tile = tile0 + tl;
            if (tile >= t_end) continue;  // wave-uniform
            if (!DENSE && (uint32_t)(i + 1) * 16u <= start) continue;  // done before the flush
            // ONE LDS round trip per tile: its tombstone word and this lane's 16 norms — accumulator row r of the lane is row
            // (r & 3) + 8 (r >> 2) + hi of the tile: four 16-byte reads, all in flight together
            const float* nrm = inv_lds + mb * (uint32_t)C::kMetaSlot + tl * 32 + hi;
            const uint32_t dead_word = a.dead ? dead_lds[mb * 64 + tl] : 0u;
            f4 nv[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                if constexpr ((DBG & 64) != 0 && !DENSE) nv[g4] = f4{1.0f, 1.0f, 1.0f, 1.0f};  // (the slow path reads nrm[] itself)
                else nv[g4] = *reinterpret_cast<const f4*>(nrm + 8 * g4);
            }
            const bool full = (uint64_t)tile * 32 + 32 <= a.row_end;
            if constexpr (DENSE) {
                if (!live) continue;
                // this lane's rows of the tile start at out[0]; row (e + 8 g4) + hi sits at out[e + 8 g4]
                float* out = a.out_dense + ((uint64_t)col * a.dense_stride + ((uint64_t)tile * 32 + hi - a.row_begin));
                const uint32_t left = full ? 32u : (uint32_t)(a.row_end - (uint64_t)tile * 32);
                uint32_t gone = (dead_word | (left >= 32u ? 0u : ~((1u << left) - 1u))) >> hi;  // bit ri: dead or past the end
                if (a.allow) {
                    // the filter as a rolled pre-pass over the lane's 16 rows (a DocumentId and a bitmap word at a time):
                    // unrolled into the stores below it needs more registers than the fragments leave
#pragma unroll 1
                    for (uint32_t r = 0; r < 16; ++r) {
                        const uint32_t ri = (r & 3) + 8 * (r >> 2);
                        if ((gone >> ri) & 1u) continue;
                        const uint64_t doc = a.row_doc[(uint64_t)tile * 32 + hi + ri];
                        if (doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull)) gone |= 1u << ri;
                    }
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t ri = (uint32_t)(e + 8 * g4);
                        if (!full && ri + hi >= left) continue;
                        out[ri] = ((gone >> ri) & 1u) ? __builtin_nanf("") : dist_of(acc[i][4 * g4 + e], nv[g4][e], qi);
                    }
                }
                continue;
            } else {
                // fast reject, one ballot: the minimum of the 16 distances against the threshold (L2) / the maximum of the 16
                // similarities s n against the column's bound (cosine: see q_lds[512 ..])
                if constexpr (L2) {
                    float best = __builtin_huge_valf();
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) best = fminf(best, dist_of(acc[i][4 * g4 + e], nv[g4][e], qi));
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(best < tau) == 0, 1)) continue;
                } else {
                    float top = -__builtin_huge_valf();
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) top = fmaxf(top, acc[i][4 * g4 + e] * nv[g4][e]);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(top > q_lds[512 + col]) == 0, 1)) continue;
                }
                // slow path: bit r of m = accumulator row r of this lane passes (recomputed through an operand the
                // optimiser cannot see through: nothing but the norms is kept from the fast path)
                const uint32_t left = full ? 32u : (uint32_t)(a.row_end - (uint64_t)tile * 32);  // rows of the tile inside the store
                const uint32_t alive = (~dead_word & (left >= 32u ? ~0u : ((1u << left) - 1u))) >> hi;
                float qi_s = qi;
                asm volatile("" : "+v"(qi_s));
                uint32_t m = 0;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        m |= ((dist_of(acc[i][4 * g4 + e], nv[g4][e], qi_s) < tau ? 1u : 0u) & (alive >> (e + 8 * g4))) << (4 * g4 + e);
                // the accumulator rows somebody passes (usually one or two of the 16): OR over the wave, then only those
                uint32_t any = wave_or_u32(m);
                if (start > (uint32_t)i * 16u) any &= ~0u << (start - (uint32_t)i * 16u);  // resuming after a flush
#pragma unroll 1
                while (any) {
                    const uint32_t r = (uint32_t)__builtin_ctz(any);  // a wave-uniform index into the accumulators
                    const bool mine = (m >> r) & 1u;
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
                    const uint32_t n_pass = (uint32_t)__popcll(bal);
                    if (staged + n_pass > kCap) return (uint32_t)i * 16u + r;  // no room: flush, then resume here
                    any &= any - 1u;
                    const uint32_t ri = (r & 3u) + 8u * (r >> 2) + hi;
                    if (mine) {
                        const uint32_t pos =
                            staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        st_dist[pos] = __float_as_uint(dist_of(acc[i][r], nrm[(r & 3u) + 8u * (r >> 2)], qi_s));
                        st_row[pos] = tile * 32u + ri;
                        st_meta[pos] = cl;
                    }
                    staged = uniform_u32(staged + n_pass);
                }
            }
        }
        return 16u * RTW;
    };

    // ---- prologue: stages 0 .. D-1 into buffers 0 .. D-1
    constexpr int LAG = C::LAG;
    const bool lagging = LAG > 0 && w >= 4;  // this wave runs LAG stages behind its SIMD's other wave
    uint32_t issued = 0;
    for (; issued < (uint32_t)D && issued < total; ++issued) issue_stage(issued);
    __syncthreads();  // q_lds is visible (the DMA in flight is not waited for: asm statements are invisible to the fence)

    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // One STEP per stage barrier; in step t the leading waves multiply stage t, the lagging ones stage t - LAG.  Before
    // barrier t every wave has waited for its share of stages <= t + 1, so behind it stages <= t + 1 are in LDS for
    // everybody (the fragment reads run P k-steps ahead of the MFMAs, also across the barrier: no LDS round trip is
    // exposed after it) and stage t - 1 - LAG has been read by everybody: its buffer takes stage t + D, issued during
    // step t — by the leading waves early in their stage, by the lagging ones late, so that the two waves of a SIMD are
    // not stuck in their DMA issue (100+ cycles per instruction under back-pressure) at the same time.
    constexpr int P = C::P;
    uint32_t step = 0;                   // barriers passed
    uint32_t buf = 0, ibuf = D % NBUF;   // ring position of the next stage to multiply / to issue
    uint32_t mb = 0;                     // metadata buffer of the block tile being multiplied
    bool owe = false;                    // the stage of the current step is still to be issued
    const bool tracing = TRACE && blockIdx.x == 0 && (w & 3) == 0 && lane == 0;
    auto stamp = [&](uint32_t slot) {  // of the step whose barrier was passed last
        if (TRACE && tracing && step >= 1 && step <= 1024) trace[(size_t)(step - 1) * 16 + (w >> 2) * 8 + slot] = __builtin_amdgcn_s_memtime();
    };
    auto step_sync = [&]() {
        if (TRACE && tracing && step < 1024) trace[(size_t)step * 16 + (w >> 2) * 8 + 0] = __builtin_amdgcn_s_memtime();
        // this wave's share of stages <= step + 1 has landed: everything but its newest D - 2 stages (tail: everything)
        if (issued - step == (uint32_t)D) qs_wait_vmcnt<IPS * (D - 2)>();
        else qs_wait_vmcnt<0>();
        if (TRACE && tracing && step < 1024) trace[(size_t)step * 16 + (w >> 2) * 8 + 5] = __builtin_amdgcn_s_memtime();
        qs_stage_barrier();
        ++step;
        stamp(1);
        owe = issued < total;
    };
    auto issue_owed = [&]() {
        if (owe) {
            issue_stage(ibuf);
            ibuf = ibuf == NBUF - 1 ? 0 : ibuf + 1;
            ++issued;
            owe = false;
            stamp(2);
        }
    };
    if (lagging) {
#pragma unroll 1
        for (int i = 0; i < LAG; ++i) {  // the leading waves are LAG stages in
            step_sync();
            issue_owed();
        }
    }
    auto finish_tile = [&](uint32_t tile_first, uint32_t tile_mb) {
        if (!(DBG & 32)) {
            uint32_t at = 0;
            while ((at = epilogue(tile_first, tile_mb, at)) < 16u * RTW) flush();
            if (!DENSE && staged * 2u > kCap) flush();  // room for the usual few rows: the next block tile does not re-run
        } else {  // ablation builds: keep every accumulator alive
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < RTW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][r];
            if (sum == 12345.678f) a.cand_count[0] = 1;
        }
    };
    const uint32_t la_wave = (uint32_t)((grp * RTW) * KS) * 1024u + (uint32_t)lane * 16u;
    uint32_t tile0 = t_first + blockIdx.x * RTB;  // first row tile of the block tile being multiplied
    for (uint32_t bi = 0; bi < my_bt; ++bi, tile0 += bt_tiles) {
        h8 fa[P + 1][RTW];
        const char* la_cur = nullptr;   // fragments of the stage being multiplied / of the one after it
        const char* la_next = lds + (size_t)buf * C::kStageBytes + la_wave;
        auto load_frags = [&](int kk) {  // k-step kk of the block tile -> register set kk % (P + 1)
            if (DBG & 8) return;
            const char* la = (kk / KS == (kk - P < 0 ? 0 : kk - P) / KS || kk < P) ? la_cur : la_next;
#pragma unroll
            for (int i = 0; i < RTW; ++i)
                fa[kk % (P + 1)][i] = *reinterpret_cast<const h8*>(la + (size_t)(i * KS + kk % KS) * 1024);
        };
        auto next_stage = [&]() {
            la_cur = la_next;
            buf = buf == NBUF - 1 ? 0 : buf + 1;
            la_next = lds + (size_t)buf * C::kStageBytes + la_wave;
        };
        // first stage of the block tile.  The previous block tile's epilogue runs HERE, behind the barrier: the SIMD's other
        // wave is LAG stages away from its own epilogue and feeds the matrix pipe meanwhile (in front of the barrier it
        // would only make everybody wait).  (Outside the unrolled k loop: with the epilogue inside its body the loop is no
        // longer unrolled and the fragment registers become an array in scratch memory.)
        step_sync();
        if (bi > 0) {
            finish_tile(tile0 - bt_tiles, mb == 0 ? C::MB - 1 : mb - 1);
            stamp(4);
        }
        next_stage();
#pragma unroll
        for (int k0 = 0; k0 < P && k0 < KSTEPS; ++k0) load_frags(k0);  // (re)start the pipeline
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (kk % KS == 0 && kk > 0) {
                step_sync();
                next_stage();
            }
            if (kk + P < KSTEPS) load_frags(kk + P);
            if (kk % KS == 1 && !lagging) issue_owed();
            if (kk % KS == KS - 3 && lagging) issue_owed();
#pragma unroll
            for (int i = 0; i < RTW; ++i) {
                if (DBG & 1) {
                    if (kk == 0) acc[i] = zero;
                    asm volatile("" ::"v"(fa[kk % (P + 1)][i]), "v"(bq[kk]));
                } else if (kk == 0) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk % (P + 1)][i], bq[kk], zero, 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk % (P + 1)][i], bq[kk], acc[i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the k-steps in this order: reads P ahead, then the MFMAs
            if (TRACE && kk % KS == KS - 1) stamp(3);
        }
        mb = mb == C::MB - 1 ? 0 : mb + 1;
    }
    finish_tile(tile0 - bt_tiles, mb == 0 ? C::MB - 1 : mb - 1);  // the last block tile
    if (LAG > 0 && !lagging) {
#pragma unroll 1
        for (int i = 0; i < LAG; ++i) {  // the lagging waves' last LAG stages
            step_sync();
            issue_owed();
        }
    }
    if (!DENSE && staged) flush();
}

template <class C, int DBG, bool DENSE, bool L2>
int qs_launch_one(const F16ScanArgs& a, const char* bfrag, const float* qinv, uint32_t blocks, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_qs_kernel<C, DBG, DENSE, L2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    unsigned long long* trace = nullptr;
    if (DBG & 16) {
        const char* e = orama::dev_env("ORAMA_K2D_TRACE");  // device pointer of >= 128 KiB (hex), set by the probe script
        if (e) trace = reinterpret_cast<unsigned long long*>(std::strtoull(e, nullptr, 16));
        ORAMA_REQUIRE(trace, "trace build needs ORAMA_K2D_TRACE");
    }
    hipLaunchKernelGGL((vec_scan_f16_qs_kernel<C, DBG, DENSE, L2>), dim3(blocks), dim3(C::kThreads), C::kLdsBytes, stream, a, bfrag,
                       qinv, f16_tile_bytes(a.dim), trace);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

template <class C, int DBG>
int qs_launch(orama_ctx* ctx, const F16ScanArgs& a, const char* bfrag, const float* qinv, hipStream_t stream) {
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    uint64_t blocks = (tiles + C::RTB - 1) / C::RTB;
    if (blocks > (uint64_t)ctx->compute_units) blocks = (uint64_t)ctx->compute_units;
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;
    if (a.out_dense) {
        return l2 ? qs_launch_one<C, DBG, true, true>(a, bfrag, qinv, (uint32_t)blocks, stream)
                  : qs_launch_one<C, DBG, true, false>(a, bfrag, qinv, (uint32_t)blocks, stream);
    }
    return l2 ? qs_launch_one<C, DBG, false, true>(a, bfrag, qinv, (uint32_t)blocks, stream)
              : qs_launch_one<C, DBG, false, false>(a, bfrag, qinv, (uint32_t)blocks, stream);
}

// 129..256 queries: 8 query tiles, block tile 64 rows, ring of 8 stages x 16 KiB.  (The geometry for <= 128 queries — 4
// query tiles, two row groups, block tile 128 rows, QsCfg<KSTEPS, 4, 8, 4> — measured behind K2d's 256 x 128 block tile and
// needs 5 more registers than the file has at 768 dimensions: those batches stay on K2d.)
// One row tile per wave (block tile 32 rows x 256 queries): 192 fragment registers + 16 accumulators leave room for fragment
// reads two k-steps ahead and for an epilogue that holds a tile's 16 norms at once — with two row tiles per wave (32
// accumulators) every variant of the kernel lived on the last register and spilled into scratch memory, whose reloads
// drain the prefetch ring (vmcnt is in order).  Stages of 16 k-steps where kpad allows (16 KiB, 3 per block tile at 768
// dimensions), else of 8.
template <int KSTEPS>
using QsWide = QsCfg<KSTEPS, 8, (KSTEPS % 16 == 0 ? 16 : 8), (KSTEPS == 48 ? 9 : (KSTEPS % 16 == 0 ? 8 : 14))>;
template <int KSTEPS>
using QsLock = QsCfg<KSTEPS, 8, (KSTEPS % 16 == 0 ? 16 : 8), (KSTEPS % 16 == 0 ? 8 : 14), false>;  // all waves in lock step (ORAMA_QS_LAG=0)

template <int KSTEPS>
int qs_dispatch(orama_ctx* ctx, const F16ScanArgs& a, const char* bfrag, const float* qinv, hipStream_t stream, int dbg) {
    if constexpr (KSTEPS == 48) {  // ablation builds of the C5 shape (timing only)
        if (dbg && !a.out_dense) {
            switch (dbg) {
                case 1: return qs_launch<QsWide<KSTEPS>, 1>(ctx, a, bfrag, qinv, stream);    // DMA + LDS reads, no MFMA
                case 2: return qs_launch<QsWide<KSTEPS>, 2>(ctx, a, bfrag, qinv, stream);    // LDS reads + MFMA + barriers
                case 9: return qs_launch<QsWide<KSTEPS>, 9>(ctx, a, bfrag, qinv, stream);    // DMA only
                case 32: return qs_launch<QsWide<KSTEPS>, 32>(ctx, a, bfrag, qinv, stream);  // everything but the epilogue
                case 16: return qs_launch<QsWide<KSTEPS>, 16>(ctx, a, bfrag, qinv, stream);  // full kernel + timeline stamps
                case 48: return qs_launch<QsWide<KSTEPS>, 48>(ctx, a, bfrag, qinv, stream);  // no epilogue + stamps
                case 34: return qs_launch<QsWide<KSTEPS>, 34>(ctx, a, bfrag, qinv, stream);  // LDS reads + MFMA + barriers, no epilogue
                case 40: return qs_launch<QsWide<KSTEPS>, 40>(ctx, a, bfrag, qinv, stream);  // DMA + MFMA, no LDS reads, no epilogue
                case 42: return qs_launch<QsWide<KSTEPS>, 42>(ctx, a, bfrag, qinv, stream);  // MFMA + barriers only
                case 64: return qs_launch<QsWide<KSTEPS>, 64>(ctx, a, bfrag, qinv, stream);  // unit-norm fast reject (rows of norm 1)
                default: break;
            }
        }
    }
    static const bool lag = [] { const char* e = orama::dev_env("ORAMA_QS_LAG"); return !e || std::atoi(e) != 0; }();
    if constexpr (KSTEPS == 48) {
        if (!lag) {
            if (dbg == 32 && !a.out_dense) return qs_launch<QsLock<KSTEPS>, 32>(ctx, a, bfrag, qinv, stream);
            if (dbg == 34 && !a.out_dense) return qs_launch<QsLock<KSTEPS>, 34>(ctx, a, bfrag, qinv, stream);
            if (dbg == 42 && !a.out_dense) return qs_launch<QsLock<KSTEPS>, 42>(ctx, a, bfrag, qinv, stream);
            return qs_launch<QsLock<KSTEPS>, 0>(ctx, a, bfrag, qinv, stream);
        }
    }
    return qs_launch<QsWide<KSTEPS>, 0>(ctx, a, bfrag, qinv, stream);
}

}  // namespace

bool vec_scan_f16_qs_supports(uint32_t dim, uint32_t q) { return f16_kpad(dim) <= 768 && q > 128 && q <= kF16WideMaxQ; }

int launch_vec_scan_f16_qs(orama_ctx* ctx, const F16ScanArgs& a_in, void* d_query_frags, hipStream_t stream) {
    static const uint32_t k2dbg = [] { const char* e = orama::dev_env("ORAMA_K2_DBG"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    F16ScanArgs a = a_in;
    if (!a.out_dense) a.dbg = k2dbg & 2u;  // timing ablation: no candidate appends
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries && d_query_frags, "vec_scan_f16_qs: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF16WideMaxQ, "vec_scan_f16_qs: q=%u outside [1, %u]", a.q, kF16WideMaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f16_qs: bad row range");
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count), "vec_scan_f16_qs: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f16_qs: filter needs row_doc");
    ORAMA_REQUIRE(vec_scan_f16_qs_supports(a.dim, a.q), "vec_scan_f16_qs: %u dimensions x %u queries outside the kernel's envelope", a.dim, a.q);
    if (a.row_begin == a.row_end) return ORAMA_OK;
    const uint32_t ksteps = f16_kpad(a.dim) / 16;
    const char* bfrag = reinterpret_cast<const char*>(d_query_frags);
    const float* qinv = reinterpret_cast<const float*>(bfrag + (size_t)8 * ksteps * 1024);
    ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
    int dbg = 0;
    if (const char* e = orama::dev_env("ORAMA_K2C_DBG")) dbg = std::atoi(e);
    switch (ksteps) {
        case 8: return qs_dispatch<8>(ctx, a, bfrag, qinv, stream, dbg);
        case 16: return qs_dispatch<16>(ctx, a, bfrag, qinv, stream, dbg);
        case 24: return qs_dispatch<24>(ctx, a, bfrag, qinv, stream, dbg);
        case 32: return qs_dispatch<32>(ctx, a, bfrag, qinv, stream, dbg);
        case 40: return qs_dispatch<40>(ctx, a, bfrag, qinv, stream, dbg);
        case 48: return qs_dispatch<48>(ctx, a, bfrag, qinv, stream, dbg);
        default: break;
    }
    set_error("vec_scan_f16_qs: kpad %u not a multiple of 128", ksteps * 16);
    return ORAMA_ERR_INVALID;
}

}  // namespace orama
