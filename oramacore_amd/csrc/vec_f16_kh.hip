// vec_f16_kh.hip — K2h: the fp16 scan for wide query batches (129..256 queries per corpus pass; config C5) with the queries
// stationary in registers AND the K loop of every output tile split over two waves that hand the accumulators on.
//
// What bounds the wide scans (profiles/r03_k2q_ablation.md): K2d reads 1 KiB of LDS per MFMA and streams the query
// fragments through LDS a second time; K2q (one 32-query tile per wave, all k-steps) keeps the queries in registers but still
// reads 1 KiB per MFMA, has a single dependent MFMA chain per wave, a barrier every 16 MFMAs and an epilogue every 48 — its
// MFMA stream alone (no memory at all) takes 2.4 of its 4.3 ms.  A wave that owns TWO query tiles reuses every corpus
// fragment for two MFMAs (0.5 KiB of LDS per MFMA, two independent accumulator chains), but two tiles x 48 k-steps of
// fragments are 384 registers.  So the K loop is cut in two:
//
//   workgroup = 8 waves (2 per SIMD, <= 256 registers), one workgroup per CU, persistent over 32-row tiles
//   wave (cp, h): query tiles 2 cp, 2 cp + 1 (64 queries) x k-steps [h KH, (h + 1) KH), KH = kpad / 32 — its B fragments
//               (2 x KH x 4 = 192 registers at 768 dims) are loaded once and stay in VGPRs
//   slot n    : the h = 0 waves multiply row tile n over the FIRST half of k starting from C = 0 and leave their 2 x 16
//               accumulator registers in LDS (X[cp], 8 KiB); the h = 1 waves pick up X[cp] of row tile n - 1, continue over
//               the SECOND half of k, and own the epilogue.  Every output element is ONE accumulator chain over
//               ascending k-steps starting from 0 — the order of K2, K2c, K2d and K2q: distances are bit-identical to
//               solo queries (tests/test_vector_f16_gpu.py::test_wide_batches_equal_solo_queries runs every form).
//   one s_barrier per slot = per 48 MFMAs of a wave (96 per SIMD, ~3 000 cycles); wave w and w + 4 — h = 0 and h = 1 of
//               the same tile pair — share a SIMD: the h = 1 wave's epilogue (behind the barrier, before it picks up X)
//               runs while the h = 0 wave multiplies, and the h = 1 wave multiplies on while the h = 0 wave is done.
//   LDS       : the corpus ring in sub-stages of 8 k-steps (8 KiB), R = 14 of them — a slot reads 6 (tile n - 1 second half,
//               tile n first half), 8 are in flight (64 KiB per CU) — filled by global->LDS DMA (1 KiB per instruction, nt)
//               issued right behind the barrier, mostly by the h = 1 waves (their MFMAs start late anyway), waited for with
//               counted vmcnt; X (32 KiB, single-buffered: a per-pair flag orders "picked up" before "overwritten");
//               row metadata; the staging areas of the h = 1 waves.
//
// global->LDS traffic = the corpus, once; LDS reads = 0.5 KiB per MFMA + 8 KiB of hand-over per 96 MFMAs.
// kpad must be a multiple of 256 (whole sub-stages per half) and <= 768; other rows take K2q / K2d.
#include "vec_f16.hpp"

#if ORAMA_COMPARISON_KERNELS  // K2h is a comparison kernel (6 % slower than K2q): not part of the product library (see _build.py)

#include <cstdlib>
#include <type_traits>

#include "device_utils.hpp"
#include "vec_f16_async.hpp"

namespace orama {

namespace {

using namespace f16async;

template <int KSTEPS_, int R_ = 14, int P_ = 2>
struct KhCfg {
    static constexpr int KSTEPS = KSTEPS_, R = R_, P = P_;
    static constexpr int KH = KSTEPS / 2;        // k-steps per half
    static constexpr int SUB = KSTEPS / 8;       // sub-stages (8 k-steps, 8 KiB) per row tile
    static constexpr int HS = SUB / 2;           // per half
    static constexpr int kWaves = 8, kThreads = kWaves * 64;
    static constexpr int kSubBytes = 8 * 1024;
    static constexpr int MB = 8;                 // metadata buffers (row tiles between "norms issued" and "epilogue done")
    // DMA instructions a wave issues per slot: the slot's SUB sub-stages in order, two fragments of each — sub-stages
    // 0 .. SUB-2 by the four h = 1 waves, the last one by the four h = 0 waves
    static constexpr int IPS1 = 2 * (SUB - 1), IPS0 = 2;
    // sub-stages issued in slot n that slot n + 1 already reads (the rest have a slot more to land)
    static constexpr int kNeedNext = 3 * SUB - R < 0 ? 0 : (3 * SUB - R > SUB ? SUB : 3 * SUB - R);
    static constexpr int kAllow1 = kNeedNext >= SUB - 1 ? 0 : 2 * (SUB - 1 - kNeedNext);  // vmcnt an h = 1 wave may keep
    static constexpr int kAllow0 = kNeedNext >= SUB ? 0 : 2;                              // ... an h = 0 wave
    static constexpr int kRingOff = 0;
    static constexpr int kXOff = R * kSubBytes;                    // [4 pairs][2 accumulators][64 lanes][16 floats]
    static constexpr int kXBytes = 4 * 8 * 1024;
    static constexpr int kMetaOff = kXOff + kXBytes;               // [MB][64] floats: 1/|x| (or |x|^2) of the tile (+ 32 more)
    static constexpr int kDeadOff = kMetaOff + MB * 256;           // [MB][64] tombstone words (a 64-lane dword DMA)
    static constexpr int kQOff = kDeadOff + MB * 256;              // [256] 1/|q| (or |q|^2), [256] threshold
    static constexpr int kFlagOff = kQOff + 2048;                  // x_read[4] (64 bytes)
    static constexpr int kFlushOff = kFlagOff + 64;                // QsFlushArgs
    static constexpr int kStageOff = kFlushOff + 64;               // per h = 1 wave: 64-bin histogram + kStageCap staged rows
    static constexpr int kStageFit = ((160 * 1024 - kStageOff) / 4 - 256) / 12 / 16 * 16;  // 12 bytes per staged row
    static constexpr int kStageCap = kStageFit > 1024 ? 1024 : kStageFit;
    static constexpr int kWaveStage = 256 + kStageCap * 12;
    static constexpr int kLdsBytes = kStageOff + 4 * kWaveStage;
    static_assert(KSTEPS % 16 == 0, "whole sub-stages of 8 k-steps per half");
    static_assert(KH * 8 <= 192, "two query tiles' fragments of one half must fit the register budget");
    static_assert(R >= 2 * SUB, "the ring holds what a slot reads plus what the next one reads");
    static_assert(kStageCap >= 128, "no room for the staging area (one accumulator row of a wave may pass 128 rows)");
    static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
    static_assert(kAllow1 < 64, "vmcnt is a 6-bit counter");
};

// DBG bits (timing ablations, ORAMA_K2C_DBG): 1 no MFMA, 2 no DMA, 8 no LDS fragment reads, 32 no epilogue
template <class C, int DBG, bool DENSE, bool L2>
__global__ __launch_bounds__(C::kThreads) void vec_scan_f16_kh_kernel(F16ScanArgs a, const char* __restrict__ bfrag,
                                                                      const float* __restrict__ qinv, uint64_t tile_bytes) {
    constexpr int KSTEPS = C::KSTEPS, KH = C::KH, SUB = C::SUB, HS = C::HS, R = C::R, P = C::P;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = (int)uniform_u32((uint32_t)(tid >> 6));
    const int cp = w & 3;   // query tiles 2 cp, 2 cp + 1
    const int h = w >> 2;   // half of the K loop
    const uint32_t t_first = (uint32_t)(a.row_begin >> 5);
    const uint32_t t_end = (uint32_t)((a.row_end + 31) >> 5);                     // row tiles [t_first, t_end)
    if (t_first + blockIdx.x >= t_end) return;
    const uint32_t my_tiles = (t_end - t_first - blockIdx.x + gridDim.x - 1) / gridDim.x;  // tile i: t_first + blockIdx.x + i gridDim.x
    const uint32_t n_sub = my_tiles * SUB;                                        // sub-stages this workgroup reads
    const float* inv_lds = reinterpret_cast<const float*>(lds + C::kMetaOff);
    const uint32_t* dead_lds = reinterpret_cast<const uint32_t*>(lds + C::kDeadOff);
    float* q_lds = reinterpret_cast<float*>(lds + C::kQOff);
    uint32_t* x_read = reinterpret_cast<uint32_t*>(lds + C::kFlagOff);
    for (uint32_t i = tid; i < 256u; i += C::kThreads) {
        q_lds[i] = i < a.q ? qinv[i] : 0.0f;
        q_lds[256 + i] = (a.tau && i < a.q) ? a.tau[i] : -__builtin_huge_valf();  // a column beyond the batch passes nothing
    }
    if (tid < 16) x_read[tid] = 0u;
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

    // ---- the stationary operand: this wave's two query tiles over its half of k, straight into registers
    h8 bq[2][KH];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < KH; ++kk)
            bq[j][kk] = *reinterpret_cast<const h8*>(bfrag + ((size_t)(2 * cp + j) * KSTEPS + (size_t)h * KH + kk) * 1024 + (size_t)lane * 16);
    // The fragments are HERE before anything else is issued, and the compiler is told so: otherwise it keeps counting these
    // loads as pending and plants `s_waitcnt vmcnt(N)` in front of their first uses — inside the K loop, on every trip —
    // where vmcnt also counts the DMA of the prefetch ring it knows nothing about: each of those waits drains the ring.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) asm volatile("" : "+v"(bq[j][kk]));

    // ---- loader state: the cursor walks the sub-stages q = SUB i + j (row tile i, k-steps 8 j .. 8 j + 7) in order
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t vlane = (uint32_t)lane * 16;
    const uint64_t tile_step = (uint64_t)gridDim.x * tile_bytes;
    uint32_t iq = 0;         // next sub-stage to issue
    uint32_t iq_j = 0;       // its position inside its row tile
    uint32_t iq_slot = 0;    // its ring slot
    uint32_t iq_tile = t_first + blockIdx.x;  // its row tile
    uint32_t iq_mb = 0;      // metadata buffer of that tile
    uint64_t iq_ptr = (uint64_t)(uintptr_t)a.tiled + (uint64_t)iq_tile * tile_bytes;  // k-step 8 j of that tile
    // Issue fragments [ks0, ks0 + nks) of the cursor's sub-stage (this wave's share), then advance the cursor.  `meta`: this
    // wave also brings the tile's norms and tombstone word when the sub-stage is the tile's first.
    auto issue_sub = [&](uint32_t ks0, uint32_t nks, bool mine, bool meta) {
        if (!(DBG & 2) && iq < n_sub && mine) {
            if (meta && iq_j == 0) {
                // 64 floats from the tile's first row (the second 32 belong to the next tile: rows past the end of the store
                // read the zero-initialised padding of the array); lane l < 1: the tile's tombstone word
                qs_dma4((uint64_t)(uintptr_t)(a.inv_norm + (uint64_t)iq_tile * 32), vlane >> 2, lds_base + C::kMetaOff + iq_mb * 256);
                if (a.dead) qs_dma4((uint64_t)(uintptr_t)(a.dead + iq_tile), 0u, lds_base + C::kDeadOff + iq_mb * 256);
            }
            for (uint32_t i = 0; i < nks; ++i)
                qs_dma16_nt(iq_ptr + (uint64_t)(ks0 + i) * 1024, vlane, lds_base + iq_slot * (uint32_t)C::kSubBytes + (ks0 + i) * 1024);
        }
        ++iq;
        iq_slot = iq_slot == R - 1 ? 0 : iq_slot + 1;
        iq_ptr += 8 * 1024;
        if (++iq_j == (uint32_t)SUB) {
            iq_j = 0;
            iq_tile += gridDim.x;
            iq_ptr += tile_step - (uint64_t)SUB * 8 * 1024;
            iq_mb = iq_mb == C::MB - 1 ? 0 : iq_mb + 1;
        }
    };

    // ---- filter mode (h = 1 waves): rows under the thresholds are staged per wave in LDS and appended in bulk with ONE global
    // atomic instruction per flush (K2's scheme, vec_f16.hip).  A wave owns 64 columns.
    constexpr uint32_t kCap = C::kStageCap;
    const uint32_t stage_off = uniform_u32((uint32_t)C::kStageOff + (uint32_t)cp * (uint32_t)C::kWaveStage);
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds + stage_off);
    uint32_t* st_dist = hist + 64;
    uint32_t* st_row = st_dist + kCap;
    uint32_t* st_meta = st_row + kCap;  // column (bits 0..7) | 1 + rank among the kept rows of its column (bits 8..), 0 = dropped
    const uint32_t col0 = (uint32_t)cp * 64;
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
        bin_store(&hist[lane], mine ? atomicAdd(&cand_count[col0 + lane], mine) : 0u);
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

    f16v acc[2];
    // Epilogue of one row tile (h = 1 waves).  Filter mode: `start` = 16 j + r of the first accumulator row still to be looked
    // at; returns 32 when the tile is done, else the position at which the staging area ran full — the caller flushes (at a
    // point where nothing of the epilogue is live) and calls again.
    auto epilogue = [&](uint32_t tile, uint32_t mb, uint32_t start) -> uint32_t {
        const uint32_t hi = (lane >> 5) ? 4u : 0u;
        // cosine: 1 - s (1/|x|)(1/|q|);  L2: (|q|^2 + |x|^2) - 2 s   (nrm / qi hold the squared norms then) — as the fused
        // operations the compiler contracts the plain expressions to (vec_f16.hip)
        auto dist_of = [&](float dot, float n, float qv) -> float {
            if constexpr (L2) return __builtin_fmaf(-2.0f, dot, qv + n);
            else return __builtin_fmaf(-dot, n * qv, 1.0f);
        };
        // This lane's 16 norms — accumulator row r of the lane is row (r & 3) + 8 (r >> 2) + hi of the tile — are read four at
        // a time and used for BOTH query tiles at once: the epilogue runs beside the other wave's MFMAs, so its LDS round
        // trips are cheap, and the 192 fragment registers + 32 accumulators leave no room to hold 16 norms.
        const float* nrm = inv_lds + mb * 64 + hi;
        const uint32_t dead_word = a.dead ? dead_lds[mb * 64] : 0u;
        const bool full = (uint64_t)tile * 32 + 32 <= a.row_end;
        const uint32_t left = full ? 32u : (uint32_t)(a.row_end - (uint64_t)tile * 32);  // rows of the tile inside the store
        const uint32_t c0 = col0 + (uint32_t)(lane & 31);  // this lane's column of query tile 0; tile 1: + 32
        const float qi0 = q_lds[c0], qi1 = q_lds[c0 + 32];
        if constexpr (DENSE) {
            uint32_t gone = (dead_word | (left >= 32u ? 0u : ~((1u << left) - 1u))) >> hi;  // bit ri: dead or past the end
            if (a.allow) {
                // the filter as a rolled pre-pass over the lane's 16 rows (a DocumentId and a bitmap word at a time)
#pragma unroll 1
                for (uint32_t r = 0; r < 16; ++r) {
                    const uint32_t ri = (r & 3) + 8 * (r >> 2);
                    if ((gone >> ri) & 1u) continue;
                    const uint64_t doc = a.row_doc[(uint64_t)tile * 32 + hi + ri];
                    if (doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull)) gone |= 1u << ri;
                }
            }
            // this lane's rows of the tile start at out[0]; row (e + 8 g4) + hi sits at out[e + 8 g4]
            float* out0 = a.out_dense + ((uint64_t)c0 * a.dense_stride + ((uint64_t)tile * 32 + hi - a.row_begin));
            float* out1 = out0 + 32 * a.dense_stride;
            const bool live0 = c0 < a.q, live1 = c0 + 32 < a.q;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f4 nv = *reinterpret_cast<const f4*>(nrm + 8 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t ri = (uint32_t)(e + 8 * g4);
                    if (!full && ri + hi >= left) continue;
                    const bool g = (gone >> ri) & 1u;
                    if (live0) out0[ri] = g ? __builtin_nanf("") : dist_of(acc[0][4 * g4 + e], nv[e], qi0);
                    if (live1) out1[ri] = g ? __builtin_nanf("") : dist_of(acc[1][4 * g4 + e], nv[e], qi1);
                }
            }
            return 32u;
        } else {
            const float tau0 = q_lds[256 + c0], tau1 = q_lds[256 + c0 + 32];
            // fast reject: the minimum of the 16 distances of each query tile against its threshold
            float best0 = __builtin_huge_valf(), best1 = __builtin_huge_valf();
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f4 nv = *reinterpret_cast<const f4*>(nrm + 8 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    best0 = fminf(best0, dist_of(acc[0][4 * g4 + e], nv[e], qi0));
                    best1 = fminf(best1, dist_of(acc[1][4 * g4 + e], nv[e], qi1));
                }
            }
            const bool hit0 = __builtin_amdgcn_ballot_w64(best0 < tau0) != 0, hit1 = __builtin_amdgcn_ballot_w64(best1 < tau1) != 0;
            if (__builtin_expect(!(hit0 || hit1), 1)) return 32u;
            const uint32_t alive = (~dead_word & (left >= 32u ? ~0u : ((1u << left) - 1u))) >> hi;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!(j ? hit1 : hit0) || (uint32_t)(j + 1) * 16u <= start) continue;  // nothing passes / done before the flush
                // slow path: bit r of m = accumulator row r of this lane passes (recomputed through an operand the
                // optimiser cannot see through: nothing is kept from the fast path)
                float qi_s = j ? qi1 : qi0;
                asm volatile("" : "+v"(qi_s));
                const float tau = j ? tau1 : tau0;
                uint32_t m = 0;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f4 nv = *reinterpret_cast<const f4*>(nrm + 8 * g4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        m |= ((dist_of(acc[j][4 * g4 + e], nv[e], qi_s) < tau ? 1u : 0u) & (alive >> (e + 8 * g4))) << (4 * g4 + e);
                }
                // the accumulator rows somebody passes (usually one or two of the 16): OR over the wave, then only those
                uint32_t any = wave_or_u32(m);
                if (start > (uint32_t)j * 16u) any &= ~0u << (start - (uint32_t)j * 16u);  // resuming after a flush
#pragma unroll 1
                while (any) {
                    const uint32_t r = (uint32_t)__builtin_ctz(any);  // a wave-uniform index into the accumulators
                    const bool mine = (m >> r) & 1u;
                    const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
                    const uint32_t n_pass = (uint32_t)__popcll(bal);
                    if (staged + n_pass > kCap) return (uint32_t)j * 16u + r;  // no room: flush, then resume here
                    any &= any - 1u;
                    const uint32_t ri = (r & 3u) + 8u * (r >> 2) + hi;
                    if (mine) {
                        const uint32_t pos =
                            staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        st_dist[pos] = __float_as_uint(dist_of(acc[j][r], nrm[(r & 3u) + 8u * (r >> 2)], qi_s));
                        st_row[pos] = tile * 32u + ri;
                        // column inside this wave's 64 — from a freshly read lane id: kept live from the top of the kernel it
                        // is the value the register allocator spills, and a reload here waits for the whole prefetch ring
                        st_meta[pos] = (uint32_t)j * 32u + (qs_lane_id_now() & 31u);
                    }
                    staged = uniform_u32(staged + n_pass);
                }
            }
            return 32u;
        }
    };
    auto finish_tile = [&](uint32_t tile, uint32_t mb) {
        if (!(DBG & 32)) {
            uint32_t at = 0;
            while ((at = epilogue(tile, mb, at)) < 32u) flush();
            if (!DENSE && staged * 2u > kCap) flush();  // room for the usual few rows: the next tile does not re-run
        } else {  // ablation builds: keep every accumulator alive
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[j][r];
            if (sum == 12345.678f) a.cand_count[0] = 1;
        }
    };

    // ---- prologue: the ring's first R - HS sub-stages (slot 0 reads sub-stages 0 .. HS-1 and leaves the rest of the window in
    // flight); every wave brings k-step w of each
    for (uint32_t q = 0; q < (uint32_t)(R - HS); ++q) issue_sub((uint32_t)w, 1u, true, w == 0);
    __syncthreads();  // q_lds, x_read, fl are visible (the DMA in flight is not waited for: asm statements are invisible to the fence)

    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    char* xp = lds + C::kXOff + (size_t)cp * 8192 + (size_t)lane * 64;  // this pair's hand-over area: [2 accumulators][64 lanes][16 floats]
    bool full_pattern = false;  // this wave issued its whole per-slot share in the slot just ended
    // barrier n (n = 0 .. my_tiles) + the refill of the ring behind it
    auto slot_sync = [&](uint32_t n) {
        // this wave's DMAs that the coming slot reads have landed: all but the newest few of a full slot's share
        if (full_pattern) {
            if (h) qs_wait_vmcnt<C::kAllow1>();
            else qs_wait_vmcnt<C::kAllow0>();
        } else {
            qs_wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the h = 0 wave's hand-over has been written
        qs_stage_barrier();
        // refill: the sub-stages that slot n - 1 freed (none in slot 0: the prologue filled the ring)
        if (n >= 1) {
            const bool have = iq < n_sub;
#pragma unroll 1
            for (uint32_t m = 0; m < (uint32_t)SUB; ++m) {
                const bool last = m == (uint32_t)SUB - 1;
                issue_sub(2u * (uint32_t)cp, 2u, last ? h == 0 : h == 1, cp == 0);
            }
            full_pattern = have && iq <= n_sub;  // (a partial last refill is waited for completely)
        }
    };
    // this wave's half of the K loop over the row tile whose half starts at ring slot `rs`: fragment reads P k-steps ahead of
    // the two MFMAs that use them
    auto multiply = [&](uint32_t rs) {
        h8 fa[P + 1];
        uint32_t rsv[HS];
#pragma unroll
        for (int s = 0; s < HS; ++s) rsv[s] = rs + (uint32_t)s >= (uint32_t)R ? rs + (uint32_t)s - R : rs + (uint32_t)s;
        auto load_frag = [&](int kk) {
            if (DBG & 8) return;
            fa[kk % (P + 1)] = *reinterpret_cast<const h8*>(lds + (size_t)rsv[kk / 8] * C::kSubBytes + (size_t)(kk % 8) * 1024 + (size_t)lane * 16);
        };
#pragma unroll
        for (int k0 = 0; k0 < P && k0 < KH; ++k0) load_frag(k0);
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            if (kk + P < KH) load_frag(kk + P);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (DBG & 1) asm volatile("" ::"v"(fa[kk % (P + 1)]), "v"(bq[j][kk]));
                else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk % (P + 1)], bq[j][kk], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the k-steps in this order: the read P ahead, then the two MFMAs
        }
    };
    const uint32_t tile_first = t_first + blockIdx.x;
    if (h == 0) {
        // ---- first halves: slot n multiplies row tile n from C = 0 and leaves the accumulators in X
        uint32_t rs = 0;  // ring slot of sub-stage 0 of the row tile
        for (uint32_t n = 0; n <= my_tiles; ++n) {
            slot_sync(n);
            if (n == my_tiles) break;  // (the last slot belongs to the second halves)
            acc[0] = zero;
            acc[1] = zero;
            multiply(rs);
            if (n >= 1) {  // hand on — once the h = 1 wave has picked up the previous accumulators
                while (__hip_atomic_load(&x_read[cp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");  // the stores below stay below the poll
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<f16v*>(xp + (size_t)j * 4096) = acc[j];
            rs = (rs + (uint32_t)SUB) % (uint32_t)R;
        }
    } else {
        // ---- second halves: slot n picks up row tile n - 1, finishes it; its epilogue runs behind the NEXT barrier, while the
        // SIMD's other wave multiplies
        uint32_t rs = (uint32_t)HS;  // ring slot of sub-stage HS of the row tile
        uint32_t mb = 0;             // its metadata buffer
        uint32_t tile = tile_first;
        slot_sync(0);
        slot_sync(1);
        for (uint32_t n = 1; n <= my_tiles; ++n) {
            // pick up the first half's accumulators (64 contiguous bytes per lane and accumulator: read straight into the
            // 16-register tuples), then tell the h = 0 wave that X may be overwritten
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = *reinterpret_cast<const f16v*>(xp + (size_t)j * 4096);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&x_read[cp], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            multiply(rs);
            if (n < my_tiles) slot_sync(n + 1);
            finish_tile(tile, mb);
            rs = (rs + (uint32_t)SUB) % (uint32_t)R;
            mb = mb == C::MB - 1 ? 0 : mb + 1;
            tile += gridDim.x;
        }
        if (!DENSE && staged) flush();
    }
}

template <class C, int DBG, bool DENSE, bool L2>
int kh_launch_one(const F16ScanArgs& a, const char* bfrag, const float* qinv, uint32_t blocks, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_kh_kernel<C, DBG, DENSE, L2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((vec_scan_f16_kh_kernel<C, DBG, DENSE, L2>), dim3(blocks), dim3(C::kThreads), C::kLdsBytes, stream, a, bfrag,
                       qinv, f16_tile_bytes(a.dim));
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

template <class C, int DBG>
int kh_launch(orama_ctx* ctx, const F16ScanArgs& a, const char* bfrag, const float* qinv, hipStream_t stream) {
    uint64_t blocks = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    if (blocks > (uint64_t)ctx->compute_units) blocks = (uint64_t)ctx->compute_units;
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;
    if (a.out_dense) {
        return l2 ? kh_launch_one<C, DBG, true, true>(a, bfrag, qinv, (uint32_t)blocks, stream)
                  : kh_launch_one<C, DBG, true, false>(a, bfrag, qinv, (uint32_t)blocks, stream);
    }
    return l2 ? kh_launch_one<C, DBG, false, true>(a, bfrag, qinv, (uint32_t)blocks, stream)
              : kh_launch_one<C, DBG, false, false>(a, bfrag, qinv, (uint32_t)blocks, stream);
}

template <int KSTEPS>
int kh_dispatch(orama_ctx* ctx, const F16ScanArgs& a, const char* bfrag, const float* qinv, hipStream_t stream, int dbg) {
    using Cfg = KhCfg<KSTEPS, (KSTEPS == 48 ? 14 : (KSTEPS == 32 ? 14 : 12))>;
    if constexpr (KSTEPS == 48) {  // ablation builds of the C5 shape (timing only)
        if (dbg && !a.out_dense) {
            switch (dbg) {
                case 1: return kh_launch<Cfg, 1>(ctx, a, bfrag, qinv, stream);    // DMA + LDS reads, no MFMA
                case 9: return kh_launch<Cfg, 9>(ctx, a, bfrag, qinv, stream);    // DMA only
                case 32: return kh_launch<Cfg, 32>(ctx, a, bfrag, qinv, stream);  // everything but the epilogue
                case 34: return kh_launch<Cfg, 34>(ctx, a, bfrag, qinv, stream);  // LDS reads + MFMA + barriers, no epilogue
                case 40: return kh_launch<Cfg, 40>(ctx, a, bfrag, qinv, stream);  // DMA + MFMA, no LDS reads, no epilogue
                case 42: return kh_launch<Cfg, 42>(ctx, a, bfrag, qinv, stream);  // MFMA + barriers only
                default: break;
            }
        }
    }
    return kh_launch<Cfg, 0>(ctx, a, bfrag, qinv, stream);
}

}  // namespace

bool vec_scan_f16_kh_supports(uint32_t dim, uint32_t q) {
    const uint32_t kpad = f16_kpad(dim);
    return kpad <= 768 && kpad % 256 == 0 && q > 128 && q <= kF16WideMaxQ;
}

int launch_vec_scan_f16_kh(orama_ctx* ctx, const F16ScanArgs& a_in, void* d_query_frags, hipStream_t stream) {
    static const uint32_t k2dbg = [] { const char* e = orama::dev_env("ORAMA_K2_DBG"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    F16ScanArgs a = a_in;
    if (!a.out_dense) a.dbg = k2dbg & 2u;  // timing ablation: no candidate appends
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries && d_query_frags, "vec_scan_f16_kh: bad arguments");
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f16_kh: bad row range");
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count), "vec_scan_f16_kh: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f16_kh: filter needs row_doc");
    ORAMA_REQUIRE(vec_scan_f16_kh_supports(a.dim, a.q), "vec_scan_f16_kh: %u dimensions x %u queries outside the kernel's envelope", a.dim, a.q);
    if (a.row_begin == a.row_end) return ORAMA_OK;
    const uint32_t ksteps = f16_kpad(a.dim) / 16;
    const char* bfrag = reinterpret_cast<const char*>(d_query_frags);
    const float* qinv = reinterpret_cast<const float*>(bfrag + (size_t)8 * ksteps * 1024);
    ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
    int dbg = 0;
    if (const char* e = orama::dev_env("ORAMA_K2C_DBG")) dbg = std::atoi(e);
    switch (ksteps) {
        case 16: return kh_dispatch<16>(ctx, a, bfrag, qinv, stream, dbg);
        case 32: return kh_dispatch<32>(ctx, a, bfrag, qinv, stream, dbg);
        case 48: return kh_dispatch<48>(ctx, a, bfrag, qinv, stream, dbg);
        default: break;
    }
    set_error("vec_scan_f16_kh: kpad %u not a multiple of 256", ksteps * 16);
    return ORAMA_ERR_INVALID;
}

}  // namespace orama

#endif  // ORAMA_COMPARISON_KERNELS
