// vec_f16.hip — K2: batched-query cosine scan over an fp16 corpus on the gfx950 matrix cores.
//
// GEMM shape per launch: scores[rows x Q] = corpus[rows x K] · queriesᵀ[K x Q], Q <= 64, never
// materialised: each wave owns 32-row tiles, streams the tile's A fragments straight from HBM into
// VGPRs (the HBM layout IS the fragment layout — one contiguous 1 KiB per wave-load, no LDS staging of
// the corpus), multiplies against the query fragments that sit in LDS (converted to fp16 once per
// block, conflict-free ds_read_b128: lane-linear 16 B), accumulates in f32 with
// v_mfma_f32_32x32x16_f16 and applies the epilogue (1 - s/|x||q|, filter, per-query threshold test)
// directly on the accumulator registers.
//
// Roofline: HBM.  Arithmetic intensity = Q flop/B (64 at C3) against a ridge of ~312 flop/B, so at
// the HBM roof the matrix pipe is ~Q/312 busy (20 % at Q = 64).  Algorithmic bytes = rows·kpad·2 per
// launch (one pass serves the whole batch).  Per CU: 8 waves (2/SIMD), each with two 8-KiB chunks of
// A fragments in flight (double-buffered across tile boundaries) → 128 KiB of loads outstanding.
//
// The A/B element order inside a 16-wide k-step only has to be the SAME for both operands (the dot
// product is invariant under a common permutation of k), and it is: both are filled with
// lane = (kgrp << 5) | (row or query), element e ↔ k = 16·kstep + 8·kgrp + e.
#include "vec_f16.hpp"

#include <cstdlib>

#include "device_utils.hpp"
#include "vec_f16_async.hpp"

namespace orama {

namespace {

using f16async::f16v;
using f16async::f4;
using f16async::h8;
using f16async::wave_or_u32;

constexpr int kBlock = 512;          // 8 waves: 2 per SIMD
constexpr int kWavesPerBlock = kBlock / 64;
// KC = k-steps per register chunk (KC x 1 KiB per wave-load group), NBUF = chunks in the register ring
// (NBUF-1 chunks stay in flight while one is being multiplied).

__device__ __forceinline__ h8 as_h8(f4 v) { return __builtin_bit_cast(h8, v); }

// DENSE: the launch writes every distance (the head of the store) instead of appending the rows under the thresholds.
template <int NQT, int KC, int NBUF, bool DENSE>
__global__ __launch_bounds__(kBlock) void vec_scan_f16_kernel(F16ScanArgs a, uint32_t ksteps, uint64_t tile_bytes) {
    constexpr int kChunk = KC;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t frag_total = ksteps * NQT * 64;
    float* qinv = reinterpret_cast<float*>(lds + (size_t)frag_total * 16);

    // ---- prologue: queries (f32, HBM/L2) → fp16 B fragments in LDS
    for (uint32_t idx = tid; idx < frag_total; idx += kBlock) {
        const uint32_t ks = idx / (NQT * 64);
        const uint32_t rem = idx - ks * (NQT * 64);
        const uint32_t qt = rem >> 6, l = rem & 63;
        const uint32_t j = qt * 32 + (l & 31);
        const uint32_t k0 = ks * 16 + (l >> 5) * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            const float x = (j < a.q && k < a.dim) ? a.queries[(size_t)j * a.dim + k] : 0.0f;
            v[e] = (_Float16)x;
        }
        *reinterpret_cast<h8*>(lds + (size_t)idx * 16) = v;
    }
    __syncthreads();
    if (tid < NQT * 32) {  // |q| of the fp16-rounded query, f32 accumulation, fixed order
        const uint32_t j = tid;
        float ss = 0.0f;
        for (uint32_t k = 0; k < ksteps * 16; ++k) {
            const uint32_t ks = k >> 4, g = (k >> 3) & 1, e = k & 7;
            const uint32_t l = g * 32 + (j & 31), qt = j >> 5;
            const float x = (float)reinterpret_cast<const _Float16*>(lds + (size_t)((ks * NQT + qt) * 64 + l) * 16)[e];
            ss = fmaf(x, x, ss);
        }
        qinv[j] = a.metric == ORAMA_METRIC_L2SQ ? ss : (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f);
    }
    __syncthreads();
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;

    // ---- tiles of this wave
    const uint32_t gw = uniform_u32(blockIdx.x * kWavesPerBlock + (tid >> 6));
    const uint32_t gwaves = gridDim.x * kWavesPerBlock;
    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;
    const uint64_t n_tiles = t_end - t_first;
    if (gw >= n_tiles) return;
    const uint64_t tile0 = t_first + gw;  // wave w takes tiles w, w + W, …: the chip sweeps one window
    const uint64_t tile_step = gwaves;
    const uint64_t my_tiles = (n_tiles - gw + gwaves - 1) / gwaves;
    const uint32_t nc = ksteps / kChunk;  // chunks per tile
    const char* base = reinterpret_cast<const char*>(a.tiled);

    f16v acc[NQT];
    f4 buf[NBUF][KC];
    uint64_t ld_tile = tile0;  // (tile, chunk) cursor of the NEXT load
    uint32_t ld_c = 0;
    uint64_t ld_more = my_tiles * nc - 1;  // chunks still to load after the one under the cursor
    uint64_t cp_tile = tile0;              // cursor of the NEXT compute
    uint32_t cp_c = 0;

    // The chunk loads are UNCONDITIONAL (past the end the cursor stays on the wave's last chunk and re-reads
    // it): with a fixed number of loads between a chunk's issue and its use, the compiler's s_waitcnt pass
    // emits counted vmcnt(N) waits that leave the NBUF-1 prefetched chunks in flight.  (Conditional loads made
    // it fall back to vmcnt(<KC), which drained the ring before every chunk: 5.4 → see profiles/.)
    // Tile metadata (1/|x| or |x|^2 of the 32 rows + the tombstone word) travels with the corpus stream: every load
    // group also moves the 256-byte metadata record of the tile of the NEXT chunk straight into LDS
    // (global_load_lds_dword: no destination registers, counted by vmcnt like the chunk loads, in order).  When the
    // compiler's counted wait for the last register of chunk g has passed, everything issued in group g-1 has landed,
    // so the epilogue of chunk g reads the record group g-1 wrote.  NBUF + 1 slots: group g + NBUF - 1 is issued just
    // before chunk g is multiplied and must not overwrite the record that chunk's epilogue is about to read.
    // (Scalar loads issued at epilogue time stalled the wave for a full memory latency per tile: 0.58 of 3.0 ms at
    // 10 M x 768, 64 queries — profiles/r02_k2_epilogue.log.)
    constexpr int kMetaSlots = NBUF + 1;
    char* meta = lds + (size_t)frag_total * 16 + 64 * sizeof(float) + (size_t)(tid >> 6) * kMetaSlots * kF16MetaBytes;
    const uint32_t meta_addr = uniform_u32((uint32_t)(size_t)(__attribute__((address_space(3))) char*)meta);
    uint32_t m_w = 0, m_r = 0;  // next slot to write / to read (wave-uniform)
    const uint32_t* meta_norm = reinterpret_cast<const uint32_t*>(a.inv_norm) + (lane & 31);
    const bool meta_dead_lane = lane == 32 && a.dead != nullptr;
    auto load_meta = [&]() {
        const uint32_t* src = meta_dead_lane ? a.dead + ld_tile : meta_norm + ld_tile * 32;
        // An asm statement, not __builtin_amdgcn_global_load_lds: the compiler's wait-count pass makes every later
        // LDS read that may alias an LDS-DMA destination (here: the fragment reads of the MFMA loop) wait for the DMA,
        // which turned the counted waits of the ring into one vmcnt(0) drain per cycle.  Unseen by that pass, the DMA
        // only makes its counted waits stricter by the <= NBUF records in flight (completion is in order).
        // m0 is an input operand ("{m0}"): the compiler materialises it and knows it is live; the s_nop is the wait
        // state gfx9 wants between a write of m0 and an LDS-DMA instruction reading it.
        asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, off"
                     :
                     : "v"(src), "{m0}"(meta_addr + m_w * kF16MetaBytes)
                     : "memory");
        m_w = m_w + 1 == kMetaSlots ? 0 : m_w + 1;
    };
    auto load_chunk = [&](f4* b) {
        const f4* p = reinterpret_cast<const f4*>(base + ld_tile * tile_bytes + (uint64_t)ld_c * kChunk * 1024) + lane;
#pragma unroll
        for (int s = 0; s < kChunk; ++s) b[s] = __builtin_nontemporal_load(p + s * 64);
        if (ld_more) {
            --ld_more;
            if (++ld_c == nc) {
                ld_c = 0;
                ld_tile += tile_step;
            }
        }
        load_meta();
    };

    // per-lane constants of the epilogue: this lane's query column per tile, its 1/|q| and threshold
    float qi_reg[NQT], tau_reg[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const uint32_t j = qt * 32 + (lane & 31);
        qi_reg[qt] = qinv[j];
        tau_reg[qt] = (a.tau && j < a.q) ? a.tau[j] : 0.0f;
    }
    // Rows that pass the threshold are STAGED per wave in LDS (distance, row, query) and appended to the per-query
    // candidate lists when the staging area is full — with ONE global atomic instruction per flush, however many rows
    // it holds.  The append needs the value a global atomic returns, and on this ISA that wait (vmcnt, in order) is
    // also a wait for every corpus load the wave has in flight, behind a memory system the scan keeps saturated:
    // appended from the epilogue directly, ~1.5 passing rows per tile drained the prefetch ring at nearly every tile
    // (0.45 of 3.0 ms at 10 M x 768, 64 queries, k = 100), and flushing 64 rows per atomic instruction still cost 0.36
    // (profiles/r02_k2_epilogue.log).  A flush ranks the staged rows within their query through an LDS histogram
    // (ds_add_rtn: lgkmcnt, not vmcnt), lane j reserves the hist[j] slots of query j's list with one atomic, and the
    // rows go to base[j] + rank.  The staging area takes what LDS is left (a.stage_cap entries of 12 bytes, >= 128):
    // at 768 dimensions a wave stages 512 rows and most waves flush once, when their tiles are done.  With a filter,
    // the lookup (row -> DocumentId -> bitmap word: two dependent loads with the same problem) happens here as well.
    const uint32_t cap = a.stage_cap;
    // (wave-uniform offsets, made so explicitly: as per-lane values the three base addresses were spilled to scratch
    // memory and every staged row waited for their reloads)
    const uint32_t wave_in_block = uniform_u32((uint32_t)tid >> 6);
    const uint32_t side_off = uniform_u32(frag_total * 16 + 64 * (uint32_t)sizeof(float) +
                                          (uint32_t)kWavesPerBlock * kMetaSlots * kF16MetaBytes);
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds + side_off + wave_in_block * 256u);
    uint32_t* stage = reinterpret_cast<uint32_t*>(lds + side_off + (uint32_t)kWavesPerBlock * 256u + wave_in_block * 12u * cap);
    uint32_t staged = 0;  // wave-uniform
    // The lanes of the wave exchange data through LDS here (rows staged by one lane are appended by another, bins
    // counted by many lanes are read by one): wavefront-scope atomics and fences keep the compiler from treating the
    // wave's LDS as one thread's private memory (it had forwarded this lane's `hist[lane] = 0` to its own read of the
    // bin for lanes without a staged row).  They cost nothing: one wave's LDS operations execute in order.
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    auto bin_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto bin_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto flush = [&]() {
        bin_store(&hist[lane], 0u);
        wave_fence();
        for (uint32_t i = lane; i < staged; i += 64) {
            const uint32_t j = stage[2 * cap + i];
            bool keep = !(a.dbg & 2u);
            if (a.allow) {
                const uint64_t doc = a.row_doc[stage[cap + i]];
                keep = keep && doc < a.allow_bits && ((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
            }
            // query | rank within the query
            stage[2 * cap + i] =
                keep ? (j | (__hip_atomic_fetch_add(&hist[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) << 6)) : ~0u;
        }
        wave_fence();
        const uint32_t mine = bin_load(&hist[lane]);
        bin_store(&hist[lane], mine ? atomicAdd(&a.cand_count[lane], mine) : 0u);  // lane = query: first slot of its rows
        wave_fence();
        for (uint32_t i = lane; i < staged; i += 64) {
            const uint32_t jr = stage[2 * cap + i];
            if (jr == ~0u) continue;
            const uint32_t j = jr & 63u;
            const uint64_t pos = (uint64_t)j * a.cand_stride + bin_load(&hist[j]) + (jr >> 6);
            a.cand_dist[pos] = __uint_as_float(stage[i]);
            a.cand_row[pos] = stage[cap + i];
        }
        wave_fence();
        staged = 0;
    };

    // Filter mode: `start` = 16 qt + r of the first accumulator row still to be looked at; returns 16 NQT when the tile is
    // done, else the position at which the staging area ran full — the caller flushes (at a point where nothing of the
    // epilogue is live: inlined into the row loop the flush needs registers the prefetch ring does not leave, and a spilled
    // value's reload waits for the whole ring: vmcnt is in order) and calls again.  Dense mode: always done.
    auto epilogue = [&](uint64_t tile, uint32_t start) -> uint32_t {
        // this lane's 16 accumulator rows are (r & 3) + 8 (r >> 2) + 4 hi_half: four 16-byte LDS reads of the record.
        // The statement takes an accumulator as a (never used) operand so that it stays behind the tile's last MFMA
        // and with it behind the counted wait described above.
        f4 n4[4];
        uint32_t dead_word;
        {
            const uint32_t rec = meta_addr + m_r * kF16MetaBytes;
            const uint32_t mine = rec + ((lane >> 5) ? 16u : 0u);
            asm volatile(
                "ds_read_b128 %0, %5\n\t"
                "ds_read_b128 %1, %5 offset:32\n\t"
                "ds_read_b128 %2, %5 offset:64\n\t"
                "ds_read_b128 %3, %5 offset:96\n\t"
                "ds_read_b32 %4, %6 offset:128\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(n4[0]), "=&v"(n4[1]), "=&v"(n4[2]), "=&v"(n4[3]), "=&v"(dead_word)
                : "v"(mine), "v"(rec), "v"(acc[NQT - 1][15]));
            if (!a.dead) dead_word = 0u;
        }
        const uint32_t hi4 = (lane >> 5) ? 4u : 0u;
        const bool full = tile * 32 + 32 <= a.row_end;  // wave-uniform: only the last tile of the store is partial
        const uint32_t left = full ? 32u : (uint32_t)(a.row_end - tile * 32);  // rows of the tile inside the store
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            if (!DENSE && (uint32_t)(qt + 1) * 16u <= start) continue;  // done before the flush
            const uint32_t j = qt * 32 + (lane & 31);
            const bool live = j < a.q;  // (a predicate, not a branch: the staging below is wave-synchronous)
            if (DENSE && !live) continue;
            const float qi = qi_reg[qt];
            const float tau = tau_reg[qt];
            // |q|^2 + |x|^2 - 2 q.x, or 1 - q.x / (|q||x|): written as the fused operations the compiler contracts the
            // plain expressions to (2 q.x is exact, so the first is the same number either way) — explicit, so that the
            // slow path below shares no sub-expression with the fast path above it (see there)
            auto dist_of = [&](float dot, float n, float qv) -> float {
                return l2 ? __builtin_fmaf(-2.0f, dot, qv + n) : __builtin_fmaf(-dot, n * qv, 1.0f);
            };
            if constexpr (DENSE) {  // the head of the store: every distance is written, NaN = excluded
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t i = (uint32_t)((r & 3) + 8 * (r >> 2)) + hi4;
                    const uint64_t row = tile * 32 + i;
                    if (!full && row >= a.row_end) continue;
                    bool excluded = (dead_word >> i) & 1u;
                    if (!excluded && a.allow) {
                        const uint64_t doc = a.row_doc[row];
                        excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                    }
                    a.out_dense[(uint64_t)j * a.dense_stride + (row - a.row_begin)] =
                        excluded ? __builtin_nanf("") : dist_of(acc[qt][r], n4[r >> 2][r & 3], qi);
                }
                continue;
            }
            // filter mode, fast reject: almost no row beats the running k-th best distance, so take the minimum of the
            // 16 distances first and look closer only when it passes.
            if (a.dbg & 8u) {  // timing ablation: metadata read only
                if (n4[0][0] + n4[1][1] + n4[2][2] + n4[3][3] == 12345.678f) a.cand_count[0] = 1;
                continue;
            }
            float best = __builtin_huge_valf();
            if (l2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) best = fminf(best, __builtin_fmaf(-2.0f, acc[qt][r], qi + n4[r >> 2][r & 3]));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) best = fminf(best, __builtin_fmaf(-acc[qt][r], n4[r >> 2][r & 3] * qi, 1.0f));
            }
            if (a.dbg & 4u) {  // timing ablation: fast reject only, never the slow path
                if (best == 12345.678f) a.cand_count[0] = 1;
                continue;
            }
            // (a wave-uniform decision from here on: the staging below uses ballots)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && best < tau) == 0, 1)) continue;
            // which of this lane's 16 elements pass: threshold, tombstone, row range
            // (the distances are computed again, from operands the compiler cannot see through: kept from the fast path
            // above — or hoisted into it — the 16 values would live in scratch memory: stores on every tile, loads here)
            float qi_s = qi;
            asm volatile("" : "+v"(qi_s));
            const uint32_t alive = (~dead_word & (left >= 32u ? ~0u : ((1u << left) - 1u))) >> hi4;
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t i = (uint32_t)((r & 3) + 8 * (r >> 2));
                const float d = dist_of(acc[qt][r], n4[r >> 2][r & 3], qi_s);
                m |= ((live && d < tau ? 1u : 0u) & (alive >> i)) << r;
            }
            // the accumulator rows somebody passes (usually one or two of the 16): OR over the wave, then only those —
            // element by element with a wave-uniform index (a per-lane index into the accumulators would go through
            // scratch memory, whose loads wait on vmcnt like the appends this staging exists to avoid)
            uint32_t any = wave_or_u32(m);
            if (start > (uint32_t)qt * 16u) any &= ~0u << (start - (uint32_t)qt * 16u);  // resuming after a flush
#pragma unroll 1
            while (any) {
                const uint32_t r = (uint32_t)__builtin_ctz(any);
                const bool mine = (m >> r) & 1u;
                const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
                const uint32_t n_pass = (uint32_t)__popcll(bal);
                if (staged + n_pass > cap) return (uint32_t)qt * 16u + r;  // no room: flush, then resume here
                any &= any - 1u;
                if (mine) {
                    // (the norm of this one row again, from the record in LDS: a run-time index into the four norm
                    // registers quads would put them in scratch memory)
                    const float nr = *reinterpret_cast<const float*>(meta + (size_t)m_r * kF16MetaBytes + (size_t)(((r & 3u) + 8u * (r >> 2)) + hi4) * 4);
                    const float dist = dist_of(acc[qt][r], nr, qi_s);
                    const uint32_t pos =
                        staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    stage[pos] = __float_as_uint(dist);
                    stage[cap + pos] = (uint32_t)(tile * 32) + ((r & 3u) + 8u * (r >> 2)) + hi4;
                    stage[2 * cap + pos] = j;
                }
                staged = uniform_u32(staged + n_pass);
            }
        }
        return 16u * NQT;
    };
    auto finish_tile = [&](uint64_t tile) {
        uint32_t at = 0;
        while ((at = epilogue(tile, at)) < 16u * NQT) {
            wave_fence();
            flush();
        }
        if (!DENSE && staged > cap - 64) {  // keep a tile's usual few rows' worth of room
            wave_fence();
            flush();
        }
    };

    auto compute_chunk = [&](const f4* b) {
        if (cp_c == 0) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qt][r] = 0.0f;
        }
        const char* bl = lds + ((size_t)cp_c * kChunk * NQT * 64 + lane) * 16;
#pragma unroll
        for (int s = 0; s < kChunk; ++s) {
            const h8 av = as_h8(b[s]);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                // (reading the fragments of step s + 1 ahead of the MFMAs of step s changes nothing: 2.54-2.61 ms either
                // way — the wave waits for HBM, not for LDS)
                const h8 bv = *reinterpret_cast<const h8*>(bl + (size_t)(s * NQT + qt) * 1024);
                acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[qt], 0, 0, 0);
            }
        }
        const bool tile_done = ++cp_c == nc;
        if (tile_done) {
            if (a.dbg & 1u) {  // timing ablation: no epilogue (the accumulators stay live through a never-true store)
                float sum = 0.0f;
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum += acc[qt][r];
                if (sum == 12345.678f) a.cand_count[0] = 1;
            } else {
                finish_tile(cp_tile);
            }
            cp_c = 0;
            cp_tile += tile_step;
        }
        m_r = m_r + 1 == kMetaSlots ? 0 : m_r + 1;
    };

    const uint64_t total = my_tiles * nc;
    load_meta();  // the record of the first tile (the one "group -1" would have brought)
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) load_chunk(buf[b]);
    // ONE loop, no separate tail: the loads are unconditional anyway (past the end the cursor re-reads the wave's last chunk),
    // so the last trip just skips the chunks that do not exist.  (A tail that picked the ring up where the loop left it made
    // the register allocator carry the prologue's chunks around the loop in scratch memory.)
    for (uint64_t g = 0; g < total; g += NBUF) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            load_chunk(buf[(b + NBUF - 1) % NBUF]);
            if (g + b < total) compute_chunk(buf[b]);
        }
    }
    if (!DENSE && staged) {
        wave_fence();
        flush();
    }
}

// ---------------------------------------------------------------- K1h: 1..4 queries without the matrix cores
// The candidate stage of the two-stage plan scans the fp16 shadow for ONE query most of the time.  K2 pads that to a
// 32-column MFMA tile and streams at 5.9 TB/s; this kernel keeps K2's data path (fragment-ordered rows streamed through
// a register ring with counted waits) and replaces the MFMAs by v_dot2_f32_f16: lane l owns row l & 31 of the tile and
// the k-half l >> 5 of every k-step — exactly the 8 halves one 16-byte load delivers — multiplies them with the query's
// 8 halves of the same k-half (LDS, two distinct addresses per wave: broadcast reads) and keeps one f32 partial sum per
// query; the two halves of a row meet through one cross-lane add per tile.  Used only where no bit-identity with K2 is
// promised (F16ScanArgs::solo: the shadow scan — its error bound holds for any summation order).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int NQ, int KC, int NBUF>
__global__ __launch_bounds__(kBlock) void vec_scan_f16_solo_kernel(F16ScanArgs a, uint32_t ksteps, uint64_t tile_bytes) {
    constexpr int kChunk = KC;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t frag_total = ksteps * 2 * NQ;  // [k-step][k-half][query] x 16 B
    float* qinv = reinterpret_cast<float*>(lds + (size_t)frag_total * 16);
    for (uint32_t idx = tid; idx < frag_total; idx += kBlock) {
        const uint32_t j = idx % NQ, kg = idx / NQ;  // kg = k-step * 2 + k-half
        const uint32_t k0 = kg * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            v[e] = (_Float16)((j < a.q && k < a.dim) ? a.queries[(size_t)j * a.dim + k] : 0.0f);
        }
        *reinterpret_cast<h8*>(lds + (size_t)idx * 16) = v;
    }
    __syncthreads();
    if (tid < NQ) {  // |q| of the fp16-rounded query, f32 accumulation
        float ss = 0.0f;
        for (uint32_t k = 0; k < ksteps * 16; ++k) {
            const float x = (float)reinterpret_cast<const _Float16*>(lds + (size_t)((k >> 3) * NQ + tid) * 16)[k & 7];
            ss = fmaf(x, x, ss);
        }
        qinv[tid] = a.metric == ORAMA_METRIC_L2SQ ? ss : (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f);
    }
    __syncthreads();
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;
    const uint32_t gw = uniform_u32(blockIdx.x * kWavesPerBlock + (tid >> 6));
    const uint32_t gwaves = gridDim.x * kWavesPerBlock;
    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;
    const uint64_t n_tiles = t_end - t_first;
    if (gw >= n_tiles) return;
    const uint64_t tile0 = t_first + gw;
    const uint64_t tile_step = gwaves;
    const uint64_t my_tiles = (n_tiles - gw + gwaves - 1) / gwaves;
    const uint32_t nc = ksteps / kChunk;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    const uint32_t half = (uint32_t)lane >> 5, rlane = (uint32_t)lane & 31u;

    float acc[NQ];
    f4 buf[NBUF][KC];
    float nrm[NBUF];  // 1/|x| (or |x|^2) of this lane's row, re-read with every chunk: one more load per chunk, but in a
                      // fixed place of the load stream, so the compiler keeps its counted waits (see load_chunk in K2)
    uint64_t ld_tile = tile0;
    uint32_t ld_c = 0;
    uint64_t ld_more = my_tiles * nc - 1;
    uint64_t cp_tile = tile0;
    uint32_t cp_c = 0;
    auto load_chunk = [&](f4* b, float* nr) {
        const f4* p = reinterpret_cast<const f4*>(base + ld_tile * tile_bytes + (uint64_t)ld_c * kChunk * 1024) + lane;
#pragma unroll
        for (int s = 0; s < kChunk; ++s) b[s] = __builtin_nontemporal_load(p + s * 64);
        *nr = a.inv_norm[ld_tile * 32 + rlane];
        if (ld_more) {
            --ld_more;
            if (++ld_c == nc) {
                ld_c = 0;
                ld_tile += tile_step;
            }
        }
    };
    float qi[NQ], tau[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        qi[j] = qinv[j];
        tau[j] = (a.tau && (uint32_t)j < a.q) ? a.tau[j] : 0.0f;
    }
    typedef const uint32_t __attribute__((address_space(4))) cu32;
    auto epilogue = [&](uint64_t tile, float inv) {
        const uint32_t dead_word = a.dead ? ((cu32*)(uintptr_t)a.dead)[tile] : 0u;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const float dot = acc[j] + __shfl_xor(acc[j], 32, 64);  // the two k-halves of the row
            const uint64_t row = tile * 32 + rlane;
            if (half != 0 || row >= a.row_end || (uint32_t)j >= a.q) continue;
            bool excluded = (dead_word >> rlane) & 1u;
            const float dist = l2 ? (qi[j] + inv) - 2.0f * dot : 1.0f - dot * (inv * qi[j]);
            if (a.out_dense) {
                if (!excluded && a.allow) {
                    const uint64_t doc = a.row_doc[row];
                    excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                }
                a.out_dense[(uint64_t)j * a.dense_stride + (row - a.row_begin)] = excluded ? __builtin_nanf("") : dist;
            } else if (!excluded && dist < tau[j]) {
                if (a.allow) {
                    const uint64_t doc = a.row_doc[row];
                    excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                }
                if (!excluded) {
                    const uint32_t pos = atomicAdd(&a.cand_count[j], 1u);
                    a.cand_dist[(uint64_t)j * a.cand_stride + pos] = dist;
                    a.cand_row[(uint64_t)j * a.cand_stride + pos] = (uint32_t)row;
                }
            }
        }
    };
    auto compute_chunk = [&](const f4* b, float nr) {
        if (cp_c == 0) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) acc[j] = 0.0f;
        }
        const char* ql = lds + ((size_t)(cp_c * kChunk * 2 + half) * NQ) * 16;
#pragma unroll
        for (int s = 0; s < kChunk; ++s) {
            const h8 av = as_h8(b[s]);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const h8 qv = *reinterpret_cast<const h8*>(ql + ((size_t)s * 2 * NQ + j) * 16);
                float c = acc[j];
                c = __builtin_amdgcn_fdot2(h2{av[0], av[1]}, h2{qv[0], qv[1]}, c, false);
                c = __builtin_amdgcn_fdot2(h2{av[2], av[3]}, h2{qv[2], qv[3]}, c, false);
                c = __builtin_amdgcn_fdot2(h2{av[4], av[5]}, h2{qv[4], qv[5]}, c, false);
                c = __builtin_amdgcn_fdot2(h2{av[6], av[7]}, h2{qv[6], qv[7]}, c, false);
                acc[j] = c;
            }
        }
        if (++cp_c == nc) {
            epilogue(cp_tile, nr);
            cp_c = 0;
            cp_tile += tile_step;
        }
    };
    const uint64_t total = my_tiles * nc;
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) load_chunk(buf[b], &nrm[b]);
    uint64_t g = 0;
    for (; g + NBUF <= total; g += NBUF) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            load_chunk(buf[(b + NBUF - 1) % NBUF], &nrm[(b + NBUF - 1) % NBUF]);
            compute_chunk(buf[b], nrm[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) {
        if (g + b < total) {
            load_chunk(buf[(b + NBUF - 1) % NBUF], &nrm[(b + NBUF - 1) % NBUF]);
            compute_chunk(buf[b], nrm[b]);
        }
    }
}

// K1h, second form (ORAMA_F16_SOLO=2): K1's loop shape instead of K2's register ring — a wave takes TWO tiles at a time,
// issues one group of G loads per tile, multiplies, next group; no software pipeline, 256-thread blocks, several per CU.
template <int NQ, int G>
__global__ __launch_bounds__(256) void vec_scan_f16_solo2_kernel(F16ScanArgs a, uint32_t ksteps, uint64_t tile_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t frag_total = ksteps * 2 * NQ;
    float* qinv = reinterpret_cast<float*>(lds + (size_t)frag_total * 16);
    for (uint32_t idx = tid; idx < frag_total; idx += 256) {
        const uint32_t j = idx % NQ, kg = idx / NQ;
        const uint32_t k0 = kg * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            v[e] = (_Float16)((j < a.q && k < a.dim) ? a.queries[(size_t)j * a.dim + k] : 0.0f);
        }
        *reinterpret_cast<h8*>(lds + (size_t)idx * 16) = v;
    }
    __syncthreads();
    if (tid < NQ) {
        float ss = 0.0f;
        for (uint32_t k = 0; k < ksteps * 16; ++k) {
            const float x = (float)reinterpret_cast<const _Float16*>(lds + (size_t)((k >> 3) * NQ + tid) * 16)[k & 7];
            ss = fmaf(x, x, ss);
        }
        qinv[tid] = a.metric == ORAMA_METRIC_L2SQ ? ss : (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f);
    }
    __syncthreads();
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;
    const uint32_t gw = uniform_u32(blockIdx.x * 4 + (tid >> 6));
    const uint32_t gwaves = gridDim.x * 4;
    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    const uint32_t half = (uint32_t)lane >> 5, rlane = (uint32_t)lane & 31u;
    float qi[NQ], tau[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        qi[j] = qinv[j];
        tau[j] = (a.tau && (uint32_t)j < a.q) ? a.tau[j] : 0.0f;
    }
    constexpr int TW = 2;  // tiles per wave iteration
    for (uint64_t t0 = t_first + (uint64_t)gw * TW; t0 < t_end; t0 += (uint64_t)gwaves * TW) {
        float acc[TW][NQ];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int j = 0; j < NQ; ++j) acc[t][j] = 0.0f;
        for (uint32_t c0 = 0; c0 < ksteps; c0 += G) {
            f4 x[TW][G];
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const uint64_t tile = min(t0 + t, t_end - 1);  // the odd last tile is read twice, emitted once
                const f4* p = reinterpret_cast<const f4*>(base + tile * tile_bytes + (uint64_t)c0 * 1024) + lane;
#pragma unroll
                for (int s = 0; s < G; ++s) x[t][s] = __builtin_nontemporal_load(p + s * 64);
            }
            const char* ql = lds + ((size_t)(c0 * 2 + half) * NQ) * 16;
#pragma unroll
            for (int s = 0; s < G; ++s) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const h8 qv = *reinterpret_cast<const h8*>(ql + ((size_t)s * 2 * NQ + j) * 16);
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
                        const h8 av = as_h8(x[t][s]);
                        float c = acc[t][j];
                        c = __builtin_amdgcn_fdot2(h2{av[0], av[1]}, h2{qv[0], qv[1]}, c, false);
                        c = __builtin_amdgcn_fdot2(h2{av[2], av[3]}, h2{qv[2], qv[3]}, c, false);
                        c = __builtin_amdgcn_fdot2(h2{av[4], av[5]}, h2{qv[4], qv[5]}, c, false);
                        c = __builtin_amdgcn_fdot2(h2{av[6], av[7]}, h2{qv[6], qv[7]}, c, false);
                        acc[t][j] = c;
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const uint64_t tile = t0 + t;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float dot = acc[t][j] + __shfl_xor(acc[t][j], 32, 64);
                const uint64_t row = tile * 32 + rlane;
                if (tile >= t_end || half != 0 || row >= a.row_end || (uint32_t)j >= a.q) continue;
                bool excluded = a.dead && ((a.dead[tile] >> rlane) & 1u);
                const float inv = a.inv_norm[row];
                const float dist = l2 ? (qi[j] + inv) - 2.0f * dot : 1.0f - dot * (inv * qi[j]);
                if (a.out_dense) {
                    if (!excluded && a.allow) {
                        const uint64_t doc = a.row_doc[row];
                        excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                    }
                    a.out_dense[(uint64_t)j * a.dense_stride + (row - a.row_begin)] = excluded ? __builtin_nanf("") : dist;
                } else if (!excluded && dist < tau[j]) {
                    if (a.allow) {
                        const uint64_t doc = a.row_doc[row];
                        excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                    }
                    if (!excluded) {
                        const uint32_t pos = atomicAdd(&a.cand_count[j], 1u);
                        a.cand_dist[(uint64_t)j * a.cand_stride + pos] = dist;
                        a.cand_row[(uint64_t)j * a.cand_stride + pos] = (uint32_t)row;
                    }
                }
            }
        }
    }
}

// K1h fused: ONE query, every wave keeps its best min(topk, 64) rows in registers (one key per lane) — one launch over
// the whole store instead of dense head + selection + filter scan + selection.  A wave scans 1/2048 of the rows, so it
// holds a fraction of a row of the global top-k on average; 64 slots (not topk = 228) cut the insertions — each one a
// wave-wide min reduction — from ~920 to ~350 per wave (0.38 ms of a 2.6 ms scan).  Exactness does not rest on that
// expectation: a wave that evicted anything reports the key of its worst kept row (wave_thr), and the caller treats the
// answer as incomplete when that row ranks inside the global top-k (shadow_wave_check_kernel) — the two-stage plan then
// answers the query by the fp32 scan, like any other candidate list it cannot prove complete.
template <int G, int DBG>
__global__ __launch_bounds__(256) void vec_scan_f16_solo_fused_kernel(F16ScanArgs a, uint32_t ksteps, uint64_t tile_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t frag_total = ksteps * 2;
    float* qinv = reinterpret_cast<float*>(lds + (size_t)frag_total * 16);
    for (uint32_t idx = tid; idx < frag_total; idx += 256) {
        const uint32_t k0 = idx * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t k = k0 + e;
            v[e] = (_Float16)(k < a.dim ? a.queries[k] : 0.0f);
        }
        *reinterpret_cast<h8*>(lds + (size_t)idx * 16) = v;
    }
    __syncthreads();
    if (tid == 0) {
        float ss = 0.0f;
        for (uint32_t k = 0; k < ksteps * 16; ++k) {
            const float x = (float)reinterpret_cast<const _Float16*>(lds + (size_t)(k >> 3) * 16)[k & 7];
            ss = fmaf(x, x, ss);
        }
        qinv[0] = a.metric == ORAMA_METRIC_L2SQ ? ss : (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f);
    }
    __syncthreads();
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;
    const uint32_t gw = uniform_u32(blockIdx.x * 4 + (tid >> 6));
    const uint32_t gwaves = gridDim.x * 4;
    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    const uint32_t half = (uint32_t)lane >> 5, rlane = (uint32_t)lane & 31u;
    const float qi = qinv[0];
    const uint32_t k = a.topk < 64u ? a.topk : 64u;  // per-wave capacity
    WaveTopKN<1> best;
    constexpr int TW = 2;
    for (uint64_t t0 = t_first + (uint64_t)gw * TW; t0 < t_end; t0 += (uint64_t)gwaves * TW) {
        float acc[TW], inv_pre[TW];
        uint32_t dead_pre[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            acc[t] = 0.0f;
            // the row's norm and the tile's tombstone word are fetched NOW, under the tile's own loads: read in the
            // epilogue they are a dependent miss every wave sits out (0.38 of 2.63 ms at 10 M rows)
            const uint64_t tile = min(t0 + t, t_end - 1);
            inv_pre[t] = a.inv_norm[tile * 32 + rlane];
            dead_pre[t] = a.dead ? a.dead[tile] : 0u;
        }
        for (uint32_t c0 = 0; c0 < ksteps; c0 += G) {
            f4 x[TW][G];
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const uint64_t tile = min(t0 + t, t_end - 1);
                const f4* p = reinterpret_cast<const f4*>(base + tile * tile_bytes + (uint64_t)c0 * 1024) + lane;
#pragma unroll
                for (int s = 0; s < G; ++s) x[t][s] = __builtin_nontemporal_load(p + s * 64);
            }
            const char* ql = lds + (size_t)(c0 * 2 + half) * 16;
#pragma unroll
            for (int s = 0; s < G; ++s) {
                const h8 qv = (DBG & 1) ? h8{1, 1, 1, 1, 1, 1, 1, 1} : *reinterpret_cast<const h8*>(ql + (size_t)s * 2 * 16);
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    const h8 av = as_h8(x[t][s]);
                    float c = acc[t];
                    c = __builtin_amdgcn_fdot2(h2{av[0], av[1]}, h2{qv[0], qv[1]}, c, false);
                    c = __builtin_amdgcn_fdot2(h2{av[2], av[3]}, h2{qv[2], qv[3]}, c, false);
                    c = __builtin_amdgcn_fdot2(h2{av[4], av[5]}, h2{qv[4], qv[5]}, c, false);
                    c = __builtin_amdgcn_fdot2(h2{av[6], av[7]}, h2{qv[6], qv[7]}, c, false);
                    acc[t] = c;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const uint64_t tile = t0 + t;
            const float dot = acc[t] + __shfl_xor(acc[t], 32, 64);
            const uint64_t row = tile * 32 + rlane;
            bool live = tile < t_end && half == 0 && row < a.row_end;
            if (DBG & 2) live = live && dot == 12345.678f;  // ablation: no epilogue work
            if (live) live = !((dead_pre[t] >> rlane) & 1u);
            float dist = 0.0f;
            if (live) {
                const float inv = inv_pre[t];
                dist = l2 ? (qi + inv) - 2.0f * dot : 1.0f - dot * (inv * qi);
                live = dist == dist;
            }
            // smaller distance wins, then the lower row: key = ~ordered(d) << 32 | ~row (as K1's fused mode)
            const unsigned long long key =
                ((unsigned long long)(~f32_to_ordered(dist)) << 32) | (unsigned long long)(uint32_t)(~(uint32_t)row);
            unsigned long long pending = __ballot(live && (best.count < k || key > best.thr));
            while (pending) {
                const int l = __ffsll((long long)pending) - 1;
                pending &= pending - 1;
                const unsigned long long kk = __shfl(key, l, 64);
                bool ok = best.count < k || kk > best.thr;  // the threshold may have risen since the ballot
                if (ok && a.allow) {  // the filter lookup only for rows that would enter the list
                    const uint64_t rr = (uint64_t)(uint32_t)(~(uint32_t)kk);
                    const uint64_t doc = a.row_doc[rr];
                    ok = doc < a.allow_bits && ((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                }
                if (ok) best.insert(kk, k, lane);
            }
        }
    }
    unsigned long long* out = a.wave_lists + (uint64_t)gw * kF16WaveListKeys;
    out[lane] = best.s[0];
    // full list = something may have been evicted: everything evicted ranks below thr
    if (lane == 0 && a.wave_thr) a.wave_thr[gw] = (best.count >= k && k < a.topk) ? best.thr : 0ull;
}

// flag[0] |= 1 when a wave's worst kept row is at least as good as the k-th best of the merged result: rows it evicted
// could belong to the top-k.  (Compared on the distance half of the keys: ties count as "could".)
__global__ void shadow_wave_check_kernel(const unsigned long long* __restrict__ wave_thr, uint32_t waves,
                                         const float* __restrict__ out_dist, const uint32_t* __restrict__ out_n, uint32_t k,
                                         uint32_t* __restrict__ flag) {
    __shared__ uint32_t bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const uint32_t n = out_n[0];
    // key of the k-th best distance (larger key = smaller distance); fewer than k results: every evicted row matters
    const uint32_t kth_hi = n >= k ? ~f32_to_ordered(out_dist[k - 1]) : 0u;
    for (uint32_t w = threadIdx.x; w < waves; w += blockDim.x) {
        const unsigned long long t = wave_thr[w];
        if (t != 0ull && (uint32_t)(t >> 32) >= kth_hi) bad = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) flag[0] = bad;
}

// ---------------------------------------------------------------- store / norms / gather
__global__ __launch_bounds__(256) void f16_store_rows_kernel(char* __restrict__ tiled,
                                                             const float* __restrict__ src, uint64_t first,
                                                             uint64_t n, uint32_t dim, uint32_t kpad,
                                                             uint64_t tile_bytes) {
    const uint32_t pieces = kpad / 8;
    const uint64_t total = n * pieces;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / pieces;
        const uint32_t p = (uint32_t)(t - r * pieces);
        const uint32_t k0 = p * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)((k0 + e < dim) ? src[r * dim + k0 + e] : 0.0f);
        const uint64_t row = first + r;
        const uint32_t ks = k0 >> 4, g = (k0 >> 3) & 1;
        const uint32_t lane = g * 32 + (uint32_t)(row & 31);
        *reinterpret_cast<h8*>(tiled + (row >> 5) * tile_bytes + ((uint64_t)ks * 64 + lane) * 16) = v;
    }
}

__device__ __forceinline__ h8 f16_piece(const char* tiled, uint64_t row, uint32_t p, uint64_t tile_bytes) {
    const uint32_t k0 = p * 8;
    const uint32_t ks = k0 >> 4, g = (k0 >> 3) & 1;
    const uint32_t lane = g * 32 + (uint32_t)(row & 31);
    return *reinterpret_cast<const h8*>(tiled + (row >> 5) * tile_bytes + ((uint64_t)ks * 64 + lane) * 16);
}

__global__ __launch_bounds__(256) void f16_inv_norm_kernel(const char* __restrict__ tiled, uint64_t first,
                                                           uint64_t n, uint32_t kpad, uint64_t tile_bytes,
                                                           float* __restrict__ inv_norm, bool l2) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint32_t pieces = kpad / 8;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const uint64_t row = first + i;
        float ss = 0.0f;
        for (uint32_t p = lane; p < pieces; p += 64) {
            const h8 v = f16_piece(tiled, row, p, tile_bytes);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) inv_norm[row] = l2 ? ss : (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f);
    }
}

__global__ __launch_bounds__(256) void f16_gather_rows_kernel(const char* __restrict__ tiled,
                                                              const uint64_t* __restrict__ idx, uint64_t n,
                                                              uint32_t dim, uint32_t kpad, uint64_t tile_bytes,
                                                              float* __restrict__ out) {
    const uint32_t pieces = kpad / 8;
    const uint64_t total = n * pieces;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / pieces;
        const uint32_t p = (uint32_t)(t - r * pieces);
        const h8 v = f16_piece(tiled, idx[r], p, tile_bytes);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (p * 8 + e < dim) out[r * dim + p * 8 + e] = (float)v[e];
    }
}

__global__ void f16_seed_candidates_kernel(const float* __restrict__ best_dist,
                                           const uint32_t* __restrict__ best_row,
                                           const uint32_t* __restrict__ best_n, uint32_t k, float* tau,
                                           float* cand_dist, uint32_t* cand_row, uint32_t* cand_count,
                                           uint64_t cand_stride, const float* __restrict__ tau_cap) {
    const uint32_t j = blockIdx.x;
    const uint32_t n = best_n[j];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        cand_dist[(uint64_t)j * cand_stride + i] = best_dist[(uint64_t)j * k + i];
        cand_row[(uint64_t)j * cand_stride + i] = best_row[(uint64_t)j * k + i];
    }
    if (threadIdx.x == 0) {
        cand_count[j] = n;
        // strict '<' against the k-th best: a later row with an equal distance has a higher row index
        // and can never displace it (tie rule: distance asc, row asc)
        float t = (n == k) ? best_dist[(uint64_t)j * k + (k - 1)] : __builtin_huge_valf();
        // (tau_cap: a bound from OUTSIDE this store's own rows so far — another shard's head, or the ceiling experiment's best-case
        // bound: rows AT the bound may still belong to the answer there, so the cap is passed one ulp up by its producer)
        if (tau_cap) t = fminf(t, tau_cap[j]);
        tau[j] = t;
    }
}

// ORAMA_F16_TAU_ORACLE=1 (scripts/f16_tau_ceiling_probe.py): remember each query's final k-th distance, one ulp up — the
// tightest threshold ANY exchange between shards could hand the next identical batch.
__global__ void f16_remember_kth_kernel(const float* __restrict__ out_dist, const uint32_t* __restrict__ out_n, uint32_t k, uint32_t q,
                                        float* __restrict__ cap) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= q) return;
    cap[j] = out_n[j] == k ? nextafterf(out_dist[(uint64_t)j * k + (k - 1)], __builtin_huge_valf()) : __builtin_huge_valf();
}

}  // namespace

uint64_t f16_tile_pad() {
    static const uint64_t pad = [] {
        uint64_t v = 0;  // measured: no effect on MI355X (profiles/r01_f16_tile_pad_sweep.md), so no padding
        if (const char* e = orama::dev_env("ORAMA_F16_TILE_PAD")) v = (uint64_t)std::strtoull(e, nullptr, 10);
        return (v + 15) & ~15ull;
    }();
    return pad;
}

namespace {

uint32_t blocks_for(uint64_t items, uint32_t per_block, uint32_t cap) {
    uint64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (uint32_t)b;
}

}  // namespace

int launch_f16_store_rows(void* tiled, const float* src, uint64_t first, uint64_t n, uint32_t dim,
                          hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    const uint32_t kpad = f16_kpad(dim);
    hipLaunchKernelGGL(f16_store_rows_kernel, dim3(blocks_for(n * (kpad / 8), 256, 16384)), dim3(256), 0,
                       stream, reinterpret_cast<char*>(tiled), src, first, n, dim, kpad, f16_tile_bytes(dim));
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_f16_inv_norm(const void* tiled, uint64_t first, uint64_t n, uint32_t dim, float* inv_norm,
                        hipStream_t stream, int metric) {
    if (n == 0) return ORAMA_OK;
    hipLaunchKernelGGL(f16_inv_norm_kernel, dim3(blocks_for(n, 4, 8192)), dim3(256), 0, stream,
                       reinterpret_cast<const char*>(tiled), first, n, f16_kpad(dim), f16_tile_bytes(dim), inv_norm,
                       metric == ORAMA_METRIC_L2SQ);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_f16_gather_rows(const void* tiled, const uint64_t* d_row_idx, uint64_t n, uint32_t dim,
                           float* d_out, hipStream_t stream) {
    if (n == 0) return ORAMA_OK;
    const uint32_t kpad = f16_kpad(dim);
    hipLaunchKernelGGL(f16_gather_rows_kernel, dim3(blocks_for(n * (kpad / 8), 256, 16384)), dim3(256), 0,
                       stream, reinterpret_cast<const char*>(tiled), d_row_idx, n, dim, kpad, f16_tile_bytes(dim), d_out);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

uint32_t vec_scan_f16_fused_waves(orama_ctx* ctx, const F16ScanArgs& a) {
    const uint32_t ksteps = f16_kpad(a.dim) / 16;
    if (a.q != 1 || a.topk < 1 || a.topk > kSelectMaxKeysFused || ksteps % 12 != 0 || !a.tiled || !a.inv_norm || !a.queries) return 0;
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    if (tiles == 0) return 4;
    return blocks_for((tiles + 1) / 2, 4, (uint32_t)ctx->compute_units * 2u) * 4u;
}

int launch_shadow_wave_check(const unsigned long long* d_wave_thr, uint32_t waves, const float* d_out_dist, const uint32_t* d_out_n,
                             uint32_t k, uint32_t* d_flag, hipStream_t stream) {
    hipLaunchKernelGGL(shadow_wave_check_kernel, dim3(1), dim3(256), 0, stream, d_wave_thr, waves, d_out_dist, d_out_n, k, d_flag);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_vec_scan_f16(orama_ctx* ctx, const F16ScanArgs& a_in, hipStream_t stream) {
    static const uint32_t k2dbg = [] { const char* e = orama::dev_env("ORAMA_K2_DBG"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    F16ScanArgs a = a_in;
    if (!a.out_dense && !a.wave_lists) a.dbg = k2dbg;  // timing ablation of the filter-mode launches
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries, "vec_scan_f16: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF16MaxQ, "vec_scan_f16: q=%u outside [1, %u]", a.q, kF16MaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f16: bad row range");
    if (a.wave_lists) {
        const uint32_t waves = vec_scan_f16_fused_waves(ctx, a);
        ORAMA_REQUIRE(waves > 0, "vec_scan_f16: the fused mode does not apply to these arguments");
        if (a.row_begin == a.row_end) return ORAMA_OK;
        ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
        const uint32_t ksteps = f16_kpad(a.dim) / 16;
        const size_t lds_bytes = (size_t)ksteps * 2 * 16 + 16 * sizeof(float);
        static const int dbg = [] { const char* e = orama::dev_env("ORAMA_K1H_DBG"); return e ? std::atoi(e) : 0; }();
        if (dbg == 1) hipLaunchKernelGGL((vec_scan_f16_solo_fused_kernel<12, 1>), dim3(waves / 4), dim3(256), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
        else if (dbg == 2) hipLaunchKernelGGL((vec_scan_f16_solo_fused_kernel<12, 2>), dim3(waves / 4), dim3(256), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
        else if (dbg == 3) hipLaunchKernelGGL((vec_scan_f16_solo_fused_kernel<12, 3>), dim3(waves / 4), dim3(256), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
        else hipLaunchKernelGGL((vec_scan_f16_solo_fused_kernel<12, 0>), dim3(waves / 4), dim3(256), lds_bytes, stream, a, ksteps,
                           f16_tile_bytes(a.dim));
        ORAMA_HIP_TRY(hipGetLastError());
        return ORAMA_OK;
    }
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count),
                  "vec_scan_f16: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f16: filter needs row_doc");
    if (a.row_begin == a.row_end) return ORAMA_OK;
    const uint32_t kpad = f16_kpad(a.dim);
    const uint32_t ksteps = kpad / 16;
    if (a.solo && a.q <= 4 && ksteps % 8 == 0) {  // K1h: no matrix cores for a handful of queries (shadow scans only)
        ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
        const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
        const dim3 grid(blocks_for(tiles, kWavesPerBlock, (uint32_t)ctx->compute_units));
        const int nq = a.q == 1 ? 1 : (a.q == 2 ? 2 : 4);
        const size_t lds_bytes = (size_t)ksteps * 2 * nq * 16 + 16 * sizeof(float);
        if (ctx->f16_solo == 2 && ksteps % 12 == 0 && nq <= 2) {  // K1's loop shape: 256-thread blocks, a few per CU
            static const int bpc = [] { const char* e = orama::dev_env("ORAMA_F16_SOLO_BPC"); return e ? std::atoi(e) : 2; }();
            const dim3 g2(blocks_for((tiles + 1) / 2, 4, (uint32_t)ctx->compute_units * (uint32_t)bpc));
            if (nq == 1) hipLaunchKernelGGL((vec_scan_f16_solo2_kernel<1, 12>), g2, dim3(256), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
            else hipLaunchKernelGGL((vec_scan_f16_solo2_kernel<2, 12>), g2, dim3(256), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
            ORAMA_HIP_TRY(hipGetLastError());
            return ORAMA_OK;
        }
        if (nq == 1) hipLaunchKernelGGL((vec_scan_f16_solo_kernel<1, 8, 3>), grid, dim3(kBlock), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
        else if (nq == 2) hipLaunchKernelGGL((vec_scan_f16_solo_kernel<2, 8, 3>), grid, dim3(kBlock), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
        else hipLaunchKernelGGL((vec_scan_f16_solo_kernel<4, 8, 3>), grid, dim3(kBlock), lds_bytes, stream, a, ksteps, f16_tile_bytes(a.dim));
        ORAMA_HIP_TRY(hipGetLastError());
        return ORAMA_OK;
    }
    const int nqt = a.q <= 32 ? 1 : 2;
    a.stage_cap = vec_scan_f16_stage_entries(a.dim, nqt);
    const size_t lds_bytes = vec_scan_f16_lds_bytes(a.dim, nqt, a.stage_cap);
    ORAMA_REQUIRE(a.stage_cap >= 128 && lds_bytes <= kF16LdsLimit, "vec_scan_f16: dim %u too large for the LDS query tile", a.dim);
    ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    const dim3 grid(blocks_for(tiles, kWavesPerBlock, (uint32_t)ctx->compute_units));
    // chunk geometry: preferred (KC, NBUF) from the context tuning, constrained by ksteps % KC == 0
    int kc = ctx->f16_kc, nbuf = ctx->f16_nbuf;
    if (ksteps % (uint32_t)kc != 0) kc = 8;
#define ORAMA_F16_LAUNCH(NQT_, KC_, NB_)                                                                   \
    do {                                                                                                   \
        static bool attr_done = false;                                                                     \
        if (!attr_done) {                                                                                  \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_kernel<NQT_, KC_, NB_, false>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));    \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_kernel<NQT_, KC_, NB_, true>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));    \
            attr_done = true;                                                                              \
        }                                                                                                  \
        if (a.out_dense)                                                                                   \
            hipLaunchKernelGGL((vec_scan_f16_kernel<NQT_, KC_, NB_, true>), grid, dim3(kBlock), lds_bytes, stream, a, \
                               ksteps, f16_tile_bytes(a.dim));                                             \
        else                                                                                               \
            hipLaunchKernelGGL((vec_scan_f16_kernel<NQT_, KC_, NB_, false>), grid, dim3(kBlock), lds_bytes, stream, a, \
                               ksteps, f16_tile_bytes(a.dim));                                             \
    } while (0)
#define ORAMA_F16_DISPATCH(NQT_)                                  \
    do {                                                          \
        if (kc == 16 && nbuf == 2) ORAMA_F16_LAUNCH(NQT_, 16, 2); \
        else if (kc == 12 && nbuf == 3) ORAMA_F16_LAUNCH(NQT_, 12, 3); \
        else if (kc == 12) ORAMA_F16_LAUNCH(NQT_, 12, 2);         \
        else if (nbuf == 4) ORAMA_F16_LAUNCH(NQT_, 8, 4);         \
        else if (nbuf == 3) ORAMA_F16_LAUNCH(NQT_, 8, 3);         \
        else ORAMA_F16_LAUNCH(NQT_, 8, 2);                        \
    } while (0)
    if (nqt == 1) ORAMA_F16_DISPATCH(1);
    else ORAMA_F16_DISPATCH(2);
#undef ORAMA_F16_DISPATCH
#undef ORAMA_F16_LAUNCH
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_f16_seed_candidates(const float* best_dist, const uint32_t* best_row, const uint32_t* best_n,
                               uint32_t q, uint32_t k, float* tau, float* cand_dist, uint32_t* cand_row,
                               uint32_t* cand_count, uint64_t cand_stride, hipStream_t stream, const float* tau_cap) {
    hipLaunchKernelGGL(f16_seed_candidates_kernel, dim3(q), dim3(256), 0, stream, best_dist, best_row, best_n,
                       k, tau, cand_dist, cand_row, cand_count, cand_stride, tau_cap);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

int launch_f16_remember_kth(const float* out_dist, const uint32_t* out_n, uint32_t k, uint32_t q, float* cap, hipStream_t stream) {
    hipLaunchKernelGGL(f16_remember_kth_kernel, dim3((q + 255) / 256), dim3(256), 0, stream, out_dist, out_n, k, q, cap);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
