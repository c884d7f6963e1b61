// vec_f32_mfma.hip — K1m: batched-query cosine scan over the plain fp32 store on the gfx950 matrix cores.
//
// GEMM shape per launch: scores[rows x Q] = corpus[rows x K] · queriesᵀ[K x Q], Q <= 32, never materialised.  The store is
// the reference's own layout — row-major f32, one row = dim x 4 contiguous bytes (embedding_field.rs:232-237 hands over
// Vec<Vec<f32>>) — so, unlike K2's fragment-tiled fp16 store, the HBM layout is NOT the operand layout of
// v_mfma_f32_32x32x2_f32 (lane = row, one float per lane and k): a wave
//   1. streams its 32-row tile in chunks of 32 floats per row with FULL-LINE loads (one global_load_dwordx4 moves 8 rows x
//      128 B: lane l holds floats 4 (l & 7) .. +3 of row l >> 3) into a register ring — 4 chunks of 4 KiB per wave, 3-4 in
//      flight, no LDS reserved for data that has not arrived;
//   2. turns a landed chunk through a 4-KiB LDS transposer of its own (ds_write_b128 / ds_read_b128, XOR-swizzled by
//      (row >> 1) & 7 so that both directions are bank-conflict free — MI355X_MICROARCH.md §LDS lane groups; measured:
//      SQ_LDS_BANK_CONFLICT 19 K cycles of 480 M active) into A fragments: lane (h = l >> 5, r = l & 31) receives floats
//      8c + 4h .. +3 of row r for c = 0..3;
//   3. multiplies them against the query fragments that sit in LDS in the same k order ([k / 8][lane = (h, query)][4 floats],
//      lane-linear ds_read_b128): MFMA e of step c consumes k = 8c + e (h = 0) and k = 8c + 4 + e (h = 1) of both operands —
//      any k order is a valid dot product as long as A and B agree, and it is FIXED, so the distance of a (row, query) pair
//      does not depend on the batch it was asked in;
//   4. applies K2's epilogue on the accumulator registers (1 - s / (|x||q|), tombstones, per-query threshold test, passing
//      rows staged in LDS and appended to the candidate lists in bulk).
// Steps 2 (for chunk g + 1) and 3 (for chunk g) are ONE straight-line block in which every matrix instruction is followed by
// one memory operation of the other chunk (sched_group_barrier): a wave issues in order, and anything that can wait — the
// vmcnt for a landed chunk, an LDS round trip — must not stand in front of matrix instructions that are ready.
// No inter-wave synchronisation after the prologue: a wave's LDS operations execute in order, the transposer is private.
//
// Roofline: the f32 MATRIX PIPE at 32 queries, HBM below that only on paper.  One v_mfma_f32_32x32x2_f32 takes 64 cycles of a
// SIMD's matrix pipe for 32 rows x 2 k = 256 B of corpus whatever the number of live query columns: 16 B per clock per CU =
// 7.4 M cycles for 10 M x 768 — 3.1 ms at the 2.4 GHz the data sheet's 157.3 TF assumes, 4.1-4.4 ms at the 1.67-1.82 GHz the
// package's power limit leaves this kernel (0.96-1.22 kW, PPT residency 85-100 %: profiles/r06_k1m_*.log), against 3.84 ms
// of HBM time.  Measured 5.33-5.50 ms per pass = 76-80 % of the matrix pipe at the clock it ran at, 0.70-0.72 of the HBM peak.
// `roofline` quotes both; algorithmic bytes = rows x dim x 4 per launch, flops = 2 x 32 x rows x dim.
#include "vec_f32_mfma.hpp"

#include <cstdlib>

#include "device_utils.hpp"
#include "vec_f16_async.hpp"

namespace orama {

namespace {

uint32_t blocks_for(uint64_t items, uint32_t per_block, uint32_t cap) {
    uint64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (uint32_t)b;
}

using f16async::f16v;
using f16async::f4;
using f16async::wave_or_u32;

constexpr int kBlock = (int)kF32MfmaWaves * 64;
constexpr int kWavesPerBlock = (int)kF32MfmaWaves;
constexpr int kLoads = 4;  // global_load_dwordx4 per chunk (8 rows x 128 B each)

// NBUF: chunks in the register ring.  BDBL: all four query fragments of a chunk are fetched a chunk ahead (16 more registers)
// instead of one step ahead.  (Measured in one lease, profiles/r06_k1m_variants_ab*.log: rings of 4 / 6 / 8 chunks, either
// fragment schedule, and loads issued as pairs of chunks — 256 contiguous bytes of a row at a time — all land within 2 % of each
// other: the kernel is bound by the matrix pipe at the clock the package's power limit leaves, not by what is in flight.  The
// pair form is gone; 4 chunks + fragments a chunk ahead is the default.)
// ABL (comparison builds): timing ablations with wrong answers — 16 no matrix instructions, 128 half of them, 64 no HBM traffic
// (every load re-reads the wave's first chunk: L2 hits).  Compile-time: a run-time test would split the ring's straight-line trips.
template <bool DENSE, int NBUF, bool BDBL, int ABL = 0>
__global__ __launch_bounds__(kBlock) void vec_scan_f32_mfma_kernel(F16ScanArgs a, uint32_t nc /* chunks per row = dim / 32 */,
                                                                    uint64_t tile_bytes /* 32 rows */) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t frag_total = nc * 4 * 64;  // 16-byte query fragments: [k / 8][lane]
    float* qinv = reinterpret_cast<float*>(lds + (size_t)frag_total * 16);

    // ---- prologue: queries (f32, HBM/L2) -> B fragments in LDS
    for (uint32_t idx = tid; idx < frag_total; idx += kBlock) {
        const uint32_t cc = idx >> 6, l = idx & 63;
        const uint32_t j = l & 31, k0 = cc * 8 + (l >> 5) * 4;
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (j < a.q) v = *reinterpret_cast<const f4*>(a.queries + (size_t)j * a.dim + k0);
        *reinterpret_cast<f4*>(lds + (size_t)idx * 16) = v;
    }
    // 1 / |q|: one wave per query at a time, lanes strided over the dimensions (the answer is K1's whatever the last ulp here: a
    // serial chain per query cost ~40 us per launch)
    for (uint32_t j = (uint32_t)tid >> 6; j < 32u; j += kWavesPerBlock) {
        float ss = 0.0f;
        if (j < a.q)
            for (uint32_t k = lane; k < a.dim; k += 64) {
                const float x = a.queries[(size_t)j * a.dim + k];
                ss = fmaf(x, x, ss);
            }
        ss = wave_sum(ss);
        if (lane == 0) qinv[j] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
    }
    __syncthreads();

    // ---- tiles of this wave
    const uint32_t gw = uniform_u32(blockIdx.x * kWavesPerBlock + (tid >> 6));
    const uint32_t gwaves = gridDim.x * kWavesPerBlock;
    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;
    const uint64_t n_tiles = t_end - t_first;
    if (gw >= n_tiles) return;
    const uint64_t tile0 = t_first + gw;  // wave w takes tiles w, w + W, ...: the chip sweeps one window
    const uint64_t tile_step = gwaves;
    const uint64_t my_tiles = (n_tiles - gw + gwaves - 1) / gwaves;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    const uint32_t row_pitch = a.dim * 4u;

    f16v acc;
    f4 buf[NBUF][kLoads];
    uint64_t ld_tile = tile0;  // (tile, chunk) cursor of the NEXT load
    uint32_t ld_c = 0;
    uint64_t ld_more = my_tiles * nc - 1;  // chunks still to load after the one under the cursor
    uint64_t cp_tile = tile0;              // cursor of the NEXT compute
    uint32_t cp_c = 0;

    // per-wave LDS: transposer | metadata ring | histogram | staging area
    const uint32_t wave_in_block = uniform_u32((uint32_t)tid >> 6);
    constexpr int kMetaSlots = (int)f32_mfma_meta_slots(NBUF);  // records in flight: one per load group issued ahead of the epilogue
    constexpr uint32_t kMetaBytes = kF32MfmaMetaBytes;
    constexpr uint32_t kWaveFixed = 4096u + (uint32_t)kMetaSlots * kMetaBytes + 256u;
    const uint32_t cap = a.stage_cap;
    const uint32_t wave_off = uniform_u32(frag_total * 16u + 64u * (uint32_t)sizeof(float) + wave_in_block * (kWaveFixed + 12u * cap));
    char* tr = lds + wave_off;
    char* meta = tr + 4096;
    const uint32_t meta_addr = uniform_u32((uint32_t)(size_t)(__attribute__((address_space(3))) char*)meta);
    uint32_t* hist = reinterpret_cast<uint32_t*>(meta + (size_t)kMetaSlots * kMetaBytes);
    uint32_t* stage = hist + 64;
    uint32_t m_w = 0, m_r = 0;  // next metadata slot to write / to read (wave-uniform)

    // transposer addresses (bytes).  slot(row, quad) = row * 8 + (quad ^ ((row >> 1) & 7)), 16 B each.
    //   write, load i (rows 8i .. 8i+7): row = 8i + (l >> 3), quad = l & 7; (row >> 1) & 7 = (4 (i & 1) + (l >> 4)) & 7
    //   read, step c: row = l & 31, quad = 2c + (l >> 5)
    const uint32_t wr_row = (uint32_t)lane >> 3, wr_quad = (uint32_t)lane & 7u;
    const uint32_t waddr0 = wr_row * 128u + ((wr_quad ^ (((uint32_t)lane >> 4) & 7u)) << 4);
    const uint32_t waddr1 = wr_row * 128u + ((wr_quad ^ ((4u + ((uint32_t)lane >> 4)) & 7u)) << 4);
    const uint32_t rd_row = (uint32_t)lane & 31u, rd_h = (uint32_t)lane >> 5, rd_f = (rd_row >> 1) & 7u;
    uint32_t raddr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) raddr[c] = rd_row * 128u + (((2u * c + rd_h) ^ rd_f) << 4);
    // this lane's byte offset inside a tile chunk: row (l >> 3) of the first 8, floats 4 (l & 7) ..
    const uint32_t lane_off = wr_row * row_pitch + wr_quad * 16u;

    const uint32_t* meta_norm = reinterpret_cast<const uint32_t*>(a.inv_norm) + (lane & 31);
    const bool meta_dead_lane = lane == 32 && a.dead != nullptr;
    auto load_meta = [&]() {
        // the metadata record (32 x 1/|x| + the tombstone word: 132 of 144 bytes) of the tile of the NEXT chunk, straight into LDS
        // (see vec_f16.hip load_meta for why this is an asm statement and how its completion is ordered with the ring)
        // (lanes 0..32 only: the instruction writes 4 bytes per ACTIVE lane at m0 + 4 lane — with all 64 lanes active it would
        // write 256 bytes into a 144-byte slot and run over the record the epilogue of the oldest chunk in flight reads next.
        // The mask is set inside the statement — an `if` around it would be a branch, and the ring's trips must stay one
        // straight-line block; the main loop runs with every lane active, so -1 restores it.)
        const uint32_t* src = meta_dead_lane ? a.dead + ld_tile : meta_norm + ld_tile * 32;
        asm volatile("s_mov_b32 exec_hi, 1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\ts_mov_b32 exec_hi, -1"
                     :
                     : "v"(src), "{m0}"(meta_addr + m_w * kMetaBytes)
                     : "memory");
        m_w = m_w + 1 == kMetaSlots ? 0 : m_w + 1;
    };
    auto load_chunk = [&](f4* b) {
        const char* p = base + ld_tile * tile_bytes + (uint64_t)ld_c * (kF32MfmaChunk * 4u) + lane_off;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) b[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + (size_t)i * 8u * row_pitch));
        // advance the cursor unless it stands on the wave's last chunk (branch-free: scalar selects)
        const uint32_t adv = (ld_more != 0 && !(ABL & 64)) ? 1u : 0u;
        ld_more -= adv;
        const uint32_t c1 = ld_c + adv;
        const bool wrap = c1 == nc;
        ld_c = wrap ? 0u : c1;
        ld_tile += wrap ? tile_step : 0ull;
        load_meta();
    };

    const float qi_reg = qinv[lane & 31];
    const float tau_reg = (a.tau && (uint32_t)(lane & 31) < a.q) ? a.tau[lane & 31] : 0.0f;

    uint32_t staged = 0;  // wave-uniform
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    auto bin_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto bin_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    // K2's bulk append (vec_f16.hip flush): rank the staged rows within their query through an LDS histogram, one global
    // atomic per query with rows, then the rows go to base[j] + rank
    auto flush = [&]() {
        bin_store(&hist[lane], 0u);
        wave_fence();
        for (uint32_t i = lane; i < staged; i += 64) {
            const uint32_t j = stage[2 * cap + i];
            bool keep = true;
            if (a.allow) {
                const uint64_t doc = a.row_doc[stage[cap + i]];
                keep = doc < a.allow_bits && ((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
            }
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

    // `start` = index of the first accumulator row still to be looked at; returns 16 when the tile is done, else the position at
    // which the staging area ran full (the caller flushes and calls again).  Dense mode: always done.
    auto epilogue = [&](uint64_t tile, uint32_t start) -> uint32_t {
        f4 n4[4];
        uint32_t dead_word;
        {
            // this lane's 16 accumulator rows are (r & 3) + 8 (r >> 2) + 4 (l >> 5): four 16-byte reads of the record.  The
            // statement takes an accumulator as a (never used) operand so that it stays behind the tile's last MFMA and with
            // it behind the counted wait that proves the record has landed.
            const uint32_t rec = meta_addr + m_r * kMetaBytes;
            const uint32_t mine = rec + ((lane >> 5) ? 16u : 0u);
            asm volatile(
                "ds_read_b128 %0, %5\n\t"
                "ds_read_b128 %1, %5 offset:32\n\t"
                "ds_read_b128 %2, %5 offset:64\n\t"
                "ds_read_b128 %3, %5 offset:96\n\t"
                "ds_read_b32 %4, %6 offset:128\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(n4[0]), "=&v"(n4[1]), "=&v"(n4[2]), "=&v"(n4[3]), "=&v"(dead_word)
                : "v"(mine), "v"(rec), "v"(acc[15]));
            if (!a.dead) dead_word = 0u;
        }
        const uint32_t hi4 = (lane >> 5) ? 4u : 0u;
        const bool full = tile * 32 + 32 <= a.row_end;  // wave-uniform: only the last tile of the range may be partial
        const uint32_t left = full ? 32u : (uint32_t)(a.row_end - tile * 32);
        const uint32_t j = (uint32_t)lane & 31u;
        const bool live = j < a.q;
        const float qi = qi_reg;
        auto dist_of = [&](float dot, float n, float qv) -> float { return __builtin_fmaf(-dot, n * qv, 1.0f); };
        if constexpr (DENSE) {  // every distance is written, NaN = excluded
            if (!live) return 16u;
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
                    excluded ? __builtin_nanf("") : dist_of(acc[r], n4[r >> 2][r & 3], qi);
            }
            return 16u;
        } else {
            // fast reject: almost no row beats the running k-th best distance
            float best = __builtin_huge_valf();
#pragma unroll
            for (int r = 0; r < 16; ++r) best = fminf(best, __builtin_fmaf(-acc[r], n4[r >> 2][r & 3] * qi, 1.0f));
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && best < tau_reg) == 0, 1)) return 16u;
            float qi_s = qi;
            asm volatile("" : "+v"(qi_s));
            const uint32_t alive = (~dead_word & (left >= 32u ? ~0u : ((1u << left) - 1u))) >> hi4;
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t i = (uint32_t)((r & 3) + 8 * (r >> 2));
                const float d = dist_of(acc[r], n4[r >> 2][r & 3], qi_s);
                m |= ((live && d < tau_reg ? 1u : 0u) & (alive >> i)) << r;
            }
            uint32_t any = wave_or_u32(m);
            if (start) any &= ~0u << start;  // resuming after a flush
#pragma unroll 1
            while (any) {
                const uint32_t r = (uint32_t)__builtin_ctz(any);
                const bool mine = (m >> r) & 1u;
                const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
                const uint32_t n_pass = (uint32_t)__popcll(bal);
                if (staged + n_pass > cap) return r;  // no room: flush, then resume here
                any &= any - 1u;
                if (mine) {
                    const uint32_t i = ((r & 3u) + 8u * (r >> 2)) + hi4;
                    const float nr = *reinterpret_cast<const float*>(meta + (size_t)m_r * kMetaBytes + (size_t)i * 4);
                    // (the accumulator row through a wave-uniform switch: a per-lane index would go through scratch memory)
                    float dot = 0.0f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr)
                        if (r == (uint32_t)rr) dot = acc[rr];
                    const float dist = dist_of(dot, nr, qi_s);
                    const uint32_t pos =
                        staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    stage[pos] = __float_as_uint(dist);
                    stage[cap + pos] = (uint32_t)(tile * 32) + i;
                    stage[2 * cap + pos] = j;
                }
                staged = uniform_u32(staged + n_pass);
            }
            return 16u;
        }
    };
    auto finish_tile = [&](uint64_t tile) {
        uint32_t at = 0;
        while ((at = epilogue(tile, at)) < 16u) {
            wave_fence();
            flush();
        }
        if (!DENSE && staged > cap - 64) {
            wave_fence();
            flush();
        }
    };

    // Two stages per chunk, one chunk apart: stage(g + 1) turns the landed chunk through the transposer into the A fragments and
    // fetches the matching query fragments — all LDS traffic of chunk g + 1 is ISSUED before the 16 matrix instructions of chunk
    // g and completes while they run (1 024 cycles of the SIMD's matrix pipe); the instructions of chunk g wait only for
    // fragments that were requested a whole chunk earlier.  (First version: reads and MFMAs of the same chunk alternated in
    // groups of 8 and both waves of a SIMD sat out the LDS latency twice per chunk — 5.7 ms per 32-query pass at 10 M x 768.)
    f4 af[2][4], bf[2][BDBL ? 4 : 1];  // A fragments of two chunks; query fragments: all four steps, or step 0 only, a chunk ahead
    uint32_t st_c = 0;                  // chunk-in-row of the NEXT stage (the query fragments' k offset)
    auto stage_chunk = [&](const f4* b, f4* a_out, f4* b_out) {
        wave_fence();
#pragma unroll
        for (int i = 0; i < kLoads; ++i) *reinterpret_cast<f4*>(tr + ((i & 1) ? waddr1 : waddr0) + i * 1024) = b[i];
        wave_fence();
#pragma unroll
        for (int c = 0; c < 4; ++c) a_out[c] = *reinterpret_cast<const f4*>(tr + raddr[c]);
        const char* bl = lds + ((size_t)st_c * 4 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < (BDBL ? 4 : 1); ++c) b_out[c] = *reinterpret_cast<const f4*>(bl + (size_t)c * 1024);
        st_c = st_c + 1 == nc ? 0 : st_c + 1;
    };
    uint64_t tiles_left = my_tiles;
    auto multiply = [&](const f4* a_in, const f4* b_in) {
        // (!BDBL: query fragments of steps 1..3 are requested one step — 4 matrix instructions, 256 cycles — ahead of their use)
        const char* bl = lds + ((size_t)cp_c * 4 * 64 + lane) * 16;
        f4 bcur = b_in[0];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f4 bnext = bcur;
            if (c < 3) bnext = BDBL ? b_in[BDBL ? c + 1 : 0] : *reinterpret_cast<const f4*>(bl + (size_t)(c + 1) * 1024);
            if constexpr ((ABL & 16) != 0) {  // no matrix instructions (the data is still consumed)
                acc[c] += a_in[c][0] + a_in[c][1] + a_in[c][2] + a_in[c][3] + bcur[0];
            } else if constexpr ((ABL & 128) != 0) {  // half of the matrix instructions
#pragma unroll
                for (int e = 0; e < 2; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_in[c][e] + a_in[c][e + 2], bcur[e], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_in[c][e], bcur[e], acc, 0, 0, 0);
            }
            bcur = bnext;
        }
    };
    // One trip of the ring = ONE straight-line block: the matrix instructions of chunk g with the memory operations of the stage
    // of chunk g + 1 and of the reload between them, one per matrix instruction.  A wave issues in order: written as "stage, then
    // 16 MFMAs" (rounds 6a/6b of this kernel) the stage's waits (vmcnt for the landed chunk, the LDS round trips) sat in front
    // of sixteen instructions that were ready, and the two waves of a SIMD — same code, same phase — waited together: 94-98
    // cycles per matrix instruction where the pipe needs 64 (scripts/micro/mfma_f32_chain_probe.hip: a dependent chain alone
    // runs at 64.8; 8 LDS reads whose results are waited for before the next 16 make it 74.9).  Interleaved, a non-matrix
    // instruction issues in the shadow of the matrix instruction before it.
    auto interleave = [] {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 DS write
        }
#pragma unroll
        for (int i = 0; i < 4 + (BDBL ? 4 : 1); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
        }
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
        }
    };
    auto finish_if_done = [&]() {
        if (++cp_c == nc) {  // the tile is done (past the wave's last tile the trailing trips multiply the re-read last chunk for nobody)
            if (tiles_left) {
                --tiles_left;
                finish_tile(cp_tile);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            cp_c = 0;
            cp_tile += tile_step;
        }
        m_r = m_r + 1 == kMetaSlots ? 0 : m_r + 1;
    };

    const uint64_t total = my_tiles * nc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    load_meta();  // the record of the first tile (the one "group -1" would have brought)
#pragma unroll
    for (int b = 0; b < NBUF; ++b) load_chunk(buf[b]);
    stage_chunk(buf[0], af[0], bf[0]);
    load_chunk(buf[0]);
    // Chunk x lives in buf[x % NBUF] from its load to its stage; per trip: stage(g + 1), reload the freed buffer, multiply(g).
    // Every trip does all of it (past the end the load cursor re-reads the wave's last chunk, see vec_f16.hip, and stage and
    // multiply work on it for nobody): a skipped part would change the number of loads in flight on one path and the compiler's
    // counted vmcnt waits would fall back to draining the ring at every trip — and would split the block the interleave needs.
    static_assert(NBUF % 2 == 0, "the fragment double buffer is indexed statically inside the unrolled ring");
    for (uint64_t g = 0; g < total; g += NBUF) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            stage_chunk(buf[(b + 1) % NBUF], af[(b + 1) & 1], bf[(b + 1) & 1]);
            load_chunk(buf[(b + 1) % NBUF]);
            multiply(af[b & 1], bf[b & 1]);
            interleave();
            finish_if_done();
        }
    }
    if (!DENSE && staged) {
        wave_fence();
        flush();
    }
}

}  // namespace

int launch_vec_scan_f32_mfma(orama_ctx* ctx, const F16ScanArgs& a_in, hipStream_t stream) {
    F16ScanArgs a = a_in;
#if ORAMA_COMPARISON_KERNELS
    static const uint32_t k1mdbg = [] { const char* e = orama::dev_env("ORAMA_K1M_DBG"); return e ? (uint32_t)std::atoi(e) : 0u; }();
#endif
    a.dbg = 0;
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries, "vec_scan_f32_mfma: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF32MfmaMaxQ, "vec_scan_f32_mfma: q=%u outside [1, %u]", a.q, kF32MfmaMaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f32_mfma: bad row range");
    ORAMA_REQUIRE(vec_scan_f32_mfma_supports(a.dim, a.metric), "vec_scan_f32_mfma: dim %u / metric %d not supported", a.dim, a.metric);
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count), "vec_scan_f32_mfma: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f32_mfma: filter needs row_doc");
    if (a.row_begin == a.row_end) return ORAMA_OK;
    ProfScope prof(&ctx->prof, "vec_scan_f32_mfma", stream);
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    const dim3 grid(blocks_for(tiles, kWavesPerBlock, (uint32_t)ctx->compute_units));
    const uint32_t nc = a.dim / kF32MfmaChunk;
    const uint64_t tile_bytes = (uint64_t)a.dim * 4u * 32u;
#define ORAMA_K1M_LAUNCH(NB_, BDBL_) ORAMA_K1M_LAUNCH_ABL(NB_, BDBL_, 0)
#define ORAMA_K1M_LAUNCH_ABL(NB_, BDBL_, ABL_)                                                                                          \
    do {                                                                                                                             \
        a.stage_cap = vec_scan_f32_mfma_stage_entries(a.dim, NB_);                                                                   \
        const size_t lds_bytes = vec_scan_f32_mfma_lds_bytes(a.dim, a.stage_cap, NB_);                                               \
        ORAMA_REQUIRE(a.stage_cap >= 128 && lds_bytes <= kF16LdsLimit, "vec_scan_f32_mfma: dim %u too large for the LDS query tile", a.dim); \
        static bool attr_done = false;                                                                                               \
        if (!attr_done) {                                                                                                            \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f32_mfma_kernel<false, NB_, BDBL_, ABL_>),    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                              \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f32_mfma_kernel<true, NB_, BDBL_, ABL_>),     \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                              \
            attr_done = true;                                                                                                        \
        }                                                                                                                            \
        if (a.out_dense)                                                                                                             \
            hipLaunchKernelGGL((vec_scan_f32_mfma_kernel<true, NB_, BDBL_, ABL_>), grid, dim3(kBlock), lds_bytes, stream, a, nc, tile_bytes);  \
        else                                                                                                                         \
            hipLaunchKernelGGL((vec_scan_f32_mfma_kernel<false, NB_, BDBL_, ABL_>), grid, dim3(kBlock), lds_bytes, stream, a, nc, tile_bytes); \
    } while (0)
#if ORAMA_COMPARISON_KERNELS
    // A/B builds: ORAMA_K1M_VARIANT = ring depth x 100 + 1 (query fragments a chunk ahead); ORAMA_K1M_DBG = a timing ablation
    static const int variant = [] { const char* e = orama::dev_env("ORAMA_K1M_VARIANT"); return e ? std::atoi(e) : 0; }();
    if (k1mdbg) {
        switch (k1mdbg) {
            case 16: ORAMA_K1M_LAUNCH_ABL(4, true, 16); break;
            case 128: ORAMA_K1M_LAUNCH_ABL(4, true, 128); break;
            case 64: ORAMA_K1M_LAUNCH_ABL(4, true, 64); break;
            case 192: ORAMA_K1M_LAUNCH_ABL(4, true, 192); break;
            default: ORAMA_K1M_LAUNCH_ABL(4, true, 80); break;
        }
    } else switch (variant) {
        case 400: ORAMA_K1M_LAUNCH(4, false); break;
        case 600: ORAMA_K1M_LAUNCH(6, false); break;
        case 601: ORAMA_K1M_LAUNCH(6, true); break;
        case 800: ORAMA_K1M_LAUNCH(8, false); break;
        case 801: ORAMA_K1M_LAUNCH(8, true); break;
        default: ORAMA_K1M_LAUNCH(4, true); break;
    }
#else
    ORAMA_K1M_LAUNCH(4, true);
#endif
#undef ORAMA_K1M_LAUNCH
#undef ORAMA_K1M_LAUNCH_ABL
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
