// vec_f32_mfma.hip — K1m: batched-query cosine scan over the plain fp32 store on the gfx950 matrix cores.
//
// GEMM shape per launch: scores[rows x Q] = corpus[rows x K] · queriesᵀ[K x Q], Q <= 32, never materialised.  The store is
// the reference's own layout — row-major f32, one row = dim x 4 contiguous bytes (embedding_field.rs:232-237 hands over
// Vec<Vec<f32>>) — so, unlike K2's fragment-tiled fp16 store, the HBM layout is NOT the operand layout of
// v_mfma_f32_32x32x2_f32 (lane = row, one float per lane and k): a wave
//   1. streams its 32-row tile in chunks of 32 floats per row with FULL-LINE loads (one global_load_dwordx4 moves 8 rows x
//      128 B: lane l holds floats 4 (l & 7) .. +3 of row l >> 3) into a register ring — 5 chunks = 20 KiB per wave in
//      flight, 160 KiB per CU, no LDS reserved for data that has not arrived;
//   2. turns a landed chunk through a 4-KiB LDS transposer of its own (ds_write_b128 / ds_read_b128, XOR-swizzled by
//      (row >> 1) & 7 so that both directions are bank-conflict free — MI355X_MICROARCH.md §LDS lane groups) into A
//      fragments: lane (h = l >> 5, r = l & 31) receives floats 8c + 4h .. +3 of row r for c = 0..3;
//   3. multiplies them against the query fragments that sit in LDS in the same k order ([k / 8][lane = (h, query)][4 floats],
//      lane-linear ds_read_b128): MFMA e of step c consumes k = 8c + e (h = 0) and k = 8c + 4 + e (h = 1) of both operands —
//      any k order is a valid dot product as long as A and B agree, and it is FIXED, so the distance of a (row, query) pair
//      does not depend on the batch it was asked in;
//   4. applies K2's epilogue on the accumulator registers (1 - s / (|x||q|), tombstones, per-query threshold test, passing
//      rows staged in LDS and appended to the candidate lists in bulk).
// No inter-wave synchronisation after the prologue: a wave's LDS operations execute in order, the transposer is private.
//
// Roofline: HBM — but only just.  One v_mfma_f32_32x32x2_f32 takes 64 cycles of a SIMD's matrix pipe (16 per CU) for 32 rows
// x 2 k = 256 B of corpus: 16 B per clock per CU = 8.2 TB/s at 2.0 GHz over 256 CUs, against an HBM peak of 8.0 TB/s.  At 32
// queries the matrix pipe must be ~85-100 % busy to keep up with the memory system; fewer queries do not make it cheaper (the
// instruction computes all 32 columns).  Algorithmic bytes = rows x dim x 4 per launch.  mfma_frac is quoted against the
// 157.3 TF f32-input peak (MI355X_MICROARCH.md).
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
constexpr int NBUF = (int)kF32MfmaRing;
constexpr int kLoads = 4;  // global_load_dwordx4 per chunk (8 rows x 128 B each)

template <bool DENSE>
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
    __syncthreads();
    if (tid < 32) {  // |q|: f32, one fmaf chain in k order
        const uint32_t j = tid;
        float ss = 0.0f;
        for (uint32_t k = 0; k < nc * 32; ++k) {
            const uint32_t cc = k >> 3, h = (k >> 2) & 1, e = k & 3;
            const float x = reinterpret_cast<const float*>(lds + (size_t)(cc * 64 + h * 32 + j) * 16)[e];
            ss = fmaf(x, x, ss);
        }
        qinv[j] = ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f;
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
    constexpr int kMetaSlots = NBUF + 1;
    constexpr uint32_t kWaveFixed = 4096u + (uint32_t)kMetaSlots * kF16MetaBytes + 256u;
    const uint32_t cap = a.stage_cap;
    const uint32_t wave_off = uniform_u32(frag_total * 16u + 64u * (uint32_t)sizeof(float) + wave_in_block * (kWaveFixed + 12u * cap));
    char* tr = lds + wave_off;
    char* meta = tr + 4096;
    const uint32_t meta_addr = uniform_u32((uint32_t)(size_t)(__attribute__((address_space(3))) char*)meta);
    uint32_t* hist = reinterpret_cast<uint32_t*>(meta + (size_t)kMetaSlots * kF16MetaBytes);
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
        // the 256-byte metadata record (32 x 1/|x| + the tombstone word) of the tile of the NEXT chunk, straight into LDS
        // (see vec_f16.hip load_meta for why this is an asm statement and how its completion is ordered with the ring)
        const uint32_t* src = meta_dead_lane ? a.dead + ld_tile : meta_norm + ld_tile * 32;
        asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, off"
                     :
                     : "v"(src), "{m0}"(meta_addr + m_w * kF16MetaBytes)
                     : "memory");
        m_w = m_w + 1 == kMetaSlots ? 0 : m_w + 1;
    };
    auto load_chunk = [&](f4* b) {
        const char* p = base + ld_tile * tile_bytes + (uint64_t)ld_c * (kF32MfmaChunk * 4u) + lane_off;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) b[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + (size_t)i * 8u * row_pitch));
        if (ld_more) {
            --ld_more;
            if (++ld_c == nc) {
                ld_c = 0;
                ld_tile += tile_step;
            }
        }
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
                    const float nr = *reinterpret_cast<const float*>(meta + (size_t)m_r * kF16MetaBytes + (size_t)i * 4);
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

    auto compute_chunk = [&](const f4* b) {
        if (cp_c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        }
        // through the transposer: rows as they arrived (8 rows x 128 B per load) -> A fragments (lane = row)
        wave_fence();
#pragma unroll
        for (int i = 0; i < kLoads; ++i) *reinterpret_cast<f4*>(tr + ((i & 1) ? waddr1 : waddr0) + i * 1024) = b[i];
        wave_fence();
        f4 af[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) af[c] = *reinterpret_cast<const f4*>(tr + raddr[c]);
        const char* bl = lds + ((size_t)cp_c * 4 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4 bv = *reinterpret_cast<const f4*>(bl + (size_t)c * 1024);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][e], bv[e], acc, 0, 0, 0);
        }
        const bool tile_done = ++cp_c == nc;
        if (tile_done) {
            finish_tile(cp_tile);
            cp_c = 0;
            cp_tile += tile_step;
        }
        m_r = m_r + 1 == kMetaSlots ? 0 : m_r + 1;
    };

    const uint64_t total = my_tiles * nc;
    load_meta();  // the record of the first tile (the one "group -1" would have brought)
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) load_chunk(buf[b]);
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

}  // namespace

int launch_vec_scan_f32_mfma(orama_ctx* ctx, const F16ScanArgs& a_in, hipStream_t stream) {
    F16ScanArgs a = a_in;
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries, "vec_scan_f32_mfma: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF32MfmaMaxQ, "vec_scan_f32_mfma: q=%u outside [1, %u]", a.q, kF32MfmaMaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f32_mfma: bad row range");
    ORAMA_REQUIRE(vec_scan_f32_mfma_supports(a.dim, a.metric), "vec_scan_f32_mfma: dim %u / metric %d not supported", a.dim, a.metric);
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count), "vec_scan_f32_mfma: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f32_mfma: filter needs row_doc");
    if (a.row_begin == a.row_end) return ORAMA_OK;
    a.stage_cap = vec_scan_f32_mfma_stage_entries(a.dim);
    const size_t lds_bytes = vec_scan_f32_mfma_lds_bytes(a.dim, a.stage_cap);
    ORAMA_REQUIRE(a.stage_cap >= 128 && lds_bytes <= kF16LdsLimit, "vec_scan_f32_mfma: dim %u too large for the LDS query tile", a.dim);
    static bool attr_done = false;
    if (!attr_done) {
        ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f32_mfma_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f32_mfma_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    ProfScope prof(&ctx->prof, "vec_scan_f32_mfma", stream);
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    const dim3 grid(blocks_for(tiles, kWavesPerBlock, (uint32_t)ctx->compute_units));
    const uint32_t nc = a.dim / kF32MfmaChunk;
    const uint64_t tile_bytes = (uint64_t)a.dim * 4u * 32u;
    if (a.out_dense)
        hipLaunchKernelGGL((vec_scan_f32_mfma_kernel<true>), grid, dim3(kBlock), lds_bytes, stream, a, nc, tile_bytes);
    else
        hipLaunchKernelGGL((vec_scan_f32_mfma_kernel<false>), grid, dim3(kBlock), lds_bytes, stream, a, nc, tile_bytes);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
