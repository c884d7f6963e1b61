// vec_f32_cvt.hip — K1x: the candidate scan of fp32 batches with the rows converted to fp16 IN REGISTERS (round 6).
//
// K1m (vec_f32_mfma.hip) multiplies f32 by f32 on the matrix cores and is bound by them: one v_mfma_f32_32x32x2_f32 per 256 B
// of corpus = 7.4 M cycles per 10 M x 768 pass, 5.3-5.5 ms at the clock the package's power limit leaves.  Since the batch path
// answers with K1's bits anyway (vec_store.hip search_enqueue_f32_batch: the scan only PROPOSES, rerank_f32_kernel decides, an
// unproven list is re-answered by K1), the proposal may be approximate as long as its error has a bound — and the fp16 two-stage
// plan (K1s) already lives on one: both operands rounded to fp16 = 2^-11 relative per element, <= 1.0e-3 on the cosine with the
// exact f32 norms used here (kShadowEps = 2.5e-3 is the plan's bound, proven for its rounded norms).  This kernel is that plan
// WITHOUT the second copy of the rows: it reads the plain store's fp32 rows (the reference's own layout, embedding_field.rs:66,88),
// rounds them to fp16 (RNE) on their way through the LDS transposer and multiplies with v_mfma_f32_32x32x16_f16 — 16 x the rate
// of the f32 instruction, 4 matrix instructions per chunk instead of 16 x 4.  The matrix pipe is idle most of the time; the pass
// is bound by HBM again, and the query tile in LDS (2 bytes per element) holds 64 queries where K1m's holds 32.
//
// Data path, per wave (8 per workgroup, one workgroup per CU), tile = 32 rows:
//   1. full-line loads as K1m: one global_load_dwordx4 = 8 rows x 128 B, chunk = 32 rows x 32 floats, register ring of NBUF chunks;
//   2. v_cvt (RNE) + ds_write_b64 into a private 2-KiB transposer [row][32 halves], 16-byte slots XOR-swizzled by (row >> 2) & 3 so
//      that the A-fragment reads (lane (h, r): halves 16 s + 8 h .. + 7 of row r, ds_read_b128) are bank-conflict free;
//   3. B fragments [k step][query tile][lane][8 halves] in LDS (the layout of K2: 96 KiB for 64 queries x 768), 1/|q| from the
//      f32 queries;
//   4. K2's epilogue over NQT query tiles (threshold filter, tombstones, candidates staged in LDS, bulk append).
// Roofline: HBM.  Algorithmic bytes = rows x dim x 4 per pass of <= 64 queries.  Rows and queries must be fp16-safe (|x_i| < 6e4,
// |x|^2 >= 1e-4: tracked per store at insert, per query on the device) — otherwise K1m.
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
using f16async::h8;
using f16async::wave_or_u32;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = (int)kF32MfmaWaves * 64;
constexpr int kWavesPerBlock = (int)kF32MfmaWaves;
constexpr int kLoads = 4;

template <bool DENSE, int NQT, int NBUF>
__global__ __launch_bounds__(kBlock) void vec_scan_f32_cvt_kernel(F16ScanArgs a, uint32_t nc /* chunks per row = dim / 32 */,
                                                                   uint64_t tile_bytes /* 32 rows */) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t ksteps = nc * 2;                  // 16 k per matrix instruction
    const uint32_t frag_total = ksteps * NQT * 64;   // 16-byte query fragments: [k step][query tile][lane]
    float* qinv = reinterpret_cast<float*>(lds + (size_t)frag_total * 16);

    // ---- prologue: queries (f32, HBM/L2) -> fp16 B fragments in LDS; 1/|q| from the f32 values
    for (uint32_t idx = tid; idx < frag_total; idx += kBlock) {
        const uint32_t ks = idx / (NQT * 64);
        const uint32_t rem = idx - ks * (NQT * 64);
        const uint32_t qt = rem >> 6, l = rem & 63;
        const uint32_t j = qt * 32 + (l & 31);
        const uint32_t k0 = ks * 16 + (l >> 5) * 8;
        h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)((j < a.q) ? a.queries[(size_t)j * a.dim + k0 + e] : 0.0f);
        *reinterpret_cast<h8*>(lds + (size_t)idx * 16) = v;
    }
    // 1 / |q| from the f32 values: one wave per query at a time, lanes strided over the dimensions (a serial loop per query — 768
    // dependent steps — cost 40-100 us per launch: more than the whole dense head's corpus traffic, profiles/r06_driver_command_*)
    for (uint32_t j = (uint32_t)tid >> 6; j < NQT * 32u; j += kWavesPerBlock) {
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
    const uint64_t tile0 = t_first + gw;
    const uint64_t tile_step = gwaves;
    const uint64_t my_tiles = (n_tiles - gw + gwaves - 1) / gwaves;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    const uint32_t row_pitch = a.dim * 4u;

    f16v acc[NQT];
    f4 buf[NBUF][kLoads];
    uint64_t ld_tile = tile0;
    uint32_t ld_c = 0;
    uint64_t ld_more = my_tiles * nc - 1;
    uint64_t cp_tile = tile0;
    uint32_t cp_c = 0;

    // per-wave LDS: transposer (2 KiB) | metadata ring | histogram | staging area
    const uint32_t wave_in_block = uniform_u32((uint32_t)tid >> 6);
    constexpr int kMetaSlots = (int)f32_mfma_meta_slots(NBUF);
    constexpr uint32_t kMetaBytes = kF32MfmaMetaBytes;
    constexpr uint32_t kWaveFixed = kF32CvtTransposerBytes + (uint32_t)kMetaSlots * kMetaBytes + 256u;
    const uint32_t cap = a.stage_cap;
    const uint32_t wave_off = uniform_u32(frag_total * 16u + 64u * (uint32_t)sizeof(float) + wave_in_block * (kWaveFixed + 12u * cap));
    char* tr = lds + wave_off;
    char* meta = tr + kF32CvtTransposerBytes;
    const uint32_t meta_addr = uniform_u32((uint32_t)(size_t)(__attribute__((address_space(3))) char*)meta);
    uint32_t* hist = reinterpret_cast<uint32_t*>(meta + (size_t)kMetaSlots * kMetaBytes);
    uint32_t* stage = hist + 64;
    uint32_t m_w = 0, m_r = 0;

    // transposer addresses (bytes): [row][64 B], 16-byte slot s of row r stored at slot s ^ ((r >> 2) & 3).
    //   write, load i (rows 8 i .. 8 i + 7): row = 8 i + (l >> 3); the lane's 4 halves = half (l & 1) of slot (l & 7) >> 1;
    //          (row >> 2) & 3 = (2 i + (l >> 5)) & 3
    //   read, k step s of the chunk: row = l & 31, slot 2 s + (l >> 5)
    uint32_t waddr[kLoads];
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        const uint32_t row = 8u * i + ((uint32_t)lane >> 3);
        const uint32_t slot = (((uint32_t)lane & 7u) >> 1) ^ ((2u * i + ((uint32_t)lane >> 5)) & 3u);
        waddr[i] = row * 64u + slot * 16u + ((uint32_t)lane & 1u) * 8u;
    }
    const uint32_t rd_row = (uint32_t)lane & 31u, rd_h = (uint32_t)lane >> 5;
    uint32_t raddr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) raddr[s] = rd_row * 64u + (((2u * s + rd_h) ^ ((rd_row >> 2) & 3u)) << 4);
    const uint32_t lane_off = ((uint32_t)lane >> 3) * row_pitch + ((uint32_t)lane & 7u) * 16u;

    const uint32_t* meta_norm = reinterpret_cast<const uint32_t*>(a.inv_norm) + (lane & 31);
    const bool meta_dead_lane = lane == 32 && a.dead != nullptr;
    auto load_meta = [&]() {  // (vec_f32_mfma.hip load_meta: the record of the tile of the NEXT chunk by LDS-DMA, lanes 0..32)
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
        const uint32_t adv = ld_more != 0 ? 1u : 0u;  // (branch-free: past the end the cursor re-reads the wave's last chunk)
        ld_more -= adv;
        const uint32_t c1 = ld_c + adv;
        const bool wrap = c1 == nc;
        ld_c = wrap ? 0u : c1;
        ld_tile += wrap ? tile_step : 0ull;
        load_meta();
    };

    float qi_reg[NQT], tau_reg[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const uint32_t j = qt * 32 + (lane & 31);
        qi_reg[qt] = qinv[j];
        tau_reg[qt] = (a.tau && j < a.q) ? a.tau[j] : 0.0f;
    }

    uint32_t staged = 0;  // wave-uniform
    auto wave_fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    auto bin_load = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto bin_store = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto flush = [&]() {  // K2's bulk append (vec_f16.hip flush)
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

    // `start` = 16 qt + r of the first accumulator row still to be looked at; returns 16 NQT when the tile is done, else the
    // position at which the staging area ran full (the caller flushes and calls again).  Dense mode: always done.
    auto epilogue = [&](uint64_t tile, uint32_t start) -> uint32_t {
        f4 n4[4];
        uint32_t dead_word;
        {
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
                : "v"(mine), "v"(rec), "v"(acc[NQT - 1][15]));
            if (!a.dead) dead_word = 0u;
        }
        const uint32_t hi4 = (lane >> 5) ? 4u : 0u;
        const bool full = tile * 32 + 32 <= a.row_end;
        const uint32_t left = full ? 32u : (uint32_t)(a.row_end - tile * 32);
        auto dist_of = [&](float dot, float n, float qv) -> float { return __builtin_fmaf(-dot, n * qv, 1.0f); };
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            if (!DENSE && (uint32_t)(qt + 1) * 16u <= start) continue;  // done before the flush
            const uint32_t j = qt * 32 + ((uint32_t)lane & 31u);
            const bool live = j < a.q;
            const float qi = qi_reg[qt];
            if constexpr (DENSE) {
                if (!live) continue;
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
            const float tau = tau_reg[qt];
            float best = __builtin_huge_valf();
#pragma unroll
            for (int r = 0; r < 16; ++r) best = fminf(best, __builtin_fmaf(-acc[qt][r], n4[r >> 2][r & 3] * qi, 1.0f));
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(live && best < tau) == 0, 1)) continue;
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
                    const uint32_t i = ((r & 3u) + 8u * (r >> 2)) + hi4;
                    const float nr = *reinterpret_cast<const float*>(meta + (size_t)m_r * kMetaBytes + (size_t)i * 4);
                    float dot = 0.0f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr)
                        if (r == (uint32_t)rr) dot = acc[qt][rr];
                    const float dist = dist_of(dot, nr, qi_s);
                    const uint32_t pos =
                        staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    stage[pos] = __float_as_uint(dist);
                    stage[cap + pos] = (uint32_t)(tile * 32) + i;
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
        if (!DENSE && staged > cap - 64) {
            wave_fence();
            flush();
        }
    };

    // stage(g + 1): the landed chunk, rounded to fp16 (RNE), through the transposer into the A fragments of its two k steps, and the
    // query fragments of those steps; multiply(g): 2 x NQT matrix instructions.  One chunk apart, as in K1m.
    h8 af[2][2], bf[2][2][NQT];
    uint32_t st_c = 0;
    auto stage_chunk = [&](const f4* b, h8* a_out, h8 (*b_out)[NQT]) {
        wave_fence();
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const f2 lo = f2{b[i][0], b[i][1]}, hi = f2{b[i][2], b[i][3]};
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 l2 = __builtin_convertvector(lo, h2), u2 = __builtin_convertvector(hi, h2);
            *reinterpret_cast<h4*>(tr + waddr[i]) = h4{l2[0], l2[1], u2[0], u2[1]};
        }
        wave_fence();
#pragma unroll
        for (int s = 0; s < 2; ++s) a_out[s] = *reinterpret_cast<const h8*>(tr + raddr[s]);
        const char* bl = lds + ((size_t)st_c * 2 * NQT * 64 + lane) * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) b_out[s][qt] = *reinterpret_cast<const h8*>(bl + (size_t)(s * NQT + qt) * 1024);
        st_c = st_c + 1 == nc ? 0 : st_c + 1;
    };
    uint64_t tiles_left = my_tiles;
    auto multiply = [&](const h8* a_in, const h8 (*b_in)[NQT]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) acc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_in[s], b_in[s][qt], acc[qt], 0, 0, 0);
    };
    auto finish_if_done = [&]() {
        if (++cp_c == nc) {
            if (tiles_left) {
                --tiles_left;
                finish_tile(cp_tile);
            }
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qt][r] = 0.0f;
            cp_c = 0;
            cp_tile += tile_step;
        }
        m_r = m_r + 1 == kMetaSlots ? 0 : m_r + 1;
    };

    const uint64_t total = my_tiles * nc;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[qt][r] = 0.0f;
    load_meta();
#pragma unroll
    for (int b = 0; b < NBUF; ++b) load_chunk(buf[b]);
    stage_chunk(buf[0], af[0], bf[0]);
    load_chunk(buf[0]);
    static_assert(NBUF % 2 == 0, "the fragment double buffer is indexed statically inside the unrolled ring");
    for (uint64_t g = 0; g < total; g += NBUF) {
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
            stage_chunk(buf[(b + 1) % NBUF], af[(b + 1) & 1], bf[(b + 1) & 1]);
            load_chunk(buf[(b + 1) % NBUF]);
            multiply(af[b & 1], bf[b & 1]);
            finish_if_done();
        }
    }
    if (!DENSE && staged) {
        wave_fence();
        flush();
    }
}

}  // namespace

int launch_vec_scan_f32_cvt(orama_ctx* ctx, const F16ScanArgs& a_in, hipStream_t stream) {
    F16ScanArgs a = a_in;
    a.dbg = 0;
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries, "vec_scan_f32_cvt: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF32CvtMaxQ, "vec_scan_f32_cvt: q=%u outside [1, %u]", a.q, kF32CvtMaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f32_cvt: bad row range");
    ORAMA_REQUIRE(vec_scan_f32_cvt_supports(a.dim, a.metric), "vec_scan_f32_cvt: dim %u / metric %d not supported", a.dim, a.metric);
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count), "vec_scan_f32_cvt: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f32_cvt: filter needs row_doc");
    if (a.row_begin == a.row_end) return ORAMA_OK;
    const int nqt = a.q <= 32 ? 1 : 2;
    ProfScope prof(&ctx->prof, "vec_scan_f32_cvt", stream);
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    const dim3 grid(blocks_for(tiles, kWavesPerBlock, (uint32_t)ctx->compute_units));
    const uint32_t nc = a.dim / kF32MfmaChunk;
    const uint64_t tile_bytes = (uint64_t)a.dim * 4u * 32u;
#define ORAMA_K1X_LAUNCH(NQT_) ORAMA_K1X_LAUNCH_RING(NQT_, (int)kF32CvtRing)
#define ORAMA_K1X_LAUNCH_RING(NQT_, RING_)                                                                                            \
    do {                                                                                                                             \
        a.stage_cap = vec_scan_f32_cvt_stage_entries(a.dim, NQT_, RING_);                                                            \
        const size_t lds_bytes = vec_scan_f32_cvt_lds_bytes(a.dim, NQT_, a.stage_cap, RING_);                                        \
        ORAMA_REQUIRE(a.stage_cap >= 128 && lds_bytes <= kF16LdsLimit, "vec_scan_f32_cvt: dim %u too large for the LDS query tile", a.dim); \
        static bool attr_done = false;                                                                                               \
        if (!attr_done) {                                                                                                            \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f32_cvt_kernel<false, NQT_, RING_>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                              \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f32_cvt_kernel<true, NQT_, RING_>),  \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                              \
            attr_done = true;                                                                                                        \
        }                                                                                                                            \
        if (a.out_dense)                                                                                                             \
            hipLaunchKernelGGL((vec_scan_f32_cvt_kernel<true, NQT_, RING_>), grid, dim3(kBlock), lds_bytes, stream, a, nc, tile_bytes); \
        else                                                                                                                         \
            hipLaunchKernelGGL((vec_scan_f32_cvt_kernel<false, NQT_, RING_>), grid, dim3(kBlock), lds_bytes, stream, a, nc, tile_bytes); \
    } while (0)
#if ORAMA_COMPARISON_KERNELS
    static const int ring = [] { const char* e = orama::dev_env("ORAMA_K1X_RING"); return e ? std::atoi(e) : 0; }();  // A/B: 4 / 8
    if (ring == 4) {
        if (nqt == 1) ORAMA_K1X_LAUNCH_RING(1, 4);
        else ORAMA_K1X_LAUNCH_RING(2, 4);
    } else if (ring == 8 && nqt == 1) {
        ORAMA_K1X_LAUNCH_RING(1, 8);
    } else
#endif
    if (nqt == 1) ORAMA_K1X_LAUNCH(1);
    else ORAMA_K1X_LAUNCH(2);
#undef ORAMA_K1X_LAUNCH
#undef ORAMA_K1X_LAUNCH_RING
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
