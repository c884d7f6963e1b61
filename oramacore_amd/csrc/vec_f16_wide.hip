// vec_f16_wide.hip — K2c: the fp16 scan for WIDE query batches (65..256 queries per corpus pass; config C5).
//
// K2 keeps the whole query batch as B fragments in LDS (<= 64 queries fit) and feeds one A fragment into two
// MFMAs.  At 256 queries the batch no longer fits LDS (384 KB) and one pass per 64 queries re-reads the corpus
// four times, so this kernel is a register-blocked GEMM instead:
//
//   workgroup (512 threads, 8 waves, one per CU, persistent) : block tile = 256 rows x 256 queries
//   wave w                                                    : 64 rows x 128 queries = 2 x 4 MFMA tiles,
//                                                               128 accumulator VGPRs, never spilled to memory
//   K loop                                                    : stages of KS = 3 k-steps (2 for 1024 dims); per
//       stage the workgroup moves 24 KB of corpus fragments (HBM) and 24 KB of query fragments (L2-resident,
//       prepared once per launch) straight into a 3-deep LDS ring with global_load_lds_dwordx4 (no staging
//       registers); the two wave groups load alternate stages, each in the shadow of the MFMAs of the stage two
//       positions earlier; one barrier per stage
//   per k-step and wave : 2 A + 4 B fragment reads (ds_read_b128, lane-linear) feed 8 MFMAs (0.75 KB of LDS
//                         traffic per v_mfma_f32_32x32x16_f16 instead of 1 KB)
//
// The corpus is read from HBM exactly once per 256 queries; flops = 2·256·rows·kpad (AI = 256 flop/B against a
// ridge of ~312): the kernel sits at the corner of the roofline.  Measured bound: the CU's L1-miss queue, which the
// corpus fragments share with an equal volume of query fragments coming from L2 (DMA-only ablation 3.6 ms per pass
// of 10 M x 768; MFMA-only bound 1.6 ms; measured 5.3 ms = 0.74 PFLOP/s).  Epilogue (1 - s/|x||q|, tombstones, allow-bit, threshold filter / dense store) is K2's, applied
// to the accumulator registers; the accumulation order over k is K2's too, so a wide batch returns bit-identical
// distances to solo queries.
#include "vec_f16.hpp"

#if ORAMA_COMPARISON_KERNELS  // K2c is a superseded comparison kernel: not part of the product library (see _build.py)

#include <cstdlib>

#include "device_utils.hpp"

namespace orama {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kWB = 512;                    // threads per workgroup
// KS = k-steps per stage: 3 when the padded dimension allows it (768, 384: 48 / 24 k-steps), else 2 (1024: 64)
constexpr int NBUF = 3;                     // LDS ring: stage g lives in buffer g % 3, two stages in flight
// operand A of a stage: 8 row tiles x KS fragments of 1 KiB (24 KiB at KS = 3)
constexpr int op_a_bytes(int ks) { return 8 * ks * 1024; }
// operand B of a stage: 2 query groups x QT query tiles x KS fragments (QT = 4: 256 queries, QT = 2: 128)
constexpr int op_b_bytes(int qt, int ks) { return 2 * qt * ks * 1024; }
constexpr int stage_bytes(int qt, int ks) { return op_a_bytes(ks) + op_b_bytes(qt, ks); }
constexpr int kMetaBytes = 2 * 256 * 4;     // 1/|x| of the block tile's 256 rows, double-buffered over block tiles
constexpr int lds_bytes_for(int qt, int ks) { return NBUF * stage_bytes(qt, ks) + kMetaBytes; }
// DMA scheduling: the two wave groups (waves 0-3 / 4-7) load alternate stages, so a wave only ever has the DMA of
// ONE stage outstanding when it must wait for it — the waits are plain vmcnt(0).  (Counted waits that leave a newer
// stage in flight are not safe: LDS-DMA loads were observed to complete out of issue order.)
// DMA instructions (one 1-KiB fragment each) per wave and loaded stage: A 8 KS / 4 waves; B 2 QT KS / 4 waves.

// s_waitcnt immediates (gfx9 encoding): vmcnt only, expcnt / lgkmcnt untouched
constexpr int kWaitVm0 = 0x0F70;  // vmcnt(0)

// 16 bytes per lane, global -> LDS, asynchronous (completion is counted by vmcnt): lane l's data lands at LDS
// address m0 + 16 l.  Issued through inline asm on purpose: for the builtin the compiler's wait-count pass assumes
// any later LDS read may alias the DMA and drains vmcnt(0) before every ds_read — which serialises the ring.  Here
// the ring discipline guarantees disjoint buffers and the waits are explicit (kWaitVm*).
// m0 is bound as an INPUT operand ("{m0}"): the compiler materialises the LDS address in m0 itself and knows the
// register is live into the statement — nothing reserved is clobbered behind its back.  The leading s_nop is the
// wait state gfx9 requires between a write of m0 and an LDS-DMA instruction that reads it (the hazard recogniser
// cannot see inside the asm string).
__device__ __forceinline__ void dma16(uint64_t saddr_uniform, uint32_t voff, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(saddr_uniform), "{m0}"(lds_addr_uniform)
                 : "memory");
}
// the same with the non-temporal hint, for the corpus stream (read once; +10 % streaming bandwidth on MI355X:
// scripts/micro/stream_probe.hip).  Mixing policies is fine here because every wait is vmcnt(0).
__device__ __forceinline__ void dma16_nt(uint64_t saddr_uniform, uint32_t voff, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                 :
                 : "v"(voff), "s"(saddr_uniform), "{m0}"(lds_addr_uniform)
                 : "memory");
}

// DBG (ablation builds, ORAMA_K2C_DBG, results are garbage — timing only): bit 0 no MFMAs, bit 1 no DMA at all,
// bit 2 no DMA of the query fragments, bit 3 no LDS fragment reads; any bit set skips the epilogue.
template <int QT, int KS, int DBG = 0>
__global__ __launch_bounds__(kWB) void vec_scan_f16_wide_kernel(F16ScanArgs a, const char* __restrict__ bfrag,
                                                                const float* __restrict__ qinv, uint32_t ksteps,
                                                                uint64_t tile_bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // [NBUF][A | B] + tile metadata
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = uniform_u32(tid >> 6);
    const int wr = w & 3;   // row group: row tiles 2wr, 2wr+1 of the block tile
    const int wq = w >> 2;  // query group: query tiles QT wq .. QT wq + QT - 1
    constexpr int kStageBytes = stage_bytes(QT, KS);
    constexpr int kOpBytes = op_a_bytes(KS);
    constexpr int kInstrA = 8 * KS / 4;
    constexpr int kMetaOff = NBUF * kStageBytes;

    const uint64_t t_first = a.row_begin >> 5;
    const uint64_t t_end = (a.row_end + 31) >> 5;            // row tiles [t_first, t_end)
    const uint64_t n_bt = (t_end - t_first + 7) >> 3;        // block tiles of 8 row tiles
    if (blockIdx.x >= n_bt) return;
    const uint64_t my_bt = (n_bt - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const uint32_t S = ksteps / KS;                          // stages per block tile
    const uint64_t total = my_bt * S;
    const char* base = reinterpret_cast<const char*>(a.tiled);
    const float* inv_lds = reinterpret_cast<const float*>(lds + kMetaOff);  // [2][256]

    // ---- stage loader: global -> LDS DMA (global_load_lds_dwordx4), no staging registers.  A wave moves 64
    // consecutive 16-byte pieces per instruction; piece p = (tile p / (KS*64), k-step (p / 64) % KS, lane p % 64),
    // LDS offset 16 p — the fragment layout itself.
    // All addresses are wave-uniform except the lane's 16-byte slot: scalar base (running, no multiplies in the
    // loop) + one constant VGPR offset.
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t vlane = (uint32_t)lane * 16;
    // wave lw = w & 3 of a group moves the 1-KiB fragments t = lw + 4 i of each operand; t = (tile t / KS, k-step t % KS)
    constexpr int kInstrB = 2 * QT * KS / 4;
    static_assert((kInstrA + kInstrB) % KS == 0, "DMA instructions split evenly over the k-steps");
    constexpr int kPerStep = (kInstrA + kInstrB) / KS;
    const int grp = w >> 2, lw = w & 3;
    uint32_t f_lds[kInstrA + kInstrB], a_tl[kInstrA];
    uint64_t f_adr[kInstrA + kInstrB];  // A: offset inside the block tile at k-step 0; B: absolute address at k-step 0
#pragma unroll
    for (int i = 0; i < kInstrA; ++i) {
        const uint32_t t = (uint32_t)lw + 4u * i;
        a_tl[i] = t / KS;
        f_lds[i] = t * 1024;
        f_adr[i] = (uint64_t)(t / KS) * tile_bytes + (uint64_t)(t % KS) * 1024;
    }
#pragma unroll
    for (int i = 0; i < kInstrB; ++i) {
        const uint32_t t = (uint32_t)lw + 4u * i;
        f_lds[kInstrA + i] = kOpBytes + t * 1024;
        f_adr[kInstrA + i] = (uint64_t)(uintptr_t)bfrag + ((uint64_t)(t / KS) * ksteps + (t % KS)) * 1024;
    }
    const uint64_t bt_stride = (uint64_t)gridDim.x * 8 * tile_bytes;
    // load cursor of THIS wave's group: the group loads stages grp, grp + 2, grp + 4, ... of the workgroup's sequence
    uint64_t ld_bt = blockIdx.x;
    uint32_t ld_s = 0, ld_koff = 0, ld_par = 0;
    uint64_t ld_base = (uint64_t)(uintptr_t)base + (t_first + (uint64_t)blockIdx.x * 8) * tile_bytes;  // block tile, k = 0
    const bool partial_last = ((t_end - t_first) & 7) != 0;
    auto advance_cursor = [&]() {
        ld_koff += KS * 1024;
        if (++ld_s == S) {
            ld_s = 0;
            ld_koff = 0;
            ld_bt += gridDim.x;
            ld_base += bt_stride;
            ld_par ^= 1;
        }
    };
    if (grp == 1) advance_cursor();
    // part `step` (0..KS-1) of the DMA of the stage under the cursor, issued in the shadow of the MFMAs of a k-step
    auto issue_part = [&](int buf, int step) {
        const uint32_t lbuf = lds_base + (uint32_t)buf * kStageBytes;
        if (!(DBG & 2) && step == 0 && ld_s == 0 && lw == 0) {
            // first stage of a block tile: also fetch the tile's 256 inverse norms (1 KiB, contiguous; rows past the
            // end of the store read the zero-initialised padding of inv_norm)
            dma16((uint64_t)(uintptr_t)a.inv_norm + (t_first + ld_bt * 8) * 128, vlane,
                  lds_base + kMetaOff + ld_par * 1024);
        }
        const bool clamp = partial_last && ld_bt == n_bt - 1;  // rare: the last block tile has < 8 row tiles
#pragma unroll
        for (int x = 0; x < kPerStep; ++x) {
            const int i = step * kPerStep + x;
            if (DBG & 2) continue;
            if (i < kInstrA) {
                uint64_t sa = ld_base + f_adr[i] + ld_koff;
                if (clamp && t_first + ld_bt * 8 + a_tl[i] >= t_end)  // re-read a valid tile; its rows are masked later
                    sa = (uint64_t)(uintptr_t)base + (t_end - 1) * tile_bytes + (f_adr[i] - (uint64_t)a_tl[i] * tile_bytes) + ld_koff;
                dma16_nt(sa, vlane, lbuf + f_lds[i]);
            } else if (!(DBG & 4)) {
                dma16(f_adr[i] + ld_koff, vlane, lbuf + f_lds[i]);
            }
        }
        if (step == KS - 1) {  // this group's next stage is two stages further
            advance_cursor();
            advance_cursor();
        }
    };

    // ---- per-lane epilogue constants: the 4 query columns of this lane
    float qi_reg[QT], tau_reg[QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        const uint32_t col = (uint32_t)(wq * QT + j) * 32 + (lane & 31);
        qi_reg[j] = qinv[col];
        tau_reg[j] = (a.tau && col < a.q) ? a.tau[col] : 0.0f;
    }

    f16v acc[2][QT];
    const bool l2 = a.metric == ORAMA_METRIC_L2SQ;
    typedef const uint32_t __attribute__((address_space(4))) cu32;
    auto epilogue = [&](uint64_t bt, uint32_t par) {
        const uint32_t hi = (lane >> 5) ? 4u : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t tl = (uint32_t)(wr * 2 + i);
            const uint64_t tile = t_first + bt * 8 + tl;
            if (tile >= t_end) continue;  // wave-uniform
            const uint32_t dead_word = a.dead ? ((cu32*)(uintptr_t)a.dead)[tile] : 0u;  // scalar load
            float nrm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[r] = inv_lds[par * 256 + tl * 32 + (r & 3) + 8 * (r >> 2) + hi];
            const bool full = tile * 32 + 32 <= a.row_end;
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                const uint32_t col = (uint32_t)(wq * QT + j) * 32 + (lane & 31);
                if (col >= a.q) continue;
                const float qi = qi_reg[j];
                const float tau = tau_reg[j];
                if (!a.out_dense) {
                    // filter mode, fast reject: almost no row beats the running k-th best distance, so take the
                    // minimum of the 16 distances first (2-3 VALU per element, no branches) and look closer only
                    // when it passes.  Tombstones / rows past the end are sorted out on the slow path.
                    float best = __builtin_huge_valf();
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        best = fminf(best, l2 ? (qi + nrm[r]) - 2.0f * acc[i][j][r] : 1.0f - acc[i][j][r] * (nrm[r] * qi));
                    if (!(best < tau)) continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t ri = (uint32_t)((r & 3) + 8 * (r >> 2)) + hi;
                    const uint64_t row = tile * 32 + ri;
                    if (!full && row >= a.row_end) continue;
                    const float dist = l2 ? (qi + nrm[r]) - 2.0f * acc[i][j][r] : 1.0f - acc[i][j][r] * (nrm[r] * qi);
                    bool excluded = (dead_word >> ri) & 1u;
                    if (a.out_dense) {
                        if (!excluded && a.allow) {
                            const uint64_t doc = a.row_doc[row];
                            excluded = doc >= a.allow_bits || !((a.allow[doc >> 6] >> (doc & 63)) & 1ull);
                        }
                        a.out_dense[(uint64_t)col * a.dense_stride + (row - a.row_begin)] =
                            excluded ? __builtin_nanf("") : dist;
                    } else if (!excluded && dist < tau) {
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

    h8 fa[2][2], fb[2][QT];  // fragment registers, double-buffered over the k-steps of a stage
    auto load_frags = [&](int buf, int ks, int slot) {
        if (DBG & 8) return;
        const char* la = lds + (size_t)buf * kStageBytes + (size_t)lane * 16;
        const char* lb = la + kOpBytes;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            fa[slot][i] = *reinterpret_cast<const h8*>(la + (size_t)((wr * 2 + i) * KS + ks) * 1024);
#pragma unroll
        for (int j = 0; j < QT; ++j)
            fb[slot][j] = *reinterpret_cast<const h8*>(lb + (size_t)((wq * QT + j) * KS + ks) * 1024);
    };
    // the fragments of k-step 0 are already in fa[0] / fb[0] (issued before the DMA of the next stage)
    auto compute = [&](int buf, bool first, bool prefetch, int pbuf) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) load_frags(buf, ks + 1, (ks + 1) & 1);
            if (ks == 0 && first) {  // wave-uniform: a new block tile starts from C = 0 (no accumulator clearing)
                const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < QT; ++j)
                        if (DBG & 1) {
                            acc[i][j] = zero;
                            asm volatile("" ::"v"(fa[0][i]), "v"(fb[0][j]));
                        } else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], zero, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < QT; ++j)
                        if (DBG & 1)
                            asm volatile("" ::"v"(fa[ks & 1][i]), "v"(fb[ks & 1][j]));
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
            }
            if (prefetch) issue_part(pbuf, ks);  // in the shadow of the MFMAs just issued
        }
    };

    // ---- pipeline: stage g lives in LDS buffer g % 3 and was loaded by wave group g & 1, which issued it while
    // stage g - 2 was being multiplied.
    static_assert(NBUF == 3, "ring of three stages: one being read, two being filled");
    if ((uint64_t)grp < total) {
#pragma unroll
        for (int step = 0; step < KS; ++step) issue_part(grp, step);  // stage grp -> buffer grp
    }
    uint64_t cp_bt = blockIdx.x;
    uint32_t cp_s = 0, cp_par = 0;
    int buf = 0;
    for (uint64_t g = 0; g < total; ++g) {
        const bool mine = ((uint32_t)g & 1u) == (uint32_t)grp;
        if (mine) __builtin_amdgcn_s_waitcnt(kWaitVm0);  // my group's DMA of stage g (its only outstanding loads) landed
        __syncthreads();  // ... everybody's did; everybody is also done reading the buffer of stage g-1
        load_frags(buf, 0, 0);
        // my group now refills the buffer stage g-1 just left with stage g+2: (buf + 2) % 3
        compute(buf, cp_s == 0, mine && g + 2 < total, buf == 0 ? 2 : buf - 1);
        if (++cp_s == S) {
            if (DBG == 0) epilogue(cp_bt, cp_par);
            else {  // keep every accumulator alive
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < QT; ++j)
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
}

}  // namespace

int launch_vec_scan_f16_wide(orama_ctx* ctx, const F16ScanArgs& a, void* d_query_frags, bool prepare,
                             hipStream_t stream) {
    ORAMA_REQUIRE(a.tiled && a.inv_norm && a.queries && d_query_frags, "vec_scan_f16_wide: bad arguments");
    ORAMA_REQUIRE(a.q >= 1 && a.q <= kF16WideMaxQ, "vec_scan_f16_wide: q=%u outside [1, %u]", a.q, kF16WideMaxQ);
    ORAMA_REQUIRE((a.row_begin & 31) == 0 && a.row_begin <= a.row_end, "vec_scan_f16_wide: bad row range");
    ORAMA_REQUIRE(a.out_dense || (a.tau && a.cand_dist && a.cand_row && a.cand_count),
                  "vec_scan_f16_wide: no output mode");
    ORAMA_REQUIRE(!a.allow || a.row_doc, "vec_scan_f16_wide: filter needs row_doc");
    const uint32_t kpad = f16_kpad(a.dim);
    const uint32_t ksteps = kpad / 16;
    ORAMA_REQUIRE(ksteps % 2 == 0, "vec_scan_f16_wide: kpad %u not a multiple of 32", kpad);
    char* bfrag = reinterpret_cast<char*>(d_query_frags);
    float* qinv = reinterpret_cast<float*>(bfrag + (size_t)8 * ksteps * 1024);
    if (prepare) ORAMA_TRY(launch_f16_prepare_queries(a.queries, a.q, a.dim, a.metric, d_query_frags, stream));
    if (a.row_begin == a.row_end) return ORAMA_OK;
    ProfScope prof(&ctx->prof, "vec_scan_f16", stream);
    static_assert(lds_bytes_for(4, 3) <= 160 * 1024 && lds_bytes_for(4, 2) <= 160 * 1024, "K2c LDS budget");
    const uint64_t tiles = ((a.row_end + 31) >> 5) - (a.row_begin >> 5);
    uint64_t blocks = (tiles + 7) / 8;
    if (blocks > (uint64_t)ctx->compute_units) blocks = (uint64_t)ctx->compute_units;
    const dim3 grid((uint32_t)blocks);
    const uint64_t tile_bytes = f16_tile_bytes(a.dim);
    // QT: 2 query tiles per wave up to 128 queries (half the MFMAs and half the query-fragment traffic), else 4
#define ORAMA_WIDE_LAUNCH(QT_, KS_, DBG_)                                                                           \
    do {                                                                                                            \
        static bool attr_done = false;                                                                              \
        if (!attr_done) {                                                                                           \
            ORAMA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vec_scan_f16_wide_kernel<QT_, KS_, DBG_>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));             \
            attr_done = true;                                                                                       \
        }                                                                                                           \
        hipLaunchKernelGGL((vec_scan_f16_wide_kernel<QT_, KS_, DBG_>), grid, dim3(kWB), lds_bytes_for(QT_, KS_),     \
                           stream, a, (const char*)bfrag, (const float*)qinv, ksteps, tile_bytes);                  \
    } while (0)
    const bool ks3 = ksteps % 3 == 0;
    int dbg = 0;
    if (const char* e = orama::dev_env("ORAMA_K2C_DBG")) dbg = std::atoi(e);
    if (dbg && ks3 && a.q > 128 && !a.out_dense) {  // ablation builds of the Q = 256, 768-dim shape (timing only)
        switch (dbg) {
            case 1: ORAMA_WIDE_LAUNCH(4, 3, 1); break;    // DMA + fragment reads, no MFMA
            case 9: ORAMA_WIDE_LAUNCH(4, 3, 9); break;    // DMA only
            case 2: ORAMA_WIDE_LAUNCH(4, 3, 2); break;    // fragment reads + MFMA + barriers, no DMA
            case 4: ORAMA_WIDE_LAUNCH(4, 3, 4); break;    // corpus DMA only + compute
            case 11: ORAMA_WIDE_LAUNCH(4, 3, 11); break;  // barriers only
            case 13: ORAMA_WIDE_LAUNCH(4, 3, 13); break;  // corpus DMA only, no compute
            default: ORAMA_WIDE_LAUNCH(4, 3, 0); break;
        }
    } else if (a.q <= 128) {
        if (ks3) ORAMA_WIDE_LAUNCH(2, 3, 0);
        else ORAMA_WIDE_LAUNCH(2, 2, 0);
    } else {
        if (ks3) ORAMA_WIDE_LAUNCH(4, 3, 0);
        else ORAMA_WIDE_LAUNCH(4, 2, 0);
    }
#undef ORAMA_WIDE_LAUNCH
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama

#endif  // ORAMA_COMPARISON_KERNELS
