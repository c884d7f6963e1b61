// vec_f32_mfma.hpp — K1m: batched-query cosine scan over the PLAIN fp32 store on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, exact — one fmaf chain per (row, query)).
// north_star: "MFMA-backed batched-query x corpus GEMM when Q>1" for the reference's own dtype
// (Vec<f32> rows, src/collection_manager/sides/read/index/embedding_field.rs:66,88,250-278).
#pragma once

#include "common.hpp"
#include "vec_f16.hpp"  // F16ScanArgs: the same dense-head / threshold-filter interface as K2 (a.tiled = the f32 row-major rows)

namespace orama {

constexpr uint32_t kF32MfmaMaxQ = 32;       // one 32-column MFMA tile of queries per corpus pass
constexpr uint32_t kF32MfmaChunk = 32;      // floats of every row a wave moves per chunk (32 rows x 128 B = 4 KiB)
constexpr uint32_t kF32MfmaWaves = 8;       // waves per workgroup (one workgroup per CU: the queries take most of the LDS)
constexpr uint32_t kF32MfmaRing = 4;        // chunks in the register ring (comparison builds also launch 6 and 8)
constexpr uint32_t kF32MfmaMetaBytes = 144; // tile metadata record in LDS: 32 x 1/|x| (128 B), the tombstone word, padding to 16 B
// metadata records in flight: one per load group issued ahead of the epilogue = ring + 3
inline constexpr uint32_t f32_mfma_meta_slots(uint32_t ring) { return ring + 3; }
// rows a search may read past the published count: a partial last tile is read whole (its rows beyond row_end are masked
// in the epilogue), so the f32 arrays carry one tile of slack (vec_store.hip: matrix_bytes / norm_bytes)
constexpr uint64_t kF32MfmaSlackRows = 32;

// LDS: the queries as B fragments ((dim / 8) KiB: [k / 8][lane = (k-half, query)][4 floats]), 32 x 1/|q|, and per wave a
// 4-KiB transposer, the ring of metadata records, a 64-bin histogram and the staging area of passing rows.
inline size_t vec_scan_f32_mfma_lds_bytes(uint32_t dim, uint32_t stage_entries, uint32_t ring = kF32MfmaRing) {
    return (size_t)(dim / 8) * 1024 + 64 * sizeof(float) +
           kF32MfmaWaves * (4096 + (size_t)f32_mfma_meta_slots(ring) * kF32MfmaMetaBytes + 256 + 3 * (size_t)stage_entries * sizeof(uint32_t));
}
inline uint32_t vec_scan_f32_mfma_stage_entries(uint32_t dim, uint32_t ring = kF32MfmaRing) {
    const size_t fixed = vec_scan_f32_mfma_lds_bytes(dim, 0, ring);
    if (fixed >= kF16LdsLimit) return 0;
    const size_t e = ((kF16LdsLimit - fixed) / (kF32MfmaWaves * 3 * sizeof(uint32_t))) & ~(size_t)63;
    return e < 128 ? 0u : (uint32_t)(e > 1024 ? 1024 : e);
}
// cosine, rows of whole chunks whose query tile leaves room for the staging areas (dim <= 864 in 160 KiB of LDS; the deepest ring
// a comparison build launches must fit as well)
inline bool vec_scan_f32_mfma_supports(uint32_t dim, int metric) {
    return metric == ORAMA_METRIC_COSINE && dim >= kF32MfmaChunk && dim % kF32MfmaChunk == 0 && vec_scan_f32_mfma_stage_entries(dim, 8) >= 128;
}

// ---- K1x (vec_f32_cvt.hip): the same scan with the rows rounded to fp16 in registers and v_mfma_f32_32x32x16_f16 — an APPROXIMATE
// candidate scan (|d - d_K1| <= kShadowEps) over the plain fp32 store, HBM-bound, <= 64 queries per pass.
constexpr uint32_t kF32CvtMaxQ = 64;
constexpr uint32_t kF32CvtRing = 6;
constexpr uint32_t kF32CvtTransposerBytes = 2048;  // 32 rows x 32 halves per wave
inline size_t vec_scan_f32_cvt_lds_bytes(uint32_t dim, int nqt, uint32_t stage_entries, uint32_t ring = kF32CvtRing) {
    return (size_t)(dim / 16) * (size_t)nqt * 1024 + 64 * sizeof(float) +
           kF32MfmaWaves * (kF32CvtTransposerBytes + (size_t)f32_mfma_meta_slots(ring) * kF32MfmaMetaBytes + 256 +
                            3 * (size_t)stage_entries * sizeof(uint32_t));
}
inline uint32_t vec_scan_f32_cvt_stage_entries(uint32_t dim, int nqt, uint32_t ring = kF32CvtRing) {
    const size_t fixed = vec_scan_f32_cvt_lds_bytes(dim, nqt, 0, ring);
    if (fixed >= kF16LdsLimit) return 0;
    const size_t e = ((kF16LdsLimit - fixed) / (kF32MfmaWaves * 3 * sizeof(uint32_t))) & ~(size_t)63;
    return e < 128 ? 0u : (uint32_t)(e > 1024 ? 1024 : e);
}
// queries one K1x pass takes at this dimension: 64 while two column tiles of fp16 fragments fit the LDS, else 32, else none
inline uint32_t vec_scan_f32_cvt_max_q(uint32_t dim) {
    return vec_scan_f32_cvt_stage_entries(dim, 2) >= 128 ? 64u : vec_scan_f32_cvt_stage_entries(dim, 1) >= 128 ? 32u : 0u;
}
inline bool vec_scan_f32_cvt_supports(uint32_t dim, int metric) {
    return metric == ORAMA_METRIC_COSINE && dim >= kF32MfmaChunk && dim % kF32MfmaChunk == 0 && vec_scan_f32_cvt_max_q(dim) > 0;
}
int launch_vec_scan_f32_cvt(orama_ctx* ctx, const F16ScanArgs& a, hipStream_t stream);

// K1m.  a.tiled = the fp32 rows (row-major, [n][dim]); a.q <= 32; dense or filter mode as launch_vec_scan_f16.
// Algorithmic HBM traffic: (row_end - row_begin) * dim * 4 bytes per launch (serves all q queries).
int launch_vec_scan_f32_mfma(orama_ctx* ctx, const F16ScanArgs& a, hipStream_t stream);

}  // namespace orama
