// fulltext.hip — BM25F scoring (K3), hybrid combine (K5) and top-n entry points.
#include "common.hpp"
#include "select.hpp"

using namespace orama;

extern "C" {

int orama_top_n(orama_ctx* ctx, const uint64_t* doc, const float* score, uint64_t n, uint32_t top_k,
                uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
    ORAMA_REQUIRE(ctx && out_n, "null argument");
    *out_n = 0;
    if (top_k == 0 || n == 0) return ORAMA_OK;
    ORAMA_REQUIRE(doc && score && out_ids && out_scores, "null argument");
    ORAMA_REQUIRE(top_k <= kSelectMaxK, "top_k %u exceeds the supported maximum %u", top_k,
                  kSelectMaxK);
    ORAMA_REQUIRE(n < 0xffffffffull, "top_n limited to 2^32-1 entries");
    ORAMA_HIP_TRY(hipSetDevice(ctx->device));
    ScratchLease sc(ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    ORAMA_TRY(sc->misc0.reserve((size_t)n * 8));
    ORAMA_TRY(sc->dist.reserve((size_t)n * 4));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc0.p, doc, (size_t)n * 8, hipMemcpyHostToDevice, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->dist.p, score, (size_t)n * 4, hipMemcpyHostToDevice, s));
    ORAMA_TRY(sc->sel_state.reserve(sizeof(SelectState)));
    ORAMA_TRY(sc->sel_keys.reserve(sizeof(unsigned long long) * top_k));
    ORAMA_TRY(sc->out_ids.reserve((size_t)top_k * 8));
    ORAMA_TRY(sc->out_val.reserve((size_t)top_k * 4));
    ORAMA_TRY(sc->out_n.reserve(4));
    SelectPlan p;
    p.vals = sc->dist.as<float>();
    p.stride = n;
    p.n = (uint32_t)n;
    p.q = 1;
    p.k = top_k;
    p.descending = true;
    p.id_map = sc->misc0.as<uint64_t>();
    p.state = sc->sel_state.as<SelectState>();
    p.keys = sc->sel_keys.as<unsigned long long>();
    p.out_ids = sc->out_ids.as<uint64_t>();
    p.out_val = sc->out_val.as<float>();
    p.out_n = sc->out_n.as<uint32_t>();
    ORAMA_TRY(launch_select(ctx, p, s));
    ORAMA_TRY(sc->h_out.reserve((size_t)top_k * 12 + 4));
    char* h = sc->h_out.as<char>();
    ORAMA_HIP_TRY(hipMemcpyAsync(h, sc->out_ids.p, (size_t)top_k * 8, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)top_k * 8, sc->out_val.p, (size_t)top_k * 4,
                                 hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipMemcpyAsync(h + (size_t)top_k * 12, sc->out_n.p, 4, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    uint32_t cnt = *reinterpret_cast<uint32_t*>(h + (size_t)top_k * 12);
    memcpy(out_ids, h, (size_t)cnt * 8);
    memcpy(out_scores, h + (size_t)top_k * 8, (size_t)cnt * 4);
    *out_n = cnt;
    return ORAMA_OK;
}

// --- the entry points below are implemented in a later milestone of this round -------------
#define ORAMA_NOT_YET(name)                                   \
    do {                                                      \
        set_error(name ": not implemented in this build yet"); \
        return ORAMA_ERR_UNSUPPORTED;                         \
    } while (0)

int orama_bm25_score(orama_ctx*, const orama_ntf_entry*, uint32_t, const orama_bm25_params*,
                     const uint64_t*, const float*, uint64_t, uint64_t*, float*, uint32_t*,
                     uint64_t*) {
    ORAMA_NOT_YET("orama_bm25_score");
}
int orama_post_create(orama_ctx*, orama_post**) { ORAMA_NOT_YET("orama_post_create"); }
void orama_post_destroy(orama_post*) {}
int orama_post_build(orama_post*, const uint64_t*, uint64_t, uint32_t, const float*, uint32_t,
                     const uint32_t*, const uint64_t*, const uint64_t*, const uint32_t*,
                     const uint32_t*) {
    ORAMA_NOT_YET("orama_post_build");
}
int orama_post_set_omc(orama_post*, const uint64_t*, const float*, uint64_t) {
    ORAMA_NOT_YET("orama_post_set_omc");
}
int orama_post_search(orama_post*, const orama_term_ref*, uint32_t, float, const orama_bm25_params*,
                      const uint64_t*, uint64_t, int, uint64_t*, float*, uint32_t*, uint64_t*) {
    ORAMA_NOT_YET("orama_post_search");
}
int orama_post_search_hybrid(orama_post*, const orama_term_ref*, uint32_t, float,
                             const orama_bm25_params*, const uint64_t*, uint64_t, const uint64_t*,
                             const float*, uint32_t, int, uint64_t*, float*, uint32_t*, uint64_t*) {
    ORAMA_NOT_YET("orama_post_search_hybrid");
}
int orama_hybrid_combine(orama_ctx*, const uint64_t*, const float*, uint64_t, const uint64_t*,
                         const float*, uint64_t, uint32_t, uint64_t*, float*, uint32_t*, uint64_t*) {
    ORAMA_NOT_YET("orama_hybrid_combine");
}

}  // extern "C"
