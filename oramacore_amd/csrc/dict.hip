// dict.hip — term-dictionary expansion on the device.
//
// In the reference the dictionary step of collect_contributions lives in the third-party string storage (an FST
// walked with a prefix / Levenshtein automaton; behaviour visible in src/tests/fulltext_search.rs:603-753 — a
// non-exact token matches every term it is a prefix of — and :956-1018 — `tolerance: 1` finds "main" for "mxin").
// Here the sorted term list of a field is resident in HBM and one kernel tests every term against the token:
// thread per term, prefix compare + Myers' bit-parallel edit distance (the token's 256 x u64 match masks are built
// once per workgroup in LDS; tokens up to 64 bytes).  A 2^20-term dictionary is ~10 MB: the scan is a few
// microseconds — next to the postings traffic of the query the dictionary step disappears, fuzzy or not.
// Distances are over BYTES (exact for ASCII; declared deviation for multi-byte UTF-8, where the reference's
// automaton presumably counts scalar values).
#include <algorithm>

#include "common.hpp"

using namespace orama;

struct orama_dict {
    orama_ctx* ctx = nullptr;
    uint32_t n_terms = 0;
    DevBuf blob, offsets;
};

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kMaxToken = 64;

__global__ __launch_bounds__(kThreads) void dict_expand_kernel(const uint8_t* __restrict__ blob,
                                                               const uint32_t* __restrict__ offsets, uint32_t n_terms,
                                                               const uint8_t* __restrict__ token, uint32_t tlen,
                                                               int exact, uint32_t tolerance, uint32_t capacity,
                                                               uint32_t* __restrict__ out_idx,
                                                               uint32_t* __restrict__ out_count) {
    __shared__ unsigned long long peq[256];
    __shared__ uint8_t tok[kMaxToken];
    for (int i = threadIdx.x; i < 256; i += kThreads) peq[i] = 0ull;
    for (uint32_t i = threadIdx.x; i < tlen; i += kThreads) tok[i] = token[i];
    __syncthreads();
    if (threadIdx.x == 0)
        for (uint32_t i = 0; i < tlen; ++i) peq[tok[i]] |= 1ull << i;
    __syncthreads();
    const unsigned long long top = tlen ? (1ull << (tlen - 1)) : 0ull;
    for (uint32_t t = blockIdx.x * kThreads + threadIdx.x; t < n_terms; t += gridDim.x * kThreads) {
        const uint32_t b = offsets[t], len = offsets[t + 1] - b;
        const uint8_t* s = blob + b;
        bool hit = false;
        if (len >= tlen && (!exact || len == tlen)) {  // equality / prefix
            bool eq = true;
            for (uint32_t i = 0; i < tlen && eq; ++i) eq = s[i] == tok[i];
            hit = eq;
        }
        if (!hit && !exact && tolerance > 0) {
            const uint32_t diff = len > tlen ? len - tlen : tlen - len;
            if (diff <= tolerance) {
                if (tlen == 0) {
                    hit = len <= tolerance;
                } else {  // Myers 1999, global distance: the horizontal delta shifted in at the top row is +1
                    unsigned long long pv = ~0ull, mv = 0ull;
                    uint32_t score = tlen;
                    for (uint32_t i = 0; i < len; ++i) {
                        const unsigned long long eq = peq[s[i]];
                        const unsigned long long xv = eq | mv;
                        const unsigned long long xh = (((eq & pv) + pv) ^ pv) | eq;
                        unsigned long long ph = mv | ~(xh | pv);
                        unsigned long long mh = pv & xh;
                        if (ph & top) ++score;
                        if (mh & top) --score;
                        ph = (ph << 1) | 1ull;
                        mh = mh << 1;
                        pv = mh | ~(xv | ph);
                        mv = ph & xv;
                    }
                    hit = score <= tolerance;
                }
            }
        }
        if (hit) {
            const uint32_t pos = atomicAdd(out_count, 1u);
            if (pos < capacity) out_idx[pos] = t;
        }
    }
}

}  // namespace

extern "C" {

int orama_dict_create(orama_ctx* ctx, const uint8_t* blob, const uint32_t* offsets, uint32_t n_terms,
                      orama_dict** out) {
    ORAMA_REQUIRE(ctx && out, "null argument");
    *out = nullptr;
    ORAMA_REQUIRE(offsets && (n_terms == 0 || blob || offsets[n_terms] == 0), "null argument");
    ORAMA_REQUIRE(offsets[0] == 0, "offsets must start at 0");
    for (uint32_t i = 0; i < n_terms; ++i) {
        ORAMA_REQUIRE(offsets[i] <= offsets[i + 1], "offsets must be non-decreasing");
        if (i > 0) {  // strictly ascending byte-wise: the dictionary order list ids are handed out in
            const uint32_t la = offsets[i] - offsets[i - 1], lb = offsets[i + 1] - offsets[i];
            const int c = memcmp(blob + offsets[i - 1], blob + offsets[i], std::min(la, lb));
            ORAMA_REQUIRE(c < 0 || (c == 0 && la < lb), "terms must be strictly ascending (term %u)", i);
        }
    }
    ORAMA_ON_DEVICE(ctx->device);
    std::unique_ptr<orama_dict> d(new (std::nothrow) orama_dict());
    if (!d) {
        set_error("out of host memory");
        return ORAMA_ERR_OOM;
    }
    d->ctx = ctx;
    d->n_terms = n_terms;
    const size_t nb = offsets[n_terms];
    ORAMA_TRY(d->blob.reserve(std::max<size_t>(nb, 16)));
    ORAMA_TRY(d->offsets.reserve(((size_t)n_terms + 1) * 4));
    if (nb) ORAMA_HIP_TRY(hipMemcpy(d->blob.p, blob, nb, hipMemcpyHostToDevice));
    ORAMA_HIP_TRY(hipMemcpy(d->offsets.p, offsets, ((size_t)n_terms + 1) * 4, hipMemcpyHostToDevice));
    *out = d.release();
    return ORAMA_OK;
}

void orama_dict_destroy(orama_dict* d) {
    if (!d) return;
    ::orama::DeviceScope ORAMA_CAT_(dev_scope__, __LINE__)(d->ctx->device);
    (void)hipDeviceSynchronize();
    delete d;
}

int orama_dict_expand(orama_dict* d, const uint8_t* token, uint32_t token_len, int exact, uint32_t tolerance,
                      uint32_t capacity, uint32_t* out_terms, uint32_t* out_n) {
    ORAMA_REQUIRE(d && out_n && (token || token_len == 0), "null argument");
    *out_n = 0;
    ORAMA_REQUIRE(capacity == 0 || out_terms, "null output");
    ORAMA_REQUIRE(token_len <= kMaxToken, "token longer than %u bytes", kMaxToken);
    if (d->n_terms == 0) return ORAMA_OK;
    ORAMA_ON_DEVICE(d->ctx->device);
    ScratchLease sc(d->ctx);
    ORAMA_TRY(sc.init());
    hipStream_t s = sc->stream;
    ORAMA_TRY(sc->misc0.reserve((size_t)capacity * 4 + 16));
    ORAMA_TRY(sc->misc1.reserve(kMaxToken + 16));
    ORAMA_TRY(sc->h_in.reserve(kMaxToken));
    ORAMA_TRY(sc->h_out.reserve((size_t)capacity * 4 + 16));
    if (token_len) memcpy(sc->h_in.p, token, token_len);
    uint32_t* d_count = sc->misc0.as<uint32_t>();
    uint32_t* d_idx = d_count + 4;
    ORAMA_HIP_TRY(hipMemsetAsync(d_count, 0, 4, s));
    if (token_len) ORAMA_HIP_TRY(hipMemcpyAsync(sc->misc1.p, sc->h_in.p, token_len, hipMemcpyHostToDevice, s));
    uint32_t blocks = (d->n_terms + kThreads - 1) / kThreads;
    blocks = std::min<uint32_t>(blocks, (uint32_t)d->ctx->compute_units * 8u);
    hipLaunchKernelGGL(dict_expand_kernel, dim3(blocks), dim3(kThreads), 0, s, d->blob.as<uint8_t>(),
                       d->offsets.as<uint32_t>(), d->n_terms, sc->misc1.as<uint8_t>(), token_len, exact, tolerance,
                       capacity, d_idx, d_count);
    ORAMA_HIP_TRY(hipGetLastError());
    ORAMA_HIP_TRY(hipMemcpyAsync(sc->h_out.p, d_count, (size_t)capacity * 4 + 16, hipMemcpyDeviceToHost, s));
    ORAMA_HIP_TRY(hipStreamSynchronize(s));
    const uint32_t total = sc->h_out.as<uint32_t>()[0];
    const uint32_t n = std::min(total, capacity);
    const uint32_t* h = sc->h_out.as<uint32_t>() + 4;
    std::copy(h, h + n, out_terms);
    std::sort(out_terms, out_terms + n);  // dictionary order (the order the host lookup yields)
    *out_n = total;
    return ORAMA_OK;
}

}  // extern "C"
