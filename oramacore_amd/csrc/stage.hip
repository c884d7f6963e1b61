// stage.hip — small host<->device blocks by a kernel of the caller's stream (round 5).
//
// hipMemcpyAsync between pinned host memory and the device is carried by the SDMA engines.  A copy is then a packet on an
// engine's own in-order queue plus a signal each way between that queue and the stream's compute queue: ~10 us of chain per
// copy for a block of a few KB, and — what cost more — two streams with chunks in flight meet on the engines.  Measured on the
// range scorer's two-chunks-in-flight batch (profiles/r05_k3r_copy_kernel_ab.log): 213 K queries/s in a fresh process, 176 K
// once ANY other stream of the process had used an engine (one orama_hybrid_search call): the device then ran the two
// chunks' chains strictly one after the other (rocprofv3 kernel trace: sum of kernel time / wall 0.86 against 1.52).  With
// the blit kernels of the runtime instead (HSA_ENABLE_SDMA=0) 141 K.  A few workgroups of OURS reading / writing the pinned
// block over PCIe are ordinary dispatches of the chunk's stream — no engine, no cross-queue signal: 221 K before and after.
// Visibility: pinned memory is fine-grained; a dispatch acquires at system scope when it starts and releases when it ends, so
// the host's writes before the launch and the kernel's writes before hipStreamSynchronize returns are seen.
#include "stage.hpp"

namespace orama {

namespace {

struct StageArgs {
    uint32_t* dst[4];
    const uint32_t* src[4];
    uint32_t bytes[4];
    uint32_t first_block[5];  // workgroups [first_block[i], first_block[i+1]) carry part i
    int n;
};

__global__ __launch_bounds__(256) void stage_blocks_kernel(StageArgs a) {
    int part = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.n && blockIdx.x >= a.first_block[i]) part = i;
    uint32_t* __restrict__ dst = a.dst[part];
    const uint32_t* __restrict__ src = a.src[part];
    const uint32_t bytes = a.bytes[part];
    const uint32_t i = (blockIdx.x - a.first_block[part]) * 256u + threadIdx.x;
    const bool wide = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0;
    if (wide) {
        const uint32_t n16 = bytes >> 4;
        if (i < n16) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        if (i == n16)
            for (uint32_t w = n16 << 2; w < (bytes >> 2); ++w) dst[w] = src[w];
    } else {
        const uint32_t n4 = bytes >> 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = i * 4u + j;
            if (w < n4) dst[w] = src[w];
        }
    }
}

}  // namespace

// CONTRACT: the host side of every part is PINNED, MAPPED memory (PinnedBuf: hipHostMallocPortable | hipHostMallocMapped) — the
// kernel form dereferences the host pointer on the device; `kind` only tells the copy form (hipMemcpyAsync) the direction.
// Every caller passes PinnedBuf memory (Scratch::h_in / h_out / h_misc).  Comparison builds CHECK it — a host pointer the runtime does
// not know as host memory takes the copy form instead of faulting on the GPU; the product build does not pay the
// runtime's pointer lookup on the lone-call path (a lone full-text call crosses this function three times).
int stage_blocks(orama_ctx* ctx, const StagePart* parts, int n_parts, hipMemcpyKind kind, hipStream_t s) {
    ORAMA_REQUIRE(n_parts >= 1 && n_parts <= 4, "stage_blocks: 1..4 parts");
    bool by_kernel = ctx->stage_by_kernel;
#if ORAMA_COMPARISON_KERNELS
    if (by_kernel) {
        for (int i = 0; i < n_parts && by_kernel; ++i) {
            if (!parts[i].bytes) continue;
            const void* host = kind == hipMemcpyDeviceToHost ? parts[i].dst : parts[i].src;
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, host) != hipSuccess || at.type != hipMemoryTypeHost) {
                (void)hipGetLastError();
                by_kernel = false;  // pageable (or unknown) host memory: the runtime's copy handles it
            }
        }
    }
#endif
    for (int i = 0; i < n_parts; ++i) {
        const StagePart& p = parts[i];
        if (p.bytes > kStageKernelMaxBytes || (p.bytes & 3u) ||
            ((reinterpret_cast<uintptr_t>(p.dst) | reinterpret_cast<uintptr_t>(p.src)) & 3u))
            by_kernel = false;
    }
    if (!by_kernel) {
        for (int i = 0; i < n_parts; ++i)
            if (parts[i].bytes) ORAMA_HIP_TRY(hipMemcpyAsync(parts[i].dst, parts[i].src, parts[i].bytes, kind, s));
        return ORAMA_OK;
    }
    StageArgs a{};
    uint32_t blocks = 0;
    a.n = 0;
    for (int i = 0; i < n_parts; ++i) {
        if (!parts[i].bytes) continue;
        const int j = a.n++;
        a.dst[j] = static_cast<uint32_t*>(parts[i].dst);
        a.src[j] = static_cast<const uint32_t*>(parts[i].src);
        a.bytes[j] = (uint32_t)parts[i].bytes;
        a.first_block[j] = blocks;
        blocks += (uint32_t)(parts[i].bytes >> 4) / 256u + 1u;  // (16 bytes per thread on either path; +1 carries the tail)
    }
    if (!a.n) return ORAMA_OK;
    a.first_block[a.n] = blocks;
    hipLaunchKernelGGL(stage_blocks_kernel, dim3(blocks), dim3(256), 0, s, a);
    ORAMA_HIP_TRY(hipGetLastError());
    return ORAMA_OK;
}

}  // namespace orama
