// stage.hpp — small host<->device blocks moved by a KERNEL of the caller's stream instead of a copy command (round 5).
#pragma once
#include "common.hpp"

namespace orama {

// Blocks above this travel by hipMemcpyAsync (the SDMA engines move large blocks faster than a few workgroups over PCIe).
constexpr size_t kStageKernelMaxBytes = 256u << 10;

struct StagePart {
    void* dst = nullptr;
    const void* src = nullptr;
    size_t bytes = 0;  // a multiple of 4; dst and src 4-byte aligned (16-byte aligned pairs take the wide path)
};

// One launch for up to 4 blocks; pinned host memory (hipHostMalloc) on the host side.  Falls back to one hipMemcpyAsync per
// block when the context asks for copy commands (ORAMA_STAGE_COPY=dma), a block is larger than kStageKernelMaxBytes or is not
// a whole number of words.
int stage_blocks(orama_ctx* ctx, const StagePart* parts, int n_parts, hipMemcpyKind kind, hipStream_t s);

inline int stage_block(orama_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    StagePart p{dst, src, bytes};
    return stage_blocks(ctx, &p, 1, kind, s);
}

}  // namespace orama
