// vec_f16_async.hpp — device helpers shared by the register-resident-query scans (vec_f16_qs.hip, vec_f16_kh.hip):
// global -> LDS DMA statements, counted waits, the raw stage barrier, small wave-level utilities.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_utils.hpp"

namespace orama {
namespace f16async {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane, global -> LDS, asynchronous (counted by vmcnt); lane l's data lands at LDS address m0 + 16 l.
// m0 is an INPUT operand of the statement: the compiler materialises it and knows it is live; the leading s_nop is the
// wait state gfx9 wants between a write of m0 and an LDS-DMA instruction reading it.
__device__ __forceinline__ void qs_dma16_nt(uint64_t saddr_uniform, uint32_t voff, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                 :
                 : "v"(voff), "s"(saddr_uniform), "{m0}"(lds_addr_uniform)
                 : "memory");
}
// 4 bytes per lane (lane l's dword lands at m0 + 4 l), from saddr + the lane's 32-bit byte offset (no 64-bit per-lane
// pointer: such a pointer is loop-invariant, gets hoisted and then spilled next to the 192 fragment registers)
__device__ __forceinline__ void qs_dma4(uint64_t saddr_uniform, uint32_t voff, uint32_t lds_addr_uniform) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :
                 : "v"(voff), "s"(saddr_uniform), "{m0}"(lds_addr_uniform)
                 : "memory");
}
// the lane id, computed HERE (a volatile statement is neither hoisted nor merged with an earlier copy): for the rare paths of a
// kernel that cannot afford a register for a value kept since its first line
__device__ __forceinline__ uint32_t qs_lane_id_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// a wave-uniform 64-bit value that the compiler cannot know to be uniform (read from LDS): into scalar registers
__device__ __forceinline__ uint64_t qs_uniform_u64(uint64_t v) {
    return (uint64_t)uniform_u32((uint32_t)v) | ((uint64_t)uniform_u32((uint32_t)(v >> 32)) << 32);
}
// OR over the 64 lanes of a wave (wave-uniform result): the DPP steps of wave_sum (device_utils.hpp) within each row of 16
// lanes, then the four row values through readlane
template <int CTRL>
__device__ __forceinline__ uint32_t qs_dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
    v |= qs_dpp_u32<0xB1>(v);
    v |= qs_dpp_u32<0x4E>(v);
    v |= qs_dpp_u32<0x141>(v);
    v |= qs_dpp_u32<0x140>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
template <int N>
__device__ __forceinline__ void qs_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The stage barrier.  A raw s_barrier, not __syncthreads(): the fence of the latter drains lgkmcnt (the fragment reads
// that run ahead across the barrier) and may drain vmcnt (the prefetch ring).  Nothing needs to be waited for here: the
// reads of the buffer that is re-filled after this barrier fed MFMAs the wave has already issued, so they have returned.
__device__ __forceinline__ void qs_stage_barrier() { asm volatile("s_barrier" ::: "memory"); }

// The arguments only a flush reads.  Kept in LDS: as kernel arguments they would sit in 14 SGPRs for the whole launch, the
// kernel runs out of SGPRs, and SGPR spills take vector registers the query fragments need (a fragment then lives in
// scratch memory and every reload of it waits for the whole prefetch ring: vmcnt is in order).
struct QsFlushArgs {
    float* cand_dist;
    uint32_t* cand_row;
    uint32_t* cand_count;
    uint64_t cand_stride;
    const uint64_t* row_doc;
    const uint64_t* allow;
    uint64_t allow_bits;
    uint32_t no_appends;
};
static_assert(sizeof(QsFlushArgs) <= 64, "FlushArgs slot");

}  // namespace f16async
}  // namespace orama
