// device_utils.hpp — gfx950 device helpers shared by the kernels: wave64 reductions on DPP,
// order-preserving float keys, uniform-lane helpers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace orama {

constexpr int kWave = 64;  // CDNA wavefront width (hard-coded: warpSize folds to 64 on gfx950)

// DPP controls (gfx9 encoding): quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E,
// row_half_mirror=0x141, row_mirror=0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes of a wave.  4 DPP steps reduce each 16-lane row (every lane of a row
// ends up holding the row total), then the 4 row totals are combined through readlane, so the
// result is wave-uniform (lives in SGPRs) and the summation order is fixed.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Order-preserving map f32 -> u32 (larger float <=> larger key). -0.0 is canonicalised to +0.0 so
// that keys compare equal exactly when the floats do. NaN must be filtered by the caller.
__device__ __forceinline__ uint32_t f32_to_ordered(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}

// ---------------------------------------------------------------- per-wave running top-k (k <= 128)
// 64-bit keys, "larger is better", 0 = empty slot.  The list lives in REGISTERS: lane l holds entries l and
// l + 64.  `thr` (wave-uniform) is the smallest kept key once the list is full, so the hot-path test is one
// scalar compare per candidate; an insert replaces the minimum and re-reduces (rare: ~k·ln(n/k) per wave).
struct WaveTopK {
    unsigned long long s0 = 0ull, s1 = 0ull;
    unsigned long long thr = 0ull;
    uint32_t count = 0;
    uint32_t min_pos = 0;

    __device__ __forceinline__ void recompute_min(uint32_t k, int lane) {
        unsigned long long m = ~0ull;
        if ((uint32_t)lane < k) m = s0;
        if ((uint32_t)lane + 64u < k && s1 < m) m = s1;
        unsigned long long w = m;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(w, off, 64);
            w = o < w ? o : w;
        }
        thr = w;
        const unsigned long long owners = __ballot(m == w);
        const int fl = __ffsll((long long)owners) - 1;
        const uint32_t which = ((uint32_t)lane < k && s0 == w) ? 0u : 1u;
        min_pos = (uint32_t)fl + 64u * (uint32_t)__shfl((int)which, fl, 64);
    }
    __device__ __forceinline__ void set_slot(uint32_t pos, unsigned long long key, int lane) {
        if ((uint32_t)lane == (pos & 63u)) {
            if (pos < 64u) s0 = key; else s1 = key;
        }
    }
    // key must be wave-uniform; call only when (count < k || key > thr)
    __device__ __forceinline__ void insert(unsigned long long key, uint32_t k, int lane) {
        if (count < k) {
            set_slot(count, key, lane);
            ++count;
            if (count == k) recompute_min(k, lane);
        } else {
            set_slot(min_pos, key, lane);
            recompute_min(k, lane);
        }
    }
};

}  // namespace orama
