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
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    return ((unsigned long long)uniform_u32((uint32_t)(v >> 32)) << 32) | uniform_u32((uint32_t)v);
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

// ---------------------------------------------------------------- transposed wave reduction
// ---- transposed wave reduction: QB per-lane partial sums -> one wave total per query, spread over the lanes.
// wave_sum() reduces ONE value with the tree  pairs (xor 1) -> quads (xor 2) -> octets -> rows of 16 ->
// (row0 + row1) + (row2 + row3).  Reducing QB values that way costs QB x (4 DPP adds + 4 readlanes + 3 adds); here
// every level that pairs lanes also halves the values a lane is responsible for (a reduce-scatter), so the QB
// values cost ~QB + log(QB) + 6 adds in total.  The summation TREE is the same one (float add is commutative, so it
// does not matter which lane of a pair evaluates a node): results are bit-identical to wave_sum().
// After the call lane l holds the total of query  qidx(l) = 4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1)  (QB = 8)
// resp.  qidx(l) = 2*(l&1) + ((l>>1)&1)  (QB = 4),  l&1  (QB = 2).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// value of lane (l ^ 4) inside its row of 16: row_shl:4 feeds banks 0 and 2, row_shr:4 banks 1 and 3
__device__ __forceinline__ float dpp_xor4(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);
    return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float finish_rows(float e) {
    e = e + dpp_mov<0x128>(e);                 // row_ror:8 = lane ^ 8: the two octets of a row
    e = e + __shfl_xor(e, 16, 64);             // row0 + row1 | row2 + row3
    e = e + __shfl_xor(e, 32, 64);             // (row0 + row1) + (row2 + row3)
    return e;
}
template <int QB>
__device__ __forceinline__ float wave_sum_scatter(const float (&a)[QB], int lane) {
    const bool p0 = lane & 1, p1 = lane & 2, p2 = lane & 4;
    if constexpr (QB == 8) {
        float b[4], c[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float keep = p0 ? a[4 + j] : a[j], send = p0 ? a[j] : a[4 + j];
            b[j] = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2] = lane ^ 1
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = p1 ? b[2 + j] : b[j], send = p1 ? b[j] : b[2 + j];
            c[j] = keep + dpp_mov<0x4E>(send);  // quad_perm [2,3,0,1] = lane ^ 2
        }
        const float keep = p2 ? c[1] : c[0], send = p2 ? c[0] : c[1];
        return finish_rows(keep + dpp_xor4(send));
    } else if constexpr (QB == 2) {
        const float keep = p0 ? a[1] : a[0], send = p0 ? a[0] : a[1];
        float c = keep + dpp_mov<0xB1>(send);
        c = c + dpp_mov<0x4E>(c);
        (void)p1;
        (void)p2;
        return finish_rows(c + dpp_xor4(c));
    } else {
        static_assert(QB == 4, "QB must be 2, 4 or 8");
        float b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = p0 ? a[2 + j] : a[j], send = p0 ? a[j] : a[2 + j];
            b[j] = keep + dpp_mov<0xB1>(send);
        }
        const float keep = p1 ? b[1] : b[0], send = p1 ? b[0] : b[1];
        const float c = keep + dpp_mov<0x4E>(send);
        (void)p2;
        return finish_rows(c + dpp_xor4(c));
    }
}
template <int QB>
__device__ __forceinline__ int lane_query(int lane) {
    if constexpr (QB == 8) return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
    if constexpr (QB == 2) return lane & 1;
    return 2 * (lane & 1) + ((lane >> 1) & 1);
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

// The same list with SLOTS x 64 entries (lane l holds entries l, l + 64, ...): 256 keys for the candidate stage of the
// two-stage search, which asks for max(2k, k + 128) rows.
template <int SLOTS>
struct WaveTopKN {
    unsigned long long s[SLOTS];
    unsigned long long thr = 0ull;
    uint32_t count = 0;
    uint32_t min_pos = 0;

    __device__ __forceinline__ WaveTopKN() {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) s[j] = 0ull;
    }
    __device__ __forceinline__ void recompute_min(uint32_t k, int lane) {
        unsigned long long m = ~0ull;
        uint32_t which = 0;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j)
            if ((uint32_t)lane + 64u * j < k && s[j] < m) {
                m = s[j];
                which = j;
            }
        unsigned long long w = m;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(w, off, 64);
            w = o < w ? o : w;
        }
        thr = w;
        const unsigned long long owners = __ballot(m == w);
        const int fl = __ffsll((long long)owners) - 1;
        min_pos = (uint32_t)fl + 64u * (uint32_t)__shfl((int)which, fl, 64);
    }
    __device__ __forceinline__ void set_slot(uint32_t pos, unsigned long long key, int lane) {
        if ((uint32_t)lane == (pos & 63u)) {
#pragma unroll
            for (int j = 0; j < SLOTS; ++j)
                if ((pos >> 6) == (uint32_t)j) s[j] = key;
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
