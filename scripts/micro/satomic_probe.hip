// Scalar memory atomics on gfx950: does s_atomic_add (GLC: returns the old value, counted by lgkmcnt — not by vmcnt,
// so waiting for it leaves a wave's vector loads in flight) exist here, and is it coherent across the 8 XCDs?
// Every wave adds 1 to each of 64 counters `rounds` times and records what came back; the host checks that the values
// returned for a counter are a permutation of 0..total-1.  Also times it against the vector atomic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_scalar(uint32_t* counters, uint32_t* out, int rounds) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (int r = 0; r < rounds; ++r) {
        for (int j = 0; j < 64; j += 8) {
            uint32_t v[8];
            uint32_t* p = counters + j;
            // 8 scalar atomics in flight, one wait
            asm volatile(
                "s_mov_b32 s20, 1\n\ts_mov_b32 s21, 1\n\ts_mov_b32 s22, 1\n\ts_mov_b32 s23, 1\n\t"
                "s_mov_b32 s24, 1\n\ts_mov_b32 s25, 1\n\ts_mov_b32 s26, 1\n\ts_mov_b32 s27, 1\n\t"
                "s_atomic_add s20, %8, 0x0 glc\n\t"
                "s_atomic_add s21, %8, 0x4 glc\n\t"
                "s_atomic_add s22, %8, 0x8 glc\n\t"
                "s_atomic_add s23, %8, 0xc glc\n\t"
                "s_atomic_add s24, %8, 0x10 glc\n\t"
                "s_atomic_add s25, %8, 0x14 glc\n\t"
                "s_atomic_add s26, %8, 0x18 glc\n\t"
                "s_atomic_add s27, %8, 0x1c glc\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_mov_b32 %0, s20\n\ts_mov_b32 %1, s21\n\ts_mov_b32 %2, s22\n\ts_mov_b32 %3, s23\n\t"
                "s_mov_b32 %4, s24\n\ts_mov_b32 %5, s25\n\ts_mov_b32 %6, s26\n\ts_mov_b32 %7, s27"
                : "=s"(v[0]), "=s"(v[1]), "=s"(v[2]), "=s"(v[3]), "=s"(v[4]), "=s"(v[5]), "=s"(v[6]), "=s"(v[7])
                : "s"(p)
                : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory");
            if (lane == 0 && out)
                for (int t = 0; t < 8; ++t) out[((size_t)wave * rounds + r) * 64 + j + t] = v[t];
        }
    }
}

__global__ void k_vector(uint32_t* counters, uint32_t* out, int rounds, int stride) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t v = atomicAdd(&counters[(size_t)lane * stride], 1u);
        if (out) out[((size_t)wave * rounds + r) * 64 + lane] = v;
    }
}

int main() {
    const int blocks = 256, threads = 512, rounds = 8;
    const int waves = blocks * threads / 64;
    uint32_t *cnt, *out;
    CK(hipMalloc(&cnt, 64 * 4 * 4096));
    CK(hipMalloc(&out, (size_t)waves * rounds * 64 * 4));
    std::vector<uint32_t> h((size_t)waves * rounds * 64);
    const int strides[] = {1, 1, 32, 64, 1024, 4096};
    for (int mode = 0; mode < 6; ++mode) {
        const int stride = strides[mode];
        CK(hipMemset(cnt, 0, 64 * 4 * 4096));
        CK(hipMemset(out, 0xff, h.size() * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(threads), 0, 0, cnt, out, rounds);
        else hipLaunchKernelGGL(k_vector, dim3(blocks), dim3(threads), 0, 0, cnt, out, rounds, stride);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> call(64 * 4096); CK(hipMemcpy(call.data(), cnt, call.size() * 4, hipMemcpyDeviceToHost));
        uint32_t c[64]; for (int j = 0; j < 64; ++j) c[j] = call[(size_t)j * stride];
        const uint32_t total = (uint32_t)waves * rounds;
        int bad = 0;
        for (int j = 0; j < 64; ++j) {
            std::vector<uint32_t> col(total);
            for (uint32_t i = 0; i < total; ++i) col[i] = h[(size_t)i * 64 + j];
            std::sort(col.begin(), col.end());
            for (uint32_t i = 0; i < total; ++i) if (col[i] != i) { ++bad; break; }
            if (c[j] != total) ++bad;
        }
        printf("%s, counters %d B apart: %d waves x %d rounds x 64 counters: %.3f ms (%.1f ns per atomic per wave-round), %s\n",
               mode == 0 ? "scalar s_atomic_add" : "vector global_atomic_add", stride * 4, waves, rounds, ms,
               ms * 1e6 / rounds / 64, bad ? "MISMATCH" : "returned values are a permutation: coherent");
    }
    return 0;
}
