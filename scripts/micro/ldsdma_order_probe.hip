// ldsdma_order_probe.hip — does `s_waitcnt vmcnt(N)` cover global->LDS DMA loads IN ISSUE ORDER on gfx950?
//
// K2c's ring would like counted waits (leave the newest stage in flight).  Round 1 saw wrong fragments with counted
// waits and fell back to vmcnt(0)-only scheduling.  This probe isolates the question: every wave issues 4 DMAs
// from an HBM-cold region (never cached: 4 GiB swept once) followed by 4 DMAs from an L2-hot 64 KiB region, waits
// vmcnt(4) and checks that the FIRST four (the slow ones) have landed.  If completion were counted out of order the
// four fast loads would satisfy the wait and the check would read the sentinel.
//   variant 0: all loads default policy      variant 1: cold loads `nt`, hot loads default
//   variant 2: all eight cold (control)      variant 3: hot first, then cold; wait vmcnt(4) checks the hot ones
// Build: hipcc --offload-arch=gfx950 -O2 -o ldsdma_order_probe ldsdma_order_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);         \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ void dma16(uint64_t saddr, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(saddr), "{m0}"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16_nt(uint64_t saddr, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(saddr), "{m0}"(lds_addr) : "memory");
}

// piece p (16 bytes) of a buffer holds {tag ^ p, ...}
__global__ void fill(uint32_t* buf, uint64_t pieces, uint32_t tag) {
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < pieces; p += (uint64_t)gridDim.x * blockDim.x) {
        buf[p * 4 + 0] = tag ^ (uint32_t)p;
        buf[p * 4 + 1] = 0x11111111u;
        buf[p * 4 + 2] = 0x22222222u;
        buf[p * 4 + 3] = 0x33333333u;
    }
}

constexpr uint32_t kColdTag = 0xC01D0000u, kHotTag = 0x40700000u;

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(const char* cold, uint64_t cold_chunks, const char* hot, uint32_t hot_chunks,
                                             uint32_t iters, unsigned long long* violations,
                                             unsigned long long* late_violations, unsigned long long* checked) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds + (uint32_t)w * 8192;
    uint32_t* my = reinterpret_cast<uint32_t*>(lds + w * 8192);
    const uint32_t vlane = lane * 16;
    unsigned long long bad = 0, bad_late = 0, n = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        for (int s = 0; s < 8; ++s) my[s * 256 + lane * 4] = 0xFFFFFFFFu;  // sentinel in the checked dword
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // lgkmcnt(0)
        uint64_t chunk[8];
        bool is_cold[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool c = VARIANT == 2 ? true : (VARIANT == 3 ? s >= 4 : s < 4);
            is_cold[s] = c;
            if (c) {
                chunk[s] = ((((uint64_t)it * gridDim.x + blockIdx.x) * 8 + w) * 8 + s) % cold_chunks;
                if (VARIANT == 1) dma16_nt((uint64_t)(uintptr_t)cold + chunk[s] * 1024, vlane, lds_base + s * 1024);
                else dma16((uint64_t)(uintptr_t)cold + chunk[s] * 1024, vlane, lds_base + s * 1024);
            } else {
                chunk[s] = (uint64_t)((it * 8 + s + w * 3 + blockIdx.x) % hot_chunks);
                dma16((uint64_t)(uintptr_t)hot + chunk[s] * 1024, vlane, lds_base + s * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // vmcnt(4): the four oldest loads must have landed
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t got = my[s * 256 + lane * 4];
            const uint32_t exp = (is_cold[s] ? kColdTag : kHotTag) ^ (uint32_t)(chunk[s] * 64 + lane);
            bad += got != exp;
            ++n;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // vmcnt(0)
#pragma unroll
        for (int s = 4; s < 8; ++s) {
            const uint32_t got = my[s * 256 + lane * 4];
            const uint32_t exp = (is_cold[s] ? kColdTag : kHotTag) ^ (uint32_t)(chunk[s] * 64 + lane);
            bad_late += got != exp;
        }
    }
    atomicAdd(violations, bad);
    atomicAdd(late_violations, bad_late);
    atomicAdd(checked, n);
}

int main() {
    const uint64_t cold_bytes = 8ull << 30;
    const uint32_t hot_bytes = 64u << 10;
    char *cold, *hot;
    unsigned long long* d_cnt;
    CK(hipMalloc(&cold, cold_bytes));
    CK(hipMalloc(&hot, hot_bytes));
    CK(hipMalloc(&d_cnt, 24));
    fill<<<4096, 256>>>((uint32_t*)cold, cold_bytes / 16, kColdTag);
    fill<<<64, 256>>>((uint32_t*)hot, hot_bytes / 16, kHotTag);
    CK(hipDeviceSynchronize());
    const uint32_t iters = 512;  // 256 blocks x 8 waves x 512 iters x 4 KiB cold = 4 GiB of cold traffic per run
    auto run = [&](int variant) {
        CK(hipMemset(d_cnt, 0, 24));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        switch (variant) {
            case 0: probe<0><<<256, 512, 65536>>>(cold, cold_bytes / 1024, hot, hot_bytes / 1024, iters, d_cnt, d_cnt + 1, d_cnt + 2); break;
            case 1: probe<1><<<256, 512, 65536>>>(cold, cold_bytes / 1024, hot, hot_bytes / 1024, iters, d_cnt, d_cnt + 1, d_cnt + 2); break;
            case 2: probe<2><<<256, 512, 65536>>>(cold, cold_bytes / 1024, hot, hot_bytes / 1024, iters, d_cnt, d_cnt + 1, d_cnt + 2); break;
            default: probe<3><<<256, 512, 65536>>>(cold, cold_bytes / 1024, hot, hot_bytes / 1024, iters, d_cnt, d_cnt + 1, d_cnt + 2); break;
        }
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[3];
        CK(hipMemcpy(h, d_cnt, 24, hipMemcpyDeviceToHost));
        const char* names[] = {"cold x4 then hot x4, default policy", "cold(nt) x4 then hot x4", "cold x8 (control)",
                               "hot x4 then cold x4"};
        printf("variant %d (%s): checked %llu lane-words after vmcnt(4): %llu violations (%.4f %%); after vmcnt(0): %llu; %.2f ms\n",
               variant, names[variant], h[2], h[0], h[2] ? 100.0 * h[0] / h[2] : 0.0, h[1], ms);
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 4; ++v) run(v);
    return 0;
}
