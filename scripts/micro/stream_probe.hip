// stream_probe.hip — how fast can a CU stream HBM through (a) plain 16-B/lane VGPR loads, (b) global->LDS DMA?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_probe.hip -o /tmp/stream_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void vgpr_stream(const f4* __restrict__ src, size_t n_kb, float* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    f4 acc = {0, 0, 0, 0};
    for (size_t kb = wave * DEPTH; kb + DEPTH <= n_kb; kb += nwaves * DEPTH) {
        f4 v[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const f4* p = src + (kb + i) * 64 + lane;
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) acc += v[i];
    }
    if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w;
}

template <bool NT>
__device__ __forceinline__ void dma16(uint64_t saddr, uint32_t voff, uint32_t lds_addr) {
    if (NT)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(saddr), "s"(lds_addr)
                     : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(saddr), "s"(lds_addr)
                     : "memory", "m0");
}

// each wave owns DEPTH KiB of LDS and keeps DEPTH 1-KiB DMA loads in flight; the data is read back with ds_read
template <int DEPTH, bool CONSUME, bool NT = false>
__global__ __launch_bounds__(256) void dma_stream(const char* __restrict__ src, size_t n_kb, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t wave = (size_t)blockIdx.x * 4 + w, nwaves = (size_t)gridDim.x * 4;
    const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds + (uint32_t)w * DEPTH * 1024;
    f4 acc = {0, 0, 0, 0};
    for (size_t kb = wave * DEPTH; kb + DEPTH <= n_kb; kb += nwaves * DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const uint64_t ad = (uint64_t)(uintptr_t)src + (kb + i) * 1024;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)ad), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(ad >> 32));
            dma16<NT>(((uint64_t)hi << 32) | lo, lane * 16, __builtin_amdgcn_readfirstlane((int)(lbase + i * 1024)));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (CONSUME) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) acc += *reinterpret_cast<const f4*>(lds + (size_t)(w * DEPTH + i) * 1024 + lane * 16);
        }
    }
    if (acc.x == 12345.678f) out[0] = acc.y + acc.z + acc.w;
}

template <class F>
static void run(const char* name, size_t bytes, F launch) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s %7.3f ms/pass  %7.1f GB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = 30ull << 30;
    char* src;
    float* out;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(src, 1, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t n_kb = bytes / 1024;
    for (int bpc : {1, 2, 4}) {
        const int grid = cus * bpc;
        char nm[128];
        snprintf(nm, sizeof nm, "vgpr nt depth12 x%d blocks/CU", bpc);
        run(nm, bytes, [&] { hipLaunchKernelGGL((vgpr_stream<12, true>), dim3(grid), dim3(256), 0, 0, (const f4*)src, n_kb, out); });
        snprintf(nm, sizeof nm, "vgpr    depth12 x%d blocks/CU", bpc);
        run(nm, bytes, [&] { hipLaunchKernelGGL((vgpr_stream<12, false>), dim3(grid), dim3(256), 0, 0, (const f4*)src, n_kb, out); });
        snprintf(nm, sizeof nm, "dma depth8  x%d blocks/CU", bpc);
        run(nm, bytes, [&] { hipLaunchKernelGGL((dma_stream<8, false>), dim3(grid), dim3(256), 4 * 8 * 1024, 0, src, n_kb, out); });
        snprintf(nm, sizeof nm, "dma depth8 + ds_read x%d blocks/CU", bpc);
        run(nm, bytes, [&] { hipLaunchKernelGGL((dma_stream<8, true>), dim3(grid), dim3(256), 4 * 8 * 1024, 0, src, n_kb, out); });
        snprintf(nm, sizeof nm, "dma nt depth8 + ds_read x%d blocks/CU", bpc);
        run(nm, bytes, [&] { hipLaunchKernelGGL((dma_stream<8, true, true>), dim3(grid), dim3(256), 4 * 8 * 1024, 0, src, n_kb, out); });
        snprintf(nm, sizeof nm, "dma nt depth12 + ds_read x%d blocks/CU", bpc);
        run(nm, bytes, [&] { hipLaunchKernelGGL((dma_stream<12, true, true>), dim3(grid), dim3(256), 4 * 12 * 1024, 0, src, n_kb, out); });
        if (bpc <= 2) {
            snprintf(nm, sizeof nm, "dma depth16 x%d blocks/CU", bpc);
            run(nm, bytes, [&] { hipLaunchKernelGGL((dma_stream<16, false>), dim3(grid), dim3(256), 4 * 16 * 1024, 0, src, n_kb, out); });
        }
    }
    return 0;
}
