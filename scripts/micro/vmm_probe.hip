// vmm_probe.hip — can the vector store grow IN PLACE?  Reserve a large virtual range once, map physical chunks
// behind it on demand (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess): no realloc + copy, the
// base address never changes, running scans keep reading the rows they started with.
// Build: hipcc --offload-arch=gfx950 -O2 -o vmm_probe vmm_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("FAIL %s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);     \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void touch(uint32_t* p, uint64_t n, uint32_t v) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v + (uint32_t)i;
}
__global__ void check(const uint32_t* p, uint64_t n, uint32_t v, unsigned long long* bad) {
    unsigned long long b = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b += p[i] != v + (uint32_t)i;
    if (b) atomicAdd(bad, b);
}

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    int vmm = 0;
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("hipDeviceAttributeVirtualMemoryManagementSupported = %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t gran_min = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity recommended %zu, minimum %zu\n", gran, gran_min);
    const size_t va_bytes = 256ull << 30;  // 256 GiB of address space
    void* base = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    CK(hipMemAddressReserve(&base, va_bytes, gran, nullptr, 0));
    printf("reserved %zu GiB at %p\n", va_bytes >> 30, base);
    const size_t chunk = 1ull << 30;
    hipMemGenericAllocationHandle_t h[8];
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long* d_bad;
    CK(hipMalloc(&d_bad, 8));
    CK(hipMemset(d_bad, 0, 8));
    for (int c = 0; c < 4; ++c) {
        auto a = std::chrono::steady_clock::now();
        CK(hipMemCreate(&h[c], chunk, &prop, 0));
        CK(hipMemMap((char*)base + c * chunk, chunk, 0, h[c], 0));
        CK(hipMemSetAccess((char*)base + c * chunk, chunk, &acc, 1));
        auto b = std::chrono::steady_clock::now();
        printf("chunk %d mapped in %.3f ms\n", c, std::chrono::duration<double, std::milli>(b - a).count());
        // write the NEW chunk while re-checking the OLD ones through the same base pointer
        touch<<<1024, 256>>>((uint32_t*)((char*)base + c * chunk), chunk / 4, 1000u * c);
        for (int o = 0; o < c; ++o) check<<<1024, 256>>>((const uint32_t*)((char*)base + o * chunk), chunk / 4, 1000u * o, d_bad);
        CK(hipDeviceSynchronize());
    }
    // one kernel over the whole contiguous 4 GiB
    unsigned long long bad = 0;
    for (int o = 0; o < 4; ++o) check<<<1024, 256>>>((const uint32_t*)((char*)base + o * chunk), chunk / 4, 1000u * o, d_bad);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("mismatches after growth: %llu\n", bad);
    // streaming bandwidth through a VMM mapping vs hipMalloc (same kernel)
    uint32_t* plain;
    CK(hipMalloc(&plain, 4 * chunk));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        uint32_t* p = which ? plain : (uint32_t*)base;
        touch<<<2048, 256>>>(p, 4 * chunk / 4, 7);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) check<<<2048, 256>>>(p, 4 * chunk / 4, 7, d_bad);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s read: %.1f GB/s\n", which ? "hipMalloc" : "VMM-mapped", 5.0 * 4 * chunk / (ms * 1e-3) / 1e9);
    }
    for (int c = 0; c < 4; ++c) {
        CK(hipMemUnmap((char*)base + c * chunk, chunk));
        CK(hipMemRelease(h[c]));
    }
    CK(hipMemAddressFree(base, va_bytes));
    auto t1 = std::chrono::steady_clock::now();
    printf("OK total %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    return 0;
}
