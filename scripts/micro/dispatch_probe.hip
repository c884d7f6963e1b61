// How long does the chip take to merely DISPATCH a grid?  Empty workgroups (one global load of a kernel argument's target, then
// exit) in the shapes of this library's launches: workgroups x threads x LDS bytes.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/micro/dispatch_probe.hip -o scripts/micro/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(const unsigned* p, unsigned* out) {
    extern __shared__ unsigned lds[];
    if (p[0] == 0xdeadbeefu) {  // never: keeps the load and the LDS allocation alive
        lds[threadIdx.x] = 1;
        out[blockIdx.x] = lds[0];
    }
}

int main() {
    unsigned *p, *out;
    CK(hipMalloc(&p, 4));
    CK(hipMemset(p, 0, 4));
    CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Shape { unsigned wgs, threads, lds; const char* what; };
    const Shape shapes[] = {
        {14752, 256, 20000, "range_score_kernel: 32 queries x 461 ranges"},
        {14752, 256, 0, "  the same without LDS"},
        {7376, 512, 40000, "  half the workgroups, twice the threads"},
        {2336, 1024, 66000, "keys_reduce_kernel: 32 lists x 73 chunks"},
        {2048, 1024, 66000, "pairs_reduce_kernel: 256 lists x 8 parts"},
        {512, 1024, 66000, "pairs_reduce_kernel: 256 lists x 2 parts"},
        {1024, 1024, 66000, "dense heads: 256 lists x 4 parts"},
        {131072 / 4, 256, 0, "32 768 workgroups of 256 threads"},
        {32768, 64, 0, "32 768 workgroups of one wave"},
    };
    for (const Shape& s : shapes) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(empty_kernel, dim3(s.wgs), dim3(s.threads), s.lds, 0, p, out);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%6u workgroups x %4u threads, %6u B LDS: %7.1f us  = %5.1f ns per workgroup, %4.2f ns per wave   (%s)\n", s.wgs, s.threads, s.lds,
               best * 1e3, best * 1e6 / s.wgs, best * 1e6 / (s.wgs * (s.threads / 64.0)), s.what);
    }
    return 0;
}
