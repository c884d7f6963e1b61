// What does a DEPENDENT chain of v_mfma_f32_32x32x2_f32 cost per instruction on gfx950, with one and two waves per SIMD, with one
// and two independent accumulators per wave, and with the fillers K1m's loop carries (LDS reads, VALU address arithmetic)?
// K1m (csrc/vec_f32_mfma.hip) measured ~96 cycles per matrix instruction where the guide says 64: this probe says which part of the
// loop shape is responsible.  The shader clock is measured inside the kernel: s_memtime ticks against s_memrealtime (100 MHz).
//   build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_f32_chain_probe.hip -o scripts/micro/mfma_f32_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

// CHAINS independent accumulators per wave; LDSR ds_read_b128 and VALU v_add (64-bit) fillers per 16 matrix instructions
template <int CHAINS, int LDSR, int VALU>
__global__ __launch_bounds__(512) void chain_kernel(float* out, unsigned long long* clocks, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    f16v acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    float a = seed + lane * 1e-3f, b = seed - lane * 1e-3f;
    unsigned long long addr = (unsigned long long)out;
    f4 l = f4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < LDSR; ++f) {
            const f4 v = *reinterpret_cast<const f4*>(lds + ((lane * 16 + f * 1024 + it * 64) & 32767 & ~15));
            l += v;
        }
#pragma unroll
        for (int f = 0; f < VALU; ++f) addr += (unsigned long long)(it + f) << 7;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = l[0] + l[1] + l[2] + l[3] + (float)(addr & 1);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}

template <int CHAINS, int LDSR, int VALU>
int run(const char* what, int threads, float* out, unsigned long long* clocks, int cus) {
    const int iters = 4096 / CHAINS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((chain_kernel<CHAINS, LDSR, VALU>), dim3(cus), dim3(threads), 0, 0, out, clocks, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost));
        }
    }
    const double waves_per_simd = threads / 64 / 4.0;
    const double mfma_per_simd = waves_per_simd * iters * 16.0 * CHAINS;
    const double mhz_memtime = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;  // s_memtime ticks per 100 MHz tick
    printf("%-86s %7.3f ms   s_memtime/s_memrealtime %8.1f MHz-equivalent   %6.1f ns per MFMA per SIMD = %6.1f cycles at 2.0 GHz, %6.1f at the s_memtime rate\n",
           what, best, mhz_memtime, best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.0, best * 1e6 / mfma_per_simd * mhz_memtime / 1e3);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out;
    unsigned long long* clocks;
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMalloc(&clocks, 64));
    printf("%d CUs, clockRate %d kHz\n", cus, prop.clockRate);
    if (run<1, 0, 0>("1 wave/SIMD, one dependent chain", 256, out, clocks, cus)) return 1;
    if (run<1, 0, 0>("2 waves/SIMD, one dependent chain each", 512, out, clocks, cus)) return 1;
    if (run<2, 0, 0>("1 wave/SIMD, two independent chains", 256, out, clocks, cus)) return 1;
    if (run<2, 0, 0>("2 waves/SIMD, two independent chains each", 512, out, clocks, cus)) return 1;
    if (run<1, 8, 0>("2 waves/SIMD, one chain + 8 ds_read_b128 per 16 MFMAs", 512, out, clocks, cus)) return 1;
    if (run<1, 0, 32>("2 waves/SIMD, one chain + 32 64-bit VALU adds per 16 MFMAs", 512, out, clocks, cus)) return 1;
    if (run<1, 8, 32>("2 waves/SIMD, one chain + 8 ds_read_b128 + 32 VALU per 16 MFMAs", 512, out, clocks, cus)) return 1;
    if (run<2, 8, 32>("2 waves/SIMD, two chains + 8 ds_read_b128 + 32 VALU per 32 MFMAs", 512, out, clocks, cus)) return 1;
    return 0;
}
