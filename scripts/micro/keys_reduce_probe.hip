// Where does a workgroup of the key-list kernels (select.hip: keys_reduce_kernel, keys_final_kernel) spend its time?
// This file includes select.hip with ORAMA_KEYS_STAMP defined: thread 0 of every workgroup writes the 100 MHz wall clock at
// the phase boundaries.  Lists shaped like a C4 BM25 query's (590 000 keys, scores of a few clusters), 1 and 32 lists.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ioramacore_amd/csrc -Iinclude scripts/micro/keys_reduce_probe.hip \
//          -Loramacore_amd/csrc -lorama_hip -Wl,-rpath,$PWD/oramacore_amd/csrc -o scripts/micro/keys_reduce_probe
#include <hip/hip_runtime.h>
#include <cstdint>
__device__ unsigned long long g_stamps[8192 * 8];
#define ORAMA_KEYS_STAMP(i)                                                                                          \
    do {                                                                                                             \
        __builtin_amdgcn_s_waitcnt(0);                                                                               \
        if (threadIdx.x == 0) g_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64();           \
    } while (0)
#define ORAMA_KEYS_NOTE(i, v) (g_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = (v))
#include "select.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static uint32_t f32_ordered(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

int main() {
    using namespace orama;
    const uint32_t n = 590000, k = 100;
    const uint32_t stride = (n + 1) & ~1u;
    for (uint32_t q : {1u, 32u}) {
        std::mt19937_64 rng(7);
        std::vector<unsigned long long> h((size_t)q * stride, 0ull);
        for (uint32_t l = 0; l < q; ++l)
            for (uint32_t i = 0; i < n; ++i) {
                const int tok = (int)(rng() % 12);
                const float idf = 4.0f + 0.45f * tok;
                const float ntf = 0.3f + 2.5f * (float)((rng() >> 11) * (1.0 / 9007199254740992.0));
                const float score = idf * 2.2f * ntf / (1.2f + ntf);
                if (rng() % 50 == 0) continue;  // empty slot
                h[(size_t)l * stride + i] = ((unsigned long long)f32_ordered(score) << 32) | (uint32_t)~i;
            }
        unsigned long long *d_keys, *d_tmp, *d_tau;
        const uint32_t chunks = (n + 8191) / 8192;
        CK(hipMalloc(&d_keys, h.size() * 8));
        CK(hipMalloc(&d_tmp, (size_t)q * chunks * k * 8));
        CK(hipMalloc(&d_tau, q * 8));
        CK(hipMemcpy(d_keys, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        float* d_val;
        uint64_t* d_ids;
        uint32_t* d_n;
        CK(hipMalloc(&d_val, q * k * 4));
        CK(hipMalloc(&d_ids, q * k * 8));
        CK(hipMalloc(&d_n, q * 4));
        hipEvent_t e0, e1, e2;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        std::vector<unsigned long long> st(8192 * 8);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d_tau, 0, q * 8));
            CK(hipDeviceSynchronize());
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(keys_reduce_kernel, dim3(q, chunks), dim3(kSortThreads), 0, 0, d_keys, n, (uint64_t)stride, nullptr, k, d_tmp,
                               (uint64_t)chunks * k, d_tau, 1u);
            (void)hipEventRecord(e1, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                // phases per workgroup (10 ns ticks): first by class
                double sum[2][6] = {{0}};
                uint32_t cnt[2] = {0, 0};
                unsigned long long t_first = ~0ull, t_last = 0;
                for (uint32_t w = 0; w < q * chunks; ++w) {
                    const unsigned long long* s = &st[w * 8];
                    const bool ranked = s[4] != 0 && (s[5] == 0 || s[4] > s[5]);
                    const unsigned long long end = ranked ? s[4] : s[5];
                    const int c = ranked ? 0 : 1;
                    ++cnt[c];
                    sum[c][0] += (double)(s[1] - s[0]);
                    sum[c][1] += (double)(s[2] - s[1]);
                    sum[c][2] += (double)(s[3] - s[2]);
                    sum[c][3] += (double)(end - s[3]);
                    sum[c][4] += (double)(end - s[0]);
                    t_first = s[0] < t_first ? s[0] : t_first;
                    t_last = end > t_last ? end : t_last;
                }
                printf("keys_reduce  q=%u: launch %.1f us by events, first stamp to last stamp %.1f us, %u workgroups\n", q, ms * 1e3,
                       (double)(t_last - t_first) * 0.01, q * chunks);
                for (int c = 0; c < 2; ++c)
                    if (cnt[c])
                        printf("   %-24s %5u workgroups: loads %.2f us, bound/barrier %.2f, compaction %.2f, %s %.2f, whole %.2f us\n",
                               c == 0 ? "ranks by counting" : "radix rounds / take all", cnt[c], sum[c][0] / cnt[c] * 0.01, sum[c][1] / cnt[c] * 0.01,
                               sum[c][2] / cnt[c] * 0.01, c == 0 ? "ranks" : "select+write", sum[c][3] / cnt[c] * 0.01, sum[c][4] / cnt[c] * 0.01);
            }
            std::vector<unsigned long long> zero(8192 * 8, 0ull);
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero.data(), zero.size() * 8));
            (void)hipEventRecord(e1, 0);
            hipLaunchKernelGGL(keys_final_kernel, dim3(q), dim3(kSortThreads), 0, 0, d_tmp, chunks * k, (uint64_t)chunks * k, nullptr, k, true,
                               nullptr, nullptr, d_ids, d_val, d_n);
            (void)hipEventRecord(e2, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
            (void)hipEventElapsedTime(&ms, e1, e2);
            if (rep == 2) {
                double sum[6] = {0};
                for (uint32_t w = 0; w < q; ++w) {
                    const unsigned long long* s = &st[w * 8];
                    for (int i = 0; i < 6; ++i) sum[i] += (double)(s[i + 1] - s[i]);
                }
                printf("keys_final   q=%u: launch %.1f us by events; loads %.2f us, bound %.2f, compaction %.2f, ranks+records %.2f, order check %.2f, write %.2f\n",
                       q, ms * 1e3, sum[0] / q * 0.01, sum[1] / q * 0.01, sum[2] / q * 0.01, sum[3] / q * 0.01, sum[4] / q * 0.01, sum[5] / q * 0.01);
            }
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero.data(), zero.size() * 8));
        }
        // sanity: the best key of list 0 on the host
        std::vector<float> val(k);
        CK(hipMemcpy(val.data(), d_val, k * 4, hipMemcpyDeviceToHost));
        unsigned long long best = 0;
        for (uint32_t i = 0; i < n; ++i) best = h[i] > best ? h[i] : best;
        const uint32_t o = (uint32_t)(best >> 32);
        const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
        float f;
        memcpy(&f, &u, 4);
        printf("   best score of list 0: device %.7g host %.7g\n", val[0], f);
        for (void* b : {(void*)d_keys, (void*)d_tmp, (void*)d_tau, (void*)d_val, (void*)d_ids, (void*)d_n}) (void)hipFree(b);
    }

    // ---- (value, index) candidate lists as the fp16 filter scans leave them: 100 seeds (the best so far, sorted) + ~700 rows
    // that beat the worst seed; 256 lists, 8 parts, finished by the reducing workgroup
    {
        const uint32_t q = 256, n = 800, stride = 3000000, k = 100;
        std::mt19937_64 rng(11);
        std::vector<float> hv((size_t)q * n);
        std::vector<uint32_t> hi((size_t)q * n), hn(q, n);
        for (uint32_t l = 0; l < q; ++l) {
            std::vector<float> seeds(k);
            for (auto& x : seeds) x = 0.80f + 0.05f * (float)((rng() >> 11) * (1.0 / 9007199254740992.0));
            std::sort(seeds.begin(), seeds.end());
            for (uint32_t i = 0; i < n; ++i) {
                hv[(size_t)l * n + i] = i < k ? seeds[i] : 0.78f + 0.07f * (float)((rng() >> 11) * (1.0 / 9007199254740992.0));
                hi[(size_t)l * n + i] = (uint32_t)(rng() % 10000000u);
            }
        }
        float* d_v;
        uint32_t *d_i, *d_n, *d_on, *d_state;
        unsigned long long* d_keys;
        uint64_t* d_ids;
        float* d_val;
        CK(hipMalloc(&d_v, (size_t)q * stride * 4));
        CK(hipMalloc(&d_i, (size_t)q * stride * 4));
        for (uint32_t l = 0; l < q; ++l) {
            CK(hipMemcpy(d_v + (size_t)l * stride, &hv[(size_t)l * n], n * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_i + (size_t)l * stride, &hi[(size_t)l * n], n * 4, hipMemcpyHostToDevice));
        }
        CK(hipMalloc(&d_n, q * 4));
        CK(hipMemcpy(d_n, hn.data(), q * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_keys, (size_t)q * 4096 * 8));
        CK(hipMalloc(&d_ids, q * k * 8));
        CK(hipMalloc(&d_val, q * k * 4));
        CK(hipMalloc(&d_on, q * 4));
        CK(hipMalloc(&d_state, q * 4));
        PairsFinal fin;
        fin.out_ids = d_ids;
        fin.out_val = d_val;
        fin.out_n = d_on;
        fin.done = d_state;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<unsigned long long> st(8192 * 8), zero(8192 * 8, 0ull);
        for (uint32_t parts : {8u, 2u})
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero.data(), zero.size() * 8));
            CK(hipDeviceSynchronize());
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(pairs_reduce_kernel<true>, dim3(q, parts), dim3(kSortThreads), 0, 0, d_v, d_i, (uint64_t)stride, d_n, stride, false, k,
                               d_keys, fin);
            (void)hipEventRecord(e1, 0);
            CK(hipDeviceSynchronize());
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
            if (rep == 2) {
                double sum[5] = {0};
                uint32_t ordered = 0;
                unsigned long long t_first = ~0ull, t_last = 0;
                for (uint32_t l = 0; l < q; ++l) {
                    const unsigned long long* s = &st[(size_t)l * 8];  // (grid (lists, parts): workgroup (l, 0))
                    for (int i = 0; i < 5; ++i) sum[i] += (double)(s[i + 1] - s[i]);
                    ordered += (uint32_t)(s[6] >> 32);
                    t_first = s[0] < t_first ? s[0] : t_first;
                    t_last = s[5] > t_last ? s[5] : t_last;
                }
                printf("pairs_reduce q=%u lists of %u (%u workgroups per list, one of them working; fused final): launch %.1f us by events, first to last stamp of the working groups %.1f us\n"
                       "   per working workgroup: loads %.2f us, bound %.2f, compaction %.2f, cut %.2f, final order %.2f; %u of %u lists left the cut in key order\n",
                       q, n, parts, ms * 1e3, (double)(t_last - t_first) * 0.01, sum[0] / q * 0.01, sum[1] / q * 0.01, sum[2] / q * 0.01, sum[3] / q * 0.01,
                       sum[4] / q * 0.01, ordered, q);
            }
        }
    }
    return 0;
}
