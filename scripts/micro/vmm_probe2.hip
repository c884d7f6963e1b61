// vmm_probe2.hip — where does hipMemSetAccess start to fail?  (vec_store's in-place growth hit "invalid argument" on a
// 30 GB store.)  Variants: piece size, reservation alignment, one SetAccess per piece vs one over the whole mapped range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CKR(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("   FAIL %s: %s\n", #x, hipGetErrorString(e)); return false; } } while (0)

static bool run_mixed(size_t va, int n_big, size_t small, int n_small, int mode) {  // mode 0 per piece, 1 whole range each time, 2 whole range once at the end, 3 only the new tail range in one call
    const size_t G = 1ull << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    void* base = nullptr;
    CKR(hipMemAddressReserve(&base, va, 2u << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    size_t mapped = 0; bool ok = true;
    for (int i = 0; i < n_big + n_small && ok; ++i) {
        const size_t piece = i < n_big ? G : small;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, piece, &prop, 0);
        if (e != hipSuccess) { printf("   piece %d create: %s\n", i, hipGetErrorString(e)); ok = false; break; }
        hs.push_back(h);
        e = hipMemMap((char*)base + mapped, piece, 0, h, 0);
        if (e != hipSuccess) { printf("   piece %d map: %s\n", i, hipGetErrorString(e)); ok = false; break; }
        mapped += piece;
        if (mode == 0) e = hipMemSetAccess((char*)base + mapped - piece, piece, &acc, 1);
        else if (mode == 1) e = hipMemSetAccess(base, mapped, &acc, 1);
        else e = hipSuccess;
        if (e != hipSuccess) { printf("   piece %d (%zu MiB at %.2f GiB): SetAccess: %s\n", i, piece >> 20, (mapped - piece) / 1073741824.0, hipGetErrorString(e)); ok = false; break; }
    }
    if (ok && mode == 2) { hipError_t e = hipMemSetAccess(base, mapped, &acc, 1); if (e != hipSuccess) { printf("   final SetAccess: %s\n", hipGetErrorString(e)); ok = false; } }
    if (ok) { hipError_t e = hipMemset(base, 1, mapped); if (e != hipSuccess) { printf("   memset: %s\n", hipGetErrorString(e)); ok = false; } hipDeviceSynchronize(); }
    if (mapped) hipMemUnmap(base, mapped);
    for (auto h : hs) hipMemRelease(h);
    hipMemAddressFree(base, va);
    return ok;
}

static bool run(size_t va, size_t align, size_t piece, int n_pieces, bool access_whole) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    void* base = nullptr;
    CKR(hipMemAddressReserve(&base, va, align, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    bool ok = true;
    size_t mapped = 0;
    for (int i = 0; i < n_pieces && ok; ++i) {
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, piece, &prop, 0);
        if (e != hipSuccess) { printf("   piece %d: hipMemCreate: %s\n", i, hipGetErrorString(e)); ok = false; break; }
        hs.push_back(h);
        e = hipMemMap((char*)base + mapped, piece, 0, h, 0);
        if (e != hipSuccess) { printf("   piece %d: hipMemMap: %s\n", i, hipGetErrorString(e)); ok = false; break; }
        mapped += piece;
        e = access_whole ? hipMemSetAccess(base, mapped, &acc, 1) : hipMemSetAccess((char*)base + mapped - piece, piece, &acc, 1);
        if (e != hipSuccess) { printf("   piece %d (offset %.1f GiB): hipMemSetAccess: %s\n", i, (mapped - piece) / 1073741824.0, hipGetErrorString(e)); ok = false; break; }
    }
    if (ok) { hipError_t e = hipMemset(base, 1, mapped); if (e != hipSuccess) { printf("   memset: %s\n", hipGetErrorString(e)); ok = false; } hipDeviceSynchronize(); }
    if (mapped) hipMemUnmap(base, mapped);
    for (auto h : hs) hipMemRelease(h);
    hipMemAddressFree(base, va);
    return ok;
}

int main() {
    const size_t G = 1ull << 30, M = 1ull << 20;
    struct { const char* name; size_t va, align, piece; int n; bool whole; } cases[] = {
        {"va 123G align 2M piece 1G x 32, per-piece access", 123 * G, 2 * M, G, 32, false},
        {"va 256G align 4K piece 1G x 32, per-piece access", 256 * G, 4096, G, 32, false},
        {"va 123G align 2M piece 1G x 32, whole-range access", 123 * G, 2 * M, G, 32, true},
        {"va 123G align 2M piece 256M x 128, per-piece access", 123 * G, 2 * M, 256 * M, 128, false},
        {"va 64G align 2M piece 4G x 8, per-piece access", 64 * G, 2 * M, 4 * G, 8, false},
        {"va 123G align 1G piece 1G x 32, per-piece access", 123 * G, G, G, 32, false},
    };
    for (int mode = 0; mode < 3; ++mode) {
        printf("mixed: 14 x 1G then 3 x 64M, mode %d\n   -> %s\n", mode, run_mixed(64 * G, 14, 64 * M, 3, mode) ? "OK" : "failed");
        printf("mixed: 2 x 1G then 3 x 64M, mode %d\n   -> %s\n", mode, run_mixed(64 * G, 2, 64 * M, 3, mode) ? "OK" : "failed");
        printf("mixed: 14 x 1G then 1 x 512M, mode %d\n   -> %s\n", mode, run_mixed(64 * G, 14, 512 * M, 1, mode) ? "OK" : "failed");
    }
    return 0;
    for (auto& c : cases) {
        printf("%s\n", c.name);
        printf("   -> %s\n", run(c.va, c.align, c.piece, c.n, c.whole) ? "OK" : "failed");
    }
    return 0;
}
