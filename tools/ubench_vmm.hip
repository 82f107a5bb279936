// Is the HIP virtual-memory API usable here?  hipMemCreate chunks, map them (also the same chunk at two addresses), run a kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fillk(unsigned* p, size_t n, unsigned v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i; }
int main() {
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t grec = 0;
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity min %zu recommended %zu\n", gran, grec);
    const size_t chunk = 256ull << 20; const int N = 8;
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    auto t1 = std::chrono::steady_clock::now();
    printf("hipMemCreate %d x 256 MiB: %.2f ms\n", N, std::chrono::duration<double, std::milli>(t1 - t0).count());
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, chunk * (N + 2), 0, nullptr, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, h[N - 1 - i], 0));   // reversed order
    CK(hipMemSetAccess(va, chunk * N, &acc, 1));
    t1 = std::chrono::steady_clock::now();
    printf("map + access: %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    // alias: chunk 0 again at the end
    hipError_t ea = hipMemMap((char*)va + N * chunk, chunk, 0, h[0], 0);
    printf("alias map of an already mapped handle: %s\n", hipGetErrorString(ea));
    if (ea == hipSuccess) CK(hipMemSetAccess((char*)va + N * chunk, chunk, &acc, 1));
    hipLaunchKernelGGL(fillk, dim3(1024), dim3(256), 0, nullptr, (unsigned*)va, chunk * N / 4, 7u);
    CK(hipDeviceSynchronize());
    unsigned x[2] = {0, 0};
    CK(hipMemcpy(&x[0], (char*)va + (N - 1) * chunk + 40, 4, hipMemcpyDeviceToHost));      // chunk h[0] through its first mapping
    if (ea == hipSuccess) CK(hipMemcpy(&x[1], (char*)va + N * chunk + 40, 4, hipMemcpyDeviceToHost));
    printf("value through mapping 1: %u, through the alias: %u (expect equal)\n", x[0], x[1]);
    CK(hipMemUnmap(va, chunk * N));
    if (ea == hipSuccess) CK(hipMemUnmap((char*)va + N * chunk, chunk));
    for (int i = 0; i < N; ++i) CK(hipMemRelease(h[i]));
    CK(hipMemAddressFree(va, chunk * (N + 2)));
    printf("vmm ok\n");
    return 0;
}
