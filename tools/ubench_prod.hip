// tools/ubench_prod.hip -- the REAL perm producer loop (swk::producer_perm from sw_systolic.hip) run by one wave alone on
// a CU, every counter preset so that it never waits: clocks per step of the loop itself.  Build variants with
// -DPP_NO_EXPORT etc. to see what each part costs.
#include "../smith-waterman_amd/csrc/sw_systolic.hip"
#include <cstdio>
#include <vector>
using namespace swk;
__global__ void __launch_bounds__(64) prod_k(const unsigned char* codes, u32* edge, int UT, u64* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[64 * SY_LSTR];
    __shared__ __attribute__((aligned(16))) u32 halo[SY_RH];
    __shared__ __attribute__((aligned(16))) int cons[8];
    __shared__ int left_cnt, right_cnt, prog;
    const int lane = threadIdx.x;
    for (int i = lane; i < SY_RH; i += 64) halo[i] = 0x55010000u + i;
    if (lane < 8) cons[lane] = 1 << 24;
    if (lane == 0) { left_cnt = 0x7fffffff; right_cnt = 0x7fffffff; prog = 0; }
    __syncthreads();
    const u32 plo = 0x01070101u, phi = 0x9c010101u;
    int polls[2];
    const uint64_t eb = (uint64_t)(uintptr_t)edge;
    const sw_i32x4p erc = {(int)(u32)eb, (int)(u32)(eb >> 32), 0x7FFFFF00, 0x00020000};
    u64 r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    const int st = producer_perm_progress(plo, phi, 2u, (u32)(size_t)&ring[0] + lane * SY_LSTR, 63u - lane, 0x55010000u, 0x55010000u + 2 * lane, 0x55010000u + 2 * lane - 100,
                                 lane == 63 ? 0u : SY_OOB, codes + 64, (u32)(size_t)&halo[0], 0, (u32)(size_t)&left_cnt, (u32)(size_t)&cons[0],
                                 (u32)(size_t)&right_cnt, (u32)(size_t)&prog, UT, 32, 0x7ffffff0, 30 - SY_R, -SY_R - 32, SY_RH * 4 - 1, erc, 64, polls);
    u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; clk[2] = (u64)st; clk[3] = (u64)polls[0]; clk[4] = (u64)polls[1]; }
}
int main() {
    const int UT = 16384;
    unsigned char* d_codes; u32* d_edge; u64* d_clk;
    hipMalloc(&d_codes, UT + 4096); hipMemset(d_codes, 1, UT + 4096);
    hipMalloc(&d_edge, (UT + 256) * 4); hipMalloc(&d_clk, 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(prod_k, dim3(1), dim3(64), 0, 0, d_codes, d_edge, UT, d_clk);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        u64 c[5]; hipMemcpy(c, d_clk, 40, hipMemcpyDeviceToHost);
        printf("%s: %.1f clk/step, %.2f ns/step (status %llu, polls %llu/%llu)\n", VARIANT, (double)c[0] / UT, (double)c[1] * 10.0 / UT, c[2], c[3], c[4]);
    }
    return 0;
}
