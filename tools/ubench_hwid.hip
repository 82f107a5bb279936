// Where do the waves of a 768-thread workgroup run?  Each wave records HW_ID (wave slot, SIMD, CU, SE) and XCC_ID; then pairs of
// waves (0, k) run a VALU loop at the same time: the time per iteration doubles when they share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(768) k_hwid(unsigned* out, unsigned long long* tim, int other, int iters) {
    __shared__ unsigned char big[130000];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0) { out[(blockIdx.x * 12 + wave) * 2] = hw; out[(blockIdx.x * 12 + wave) * 2 + 1] = xcc; }
    big[threadIdx.x] = (unsigned char)hw;
    __syncthreads();
    if (wave == 0 || wave == other) {
        int x = lane, y = big[lane];
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { x = max(x + y, y ^ j); y = __builtin_amdgcn_update_dpp(y, x, 0x138, 0xF, 0xF, false); }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) tim[blockIdx.x * 12 + wave] = t1 - t0;
        if (x == 0x7fffffff) out[0] = y;
    }
}
int main() {
    unsigned* d_out; unsigned long long* d_t;
    hipMalloc(&d_out, 4 * 12 * 2 * 4); hipMalloc(&d_t, 4 * 12 * 8);
    for (int other = -1; other < 12; ++other) {
        hipMemset(d_t, 0, 4 * 12 * 8);
        hipLaunchKernelGGL(k_hwid, dim3(2), dim3(768), 0, 0, d_out, d_t, other, 2000);
        hipDeviceSynchronize();
        std::vector<unsigned> o(4 * 12 * 2); std::vector<unsigned long long> t(4 * 12);
        hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
        if (other < 0) {
            for (int b = 0; b < 2; ++b) { printf("block %d:", b); for (int w = 0; w < 12; ++w) { unsigned h = o[(b * 12 + w) * 2]; printf(" w%d:simd%u/slot%u/cu%u", w, (h >> 4) & 3, h & 15, (h >> 8) & 15); } printf("\n"); }
        }
        printf("waves 0 and %2d busy: clk per VALU pair: wave 0 %.2f, other %.2f\n", other, t[0] / (2000.0 * 16), other > 0 ? t[other] / (2000.0 * 16) : 0.0);
    }
    return 0;
}
