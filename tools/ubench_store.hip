// tools/ubench_store.hip -- how fast can 1..N waves of ONE CU (and of all CUs) write matrix rows?
// Pattern of the fill: every wave writes one segment per matrix row, row stride = 64 KiB+4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
template <int W>  // dwords per lane
__global__ void store_k(u32* base, long long row_stride_dw, int rows, int seg_stride_dw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * (blockDim.x >> 6) + wave;
    u32* p = base + (long long)gw * seg_stride_dw + lane * W;
    for (int r = 0; r < rows; ++r) {
        if constexpr (W == 1) { p[0] = r; }
        if constexpr (W == 2) { typedef u32 v2 __attribute__((ext_vector_type(2))); typedef v2 __attribute__((aligned(4))) v2u; *(v2u*)p = v2{(u32)r, (u32)r}; }
        if constexpr (W == 4) { typedef u32 v4 __attribute__((ext_vector_type(4))); typedef v4 __attribute__((aligned(4))) v4u; *(v4u*)p = v4{(u32)r, (u32)r, (u32)r, (u32)r}; }
        p += row_stride_dw;
    }
}
template <int W>
static void run(const char* name, int blocks, int waves, u32* d, long long stride_dw, int rows) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(store_k<W>, dim3(blocks), dim3(64 * waves), 0, 0, d, stride_dw, rows, 64 * W);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(store_k<W>, dim3(blocks), dim3(64 * waves), 0, 0, d, stride_dw, rows, 64 * W);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * waves * rows * 256.0 * W;
    printf("%-10s blocks=%3d waves/blk=%2d : %8.1f GB/s total, %7.2f GB/s per block, %6.2f per wave (%.3f ms)\n", name, blocks, waves,
           bytes / ms * 1e-6, bytes / ms * 1e-6 / blocks, bytes / ms * 1e-6 / blocks / waves, ms);
}
int main() {
    const long long stride_dw = 16385;  // row stride of the 16384^2 int32 matrix
    const int rows = 16384;
    u32* d; hipMalloc(&d, (size_t)stride_dw * (rows + 1) * 4 + (1 << 20));
    for (int waves : {1, 2, 4, 8}) { run<1>("dword", 1, waves, d, stride_dw, rows); run<2>("dwordx2", 1, waves, d, stride_dw, rows); run<4>("dwordx4", 1, waves, d, stride_dw, rows); }
    for (int blocks : {16, 64, 128, 256}) { run<1>("dword", blocks, 1, d, stride_dw, rows); run<4>("dwordx4", blocks > 64 ? 64 : blocks, 1, d, stride_dw, rows); }
    run<1>("dword", 64, 4, d, stride_dw, rows);
    run<1>("dword", 128, 2, d, stride_dw, rows);
    run<1>("dword", 256, 1, d, stride_dw, rows);
    run<2>("dwordx2", 128, 1, d, stride_dw, rows);
    return 0;
}
