// tools/ubench_ldsorder.hip -- in which order do the lanes of ONE ds_write_b32 (64 consecutive dwords) become visible to another
// wave of the workgroup?  The perm producer's validate-on-use check (sw_systolic.hip, PP_CHKA_1) relies on: if the value of a
// higher lane is visible, so are those of all lower lanes of the same instruction (and of every earlier instruction).
// Writer wave: iteration i stores i into buf[lane].  Reader wave: ds_read_b128 of four consecutive dwords per lane; counts
//   torn      reads in which the four dwords are not all equal (the write was caught half way)
//   inverted  reads in which a HIGHER dword is newer than a lower one (would break the assumption)
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef SHIFT
#define SHIFT 0   // 2: the 16-byte windows straddle the writer's lanes 31|32 (and 15|16, 47|48)
#endif
__global__ void __launch_bounds__(512) k(unsigned long long* out, int iters) {
    __shared__ unsigned buf[64 + 64];
    __shared__ int stop;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 128) buf[threadIdx.x] = 0;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    unsigned long long torn = 0, inv = 0, reads = 0;
    if (wave == 0) {
        for (int i = 1; i <= iters; ++i) {
            __hip_atomic_store(&buf[lane + SHIFT], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_sleep(1);
        }
        __hip_atomic_store(&stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (wave == 1 || wave == 5) {   // one reader on another SIMD, one on the writer's neighbour
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const int off = (lane & 15) * 4 + ((lane >> 4) & 1);          // aligned and unaligned windows
        while (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
            unsigned a, b, c, d;
            if ((off & 3) || off < SHIFT) {   // only aligned windows of written dwords
                a = b = c = d = 0;
            } else {
                u4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)&buf[off]) : "memory");
                a = v.x; b = v.y; c = v.z; d = v.w;
                ++reads;
                if (!(a == b && b == c && c == d)) ++torn;
                if (b > a || c > b || d > c) ++inv;
            }
        }
    }
    // cross-lane windows too: lane l compares buf[l] and buf[l+1] read by ONE ds_read_b64 at an 8-byte boundary
    atomicAdd(&out[0], reads); atomicAdd(&out[1], torn); atomicAdd(&out[2], inv);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 0, 0, d, 2000000);
    hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("ds_read_b128 windows read %llu, torn (write caught half way) %llu, inverted (higher dword newer) %llu\n", h[0], h[1], h[2]);
    return 0;
}
