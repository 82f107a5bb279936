// tools/ubench.hip -- micro-measurements on gfx950 that size the fill kernel's design:
//   per-instruction issue cost for ONE wave on a SIMD, the price of DPP hazards / s_nop,
//   LDS wave-to-wave hand-off latency, cross-CU {tag,value} granule hand-off latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef unsigned int u32;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__device__ __forceinline__ u64 now() { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

// kind: which instruction pattern; each pattern body is 64 copies, looped `iters` times
template <int KIND>
__global__ void issue_k(u64* out, int iters, u32 seed) {
    u32 v = seed + threadIdx.x, w = seed * 3 + threadIdx.x, x = 7, y = 9;
    u32 s = 0;
    __shared__ u32 lds[4096];
    u32 la = threadIdx.x * 4;
    u64 t0 = now();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) { asm volatile(REP64("v_add_u32 %0, %0, %1\n\t") : "+v"(v) : "v"(w)); }                       // dependent VALU
        if constexpr (KIND == 1) { asm volatile(REP16("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t") : "+v"(v), "+v"(w), "+v"(x), "+v"(y) : "v"(la)); } // independent VALU
        if constexpr (KIND == 2) { asm volatile(REP64("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t") : "+v"(v)); } // dpp chain + s_nop 1 (2 instr per rep)
        if constexpr (KIND == 3) { asm volatile(REP64("v_add_u32 %1, %1, %3\n\tv_add_u32 %2, %2, %3\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t") : "+v"(v), "+v"(w), "+v"(x) : "v"(la)); } // dpp chain + 2 fillers (3 instr per rep)
        if constexpr (KIND == 4) { asm volatile(REP64("s_nop 0\n\t")); }
        if constexpr (KIND == 5) { asm volatile(REP64("s_nop 1\n\t")); }
        if constexpr (KIND == 6) { asm volatile(REP64("v_readlane_b32 %1, %0, 63\n\tv_writelane_b32 %0, %1, 3\n\t") : "+v"(v), "=s"(s)); } // 2 instr per rep (has SGPR hazard? assembler does not pad)
        if constexpr (KIND == 7) { asm volatile(REP64("ds_write_b32 %0, %1\n\t") ::"v"(la), "v"(v) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)"); }
        if constexpr (KIND == 8) { asm volatile(REP64("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0xffc, %0\n\t") : "+v"(v), "+v"(la)::"memory"); } // dependent LDS read chain (3 instr)
        if constexpr (KIND == 9) { asm volatile(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s)); }
        if constexpr (KIND == 10) { asm volatile(REP64("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(v), "+s"(s) : "v"(w)); } // VALU+SALU pairs (2 instr per rep)
        if constexpr (KIND == 11) { asm volatile(REP64("v_max3_u32 %0, %0, %1, %2\n\t") : "+v"(v) : "v"(w), "v"(x)); }
        if constexpr (KIND == 12) { asm volatile(REP64("v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc\n\t") : "+v"(v) : "v"(w), "v"(x) : "vcc"); } // 2 instr
        if constexpr (KIND == 13) { asm volatile(REP64("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t") : "+v"(v)); } // dpp chain WITHOUT nops (hazard violated; timing only)
        if constexpr (KIND == 14) { asm volatile(REP64("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32 %1, %1, %0\n\ts_nop 0\n\t") : "+v"(v), "+v"(w)); } // 3 instr
    }
    u64 t1 = now();
    lds[threadIdx.x] = v + w + x + y + s;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = lds[(v & 63)]; }
}

// LDS ping-pong between wave 0 and wave 1 of one workgroup: round trips of a counter
__global__ void lds_pingpong(u64* out, int iters) {
    __shared__ volatile u32 flag[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    u64 t0 = now();
    for (int i = 1; i <= iters; ++i) {
        if (wave == 0) {
            if (lane == 0) flag[0] = i;
            while (flag[1] != (u32)i) {}
        } else {
            while (flag[0] != (u32)i) {}
            if (lane == 0) flag[1] = i;
        }
    }
    u64 t1 = now();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

// cross-workgroup granule ping-pong: block 0 <-> block `peer` through sc1 8-byte granules
typedef __attribute__((address_space(1))) u64 gu64;
__global__ void gl_pingpong(u64* out, u64* cell, int iters, int peer) {
    if (blockIdx.x != 0 && (int)blockIdx.x != peer) return;
    const int me = blockIdx.x == 0 ? 0 : 1;
    u64 t0 = now();
    for (int i = 1; i <= iters; ++i) {
        if (me == 0) {
            if (threadIdx.x == 0) __hip_atomic_store((gu64*)&cell[0], (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load((gu64*)&cell[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)i) {}
        } else {
            while (__hip_atomic_load((gu64*)&cell[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)i) {}
            if (threadIdx.x == 0) __hip_atomic_store((gu64*)&cell[16], (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    u64 t1 = now();
    if (threadIdx.x == 0 && me == 0) { out[0] = t1 - t0; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); out[1] = xcc; }
    if (threadIdx.x == 0 && me == 1) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); out[2] = xcc; }
}

template <int KIND>
static void run_issue(const char* name, int per_rep, u64* d_out, int blocks, int threads) {
    const int iters = 200;
    hipLaunchKernelGGL(issue_k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 10, 1u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(issue_k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 h[2]; hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64 * per_rep;
    printf("%-44s blocks=%3d thr=%4d : %.2f memtime-ticks/instr, %.2f ns/instr (wall %.3f ms)\n", name, blocks, threads,
           (double)h[0] / n, ms * 1e6 / n, ms);
}

int main() {
    u64* d_out; hipMalloc(&d_out, 1 << 16);
    u64* d_cell; hipMalloc(&d_cell, 4096); hipMemset(d_cell, 0, 4096);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int thr : {64, 256, 512}) {
        run_issue<0>("dependent v_add_u32", 1, d_out, 1, thr);
        run_issue<1>("independent v_add_u32 x4", 4, d_out, 1, thr);
    }
    run_issue<11>("dependent v_max3_u32", 1, d_out, 1, 64);
    run_issue<12>("v_cmp+v_cndmask pair", 2, d_out, 1, 64);
    run_issue<2>("dpp chain: s_nop 1 + v_max_u32_dpp", 2, d_out, 1, 64);
    run_issue<13>("dpp chain, no nop (hazard ignored)", 1, d_out, 1, 64);
    run_issue<3>("dpp chain: 2 fillers + v_max_u32_dpp", 3, d_out, 1, 64);
    run_issue<14>("mov_dpp wave_shr + add + s_nop 0", 3, d_out, 1, 64);
    run_issue<4>("s_nop 0", 1, d_out, 1, 64);
    run_issue<5>("s_nop 1", 1, d_out, 1, 64);
    run_issue<9>("dependent s_add_u32", 1, d_out, 1, 64);
    run_issue<10>("v_add + s_add pairs", 2, d_out, 1, 64);
    run_issue<6>("v_readlane + v_writelane pair", 2, d_out, 1, 64);
    run_issue<7>("ds_write_b32 stream", 1, d_out, 1, 64);
    run_issue<8>("ds_read->wait->and chain", 3, d_out, 1, 64);
    run_issue<0>("dependent v_add_u32, all CUs busy", 1, d_out, 256, 256);
    {
        const int iters = 2000;
        hipLaunchKernelGGL(lds_pingpong, dim3(1), dim3(128), 0, 0, d_out, iters);
        hipDeviceSynchronize();
        u64 h; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        printf("LDS ping-pong (2 waves, 1 WG): %.1f ticks per round trip (100 MHz memtime? see ns) \n", (double)h / iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(lds_pingpong, dim3(1), dim3(128), 0, 0, d_out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("LDS ping-pong: %.1f ns per round trip (one-way ~half)\n", ms * 1e6 / iters);
    }
    for (int peer : {1, 8, 9, 64, 255}) {
        const int iters = 2000;
        hipMemset(d_cell, 0, 4096);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(gl_pingpong, dim3(256), dim3(64), 0, 0, d_out, d_cell, 10, peer);
        hipDeviceSynchronize(); hipMemset(d_cell, 0, 4096);
        hipEventRecord(e0);
        hipLaunchKernelGGL(gl_pingpong, dim3(256), dim3(64), 0, 0, d_out, d_cell, iters, peer);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        u64 h[3]; hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
        printf("global granule ping-pong block0(xcc %llu) <-> block%d(xcc %llu): %.1f ns per round trip\n", h[1], peer, h[2], ms * 1e6 / iters);
    }
    return 0;
}
