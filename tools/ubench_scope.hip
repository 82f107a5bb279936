// tools/ubench_scope.hip -- which cache-scope bits does a cross-workgroup {flag} hand-off need when both
// workgroups run on the SAME XCD, and what does it cost?  Block 0 <-> block `peer` ping-pong through two
// 8-byte cells with every combination of store / load scope bits; bounded polling (a combination that never
// becomes visible is reported as "not visible", nothing hangs).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_scope tools/ubench_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

template <int LD>
__device__ __forceinline__ u64 ld(const u64* p) {
    u64 v;
    if constexpr (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == 4) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LD == 5 || LD == 6) {   // the scalar path (its own way to the L2, not behind the CU's vector stores): glc / plain + cache invalidate
        u64 sv;
        const u64* sp = (const u64*)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)p) + 0;   // (placeholder, replaced below)
        (void)sp;
        unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)p & 0xffffffffu)), hi = __builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)p >> 32));
        u64 base = ((u64)hi << 32) | lo;
        if constexpr (LD == 5) asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(base) : "memory");
        else asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(base) : "memory");
        v = sv;
    }
    return v;
}
template <int ST>
__device__ __forceinline__ void st(u64* p, u64 v) {
    if constexpr (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}

template <int ST, int LD>
__global__ void pingpong(u64* out, u64* cell, int iters, int peer, u64 base) {
    if (blockIdx.x != 0 && (int)blockIdx.x != peer) return;
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x == 0 ? 0 : 1;
    u64* mine = cell + (me ? 16 : 0);
    u64* theirs = cell + (me ? 0 : 16);
    u64* abortf = cell + 32;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    int done = 0;
    for (int i = 1; i <= iters; ++i) {
        const u64 want = base + (u64)i;
        if (me == 0) st<ST>(mine, want);
        int polls = 0;
        while (ld<LD>(theirs) != want) {
            if (++polls > 200000 || ld<3>(abortf) != 0) { st<3>(abortf, 1); goto out; }
        }
        if (me == 1) st<ST>(mine, want);
        done = i;
    }
out:
    const u64 t1 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[me * 4 + 0] = t1 - t0; out[me * 4 + 1] = xcc; out[me * 4 + 2] = done;
}

template <int ST, int LD>
static void run(u64* d_out, u64* d_cell, int peer, u64& base) {
    const int iters = 2000;
    static const char* nm[] = {"-", "sc0", "sc1", "sc0 sc1", "nt", "s_load glc", "s_dcache_inv+s_load"};
    hipMemset(d_cell, 0, 512);
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(256), dim3(64), 0, 0, d_out, d_cell, iters, peer, base);
    hipDeviceSynchronize();
    u64 h[8];
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    base += 1u << 20;
    if ((int)h[2] == iters)
        printf("peer %3d (xcc %llu/%llu)  store %-7s load %-7s : %7.1f ns per round trip\n", peer, h[1], h[5], nm[ST], nm[LD], (double)h[0] * 10.0 / iters);
    else
        printf("peer %3d (xcc %llu/%llu)  store %-7s load %-7s : NOT VISIBLE (stopped after %llu round trips)\n", peer, h[1], h[5], nm[ST], nm[LD], h[2]);
}

int main() {
    u64 *d_out, *d_cell;
    hipMalloc(&d_out, 4096); hipMalloc(&d_cell, 4096);
    u64 base = 1;
    for (int peer : {8, 64, 1}) {   // 8, 64: same XCD as block 0 (round-robin dealing); 1: the next XCD
        run<2, 2>(d_out, d_cell, peer, base);   // what the library uses today
        run<2, 1>(d_out, d_cell, peer, base);
        run<2, 0>(d_out, d_cell, peer, base);
        run<2, 4>(d_out, d_cell, peer, base);
        run<1, 1>(d_out, d_cell, peer, base);
        run<0, 1>(d_out, d_cell, peer, base);
        run<1, 2>(d_out, d_cell, peer, base);
        run<0, 0>(d_out, d_cell, peer, base);
        run<3, 3>(d_out, d_cell, peer, base);
        run<1, 5>(d_out, d_cell, peer, base);
        run<2, 5>(d_out, d_cell, peer, base);
        run<1, 6>(d_out, d_cell, peer, base);
        run<2, 6>(d_out, d_cell, peer, base);
    }
    return 0;
}
