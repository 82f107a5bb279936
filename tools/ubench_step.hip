// tools/ubench_step.hip -- cycles per anti-diagonal step of the producer's instruction sequence,
// with single instructions ablated, one wave on one CU (s_memtime ticks = shader clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
__device__ __forceinline__ u64 now() { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

#define CMP  "v_cmp_eq_u32_sdwa vcc, %[a], %[C] src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define CMPN "v_cmp_eq_u32 vcc, %[a], %[C]\n\t"
#define RDL  "v_readlane_b32 %[sh], %[Hv], 3\n\t"
#define MAXD "v_max_i32_dpp %[m], %[G1], %[G1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define CND  "v_cndmask_b32 %[sp], %[xm], %[mm], vcc\n\t"
#define WRL  "v_writelane_b32 %[d], %[sh], 0\n\t"
#define ADDD "v_add_u32_dpp %[d], %[G2], %[sp] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define MAX3 "v_max3_i32 %[G1], %[d], %[m], %[Z]\n\t"     /* writes G1 directly (G2 rotation dropped: timing only) */
#define ADDZ "v_add_u32 %[Z], %[Z], %[ngap]\n\t"
#define DSW  "ds_write_b32 %[waddr], %[G1] offset:256\n\t"
#define NOP2 "s_nop 1\n\t"

template <int KIND>
__global__ void step_k(u64* out, int iters, u32 seed) {
    u32 G1 = seed + threadIdx.x, G2 = seed * 7 + threadIdx.x, m = 0, d = 0, sp = 1, Z = 5, Hv = threadIdx.x, a = threadIdx.x & 3, C = 0x01020300u;
    u32 ngap = 2, xm = 1, mm = 7, waddr = threadIdx.x * 4;
    int sh = 0;
    __shared__ u32 lds[8192];
    lds[threadIdx.x] = 0;
    u64 t0 = now();
    for (int i = 0; i < iters; ++i) {
#define BODY(SEQ) asm volatile(REP16(SEQ) : [sh] "+s"(sh), [m] "+v"(m), [d] "+v"(d), [G1] "+v"(G1), [Z] "+v"(Z), [sp] "+v"(sp) \
                               : [Hv] "v"(Hv), [G2] "v"(G2), [C] "v"(C), [a] "v"(a), [ngap] "v"(ngap), [waddr] "v"(waddr), [xm] "v"(xm), [mm] "v"(mm) : "vcc", "memory")
        if constexpr (KIND == 0) BODY(CMP RDL MAXD CND WRL ADDD MAX3 ADDZ DSW);         // the real step
        if constexpr (KIND == 1) BODY(CMP RDL MAXD CND WRL ADDD MAX3 ADDZ);             // no ds_write
        if constexpr (KIND == 2) BODY(CMPN RDL MAXD CND WRL ADDD MAX3 ADDZ DSW);        // plain compare instead of SDWA
        if constexpr (KIND == 3) BODY(CMP MAXD CND ADDD MAX3 ADDZ DSW);                 // no readlane/writelane
        if constexpr (KIND == 4) BODY(MAXD ADDD MAX3);                                  // bare recurrence
        if constexpr (KIND == 5) BODY(MAXD MAX3);                                       // bare chain
        if constexpr (KIND == 6) BODY(MAXD NOP2 MAX3 NOP2);                             // chain + hazard nops
        if constexpr (KIND == 7) BODY(CMP RDL MAXD CND WRL ADDD MAX3 ADDZ DSW NOP2);
        if constexpr (KIND == 8) BODY(MAXD CMP RDL CND WRL ADDD MAX3 ADDZ DSW);         // chain op first
        if constexpr (KIND == 9) BODY(ADDD MAXD MAX3 CMP RDL ADDZ DSW CND WRL);         // d before m; sp/halo for the NEXT step after max3
        if constexpr (KIND == 10) BODY(MAXD ADDD MAX3 DSW);
        if constexpr (KIND == 11) BODY(MAXD ADDD MAX3 ADDZ CMP CND);
        if constexpr (KIND == 12) BODY(MAXD ADDD MAX3 RDL ADDZ WRL);
    }
    u64 t1 = now();
    lds[threadIdx.x + 64] = G1 + m + d + sp + Z + sh;
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = lds[64]; }
}
template <int KIND> static void run(const char* name, u64* d_out) {
    const int iters = 400;
    hipLaunchKernelGGL(step_k<KIND>, dim3(1), dim3(64), 0, 0, d_out, 10, 1u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(step_k<KIND>, dim3(1), dim3(64), 0, 0, d_out, iters, 1u);
    hipDeviceSynchronize();
    u64 h; hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
    printf("%-64s %6.1f clk/step\n", name, (double)h / (iters * 16.0));
}
int main() {
    u64* d; hipMalloc(&d, 64);
    run<0>("real step: cmp_sdwa rdl max_dpp cnd wrl add_dpp max3 addz ds_write", d);
    run<1>("  without ds_write", d);
    run<2>("  plain v_cmp instead of SDWA", d);
    run<3>("  without readlane/writelane", d);
    run<4>("bare recurrence: max_dpp add_dpp max3", d);
    run<5>("bare chain: max_dpp max3", d);
    run<6>("bare chain with s_nop 1 after each", d);
    run<7>("real step + s_nop 1", d);
    run<8>("real step, chain op (max_dpp) first", d);
    run<9>("reordered: add_dpp max_dpp max3 | cmp rdl addz ds_write cnd wrl", d);
    run<10>("max_dpp add_dpp max3 ds_write", d);
    run<11>("max_dpp add_dpp max3 addz cmp cnd", d);
    run<12>("max_dpp add_dpp max3 rdl addz wrl", d);
    return 0;
}
