// tools/ubench_step2.hip -- candidate 4-step groups for the producer (clk per STEP, one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define REP4(x) x x x x
__device__ __forceinline__ u64 now() { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
// literal registers: g ring v[100:107], halo/d regs v[108:111], m v112, sp v113, Z v114, tmp v115
#define STEP(GN, G1, G2, D, BYTE, FLOOR) \
    "v_max_i32_dpp v112, " G1 ", " G1 " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp " D ", " G2 ", v113 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_max3_i32 " GN ", " D ", v112, " FLOOR "\n\t" \
    "v_cmp_eq_u32_sdwa vcc, %[a], %[C] src0_sel:DWORD src1_sel:" BYTE "\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32 v113, %[xm], %[mm], vcc\n\t"
#define STEPZ(GN, G1, G2, D, BYTE) STEP(GN, G1, G2, D, BYTE, "v114") "v_add_u32 v114, v114, %[ngap]\n\t"
#define STEPM(GN, G1, G2, D, BYTE, FLOOR) \
    "v_max_i32_dpp v112, " G1 ", " G1 " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_u32_dpp " D ", " G2 ", v113 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_max3_i32 " GN ", " D ", v112, " FLOOR "\n\t" \
    "v_min_u32_sdwa v115, %[X], %[one] src0_sel:" BYTE " src1_sel:DWORD\n\t" \
    "v_mad_i32_i24 v113, v115, %[neg6], %[mm]\n\t"
template <int KIND>
__global__ void k(u64* out, int iters) {
    __shared__ __attribute__((aligned(16))) u32 lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = i;
    u32 a = threadIdx.x & 3, C = 0x01020300u, xm = 1, mm = 7, ngap = 2, waddr = threadIdx.x * 1040, raddr = 63 * 1040 + 16, one = 1, neg6 = (u32)-6, X = 0x00010200u;
    asm volatile("v_mov_b32 v100, 1\n\tv_mov_b32 v101, 2\n\tv_mov_b32 v102, 3\n\tv_mov_b32 v103, 4\n\tv_mov_b32 v104, 5\n\tv_mov_b32 v105, 6\n\tv_mov_b32 v106, 7\n\tv_mov_b32 v107, 8\n\t"
                 "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 1\n\tv_mov_b32 v114, 9\n\tv_mov_b32 v115, 0" :::
                 "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
    u64 t0 = now();
    for (int i = 0; i < iters; ++i) {
#define OPS : : [a] "v"(a), [C] "v"(C), [xm] "v"(xm), [mm] "v"(mm), [ngap] "v"(ngap), [waddr] "v"(waddr), [raddr] "v"(raddr), [one] "v"(one), [neg6] "v"(neg6), [X] "v"(X) : "vcc", "memory"
        if constexpr (KIND == 0)  // VGPR floor + addz, b128 write, b128 halo read into the d registers
            asm volatile(REP4(
                STEPZ("v100","v107","v106","v108","BYTE_0") STEPZ("v101","v100","v107","v109","BYTE_1") STEPZ("v102","v101","v100","v110","BYTE_2") STEPZ("v103","v102","v101","v111","BYTE_3")
                "ds_write_b128 %[waddr], v[100:103] offset:64\n\t" "ds_read_b128 v[108:111], %[raddr]\n\t"
                STEPZ("v104","v103","v102","v108","BYTE_0") STEPZ("v105","v104","v103","v109","BYTE_1") STEPZ("v106","v105","v104","v110","BYTE_2") STEPZ("v107","v106","v105","v111","BYTE_3")
                "ds_write_b128 %[waddr], v[104:107] offset:80\n\t" "ds_read_b128 v[108:111], %[raddr]\n\t") OPS);
        if constexpr (KIND == 1)  // inline-constant floor (re-biased space), no addz
            asm volatile(REP4(
                STEP("v100","v107","v106","v108","BYTE_0","2") STEP("v101","v100","v107","v109","BYTE_1","4") STEP("v102","v101","v100","v110","BYTE_2","6") STEP("v103","v102","v101","v111","BYTE_3","8")
                "ds_write_b128 %[waddr], v[100:103] offset:64\n\t" "ds_read_b128 v[108:111], %[raddr]\n\t"
                STEP("v104","v103","v102","v108","BYTE_0","10") STEP("v105","v104","v103","v109","BYTE_1","12") STEP("v106","v105","v104","v110","BYTE_2","14") STEP("v107","v106","v105","v111","BYTE_3","16")
                "ds_write_b128 %[waddr], v[104:107] offset:80\n\t" "ds_read_b128 v[108:111], %[raddr]\n\t") OPS);
        if constexpr (KIND == 2)  // as 1 but 4x ds_read_b32 for the halo
            asm volatile(REP4(
                STEP("v100","v107","v106","v108","BYTE_0","2") STEP("v101","v100","v107","v109","BYTE_1","4") STEP("v102","v101","v100","v110","BYTE_2","6") STEP("v103","v102","v101","v111","BYTE_3","8")
                "ds_write_b128 %[waddr], v[100:103] offset:64\n\t" "ds_read_b32 v108, %[raddr]\n\tds_read_b32 v109, %[raddr] offset:4\n\tds_read_b32 v110, %[raddr] offset:8\n\tds_read_b32 v111, %[raddr] offset:12\n\t"
                STEP("v104","v103","v102","v108","BYTE_0","10") STEP("v105","v104","v103","v109","BYTE_1","12") STEP("v106","v105","v104","v110","BYTE_2","14") STEP("v107","v106","v105","v111","BYTE_3","16")
                "ds_write_b128 %[waddr], v[104:107] offset:80\n\t" "ds_read_b32 v108, %[raddr]\n\tds_read_b32 v109, %[raddr] offset:4\n\tds_read_b32 v110, %[raddr] offset:8\n\tds_read_b32 v111, %[raddr] offset:12\n\t") OPS);
        if constexpr (KIND == 3)  // as 1 with min_sdwa + mad instead of cmp/cndmask
            asm volatile(REP4(
                STEPM("v100","v107","v106","v108","BYTE_0","2") STEPM("v101","v100","v107","v109","BYTE_1","4") STEPM("v102","v101","v100","v110","BYTE_2","6") STEPM("v103","v102","v101","v111","BYTE_3","8")
                "ds_write_b128 %[waddr], v[100:103] offset:64\n\t" "ds_read_b128 v[108:111], %[raddr]\n\t"
                STEPM("v104","v103","v102","v108","BYTE_0","10") STEPM("v105","v104","v103","v109","BYTE_1","12") STEPM("v106","v105","v104","v110","BYTE_2","14") STEPM("v107","v106","v105","v111","BYTE_3","16")
                "ds_write_b128 %[waddr], v[104:107] offset:80\n\t" "ds_read_b128 v[108:111], %[raddr]\n\t") OPS);
        if constexpr (KIND == 4)  // as 1 but no LDS ops at all
            asm volatile(REP4(
                STEP("v100","v107","v106","v108","BYTE_0","2") STEP("v101","v100","v107","v109","BYTE_1","4") STEP("v102","v101","v100","v110","BYTE_2","6") STEP("v103","v102","v101","v111","BYTE_3","8")
                STEP("v104","v103","v102","v108","BYTE_0","10") STEP("v105","v104","v103","v109","BYTE_1","12") STEP("v106","v105","v104","v110","BYTE_2","14") STEP("v107","v106","v105","v111","BYTE_3","16")) OPS);
    }
    u64 t1 = now();
    u32 r; asm volatile("v_add_u32 %0, v100, v107" : "=v"(r));
    lds[threadIdx.x] = r;
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = lds[5]; }
}
template <int KIND> static void run(const char* name, u64* d) {
    const int iters = 300;
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
    u64 h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-78s %6.1f clk/step\n", name, (double)h / (iters * 32.0));
}
int main() {
    u64* d; hipMalloc(&d, 64);
    run<0>("VGPR floor+add, cmp/cnd, write_b128 + halo read_b128 per 4 steps", d);
    run<1>("inline-constant floor, cmp/cnd, write_b128 + halo read_b128 per 4 steps", d);
    run<2>("  same with 4x ds_read_b32 for the halo", d);
    run<3>("  same with min_sdwa+mad instead of cmp/cnd", d);
    run<4>("  same without any LDS op", d);
    return 0;
}
