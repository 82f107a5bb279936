// tools/ubench_perm.hip -- the "perm" producer step (round 2): 4 VALU per anti-diagonal step
//   A: P    = max3(t[u-2], g[u-1], Z)                 v_max3_i32
//   B: Z   += ngap                                    v_add_u32
//   C: t[u-1] = g[u-1] + sext(S.byte)                 v_add_u32_sdwa   (S: 16 score bytes per block from 4 v_perm_b32)
//   D: g[u] = max(P[l-1], g[u-1][l])                  v_max_i32_dpp wave_shr:1 (lane 0 keeps the pre-loaded halo)
// one wave, verified against a host restatement of the same recurrence, then timed (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u64 now_rt() { u64 t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }
__device__ __forceinline__ u64 now() { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return t; }

// g[k] = v(100+k), t[k] = v(116 + (k&3)), P = v120, Z = v121
#define PP_STEP(GK, GP, TP2, TP1, SREG, BYTE)                                                     \
    "v_max3_i32 v120, " TP2 ", " GP ", v121\n\t"                                                  \
    "v_add_u32 v121, v121, %[ngap]\n\t"                                                           \
    "v_add_u32_sdwa " TP1 ", " GP ", sext(" SREG ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" BYTE "\n\t" \
    "v_max_i32_dpp " GK ", v120, " GP " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
// one 16-step block; SP3: the score dword holding step -1 of this block (= byte 3 of the previous block's last dword);
// S0..S3: this block's score dwords.  C of step k uses the byte of step k-1.
#define PP_G0(SP3, S0)                                                                            \
    PP_STEP("v100", "v115", "v118", "v119", SP3, "BYTE_3") PP_STEP("v101", "v100", "v119", "v116", S0, "BYTE_0")   \
    PP_STEP("v102", "v101", "v116", "v117", S0, "BYTE_1") PP_STEP("v103", "v102", "v117", "v118", S0, "BYTE_2")
#define PP_G1(S0, S1)                                                                             \
    PP_STEP("v104", "v103", "v118", "v119", S0, "BYTE_3") PP_STEP("v105", "v104", "v119", "v116", S1, "BYTE_0")   \
    PP_STEP("v106", "v105", "v116", "v117", S1, "BYTE_1") PP_STEP("v107", "v106", "v117", "v118", S1, "BYTE_2")
#define PP_G2(S1, S2)                                                                             \
    PP_STEP("v108", "v107", "v118", "v119", S1, "BYTE_3") PP_STEP("v109", "v108", "v119", "v116", S2, "BYTE_0")   \
    PP_STEP("v110", "v109", "v116", "v117", S2, "BYTE_1") PP_STEP("v111", "v110", "v117", "v118", S2, "BYTE_2")
#define PP_G3(S2, S3)                                                                             \
    PP_STEP("v112", "v111", "v118", "v119", S2, "BYTE_3") PP_STEP("v113", "v112", "v119", "v116", S3, "BYTE_0")   \
    PP_STEP("v114", "v113", "v116", "v117", S3, "BYTE_1") PP_STEP("v115", "v114", "v117", "v118", S3, "BYTE_2")
// block: codes in C0..C3 (4 dwords) -> scores S0..S3; halo groups 2,3 of this block and 0,1 of the next are fetched on
// the way; ring writes after each group.  HB: byte offset of this block's first halo value; WO: ring byte offset
#define PP_BLOCK(SP3, S0, S1, S2, S3, C0, C1, C2, C3, HB, WO)                                     \
    "v_perm_b32 " S0 ", %[phi], %[plo], " C0 "\n\t"                                               \
    "v_perm_b32 " S1 ", %[phi], %[plo], " C1 "\n\t"                                               \
    "v_perm_b32 " S2 ", %[phi], %[plo], " C2 "\n\t"                                               \
    "v_perm_b32 " S3 ", %[phi], %[plo], " C3 "\n\t"                                               \
    "s_waitcnt lgkmcnt(2)\n\t"                        /* halo group 0 landed (L0', then W3?, L1' behind it) */ \
    PP_G0(SP3, S0)                                                                                \
    "ds_write_b128 %[waddr], v[100:103] offset:" #WO "+0\n\t"                                     \
    "ds_read_b128 v[108:111], %[haddr] offset:" #HB "+32\n\t"                                     \
    "s_waitcnt lgkmcnt(2)\n\t"                        /* halo group 1 landed */                   \
    PP_G1(S0, S1)                                                                                 \
    "ds_write_b128 %[waddr], v[104:107] offset:" #WO "+16\n\t"                                    \
    "ds_read_b128 v[112:115], %[haddr] offset:" #HB "+48\n\t"                                     \
    "s_waitcnt lgkmcnt(2)\n\t"                        /* halo group 2 landed */                   \
    PP_G2(S1, S2)                                                                                 \
    "ds_write_b128 %[waddr], v[108:111] offset:" #WO "+32\n\t"                                    \
    "ds_read_b128 v[100:103], %[haddr] offset:" #HB "+64\n\t"                                     \
    "s_waitcnt lgkmcnt(2)\n\t"                        /* halo group 3 landed */                   \
    PP_G3(S2, S3)                                                                                 \
    "ds_write_b128 %[waddr], v[112:115] offset:" #WO "+48\n\t"                                    \
    "ds_read_b128 v[104:107], %[haddr] offset:" #HB "+80\n\t"

// NOTE: this first cut loads each halo group 8 steps ahead right behind a ring write; group q+2 while group q+1 runs.
//       (simplification of the schedule discussed in DESIGN: measured here first)

template <int VERIFY>
__global__ void perm_k(const unsigned char* codes, const int* halo_g, const int* g0_g, const int* tm1_g, const u32* prof, int ngap_i, int z1,
                       int nchunks, int* out, u64* clk) {
    __shared__ __attribute__((aligned(16))) int ring[64 * 260 + 64];   // lane-major: lane l at l*1040 bytes, 256 steps + pad
    __shared__ __attribute__((aligned(16))) int halo[1024 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024 + 64; i += 64) halo[i] = halo_g[i];
    __syncthreads();
    const u32 plo = prof[2 * lane], phi = prof[2 * lane + 1];
    const u32 ngap = (u32)ngap_i;
    const u32 waddr = (u32)(size_t)&ring[0] + lane * 1040;
    const u32 haddr = (u32)(size_t)&halo[0];
    // code stream: lane l, local step u (1-based) at codes[OFF + u - l]; the 16-byte window of block b starts at step 16b+1;
    // the window of "block -1" (steps -15..0) supplies the score of step 0
    const unsigned char* cp = codes + 128 - 78;   // lane l, step u at codes[128 + u - l]; per-lane offset 63 - l on top of this base
    const u32 voff = 63u - (u32)lane;
    const int g0 = g0_g[lane], tm1 = tm1_g[lane];
    u64 r0 = now_rt();
    u64 t0 = now();
    asm volatile(
        "v_mov_b32 v115, %[g0]\n\t"
        "v_mov_b32 v118, %[tm1]\n\t"
        "v_mov_b32 v121, %[z1]\n\t"
        "s_mov_b64 s[92:93], %[cp]\n\t"
        "s_mov_b32 s88, %[nch]\n\t"
        "s_mov_b32 s90, 0\n\t"
        "v_mov_b32 v96, %[voff]\n\t"
        "global_load_dwordx4 v[76:79], v96, s[92:93] offset:0\n\t"      /* block -1: only its last dword matters */
        "global_load_dwordx4 v[64:67], v96, s[92:93] offset:16\n\t"
        "global_load_dwordx4 v[68:71], v96, s[92:93] offset:32\n\t"
        "global_load_dwordx4 v[72:75], v96, s[92:93] offset:48\n\t"
        "s_add_u32 s92, s92, 64\n\t"
        "s_addc_u32 s93, s93, 0\n\t"
        "ds_read_b128 v[100:103], %[haddr] offset:0\n\t"
        "ds_read_b128 v[104:107], %[haddr] offset:16\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "v_perm_b32 v63, %[phi], %[plo], v79\n\t"
        "v_mov_b32 v98, %[haddr]\n\t"
        "v_mov_b32 v99, %[waddr]\n"
        "Lchunk_%=:\n\t"
        // chunk: 4 blocks; codes for block b+3 are loaded at the start of block b
        "s_waitcnt vmcnt(2)\n\t"
        "global_load_dwordx4 v[76:79], v96, s[92:93] offset:0\n\t"
#define HADDR "v98"
#define WADDR "v99"
        "v_perm_b32 v122, %[phi], %[plo], v64\n\t"
        "v_perm_b32 v123, %[phi], %[plo], v65\n\t"
        "v_perm_b32 v124, %[phi], %[plo], v66\n\t"
        "v_perm_b32 v125, %[phi], %[plo], v67\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        PP_G0("v63", "v122")
        "ds_write_b128 v99, v[100:103] offset:0\n\t"
        "ds_read_b128 v[108:111], v98 offset:32\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G1("v122", "v123")
        "ds_write_b128 v99, v[104:107] offset:16\n\t"
        "ds_read_b128 v[112:115], v98 offset:48\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G2("v123", "v124")
        "ds_write_b128 v99, v[108:111] offset:32\n\t"
        "ds_read_b128 v[100:103], v98 offset:64\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G3("v124", "v125")
        "ds_write_b128 v99, v[112:115] offset:48\n\t"
        "ds_read_b128 v[104:107], v98 offset:80\n\t"
        // block 1
        "s_waitcnt vmcnt(2)\n\t"
        "global_load_dwordx4 v[64:67], v96, s[92:93] offset:16\n\t"
        "v_perm_b32 v60, %[phi], %[plo], v68\n\t"
        "v_perm_b32 v61, %[phi], %[plo], v69\n\t"
        "v_perm_b32 v62, %[phi], %[plo], v70\n\t"
        "v_perm_b32 v63, %[phi], %[plo], v71\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G0("v125", "v60")
        "ds_write_b128 v99, v[100:103] offset:64\n\t"
        "ds_read_b128 v[108:111], v98 offset:96\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G1("v60", "v61")
        "ds_write_b128 v99, v[104:107] offset:80\n\t"
        "ds_read_b128 v[112:115], v98 offset:112\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G2("v61", "v62")
        "ds_write_b128 v99, v[108:111] offset:96\n\t"
        "ds_read_b128 v[100:103], v98 offset:128\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G3("v62", "v63")
        "ds_write_b128 v99, v[112:115] offset:112\n\t"
        "ds_read_b128 v[104:107], v98 offset:144\n\t"
        // block 2
        "s_waitcnt vmcnt(2)\n\t"
        "global_load_dwordx4 v[68:71], v96, s[92:93] offset:32\n\t"
        "v_perm_b32 v122, %[phi], %[plo], v72\n\t"
        "v_perm_b32 v123, %[phi], %[plo], v73\n\t"
        "v_perm_b32 v124, %[phi], %[plo], v74\n\t"
        "v_perm_b32 v125, %[phi], %[plo], v75\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G0("v63", "v122")
        "ds_write_b128 v99, v[100:103] offset:128\n\t"
        "ds_read_b128 v[108:111], v98 offset:160\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G1("v122", "v123")
        "ds_write_b128 v99, v[104:107] offset:144\n\t"
        "ds_read_b128 v[112:115], v98 offset:176\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G2("v123", "v124")
        "ds_write_b128 v99, v[108:111] offset:160\n\t"
        "ds_read_b128 v[100:103], v98 offset:192\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G3("v124", "v125")
        "ds_write_b128 v99, v[112:115] offset:176\n\t"
        "ds_read_b128 v[104:107], v98 offset:208\n\t"
        // block 3
        "s_waitcnt vmcnt(2)\n\t"
        "global_load_dwordx4 v[72:75], v96, s[92:93] offset:48\n\t"
        "v_perm_b32 v60, %[phi], %[plo], v76\n\t"
        "v_perm_b32 v61, %[phi], %[plo], v77\n\t"
        "v_perm_b32 v62, %[phi], %[plo], v78\n\t"
        "v_perm_b32 v63, %[phi], %[plo], v79\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G0("v125", "v60")
        "ds_write_b128 v99, v[100:103] offset:192\n\t"
        "ds_read_b128 v[108:111], v98 offset:224\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G1("v60", "v61")
        "ds_write_b128 v99, v[104:107] offset:208\n\t"
        "ds_read_b128 v[112:115], v98 offset:240\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G2("v61", "v62")
        "ds_write_b128 v99, v[108:111] offset:224\n\t"
        "ds_read_b128 v[100:103], v98 offset:256\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        PP_G3("v62", "v63")
        "ds_write_b128 v99, v[112:115] offset:240\n\t"
        "ds_read_b128 v[104:107], v98 offset:272\n\t"
        // next chunk: halo advances 256 bytes (the test's halo array is linear), the ring wraps at 1024 bytes
        "v_add_u32 v98, 0x100, v98\n\t"
        "s_add_i32 s90, s90, 256\n\t"
        "s_and_b32 s90, s90, 1023\n\t"
        "v_add_u32 v99, s90, %[waddr]\n\t"
        "s_add_u32 s92, s92, 64\n\t"
        "s_addc_u32 s93, s93, 0\n\t"
        "s_add_i32 s88, s88, -1\n\t"
        "s_cmp_gt_i32 s88, 0\n\t"
        "s_cbranch_scc1 Lchunk_%=\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        :
        : [g0] "v"(g0), [tm1] "v"(tm1), [z1] "v"(z1), [cp] "s"(cp), [nch] "s"(nchunks), [voff] "v"(voff), [haddr] "v"(haddr), [waddr] "v"(waddr),
          [phi] "v"(phi), [plo] "v"(plo), [ngap] "v"(ngap)
        : "vcc", "scc", "memory", "s88", "s90", "s92", "s93", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
          "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v96", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106",
          "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",
          "v123", "v124", "v125");
    u64 t1 = now();
    u64 r1 = now_rt();
    __syncthreads();
    if (VERIFY) {
        // ring holds the last 256 steps: lane l step u at l*1040 + ((u-1)&255)*4
        const int nsteps = nchunks * 64;
        for (int u = 1; u <= nsteps && u <= 256; ++u) out[(u - 1) * 64 + lane] = ring[lane * 260 + ((u - 1) & 255)];
    }
    if (lane == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    const int nch_v = 4, nsteps = nch_v * 64;   // verification: 256 steps (fits the ring)
    srand(12345);
    const int OFF = 128;
    std::vector<unsigned char> codes(OFF + 64 * 1100 + 256);
    for (auto& c : codes) c = rand() % 8;
    std::vector<int> halo(1024 + 64), g0(64), tm1(64);
    std::vector<u32> prof(128);
    signed char pb[64][8];
    for (int l = 0; l < 64; ++l) {
        for (int c = 0; c < 8; ++c) pb[l][c] = (signed char)((rand() % 40) - 30);
        pb[l][7] = -100;
        prof[2 * l] = (u32)(unsigned char)pb[l][0] | ((u32)(unsigned char)pb[l][1] << 8) | ((u32)(unsigned char)pb[l][2] << 16) | ((u32)(unsigned char)pb[l][3] << 24);
        prof[2 * l + 1] = (u32)(unsigned char)pb[l][4] | ((u32)(unsigned char)pb[l][5] << 8) | ((u32)(unsigned char)pb[l][6] << 16) | ((u32)(unsigned char)pb[l][7] << 24);
    }
    const int ngap = 2, bias = 0x55010000;
    // halo column: non-decreasing G values
    int hv = bias + 50;
    for (size_t i = 0; i < halo.size(); ++i) { hv += rand() % 4; halo[i] = hv; }
    for (int l = 0; l < 64; ++l) { g0[l] = bias + 40 + 2 * l + rand() % 3; tm1[l] = g0[l] - 100; }
    const int z1 = bias + 10 + ngap;   // floor of step 1
    // host restatement
    auto code_at = [&](int l, int u) { return codes[OFF + 1 + u - l - 1 + 0]; };   // codes[OFF + u - l]
    std::vector<std::vector<int>> g(nsteps + 1, std::vector<int>(64)), t(nsteps + 1, std::vector<int>(64));
    std::vector<int> tm(64);
    for (int l = 0; l < 64; ++l) { g[0][l] = g0[l]; tm[l] = tm1[l]; t[0][l] = g0[l] + pb[l][code_at(l, 0)]; }
    for (int u = 1; u <= nsteps; ++u) {
        const int Z = z1 + ngap * (u - 1);
        for (int l = 0; l < 64; ++l) {
            int v;
            if (l == 0) v = halo[u - 1];
            else {
                const int t2 = (u >= 2) ? t[u - 2][l - 1] : tm[l - 1];
                v = std::max(std::max(t2, g[u - 1][l - 1]), std::max(g[u - 1][l], Z));
            }
            g[u][l] = v;
            t[u][l] = v + pb[l][code_at(l, u)];
        }
    }
    unsigned char* d_codes; int *d_halo, *d_g0, *d_tm1, *d_out; u32* d_prof; u64* d_clk;
    hipMalloc(&d_codes, codes.size()); hipMalloc(&d_halo, halo.size() * 4); hipMalloc(&d_g0, 256); hipMalloc(&d_tm1, 256);
    hipMalloc(&d_prof, 512); hipMalloc(&d_out, nsteps * 64 * 4); hipMalloc(&d_clk, 64);
    hipMemcpy(d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_halo, halo.data(), halo.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_g0, g0.data(), 256, hipMemcpyHostToDevice); hipMemcpy(d_tm1, tm1.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(d_prof, prof.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(perm_k<1>, dim3(1), dim3(64), 0, 0, d_codes, d_halo, d_g0, d_tm1, d_prof, ngap, z1, nch_v, d_out, d_clk);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<int> out(nsteps * 64);
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int u = 1; u <= nsteps && bad < 10; ++u)
        for (int l = 0; l < 64; ++l)
            if (out[(u - 1) * 64 + l] != g[u][l]) {
                if (bad < 10) printf("MISMATCH step %d lane %d: got %d want %d\n", u, l, out[(u - 1) * 64 + l] - bias, g[u][l] - bias);
                ++bad;
            }
    printf("verify: %s (%d steps x 64 lanes)\n", bad ? "FAILED" : "ok", nsteps);
    // timing: 16 chunks = 1024 steps (halo array is 1024+64 long)
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(perm_k<0>, dim3(1), dim3(64), 0, 0, d_codes, d_halo, d_g0, d_tm1, d_prof, ngap, z1, 15, d_out, d_clk);
        hipDeviceSynchronize();
        u64 c[2]; hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
        printf("perm producer block (4 VALU/step + perm + ring write + halo read + code load): %.1f clk/step, %.2f ns/step (s_memtime ticks at %.0f MHz)\n",
               (double)c[0] / (15 * 64.0), (double)c[1] * 10.0 / (15 * 64.0), (double)c[0] / ((double)c[1] * 0.01));
    }
    return bad ? 1 : 0;
}
