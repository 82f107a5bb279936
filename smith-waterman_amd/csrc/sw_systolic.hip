// sw_systolic.hip -- the wavefront fill as a systolic producer/consumer pipeline (gfx950).
//
// Measured on MI355X (profiles/r01_ubench_*.log): for a lone wave every instruction of the
// recurrence costs ~4.3 clk whatever its dependences (the bare 3-op recurrence runs at 14 clk per
// step, a ds_write_b32 adds 16 clk, a ds_write_b128 per four steps ~1 clk per step), and a
// cross-workgroup hand-off costs ~0.4 us one way.  So a strip's speed is set by the NUMBER of
// instructions its producer wave issues per anti-diagonal step: the producer runs the recurrence
// and nothing else (6 instructions per step + 2 LDS ops per 4 steps), everything else (H/P
// derivation, arg-max, HBM stores, inter-workgroup traffic) lives in other waves.
//
// G-space (see sw_kernels.hip):  G = H - gap*(row+col);  Z = -gap*(row+col) is the H==0 floor and
// is the SAME for all cells of one anti-diagonal.
//
// Strip s = matrix columns 63*s .. 63*s+63.  Lane l owns column 63*s+l; lane 0 is the strip's
// left halo column (= lane 63 of strip s-1), so every lane finds its left/diagonal neighbour one
// lane down.  At local step u lane l works on row r = u - phi - l (phi = (-s) mod 4, see below):
//     m = max(G1[l-1], G1[l])          v_max_i32_dpp wave_shr:1     (G1 = values of step u-1)
//     d = G2[l-1] + s'                 v_add_u32_dpp wave_shr:1     (G2 = step u-2)
//     g = max3(d, m, Z)                v_max3_i32
// Lane 0 is never written by the two DPP ops (no source lane): m[0] stays "minus infinity" and
// d[0] was loaded with the halo value of that step (a broadcast ds_read_b128 fetches four steps
// of halo straight into the four d registers), so g[0] = halo falls out of the same max3.
//
// LDS ring of a strip, lane-major: lane l, step u at  l*SY_LSTR + ((u-1) mod R)*4  (SY_LSTR =
// 4R+16 keeps b128 accesses of 8 neighbouring lanes on distinct banks).  A producer writes four
// steps with one ds_write_b128; a consumer reads matrix row r of lane l at step r+l+phi, i.e. along
// the skew (conflict-free); the right-hand strip reads lane 63's four values of steps u+63...
// with one ds_read_b128.  phi shifts each strip's step numbering so that this read is 16-byte
// aligned: (63 + phi[s-1] - phi[s]) mod 4 == 0.
//
// Roles in one workgroup (NS strips): NS producer waves, NS*NC consumer waves (derive H and P
// from the ring, row-major coalesced HBM stores, arg-max), an importer and an exporter wave (move the
// edge column between workgroups through HBM/L2 as {tag,value} granules).  Hand-offs inside the workgroup are
// LDS counters written in order behind the data they cover.  Shapes: NS=1 with 8 consumers while the
// strips of the job fit the CUs (the producer then has a SIMD to itself: 20 ns per step), NS=2 with
// 2x4 consumers otherwise (twice the work per CU; VALU-bound at 27 ns per step).
// Both hot loops are single asm statements: producer_fast (the whole strip loop) and
// consumer_block16 (16 matrix rows: H, P, both stores; int32/int64 H, int32/int8 P, write-back or
// streaming stores).  DESIGN.md sections 5 and 6 have the measurements behind every choice.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sw_kernels.h"

namespace swk {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) u32 gu32;

constexpr int SY_R = 256;               // ring entries (steps) per lane
constexpr int SY_LSTR = SY_R * 4 + 16;  // byte stride between lanes in a ring
constexpr int SY_U = 16;                // steps per producer block / rows per consumer block
constexpr int SY_RH = 1024;             // imported-halo ring entries (steps)
constexpr int SY_W = 63;                // real columns per strip
constexpr u32 SY_OOB = 0xFFFFFF00u;
constexpr int SY_ASENT = 0x200;         // never-matching character for lanes without a column

template <int NS>
struct SysLds {
    __attribute__((aligned(16))) unsigned char ring[NS][64 * SY_LSTR];
    __attribute__((aligned(16))) u32 halo[SY_RH];      // edge column imported from the previous workgroup
    __attribute__((aligned(16))) int cons_blk[NS][8];  // latest completed 16-row block per consumer (-1 none; unused INT_MAX)
    int prod_u[NS];   // completed local steps of each producer
    int halo_ready;   // imported halo: every local step (strip s0's numbering) < halo_ready is in halo[]
    int exp_done;     // exported edge rows: every row <= exp_done is in HBM
    int never;        // INT_MAX
    int topv[NS][64]; // band-resident launch: this group's segment of the halo row (H values), polled from top_gran
};

// counters are wave-uniform by construction; readfirstlane makes that provable, so every poll loop
// is a scalar branch
__device__ __forceinline__ int lds_load(const int* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_store(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// bounded spinning: gives up after ~3 s or when another wave raised the abort flag
struct Spin {
    unsigned n = 0;
    uint64_t t0 = 0;
    unsigned code = 1;        // what the abort flag is set to when this wait gives up (diagnostics: 2 consumer, 3 importer)
    __device__ __forceinline__ bool fail(unsigned int* abort_flag) {
        __builtin_amdgcn_s_sleep(1);
        if ((++n & 127u) != 0) return false;
        const uint64_t now = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        if (t0 == 0) t0 = now;
        const bool expired = (now - t0) > 300000000ull;
        if (expired) __hip_atomic_store((gu32*)abort_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return expired || __builtin_amdgcn_readfirstlane((int)__hip_atomic_load((gu32*)abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
    }
};

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// per-strip shift of the step numbering.  generic producer: (-s) mod 4 (halo read 16-byte aligned);
// fast producer: phi_base - s (halo always at left step u+64, a 64-step chunk never wraps in the ring)
__device__ __forceinline__ int phi_of(int s, int phi_base) { return phi_base >= 0 ? phi_base - s : (4 - (s & 3)) & 3; }

// =================================================================================================
// Producer inner loop: ONE asm statement per 16-step block.  Literal registers are used only as
// temporaries inside the statement (hipcc allocates its own values anywhere, also in v100+, so no
// state may sit in a literal register between statements; tools/check_isa.py audits this):
//   v[100:107] the last 8 step results (ds_write_b128 needs them contiguous)
//   v[116:119], v[108:111]  d registers of groups 1/3 and 2 (lane 0 = halo, landed by ds_read_b128)
//   v112 m   v113 s'   v120 counter snapshot
// Crossing statements as ordinary operands: G1, G2 (last two results), Z (floor), h0..h3 (halo of
// the NEXT block's first group, loaded and waited for inside this statement), cnt (left progress).
// Hazards are hand-scheduled (hipcc pads nothing inside asm): v_cmp -> v_cndmask through VCC two
// instructions apart; a DPP source written at least two instructions earlier.
// =================================================================================================
#define SW_STEP(GN, G1, G2, D, C, BYTE)                                                       \
    "v_cmp_eq_u32_sdwa vcc, %[a], %[" C "] src0_sel:DWORD src1_sel:" BYTE "\n\t"             \
    "v_max_i32_dpp v112, " G1 ", " G1 " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
    "v_add_u32 %[Z], %[Z], %[ngap]\n\t"                                                      \
    "v_cndmask_b32 v113, %[xm], %[mm], vcc\n\t"                                              \
    "v_add_u32_dpp " D ", " G2 ", v113 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"            \
    "v_max3_i32 " GN ", " D ", v112, %[Z]\n\t"
// prologue variant: the lane whose row is 0 at this step takes the halo-row value G0
#define SW_STEP_PRO(GN, G1, G2, D, C, BYTE, KK)                                              \
    SW_STEP(GN, G1, G2, D, C, BYTE)                                                          \
    "v_cmp_eq_u32 vcc, %[kb]+" KK ", %[injv]\n\t"                                            \
    "s_nop 1\n\t"                                                                            \
    "v_cndmask_b32 " GN ", " GN ", %[g0v], vcc\n\t"                                          \
    "s_nop 0\n\t"
#define SW_S(GN, G1, G2, D, C, BYTE, KK) SW_STEP(GN, G1, G2, D, C, BYTE)
#define SW_SP(GN, G1, G2, D, C, BYTE, KK) SW_STEP_PRO(GN, G1, G2, D, C, BYTE, KK)
#define SW_GROUP_EVEN(S, C, D0, D1, D2, D3, K0, K1, K2, K3)                                   \
    S("v100", "v107", "v106", D0, C, "BYTE_0", K0) S("v101", "v100", "v107", D1, C, "BYTE_1", K1) \
    S("v102", "v101", "v100", D2, C, "BYTE_2", K2) S("v103", "v102", "v101", D3, C, "BYTE_3", K3)
#define SW_GROUP_ODD(S, C, D0, D1, D2, D3, K0, K1, K2, K3)                                    \
    S("v104", "v103", "v102", D0, C, "BYTE_0", K0) S("v105", "v104", "v103", D1, C, "BYTE_1", K1) \
    S("v106", "v105", "v104", D2, C, "BYTE_2", K2) S("v107", "v106", "v105", D3, C, "BYTE_3", K3)

#define SW_LITERALS "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
                    "v112", "v113", "v116", "v117", "v118", "v119", "v120"

// LDS traffic of a block, in issue order (lgkmcnt counts are exact because the whole block is one
// statement): R1=halo(g1) | W0 | R2=halo(g2) Rc=counter | W1 | R3=halo(g3) Rn0..3=next block's halo |
// W2 | W3 | wait for everything but the two youngest writes.
#define SW_BLOCK(S, KB)                                                                                           \
    asm volatile(                                                                                                 \
        "v_mov_b32 v107, %[G1]\n\t"                                                                               \
        "v_mov_b32 v106, %[G2]\n\t"                                                                               \
        "v_mov_b32 v112, 0xe0000000\n\t"                                                                          \
        "ds_read_b128 v[116:119], %[ra1]\n\t"                                                                     \
        SW_GROUP_EVEN(S, "c0", "%[h0]", "%[h1]", "%[h2]", "%[h3]", "1", "2", "3", "4")                             \
        "ds_write_b128 %[waddr], v[100:103] offset:%[o0]\n\t"                                                     \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                \
        "ds_read_b128 v[108:111], %[ra2]\n\t"                                                                     \
        "ds_read_b32 v120, %[cnt_addr]\n\t"                                                                       \
        SW_GROUP_ODD(S, "c1", "v116", "v117", "v118", "v119", "5", "6", "7", "8")                                  \
        "ds_write_b128 %[waddr], v[104:107] offset:%[o1]\n\t"                                                     \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                \
        "ds_read_b128 v[116:119], %[ra3]\n\t"                                                                     \
        "ds_read_b32 %[h0], %[ran]\n\t"                                                                           \
        "ds_read_b32 %[h1], %[ran] offset:4\n\t"                                                                  \
        "ds_read_b32 %[h2], %[ran] offset:8\n\t"                                                                  \
        "ds_read_b32 %[h3], %[ran] offset:12\n\t"                                                                 \
        SW_GROUP_EVEN(S, "c2", "v108", "v109", "v110", "v111", "9", "10", "11", "12")                              \
        "ds_write_b128 %[waddr], v[100:103] offset:%[o2]\n\t"                                                     \
        "s_waitcnt lgkmcnt(5)\n\t"                                                                                \
        SW_GROUP_ODD(S, "c3", "v116", "v117", "v118", "v119", "13", "14", "15", "16")                              \
        "ds_write_b128 %[waddr], v[104:107] offset:%[o3]\n\t"                                                     \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                \
        "v_readfirstlane_b32 %[cnt], v120\n\t"                                                                    \
        "v_mov_b32 %[G1], v107\n\t"                                                                               \
        "v_mov_b32 %[G2], v106\n\t"                                                                               \
        : [G1] "+v"(G1), [G2] "+v"(G2), [Z] "+v"(Z), [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3),   \
          [cnt] "=s"(cnt)                                                                                         \
        : [a] "v"(a_l), [c0] "v"(C.x), [c1] "v"(C.y), [c2] "v"(C.z), [c3] "v"(C.w), [xm] "v"(xm_v), [mm] "v"(mm_v), \
          [ngap] "v"(ngap_v), [injv] "v"(injv), [g0v] "v"(G0v), [waddr] "v"(waddr), [ra1] "v"(ra1), [ra2] "v"(ra2),  \
          [ra3] "v"(ra3), [ran] "v"(ran), [cnt_addr] "v"(cnt_addr), [kb] "n"(KB), [o0] "n"(KB * 4),                  \
          [o1] "n"(KB * 4 + 16), [o2] "n"(KB * 4 + 32), [o3] "n"(KB * 4 + 48)                                       \
        : "vcc", "memory", SW_LITERALS)


// =================================================================================================
// Fast producer: the WHOLE strip loop is one asm statement (hipcc's code between block statements
// -- structurizer branches, SGPR spills through v_readlane, full lgkmcnt drains -- cost as much as
// the steps themselves).  Used when the matrix has no halo row (top == NULL) and mismatch <= 0:
// then the cells above row 1 need no injection, because with never-matching characters (16-bit
// character stream, 0x100 outside the sequence) the recurrence reproduces the H == 0 floor there
// by itself.  phi = phi_base - s here, so every strip reads its halo at left step u+64 and a
// 64-step chunk never wraps inside the 256-entry ring: all LDS addresses are base + immediate.
//
// literal registers (temporaries of this one statement):
//   v[64:95]  four 8-dword character buffers (16 steps each, 16-bit characters), loaded 3 blocks ahead
//   v96 per-lane character byte offset     v56..v59 LDS addresses of the counters
//   v[100:107] last 8 step results   v[108:111], v[116:119], v[122:125] d registers (lane 0 = halo)
//   v112 m  v113 s'  v114 floor  v120/v121 scratch  v126 ring write address  v127/v115 halo read bases
//   s84 have_halo  s85..s87 scratch  s88 u0  s89 spin count  s90/s91 chunk byte offset in my ring / the halo
//   source ring  s[92:93] char base
// =================================================================================================
#define SF_STEP(GN, G1, G2, D, CREG, WSEL)                                                    \
    "v_cmp_eq_u32_sdwa vcc, %[a], " CREG " src0_sel:DWORD src1_sel:" WSEL "\n\t"             \
    "v_max_i32_dpp v112, " G1 ", " G1 " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"           \
    "v_add_u32 v114, v114, %[ngap]\n\t"                                                      \
    "v_cndmask_b32 v113, %[xm], %[mm], vcc\n\t"                                              \
    "v_add_u32_dpp " D ", " G2 ", v113 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"            \
    "v_max3_i32 " GN ", " D ", v112, v114\n\t"
#define SF_EVEN(CA, CB, D0, D1, D2, D3)                                                      \
    SF_STEP("v100", "v107", "v106", D0, CA, "WORD_0") SF_STEP("v101", "v100", "v107", D1, CA, "WORD_1") \
    SF_STEP("v102", "v101", "v100", D2, CB, "WORD_0") SF_STEP("v103", "v102", "v101", D3, CB, "WORD_1")
#define SF_ODD(CA, CB, D0, D1, D2, D3)                                                       \
    SF_STEP("v104", "v103", "v102", D0, CA, "WORD_0") SF_STEP("v105", "v104", "v103", D1, CA, "WORD_1") \
    SF_STEP("v106", "v105", "v104", D2, CB, "WORD_0") SF_STEP("v107", "v106", "v105", D3, CB, "WORD_1")
// one 16-step block.  KB: first step's index in the chunk; C0..C7: this block's character dwords;
// NB: the buffer (tuples) that receives the characters of the block three ahead; PF: their byte offset;
// RAN: where the next block's first four halo values are read from
#define SF_BLOCK(KB, O0, O1, O2, O3, R1, R2, R3, C0, C1, C2, C3, C4, C5, C6, C7, NBA, NBB, PF0, PF1, RAN, BPF) \
    "s_cmp_eq_u32 s84, 0\n\t"                                                                \
    "s_cbranch_scc1 Lslow" #KB "_%=\n"                                                       \
    "Lgo" #KB "_%=:\n\t"                                                                     \
    "s_waitcnt vmcnt(4)\n\t"                                                                 \
    "global_load_dwordx4 " NBA ", v96, s[92:93] offset:" PF0 "\n\t"                          \
    "global_load_dwordx4 " NBB ", v96, s[92:93] offset:" PF1 "\n\t"                          \
    "ds_read_b128 v[116:119], v127 offset:" R1 "\n\t"                                        \
    SF_EVEN(C0, C1, "v122", "v123", "v124", "v125")                                          \
    "ds_write_b128 v126, v[100:103] offset:" O0 "\n\t"                                       \
    "s_waitcnt lgkmcnt(1)\n\t"                                                               \
    "ds_read_b128 v[108:111], v127 offset:" R2 "\n\t"                                        \
    SF_ODD(C2, C3, "v116", "v117", "v118", "v119")                                           \
    "ds_write_b128 v126, v[104:107] offset:" O1 "\n\t"                                       \
    "s_waitcnt lgkmcnt(1)\n\t"                                                               \
    "ds_read_b128 v[116:119], v127 offset:" R3 "\n\t"                                        \
    SF_EVEN(C4, C5, "v108", "v109", "v110", "v111")                                          \
    "ds_write_b128 v126, v[100:103] offset:" O2 "\n\t"                                       \
    "s_waitcnt lgkmcnt(1)\n\t"                                                               \
    /* as late as the LDS latency allows: the left neighbour's progress, THEN (in order behind it) the next     \
       block's first four halo values -- every step this look-ahead is shorter is a step less lag per strip */  \
    "ds_read_b32 v120, v56\n\t"                                                              \
    "ds_read_b128 v[122:125], " RAN "\n\t"                                                   \
    BPF                                                                                      \
    SF_ODD(C6, C7, "v116", "v117", "v118", "v119")                                           \
    "ds_write_b128 v126, v[104:107] offset:" O3 "\n\t"                                       \
    "s_waitcnt lgkmcnt(1)\n\t"                                                               \
    "v_readfirstlane_b32 s85, v120\n\t"                                                      \
    "s_add_i32 s86, s88, %[k1]\n\t"                                                          \
    "s_min_i32 s86, s86, %[k2]\n\t"                                                          \
    "s_cmp_ge_i32 s85, s86\n\t"                                                              \
    "s_cselect_b32 s84, 1, 0\n\t"                                                            \
    "s_add_i32 s87, s88, 15\n\t"                                                             \
    "v_mov_b32 v121, s87\n\t"                                                                \
    "ds_write_b32 v59, v121\n\t"                                                             \
    "s_add_i32 s88, s88, 16\n\t"                                                             \
    "s_cmp_gt_i32 s88, %[ut]\n\t"                                                            \
    "s_cbranch_scc1 Lexit_%=\n\t"
// Ring back-pressure, every 32 steps: my consumers (four counters) and my right-hand reader (the next producer or
// the exporter) must be done with the slots of the coming 32 steps.  The counters were fetched from LDS during the
// block before (SF_BPFETCH: no LDS round trip on the fast path); only a failed check re-reads them and polls.
#define SF_BPFETCH "ds_read_b128 v[60:63], v57\n\tds_read_b128 v[52:55], v57 offset:16\n\tds_read_b32 v97, v58\n\t"
#define SF_BPCHECK(T)                                                                         \
    "v_min3_i32 v116, v60, v61, v62\n\t"                                                     \
    "v_min3_i32 v117, v63, v52, v53\n\t"                                                     \
    "v_min3_i32 v116, v116, v54, v55\n\t"                                                    \
    "v_min_i32 v116, v116, v117\n\t"                                                         \
    "s_mov_b32 s89, 0\n\t"                                                                   \
    "v_readfirstlane_b32 s86, v97\n\t"                                                       \
    "v_readfirstlane_b32 s85, v116\n"                                                        \
    "Lbpe" T "_%=:\n\t"                                                                      \
    "s_lshl_b32 s85, s85, 4\n\t"                                                             \
    "s_add_i32 s85, s85, 16\n\t"                                                             \
    "s_add_i32 s87, s88, %[kc]\n\t"                                                          \
    "s_cmp_ge_i32 s85, s87\n\t"                                                              \
    "s_cbranch_scc0 Lbpw" T "_%=\n\t"                                                        \
    "s_add_i32 s87, s88, %[kr]\n\t"                                                          \
    "s_cmp_ge_i32 s86, s87\n\t"                                                              \
    "s_cbranch_scc1 Lbpok" T "_%=\n"                                                         \
    "Lbpw" T "_%=:\n\t"                                                                      \
    "s_sleep 1\n\t"                                                                          \
    "s_add_i32 s89, s89, 1\n\t"                                                              \
    "s_add_i32 s94, s94, 1\n\t"                                                              \
    "s_cmp_lt_u32 s89, 0x1000000\n\t"                                                        \
    "s_cbranch_scc0 Lbpfail_%=\n\t"                                                          \
    "ds_read_b128 v[116:119], v57\n\t"                                                       \
    "ds_read_b128 v[52:55], v57 offset:16\n\t"                                               \
    "ds_read_b32 v120, v58\n\t"                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "v_min3_i32 v116, v116, v117, v118\n\t"                                                  \
    "v_min3_i32 v117, v119, v52, v53\n\t"                                                    \
    "v_min3_i32 v116, v116, v54, v55\n\t"                                                    \
    "v_min_i32 v116, v116, v117\n\t"                                                         \
    "s_nop 0\n\t"                                                                            \
    "v_readfirstlane_b32 s85, v116\n\t"                                                      \
    "v_readfirstlane_b32 s86, v120\n\t"                                                      \
    "s_branch Lbpe" T "_%=\n"                                                                \
    "Lbpok" T "_%=:\n\t"
// out-of-line: wait until the left neighbour has produced this block's halo, then fetch its first group
#define SF_SLOW(KB, R0)                                                                       \
    "Lslow" #KB "_%=:\n\t"                                                                   \
    "s_mov_b32 s89, 0\n"                                                                     \
    "Lpoll" #KB "_%=:\n\t"                                                                   \
    "ds_read_b32 v120, v56\n\t"                                                              \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "v_readfirstlane_b32 s85, v120\n\t"                                                      \
    "s_add_i32 s86, s88, %[k1]\n\t"                                                          \
    "s_add_i32 s86, s86, -16\n\t"                                                            \
    "s_min_i32 s86, s86, %[k2]\n\t"                                                          \
    "s_cmp_ge_i32 s85, s86\n\t"                                                              \
    "s_cbranch_scc1 Lrd" #KB "_%=\n\t"                                                       \
    "s_sleep 1\n\t"                                                                          \
    "s_add_i32 s89, s89, 1\n\t"                                                              \
    "s_add_i32 s95, s95, 1\n\t"                                                              \
    "s_cmp_lt_u32 s89, 0x1000000\n\t"                                                        \
    "s_cbranch_scc1 Lpoll" #KB "_%=\n\t"                                                     \
    "s_mov_b32 %[status], 1\n\t"                                                             \
    "s_branch Lexit_%=\n"                                                                    \
    "Lrd" #KB "_%=:\n\t"                                                                     \
    "ds_read_b128 v[122:125], v127 offset:" R0 "\n\t"                                        \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "s_branch Lgo" #KB "_%=\n"

__device__ __forceinline__ int producer_fast(u32 a_l, u32 xm_v, u32 mm_v, u32 ngap_v, u32 wbase, u32 voff, u32 z0,
                                             const unsigned short* cbase, u32 hbase, int hoff4, u32 cnt_addr,
                                             u32 cons_addr, u32 right_addr, u32 prog_addr, int UT, int k1, int k2, int kc, int kr, int hmask,
                                             int (&polls)[2]) {
    int status;
    asm volatile(
        "s_setprio 3\n\t"                          /* producers own the critical path: win VALU arbitration on their SIMD */
        "s_mov_b32 %[status], 0\n\t"
        "s_mov_b32 s84, 0\n\t"
        "s_mov_b32 s94, 0\n\t"
        "s_mov_b32 s95, 0\n\t"
        "s_mov_b32 s88, 1\n\t"
        "s_mov_b32 s90, 0\n\t"
        "s_and_b32 s91, %[hoff4], %[hmask]\n\t"   /* byte offset of this chunk's halo inside the halo source ring */
        "s_mov_b64 s[92:93], %[cbase]\n\t"
        "v_mov_b32 v96, %[voff]\n\t"
        "v_mov_b32 v56, %[cntaddr]\n\t"
        "v_mov_b32 v57, %[consaddr]\n\t"
        "v_mov_b32 v58, %[rightaddr]\n\t"
        "v_mov_b32 v59, %[progaddr]\n\t"
        "v_mov_b32 v107, %[z0]\n\t"               /* step 0: every lane's cell is above the matrix: the H == 0 floor */
        "v_sub_u32 v106, %[z0], %[ngap]\n\t"      /* step -1 */
        "v_mov_b32 v112, 0xe0000000\n\t"
        "v_mov_b32 v114, %[z0]\n\t"
        "v_mov_b32 v60, 0\n\t"                   /* back-pressure counters as seen "before the start": nothing to wait for */
        "v_mov_b32 v61, 0\n\t"
        "v_mov_b32 v62, 0\n\t"
        "v_mov_b32 v63, 0\n\t"
        "v_mov_b32 v52, 0\n\t"
        "v_mov_b32 v53, 0\n\t"
        "v_mov_b32 v54, 0\n\t"
        "v_mov_b32 v55, 0\n\t"
        "v_mov_b32 v97, 0\n\t"
        "s_nop 4\n\t"
        "global_load_dwordx4 v[64:67], v96, s[92:93] offset:0\n\t"
        "global_load_dwordx4 v[68:71], v96, s[92:93] offset:16\n\t"
        "global_load_dwordx4 v[72:75], v96, s[92:93] offset:32\n\t"
        "global_load_dwordx4 v[76:79], v96, s[92:93] offset:48\n\t"
        "global_load_dwordx4 v[80:83], v96, s[92:93] offset:64\n\t"
        "global_load_dwordx4 v[84:87], v96, s[92:93] offset:80\n"
        "Lchunk_%=:\n\t"
        SF_BPCHECK("A")
        // ---- this chunk's LDS addresses: ring write base, halo read base, next chunk's halo read base
        "v_add_u32 v126, s90, %[wbase]\n\t"
        "v_mov_b32 v127, %[hbase]\n\t"
        "v_add_u32 v127, s91, v127\n\t"
        "s_add_i32 s91, s91, 256\n\t"
        "s_and_b32 s91, s91, %[hmask]\n\t"
        "v_mov_b32 v115, %[hbase]\n\t"
        "v_add_u32 v115, s91, v115\n\t"
        SF_BLOCK(0, "0", "16", "32", "48", "16", "32", "48", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
                 "v[88:91]", "v[92:95]", "96", "112", "v127 offset:64", "")
        SF_BLOCK(16, "64", "80", "96", "112", "80", "96", "112", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
                 "v[64:67]", "v[68:71]", "128", "144", "v127 offset:128", SF_BPFETCH)
        SF_BPCHECK("B")
        SF_BLOCK(32, "128", "144", "160", "176", "144", "160", "176", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87",
                 "v[72:75]", "v[76:79]", "160", "176", "v127 offset:192", "")
        SF_BLOCK(48, "192", "208", "224", "240", "208", "224", "240", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
                 "v[80:83]", "v[84:87]", "192", "208", "v115", SF_BPFETCH)
        "s_add_i32 s90, s90, 256\n\t"
        "s_and_b32 s90, s90, 1023\n\t"
        "s_add_u32 s92, s92, 128\n\t"
        "s_addc_u32 s93, s93, 0\n\t"
        "s_branch Lchunk_%=\n"
        SF_SLOW(0, "0") SF_SLOW(16, "64") SF_SLOW(32, "128") SF_SLOW(48, "192")
        "Lbpfail_%=:\n\t"
        "s_mov_b32 %[status], 2\n"
        "Lexit_%=:\n\t"
        "s_setprio 0\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "s_mov_b32 %[nbp], s94\n\t"
        "s_mov_b32 %[nhalo], s95\n\t"
        : [status] "=&s"(status), [nbp] "=&s"(polls[0]), [nhalo] "=&s"(polls[1])
        : [a] "v"(a_l), [xm] "v"(xm_v), [mm] "v"(mm_v), [ngap] "v"(ngap_v), [wbase] "v"(wbase), [voff] "v"(voff), [z0] "v"(z0),
          [cbase] "s"(cbase), [hbase] "s"(hbase), [hoff4] "s"(hoff4), [cntaddr] "s"(cnt_addr), [consaddr] "s"(cons_addr),
          [rightaddr] "s"(right_addr), [progaddr] "s"(prog_addr), [ut] "s"(UT), [k1] "s"(k1), [k2] "s"(k2), [kc] "s"(kc), [kr] "s"(kr), [hmask] "s"(hmask)
        : "vcc", "scc", "memory", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95",
          "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v97", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",
          "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91",
          "v92", "v93", "v94", "v95", "v96", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
          "v124", "v125", "v126", "v127");
    return status;
}


// =================================================================================================
// "Perm" producer (round 2): FOUR VALU instructions per anti-diagonal step instead of six.
//     A: P      = max3(t[u-2], g[u-1], Z)      v_max3_i32        t = g + s' of the cell diagonally below-right
//     B: Z     += ngap                          v_add_u32
//     C: t[u-1] = g[u-1] + sext(S.byte)         v_add_u32_sdwa    S: this block's 16 score bytes
//     D: g[u]   = max(P[l-1], g[u-1][l])        v_max_i32_dpp wave_shr:1   (lane 0 is not written: it keeps the halo value
//                                                                   a broadcast ds_read_b128 put there 8 steps earlier)
// i.e. g[l] = max(t2[l-1], g1[l-1], g1[l], Z): the shifted maximum is formed BEFORE the shift, so one DPP op does both
// lane moves.  The scores are not computed per step: letters are coded 0..6 (7 = outside the sequences) by sw_pad_b,
// every lane holds its column's profile (8 score bytes, one per code) in two VGPRs, and ONE v_perm_b32 with four code
// bytes of b as the selector yields four steps of scores -- 4 v_perm + 1 code load per 16 steps.  Cells outside the
// sequences score SW_PERM_PAD: above the matrix every lane then simply keeps its row-0 value (G is non-decreasing along
// row 0), so a halo row needs no injection and the mismatch score may be positive.
// All G values carry p.gbias (launch tag << 24 | 2^16; G-space is translation invariant), which makes every exported
// value self-validating: lane 63 stores its own four results per group with one buffer_store_dwordx4 (the other lanes'
// offsets are out of range) -- no exporter wave, no progress poll, no ring read on the way out.
// Checked step by step against a host restatement and timed in tools/ubench_perm.hip (28.5 clk per step alone).
//
// literal registers (temporaries of this one statement):
//   v[100:115] g of the 16 steps of a block (lane 0: halo)    v[116:119] t    v120 P    v121 Z
//   v[122:125] / v[60:63] score bytes of even / odd blocks     v[64:79] four code buffers (loaded 3 blocks ahead)
//   v[80:83], v84 back-pressure counters   v86 left progress   v87, v92 scratch   v88..v91 LDS addresses of the counters
//   v96 code offset   v97 export offset (lane 63: 0, others out of range)   v98/v99 halo read base of this/next chunk
//   v126 ring write address
//   s84 have_halo  s85..s87 scratch  s88 u0  s89 spin  s90/s91 chunk byte offset in my ring / the halo ring
//   s[92:93] code base  s94/s95 slow-path poll counts  s[76:79] export buffer  s75 export chunk offset
// =================================================================================================
#define PP_STEP(GK, GP, TP2, TP1, SREG, BYTE)                                                     \
    "v_max3_i32 v120, " TP2 ", " GP ", v121\n\t"                                                  \
    "v_add_u32 v121, v121, %[ngap]\n\t"                                                           \
    "v_add_u32_sdwa " TP1 ", " GP ", sext(" SREG ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" BYTE "\n\t" \
    "v_max_i32_dpp " GK ", v120, " GP " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
// a group = 4 steps; xA is its first step (after which the registers of the group before it are dead), xB the other three
#define PP_G0A(SP3) PP_STEP("v100", "v115", "v118", "v119", SP3, "BYTE_3")
#define PP_G0B(S0) PP_STEP("v101", "v100", "v119", "v116", S0, "BYTE_0") PP_STEP("v102", "v101", "v116", "v117", S0, "BYTE_1") PP_STEP("v103", "v102", "v117", "v118", S0, "BYTE_2")
#define PP_G1A(S0) PP_STEP("v104", "v103", "v118", "v119", S0, "BYTE_3")
#define PP_G1B(S1) PP_STEP("v105", "v104", "v119", "v116", S1, "BYTE_0") PP_STEP("v106", "v105", "v116", "v117", S1, "BYTE_1") PP_STEP("v107", "v106", "v117", "v118", S1, "BYTE_2")
#define PP_G2A(S1) PP_STEP("v108", "v107", "v118", "v119", S1, "BYTE_3")
#define PP_G2B(S2) PP_STEP("v109", "v108", "v119", "v116", S2, "BYTE_0") PP_STEP("v110", "v109", "v116", "v117", S2, "BYTE_1") PP_STEP("v111", "v110", "v117", "v118", S2, "BYTE_2")
#define PP_G3A(S2) PP_STEP("v112", "v111", "v118", "v119", S2, "BYTE_3")
#define PP_G3B(S3) PP_STEP("v113", "v112", "v119", "v116", S3, "BYTE_0") PP_STEP("v114", "v113", "v116", "v117", S3, "BYTE_1") PP_STEP("v115", "v114", "v117", "v118", S3, "BYTE_2")
#ifdef PP_NO_EXPORT   /* tools/ubench_prod.hip: what does each part of the block cost? */
#define PP_KX(X) ""
#else
#define PP_KX(X) X
#endif
// One 16-step block.  KB: first step's index in the chunk.  SP3: score dword holding the step before the block;
// S0..S3 / C0..C3: this block's score and code dwords; NB, PF: code buffer that receives the block three ahead and its
// byte offset; O0..O3: ring / export byte offsets of the four groups; H3: halo offset of this block's last group; HN0..HN2:
// operands of the NEXT block's first three halo groups; LA, LB: lgkmcnt of the two waits (they differ where back-pressure
// fetches sit in the LDS queue); BPF: those fetches.
// Every halo group is fetched 11 steps before its first use, right after the step that last read its registers, so no
// wait ever stalls (measured: fetching 4-5 steps ahead cost ~20 clk per wait, 5 clk per step).  The left neighbour's
// progress is sampled (Rc) BEFORE the next block's groups are fetched and evaluated at the next block's start: if it
// covered that whole block then, what was fetched is valid; else the slow path polls and fetches again.
// LDS order: [k0] R3 [k1-3] W0 [k4] Rc R0' [k5-7] W1 [k8] R1' [k9-11] W2 (BPF) [k12] R2' [k13-15] W3 Wp
/* tools/ubench_prod.hip: knock single parts of the block out to see what each costs (never in the product build) */
#ifdef PP_NO_CHK
#define PP_K_CHK(X) ""
#else
#define PP_K_CHK(X) X
#endif
#ifdef PP_NO_CODE
#define PP_K_CODE(X) ""
#else
#define PP_K_CODE(X) X
#endif
#ifdef PP_NO_RING
#define PP_K_RING(X) ""
#else
#define PP_K_RING(X) X
#endif
#ifdef PP_NO_PROG
#define PP_K_PROG(X) ""
#else
#define PP_K_PROG(X) X
#endif
#ifdef PP_NO_BP
#define PP_K_BP(X) ""
#else
#define PP_K_BP(X) X
#endif
#ifdef PP_LATE_R   /* experiment: fetch the halo groups three steps later */
#define PP_RA(X) ""
#define PP_RB(X) X
#else
#define PP_RA(X) X
#define PP_RB(X) ""
#endif
// The halo check comes in two forms (V):
//   V = 0  progress: the left neighbour's progress counter, sampled (Rc) before the next block's groups are fetched, must
//          have covered that whole block -- a look-ahead of 28 steps.  For a left neighbour in the same workgroup.
//   V = 1  validate on use: the importer keeps every halo slot it has not filled yet at 0 (no G is 0: they all carry the
//          launch bias), so a fetched group says by itself whether it was there; groups are checked in pairs (the last
//          value of each -- the importer fills in step order) right before their first use: a look-ahead of 12 steps.
#define PP_CHKA_0(KB)                                                                         \
    "v_readfirstlane_b32 s85, v86\n\t"                                                        \
    "s_add_i32 s83, s83, 16\n\t"                                                              \
    "s_min_i32 s86, s83, %[k2]\n\t"                                                           \
    "s_cmp_ge_i32 s85, s86\n\t"                                                               \
    "s_cbranch_scc0 Lslow" #KB "_%=\n"
#define PP_CHKA_1(KB)     /* both there <=> the AND keeps the tag byte: s_and sets SCC = (result != 0).  (No VALU in it: a */  \
    "v_readfirstlane_b32 s85, v103\n\t"      /* VGPR written by a VALU op cannot be read by v_readfirstlane in the next slot) */     \
    "v_readfirstlane_b32 s86, v107\n\t"                                                       \
    "s_and_b32 s85, s85, s86\n\t"                                                             \
    "s_cbranch_scc0 Lslow" #KB "_%=\n"
#define PP_CHKB_0(KB) ""
#define PP_CHKB_1(KB)                                                                         \
    "v_readfirstlane_b32 s85, v111\n\t"                                                       \
    "v_readfirstlane_b32 s86, v115\n\t"                                                       \
    "s_and_b32 s85, s85, s86\n\t"                                                             \
    "s_cbranch_scc0 LslowB" #KB "_%=\n"                                                        \
    "LgoB" #KB "_%=:\n\t"
#define PP_RC_0 "ds_read_b32 v86, v88\n\t"
#define PP_RC_1 ""
#define PP_LB_0 "4"
#define PP_LB_1 "3"      /* no Rc in the queue */
#define PP_BLOCK_(V, KB, SP3, S0, S1, S2, S3, C0, C1, C2, C3, NB, PF, O0, O1, O2, O3, H3, HN0, HN1, HN2, LA, BPF)  \
    "s_waitcnt lgkmcnt(" LA ")\n\t"                   /* (Rc and) this block's groups 0, 1 */   \
    PP_K_CHK(PP_CHKA_##V(KB))                                                                \
    "Lgo" #KB "_%=:\n\t"                                                                     \
    PP_K_CODE("s_waitcnt vmcnt(14)\n\t"                                                      \
    "global_load_dwordx4 " NB ", v96, s[92:93] offset:" PF "\n\t"                            \
    "v_perm_b32 " S0 ", %[phi], %[plo], " C0 "\n\t"                                          \
    "v_perm_b32 " S1 ", %[phi], %[plo], " C1 "\n\t"                                          \
    "v_perm_b32 " S2 ", %[phi], %[plo], " C2 "\n\t"                                          \
    "v_perm_b32 " S3 ", %[phi], %[plo], " C3 "\n\t")                                         \
    PP_G0A(SP3)                                                                              \
    PP_RA("ds_read_b128 v[112:115], v98 offset:" H3 "\n\t")                                  \
    PP_G0B(S0)                                                                               \
    PP_RB("ds_read_b128 v[112:115], v98 offset:" H3 "\n\t")                                  \
    PP_K_RING("ds_write_b128 v126, v[100:103] offset:" O0 "\n\t")                            \
    PP_KX("buffer_store_dwordx4 v[100:103], v97, s[76:79], s75 offen offset:" O0 " sc1\n\t") \
    PP_G1A(S0)                                                                               \
    PP_K_CHK(PP_RC_##V)                                                                      \
    PP_RA("ds_read_b128 v[100:103], " HN0 "\n\t")                                            \
    PP_G1B(S1)                                                                               \
    PP_RB("ds_read_b128 v[100:103], " HN0 "\n\t")                                            \
    PP_K_RING("ds_write_b128 v126, v[104:107] offset:" O1 "\n\t")                            \
    PP_KX("buffer_store_dwordx4 v[104:107], v97, s[76:79], s75 offen offset:" O1 " sc1\n\t") \
    "s_waitcnt lgkmcnt(" PP_LB_##V ")\n\t"            /* this block's groups 2, 3 */            \
    PP_K_CHK(PP_CHKB_##V(KB))                                                                \
    PP_G2A(S1)                                                                               \
    PP_RA("ds_read_b128 v[104:107], " HN1 "\n\t")                                            \
    PP_G2B(S2)                                                                               \
    PP_RB("ds_read_b128 v[104:107], " HN1 "\n\t")                                            \
    PP_K_RING("ds_write_b128 v126, v[108:111] offset:" O2 "\n\t")                            \
    PP_KX("buffer_store_dwordx4 v[108:111], v97, s[76:79], s75 offen offset:" O2 " sc1\n\t") \
    PP_K_BP(BPF)                                                                             \
    PP_G3A(S2)                                                                               \
    PP_RA("ds_read_b128 v[108:111], " HN2 "\n\t")                                            \
    PP_G3B(S3)                                                                               \
    PP_RB("ds_read_b128 v[108:111], " HN2 "\n\t")                                            \
    PP_K_RING("ds_write_b128 v126, v[112:115] offset:" O3 "\n\t")                            \
    PP_KX("buffer_store_dwordx4 v[112:115], v97, s[76:79], s75 offen offset:" O3 " sc1\n\t") \
    PP_K_PROG("v_add_u32 v87, 16, v87\n\t"                                                   \
    "ds_write_b32 v91, v87\n\t")
#define PP_BPFETCH "ds_read_b128 v[80:83], v89\n\tds_read_b128 v[52:55], v89 offset:16\n\tds_read_b32 v84, v90\n\t"
// ring back-pressure before the 32 steps that start at local step s88 + D (D = 0 / 32: operands kc/kr, kc32/kr32)
#define PP_BPCHECK(T, KC, KR)                                                                 \
    "s_waitcnt lgkmcnt(3)\n\t"                        /* the fetches sit in front of R2', W3, Wp */ \
    "v_min3_i32 v92, v80, v81, v82\n\t"                                                      \
    "v_min3_i32 v93, v83, v52, v53\n\t"                                                      \
    "v_min3_i32 v92, v92, v54, v55\n\t"                                                      \
    "v_min_i32 v92, v92, v93\n\t"                                                            \
    "s_mov_b32 s89, 0\n\t"                                                                   \
    "v_readfirstlane_b32 s86, v84\n\t"                                                       \
    "v_readfirstlane_b32 s85, v92\n"                                                         \
    "Lbpe" T "_%=:\n\t"                                                                      \
    "s_lshl_b32 s85, s85, 4\n\t"                                                             \
    "s_add_i32 s85, s85, 16\n\t"                                                             \
    "s_add_i32 s87, s88, " KC "\n\t"                                                         \
    "s_cmp_ge_i32 s85, s87\n\t"                                                              \
    "s_cbranch_scc0 Lbpw" T "_%=\n\t"                                                        \
    "s_add_i32 s87, s88, " KR "\n\t"                                                         \
    "s_cmp_ge_i32 s86, s87\n\t"                                                              \
    "s_cbranch_scc1 Lbpok" T "_%=\n"                                                         \
    "Lbpw" T "_%=:\n\t"                                                                      \
    "s_sleep 1\n\t"                                                                          \
    "s_add_i32 s89, s89, 1\n\t"                                                              \
    "s_add_i32 s94, s94, 1\n\t"                                                              \
    "s_cmp_lt_u32 s89, 0x1000000\n\t"                                                        \
    "s_cbranch_scc0 Lbpfail_%=\n\t"                                                          \
    "ds_read_b128 v[80:83], v89\n\t"                                                         \
    "ds_read_b128 v[52:55], v89 offset:16\n\t"                                               \
    "ds_read_b32 v84, v90\n\t"                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "v_min3_i32 v92, v80, v81, v82\n\t"                                                      \
    "v_min3_i32 v93, v83, v52, v53\n\t"                                                      \
    "v_min3_i32 v92, v92, v54, v55\n\t"                                                      \
    "v_min_i32 v92, v92, v93\n\t"                                                            \
    "s_nop 0\n\t"                                                                            \
    "v_readfirstlane_b32 s85, v92\n\t"                                                       \
    "v_readfirstlane_b32 s86, v84\n\t"                                                       \
    "s_branch Lbpe" T "_%=\n"                                                                \
    "Lbpok" T "_%=:\n\t"
// out-of-line: wait until the left neighbour has produced this block's halo, then fetch its first three groups
#define PP_SLOW_0(KB, R0, R1, R2, R3)                                                         \
    "Lslow" #KB "_%=:\n\t"                                                                   \
    "s_mov_b32 s89, 0\n"                                                                     \
    "Lpoll" #KB "_%=:\n\t"                                                                   \
    "ds_read_b32 v86, v88\n\t"                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "v_readfirstlane_b32 s85, v86\n\t"                                                       \
    "s_nop 3\n\t"                                                                            \
    "s_cmp_ge_i32 s85, s86\n\t"                                                              \
    "s_cbranch_scc1 Lrd" #KB "_%=\n\t"                                                       \
    "s_sleep 1\n\t"                                                                          \
    "s_add_i32 s89, s89, 1\n\t"                                                              \
    "s_add_i32 s95, s95, 1\n\t"                                                              \
    "s_cmp_lt_u32 s89, 0x1000000\n\t"                                                        \
    "s_cbranch_scc1 Lpoll" #KB "_%=\n\t"                                                     \
    "s_mov_b32 %[status], 1\n\t"                                                             \
    "s_branch Lexit_%=\n"                                                                    \
    "Lrd" #KB "_%=:\n\t"                                                                     \
    "ds_read_b128 v[100:103], v98 offset:" R0 "\n\t"                                         \
    "ds_read_b128 v[104:107], v98 offset:" R1 "\n\t"                                         \
    "ds_read_b128 v[108:111], v98 offset:" R2 "\n\t"                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "s_branch Lgo" #KB "_%=\n"
// validate on use: fetch the pair of groups again until both are there.  (Group 2 is fetched along with 0 and 1: the
// first block of a strip has no block before it that would have done so.)
#define PP_SLOW_1(KB, R0, R1, R2, R3)                                                         \
    "Lslow" #KB "_%=:\n\t"                                                                   \
    "s_mov_b32 s89, 0\n"                                                                     \
    "Lpoll" #KB "_%=:\n\t"                                                                   \
    "ds_read_b128 v[100:103], v98 offset:" R0 "\n\t"                                         \
    "ds_read_b128 v[104:107], v98 offset:" R1 "\n\t"                                         \
    "ds_read_b128 v[108:111], v98 offset:" R2 "\n\t"                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "v_readfirstlane_b32 s85, v103\n\t"                                                      \
    "v_readfirstlane_b32 s86, v107\n\t"                                                      \
    "s_and_b32 s85, s85, s86\n\t"                                                            \
    "s_cbranch_scc1 Lgo" #KB "_%=\n\t"                                                       \
    "s_add_i32 s89, s89, 1\n\t"                                                              \
    "s_add_i32 s95, s95, 1\n\t"                                                              \
    "s_cmp_lt_u32 s89, 0x1000000\n\t"                                                        \
    "s_cbranch_scc1 Lpoll" #KB "_%=\n\t"                                                     \
    "s_mov_b32 %[status], 1\n\t"                                                             \
    "s_branch Lexit_%=\n"                                                                    \
    "LslowB" #KB "_%=:\n\t"                                                                  \
    "s_mov_b32 s89, 0\n"                                                                     \
    "LpollB" #KB "_%=:\n\t"                                                                  \
    "ds_read_b128 v[108:111], v98 offset:" R2 "\n\t"                                         \
    "ds_read_b128 v[112:115], v98 offset:" R3 "\n\t"                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                               \
    "v_readfirstlane_b32 s85, v111\n\t"                                                      \
    "v_readfirstlane_b32 s86, v115\n\t"                                                      \
    "s_and_b32 s85, s85, s86\n\t"                                                            \
    "s_cbranch_scc1 LgoB" #KB "_%=\n\t"                                                      \
    "s_add_i32 s89, s89, 1\n\t"                                                              \
    "s_add_i32 s95, s95, 1\n\t"                                                              \
    "s_cmp_lt_u32 s89, 0x1000000\n\t"                                                        \
    "s_cbranch_scc1 LpollB" #KB "_%=\n\t"                                                    \
    "s_mov_b32 %[status], 1\n\t"                                                             \
    "s_branch Lexit_%=\n"

typedef int sw_i32x4p __attribute__((ext_vector_type(4)));
#define PP_BLOCK(V, ...) PP_BLOCK_(V, __VA_ARGS__)
#define PP_SLOW_(V, ...) PP_SLOW_##V(__VA_ARGS__)
#define PP_SLOW(V, ...) PP_SLOW_(V, __VA_ARGS__)
#define PPV 0
#define PP_FN producer_perm_progress
#include "sw_perm_producer.inc"
#undef PPV
#undef PP_FN
#define PPV 1
#define PP_FN producer_perm_valid
#include "sw_perm_producer.inc"
#undef PPV
#undef PP_FN


// =================================================================================================
// Consumer: four matrix rows per asm statement, hand-scheduled (literal registers are temporaries
// of the statement only).  Per row: s' (cmp+cndmask), diagonal candidate (DPP-fused add), z, H,
// three compares + three selects for P (serial_smithW.c:204-234: first of DIAGONAL, UP, LEFT that
// attains a positive maximum) and one v_max for the arg-max.  Compares are batched so that every
// VALU-written mask is read at least two instructions later (gfx940 hazard), i.e. no s_nop at all.
//   v[100:103] s'   v[104:107] diagonal candidates   v[116:119] z of the four rows
//   s[60:61]..s[74:75] compare masks, s76..s79 the four row characters
// =================================================================================================
#define SC_PRED(GI, UPI, DDI, M1, M3)                                                         \
    "v_cmp_eq_u32_e64 " M1 ", " UPI ", " GI "\n\t"                                           \
    "v_cmp_eq_u32_e64 " M3 ", " DDI ", " GI "\n\t"
#define SC_SEL(GI, ZI, PI, M1, M3)                                                            \
    "v_cmp_eq_u32_e32 vcc, " GI ", " ZI "\n\t"                                               \
    "v_cndmask_b32_e64 " PI ", 2, 1, " M1 "\n\t"                                             \
    "v_cndmask_b32_e64 " PI ", " PI ", 3, " M3 "\n\t"                                        \
    "v_cndmask_b32_e64 " PI ", " PI ", 0, vcc\n\t"
struct Rows4 { u32 h0, h1, h2, h3, p0, p1, p2, p3; };
__device__ __forceinline__ Rows4 consumer_rows4(u32 U, u32 G0, u32 G1, u32 G2, u32 G3, u32& z, u32& blkmax, u32 a_l, u32 mm_v,
                                                u32 xm_v, u32 ngap_v, u32 bw) {
    Rows4 o;  // NB: no constant operands here: hipcc may give an input the same register as an in/out operand of equal value
    asm volatile(
        "s_bfe_u32 s76, %[bw], 0x80000\n\t"
        "s_bfe_u32 s77, %[bw], 0x80008\n\t"
        "s_bfe_u32 s78, %[bw], 0x80010\n\t"
        "s_lshr_b32 s79, %[bw], 24\n\t"
        "v_cmp_eq_u32_e64 s[60:61], s76, %[a]\n\t"
        "v_cmp_eq_u32_e64 s[62:63], s77, %[a]\n\t"
        "v_cmp_eq_u32_e64 s[64:65], s78, %[a]\n\t"
        "v_cmp_eq_u32_e64 s[66:67], s79, %[a]\n\t"
        "v_add_u32 v116, %[z], %[ngap]\n\t"
        "v_cndmask_b32_e64 v100, %[xm], %[mm], s[60:61]\n\t"
        "v_cndmask_b32_e64 v101, %[xm], %[mm], s[62:63]\n\t"
        "v_cndmask_b32_e64 v102, %[xm], %[mm], s[64:65]\n\t"
        "v_cndmask_b32_e64 v103, %[xm], %[mm], s[66:67]\n\t"
        "v_add_u32 v117, v116, %[ngap]\n\t"
        "v_add_u32_dpp v104, %[U], v100 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_u32_dpp v105, %[G0], v101 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_u32_dpp v106, %[G1], v102 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_u32_dpp v107, %[G2], v103 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "v_add_u32 v118, v117, %[ngap]\n\t"
        "v_add_u32 v119, v118, %[ngap]\n\t"
        SC_PRED("%[G0]", "%[U]", "v104", "s[60:61]", "s[62:63]")
        SC_PRED("%[G1]", "%[G0]", "v105", "s[64:65]", "s[66:67]")
        "v_sub_u32 %[h0], %[G0], v116\n\t"
        SC_SEL("%[G0]", "v116", "%[p0]", "s[60:61]", "s[62:63]")
        SC_PRED("%[G2]", "%[G1]", "v106", "s[68:69]", "s[70:71]")
        "v_sub_u32 %[h1], %[G1], v117\n\t"
        SC_SEL("%[G1]", "v117", "%[p1]", "s[64:65]", "s[66:67]")
        SC_PRED("%[G3]", "%[G2]", "v107", "s[72:73]", "s[74:75]")
        "v_sub_u32 %[h2], %[G2], v118\n\t"
        SC_SEL("%[G2]", "v118", "%[p2]", "s[68:69]", "s[70:71]")
        "v_sub_u32 %[h3], %[G3], v119\n\t"
        "v_max_i32 %[bm], %[bm], %[h0]\n\t"
        SC_SEL("%[G3]", "v119", "%[p3]", "s[72:73]", "s[74:75]")
        "v_max3_i32 %[bm], %[bm], %[h1], %[h2]\n\t"
        "v_max_i32 %[bm], %[bm], %[h3]\n\t"
        "v_mov_b32 %[z], v119\n\t"
        : [z] "+v"(z), [bm] "+v"(blkmax), [h0] "=&v"(o.h0), [h1] "=&v"(o.h1), [h2] "=&v"(o.h2), [h3] "=&v"(o.h3),
          [p0] "=&v"(o.p0), [p1] "=&v"(o.p1), [p2] "=&v"(o.p2), [p3] "=&v"(o.p3)
        : [U] "v"(U), [G0] "v"(G0), [G1] "v"(G1), [G2] "v"(G2), [G3] "v"(G3), [a] "v"(a_l), [mm] "v"(mm_v), [xm] "v"(xm_v),
          [ngap] "v"(ngap_v), [bw] "s"(bw)
        : "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
          "s74", "s75", "s76", "s77", "s78", "s79", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",
          "v116", "v117", "v118", "v119");
    return o;
}

// =================================================================================================
// Consumer, whole 16-row block in ONE asm statement (int32 H, the common case).  hipcc's own code
// around four consumer_rows4 statements and 32 buffer stores came to ~650 instructions per block
// (SGPR spills through v_readlane, per-row branches, and an s_waitcnt vmcnt(0) for the row
// characters that also waited for every outstanding H/P store); this is ~270.
//   phase 1: z, H = g - z, store H (needs no characters: hides the scalar load of the 16 row characters)
//   phase 2: per row s', diagonal candidate, P (serial_smithW.c:204-234), store P
// literal temporaries: v[100:115] H of the 16 rows, v[116:119] s', v[120:123] diagonal candidates,
// v[124:127] P; s[60:75] compare masks, s[76:79] the row characters, s80 row byte offset
// =================================================================================================
// A/B hooks: -DCB_FORCE_H='" nt"' / -DCB_FORCE_P='""' pin the cache policy of the H or P stores
#ifdef CB_FORCE_H
#define CB_POLH(POL) CB_FORCE_H
#else
#define CB_POLH(POL) POL
#endif
#ifdef CB_FORCE_P
#define CB_POLP(POL) CB_FORCE_P
#else
#define CB_POLP(POL) POL
#endif
#ifdef CB_NOSTORE
#define CB_ST(X) ""
#else
#define CB_ST(X) X
#endif
// (the x-macros take complete operand strings, so that the same text serves the one-block statement -- ring values
//  and buffer descriptors as compiler operands -- and the whole-loop statement -- literal registers)
#define CB_Hx(POL, K, GOP, RH)                                                                \
    "v_add_u32 %[z], %[z], %[ngap]\n\t"                                                      \
    "v_sub_u32 v" #K ", " GOP ", %[z]\n\t"                                                    \
    CB_ST("buffer_store_dword v" #K ", %[voffH], " RH ", s81 offen" CB_POLH(POL) "\n\t")      \
    "s_add_u32 s81, s81, %[strideH]\n\t"
#define CB_H(POL, K, G) CB_Hx(POL, K, "%[" G "]", "%[rH]")
// int64 H: the score is never negative, so the high dword is a zero register paired with each H register
#define CB_H64x(POL, K, K1, GOP, RH)                                                          \
    "v_add_u32 %[z], %[z], %[ngap]\n\t"                                                      \
    "v_sub_u32 v" #K ", " GOP ", %[z]\n\t"                                                    \
    "v_mov_b32 v" #K1 ", 0\n\t"                                                              \
    CB_ST("buffer_store_dwordx2 v[" #K ":" #K1 "], %[voffH], " RH ", s81 offen" POL "\n\t")   \
    "s_add_u32 s81, s81, %[strideH]\n\t"
#define CB_H64(POL, K, K1, G) CB_H64x(POL, K, K1, "%[" G "]", "%[rH]")
#define CB_PREDx(GOP, UPOP, DD, M1, M3)                                                       \
    "v_cmp_eq_u32_e64 " M1 ", " UPOP ", " GOP "\n\t"                                         \
    "v_cmp_eq_u32_e64 " M3 ", " DD ", " GOP "\n\t"
#define CB_SELx(POL, PST, HK, PI, M1, M3, RP)                                                 \
    "v_cmp_eq_u32_e32 vcc, 0, " HK "\n\t"                                                    \
    "v_cndmask_b32_e64 " PI ", 2, 1, " M1 "\n\t"                                             \
    "v_cndmask_b32_e64 " PI ", " PI ", 3, " M3 "\n\t"                                        \
    "v_cndmask_b32_e64 " PI ", " PI ", 0, vcc\n\t"                                           \
    CB_ST(PST " " PI ", %[voff], " RP ", s80 offen" CB_POLP(POL) "\n\t")                      \
    "s_add_u32 s80, s80, %[stride]\n\t"
#define CB_GROUPx(POL, PST, CH, U, G0, G1, G2, G3, H0, H1, H2, H3, RP)                        \
    "v_cmp_eq_u32_sdwa s[60:61], %[a], " CH " src0_sel:DWORD src1_sel:BYTE_0\n\t"            \
    "v_cmp_eq_u32_sdwa s[62:63], %[a], " CH " src0_sel:DWORD src1_sel:BYTE_1\n\t"            \
    "v_cmp_eq_u32_sdwa s[64:65], %[a], " CH " src0_sel:DWORD src1_sel:BYTE_2\n\t"            \
    "v_cmp_eq_u32_sdwa s[66:67], %[a], " CH " src0_sel:DWORD src1_sel:BYTE_3\n\t"            \
    "v_cndmask_b32_e64 v116, %[xm], %[mm], s[60:61]\n\t"                                     \
    "v_cndmask_b32_e64 v117, %[xm], %[mm], s[62:63]\n\t"                                     \
    "v_cndmask_b32_e64 v118, %[xm], %[mm], s[64:65]\n\t"                                     \
    "v_cndmask_b32_e64 v119, %[xm], %[mm], s[66:67]\n\t"                                     \
    "v_add_u32_dpp v120, " U ", v116 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"  \
    "v_add_u32_dpp v121, " G0 ", v117 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
    "v_add_u32_dpp v122, " G1 ", v118 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
    "v_add_u32_dpp v123, " G2 ", v119 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
    CB_PREDx(G0, U, "v120", "s[60:61]", "s[62:63]")                                           \
    CB_PREDx(G1, G0, "v121", "s[64:65]", "s[66:67]")                                          \
    CB_SELx(POL, PST, H0, "v124", "s[60:61]", "s[62:63]", RP)                                 \
    CB_PREDx(G2, G1, "v122", "s[68:69]", "s[70:71]")                                          \
    CB_SELx(POL, PST, H1, "v125", "s[64:65]", "s[66:67]", RP)                                 \
    CB_PREDx(G3, G2, "v123", "s[72:73]", "s[74:75]")                                          \
    CB_SELx(POL, PST, H2, "v126", "s[68:69]", "s[70:71]", RP)                                 \
    CB_SELx(POL, PST, H3, "v127", "s[72:73]", "s[74:75]", RP)
#define CB_GROUP(POL, PST, CH, U, G0, G1, G2, G3, H0, H1, H2, H3)                             \
    CB_GROUPx(POL, PST, CH, "%[" U "]", "%[" G0 "]", "%[" G1 "]", "%[" G2 "]", "%[" G3 "]", H0, H1, H2, H3, "%[rP]")
typedef int sw_i32x4 __attribute__((ext_vector_type(4)));
#define CB_ASM(POL, PST) \
    asm volatile( \
        "s_load_dwordx4 s[76:79], %[cptr], 0x0\n\t" \
        "s_mov_b32 s81, 0\n\t" \
        CB_H(POL, 100, "g1") CB_H(POL, 101, "g2") CB_H(POL, 102, "g3") CB_H(POL, 103, "g4") CB_H(POL, 104, "g5") CB_H(POL, 105, "g6") CB_H(POL, 106, "g7") CB_H(POL, 107, "g8") \
        CB_H(POL, 108, "g9") CB_H(POL, 109, "g10") CB_H(POL, 110, "g11") CB_H(POL, 111, "g12") CB_H(POL, 112, "g13") CB_H(POL, 113, "g14") CB_H(POL, 114, "g15") CB_H(POL, 115, "g16") \
        "v_max3_i32 %[bm], %[bm], v100, v101\n\t" \
        "v_max3_i32 v116, v102, v103, v104\n\t" \
        "v_max3_i32 v117, v105, v106, v107\n\t" \
        "v_max3_i32 v118, v108, v109, v110\n\t" \
        "v_max3_i32 v119, v111, v112, v113\n\t" \
        "v_max3_i32 %[bm], %[bm], v114, v115\n\t" \
        "v_max3_i32 v116, v116, v117, v118\n\t" \
        "v_max3_i32 %[bm], %[bm], v116, v119\n\t" \
        "s_mov_b32 s80, 0\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        CB_GROUP(POL, PST, "s76", "g0", "g1", "g2", "g3", "g4", "v100", "v101", "v102", "v103") \
        CB_GROUP(POL, PST, "s77", "g4", "g5", "g6", "g7", "g8", "v104", "v105", "v106", "v107") \
        CB_GROUP(POL, PST, "s78", "g8", "g9", "g10", "g11", "g12", "v108", "v109", "v110", "v111") \
        CB_GROUP(POL, PST, "s79", "g12", "g13", "g14", "g15", "g16", "v112", "v113", "v114", "v115") \
        : [z] "+v"(z), [bm] "+v"(blkmax) \
        : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]), [g7] "v"(g[7]), \
          [g8] "v"(g[8]), [g9] "v"(g[9]), [g10] "v"(g[10]), [g11] "v"(g[11]), [g12] "v"(g[12]), [g13] "v"(g[13]), [g14] "v"(g[14]), \
          [g15] "v"(g[15]), [g16] "v"(g[16]), [a] "v"(a_l), [mm] "v"(mm_v), [xm] "v"(xm_v), [ngap] "v"(ngap_v), [voff] "v"(voff), \
          [voffH] "v"(voffH), [rH] "s"(rH), [rP] "s"(rP), [stride] "s"(stride), [strideH] "s"(strideH), [cptr] "s"(chars) \
        : "vcc", "scc", "memory", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", \
          "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", \
          "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", \
          "v124", "v125", "v126", "v127")
#define CB_ASM64(POL, PST) \
    asm volatile( \
        "s_load_dwordx4 s[76:79], %[cptr], 0x0\n\t" \
        "s_mov_b32 s81, 0\n\t" \
        CB_H64(POL, 64, 65, "g1") CB_H64(POL, 66, 67, "g2") CB_H64(POL, 68, 69, "g3") CB_H64(POL, 70, 71, "g4") \
        CB_H64(POL, 72, 73, "g5") CB_H64(POL, 74, 75, "g6") CB_H64(POL, 76, 77, "g7") CB_H64(POL, 78, 79, "g8") \
        CB_H64(POL, 80, 81, "g9") CB_H64(POL, 82, 83, "g10") CB_H64(POL, 84, 85, "g11") CB_H64(POL, 86, 87, "g12") \
        CB_H64(POL, 88, 89, "g13") CB_H64(POL, 90, 91, "g14") CB_H64(POL, 92, 93, "g15") CB_H64(POL, 94, 95, "g16") \
        "v_max3_i32 %[bm], %[bm], v64, v66\n\t" \
        "v_max3_i32 v116, v68, v70, v72\n\t" \
        "v_max3_i32 v117, v74, v76, v78\n\t" \
        "v_max3_i32 v118, v80, v82, v84\n\t" \
        "v_max3_i32 v119, v86, v88, v90\n\t" \
        "v_max3_i32 %[bm], %[bm], v92, v94\n\t" \
        "v_max3_i32 v116, v116, v117, v118\n\t" \
        "v_max3_i32 %[bm], %[bm], v116, v119\n\t" \
        "s_mov_b32 s80, 0\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        CB_GROUP(POL, PST, "s76", "g0", "g1", "g2", "g3", "g4", "v64", "v66", "v68", "v70") \
        CB_GROUP(POL, PST, "s77", "g4", "g5", "g6", "g7", "g8", "v72", "v74", "v76", "v78") \
        CB_GROUP(POL, PST, "s78", "g8", "g9", "g10", "g11", "g12", "v80", "v82", "v84", "v86") \
        CB_GROUP(POL, PST, "s79", "g12", "g13", "g14", "g15", "g16", "v88", "v90", "v92", "v94") \
        : [z] "+v"(z), [bm] "+v"(blkmax) \
        : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]), [g7] "v"(g[7]), \
          [g8] "v"(g[8]), [g9] "v"(g[9]), [g10] "v"(g[10]), [g11] "v"(g[11]), [g12] "v"(g[12]), [g13] "v"(g[13]), [g14] "v"(g[14]), \
          [g15] "v"(g[15]), [g16] "v"(g[16]), [a] "v"(a_l), [mm] "v"(mm_v), [xm] "v"(xm_v), [ngap] "v"(ngap_v), [voff] "v"(voff), \
          [voffH] "v"(voffH), [rH] "s"(rH), [rP] "s"(rP), [stride] "s"(stride), [strideH] "s"(strideH), [cptr] "s"(chars) \
        : "vcc", "scc", "memory", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", \
          "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", \
          "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", \
          "v92", "v93", "v94", "v95", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")
// NT: streaming (nt) stores -- H and P are written once and never re-read by the fill; which policy is faster
// depends on the problem size (measured: nt wins at 16384^2, plain write-back at 32768^2).
// P8: the predecessor matrix as one byte per cell (compact P) instead of the reference's int32.
template <bool NT, bool P8>
__device__ __forceinline__ void consumer_block16(const u32 (&g)[SY_U + 1], u32& z, u32& blkmax, u32 a_l, u32 mm_v, u32 xm_v, u32 ngap_v,
                                                 u32 voff, u32 voffH, sw_i32x4 rH, sw_i32x4 rP, u32 stride, u32 strideH,
                                                 const unsigned char* chars) {
    if constexpr (NT && P8) { CB_ASM(" nt", "buffer_store_byte"); }
    else if constexpr (NT) { CB_ASM(" nt", "buffer_store_dword"); }
    else if constexpr (P8) { CB_ASM("", "buffer_store_byte"); }
    else { CB_ASM("", "buffer_store_dword"); }
}
template <bool NT, bool P8>
__device__ __forceinline__ void consumer_block16_h64(const u32 (&g)[SY_U + 1], u32& z, u32& blkmax, u32 a_l, u32 mm_v, u32 xm_v,
                                                     u32 ngap_v, u32 voff, u32 voffH, sw_i32x4 rH, sw_i32x4 rP, u32 stride, u32 strideH,
                                                     const unsigned char* chars) {
    if constexpr (NT && P8) { CB_ASM64(" nt", "buffer_store_byte"); }
    else if constexpr (NT) { CB_ASM64(" nt", "buffer_store_dword"); }
    else if constexpr (P8) { CB_ASM64("", "buffer_store_byte"); }
    else { CB_ASM64("", "buffer_store_dword"); }
}

// =================================================================================================
// Consumer, the WHOLE loop over a wave's full 16-row blocks in one asm statement (int32 H).  hipcc's own code around
// the one-block statement -- SGPR spills through v_readlane, 64-bit pointer arithmetic, the wrap test of the ring reads
// -- made a block cost 1.4 us where its ~300 instructions need 0.6 (measured round 2: six consumers could not keep up
// with a producer running 18 ns per row).  Per block here: poll the producer's progress, 17 ring reads along the skew
// (base + immediate when no lane wraps inside the block, else per-entry masked addresses), the block body, arg-max
// bookkeeping, publish, advance descriptors / pointers by NC blocks.
// literal registers: v[80:96] ring values of rows r0-1..r0+15, v97..v99 scratch, v[100:127] as in the one-block
// statement; s[44:47] / s[48:51] H / P buffer descriptors, s[52:53] row characters, s54 block index, s55 producer step
// that completes the block, s56 scratch, s58 spin count, s59 ring entry of row r0-1 in lane 0 (unmasked)
// =================================================================================================
#define CL_RD(K, OFF) "ds_read_b32 v" #K ", v99 offset:" #OFF "\n\t"
#define CL_RDW(K, I)                                                                          \
    "v_add_u32 v98, " #I ", %[E]\n\t"                                                         \
    "v_and_b32 v98, %[rmask], v98\n\t"                                                        \
    "v_lshl_add_u32 v98, v98, 2, %[lanebase]\n\t"                                             \
    "ds_read_b32 v" #K ", v98\n\t"
#define CL_ROW(HK, K)                                                                         \
    "v_cmp_eq_i32 s[60:61], v" #HK ", v97\n\t"                                                \
    "s_nop 1\n\t"                                                                             \
    "v_cndmask_b32_e64 v99, v99, " #K ", s[60:61]\n\t"
#define CL_ASM(POL, PST)                                                                      \
    asm volatile(                                                                             \
        "s_mov_b32 %[status], 0\n\t"                                                          \
        "s_mov_b32 s44, %[h0]\n\t" "s_mov_b32 s45, %[h1]\n\t" "s_mov_b32 s46, %[h2]\n\t" "s_mov_b32 s47, %[h3]\n\t" \
        "s_mov_b32 s48, %[p0]\n\t" "s_mov_b32 s49, %[p1]\n\t" "s_mov_b32 s50, %[p2]\n\t" "s_mov_b32 s51, %[p3]\n\t" \
        "s_mov_b64 s[52:53], %[cptr]\n\t"                                                     \
        "s_mov_b32 s54, %[q0]\n\t"                                                            \
        "s_mov_b32 s55, %[need0]\n\t"                                                         \
        "s_mov_b32 s59, %[es0]\n"                                                             \
        "Lcl_loop_%=:\n\t"                                                                    \
        "s_mov_b32 s58, 0\n"                                                                  \
        "Lcl_poll_%=:\n\t"                                                                    \
        "ds_read_b32 v97, %[prodaddr]\n\t"                                                    \
        "s_waitcnt lgkmcnt(0)\n\t"                                                            \
        "v_readfirstlane_b32 s56, v97\n\t"                                                    \
        "s_cmp_ge_i32 s56, s55\n\t"                                                           \
        "s_cbranch_scc1 Lcl_go_%=\n\t"                                                        \
        "s_sleep 1\n\t"                                                                       \
        "s_add_i32 s58, s58, 1\n\t"                                                           \
        "s_cmp_lt_u32 s58, 0x1000000\n\t"                                                     \
        "s_cbranch_scc1 Lcl_poll_%=\n\t"                                                      \
        "s_mov_b32 %[status], 1\n\t"                                                          \
        "s_branch Lcl_exit_%=\n"                                                              \
        "Lcl_go_%=:\n\t"                                                                      \
        "s_load_dwordx4 s[76:79], s[52:53], 0x0\n\t"                                          \
        "s_and_b32 s56, s59, %[srmask]\n\t"                                                   \
        "s_cmp_lt_u32 s56, %[nowrap]\n\t"                                                     \
        "s_cbranch_scc0 Lcl_wrap_%=\n\t"                                                      \
        "v_and_b32 v99, %[rmask], %[E]\n\t"                                                   \
        "v_lshl_add_u32 v99, v99, 2, %[lanebase]\n\t"                                         \
        CL_RD(80, 0) CL_RD(81, 4) CL_RD(82, 8) CL_RD(83, 12) CL_RD(84, 16) CL_RD(85, 20) CL_RD(86, 24) CL_RD(87, 28) CL_RD(88, 32) \
        CL_RD(89, 36) CL_RD(90, 40) CL_RD(91, 44) CL_RD(92, 48) CL_RD(93, 52) CL_RD(94, 56) CL_RD(95, 60) CL_RD(96, 64)           \
        "s_branch Lcl_have_%=\n"                                                              \
        "Lcl_wrap_%=:\n\t"                                                                    \
        CL_RDW(80, 0) CL_RDW(81, 1) CL_RDW(82, 2) CL_RDW(83, 3) CL_RDW(84, 4) CL_RDW(85, 5) CL_RDW(86, 6) CL_RDW(87, 7) CL_RDW(88, 8) \
        CL_RDW(89, 9) CL_RDW(90, 10) CL_RDW(91, 11) CL_RDW(92, 12) CL_RDW(93, 13) CL_RDW(94, 14) CL_RDW(95, 15) CL_RDW(96, 16)       \
        "Lcl_have_%=:\n\t"                                                                    \
        "s_mov_b32 s81, 0\n\t"                                                                \
        "s_waitcnt lgkmcnt(0)\n\t"                                                            \
        CB_Hx(POL, 100, "v81", "s[44:47]") CB_Hx(POL, 101, "v82", "s[44:47]") CB_Hx(POL, 102, "v83", "s[44:47]") CB_Hx(POL, 103, "v84", "s[44:47]") \
        CB_Hx(POL, 104, "v85", "s[44:47]") CB_Hx(POL, 105, "v86", "s[44:47]") CB_Hx(POL, 106, "v87", "s[44:47]") CB_Hx(POL, 107, "v88", "s[44:47]") \
        CB_Hx(POL, 108, "v89", "s[44:47]") CB_Hx(POL, 109, "v90", "s[44:47]") CB_Hx(POL, 110, "v91", "s[44:47]") CB_Hx(POL, 111, "v92", "s[44:47]") \
        CB_Hx(POL, 112, "v93", "s[44:47]") CB_Hx(POL, 113, "v94", "s[44:47]") CB_Hx(POL, 114, "v95", "s[44:47]") CB_Hx(POL, 115, "v96", "s[44:47]") \
        "v_max3_i32 v97, v100, v101, v102\n\t"                                                \
        "v_max3_i32 v116, v103, v104, v105\n\t"                                               \
        "v_max3_i32 v117, v106, v107, v108\n\t"                                               \
        "v_max3_i32 v118, v109, v110, v111\n\t"                                               \
        "v_max3_i32 v119, v112, v113, v114\n\t"                                               \
        "v_max3_i32 v97, v97, v115, v116\n\t"                                                 \
        "v_max3_i32 v97, v97, v117, v118\n\t"                                                 \
        "v_max_i32 v97, v97, v119\n\t"                                                        \
        "s_mov_b32 s80, 0\n\t"                                                                \
        CB_GROUPx(POL, PST, "s76", "v80", "v81", "v82", "v83", "v84", "v100", "v101", "v102", "v103", "s[48:51]") \
        CB_GROUPx(POL, PST, "s77", "v84", "v85", "v86", "v87", "v88", "v104", "v105", "v106", "v107", "s[48:51]") \
        CB_GROUPx(POL, PST, "s78", "v88", "v89", "v90", "v91", "v92", "v108", "v109", "v110", "v111", "s[48:51]") \
        CB_GROUPx(POL, PST, "s79", "v92", "v93", "v94", "v95", "v96", "v112", "v113", "v114", "v115", "s[48:51]") \
        /* arg-max: the first of my blocks that reached the best value of my column.  With H in HBM the row is re-read at the  \
           end (rr != 0); without it the lowest row holding the block maximum is taken from the H registers right here, in  \
           the lanes that improved (the block index operand then carries the ROW) */                                         \
        "v_cmp_gt_i32 vcc, v97, %[bestv]\n\t"                                                 \
        "v_mov_b32 v98, s54\n\t"                                                              \
        "s_cmp_lg_u32 %[rr], 0\n\t"                                                           \
        "s_cbranch_scc1 Lcl_upd_%=\n\t"                                                       \
        "s_cbranch_vccz Lcl_pub_%=\n\t"                                                       \
        "s_mov_b64 s[62:63], vcc\n\t"                                                         \
        "v_mov_b32 v99, 15\n\t"                                                               \
        CL_ROW(114, 14) CL_ROW(113, 13) CL_ROW(112, 12) CL_ROW(111, 11) CL_ROW(110, 10) CL_ROW(109, 9) CL_ROW(108, 8) CL_ROW(107, 7)        \
        CL_ROW(106, 6) CL_ROW(105, 5) CL_ROW(104, 4) CL_ROW(103, 3) CL_ROW(102, 2) CL_ROW(101, 1) CL_ROW(100, 0)                          \
        "s_lshl_b32 s56, s54, 4\n\t"                                                          \
        "s_add_i32 s56, s56, 1\n\t"                                                           \
        "v_add_u32 v98, s56, v99\n\t"               /* row = 16 q + 1 + k */                  \
        "s_mov_b64 vcc, s[62:63]\n\t"                                                         \
        "s_nop 1\n"                                                                            \
        "Lcl_upd_%=:\n\t"                                                                     \
        "v_cndmask_b32 %[bestv], %[bestv], v97, vcc\n\t"                                      \
        "v_cndmask_b32 %[bestblk], %[bestblk], v98, vcc\n"                                     \
        "Lcl_pub_%=:\n\t"                                                                     \
        "v_mov_b32 v98, s54\n\t"                                                              \
        "ds_write_b32 %[slotaddr], v98\n\t"        /* in LDS order behind this block's ring reads */ \
        "v_add_u32 %[z], %[z], %[zstep]\n\t"                                                  \
        "v_add_u32 %[E], %[E], %[estep]\n\t"                                                  \
        "s_add_i32 s59, s59, %[sestep]\n\t"                                                   \
        "s_add_i32 s55, s55, %[sestep]\n\t"                                                   \
        "s_add_u32 s44, s44, %[bh0]\n\t"                                                      \
        "s_addc_u32 s45, s45, %[bh1]\n\t"                                                     \
        "s_add_u32 s48, s48, %[bp0]\n\t"                                                      \
        "s_addc_u32 s49, s49, %[bp1]\n\t"                                                     \
        "s_add_u32 s52, s52, %[sestep]\n\t"                                                   \
        "s_addc_u32 s53, s53, 0\n\t"                                                          \
        "s_add_i32 s54, s54, %[qstep]\n\t"                                                    \
        "s_cmp_lt_i32 s54, %[qend]\n\t"                                                       \
        "s_cbranch_scc1 Lcl_loop_%=\n"                                                        \
        "Lcl_exit_%=:\n\t"                                                                    \
        : [status] "=&s"(status), [z] "+v"(z), [E] "+v"(E), [bestv] "+v"(bestv), [bestblk] "+v"(bestblk)                                   \
        : [a] "v"(a_l), [mm] "v"(mm_v), [xm] "v"(xm_v), [ngap] "v"(ngap_v), [voff] "v"(voff), [voffH] "v"(voffH), [lanebase] "v"(lanebase),  \
          [zstep] "v"(zstep), [estep] "v"(estep), [prodaddr] "v"(prod_addr), [slotaddr] "v"(slot_addr), [rmask] "v"(rmask),                  \
          [h0] "s"(rH.x), [h1] "s"(rH.y), [h2] "s"(rH.z), [h3] "s"(rH.w), [p0] "s"(rP.x), [p1] "s"(rP.y), [p2] "s"(rP.z), [p3] "s"(rP.w),     \
          [cptr] "s"(chars), [q0] "s"(q0), [need0] "s"(need0), [es0] "s"(es0), [srmask] "s"((int)rmask), [nowrap] "s"(nowrap),               \
          [sestep] "s"((int)estep), [bh0] "s"(bh0), [bh1] "s"(bh1), [bp0] "s"(bp0), [bp1] "s"(bp1), [qstep] "s"(qstep), [qend] "s"(qend),     \
          [stride] "s"(stride), [strideH] "s"(strideH), [rr] "s"(rr)                                                                       \
        : "vcc", "scc", "memory", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s58", "s59", "s60", \
          "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",   \
          "s80", "s81", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",   \
          "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",  \
          "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")
// blocks q0, q0+qstep, ... < qend of one consumer wave; E: ring entry of row r0-1 for this lane (unmasked), advanced here
template <bool NT, bool P8>
__device__ __forceinline__ int consumer_loop32(u32& z, u32& E, u32& bestv, u32& bestblk, u32 lanebase, u32 a_l, u32 mm_v, u32 xm_v, u32 ngap_v,
                                               u32 voff, u32 voffH, u32 zstep, u32 estep, u32 rmask, sw_i32x4 rH, sw_i32x4 rP, uint64_t blkH,
                                               uint64_t blkP, u32 stride, u32 strideH, const unsigned char* chars, u32 prod_addr, u32 slot_addr,
                                               int q0, int qend, int qstep, int need0, int es0, int rr) {
    int status;
    const int nowrap = (int)rmask + 1 - 63 - SY_U;   // ring entry (lane 0) below which no lane's 17 entries wrap
    const int bh0 = (int)(u32)blkH, bh1 = (int)(u32)(blkH >> 32), bp0 = (int)(u32)blkP, bp1 = (int)(u32)(blkP >> 32);
    if constexpr (NT && P8) { CL_ASM(" nt", "buffer_store_byte"); }
    else if constexpr (NT) { CL_ASM(" nt", "buffer_store_dword"); }
    else if constexpr (P8) { CL_ASM("", "buffer_store_byte"); }
    else { CL_ASM("", "buffer_store_dword"); }
    return status;
}

// (the kernel proper: sw_systolic below, behind sw_systolic2.inc, wraps this body -- as the fall-back of a one-launch fill it also runs
// that file's epilogue)
template <typename HT, int NS, int NC>
__device__ __forceinline__ void sw_systolic_body(const unsigned char* seq_a, const unsigned char* seq_b, const unsigned char* bpad, FillParams p) {
    __shared__ SysLds<NS> lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t M = p.M;
    const int rows = (int)p.rows;
    const int ngap = p.ngap, mm = p.mm, xm = p.xm;
    const int ngroups = (p.nstrips + NS - 1) / NS;
    const u64 tag_base = p.tag_base;
    const int64_t estride = p.rows + 1;
    // perm producer: eligible on the host side (score range -> gbias != 0) AND an alphabet of at most 7 letters (found on
    // the device by sw_pad_b); it always uses the fast step numbering
    const bool perm = (p.gbias != 0) && (*(const unsigned int*)(p.atab + 256) <= 7u) && !(p.debug_flags & 16);
    const int phib = perm ? p.nstrips - 1 : p.phi_base;
    const int gb = perm ? (int)p.gbias : 0;       // carried by every G value
    const int ugran = perm ? 64 : SY_U;   // the perm producer tests for the end once per 64-step chunk
    auto u_total = [&](int s) { return (rows + SY_W + phi_of(s, phib) + ugran - 1) / ugran * ugran; };  // local steps 1..u_total

    const unsigned char* const seq_a0 = seq_a; const unsigned char* const seq_b0 = seq_b; const unsigned char* const bpad0 = bpad;
    const FillParams p0 = p;
    const int64_t ntotal = (int64_t)ngroups * p0.npairs;
    for (int64_t gbase = 0; gbase < ntotal; gbase += gridDim.x) {
        // Optional XCD-aware order inside one pass of gridDim.x workgroups (option xcd_order): the hardware deals
        // workgroups to the 8 XCDs round-robin (blockIdx % 8); give neighbouring strip groups to one XCD so that
        // cache lines straddling two strips can be completed inside one L2.
        const int n_pass = (int)min((int64_t)gridDim.x, ntotal - gbase);
        if ((int)blockIdx.x >= n_pass) break;
        int local = (int)blockIdx.x;
        if (p0.xcd_order && n_pass >= 16) {
            const int x = (int)blockIdx.x & 7;
            int off = 0;
            for (int y = 0; y < x; ++y) off += (n_pass - y + 7) >> 3;
            local = off + ((int)blockIdx.x >> 3);
        }
        const int64_t gidx = gbase + local;
        const int grp = (int)(gidx % ngroups);
        const int64_t pair = gidx / ngroups;
        // per-problem views (batch of independent pairs: BASELINE config 5)
        seq_a = seq_a0 + pair * p0.a_pstride;
        seq_b = seq_b0 + pair * p0.b_pstride;
        bpad = bpad0 + pair * p0.bpad_pstride;
        p.bpad16 = p0.bpad16 + pair * p0.bpad_pstride;
        p.edge = p0.edge + pair * p0.edge_pstride;
        p.edge4 = p0.edge4 + pair * p0.edge4_pstride;
        p.bcode = p0.bcode + pair * p0.bpad_pstride;
        p.result_key = p0.result_key + pair;
        p.H = p0.H ? (void*)((char*)p0.H + pair * p0.hp_pstride * (int64_t)sizeof(HT)) : nullptr;
        p.P = p0.P ? (int32_t*)((char*)p0.P + pair * p0.hp_pstride * (int64_t)p0.p_bytes) : nullptr;
        if (threadIdx.x < NS) lds.prod_u[threadIdx.x] = 0;
        if (threadIdx.x < NS * 8) lds.cons_blk[threadIdx.x / 8][threadIdx.x % 8] = ((int)(threadIdx.x % 8) < NC) ? -1 : 0x7fffffff;
        if (threadIdx.x == 0) { lds.halo_ready = 1; lds.exp_done = 0; lds.never = 0x7fffffff; }
        // the halo ring starts empty: the perm producer takes a 0 for "not imported yet" (every G carries the launch bias)
        for (int i = threadIdx.x; i < SY_RH; i += blockDim.x) lds.halo[i] = 0u;
        if ((p.debug_flags & 2048) && p.gbias == 0)   // round-1 producers: poison the halo ring, so that any slot consumed before an
            for (int i = threadIdx.x; i < SY_RH; i += blockDim.x) lds.halo[i] = 0x700000u + (u32)i;   // importer wrote it shows up in the output
        const int s0 = grp * NS;
        const int nact = min(NS, p.nstrips - s0);  // active strips of this group
        const bool has_top = (p.top != nullptr) || (p.top_gran != nullptr);
        if (p.top_gran && wave < NS) {
            // band-resident launch: this strip's 64 columns of the row above, {tag, value} granules written by whoever
            // receives the neighbour band's last row (any order, any time): the data is its own flag
            const int64_t jt = (int64_t)(s0 + wave) * SY_W + lane;
            const bool want = (wave < nact) && jt <= p.cols;
            u64 gr = 0;
            uint64_t t0 = 0;
            unsigned n = 0;
            for (;;) {
                if (want) gr = __hip_atomic_load((gu64*)(p.top_gran + jt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const bool ok = !want || (u32)(gr >> 32) == p.top_tag;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(8);
                if ((++n & 63u) == 0) {
                    const uint64_t now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    if (((now - t0) >> 10) > (uint64_t)p.top_wait_ticks)
                        __hip_atomic_store((gu32*)p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((gu32*)p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
                }
            }
            lds.topv[wave][lane] = (int)(u32)gr;
        }
        __syncthreads();
        if (p.top_gran && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load((gu32*)p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
        // H of the halo row above column (strip ls, lane l)
        auto top_at = [&](int ls_, int lane_, int64_t j_) -> int {
            return p.top_gran ? lds.topv[ls_][lane_] : (p.top ? p.top[j_] : 0);
        };

        // Role of each wave.  Waves are dealt to the four SIMDs cyclically, so waves w, w+4, w+8, w+12 share a SIMD.
        // A producer issues one instruction per ~4.5 clk, i.e. it fills the VALU pipe of its SIMD almost alone; measured
        // (round 2): the same producer loop takes 12.0 ns per step alone on a SIMD and 20.8 ns beside a polling importer.
        // NS == 1: the producer (wave 0) gets SIMD 0 for itself -- waves 4, 8, 12 go straight to the barrier and sleep
        // there --, consumers, the exporter and the importers fill the other three SIMDs.
        // NS == 2 (batches, VALU-bound anyway): the two helper waves take slots 4 and 5, the SIMDs of the two producers.
        constexpr int NWAVES = NS * (1 + NC) + 2;
        static_assert(NWAVES <= 16 && NC <= 8, "at most 1024 threads; four counter slots shared by up to two consumers each");
        enum { R_PROD, R_CONS, R_IMP, R_EXP, R_IDLE };
        int role, cw = 0;
        if constexpr (NS == 1) {
            const int k = wave - 1 - (wave >> 2);   // ordinal among the waves off SIMD 0
            role = (wave == 0) ? R_PROD : ((wave & 3) == 0) ? R_IDLE : (k < NC) ? R_CONS : (k == NC) ? R_EXP : R_IMP;
            cw = k;
        } else {
            constexpr bool helpers_mid = (NWAVES > 6);
            const int h_imp = helpers_mid ? 4 : NWAVES - 2, h_exp = helpers_mid ? 5 : NWAVES - 1;
            role = (wave < NS) ? R_PROD : (wave == h_imp || wave >= NWAVES) ? R_IMP : (wave == h_exp) ? R_EXP : R_CONS;
            cw = wave - NS - (wave > h_imp ? 1 : 0) - (wave > h_exp ? 1 : 0);  // consumer ordinal 0..NS*NC-1
        }
        if (p.dbg && (p.debug_flags & 64) && blockIdx.x == 0 && lane == 0) {   // where did the hardware put my waves?
            u32 hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            p.dbg[6 * p.nstrips + 32 + wave] = ((u64)role << 32) | hwid;
        }
        if (role == R_PROD) {
            // ================================ producer ================================
            const int ls = wave, s = s0 + ls;
            asm volatile("; SW_PRODUCER_PATH_BEGIN ");
            if (ls < nact && !((p.debug_flags & 8) && s == 1)) {   // debug bit 3: strip 1 never runs (tests the abort path)
                const int phi = phi_of(s, phib);
                const int UT = u_total(s);
                const u32 j = (u32)s * SY_W + (u32)lane;
                const bool jvalid = (int64_t)j <= p.cols;
                const u32 a_l = (lane >= 1 && jvalid) ? (u32)seq_a[j - 1] : (u32)SY_ASENT;
                const u32 G0v = jvalid ? (u32)(gb + top_at(ls, lane, j) + ngap * (int)j) : (u32)gb;  // row 0 in G-space
                const u32 mm_v = (u32)mm, xm_v = (u32)xm, ngap_v = (u32)ngap;
                const bool lefthalo = (ls == 0);         // halo comes from the import ring
                const bool has_right = (ls + 1 < nact);  // another producer reads my lane-63 column
                const bool has_export = (ls + 1 == nact) && (s + 1 < p.nstrips);
                const u32 ringbase = (u32)(size_t)&lds.ring[ls][0];
                const int* left_cnt = lefthalo ? &lds.halo_ready : &lds.prod_u[ls - 1];
                const int* right_cnt = has_right ? &lds.prod_u[ls + 1] : has_export ? &lds.exp_done : &lds.never;
                const u32 cnt_addr = (u32)(size_t)left_cnt;
                // halo of my local step u: left strip's lane 63 at ITS local step u + hoff (hoff % 4 == 0)
                const int hoff = lefthalo ? 0 : SY_W + phi_of(s - 1, phib) - phi;
                const u32 hbase = lefthalo ? (u32)(size_t)&lds.halo[0] : (u32)(size_t)&lds.ring[ls - 1][63 * SY_LSTR];
                const int hmask = lefthalo ? SY_RH - 1 : SY_R - 1;
                auto halo_addr = [&](int u) -> u32 { return hbase + (u32)((u - 1 + hoff) & hmask) * 4u; };
                const int left_total = lefthalo ? 0x7fffffff : u_total(s - 1);
                // counter value that makes the halo of local steps < u_end readable
                auto halo_need = [&](int u_end) -> int {
                    return lefthalo ? min(u_end, rows + phi + 1) : min(u_end - 1 + hoff, left_total);
                };
                auto cons_rows_done = [&]() -> int {
                    const int a0 = min(lds_load(&lds.cons_blk[ls][0]), lds_load(&lds.cons_blk[ls][4])), a1 = min(lds_load(&lds.cons_blk[ls][1]), lds_load(&lds.cons_blk[ls][5]));
                    const int a2 = min(lds_load(&lds.cons_blk[ls][2]), lds_load(&lds.cons_blk[ls][6])), a3 = min(lds_load(&lds.cons_blk[ls][3]), lds_load(&lds.cons_blk[ls][7]));
                    return SY_U * (min(min(a0, a1), min(a2, a3)) + 1);
                };

                if (p.dbg && lane == 0) p.dbg[2 * s] = __builtin_amdgcn_s_memrealtime();
                if (p.dbg && lane == 0 && s == 0) p.dbg[6 * p.nstrips + 48] = __builtin_amdgcn_s_memtime();   // shader clocks too: what does the chip run at?
                if (perm) {
                    // ---- perm producer (whole strip loop in one asm statement) ----
                    // profile of this lane: score of its NEXT column's letter against every letter code (7: outside)
                    const bool nvalid = (int64_t)j < p.cols;
                    const u32 acode = nvalid ? (u32)p.atab[seq_a[j]] : 8u;
                    u32 prof_lo = 0, prof_hi = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) prof_lo |= ((u32)(unsigned char)(signed char)((u32)c == acode ? mm : xm)) << (8 * c);
#pragma unroll
                    for (int c = 4; c < 7; ++c) prof_hi |= ((u32)(unsigned char)(signed char)((u32)c == acode ? mm : xm)) << (8 * (c - 4));
                    prof_hi |= ((u32)(unsigned char)(signed char)SW_PERM_PAD) << 24;
                    const u32 wbase = ringbase + (u32)lane * (u32)SY_LSTR;
                    // lane l, local step u scores b[u - phi - l] (the row BELOW its own): byte offset of the window that
                    // ends at step 0; the scalar base moves with the blocks
                    const unsigned char* cb = p.bcode + (p.bfront - 78 - phi);
                    const u32 voff = 63u - (u32)lane;
                    const u32 z1 = (u32)(gb + ngap * (1 - phi + s * SY_W));       // floor of local step 1
                    const int k1 = lefthalo ? 2 * SY_U : 2 * SY_U - 1 + hoff;
                    const int k2 = lefthalo ? 0x7fffff00 : left_total;   // the importer defines the halo of every step, below the matrix too
                    const int kc = 30 - SY_R - phi;
                    const int kr = (has_right ? -(SY_R + 1) + 16 : -SY_R - phi) - 32;
                    const u32 expoff = (has_export && lane == 63) ? 0u : SY_OOB;
                    const uint64_t eb = (uint64_t)(uintptr_t)(p.edge4 + (int64_t)s * p.e4stride);
                    const sw_i32x4p erc = {(int)(u32)eb, (int)(u32)(eb >> 32), 0x7FFFFF00, 0x00020000};
                    int polls[2];
                    // halo from the importer's ring: validated on use (unfilled slots read 0); from the left producer's ring in
                    // this workgroup: by its progress counter
                    const int st = (lefthalo && !(p.debug_flags & 8192))   // (debug bit 13: the progress form everywhere, for A/B runs)
                        ? producer_perm_valid(prof_lo, prof_hi, ngap_v, wbase, voff, z1, G0v, G0v + (u32)SW_PERM_PAD, expoff, cb, hbase, hoff * 4, cnt_addr,
                                              (u32)(size_t)&lds.cons_blk[ls][0], (u32)(size_t)(has_right ? right_cnt : &lds.never),
                                              (u32)(size_t)&lds.prod_u[ls], UT, k1, k2, kc, kr, hmask * 4 + 3, erc, (p.debug_flags & 32) ? 0 : 64, polls)
                        : producer_perm_progress(prof_lo, prof_hi, ngap_v, wbase, voff, z1, G0v, G0v + (u32)SW_PERM_PAD, expoff, cb, hbase, hoff * 4, cnt_addr,
                                                 (u32)(size_t)&lds.cons_blk[ls][0], (u32)(size_t)(has_right ? right_cnt : &lds.never),
                                                 (u32)(size_t)&lds.prod_u[ls], UT, k1, k2, kc, kr, hmask * 4 + 3, erc, (p.debug_flags & 32) ? 0 : 64, polls);
                    if (p.dbg && lane == 0) {
                        p.dbg[2 * s + 1] = __builtin_amdgcn_s_memrealtime();
                        if (s == 0) p.dbg[6 * p.nstrips + 49] = __builtin_amdgcn_s_memtime();
                        p.dbg[4 * p.nstrips + 16 + 2 * s] = (u64)polls[0];
                        p.dbg[4 * p.nstrips + 17 + 2 * s] = (u64)polls[1];
                    }
                    if (st) {
                        __hip_atomic_store((gu32*)p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                } else if (phib >= 0) {
                    // ---- fast producer (whole strip loop in one asm statement) ----
                    const u32 wbase = ringbase + (u32)lane * (u32)SY_LSTR;
                    const u32 voff = 2u * (u32)(p.bfront - phi - lane);           // byte offset of local step 1's character
                    const u32 z0 = (u32)(ngap * (0 - phi + s * SY_W));            // floor of local step 0
                    const int k1 = lefthalo ? 2 * SY_U : 2 * SY_U - 1 + hoff;     // next block readable when left >= min(u0+k1, k2)
                    const int k2 = lefthalo ? rows + phi + 1 : left_total;
                    const int kc = 30 - SY_R - phi;           // back-pressure is checked every 32 steps
                    // my lane-63 entries of steps <= uc-193 get overwritten: the right-hand producer read them at its
                    // local step (mine - 64); the exporter counts ROWS (= my step - 63 - phi)
                    const int kr = (has_right ? -(SY_R + 1) + 16 : -SY_R - phi) - 32;
                    int polls[2];
                    const int st = producer_fast(a_l, xm_v, mm_v, ngap_v, wbase, voff, z0, p.bpad16, hbase, hoff * 4, cnt_addr,
                                                 (u32)(size_t)&lds.cons_blk[ls][0], (u32)(size_t)right_cnt, (u32)(size_t)&lds.prod_u[ls],
                                                 UT, k1, k2, kc, kr, hmask * 4 + 3, polls);
                    if (p.dbg && lane == 0) {
                        p.dbg[2 * s + 1] = __builtin_amdgcn_s_memrealtime();
                        // slow-path polls of this strip: ring back-pressure / halo not there yet (after the strip times and hop stamps)
                        p.dbg[4 * p.nstrips + 16 + 2 * s] = (u64)polls[0];
                        p.dbg[4 * p.nstrips + 17 + 2 * s] = (u64)polls[1];
                    }
                    if (st) {
                        __hip_atomic_store((gu32*)p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                } else {
                // ---- generic producer (halo row / positive mismatch: needs the row-0 injection) ----
                // this lane's row characters: at local step u it works on row u-phi-lane, i.e. b[u-phi-lane-1]
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                typedef u32x4 __attribute__((aligned(1))) u32x4_u;
                const unsigned char* bp = bpad + p.bfront - phi - lane - 1;  // + u
                // prologue injection: lane l takes G0 at local step l + phi
                u32 injv = (u32)(lane + phi);

                // state crossing block boundaries: results of the last two steps, the floor of the last
                // step, and the halo values of the coming block's first four steps
                u32 G1 = (lane == 0) ? G0v : 0u, G2 = 0u;        // "step 0": only lane 0 (row 0 of the halo column) matters
                u32 Z = (u32)(ngap * (0 - phi + s * SY_W));      // floor of local step 0
                u32 h0 = 0, h1 = 0, h2 = 0, h3 = 0;
                Spin spin;
                bool have_halo = false;  // h0..h3 hold the halo of the coming block's first group

                // one 16-step block: local steps u0..u0+15, KB = index of u0 inside its 64-step chunk
                auto run_block = [&](auto PRO, auto KBt, int u0, const u32x4& C) -> bool {
                    constexpr int KB = decltype(KBt)::value;
                    if (!have_halo) {
                        while (lds_load(left_cnt) < halo_need(u0 + SY_U))
                            if (spin.fail(p.abort_flag)) return false;
                        asm volatile("" ::: "memory");
                        const u32* hp = (const u32*)(lefthalo ? (const void*)&lds.halo[(u0 - 1 + hoff) & hmask]
                                                              : (const void*)(&lds.ring[ls - 1][63 * SY_LSTR] + 4 * ((u0 - 1 + hoff) & hmask)));
                        h0 = __hip_atomic_load(hp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        h1 = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        h2 = __hip_atomic_load(hp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        h3 = __hip_atomic_load(hp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    const u32 waddr = ringbase + (u32)lane * (u32)SY_LSTR + (u32)((u0 - 1 - KB) & (SY_R - 1)) * 4u;
                    const u32 ra1 = halo_addr(u0 + 4), ra2 = halo_addr(u0 + 8), ra3 = halo_addr(u0 + 12), ran = halo_addr(u0 + 16);
                    int cnt;
                    if constexpr (decltype(PRO)::value) SW_BLOCK(SW_SP, KB); else SW_BLOCK(SW_S, KB);
                    // cnt = the left neighbour's progress as it was BEFORE the next block's halo (h0..h3)
                    // was read: if it already covered that block, what we read is valid
                    have_halo = (u0 + SY_U <= UT) && (cnt >= halo_need(u0 + 2 * SY_U));
                    lds_store(&lds.prod_u[ls], u0 + SY_U - 1);  // in order behind the ring writes
                    return true;
                };
                // one 64-step chunk from local step uc (uc % 64 == 1)
                u32x4 c0 = *(const u32x4_u*)(bp + 1), c1 = *(const u32x4_u*)(bp + 17), c2 = *(const u32x4_u*)(bp + 33), c3 = *(const u32x4_u*)(bp + 49);
                auto run_chunk = [&](auto PRO, int uc) -> bool {
                    // ring slots of this chunk hold steps uc-R.. : their rows (<= uc+62-R-phi) must have been
                    // consumed, and their lane-63 entries read by the right-hand strip / the exporter
                    while (cons_rows_done() < uc + 62 - SY_R - phi)
                        if (spin.fail(p.abort_flag)) return false;
                    while (lds_load(right_cnt) < uc + 63 - SY_R)
                        if (spin.fail(p.abort_flag)) return false;
                    const unsigned char* nb_ = bp + uc + 64;  // next chunk's characters
                    const u32x4 n0 = *(const u32x4_u*)(nb_), n1 = *(const u32x4_u*)(nb_ + 16), n2 = *(const u32x4_u*)(nb_ + 32), n3 = *(const u32x4_u*)(nb_ + 48);
                    if constexpr (decltype(PRO)::value) injv = (u32)(lane + phi - (uc - 1));
                    if (!run_block(PRO, std::integral_constant<int, 0>{}, uc, c0)) return false;
                    if (uc + 16 <= UT && !run_block(PRO, std::integral_constant<int, 16>{}, uc + 16, c1)) return false;
                    if (uc + 32 <= UT && !run_block(PRO, std::integral_constant<int, 32>{}, uc + 32, c2)) return false;
                    if (uc + 48 <= UT && !run_block(PRO, std::integral_constant<int, 48>{}, uc + 48, c3)) return false;
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                    return true;
                };
                int uc = 1;
                if (!run_chunk(std::true_type{}, uc)) return;  // local steps 1..128 contain every row-0 injection
                uc += 64;
                if (uc <= UT) { if (!run_chunk(std::true_type{}, uc)) return; uc += 64; }
                for (; uc <= UT; uc += 64)
                    if (!run_chunk(std::false_type{}, uc)) return;
                }
            }
            asm volatile("; SW_PRODUCER_PATH_END");
        } else if (role == R_CONS) {
            // ================================ consumer ================================
            const int ls = cw % NS, ci = cw / NS, s = s0 + ls;
            if (ls < nact && (p.debug_flags & 2)) {
                lds_store(&lds.cons_blk[ls][ci], 1 << 24);  // timing experiment: producer alone
            } else if (ls < nact) {
                const int phi = phi_of(s, phib);
                const u32 j = (u32)s * SY_W + (u32)lane;
                const bool jvalid = (int64_t)j <= p.cols;
                // lane 0 of strip 0 is the tile's own column 0 only for a whole matrix (left == NULL)
                const bool cell_ok = jvalid && (lane >= 1 || (s == 0 && p.left == nullptr));
                const bool store_h = cell_ok && p.H != nullptr && !(p.debug_flags & 1);
                const bool store_p = cell_ok && p.P != nullptr && !(p.debug_flags & 1);
                const bool last_strip = (s + 1 == p.nstrips);
                const int lc = (int)p.cols - s * SY_W;          // lane of the tile's last column (in the last strip)
                const bool right_strip = last_strip && p.right != nullptr;   // this strip also returns the tile's right edge column
                const bool right_lane = right_strip && lane == lc;
                const u32 voffH = store_h ? j * (u32)sizeof(HT) : SY_OOB;
                const bool p8 = (p.p_bytes == 1);                 // compact predecessor matrix: one byte per cell
                const u32 voffP = store_p ? j * (p8 ? 1u : 4u) : SY_OOB;
                const int a_l = (lane >= 1 && jvalid) ? (int)seq_a[j - 1] : SY_ASENT;
                const int cz = gb + ngap * (int)j;
                const int topj = jvalid ? top_at(ls, lane, j) : 0;
                const int G0v = topj + (jvalid ? cz : 0);
                HT* H = (HT*)p.H;
                int32_t* P = p.P;
                if (ci == 0) {  // row 0: the halo row itself (P of a row owned by the tile above is left alone)
                    if (store_h) H[j] = (HT)topj;
                    if (store_p && !has_top) { if (p8) ((signed char*)P)[j] = 0; else P[j] = 0; }
                }
                if (ci == 0 && right_lane) p.right[0] = topj;
                // arg-max: best value of my column and where it first occurred.  With H in HBM only the 16-row block is
                // noted and its exact row is re-read at the end (free inside the loop); without H (score-only, P-only)
                // the row is resolved on the spot from the ring values, but only when some lane beats the wave's best so far
                int bestv = 0, bestblk = 0, bestrow = 0, wbest = 0;
                const bool reread = (p.H != nullptr);
                const u32 a_lu = (u32)a_l, mm_v = (u32)mm, xm_v = (u32)xm, ngap_v = (u32)ngap;
                const int nblk = (rows + SY_U - 1) / SY_U;
                const u32* myring = (const u32*)&lds.ring[ls][lane * SY_LSTR];
                int snap_prod = 0;
                Spin spin;
                const int slot = ci;   // progress counters: one slot per consumer
                // every load issued so far (a, top) has landed before the loop: inside it, a wait for one of them
                // would be a vmcnt(0), i.e. a wait for all the H/P stores of the previous block as well
                __builtin_amdgcn_s_waitcnt(0x0F70);
                // My blocks are ci, ci+NC, ...  The first block of the strip (row 0 comes from the halo row) and the last
                // one (partial rows, band / tile exports) take the compiler-scheduled path below; everything between runs in
                // ONE asm statement (consumer_loop32) in the common configuration: int32 H stored, no right-edge output.
                bool looped = false;
                for (int q = ci; q < nblk; q += NC) {
                    if constexpr (sizeof(HT) == 4) {
                        if (!looped && q > 0 && q < nblk - 1 && !right_strip && !(p.debug_flags & (128 | 256))) {
                            looped = true;
                            const int qs = __builtin_amdgcn_readfirstlane(q);   // (wave-uniform by construction; say so)
                            const int nmine = (nblk - 1 - qs + NC - 1) / NC;    // my blocks below the last one
                            const int qend = qs + nmine * NC;
                            const int r0 = qs * SY_U + 1;
                            u32 z = (u32)(cz + ngap * (r0 - 1)), E = (u32)(r0 - 2 + lane + phi);
                            u32 bv = (u32)bestv, bb = (u32)(reread ? bestblk : bestrow);
                            const uint64_t bH = (uint64_t)(uintptr_t)(H + (int64_t)r0 * M);
                            const uint64_t bP = (uint64_t)(uintptr_t)(p8 ? (char*)P + (int64_t)r0 * M : (char*)(P + (int64_t)r0 * M));
                            const sw_i32x4 dH = {(int)(u32)bH, (int)(u32)(bH >> 32), 0x7FFFFF00, 0x00020000};
                            const sw_i32x4 dP = {(int)(u32)bP, (int)(u32)(bP >> 32), 0x7FFFFF00, 0x00020000};
                            const u32 rowH = (u32)(M * 4), rowP = (u32)(M * (p8 ? 1 : 4));
                            const uint64_t blkH = (uint64_t)M * 4u * (uint64_t)(SY_U * NC), blkP = (uint64_t)M * (p8 ? 1u : 4u) * (uint64_t)(SY_U * NC);
                            const u32 lanebase = (u32)(size_t)&lds.ring[ls][lane * SY_LSTR];
                            const u32 zstep = (u32)(ngap * SY_U * (NC - 1));
                            int st;
#define SW_CLOOP(NTV, P8V)                                                                                                              \
    st = consumer_loop32<NTV, P8V>(z, E, bv, bb, lanebase, a_lu, mm_v, xm_v, ngap_v, voffP, voffH, zstep, (u32)(SY_U * NC), (u32)(SY_R - 1), dH, \
                                   dP, blkH, blkP, rowP, rowH, seq_b + (r0 - 1), (u32)(size_t)&lds.prod_u[ls],                            \
                                   (u32)(size_t)&lds.cons_blk[ls][slot], qs, qend, NC, r0 + SY_U - 1 + SY_W + phi, r0 - 2 + phi, __builtin_amdgcn_readfirstlane(reread ? 1 : 0))
                            if (p.store_nt) { if (p8) SW_CLOOP(true, true); else SW_CLOOP(true, false); }
                            else { if (p8) SW_CLOOP(false, true); else SW_CLOOP(false, false); }
#undef SW_CLOOP
                            if (st) {
                                __hip_atomic_store((gu32*)p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                return;
                            }
                            bestv = (int)bv;
                            if (reread) bestblk = (int)bb; else { bestrow = (int)bb; wbest = 0; }   // (wbest: the glue path's shortcut restarts)
                            q = qend - NC;   // the loop's increment lands on my next block
                            continue;
                        }
                    }
                    const int qu = __builtin_amdgcn_readfirstlane(q);   // wave-uniform by construction; say so (scalar operands below)
                    const int r0 = qu * SY_U + 1;
                    const int nb = min(SY_U, rows - r0 + 1);
                    const int need = r0 + nb - 1 + SY_W + phi;  // the local step that completes row r0+nb-1
                    if (snap_prod < need)
                        while ((snap_prod = lds_load(&lds.prod_u[ls])) < need)
                            if (spin.fail(p.abort_flag)) return;
                    asm volatile("" ::: "memory");
                    // rows r0-1 .. r0+15 along the skew: row r of lane l is step r+l+phi, ring entry (step-1) mod R
                    const u32 e0 = (u32)(r0 - 2 + lane + phi);
                    u32 gv[SY_U + 1];
                    if ((((u32)(r0 - 2 + phi) & (SY_R - 1)) + 63 + SY_U) < (u32)SY_R) {
                        const u32* q0 = myring + (e0 & (SY_R - 1));  // no lane wraps inside this block
#pragma unroll
                        for (int k = 0; k <= SY_U; ++k) gv[k] = __hip_atomic_load(q0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
#pragma unroll
                        for (int k = 0; k <= SY_U; ++k)
                            gv[k] = __hip_atomic_load(myring + ((e0 + k) & (SY_R - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    snap_prod = lds_load(&lds.prod_u[ls]);  // looked at again one block later
                    if (r0 == 1) gv[0] = (u32)G0v;
                    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)(H + (int64_t)r0 * M), 0, 0x7FFFFF00, 0x00020000);
                    char* const Prow = p8 ? (char*)P + (int64_t)r0 * M : (char*)(P + (int64_t)r0 * M);
                    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)Prow, 0, 0x7FFFFF00, 0x00020000);
                    const u32 rowH = (u32)(M * (int64_t)sizeof(HT)), rowP = (u32)(M * (p8 ? 1 : 4));
                    auto store_row = [&](int k, u32 h, u32 pr) {
                        if constexpr (sizeof(HT) == 8) {
                            typedef int v2i __attribute__((ext_vector_type(2)));
                            v2i hv; hv.x = (int)h; hv.y = (int)h >> 31;
                            __builtin_amdgcn_raw_buffer_store_b64(hv, rH, voffH, (int)(rowH * (u32)k), 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32((int)h, rH, voffH, (int)(rowH * (u32)k), 0);
                        }
                        if (p8) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)pr, rP, voffP, (int)(rowP * (u32)k), 0);
                        else __builtin_amdgcn_raw_buffer_store_b32((int)pr, rP, voffP, (int)(rowP * (u32)k), 0);
                        if (right_lane) p.right[r0 + k] = (int)h;   // the tile's right edge column, for the next tile of this band
                    };
                    u32 blkmax = 0;
                    if (nb == SY_U && !right_strip) {
                        // the common case: whole block in one asm statement
                        u32 z = (u32)(cz + ngap * (r0 - 1));
                        const uint64_t bH = (uint64_t)(uintptr_t)(H + (int64_t)r0 * M), bP = (uint64_t)(uintptr_t)Prow;
                        const sw_i32x4 dH = {(int)(u32)bH, (int)(u32)(bH >> 32), 0x7FFFFF00, 0x00020000};
                        const sw_i32x4 dP = {(int)(u32)bP, (int)(u32)(bP >> 32), 0x7FFFFF00, 0x00020000};
                        const unsigned char* ch = seq_b + (r0 - 1);
                        auto run = [&](auto NT, auto P8) {
                            if constexpr (sizeof(HT) == 8)
                                consumer_block16_h64<decltype(NT)::value, decltype(P8)::value>(gv, z, blkmax, a_lu, mm_v, xm_v, ngap_v, voffP, voffH, dH, dP, rowP, rowH, ch);
                            else
                                consumer_block16<decltype(NT)::value, decltype(P8)::value>(gv, z, blkmax, a_lu, mm_v, xm_v, ngap_v, voffP, voffH, dH, dP, rowP, rowH, ch);
                        };
                        if (p.store_nt) { if (p8) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
                        else { if (p8) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }
                    } else if (nb == SY_U) {
                        const uint4 w = *reinterpret_cast<const uint4*>(seq_b + (r0 - 1));  // this block's 16 row characters
                        const u32 bw[4] = {(u32)__builtin_amdgcn_readfirstlane((int)w.x), (u32)__builtin_amdgcn_readfirstlane((int)w.y),
                                           (u32)__builtin_amdgcn_readfirstlane((int)w.z), (u32)__builtin_amdgcn_readfirstlane((int)w.w)};
                        u32 z = (u32)(cz + ngap * (r0 - 1));
                        sfor<0, 4>([&](auto Q) {
                            constexpr int k = Q.value * 4;
                            const Rows4 o = consumer_rows4(gv[k], gv[k + 1], gv[k + 2], gv[k + 3], gv[k + 4], z, blkmax, a_lu, mm_v, xm_v,
                                                           ngap_v, bw[Q.value]);
                            store_row(k, o.h0, o.p0); store_row(k + 1, o.h1, o.p1); store_row(k + 2, o.h2, o.p2); store_row(k + 3, o.h3, o.p3);
                        });
                    } else {  // the last, partial block of the matrix: plain per-row code
                        for (int k = 0; k < nb; ++k) {
                            int g = 0, U = 0;
                            sfor<0, SY_U>([&](auto K) { if (K.value == k) { g = (int)gv[K.value + 1]; U = (int)gv[K.value]; } });
                            const int b_i = (int)seq_b[r0 - 1 + k];
                            const int D = __builtin_amdgcn_update_dpp(0, U, 0x138, 0xF, 0xF, false);  // G[r-1][j-1]
                            const int dd = D + ((a_l == b_i) ? mm : xm);
                            const int zz = cz + ngap * (r0 + k);
                            const int h = g - zz;
                            const int pred = (g == zz) ? 0 : (dd == g) ? 3 : (U == g) ? 1 : 2;  // serial_smithW.c:204-234
                            store_row(k, (u32)h, (u32)pred);
                            blkmax = (u32)max((int)blkmax, h);
                        }
                    }
                    if (reread) {
                        if ((int)blkmax > bestv) { bestv = (int)blkmax; bestblk = qu; }
                    } else {
                        const int bm = cell_ok ? (int)blkmax : 0;
                        if (__builtin_amdgcn_ballot_w64(bm > wbest) != 0) {
                            if (bm > bestv) {
                                bestv = bm;
                                int row = r0;
                                sfor<0, SY_U>([&](auto K) {
                                    constexpr int k = SY_U - 1 - K.value;   // descending: the lowest row wins
                                    if (k < nb && (int)gv[k + 1] - (cz + ngap * (r0 + k)) == bm) row = r0 + k;
                                });
                                bestrow = row;
                            }
                            int v = cell_ok ? bestv : 0;   // (pad lanes may carry block maxima from the asm loop)
                            for (int off = 32; off; off >>= 1) v = max(v, __shfl_xor(v, off));
                            wbest = __builtin_amdgcn_readfirstlane(v);
                        }
                    }
                    if (p.dbg && (p.debug_flags & 128) && lane == 0)   // per-block completion stamps (after the strip stamps and counters)
                        p.dbg[6 * p.nstrips + 64 + (int64_t)s * nblk + qu] = __builtin_amdgcn_s_memrealtime();
                    if (p.bot_gran && qu == nblk - 1) {
                        // band-resident launch: the band's last row leaves as {tag, H} granules (the next band's halo row)
                        u32 gl = 0;
                        sfor<0, SY_U>([&](auto K) { if (K.value + 1 == nb) gl = gv[K.value + 1]; });
                        const int hl = (int)gl - (cz + ngap * rows);
                        if (cell_ok)
                            __hip_atomic_store((gu64*)(p.bot_gran + j), ((u64)p.bot_tag << 32) | (u64)(u32)hl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (p.bot_done) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (lane == 0) __hip_atomic_store((gu32*)(p.bot_done + s), p.bot_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                    lds_store(&lds.cons_blk[ls][slot], qu);
                }
                // arg-max: the lowest row of block `bestblk` holding bestv in my column (re-read what this wave stored)
                if (cell_ok && !(p.debug_flags & 1) && bestv > 0) {
                    if (reread) {
                        bestrow = bestblk * SY_U + 1;
                        __builtin_amdgcn_s_waitcnt(0);  // my own stores have reached L2
                        for (int k = SY_U - 1; k >= 0; --k) {
                            const int r = bestblk * SY_U + 1 + k;
                            if (r <= rows && (int)__builtin_nontemporal_load(&H[(int64_t)r * M + j]) == bestv) bestrow = r;
                        }
                    }
                    const u64 idx = (u64)bestrow * (u64)M + (u64)j;
                    atomicMax(p.result_key, ((u64)(u32)bestv << 40) | (SW_KEY_IDX_MASK - idx));
                }
            }
        } else if (role != R_IDLE) {
            // ================================= helper ==================================
            // import: edge column of strip s0-1 (HBM granules; column 0 / row 0 synthesised) -> lds.halo,
            //         indexed by strip s0's local step: row t lives at entry (t + phi0 - 1) mod RH
            // export: lane-63 column of the group's last strip -> HBM granules for the next group
            const int slast = s0 + nact - 1;
            __builtin_amdgcn_s_setprio(2);  // short, latency-critical loops: ahead of the consumers on their SIMD
            if (perm) {
                // ---- perm path: the left group's last producer stored its lane-63 results itself, as 4-byte values that
                // carry the launch tag, indexed by ITS local step; my strip's local step u needs its step u + 64.
                // There is no exporter.  Steps above the matrix (rows <= 0) hold the row-0 value of the halo column.
                // (debug bit 12: a single importer wave)
                const bool importer_w = (p.debug_flags & 4096) ? (role == R_EXP) : (role == R_IMP || role == R_EXP);   // nothing to export here: one more importer
                if (importer_w) {
                    const int phi0 = phi_of(s0, phib);
                    const u32 row0v = (u32)(gb + top_at(0, 0, (int64_t)s0 * SY_W) + ngap * s0 * SY_W);
                    // Steps whose halo row lies BELOW the matrix (the other lanes are still inside it) get the launch bias, a
                    // value below every floor: the producer must never meet left-over LDS contents there.  Round 2 found what
                    // happens otherwise: such garbage rode through the (unused) cells below the matrix into the exported
                    // values, grew past 24 bits, and with the tag byte bumped by one was accepted by the NEXT launch.
                    const int ulast = u_total(s0) + 64;            // every local step the producer may fetch a halo for
                    const u32 tag = (u32)gb >> 24;
                    const u32* e4 = (s0 > 0) ? p.edge4 + (int64_t)(s0 - 1) * p.e4stride : nullptr;
                    int impu = 1;
                    bool wide = true;
                    Spin spin;
                    while (impu <= ulast) {
                        const int front = lds_load(&lds.halo_ready);
                        if (front > 1) impu = max(impu, front);
                        if (impu > ulast) break;
                        // steps < lim may be written.  Every valid write also zeroes the slot half a ring further on (the producer
                        // tells by the zero that a slot has not been filled yet), so only half of the ring holds unconsumed steps.
                        int lim = min(ulast + 1, lds_load(&lds.prod_u[0]) + SY_RH / 2 - 2 * SY_U);
                        int base = impu;
                        auto windows = [&](auto NB) {
                            constexpr int nbat = decltype(NB)::value;
                            u32 vals[nbat];
                            bool oks[nbat];
#pragma unroll
                            for (int b4 = 0; b4 < nbat; ++b4) {
                                const int u = impu + b4 * 64 + lane;
                                const int r = u - phi0;
                                u32 v = row0v;
                                bool ok = true;
                                if (r > rows) v = (u32)gb;
                                else if (r > 0) {
                                    if (s0 == 0) v = (u32)(gb + ngap * r + (p.left ? p.left[r] : 0));
                                    else {
                                        v = __hip_atomic_load((gu32*)(e4 + u + 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        ok = (v >> 24) == tag;
                                    }
                                }
                                vals[b4] = v;
                                oks[b4] = ok && (u < lim);
                            }
                            const int front2 = lds_load(&lds.halo_ready);
#pragma unroll
                            for (int b4 = 0; b4 < nbat; ++b4) {
                                const int u = impu + b4 * 64 + lane;
                                const u64 okm = __ballot(oks[b4]);
                                const int npre = (okm == ~0ull) ? 64 : __builtin_ctzll(~okm);
                                if (base == impu + b4 * 64 && npre > 0) {
                                    if (lane < npre && (front2 <= 1 || u >= front2)) {
                                        __hip_atomic_store(&lds.halo[(u - 1) & (SY_RH - 1)], vals[b4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        __hip_atomic_store(&lds.halo[(u - 1 + SY_RH / 2) & (SY_RH - 1)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        if (p.dbg && (p.debug_flags & 512) && u < 256) p.dbg[1024 + (int64_t)s0 * 1024 + u] = ((u64)(u32)wave << 32) | vals[b4];
                                    }
                                    base += npre;
                                }
                            }
                        };
                        if (wide) windows(std::integral_constant<int, 4>{});
                        else windows(std::integral_constant<int, 1>{});
                        wide = (base - impu) >= 64;
                        if (base > impu) {
                            asm volatile("" ::: "memory");
                            if (p.dbg && lane == 0 && impu - phi0 <= rows / 2 && base - phi0 > rows / 2) p.dbg[2 * p.nstrips + 8 + 2 * grp] = __builtin_amdgcn_s_memrealtime();
                            impu = base;
                            __hip_atomic_fetch_max(&lds.halo_ready, impu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else if (spin.fail(p.abort_flag)) {
                            return;
                        }
                    }
                    lds_store(&lds.halo_ready, 0x7fffffff);
                }
                __builtin_amdgcn_s_setprio(0);
            } else {
            const bool do_export = (slast + 1 < p.nstrips);
            const int phi0 = phi_of(s0, phib), phil = phi_of(slast, phib);
            const u32 halo_row0 = (u32)(top_at(0, 0, (int64_t)s0 * SY_W) + ngap * s0 * SY_W);
            // import starts at row 0 (the diagonal neighbour of row 1); with the fast producers at the row of
            // strip s0's local step 1: the cells above the matrix hold the H == 0 floor there
            int imp = (phib >= 0) ? 1 - phi0 : 0, exp = 1;
            bool imp_wide = true;
            u64 pace_t0 = 0;
            Spin spin;
            auto edge_val = [&](int r, bool& ok) -> u32 {  // G of (row r, column 63*s0): the halo of strip s0
                ok = true;
                if (r <= 0) return (r == 0) ? halo_row0 : (u32)(ngap * (r + s0 * SY_W));  // above the matrix: H == 0 floor
                if (s0 == 0) return (u32)(ngap * r + (p.left ? p.left[r] : 0));           // column 0: H == 0, or the tile's left halo column
                const u64 gr = __hip_atomic_load((gu64*)(p.edge + (int64_t)(s0 - 1) * estride + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (gr >> 32) == (tag_base | (u64)r);
                return (u32)gr;
            };
            const bool importer = (role == R_IMP), exporter = (role == R_EXP);
            while ((importer && imp <= rows) || (exporter && do_export && exp <= rows)) {
                bool progressed = false;
                if (importer && imp <= rows) {
                    // with several importers: never work behind the published front (a wave that fell a ring length behind
                    // would overwrite slots the leader has refilled)
                    const int front = lds_load(&lds.halo_ready);
                    if (front > 1) imp = max(imp, front - phi0);
                    // halo[] holds RH steps and strip s0's producer has consumed every step <= its progress:
                    // rows < lim may be written.  One 64-row window per round trip to HBM/L2 while the importer
                    // rides the left neighbour's front (latency matters); four windows while it is catching up.
                    int lim = min(rows + 1, lds_load(&lds.prod_u[0]) - phi0 + SY_RH - 2 * SY_U);
                    if (s0 == 0 && p.pace_ps > 0) {
                        // pacing: strip 0 runs a few percent below full speed, so every later strip has headroom to
                        // win back what a late hand-off cost it (otherwise each hop keeps its worst delay for good)
                        const u64 now = __builtin_amdgcn_s_memrealtime();  // 100 MHz
                        if (pace_t0 == 0) pace_t0 = now;
                        lim = min(lim, 64 + (int)min((u64)0x7fffff00, (now - pace_t0) * 10000ull / (u64)p.pace_ps));
                    }
                    int base = imp;
                    auto import_windows = [&](auto NB) {
                        constexpr int nbat = decltype(NB)::value;
                        u32 vals[nbat];
                        bool oks[nbat];
                        if (s0 > 0) {
                            // branch-free: all granule loads in flight at once (clamped addresses), validity by selects
                            u64 gr[nbat];
                            const u64* eb = p.edge + (int64_t)(s0 - 1) * estride;
#pragma unroll
                            for (int b4 = 0; b4 < nbat; ++b4) {
                                const int r = imp + b4 * 64 + lane;
                                gr[b4] = __hip_atomic_load((gu64*)(eb + min(max(r, 1), rows)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
#pragma unroll
                            for (int b4 = 0; b4 < nbat; ++b4) {
                                const int r = imp + b4 * 64 + lane;
                                const bool tag_ok = (gr[b4] >> 32) == (tag_base | (u64)(u32)max(r, 1));
                                const u32 above = (r == 0) ? halo_row0 : (u32)(ngap * (r + s0 * SY_W));   // rows <= 0: H == 0 floor / halo row
                                vals[b4] = (r <= 0) ? above : (u32)gr[b4];
                                oks[b4] = (r < lim) && (r <= 0 || tag_ok);
                            }
                        } else {
#pragma unroll
                            for (int b4 = 0; b4 < nbat; ++b4) {
                                const int r = imp + b4 * 64 + lane;
                                bool ok = false;
                                vals[b4] = 0;
                                if (r < lim) vals[b4] = edge_val(r, ok);
                                oks[b4] = ok;
                            }
                        }
                        const int front2 = lds_load(&lds.halo_ready);   // rows below it are published (and may be consumed): leave them alone
#pragma unroll
                        for (int b4 = 0; b4 < nbat; ++b4) {
                            const int r = imp + b4 * 64 + lane;
                            const u64 okm = __ballot(oks[b4]);
                            const int npre = (okm == ~0ull) ? 64 : __builtin_ctzll(~okm);  // leading run of valid rows
                            if (base == imp + b4 * 64 && npre > 0) {                       // contiguous with what is imported
                                if (lane < npre && (front2 <= 1 || r + phi0 >= front2))
                                    __hip_atomic_store(&lds.halo[(r + phi0 - 1) & (SY_RH - 1)], vals[b4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                base += npre;
                            }
                        }
                    };
                    if (imp_wide) import_windows(std::integral_constant<int, 4>{});
                    else import_windows(std::integral_constant<int, 1>{});
                    imp_wide = (base - imp) >= 64;   // a full window came back: there may be more behind it
                    if (base > imp) {
                        asm volatile("" ::: "memory");  // LDS executes a wave's ops in order: data before counter
                        if (p.dbg && lane == 0 && imp <= rows / 2 && base > rows / 2) p.dbg[2 * p.nstrips + 8 + 2 * grp] = __builtin_amdgcn_s_memrealtime();
                        imp = base;
                        // local steps < imp+phi0 are in halo[] (a maximum: with two importers the slower one must not take it back)
                        __hip_atomic_fetch_max(&lds.halo_ready, imp + phi0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        progressed = true;
                    }
                }
                if (exporter && do_export && exp <= rows) {
                    // rows whose lane-63 value the last producer has written: step r+63+phi <= its progress
                    const int avail = lds_load(&lds.prod_u[nact - 1]) - SY_W - phil;
                    const int n = min(64, min(rows, avail) - exp + 1);
                    if (n > 0) {
                        asm volatile("" ::: "memory");
                        const int r = exp + lane;
                        const u32* e = (const u32*)&lds.ring[nact - 1][63 * SY_LSTR];
                        if (lane < n) {
                            const u32 v = __hip_atomic_load(e + ((r + SY_W + phil - 1) & (SY_R - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store((gu64*)(p.edge + (int64_t)slast * estride + r), ((tag_base | (u64)r) << 32) | (u64)v,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (p.dbg && lane == 0 && exp <= rows / 2 && exp + n > rows / 2) p.dbg[2 * p.nstrips + 9 + 2 * grp] = __builtin_amdgcn_s_memrealtime();
                        exp += n;
                        lds_store(&lds.exp_done, exp - 1);
                        progressed = true;
                    }
                }
                if (!progressed && spin.fail(p.abort_flag)) return;
            }
            __builtin_amdgcn_s_setprio(0);
            if (exporter && do_export) lds_store(&lds.exp_done, 0x7fffffff);
            if (importer) lds_store(&lds.halo_ready, 0x7fffffff);
            }
        }
        __syncthreads();
        if (p.dbg && (p.debug_flags & 512) && threadIdx.x < 256) {
            const u32* r0 = (const u32*)&lds.ring[0][0];
            p.dbg[1024 + (int64_t)s0 * 1024 + 256 + threadIdx.x] = r0[threadIdx.x];          // lane 0, steps 1..256 (ring slot = step-1)
            p.dbg[1024 + (int64_t)s0 * 1024 + 512 + threadIdx.x] = lds.halo[threadIdx.x];    // the halo ring itself
        }
    }
}


// Wipe the self-tagged edge buffer with the SAME kind of store the producers use (agent-scope, written through): the
// importers read it with agent-scope loads that are served from memory, and a plain memset's zeros can still sit in the
// writing XCD's L2 at that time -- stale words of an earlier owner of the memory whose top byte happens to equal the
// launch tag were then taken for data (seen as one wrong fill in ~200 around tags 0x38..0x40, the top bytes of floats).
// (16 bytes per store: the 4-byte version took 8 ms for the 70 MB of a 32768^2 fill -- once every 255 fills, but inside somebody's timing.)
__global__ void __launch_bounds__(256) sw_wipe_u32(unsigned int* __restrict__ buf, size_t n) {
    const size_t n4 = n / 4;
    const sw_i32x4p z4 = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(buf + 4 * i), "v"(z4) : "memory");
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store((gu32*)(buf + i), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Which XCD does workgroup i of a launch run on?  (sw_create checks that it is i % 8 before sw_systolic2 may rely on it.)
__global__ void sw_xcc_probe(unsigned int* xcc_of_block) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = xcc & 15u;
}

// ---- input preparation: two small kernels per fill (they used to be seven dispatches: two memsets + scan + coding + two row-0
// memsets + a column-0 kernel, 72 us of a 0.97 ms fill in round 2's trace) ---------------------------------------------------
// sw_prep_scan: which byte values occur in a and b (all pairs of a batch).  Every block writes ITS 256-bit presence map to
// part[block][8] -- no atomics, nothing to clear beforehand; the readers OR the `gridDim.x` maps together.
__global__ void __launch_bounds__(256) sw_prep_scan(const unsigned char* __restrict__ a, int64_t cols, int64_t a_pstride,
                                                    const unsigned char* __restrict__ b, int64_t rows, int64_t b_pstride, int64_t npairs,
                                                    unsigned int* __restrict__ part) {
    __shared__ unsigned int seen[8];
    if (threadIdx.x < 8) seen[threadIdx.x] = 0;
    __syncthreads();
    unsigned int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t per = cols + rows, total = per * npairs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pair = i / per, k = i - pair * per;
        const unsigned char ch = k < cols ? a[pair * a_pstride + k] : b[pair * b_pstride + (k - cols)];
#pragma unroll
        for (int w = 0; w < 8; ++w) mine[w] |= ((ch >> 5) == w) ? (1u << (ch & 31)) : 0u;
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) if (mine[w]) atomicOr(&seen[w], mine[w]);
    __syncthreads();
    if (threadIdx.x < 8) part[blockIdx.x * 8 + threadIdx.x] = seen[threadIdx.x];
}

// ORs the npart presence maps of sw_prep_scan into the first one (one block of 256 threads).
__global__ void __launch_bounds__(256) sw_prep_reduce(unsigned int* part, int npart) {
    __shared__ unsigned int acc[256];
    const int w = threadIdx.x & 7;
    unsigned int m = 0;
    for (int k = threadIdx.x >> 3; k < npart; k += 32) m |= part[k * 8 + w];
    acc[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < 8) {
        for (int k = 1; k < 32; ++k) m |= acc[k * 8 + w];
        part[w] = m;
    }
}

// the letter-code table from the partial presence maps: tab[v] = rank of byte value v among the values present (`maxcode` letters at
// most, else `pad` for every value); returns the number of letters.  All 256 threads of the block take part.
__device__ __forceinline__ int sw_code_table(const unsigned int* __restrict__ part, int npart, unsigned char* tab, int maxletters, unsigned char pad) {
    __shared__ unsigned int present[8];
    if (threadIdx.x < 8) {
        unsigned int m = 0;
        for (int k = 0; k < npart; ++k) m |= part[k * 8 + threadIdx.x];
        present[threadIdx.x] = m;
    }
    __syncthreads();
    const int t = threadIdx.x;
    int rank = 0, nletters = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const unsigned int m = present[w];
        nletters += __popc(m);
        if (w < (t >> 5)) rank += __popc(m);
        else if (w == (t >> 5)) rank += __popc(m & ((1u << (t & 31)) - 1u));
    }
    const bool here = (present[t >> 5] >> (t & 31)) & 1u;
    tab[t] = (here && nletters <= maxletters) ? (unsigned char)rank : pad;
    __syncthreads();
    return nletters;
}

// sw_prep_code: blocks x < npad of pair y: bpad[front + i] = b[i] (bytes, zero padded), bpad16 (16-bit, padded with the never-matching
// 0x100), bcode (letter codes 0..6, 7 outside the sequence and for alphabets of more than 7 letters) -- producer lane l reads
// b[u-phi-l-1] for steps that reach phi+63 rows above and ~200 rows below the matrix (those cells are never stored).  Block (0,0)
// also publishes the code table atab and zeroes the arg-max key.  Blocks x >= npad (pair 0 only): row 0 and column 0 of the
// matrices the two-column kernel does not write itself (H int32 / int64, P int32 / int8; either may be NULL; skip_row0: a band's
// row 0 is its halo row -- H comes from the kernel, P belongs to the band above).
__global__ void __launch_bounds__(256) sw_prep_code(const unsigned char* __restrict__ b, int64_t rows, int64_t front, int64_t b_pstride,
                                                    unsigned char* __restrict__ bpad, unsigned short* __restrict__ bpad16, unsigned char* __restrict__ bcode,
                                                    const unsigned int* __restrict__ part, int npart, unsigned char* __restrict__ atab, int64_t per, int npad,
                                                    void* H, int h_bytes, void* P, int p_bytes, int64_t M, int64_t rows1, int skip_row0,
                                                    unsigned long long* key) {
    if ((int)blockIdx.x >= npad) {
        const int64_t e = ((int64_t)blockIdx.x - npad) * blockDim.x + threadIdx.x;   // element of [row 0 | column 0]
        if (e < M && !skip_row0) {
            if (H) { if (h_bytes == 8) ((int64_t*)H)[e] = 0; else ((int32_t*)H)[e] = 0; }
            if (P) { if (p_bytes == 1) ((signed char*)P)[e] = 0; else ((int32_t*)P)[e] = 0; }
        } else if (e >= M && e - M < rows1 - 1) {
            const int64_t r = e - M + 1;
            if (H) { if (h_bytes == 8) ((int64_t*)H)[r * M] = 0; else ((int32_t*)H)[r * M] = 0; }
            if (P) { if (p_bytes == 1) ((signed char*)P)[r * M] = 0; else ((int32_t*)P)[r * M] = 0; }
        }
        return;
    }
    __shared__ unsigned char tab[256];
    const int nletters = sw_code_table(part, npart, tab, 7, (unsigned char)7);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        atab[threadIdx.x] = tab[threadIdx.x];
        if (threadIdx.x == 0) *(unsigned int*)(atab + 256) = (unsigned int)nletters;
        if (key && threadIdx.x < 2) key[threadIdx.x] = 0ull;
        ((unsigned int*)(atab + SW_XTAB_OFF))[threadIdx.x] = 0u;   // sw_systolic2, xcd_mode: no workgroup of the coming launch has said where it runs
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pair = blockIdx.y;  // batch: one padded copy per problem
    if (i < per) {
        const bool in = (i >= front && i - front < rows);
        const unsigned char ch = in ? b[pair * b_pstride + i - front] : (unsigned char)0;
        bpad[pair * per + i] = ch;
        bpad16[pair * per + i] = in ? (unsigned short)ch : (unsigned short)0x100;  // 0x100 never equals a character
        bcode[pair * per + i] = in ? tab[ch] : (unsigned char)7;
    }
}

#include "sw_systolic2.inc"

template <typename HT, int NS, int NC>
__global__ void __launch_bounds__(NS == 1 ? 768 : 64 * (NS * (1 + NC) + 2))   // NS == 1: 12 waves, SIMD 0 belongs to the producer (3 waves per SIMD: 168 VGPRs)
sw_systolic(const unsigned char* seq_a, const unsigned char* seq_b, const unsigned char* bpad, FillParams p) {
    // enqueued behind sw_systolic2 (skip_if_perm): that kernel has done the fill unless the alphabet (found on the device, by its prologue)
    // has more than 7 letters -- then this one fills, from the shared padded copies workgroup 0 of that launch left, and reports like it
    if (p.skip_if_perm && p.gbias != 0 && *(const unsigned int*)(p.atab + 256) <= 7u && !(p.debug_flags & 16)) return;
    sw_systolic_body<HT, NS, NC>(seq_a, seq_b, bpad, p);
    if (p.skip_if_perm && p.sync) s2_epilogue(p, true);
}
#define SW_INST(NS, NC)                                                                                                   \
    template __global__ void sw_systolic<int32_t, NS, NC>(const unsigned char*, const unsigned char*, const unsigned char*, FillParams); \
    template __global__ void sw_systolic<int64_t, NS, NC>(const unsigned char*, const unsigned char*, const unsigned char*, FillParams);
SW_INST(2, 2)
SW_INST(2, 3)
SW_INST(2, 4)
SW_INST(1, 2)
SW_INST(1, 3)
SW_INST(1, 4)
SW_INST(1, 6)
SW_INST(1, 7)
#undef SW_INST

}  // namespace swk
