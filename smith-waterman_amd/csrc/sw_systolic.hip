// sw_systolic.hip -- the wavefront fill as a systolic producer/consumer pipeline (gfx950).
//
// Measured on MI355X (profiles/r01_ubench_issue_latency.log): one wave issues INDEPENDENT VALU
// ops every ~1.15 clk but a DEPENDENT op only every ~4.5 clk (~12.5 clk when the dependence goes
// through a DPP lane shift).  The DP recurrence is one long dependence chain, so the fill is
// latency-bound and the design goal is: as few dependent ops per anti-diagonal step as possible,
// everything else issued in their shadow or moved to other waves.
//
// G-space (see sw_kernels.hip):  G = H - gap*(row+col);  Z = -gap*(row+col) is the H==0 floor and
// is the SAME for all cells of one anti-diagonal.
//
// Strip s = matrix columns 63*s .. 63*s+63.  Lane l owns column 63*s+l; lane 0 is the strip's
// left halo column (= lane 63 of strip s-1), so every lane finds its left/diagonal neighbour
// one lane down.  At step t lane l works on row r = t - l, i.e. one wave sweeps an
// anti-diagonal down the strip:
//     m = max(G1[l-1], G1[l])          v_max_i32_dpp wave_shr:1     (G1 = values of step t-1)
//     d = G2[l-1] + s'                 v_add_u32_dpp wave_shr:1     (G2 = step t-2; off the chain)
//     g = max3(d, m, Z_t)              v_max3_i32                    -> 2 dependent ops per step
// (signed: G >= 0 always, but d can dip below 0 when mismatch - 2*gap < 0)
// Lane 0 is never written by the two DPP ops (no source lane), so m[0] stays 0 and d[0] is
// preloaded with the halo value of row t: g[0] = halo(t) falls out of the same max3.
//
// Roles inside one workgroup (NS strips): NS producer waves run the recurrence only and write
// each step's 64 values to an LDS ring (diagonal-major); NS*NC consumer waves read the ring
// row-major (skewed addresses, conflict-free), derive H and P and write them to HBM with one
// coalesced 252-byte store per matrix row; one helper wave moves the strip-edge column between
// workgroups through HBM/L2 as {tag,value} granules.  All intra-workgroup hand-offs are LDS
// counters written in order behind the data they cover.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sw_kernels.h"

namespace swk {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) u32 gu32;

#ifndef SW_RING
#define SW_RING 256
#endif
constexpr int SY_R = SW_RING;  // ring slots (steps) per strip: 256 B each
constexpr int SY_U = 16;     // steps per producer block / rows per consumer block
constexpr int SY_RH = 256;   // imported-halo ring entries (rows)
constexpr int SY_W = 63;     // real columns per strip
constexpr u32 SY_OOB = 0xFFFFFF00u;
constexpr int SY_ASENT = 0x200;  // never-matching character for lanes without a column

template <int NS, int NC>
struct SysLds {
    u32 ring[NS][SY_R][64];
    u32 halo[SY_RH];
    __attribute__((aligned(16))) int cons_blk[NS][4];  // latest completed 16-row block per consumer (-1: none; unused: INT_MAX)
    int prod_t[NS];        // completed steps of each producer
    int halo_ready;        // imported halo rows: every row < halo_ready is in halo[]
    int exp_done;          // exported edge rows: every row <= exp_done is in HBM
    int never;             // INT_MAX: "neighbour" of a strip nobody waits behind
};

// counters are wave-uniform by construction; readfirstlane makes that provable, so every poll
// loop is a scalar branch and the SGPR state of the producer's asm stays in SGPRs
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
    __device__ __forceinline__ bool fail(unsigned int* abort_flag) {
        __builtin_amdgcn_s_sleep(1);
        if ((++n & 127u) != 0) return false;
        const uint64_t now = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        if (t0 == 0) t0 = now;
        const bool expired = (now - t0) > 300000000ull;
        if (expired) __hip_atomic_store((gu32*)abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// One anti-diagonal step of a producer, hand-scheduled (hipcc pads no hazards inside asm):
//  * v_cmp(sdwa) -> v_cndmask via VCC and v_readlane -> v_writelane via an SGPR: 2 instructions apart
//    (gfx940+: a VALU write of an SGPR/VCC needs 2 wait states before a VALU read)
//  * a DPP source VGPR must have been written >= 2 instructions earlier: G1 (= previous step's g)
//    is followed by v_add/ds_write/v_cmp/v_readlane, G2 is two steps old.
// C holds 4 of this lane's next row characters; byte K&3 is the character of the row this lane
// works on at this step (each lane walks b[] at its own offset, so nothing has to be shifted).
#define SW_PRODUCER_STEP_ASM(BYTE)                                                                  \
    asm volatile(                                                                                   \
        "v_cmp_eq_u32_sdwa vcc, %[a], %[C] src0_sel:DWORD src1_sel:" BYTE "\n\t"                    \
        "v_readlane_b32 %[sh], %[Hv], %[k]\n\t"                                                     \
        "v_max_i32_dpp %[m], %[G1], %[G1] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                \
        "v_cndmask_b32 %[sp], %[xm], %[mm], vcc\n\t"                                                \
        "v_writelane_b32 %[d], %[sh], 0\n\t"                                                        \
        "v_add_u32_dpp %[d], %[G2], %[sp] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                \
        "v_max3_i32 %[g], %[d], %[m], %[Z]\n\t"                                                     \
        "v_add_u32 %[Z], %[Z], %[ngap]\n\t"                                                         \
        "ds_write_b32 %[waddr], %[g] offset:%[off]\n\t"                                             \
        : [sh] "=&s"(sh), [m] "+v"(m), [d] "+v"(d), [g] "=&v"(g), [Z] "+v"(Z), [sp] "=&v"(sp)       \
        : [Hv] "v"(Hv), [G1] "v"(G1), [G2] "v"(G2), [C] "v"(C), [a] "v"(a_l), [ngap] "v"(ngap_v),   \
          [waddr] "v"(waddr), [xm] "v"(xm_v), [mm] "v"(mm_v), [k] "n"(K & 15), [off] "n"((K & 15) * 256) \
        : "vcc", "memory")

template <int K>
__device__ __forceinline__ u32 producer_step(u32 G1, u32 G2, u32& m, u32& d, u32 Hv, u32 a_l, u32 C, u32& Z,
                                             u32 ngap_v, u32 xm_v, u32 mm_v, u32 waddr) {
    u32 g, sp;
    int sh;
    if constexpr ((K & 3) == 0) SW_PRODUCER_STEP_ASM("BYTE_0");
    if constexpr ((K & 3) == 1) SW_PRODUCER_STEP_ASM("BYTE_1");
    if constexpr ((K & 3) == 2) SW_PRODUCER_STEP_ASM("BYTE_2");
    if constexpr ((K & 3) == 3) SW_PRODUCER_STEP_ASM("BYTE_3");
    return g;
}

// ---- producer-side counter snapshots ------------------------------------------------------------
// Polling an LDS counter costs a ~64-clk round trip; done at every block boundary it would cost
// more than the 16 steps themselves.  So the producer issues its reads (left/right neighbour
// progress, its consumers' progress, next block's halo values) in the MIDDLE of a block and picks
// the results up at the next block boundary with a counted s_waitcnt: exactly 9 younger LDS ops
// follow the reads (8 ring writes + the progress store; lgkmcnt is 4 bits on gfx9).  The landing
// registers are literal v120..v126, which no compiler value ever uses (tools/check_isa.py), so the
// asynchronously landing data cannot hit a live register (cdna_hip_programming.md 5.7 item 1).
__device__ __forceinline__ void snap_issue(u32 left_addr, u32 cons_addr, u32 right_addr) {
    asm volatile("ds_read_b32 v124, %0\n\tds_read_b128 v[120:123], %1\n\tds_read_b32 v125, %2"
                 :: "v"(left_addr), "v"(cons_addr), "v"(right_addr)
                 : "memory", "v120", "v121", "v122", "v123", "v124", "v125");
}
__device__ __forceinline__ void halo_issue(u32 addr) {
    asm volatile("ds_read_b32 v126, %0" :: "v"(addr) : "memory", "v126");
}
struct Snap { int left, c0, c1, c2, c3, right; u32 hv; };
__device__ __forceinline__ Snap snap_collect() {
    Snap r;
    asm volatile("s_waitcnt lgkmcnt(9)\n\t"
                 "v_readfirstlane_b32 %0, v124\n\t"
                 "v_readfirstlane_b32 %1, v120\n\t"
                 "v_readfirstlane_b32 %2, v121\n\t"
                 "v_readfirstlane_b32 %3, v122\n\t"
                 "v_readfirstlane_b32 %4, v123\n\t"
                 "v_readfirstlane_b32 %5, v125\n\t"
                 "v_mov_b32 %6, v126"
                 : "=s"(r.left), "=s"(r.c0), "=s"(r.c1), "=s"(r.c2), "=s"(r.c3), "=s"(r.right), "=v"(r.hv)
                 :: "memory");
    return r;
}

template <typename HT, int NS, int NC>
__global__ void __launch_bounds__(64 * (NS * (1 + NC) + 1))
sw_systolic(const unsigned char* __restrict__ seq_a, const unsigned char* __restrict__ seq_b,
            const unsigned char* __restrict__ bpad, FillParams p) {
    __shared__ SysLds<NS, NC> lds;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t M = p.M;
    const int rows = (int)p.rows;
    const int ngap = p.ngap, mm = p.mm, xm = p.xm;
    const int T_total = (rows + SY_W + SY_U - 1) / SY_U * SY_U;  // steps 1..T_total
    const int ngroups = (p.nstrips + NS - 1) / NS;
    const u64 tag_base = p.tag_base;
    const int64_t estride = p.rows + 1;

    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        if (threadIdx.x < NS) lds.prod_t[threadIdx.x] = 0;
        if (threadIdx.x < NS * 4) lds.cons_blk[threadIdx.x / 4][threadIdx.x % 4] = ((int)(threadIdx.x % 4) < NC) ? -1 : 0x7fffffff;
        if (threadIdx.x == 0) { lds.halo_ready = 1; lds.exp_done = 0; lds.never = 0x7fffffff; }
        __syncthreads();

        const int s0 = grp * NS;
        const int nact = min(NS, p.nstrips - s0);  // active strips of this group

        if (wave < NS) {
            // ================================ producer ================================
            const int ls = wave, s = s0 + ls;
            if (ls < nact) {
                const u32 j = (u32)s * SY_W + (u32)lane;
                const bool jvalid = (int64_t)j < M;
                const u32 a_l = (lane >= 1 && jvalid) ? (u32)seq_a[j - 1] : (u32)SY_ASENT;
                const u32 G0v = jvalid ? (u32)((p.top ? p.top[j] : 0) + ngap * (int)j) : 0u;  // row 0 in G-space
                u32 G1 = (lane == 0) ? G0v : 0u, G2 = 0u, m = 0u, d = 0u;
                u32 Z = (u32)(ngap * (1 + s * SY_W));
                const u32 mm_v = (u32)mm, xm_v = (u32)xm, ngap_v = (u32)ngap;
                const bool lefthalo = (ls == 0);         // halo comes from the import ring
                const bool has_right = (ls + 1 < nact);  // another producer reads my lane-63 column
                const bool has_export = (ls + 1 == nact) && (s + 1 < p.nstrips);
                const u32 ringbase = (u32)(size_t)&lds.ring[ls][0][0];
                const int* left_cnt = lefthalo ? &lds.halo_ready : &lds.prod_t[ls - 1];
                const int* right_cnt = has_right ? &lds.prod_t[ls + 1] : has_export ? &lds.exp_done : &lds.never;
                const u32 left_addr = (u32)(size_t)left_cnt, right_addr = (u32)(size_t)right_cnt;
                const u32 cons_addr = (u32)(size_t)&lds.cons_blk[ls][0];

                auto halo_ptr = [&](int t0) -> const u32* {  // where halo(t0 + lane&15) lives
                    const int t = t0 + (lane & 15);
                    return lefthalo ? &lds.halo[t & (SY_RH - 1)] : &lds.ring[ls - 1][(t + SY_W - 1) & (SY_R - 1)][63];
                };
                auto halo_need = [&](int t0) -> int {  // counter value that makes block t0's halo readable
                    return lefthalo ? min(t0 + SY_U - 1, rows) + 1 : min(t0 + SY_U - 1 + SY_W, T_total);
                };
                auto cons_rows_done = [&]() -> int {
                    const int a0 = lds_load(&lds.cons_blk[ls][0]), a1 = lds_load(&lds.cons_blk[ls][1]);
                    const int a2 = lds_load(&lds.cons_blk[ls][2]), a3 = lds_load(&lds.cons_blk[ls][3]);
                    return SY_U * (min(min(a0, a1), min(a2, a3)) + 1);
                };

                // this lane's row characters: at step t it needs b[t-1-lane]; bp points at that byte for
                // the current 64-step chunk.  16 dwords = 64 steps, loaded one chunk ahead.
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                typedef u32x4 __attribute__((aligned(1))) uint4_u;
                const unsigned char* bp = bpad + 64 - lane;  // + (t-1)
                u32x4 c0 = *(const uint4_u*)(bp + 0), c1 = *(const uint4_u*)(bp + 16), c2 = *(const uint4_u*)(bp + 32),
                      c3 = *(const uint4_u*)(bp + 48);

                Snap snap = {0, -1, -1, -1, -1, 0, 0u};
                bool snap_pending = false, have_next = false;
                Spin spin;

                // 16 steps t0..t0+15 (KB = first step's index inside the 64-step chunk)
                auto run_block = [&](auto PRO, auto KB, int t0, const u32x4& C) -> bool {
                    constexpr int kb = decltype(KB)::value;
                    if (snap_pending) snap = snap_collect();
                    u32 Hv;
                    if (have_next) {
                        Hv = snap.hv;
                    } else {
                        while (lds_load(left_cnt) < halo_need(t0))
                            if (spin.fail(p.abort_flag)) return false;
                        asm volatile("" ::: "memory");
                        Hv = __hip_atomic_load(halo_ptr(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    // ring slots of steps t0..t0+15 may be overwritten once their old contents (steps
                    // t0-R ..) were consumed: rows <= t0+14-R by my consumers, halo rows <= t0+15-R-63
                    // by the strip to the right / the exporter.  The snapshot is older, hence conservative.
                    if (SY_U * (min(min(snap.c0, snap.c1), min(snap.c2, snap.c3)) + 1) < t0 + SY_U - 2 - SY_R)
                        while (cons_rows_done() < t0 + SY_U - 2 - SY_R)
                            if (spin.fail(p.abort_flag)) return false;
                    if (snap.right < t0 + SY_U - 1 - SY_R - SY_W)
                        while (lds_load(right_cnt) < t0 + SY_U - 1 - SY_R - SY_W)
                            if (spin.fail(p.abort_flag)) return false;
                    const bool next_ok = (t0 + SY_U <= T_total) && snap_pending && (snap.left >= halo_need(t0 + SY_U));
                    const u32 waddr = ringbase + (u32)lane * 4u + (u32)((t0 - 1) & (SY_R - 1)) * 256u;
                    const u32 next_halo_addr = (u32)(size_t)halo_ptr(t0 + SY_U);
                    sfor<0, SY_U>([&](auto K) {
                        constexpr int k = kb + K.value;  // index in the chunk
                        const u32 Cw = (k / 4) % 4 == 0 ? C.x : (k / 4) % 4 == 1 ? C.y : (k / 4) % 4 == 2 ? C.z : C.w;
                        u32 g = producer_step<k>(G1, G2, m, d, Hv, a_l, Cw, Z, ngap_v, xm_v, mm_v, waddr);
                        // first 63 steps only: the lane whose row is 0 takes the halo-row value (kept out
                        // of the main loop: a select here would be a third dependent op per step)
                        if constexpr (decltype(PRO)::value) g = (lane == t0 + K.value) ? G0v : g;
                        G2 = G1;
                        G1 = g;
                        if constexpr (K.value == 7) {  // mid-block: ask for the next boundary's counters
                            snap_issue(left_addr, cons_addr, right_addr);
                            if (next_ok) halo_issue(next_halo_addr);
                        }
                    });
                    lds_store(&lds.prod_t[ls], t0 + SY_U - 1);  // in order behind the ring writes
                    snap_pending = true;
                    have_next = next_ok;
                    return true;
                };
                auto run_chunk = [&](auto PRO, int tc) -> bool {  // 64 steps from step tc
                    const unsigned char* nb_ = bp + (tc - 1) + 64;  // next chunk's characters
                    const u32x4 n0 = *(const uint4_u*)(nb_ + 0), n1 = *(const uint4_u*)(nb_ + 16), n2 = *(const uint4_u*)(nb_ + 32),
                                n3 = *(const uint4_u*)(nb_ + 48);
                    if (!run_block(PRO, std::integral_constant<int, 0>{}, tc, c0)) return false;
                    if (tc + 16 <= T_total && !run_block(PRO, std::integral_constant<int, 16>{}, tc + 16, c1)) return false;
                    if (tc + 32 <= T_total && !run_block(PRO, std::integral_constant<int, 32>{}, tc + 32, c2)) return false;
                    if (tc + 48 <= T_total && !run_block(PRO, std::integral_constant<int, 48>{}, tc + 48, c3)) return false;
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                    return true;
                };
                int tc = 1;
                if (!run_chunk(std::true_type{}, tc)) return;   // steps 1..64 contain every row-0 injection
                for (tc += 64; tc <= T_total; tc += 64)
                    if (!run_chunk(std::false_type{}, tc)) return;
            }
        } else if (wave < NS + NS * NC) {
            // ================================ consumer ================================
            const int ls = (wave - NS) % NS, ci = (wave - NS) / NS, s = s0 + ls;
            if (ls < nact && (p.debug_flags & 2)) {
                lds_store(&lds.cons_blk[ls][ci], 1 << 24);  // timing experiment: producer alone
            } else if (ls < nact) {
                const u32 j = (u32)s * SY_W + (u32)lane;
                const bool jvalid = (int64_t)j < M;
                const bool store_ok = jvalid && (lane >= 1 || s == 0) && !(p.debug_flags & 1);
                const u32 voffH = store_ok ? j * (u32)sizeof(HT) : SY_OOB;
                const u32 voffP = store_ok ? j * 4u : SY_OOB;
                const int a_l = (lane >= 1 && jvalid) ? (int)seq_a[j - 1] : SY_ASENT;
                const int cz = ngap * (int)j;
                const int G0v = jvalid ? (p.top ? p.top[j] : 0) + cz : 0;
                HT* H = (HT*)p.H;
                int32_t* P = p.P;
                if (ci == 0 && store_ok) {  // row 0: the halo row itself
                    H[j] = (HT)(p.top ? p.top[j] : 0);
                    P[j] = 0;
                }
                int bestv = store_ok ? 0 : 0x7fffffff, bestrow = 0;
                const int nblk = (rows + SY_U - 1) / SY_U;
                int snap_prod = 0;
                Spin spin;
                for (int q = ci; q < nblk; q += NC) {
                    const int r0 = q * SY_U + 1;
                    const int nb = min(SY_U, rows - r0 + 1);
                    const int need = r0 + nb - 1 + SY_W;  // the step that completes row r0+nb-1
                    if (snap_prod < need)
                        while ((snap_prod = lds_load(&lds.prod_t[ls])) < need)
                            if (spin.fail(p.abort_flag)) return;
                    // rows r0-1 .. r0+15 of the ring, read along the skew: row r, lane l sits in slot r+l-1
                    asm volatile("" ::: "memory");
                    u32 slot = (u32)(r0 - 2 + lane);
                    u32 gv[SY_U + 1];
#pragma unroll
                    for (int k = 0; k <= SY_U; ++k, ++slot)
                        gv[k] = __hip_atomic_load(&lds.ring[ls][slot & (SY_R - 1)][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    snap_prod = lds_load(&lds.prod_t[ls]);  // looked at again one block later
                    int U = (r0 == 1) ? G0v : (int)gv[0];
                    // this block's b characters
                    u32 bw[4];
                    if (nb == SY_U) {
                        const uint4 w = *reinterpret_cast<const uint4*>(seq_b + (r0 - 1));
                        bw[0] = w.x; bw[1] = w.y; bw[2] = w.z; bw[3] = w.w;
                    } else {
                        bw[0] = bw[1] = bw[2] = bw[3] = 0;
                        for (int r = 0; r < nb; ++r) bw[r >> 2] |= (u32)seq_b[r0 - 1 + r] << (8 * (r & 3));
                    }
                    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)(H + (int64_t)r0 * M), 0, 0x7FFFFF00, 0x00020000);
                    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(P + (int64_t)r0 * M), 0, 0x7FFFFF00, 0x00020000);
                    const u32 rowH = (u32)(M * (int64_t)sizeof(HT)), rowP = (u32)(M * 4);
                    auto do_row = [&](int k) {
                        const int g = (int)gv[k + 1];
                        const int b_i = (int)((bw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                        const int D = __builtin_amdgcn_update_dpp(0, U, 0x138, 0xF, 0xF, false);  // G[r-1][j-1]
                        const int dd = D + ((a_l == b_i) ? mm : xm);
                        const int z = cz + ngap * (r0 + k);
                        const int h = g - z;
                        const int pred = (g == z) ? 0 : (dd == g) ? 3 : (U == g) ? 1 : 2;  // serial_smithW.c:204-234
                        if constexpr (sizeof(HT) == 8) {
                            typedef int v2i __attribute__((ext_vector_type(2)));
                            v2i hv; hv.x = h; hv.y = h >> 31;
                            __builtin_amdgcn_raw_buffer_store_b64(hv, rH, voffH, (int)(rowH * (u32)k), 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(h, rH, voffH, (int)(rowH * (u32)k), 0);
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(pred, rP, voffP, (int)(rowP * (u32)k), 0);
                        if (h > bestv) { bestv = h; bestrow = r0 + k; }
                        U = g;
                    };
                    if (nb == SY_U) {  // one basic block: the 16 rows are independent, hipcc interleaves them
                        sfor<0, SY_U>([&](auto K) { do_row(K.value); });
                    } else {
                        sfor<0, SY_U>([&](auto K) { if (K.value < nb) do_row(K.value); });
                    }
                    lds_store(&lds.cons_blk[ls][ci], q);
                }
                if (store_ok && bestv > 0) {
                    const u64 idx = (u64)bestrow * (u64)M + (u64)j;
                    atomicMax(p.result_key, ((u64)(u32)bestv << 40) | (SW_KEY_IDX_MASK - idx));
                }
            }
        } else {
            // ================================= helper ==================================
            // import: edge column of strip s0-1 (HBM granules, or column 0 synthesised) -> lds.halo
            // export: lane-63 column of the group's last strip -> HBM granules for the next group
            const int slast = s0 + nact - 1;
            const bool do_export = (slast + 1 < p.nstrips);
            const int k16 = lane & 15;
            int imp = 1, exp = 1;
            Spin spin;
            while (imp <= rows || (do_export && exp <= rows)) {
                bool progressed = false;
                if (imp <= rows && imp + SY_U - 1 - SY_RH <= lds_load(&lds.prod_t[0])) {
                    const int n = min(SY_U, rows - imp + 1);
                    const int r = min(imp + k16, rows);
                    u32 val;
                    bool ok = true;
                    if (s0 == 0) {
                        val = (u32)(ngap * r);  // column 0: H == 0
                    } else {
                        const u64 gr = __hip_atomic_load((gu64*)(p.edge + (int64_t)(s0 - 1) * estride + r), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                        ok = (gr >> 32) == (tag_base | (u64)r);
                        val = (u32)gr;
                    }
                    if (__all(ok)) {
                        if (lane < n) __hip_atomic_store(&lds.halo[(imp + lane) & (SY_RH - 1)], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        asm volatile("" ::: "memory");  // LDS executes a wave's ops in order: data before counter
                        imp += n;
                        lds_store(&lds.halo_ready, imp);
                        progressed = true;
                    }
                }
                if (do_export && exp <= rows) {
                    const int n = min(SY_U, rows - exp + 1);
                    if (lds_load(&lds.prod_t[nact - 1]) >= exp + n - 1 + SY_W) {
                        const int r = min(exp + k16, rows);
                        asm volatile("" ::: "memory");
                        const u32 v = __hip_atomic_load(&lds.ring[nact - 1][(r + SY_W - 1) & (SY_R - 1)][63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (lane < n)
                            __hip_atomic_store((gu64*)(p.edge + (int64_t)slast * estride + r), ((tag_base | (u64)r) << 32) | (u64)v,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        exp += n;
                        lds_store(&lds.exp_done, exp - 1);
                        progressed = true;
                    }
                }
                if (!progressed && spin.fail(p.abort_flag)) return;
            }
        }
        __syncthreads();
    }
}

#define SW_INST(NS, NC)                                                                                                   \
    template __global__ void sw_systolic<int32_t, NS, NC>(const unsigned char*, const unsigned char*, const unsigned char*, FillParams); \
    template __global__ void sw_systolic<int64_t, NS, NC>(const unsigned char*, const unsigned char*, const unsigned char*, FillParams);
SW_INST(2, 2)
SW_INST(2, 3)
SW_INST(2, 4)
SW_INST(1, 2)
SW_INST(1, 4)

#undef SW_INST

// bpad[64 + i] = b[i], zero padded on both sides: producer lane l reads b[t-1-l] for steps t that
// reach 63 rows above and ~130 rows below the matrix (those cells are never stored).
__global__ void sw_pad_b(const unsigned char* __restrict__ b, int64_t rows, unsigned char* __restrict__ bpad, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bpad[i] = (i >= 64 && i - 64 < rows) ? b[i - 64] : (unsigned char)0;
}

}  // namespace swk
