// sw_traceback.hip -- backtrack(), serial_smithW.c:262-277, wave-cooperative: ONE WAVE per problem (gfx950).
//   do { pred = P[pos] == DIAGONAL ? pos-m-1 : P[pos] == UP ? pos-m : pos-1;  P[pos] *= -1;  pos = pred; } while (P[pos] != NONE)
// A single lane chasing P through HBM pays a full memory latency per step (441 ns measured in round 2).  Here the wave loads a
// 64-row x 64-column window of P whose bottom-right part holds the cursor -- 64 coalesced row loads in flight at once, row q in
// the literal register v[64+q], column = lane -- and walks it without touching memory: the cursor (ti, tj) lives in SGPRs, a step
// is one VGPR-indexed v_mov_b32 (row ti) + v_readlane_b32 (lane tj) + a handful of SALU ops, and the codes of up to 64 steps are collected in
// one VGPR (v_writelane_b32, lane = step).  A flush turns them into positions with one wave prefix sum and negates the cells /
// writes the path with ONE 64-lane store each.  The walk only moves up and left, so it leaves a window through its top or left
// edge, and the next window is anchored at the new cursor.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sw_kernels.h"

namespace swk {

typedef unsigned int u32;

#define TB_WINDOW_REGS                                                                                                       \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", \
    "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104",   \
    "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",   \
    "v123", "v124", "v125", "v126", "v127"

// rows q = 0..63 of the window into v[64+q]; bytes beyond the matrix read as 0 (buffer bounds check)
template <typename PT>
__device__ __forceinline__ void tb_load_window(const PT* base, u32 num_records, u32 row_pitch_bytes, int lane) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const int d0 = __builtin_amdgcn_readfirstlane((int)(u32)b), d1 = __builtin_amdgcn_readfirstlane((int)(u32)(b >> 32));
    const u32 voff = (u32)lane * (u32)sizeof(PT);
#define TB_LD4(LD, R0, R1, R2, R3)                                                                                          \
    LD " v" #R0 ", %[vo], s[76:79], s75 offen\n\ts_add_u32 s75, s75, %[pitch]\n\t"                                          \
    LD " v" #R1 ", %[vo], s[76:79], s75 offen\n\ts_add_u32 s75, s75, %[pitch]\n\t"                                          \
    LD " v" #R2 ", %[vo], s[76:79], s75 offen\n\ts_add_u32 s75, s75, %[pitch]\n\t"                                          \
    LD " v" #R3 ", %[vo], s[76:79], s75 offen\n\ts_add_u32 s75, s75, %[pitch]\n\t"
#define TB_LOAD(LD)                                                                                                          \
    asm volatile(                                                                                                            \
        "s_mov_b32 s76, %[d0]\n\ts_mov_b32 s77, %[d1]\n\ts_mov_b32 s78, %[nr]\n\ts_mov_b32 s79, 0x00020000\n\ts_mov_b32 s75, 0\n\t" \
        "s_nop 4\n\t"                                                                                                        \
        TB_LD4(LD, 64, 65, 66, 67) TB_LD4(LD, 68, 69, 70, 71) TB_LD4(LD, 72, 73, 74, 75) TB_LD4(LD, 76, 77, 78, 79)         \
        TB_LD4(LD, 80, 81, 82, 83) TB_LD4(LD, 84, 85, 86, 87) TB_LD4(LD, 88, 89, 90, 91) TB_LD4(LD, 92, 93, 94, 95)         \
        TB_LD4(LD, 96, 97, 98, 99) TB_LD4(LD, 100, 101, 102, 103) TB_LD4(LD, 104, 105, 106, 107) TB_LD4(LD, 108, 109, 110, 111) \
        TB_LD4(LD, 112, 113, 114, 115) TB_LD4(LD, 116, 117, 118, 119) TB_LD4(LD, 120, 121, 122, 123) TB_LD4(LD, 124, 125, 126, 127) \
        "s_waitcnt vmcnt(0)\n\t"                                                                                             \
        :: [vo] "v"(voff), [d0] "s"(d0), [d1] "s"(d1), [nr] "s"(num_records), [pitch] "s"(row_pitch_bytes)                  \
        : "memory", "scc", "s75", "s76", "s77", "s78", "s79", TB_WINDOW_REGS)
    if constexpr (sizeof(PT) == 4) { TB_LOAD("buffer_load_dword"); } else { TB_LOAD("buffer_load_sbyte"); }
#undef TB_LOAD
#undef TB_LD4
}

// The same window out of a 2-bit matrix (P2Cells: 4 codes per byte, cell k in bits 2 (k & 3) of byte k >> 2): every lane fetches the byte
// that holds its cell of row q (byte offset (cw + q M) >> 2 from the byte of the window's corner cell, cw = (corner & 3) + lane) and
// shifts its code out.  One asm statement, like the loader above: the compiler never sees the window registers.  nbytes: bytes from the
// corner's byte to the end of the matrix (beyond: 0, the buffer bounds check).
__device__ __forceinline__ void tb_load_window_p2(const unsigned char* corner_byte, u32 corner_phase, u32 M, u32 nbytes, int lane) {
    const uint64_t b = (uint64_t)(uintptr_t)corner_byte;
    const int d0 = __builtin_amdgcn_readfirstlane((int)(u32)b), d1 = __builtin_amdgcn_readfirstlane((int)(u32)(b >> 32));
    u32 cw = corner_phase + (u32)lane, cw2 = cw;
#define TB_P2_LD(Q) "v_lshrrev_b32 v62, 2, %[cw]\n\tbuffer_load_ubyte v" #Q ", v62, s[76:79], 0 offen\n\tv_add_u32 %[cw], %[cw], %[m]\n\t"
#define TB_P2_EX(Q) "v_and_b32 v62, 3, %[cw2]\n\tv_lshlrev_b32 v62, 1, v62\n\tv_lshrrev_b32 v" #Q ", v62, v" #Q "\n\tv_and_b32 v" #Q ", 3, v" #Q "\n\tv_add_u32 %[cw2], %[cw2], %[m]\n\t"
#define TB_P2_8(X, A, B, C, D, E, F, G, H) X(A) X(B) X(C) X(D) X(E) X(F) X(G) X(H)
#define TB_P2_ALL(X)                                                                                                          \
    TB_P2_8(X, 64, 65, 66, 67, 68, 69, 70, 71) TB_P2_8(X, 72, 73, 74, 75, 76, 77, 78, 79) TB_P2_8(X, 80, 81, 82, 83, 84, 85, 86, 87)           \
    TB_P2_8(X, 88, 89, 90, 91, 92, 93, 94, 95) TB_P2_8(X, 96, 97, 98, 99, 100, 101, 102, 103) TB_P2_8(X, 104, 105, 106, 107, 108, 109, 110, 111) \
    TB_P2_8(X, 112, 113, 114, 115, 116, 117, 118, 119) TB_P2_8(X, 120, 121, 122, 123, 124, 125, 126, 127)
    asm volatile(
        "s_mov_b32 s76, %[d0]\n\ts_mov_b32 s77, %[d1]\n\ts_mov_b32 s78, %[nr]\n\ts_mov_b32 s79, 0x00020000\n\t"
        "s_nop 4\n\t"
        TB_P2_ALL(TB_P2_LD)
        "s_waitcnt vmcnt(0)\n\t"
        TB_P2_ALL(TB_P2_EX)
        : [cw] "+v"(cw), [cw2] "+v"(cw2)
        : [d0] "s"(d0), [d1] "s"(d1), [nr] "s"(nbytes), [m] "s"(M)
        : "memory", "s76", "s77", "s78", "s79", "v62", TB_WINDOW_REGS);
#undef TB_P2_ALL
#undef TB_P2_8
#undef TB_P2_EX
#undef TB_P2_LD
}

// Up to 64 steps inside the window.  ti / tj: cursor (row register, lane); returns the number of steps taken, their codes in
// lane order (rec), and why it stopped: 0 = 64 steps done, 1 = the cursor left the window (ti or tj is -1), 2 = P <= 0 (end).
__device__ __forceinline__ int tb_walk(int& ti, int& tj, int& status, u32& rec) {
    int k, c, t, keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 %[k], 0\n\t"
        "s_mov_b32 %[st], 0\n\t"
        "v_mov_b32 v63, 0\n"
        "Ltb_loop_%=:\n\t"
        "s_set_gpr_idx_on %[ti], 0x1\n\t"          /* src0 of the next VALU op is v[64 + ti] */
        "v_mov_b32 v62, v64\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_mov_b32 m0, %[k]\n\t"                   /* (also the wait state between the VALU write of v62 and v_readlane) */
        "v_readlane_b32 %[c], v62, %[tj]\n\t"
        "s_cmp_lt_i32 %[c], 1\n\t"
        "s_cbranch_scc1 Ltb_end_%=\n\t"
        "v_writelane_b32 v63, %[c], m0\n\t"
        "s_add_u32 %[k], %[k], 1\n\t"
        "s_and_b32 %[t], %[c], 1\n\t"
        "s_sub_i32 %[ti], %[ti], %[t]\n\t"
        "s_lshr_b32 %[t], %[c], 1\n\t"
        "s_sub_i32 %[tj], %[tj], %[t]\n\t"
        "s_or_b32 %[t], %[ti], %[tj]\n\t"
        "s_cmp_lt_i32 %[t], 0\n\t"
        "s_cbranch_scc1 Ltb_exit_%=\n\t"
        "s_cmp_lt_u32 %[k], 64\n\t"
        "s_cbranch_scc1 Ltb_loop_%=\n\t"
        "s_branch Ltb_done_%=\n"
        "Ltb_end_%=:\n\t"
        "s_mov_b32 %[st], 2\n\t"
        "s_branch Ltb_done_%=\n"
        "Ltb_exit_%=:\n\t"
        "s_mov_b32 %[st], 1\n"
        "Ltb_done_%=:\n\t"
        "s_mov_b32 m0, %[keep]\n\t"
        "v_mov_b32 %[rec], v63\n\t"
        : [k] "=&s"(k), [c] "=&s"(c), [t] "=&s"(t), [keep] "=&s"(keep), [st] "=&s"(status), [ti] "+s"(ti), [tj] "+s"(tj), [rec] "=v"(rec)
        :
        : "memory", "scc", "v62", "v63", TB_WINDOW_REGS);
    return k;
}

// The same walk unrolled over the rows (scripts/gen_traceback_rows.py): it starts at row 63 of the window, so it serves every
// window that is not clamped at the top of the matrix.  tj: column (lane) in, column of the next cursor / the stop cell out.
// Returns `low`, the lowest row visited, and per row q >= low in lane q of `rec`: 4 * (LEFT steps in the row) + the code the row
// was left with (3 DIAGONAL, 1 UP; in row `low` also 2 = through the left edge of the window, 0 = the path ended).
// status: 1 -> next cursor (low - 1, tj), 3 -> next cursor (low, -1), 2 -> the path ended at (low, tj).
__device__ __forceinline__ int tb_walk_rows(int& tj, int& status, u32& rec) {
    int c, cnt, low;
    asm volatile(
        "v_mov_b32 v63, 3\n\t"
        "s_mov_b32 %[cnt], 0\n\t"
        "s_nop 0\n\t"
#include "sw_traceback_rows.inc"
        "v_mov_b32 %[rec], v63\n\t"
        : [c] "=&s"(c), [cnt] "=&s"(cnt), [low] "=&s"(low), [st] "=&s"(status), [tj] "+s"(tj), [rec] "=v"(rec)
        :
        : "memory", "scc", "v63", TB_WINDOW_REGS);
    return low;
}

__device__ __forceinline__ u32 tb_wave_max_u32(u32 v) {   // maximum over the 64 lanes, valid in lane 63
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true));
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 tb_wave_prefix_sum(u32 v) {   // inclusive, 64 lanes
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);   // row_bcast:15
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);   // row_bcast:31
    return v;
}

// One wave per problem (blockIdx.x = problem): walks problem k's P (at P + k * pstride, row stride M, rows1 rows) from start_pos
// (>= 0) or res[k].max_pos, negates the path, writes its length to res[k].path_len, optionally the visited indices to
// paths + k * cap and the index of the cell the walk stopped at (the first cell with P <= 0) to stop[k].
template <typename PT>
__global__ void __launch_bounds__(128) sw_traceback_wave(PT* __restrict__ P, int64_t M, int64_t rows1, int64_t pstride, int64_t start_pos,
                                                         int64_t* __restrict__ paths, int64_t cap, sw_result* __restrict__ res, int64_t* __restrict__ stop,
                                                         unsigned int* __restrict__ pathbits) {
    // PACKED: a 2-bit matrix (4 codes per byte).  It cannot hold the sign the walk leaves on its cells: the path is marked in the
    // bitmap `pathbits` instead (bit k & 31 of word k >> 5 for cell k, optional, zeroed by the caller).
    constexpr bool PACKED = std::is_same<PT, P2Cells>::value;
    // cursor of the walking wave (row, column, done) for the second wave of a two-wave launch, which reads the cells the walk is
    // heading for -- the band around the diagonal above the cursor -- ahead of it, so that the window loads hit the L2 instead
    // of paying an HBM miss per window (a freshly filled 1 GB matrix is nowhere near any cache).  Results do not depend on it.
    __shared__ volatile int64_t cur[3];
    const int lane = threadIdx.x & 63;
    const int64_t k = blockIdx.x;
    if (res[k].path_len < 0) return;            // the fill of this problem was aborted
    PT* Pk = PACKED ? P : P + k * pstride;
    const int64_t koff = PACKED ? k * pstride : 0;                  // (packed: cell offset of problem k inside the byte array)
    const unsigned char* const pbase = (const unsigned char*)P;
    auto mark = [&](int64_t idx, int neg_code) {
        if constexpr (PACKED) {
            if (pathbits) atomicOr(pathbits + ((koff + idx) >> 5), 1u << (u32)((koff + idx) & 31));
        } else {
            Pk[idx] = (PT)neg_code;
        }
    };
    const int64_t pos_l = start_pos >= 0 ? start_pos : res[k].max_pos;
    const int64_t pos = ((int64_t)__builtin_amdgcn_readfirstlane((int)(pos_l >> 32)) << 32) | (int64_t)(u32)__builtin_amdgcn_readfirstlane((int)(u32)pos_l);
    const int64_t total = rows1 * M;
    const bool warm = blockDim.x > 64;
    if (warm) {
        if (threadIdx.x == 0) { cur[0] = pos / M; cur[1] = pos - (pos / M) * M; cur[2] = 0; }
        __syncthreads();
        if (threadIdx.x >= 64) {
            constexpr int LINE = PACKED ? 512 : 128 / (int)sizeof(PT);            // matrix columns per 128-byte line
            int64_t next_row = cur[0];                               // rows above this one are still to be touched
            u32 sink = 0;
            for (;;) {
                const int64_t gi = cur[0], gj = cur[1];
                if (cur[2]) break;
                const int64_t upto = gi - 640 > 0 ? gi - 640 : 0;    // stay ~600 rows ahead of the walk
                if (next_row > gi) next_row = gi;
                int batches = 0;
                while (next_row > upto && batches < 4) {             // 64 rows per pass, lane l takes row next_row - l
                    const int64_t r = next_row - lane;
                    const int64_t pc = gj - (gi - r);                // where a pure diagonal from the cursor crosses row r
                    if (r >= 0) {
#pragma unroll
                        for (int q = -3; q < 3; ++q) {               // [pc - 3 LINE, pc + 3 LINE): 6 lines around it
                            int64_t c = pc + (int64_t)q * LINE;
                            c = c < 0 ? 0 : c;
                            const int64_t idx = r * M + c;
                            if (idx < total)   // L2-served, fills the L2
                                sink += (u32)__hip_atomic_load(PACKED ? pbase + ((koff + idx) >> 2) : (const unsigned char*)(Pk + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    next_row -= 64;
                    ++batches;
                }
                if (batches == 0) __builtin_amdgcn_s_sleep(32);
            }
            if (sink == 0x9e3779b9u && stop == (int64_t*)1) stop[0] = 0;   // (keeps the loads alive; never true)
            return;
        }
    }
    int64_t* path = paths ? paths + k * cap : nullptr;
    int64_t gi = pos / M, gj = pos - gi * M, len = 0;
#ifdef TB_PROFILE
    uint64_t t_load = 0, t_walk = 0, t_flush = 0, nwin = 0, tA, tB;
#define TB_T(x) x = __builtin_amdgcn_s_memrealtime()
#else
#define TB_T(x)
#endif
    for (;;) {
        // window: rows r0 .. r0+63, columns c0 .. c0+63, the cursor in its bottom-right part
        const int64_t r0 = gi > 63 ? gi - 63 : 0, c0 = gj > 63 ? gj - 63 : 0;
        int ti = __builtin_amdgcn_readfirstlane((int)(gi - r0)), tj = __builtin_amdgcn_readfirstlane((int)(gj - c0));
        const int64_t w0 = r0 * M + c0;
        const uint64_t remain = (uint64_t)(total - w0) * sizeof(PT);
        if (warm && lane == 0) { cur[0] = gi; cur[1] = gj; }
        TB_T(tA);
        if constexpr (PACKED)
        {   // (a window spans at most 64 M + 64 < 2^27 cells: 32-bit byte offsets; the matrix ends nbytes after the corner's byte)
            const int64_t c0k = koff + w0;
            const uint64_t nb = (uint64_t)((c0k & 3) + (total - w0) + 3) >> 2;
            tb_load_window_p2(pbase + (c0k >> 2), (u32)__builtin_amdgcn_readfirstlane((int)(c0k & 3)), (u32)__builtin_amdgcn_readfirstlane((int)(u32)M),
                              (u32)__builtin_amdgcn_readfirstlane((int)(nb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)nb)), lane);
        }
        else
            tb_load_window<PT>(Pk + w0, (u32)__builtin_amdgcn_readfirstlane((int)(remain > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)remain)),
                               (u32)__builtin_amdgcn_readfirstlane((int)(u32)(M * (int64_t)sizeof(PT))), lane);
        int status = 0;
#ifdef TB_PROFILE
        TB_T(tB); t_load += tB - tA; ++nwin;
#endif
        if (ti == 63) {
            // the row-unrolled walk: one record per ROW
            const int sj = tj;
            u32 rec;
            TB_T(tA);
            const int low = tb_walk_rows(tj, status, rec);
#ifdef TB_PROFILE
            TB_T(tB); t_walk += tB - tA;
#endif
            const bool vis = lane >= low;
            const u32 cnt = vis ? rec >> 2 : 0u, code = rec & 3u;
            const u32 ncell = vis ? cnt + ((code & 1u) ? 1u : 0u) : 0u;           // cells of my row on the path
            const u32 cmove = vis ? cnt + (code == 3u ? 1u : 0u) : 0u;            // columns the path moves left in my row
            const u32 d = (cmove << 16) | ncell;
            const u32 incl = tb_wave_prefix_sum(d);
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            const u32 above = tot - incl;                                          // rows walked before mine: lanes > mine
            const int64_t ecol = c0 + sj - (int64_t)(above >> 16);                 // column at which the walk enters my row
            const int64_t rbase = (r0 + lane) * M;
            const int64_t poff = len + (int64_t)(above & 0xffffu);
            const int maxc = __builtin_amdgcn_readfirstlane((int)tb_wave_max_u32(ncell));
            for (int i = 0; i < maxc; ++i) {
                if ((u32)i < ncell) {
                    const int64_t idx = rbase + ecol - i;
                    mark(idx, (u32)i < cnt ? -SW_LEFT : -(int)code);
                    if (path && poff + i < cap) path[poff + i] = idx;
                }
            }
            len += (int64_t)(tot & 0xffffu);
            ti = status == 1 ? low - 1 : low;
            if (status == 3) { tj = -1; status = 1; }
#ifdef TB_PROFILE
            TB_T(tA); t_flush += tA - tB;
#endif
        } else
        do {
            const int si = ti, sj = tj;            // cursor at the start of this run of steps
            u32 rec;
            TB_T(tA);
            const int n = tb_walk(ti, tj, status, rec);
#ifdef TB_PROFILE
            TB_T(tB); t_walk += tB - tA;
#endif
            if (n > 0) {
                const u32 c = lane < n ? rec : 0u;
                const u32 d = ((c & 1u) << 16) | (c >> 1);          // (rows up, columns left) of this step
                const u32 before = tb_wave_prefix_sum(d) - d;       // moves made by the earlier steps
                const int64_t idx = (r0 + si - (int64_t)(before >> 16)) * M + c0 + sj - (int64_t)(before & 0xffffu);
                if (lane < n) {
                    mark(idx, -(int)c);
                    if (path && len + lane < cap) path[len + lane] = idx;
                }
                len += n;
            }
#ifdef TB_PROFILE
            TB_T(tA); t_flush += tA - tB;
#endif
        } while (status == 0);
        gi = r0 + ti; gj = c0 + tj;
        if (status == 2 || gi < 0 || gj < 0) break;
    }
    if (warm && lane == 0) cur[2] = 1;
    if (lane == 0) {
        res[k].path_len = len;
        if (stop) stop[k] = gi * M + gj;
#ifdef TB_PROFILE
        if (stop) { stop[1] = (int64_t)t_load; stop[2] = (int64_t)t_walk; stop[3] = (int64_t)t_flush; stop[4] = (int64_t)nwin; }   // 100 MHz ticks
#endif
    }
}
template __global__ void sw_traceback_wave<int32_t>(int32_t*, int64_t, int64_t, int64_t, int64_t, int64_t*, int64_t, sw_result*, int64_t*, unsigned int*);
template __global__ void sw_traceback_wave<signed char>(signed char*, int64_t, int64_t, int64_t, int64_t, int64_t*, int64_t, sw_result*, int64_t*, unsigned int*);
template __global__ void sw_traceback_wave<P2Cells>(P2Cells*, int64_t, int64_t, int64_t, int64_t, int64_t*, int64_t, sw_result*, int64_t*, unsigned int*);

}  // namespace swk
