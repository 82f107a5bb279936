// sw_batch.hip -- BASELINE config 5: many independent pairs, ONE PAIR PER WAVE (gfx950).
//
// What is computed per pair is the reference recurrence, cell for cell (serial_smithW.c:187-256):
//   H[i][j] = max(0, H[i-1][j-1] + s(i,j), H[i-1][j] + gap, H[i][j-1] + gap);  P = first of DIAGONAL, UP, LEFT that attains a
//   positive maximum; maxPos = lowest linear index of max H.
//
// How.  A single pair is chain-bound (2N-1 dependent anti-diagonals); a batch is not, so nothing of the single-pair machinery
// (producer / consumer waves, LDS rings, strips chained through HBM) is used here.  One 64-lane wave owns one pair.  Lane l keeps
// the C adjacent columns  c0 = 64 C strip + C l + 1 .. c0 + C - 1  of the current row in registers and works, at step u, on row
// r = u - l: the wave sweeps an anti-diagonal of 64 row segments down the matrix, the only cross-lane traffic being ONE DPP move
// per step (lane l-1's last column = my left / diagonal neighbour).  Everything stays in H-space: the state of a lane is its row of
// H, the floor is the inline constant 0, rows above the matrix are inert because cells outside the sequences score -1 (0 + -1 < 0).
// Per cell: t = Hd + s (v_add_u32_sdwa, the score a byte of a v_perm_b32 profile look-up shared by 4 rows), u = max(Hu, Hl),
// u -= -gap, H = max3(t, u, 0): 4.25 VALU; the arg-max adds 0.6 (row maximum by max3 tree, one compare against the wave's best
// so far; the cell is looked for only in the rare steps that reach it), P 6 more (three compares, two selects, one SDWA select
// that deposits the code byte in place).  Matrix rows leave as 16 (int8 P) or 64 (int32) contiguous bytes per lane.
// Matrices wider than 64 C columns are swept strip after strip by the same wave; the boundary column goes through a scratch row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sw_kernels.h"

namespace swk {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr u32 SB_OOB = 0xFFFFFF00u;
#ifndef SB_RUN
#define SB_RUN 256                         // bytes of one row that a lane group stores with one instruction (int8 P)
#endif      // buffer offset beyond every descriptor: the store is dropped

// Letter codes of every pair's b (rank of the byte value among the values present in the batch, 0..7) in a padded copy:
// bcode[pair * per + front + i] = code(b[i]); 0x0D outside the sequence -- as a v_perm_b32 selector byte that yields 0xFF, the
// score -1 of a cell outside the matrix.  Block (0,0) publishes the code table (atab[0..255], letter count at atab[256]).
__global__ void __launch_bounds__(256) sw_batch_codes(const unsigned char* __restrict__ b, int64_t rows, int64_t b_pstride, unsigned char* __restrict__ bcode,
                                                      int64_t per, int front, const unsigned int* __restrict__ part, int npart, unsigned char* __restrict__ atab,
                                                      int64_t npairs) {
    __shared__ unsigned char tab[256];
    __shared__ unsigned int present[8];
    {
        if (threadIdx.x < 8) {   // the batch's presence map: OR of the partial maps of sw_prep_scan
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
        tab[t] = (here && nletters <= 8) ? (unsigned char)rank : (unsigned char)0x0D;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            atab[t] = tab[t];
            if (t == 0) *(unsigned int*)(atab + 256) = (unsigned int)nletters;
        }
    }
    __syncthreads();
    // (gridDim.y is capped at 65535 by the host -- the limit a HIP runtime may enforce: the pairs are taken grid-stride)
    for (int64_t pair = blockIdx.y; pair < npairs; pair += gridDim.y)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
            const bool in = (i >= front && i - front < rows);
            bcode[pair * per + i] = in ? tab[b[pair * b_pstride + i - front]] : (unsigned char)0x0D;
        }
}

__device__ __forceinline__ int sb_dpp_shr1(int old, int src) {   // lane l <- lane l-1; lane 0 keeps `old`
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int sb_sbyte(u32 w, int j) { return (int)(signed char)(w >> (8 * j)); }

__device__ __forceinline__ int sb_wave_max(int v) {   // max over the 64 lanes, wave-uniform result (v >= 0)
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true));   // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true));   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// P code of a cell, deposited as byte BYTE of the packed register `acc` in ONE select: acc.byte = (h == 0) ? 0 : p
template <int BYTE>
__device__ __forceinline__ void sb_deposit(u32& acc, int p, int h, int zero) {
    if constexpr (BYTE == 0)
        asm("v_cmp_eq_u32 vcc, 0, %2\n\tv_cndmask_b32_sdwa %0, %1, %3, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(acc) : "v"(p), "v"(h), "v"(zero) : "vcc");
    else if constexpr (BYTE == 1)
        asm("v_cmp_eq_u32 vcc, 0, %2\n\tv_cndmask_b32_sdwa %0, %1, %3, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(acc) : "v"(p), "v"(h), "v"(zero) : "vcc");
    else if constexpr (BYTE == 2)
        asm("v_cmp_eq_u32 vcc, 0, %2\n\tv_cndmask_b32_sdwa %0, %1, %3, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(acc) : "v"(p), "v"(h), "v"(zero) : "vcc");
    else
        asm("v_cmp_eq_u32 vcc, 0, %2\n\tv_cndmask_b32_sdwa %0, %1, %3, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
            : "+v"(acc) : "v"(p), "v"(h), "v"(zero) : "vcc");
}

template <int I, int N, typename F>
__device__ __forceinline__ void sb_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sb_for<I + 1, N>(f);
    }
}

typedef int sb_v4i __attribute__((ext_vector_type(4)));
typedef int sb_v2i __attribute__((ext_vector_type(2)));
template <int C> struct SbSeg { typedef int type; };            // C packed P codes of one lane and row
template <> struct SbSeg<8> { typedef sb_v2i type; };
template <> struct SbSeg<16> { typedef sb_v4i type; };

// C: columns per lane (4, 8, 16); PB: bytes per P element written (0: P not written, 1: int8, 4: int32)
template <int C, int PB>
__global__ void __launch_bounds__(256) sw_batch_wave(BatchParams p) {
    static_assert(C % 4 == 0 && C <= 16, "C is a multiple of 4 (packed P bytes, 16-byte stores)");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (pair >= p.npairs) return;
    if (*(const unsigned int*)(p.atab + 256) > 8u) return;   // (the host has already checked: kept as a guard)
    const int cols = (int)p.cols, rows = (int)p.rows, M = cols + 1;
    const int ngap = p.ngap;
    const unsigned char* __restrict__ a = p.a + pair * p.a_pstride;
    const unsigned char* __restrict__ bcl = p.bcode + pair * p.bcode_pstride + p.bfront - 1 - lane;   // bcl[u] = code of b[u - lane - 1]
    const bool wh = p.H != nullptr;
    constexpr bool wp = PB != 0;
    const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)(p.H ? p.H + pair * p.hp_pstride : nullptr), 0,
                                                                        wh ? (int)((int64_t)(rows + 1) * M * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.P ? (char*)p.P + pair * p.hp_pstride * (PB ? PB : 1) : nullptr), 0,
                                                                        wp ? (int)((int64_t)(rows + 1) * M * PB) : 0, 0x00020000);
    const int nstrips = (cols + 64 * C - 1) / (64 * C);
    const bool multi = nstrips > 1;
    // boundary column between strips: bnd[r + 64] = H[r][last column of the strip], rows -63 .. rows (+ slack)
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(multi ? p.bnd + pair * p.bnd_pstride : nullptr), 0,
                                                                        multi ? (int)(p.bnd_pstride * 4) : 0, 0x00020000);
    const bool ragged = (cols % C) != 0;
    const u32 mis4 = 0x01010101u * (u32)(unsigned char)(signed char)p.mismatch;
    const u32 dmm = ((u32)(unsigned char)(signed char)p.match) ^ ((u32)(unsigned char)(signed char)p.mismatch);
    u64 kbest = 0;         // per lane: best (score << 40 | MASK - index) over the strips done
    int sbest = 1;         // wave-uniform: highest valid H seen so far (at least 1: zeros never count)
    // steps 0 .. rows + 63 (lane 63's last row); delayed int8 P stores drain for up to 128 / C + 1 more
    const int G = (rows + 64 + (PB == 1 ? SB_RUN / C + 1 : 0) + 3) / 4;
    __shared__ __attribute__((aligned(16))) unsigned char sb_ring[PB == 1 ? 4 * 64 * SB_RUN : 16];   // (4 waves per workgroup; 2-wave workgroups were slower: 1963 vs 2148 GCUPS)
    unsigned char* const ring = sb_ring + (PB == 1 ? wave * 64 * SB_RUN + lane * C : 0);

    for (int st = 0; st < nstrips; ++st) {
        const int c0 = st * 64 * C + lane * C + 1;
        const int nval = min(C, max(0, cols - c0 + 1));            // columns of this lane inside the matrix
        // score profiles of my columns: byte L = score against letter code L (lo: codes 0..3, hi: 4..7)
        u32 lo[C], hi[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int c = c0 + k;
            if (c <= cols) {
                const u32 code = p.atab[a[c - 1]];
                lo[k] = mis4 ^ (code < 4u ? dmm << (8 * code) : 0u);
                hi[k] = mis4 ^ (code >= 4u && code < 8u ? dmm << (8 * (code - 4u)) : 0u);
            } else {
                lo[k] = hi[k] = 0xFFFFFFFFu;                         // outside the matrix: -1 against everything
            }
        }
        int h[C];
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = 0;
        int diag0 = 0, lbest = 0, lk = 0, lstep = 0;
        // output offsets of my row segment at step u = 0 (row -lane): wraps to a huge unsigned offset above the matrix, runs past
        // the descriptor below it -- stores outside rows 0..rows are dropped by the bounds check.  (Row 0 is stored too: H = P = 0.)
        const bool full = nval == C && !(p.debug & 1);   // (debug bit 0: drop the matrix stores, timing experiments)
        u32 voffH = (full && wh) ? (u32)((-lane * M + c0) * 4) : SB_OOB;
        u32 voffP = (full && wp) ? (u32)((-lane * M + c0) * PB) : SB_OOB;
        // int8 P: delayed stores through the LDS ring (see the store below): this lane's piece leaves 1 + dly steps late
        constexpr int GS = PB == 1 ? SB_RUN / C : 1;
        const int dly = GS - 1 - (lane & (GS - 1));
        const int dslot = (GS - dly) & (GS - 1);                     // the slot read at step u is (u - dly) mod GS
        if (PB == 1 && full) voffP -= (u32)((1 + dly) * M);
        typedef typename SbSeg<C>::type SegT;
        SegT dseg = {};
        const bool col0 = st == 0 && lane == 0;
        u32 voffH0 = (wh && col0) ? 0u : SB_OOB, voffP0 = (wp && col0) ? 0u : SB_OOB;   // column 0
        // (lanes that do not store keep their out-of-range offset: pitch 0)
        const u32 pitchH = full ? (u32)(M * 4) : 0u, pitchP = full ? (u32)(M * PB) : 0u;
        const u32 pitchH0 = col0 ? (u32)(M * 4) : 0u, pitchP0 = col0 ? (u32)(M * PB) : 0u;
        // boundary column: lane 63 writes its last column (row u - 63) for the next strip, lane 0 reads row u of the previous one
        const bool bw = multi && st + 1 < nstrips, br = multi && st > 0;
        // (sc1 loads: served from L2, which this wave's own earlier stores have reached once vmcnt has drained)
        sb_v4i bq = {0, 0, 0, 0};
        const u32 voffB = lane == 0 ? 64u * 4u : SB_OOB;
        if (br) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bq = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)voffB, 0, 16);
        }
        u32 selw_next;
        __builtin_memcpy(&selw_next, bcl, 4);

        for (int g = 0; g < G; ++g) {
            const u32 selw = selw_next;
            __builtin_memcpy(&selw_next, bcl + 4 * (g + 1), 4);     // codes of the next 4 rows (the padded copy covers the overrun)
            u32 S[C];
#pragma unroll
            for (int k = 0; k < C; ++k) S[k] = __builtin_amdgcn_perm(hi[k], lo[k], selw);
            const sb_v4i bcur = bq;
            if (br) bq = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)voffB, 16 * (g + 1), 16);

            sb_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int u = 4 * g + j;
                const int left = sb_dpp_shr1(br ? bcur[j] : 0, h[C - 1]);
                int dprev = diag0, prev = left;
                diag0 = left;
                u32 pk[C / 4];
                int p32[C];
#pragma unroll
                for (int q = 0; q < C / 4; ++q) pk[q] = 0;
                sb_for<0, C>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    const int old = h[k];
                    const int t = dprev + sb_sbyte(S[k], j);
                    const int u2 = max(old, prev);
                    const int hn = max(max(t, u2 - ngap), 0);
                    if constexpr (PB != 0) {
                        int pc = (old == u2) ? SW_UP : SW_LEFT;
                        pc = (t == hn) ? SW_DIAGONAL : pc;
                        if constexpr (PB == 1) sb_deposit<k & 3>(pk[k >> 2], pc, hn, 0);
                        else p32[k] = (hn == 0) ? SW_NONE : pc;
                    }
                    h[k] = hn;
                    dprev = old;
                    prev = hn;
                });
                // ---- stores: my row segment of H and P (dropped outside rows 0..rows), column 0 by lane 0
                if (wh) {
#pragma unroll
                    for (int q = 0; q < C / 4; ++q) {
                        const sb_v4i v = {h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rH, (int)voffH, 16 * q, 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(0, rH, (int)voffH0, 0, 0);
                    if (ragged) {
#pragma unroll
                        for (int k = 0; k < C - 1; ++k)
                            if (!full && k < nval) __builtin_amdgcn_raw_buffer_store_b32(h[k], rH, (int)((u32)((u - lane) * M + c0 + k) * 4u), 0, 0);
                    }
                    voffH += pitchH; voffH0 += pitchH0;
                }
                if constexpr (PB == 4) {
#pragma unroll
                    for (int q = 0; q < C / 4; ++q) {
                        const sb_v4i v = {p32[4 * q], p32[4 * q + 1], p32[4 * q + 2], p32[4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rP, (int)voffP, 16 * q, 0);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(0, rP, (int)voffP0, 0, 0);
                    if (ragged) {
#pragma unroll
                        for (int k = 0; k < C - 1; ++k)
                            if (!full && k < nval) __builtin_amdgcn_raw_buffer_store_b32(p32[k], rP, (int)((u32)((u - lane) * M + c0 + k) * 4u), 0, 0);
                    }
                    voffP += pitchP; voffP0 += pitchP0;
                } else if constexpr (PB == 1) {
                    // Whole 128-byte runs instead of 64 scattered C-byte pieces per store.  Lane l's piece of row r is ready at
                    // step r + l, its neighbours' one step apart each: stored at once, a wave's store instruction touches 64
                    // different rows, and the L2 writes every piece to HBM on its own (measured: 3.3x the bytes, the fill at
                    // half speed).  So every lane parks its piece in a ring in LDS and stores the one it produced d = GS-1 - l%GS
                    // steps ago (+1 step: the read-back is used a step later, its latency behind the next row's arithmetic):
                    // the GS = SB_RUN/C lanes of a group then store pieces of the SAME row with one instruction -- SB_RUN contiguous bytes.
                    SegT seg;
                    if constexpr (C == 16) seg = SegT{(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
                    else if constexpr (C == 8) seg = SegT{(int)pk[0], (int)pk[1]};
                    else seg = (int)pk[0];
                    if constexpr (C == 16) __builtin_amdgcn_raw_buffer_store_b128(dseg, rP, (int)voffP, 0, 0);
                    else if constexpr (C == 8) __builtin_amdgcn_raw_buffer_store_b64(dseg, rP, (int)voffP, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b32(dseg, rP, (int)voffP, 0, 0);
                    *(SegT*)(ring + (u & (GS - 1)) * (64 * C)) = seg;
                    dseg = *(const SegT*)(ring + ((u + dslot) & (GS - 1)) * (64 * C));
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, rP, (int)voffP0, 0, 0);
                    if (ragged) {
#pragma unroll
                        for (int k = 0; k < C - 1; ++k)
                            if (!full && k < nval)
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(pk[k >> 2] >> (8 * (k & 3))), rP, (int)((u32)((u - lane) * M + c0 + k)), 0, 0);
                    }
                    voffP += pitchP; voffP0 += pitchP0;
                }
                if (bw) __builtin_amdgcn_raw_buffer_store_b32(h[C - 1], rB, lane == 63 ? 4 : (int)SB_OOB, 4 * u, 0);   // row u - 63 at index row + 64
                // ---- arg-max: the row maximum against the wave's best so far; only a step that reaches it looks for the cell.
                // (Cells outside the matrix need no masking here: such a cell never exceeds a cell of the matrix computed before
                // it -- it scores -1 against everything -- so it cannot raise `sbest` and a record it leaves in a lane loses
                // against a real one at the end.)
                int m = h[0];
#pragma unroll
                for (int k = 1; k + 1 < C; k += 2) m = max(max(m, h[k]), h[k + 1]);
                m = max(m, h[C - 1]);
                if (__builtin_amdgcn_ballot_w64(m >= sbest) != 0) {
                    sbest = max(sbest, sb_wave_max(m));
                    int kk = 0;                                   // first column of my row that holds its maximum
#pragma unroll
                    for (int k = C - 1; k >= 0; --k) kk = (h[k] == m) ? k : kk;
                    const bool imp = m > lbest;                   // strictly: an earlier row of this lane wins a tie
                    lk = imp ? kk : lk;
                    lstep = imp ? u : lstep;
                    lbest = max(lbest, m);
                }
            });
        }
        {
            const int r = lstep - lane, c = c0 + lk;
            if (lbest > 0 && r >= 1 && r <= rows && c <= cols) {
                const u64 key = ((u64)(u32)lbest << 40) | (SW_KEY_IDX_MASK - (u64)(u32)(r * M + c));
                kbest = key > kbest ? key : kbest;
            }
        }
    }
    // the pair's arg-max: highest score, lowest linear index among equals (serial_smithW.c:240-242)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u32 olo = (u32)__shfl_xor((int)(u32)kbest, off), ohi = (u32)__shfl_xor((int)(u32)(kbest >> 32), off);
        const u64 o = ((u64)ohi << 32) | olo;
        kbest = o > kbest ? o : kbest;
    }
    if (lane == 0) {
        sw_result* res = p.results + pair;
        res->max_score = (int64_t)(kbest >> 40);
        res->max_pos = kbest ? (int64_t)(SW_KEY_IDX_MASK - (kbest & SW_KEY_IDX_MASK)) : 0;
        res->path_len = 0;
    }
}

// ---- TWO pairs per wave on packed 16-bit lanes (round 4) --------------------------------------------------------------------------------
// Where every score of a pair fits 15 bits (match * min(cols, rows) < 2^15: 1024^2 reads score at most 3072) the recurrence runs on
// VOP3P instructions: the low half of every register belongs to pair 2w, the high half to pair 2w + 1 of the batch (same shape, same
// step numbering: lane l works on row u - l of BOTH pairs).  Per pair of cells
//     u2 = v_pk_max_u16(H_up, H_left)          t.lo / t.hi = v_add_u32_sdwa(H_diag.word, sext(S.byte))   (one per half: VOP3P has no SDWA)
//     ug = v_pk_sub_u16(u2, -gap) clamp        H = v_pk_max_i16(t, ug)
// -- the unsigned saturating subtract makes ug >= 0, so the maximum with the (possibly negative) t IS the floor at 0: 5 VALU for two cells
// where the 32-bit kernel needs 8.  No P, no H: score + exact maxPos only (the other output modes stay on sw_batch_wave).
// Arg-max, exact and without a branch: the row maximum of a lane's 16 cells is a binary tree of v_pk_max_u16 (15), a descent through the
// tree in packed arithmetic (b = min(m - left child, 1) is 0 where the left child holds the maximum; children are picked with
// v_pk_mad_u16) yields the FIRST column that holds it (30), and the lane's record (best, step, column) is updated with a strict "better
// than before" (7) -- all for both pairs at once, every step.  A lane's record is exact for the lane; the lanes are merged at the end by
// the reference's rule (highest score, lowest linear index).  Cells outside the matrix need no masking: they derive from cells of the
// matrix by strictly negative moves, so they stay below the pair's maximum and can only spoil the record of a lane that does not hold it.
#define SB_PK2(OP, D, A, B) asm(OP " %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) { u32 d; SB_PK2("v_pk_max_u16", d, a, b); return d; }
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) { u32 d; SB_PK2("v_pk_min_u16", d, a, b); return d; }
__device__ __forceinline__ u32 pk_max_i16(u32 a, u32 b) { u32 d; SB_PK2("v_pk_max_i16", d, a, b); return d; }
__device__ __forceinline__ u32 pk_sub_u16(u32 a, u32 b) { u32 d; SB_PK2("v_pk_sub_u16", d, a, b); return d; }             // wraps
__device__ __forceinline__ u32 pk_subs_u16(u32 a, u32 b) { u32 d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }   // saturates at 0
__device__ __forceinline__ u32 pk_mad_u16(u32 a, u32 b, u32 c) { u32 d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ u32 pk_mad_u16_sc(u32 a, u32 b, u32 c) { u32 d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c)); return d; }   // c: a wave-uniform constant (SGPR)
// t = {dprev.lo + sext(SA.byte J), dprev.hi + sext(SB.byte J)}
template <int J>
__device__ __forceinline__ u32 pk_add_sbytes(u32 dprev, u32 SA, u32 SB) {
    u32 t;
    if constexpr (J == 0) {
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "=v"(t) : "v"(dprev), "v"(SA));
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_0" : "+v"(t) : "v"(dprev), "v"(SB));
    } else if constexpr (J == 1) {
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_1" : "=v"(t) : "v"(dprev), "v"(SA));
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1" : "+v"(t) : "v"(dprev), "v"(SB));
    } else if constexpr (J == 2) {
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2" : "=v"(t) : "v"(dprev), "v"(SA));
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2" : "+v"(t) : "v"(dprev), "v"(SB));
    } else {
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_3" : "=v"(t) : "v"(dprev), "v"(SA));
        asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3" : "+v"(t) : "v"(dprev), "v"(SB));
    }
    return t;
}

// one cell of both pairs as ONE statement (the compiler pads every dependent pair of asm statements with an s_nop it cannot prove
// unnecessary: 3 per cell when the five instructions are five statements)
#define SB_PK_CELL(BYTE)                                                                                                              \
    asm("v_add_u32_sdwa %0, %3, sext(%4) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:" BYTE "\n\t"                    \
        "v_pk_max_u16 %1, %6, %7\n\t"                                                                                                 \
        "v_add_u32_sdwa %0, %3, sext(%5) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:" BYTE "\n\t"               \
        "v_pk_sub_u16 %1, %1, %8 clamp\n\t"                                                                                           \
        "v_pk_max_i16 %2, %0, %1"                                                                                                     \
        : "=&v"(t), "=&v"(u2), "=v"(hn) : "v"(dprev), "v"(SA), "v"(SB), "v"(old), "v"(prev), "v"(gg))
__device__ __forceinline__ u32 pk_mad_u16_sb(u32 a, u32 b, u32 c) { u32 d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c)); return d; }   // b: a wave-uniform constant (SGPR)
// the cell AND its P code (serial_smithW.c:187-256: DIAGONAL before UP before LEFT, NONE where H = 0) for both pairs, without a compare:
//     e = min(u2 - old, 1)   0: UP attains max(up, left)          d = min(H - t, 1)   0: the diagonal term attains H        z = min(H, 1)
//     code = z * (3 + d * (e - 2))     ->  3 (d = 0), 1 (d = 1, e = 0), 2 (d = 1, e = 1), 0 (H = 0)
// 13 instructions for two cells, one statement
#define SB_PK_CELL_P(BYTE)                                                                                                            \
    asm("v_add_u32_sdwa %0, %4, sext(%5) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:" BYTE "\n\t"                    \
        "v_pk_max_u16 %1, %7, %8\n\t"                                                                                                 \
        "v_add_u32_sdwa %0, %4, sext(%6) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:" BYTE "\n\t"               \
        "v_pk_sub_u16 %3, %1, %7\n\t"                                                                                                 \
        "v_pk_sub_u16 %1, %1, %9 clamp\n\t"                                                                                           \
        "v_pk_min_u16 %3, %3, %10\n\t"                                                                                                \
        "v_pk_max_i16 %2, %0, %1\n\t"                                                                                                 \
        "v_pk_sub_u16 %3, %3, %11\n\t"                                                                                                \
        "v_pk_sub_u16 %0, %2, %0\n\t"                                                                                                 \
        "v_pk_min_u16 %1, %2, %10\n\t"                                                                                                \
        "v_pk_min_u16 %0, %0, %10\n\t"                                                                                                \
        "v_pk_mad_u16 %0, %0, %3, %12\n\t"                                                                                            \
        "v_pk_mul_lo_u16 %3, %0, %1"                                                                                                  \
        : "=&v"(t), "=&v"(u2), "=&v"(hn), "=&v"(code) : "v"(dprev), "v"(SA), "v"(SB), "v"(old), "v"(prev), "v"(gg), "v"(one2), "v"(two2), "s"(0x00030003u))
template <int J>
__device__ __forceinline__ u32 pk_cell_p(u32 dprev, u32 SA, u32 SB, u32 old, u32 prev, u32 gg, u32 one2, u32 two2, u32& code) {
    u32 t, u2, hn;
    if constexpr (J == 0) SB_PK_CELL_P("BYTE_0");
    else if constexpr (J == 1) SB_PK_CELL_P("BYTE_1");
    else if constexpr (J == 2) SB_PK_CELL_P("BYTE_2");
    else SB_PK_CELL_P("BYTE_3");
    return hn;
}
template <int J>
__device__ __forceinline__ u32 pk_cell(u32 dprev, u32 SA, u32 SB, u32 old, u32 prev, u32 gg) {
    u32 t, u2, hn;
    if constexpr (J == 0) SB_PK_CELL("BYTE_0");
    else if constexpr (J == 1) SB_PK_CELL("BYTE_1");
    else if constexpr (J == 2) SB_PK_CELL("BYTE_2");
    else SB_PK_CELL("BYTE_3");
    return hn;
}

// C = 16 columns per lane; LE4: the batch has at most 4 distinct letters (codes 0..3: the upper half of the score profiles is never selected)
// K12: every score is below 4096 (match * min(cols, rows) < 2^12: 1024^2 with match 3), so score * 16 + (15 - column) fits a half: the tree of
// row maxima runs on these KEYS and its root already names the first column that holds the maximum -- 16 v_pk_mad_u16 instead of the
// 33-instruction descent (132 against 149.5 VALU per step)
// PB1: the predecessor matrices of both pairs are written too, one byte per cell.  The codes come out of packed arithmetic (SB_PK_CELL_P).  As in
// sw_batch_wave a lane's piece of a row waits in an LDS ring until the 16 lanes of its group can store 256 contiguous bytes of ONE row with
// one instruction -- but the ring holds the codes at TWO BITS each (4 + 4 bytes per lane and row for the two pairs: 8 KB per wave, so that
// two waves per SIMD fit; with byte rings the kernel ran one wave per SIMD and lost to its own latencies: 2400 GCUPS against 3700 with the
// stores dropped; with 128-byte runs and two waves per SIMD 1940).  Byte j of a packed word holds the codes of columns j, j + 4, j + 8, j + 12,
// so (w >> 2k) & 0x03030303 IS output dword k.
template <bool LE4, bool K12, bool PB1>
__global__ void __launch_bounds__(256, PB1 ? 2 : (LE4 ? 4 : 3)) sw_batch_wave16(BatchParams p) {   // (LE4 score-only: 4 waves per SIMD, at most 128 VGPRs)
    constexpr int C = 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t couple = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const int64_t pairA = 2 * couple;
    if (pairA >= p.npairs) return;
    const int64_t pairB = pairA + 1 < p.npairs ? pairA + 1 : pairA;   // (an odd batch: the last wave runs its pair in both halves)
    const int cols = (int)p.cols, rows = (int)p.rows, M = cols + 1;
    const unsigned char* __restrict__ aA = p.a + pairA * p.a_pstride;
    const unsigned char* __restrict__ aB = p.a + pairB * p.a_pstride;
    const unsigned char* __restrict__ bclA = p.bcode + pairA * p.bcode_pstride + p.bfront - 1 - lane;   // bcl[u] = code of b[u - lane - 1]
    const unsigned char* __restrict__ bclB = p.bcode + pairB * p.bcode_pstride + p.bfront - 1 - lane;
    const int nstrips = (cols + 64 * C - 1) / (64 * C);
    const bool multi = nstrips > 1;
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(multi ? p.bnd + pairA * p.bnd_pstride : nullptr), 0,
                                                                        multi ? (int)(p.bnd_pstride * 4) : 0, 0x00020000);   // (the couple shares pair A's boundary row: packed values)
    const u32 mis4 = 0x01010101u * (u32)(unsigned char)(signed char)p.mismatch;
    const u32 dmm = ((u32)(unsigned char)(signed char)p.match) ^ ((u32)(unsigned char)(signed char)p.mismatch);
    const u32 gg = (u32)p.ngap * 0x00010001u, one2 = 0x00010001u, sixteen2 = 0x00100010u, two2 = 0x00020002u;
    u64 kbestA = 0, kbestB = 0;
    const int G = (rows + 64 + (PB1 ? 66 : 0) + 3) / 4;   // (delayed P stores drain for up to 65 more steps)
    const int pbytes = PB1 ? (int)((int64_t)(rows + 1) * M) : 0;
    const __amdgpu_buffer_rsrc_t rPA = __builtin_amdgcn_make_buffer_rsrc((void*)(PB1 ? (char*)p.P + pairA * p.hp_pstride : nullptr), 0, pbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rPB = __builtin_amdgcn_make_buffer_rsrc((void*)(PB1 ? (char*)p.P + pairB * p.hp_pstride : nullptr), 0, pairB != pairA ? pbytes : 0, 0x00020000);
    const bool ragged = (cols % C) != 0;
    // per wave: a FIFO per lane, 63 - lane entries of (4 + 4) bytes deep -- lane l's piece of a row waits 63 - l steps, so that ALL 64 lanes store
    // their pieces of ONE row with one instruction: 1024 contiguous bytes (2016 entries = 16 128 bytes per wave)
    __shared__ __attribute__((aligned(16))) unsigned char sb_ring16[PB1 ? 4 * 2016 * 8 : 16];
    const int dly = 63 - lane, flen = 8 * dly;
    unsigned char* const fifo = sb_ring16 + (PB1 ? wave * 2016 * 8 + 8 * (63 * lane - lane * (lane - 1) / 2) : 0);

    for (int st = 0; st < nstrips; ++st) {
        const int c0 = st * 64 * C + lane * C + 1;
        u32 loA[C], loB[C], hiA[LE4 ? 1 : C], hiB[LE4 ? 1 : C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int c = c0 + k;
            u32 la = 0xFFFFFFFFu, ha = 0xFFFFFFFFu, lb = 0xFFFFFFFFu, hb = 0xFFFFFFFFu;   // outside the matrix: -1 against everything
            if (c <= cols) {
                const u32 ca = p.atab[aA[c - 1]], cb = p.atab[aB[c - 1]];
                la = mis4 ^ (ca < 4u ? dmm << (8 * ca) : 0u);
                ha = mis4 ^ (ca >= 4u && ca < 8u ? dmm << (8 * (ca - 4u)) : 0u);
                lb = mis4 ^ (cb < 4u ? dmm << (8 * cb) : 0u);
                hb = mis4 ^ (cb >= 4u && cb < 8u ? dmm << (8 * (cb - 4u)) : 0u);
            }
            loA[k] = la; loB[k] = lb;
            if constexpr (!LE4) { hiA[k] = ha; hiB[k] = hb; }
        }
        u32 h[C];
#pragma unroll
        for (int k = 0; k < C; ++k) h[k] = 0u;
        u32 diag0 = 0u, lbest = 0u, lk = 0u, lstep = 0u;
        const bool bw = multi && st + 1 < nstrips, br = multi && st > 0;
        // P stores (PB1): as in sw_batch_wave -- my piece of row u - lane leaves 1 + dly steps late, so that the 16 lanes of a group store
        // 256 contiguous bytes of ONE row per instruction
        const int nval = min(C, max(0, cols - c0 + 1));
        const bool full = PB1 && nval == C && !(p.debug & 1);
        int foff = 0;
        u32 voffP = full ? (u32)(-lane * M + c0) - (u32)((1 + dly) * M) : SB_OOB;
        const bool col0 = PB1 && st == 0 && lane == 0;
        u32 voffP0 = col0 ? 0u : SB_OOB;
        const u32 pitchP = full ? (u32)M : 0u, pitchP0 = col0 ? (u32)M : 0u;
        sb_v2i dw = {0, 0};        // the delayed row's codes, two bits each: .x pair A, .y pair B
        sb_v4i bq = {0, 0, 0, 0};
        const u32 voffB = lane == 0 ? 64u * 4u : SB_OOB;
        if (br) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bq = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)voffB, 0, 16);
        }
        u32 selA_next, selB_next;
        __builtin_memcpy(&selA_next, bclA, 4);
        __builtin_memcpy(&selB_next, bclB, 4);

        for (int g = 0; g < G; ++g) {
            const u32 selA = selA_next, selB = selB_next;
            __builtin_memcpy(&selA_next, bclA + 4 * (g + 1), 4);     // codes of the next 4 rows (the padded copy covers the overrun)
            __builtin_memcpy(&selB_next, bclB + 4 * (g + 1), 4);
            u32 SA[C], SB[C];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                SA[k] = __builtin_amdgcn_perm(LE4 ? loA[k] : hiA[k], loA[k], selA);
                SB[k] = __builtin_amdgcn_perm(LE4 ? loB[k] : hiB[k], loB[k], selB);
            }
            const sb_v4i bcur = bq;
            if (br) bq = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)voffB, 16 * (g + 1), 16);

            sb_for<0, 4>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int u = 4 * g + j;
                const u32 left = (u32)sb_dpp_shr1(br ? bcur[j] : 0, (int)h[C - 1]);
                u32 dprev = diag0, prev = left;
                diag0 = left;
                u32 pc[PB1 ? C : 1];
                sb_for<0, C>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    const u32 old = h[k];
                    u32 hn;
                    if constexpr (PB1) hn = pk_cell_p<j>(dprev, SA[k], SB[k], old, prev, gg, one2, two2, pc[k]);
                    else hn = pk_cell<j>(dprev, SA[k], SB[k], old, prev, gg);
                    h[k] = hn;
                    dprev = old;
                    prev = hn;
                });
                if constexpr (PB1) {
                    // byte j of a pair's word: the codes of columns j, j + 4, j + 8, j + 12 at two bits each (Horner in packed arithmetic,
                    // halves = pairs); two v_perm_b32 separate the pairs
                    u32 nib[4];
                    sb_for<0, 4>([&](auto Jc) {
                        constexpr int jc = decltype(Jc)::value;
                        nib[jc] = pk_mad_u16_sb(pk_mad_u16_sb(pk_mad_u16_sb(pc[jc + 12], 0x00040004u, pc[jc + 8]), 0x00040004u, pc[jc + 4]), 0x00040004u, pc[jc]);
                    });
                    const u32 X = pk_mad_u16_sb(nib[1], 0x01000100u, nib[0]), Y = pk_mad_u16_sb(nib[3], 0x01000100u, nib[2]);
                    const sb_v2i w = {(int)__builtin_amdgcn_perm(Y, X, 0x05040100u), (int)__builtin_amdgcn_perm(Y, X, 0x07060302u)};
                    // the row that left the ring a step ago, expanded to bytes: output dword k = (w >> 2k) & 0x03030303
                    const u32 m3 = 0x03030303u, dA = (u32)dw.x, dB = (u32)dw.y;
                    const sb_v4i dsegA = {(int)(dA & m3), (int)((dA >> 2) & m3), (int)((dA >> 4) & m3), (int)((dA >> 6) & m3)};
                    const sb_v4i dsegB = {(int)(dB & m3), (int)((dB >> 2) & m3), (int)((dB >> 4) & m3), (int)((dB >> 6) & m3)};
                    // (streaming: 3150-3190 GCUPS against 2890 write-back -- a row is written once and read much later, by the traceback)
                    __builtin_amdgcn_raw_buffer_store_b128(dsegA, rPA, (int)voffP, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(dsegB, rPB, (int)voffP, 0, 2);
                    dw = w;                                                          // (lane 63 waits for nobody)
                    if (dly) {
                        dw = *(const sb_v2i*)(fifo + foff);                          // what I produced 63 - lane steps ago ...
                        *(sb_v2i*)(fifo + foff) = w;                                 // ... makes room for this step's
                        foff = foff + 8 == flen ? 0 : foff + 8;
                    }
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, rPA, (int)voffP0, 0, 0);     // column 0
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)0, rPB, (int)voffP0, 0, 0);
                    if (ragged) {   // the lane that holds the matrix's last columns: byte by byte, undelayed
#pragma unroll
                        for (int k = 0; k < C - 1; ++k)
                            if (!full && k < nval && !(p.debug & 1)) {
                                const u32 off = (u32)((u - lane) * M + c0 + k);
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(pc[k] & 0xffu), rPA, (int)off, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((pc[k] >> 16) & 0xffu), rPB, (int)off, 0, 0);
                            }
                    }
                    voffP += pitchP; voffP0 += pitchP0;
                }
                if (bw) __builtin_amdgcn_raw_buffer_store_b32((int)h[C - 1], rB, lane == 63 ? 4 : (int)SB_OOB, 4 * u, 0);   // row u - 63 at index row + 64
                // ---- arg-max: tree of row maxima, first column that holds the maximum, strict update of the lane's record
                auto sel = [&](u32 b, u32 x, u32 y) { return pk_mad_u16(b, pk_sub_u16(y, x), x); };    // b ? y : x  (b = 0 / 1 per half)
                const u32 upk = (u32)__builtin_amdgcn_readfirstlane(u * 0x00010001);
                if constexpr (K12) {
                    // keys: score * 16 + (15 - column): the maximum key is the highest score in its lowest column; lbest holds the lane's best KEY
                    u32 n1[8], n2[4];
                    sb_for<0, 8>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        n1[i] = pk_max_u16(pk_mad_u16_sc(h[2 * i], sixteen2, (u32)(15 - 2 * i) * 0x00010001u), pk_mad_u16_sc(h[2 * i + 1], sixteen2, (u32)(14 - 2 * i) * 0x00010001u));
                    });
#pragma unroll
                    for (int i = 0; i < 4; ++i) n2[i] = pk_max_u16(n1[2 * i], n1[2 * i + 1]);
                    const u32 m = pk_max_u16(pk_max_u16(n2[0], n2[1]), pk_max_u16(n2[2], n2[3]));
                    const u32 imp = pk_min_u16(pk_subs_u16(m, lbest | 0x000F000Fu), one2);          // 1 where the SCORE beats the lane's best so far
                    lbest = sel(imp, lbest, m);
                    lstep = sel(imp, lstep, upk);
                } else {
                u32 n1[8], n2[4], n3[2];
#pragma unroll
                for (int i = 0; i < 8; ++i) n1[i] = pk_max_u16(h[2 * i], h[2 * i + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) n2[i] = pk_max_u16(n1[2 * i], n1[2 * i + 1]);
                n3[0] = pk_max_u16(n2[0], n2[1]); n3[1] = pk_max_u16(n2[2], n2[3]);
                const u32 m = pk_max_u16(n3[0], n3[1]);
                // b = 0 where the LEFT child holds the maximum (the lower columns win a tie), else 1; sel(b, x, y) = b ? y : x
                // (tried: the search only in steps that reach the pair's best so far, behind a wave-uniform threshold -- 7214 against 7170
                //  GCUPS: some lane of the 64 is near its pair's best in almost every step, the branch saves next to nothing)
                auto isnot = [&](u32 node) { return pk_min_u16(pk_sub_u16(m, node), one2); };
                const u32 b3 = isnot(n3[0]);
                const u32 b2 = isnot(sel(b3, n2[0], n2[2]));
                const u32 b1 = isnot(sel(b3, sel(b2, n1[0], n1[2]), sel(b2, n1[4], n1[6])));
                const u32 b0 = isnot(sel(b3, sel(b2, sel(b1, h[0], h[2]), sel(b1, h[4], h[6])), sel(b2, sel(b1, h[8], h[10]), sel(b1, h[12], h[14]))));
                const u32 two2 = 0x00020002u;
                const u32 kk = pk_mad_u16(pk_mad_u16(pk_mad_u16(b3, two2, b2), two2, b1), two2, b0);   // 8 b3 + 4 b2 + 2 b1 + b0
                const u32 imp = pk_min_u16(pk_subs_u16(m, lbest), one2);                              // 1 where m > the lane's best so far
                lbest = pk_max_u16(lbest, m);
                lk = sel(imp, lk, kk);
                lstep = sel(imp, lstep, upk);
                }
            });
        }
        {
            auto rec = [&](u32 best, u32 step, u32 k, u64& kb) {
                const int r = (int)step - lane, c = c0 + (int)k;
                if (best > 0u && r >= 1 && r <= rows && c <= cols) {
                    const u64 key = ((u64)best << 40) | (SW_KEY_IDX_MASK - (u64)(u32)(r * M + c));
                    kb = key > kb ? key : kb;
                }
            };
            if constexpr (K12) {
                rec((lbest & 0xffffu) >> 4, lstep & 0xffffu, 15u - (lbest & 15u), kbestA);
                rec(lbest >> 20, lstep >> 16, 15u - ((lbest >> 16) & 15u), kbestB);
            } else {
                rec(lbest & 0xffffu, lstep & 0xffffu, lk & 0xffffu, kbestA);
                rec(lbest >> 16, lstep >> 16, lk >> 16, kbestB);
            }
        }
    }
    // each pair's arg-max: highest score, lowest linear index among equals (serial_smithW.c:240-242)
    auto reduce = [&](u64 kb) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u32 olo = (u32)__shfl_xor((int)(u32)kb, off), ohi = (u32)__shfl_xor((int)(u32)(kb >> 32), off);
            const u64 o = ((u64)ohi << 32) | olo;
            kb = o > kb ? o : kb;
        }
        return kb;
    };
    kbestA = reduce(kbestA); kbestB = reduce(kbestB);
    if (lane == 0) {
        sw_result* res = p.results + pairA;
        res->max_score = (int64_t)(kbestA >> 40);
        res->max_pos = kbestA ? (int64_t)(SW_KEY_IDX_MASK - (kbestA & SW_KEY_IDX_MASK)) : 0;
        res->path_len = 0;
        if (pairB != pairA) {
            res = p.results + pairB;
            res->max_score = (int64_t)(kbestB >> 40);
            res->max_pos = kbestB ? (int64_t)(SW_KEY_IDX_MASK - (kbestB & SW_KEY_IDX_MASK)) : 0;
            res->path_len = 0;
        }
    }
}
template __global__ void sw_batch_wave16<true, true, false>(BatchParams);
template __global__ void sw_batch_wave16<true, false, false>(BatchParams);
template __global__ void sw_batch_wave16<false, true, false>(BatchParams);
template __global__ void sw_batch_wave16<false, false, false>(BatchParams);
template __global__ void sw_batch_wave16<true, true, true>(BatchParams);
template __global__ void sw_batch_wave16<false, true, true>(BatchParams);
template __global__ void sw_batch_wave16<true, false, true>(BatchParams);
template __global__ void sw_batch_wave16<false, false, true>(BatchParams);

#define SB_INST(C, PB) template __global__ void sw_batch_wave<C, PB>(BatchParams);
SB_INST(4, 0) SB_INST(4, 1) SB_INST(4, 4) SB_INST(8, 0) SB_INST(8, 1) SB_INST(8, 4) SB_INST(16, 0) SB_INST(16, 1) SB_INST(16, 4)
#undef SB_INST

}  // namespace swk
