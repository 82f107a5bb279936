// sw_kernels.hip -- CDNA4 (gfx950) kernels of the Smith-Waterman DP fill.
//
// What is computed (bit-exact with serial_smithW.c:187-256 of the reference):
//   H[i][j] = max(0, H[i-1][j-1] + s(i,j), H[i-1][j] + gap, H[i][j-1] + gap)
//   P[i][j] = first of {DIAGONAL, UP, LEFT} that attains a positive maximum, else NONE
//   maxPos  = lowest row-major index holding max H (0 if H == 0 everywhere)
//
// How (not a translation of any reference variant): the recurrence is rewritten in "G-space",
//   G[i][j] = H[i][j] - gap*(i + j)          (gap <= 0, so G >= H >= 0)
// where both gap moves become plain copies:
//   G[i][j] = max(Z, G[i-1][j-1] + s - 2*gap, G[i-1][j], G[i][j-1]),   Z = -gap*(i+j)  (H == 0)
// so a row is  e = max(Z, diag', up)  followed by an inclusive prefix-max along the row.
// One 64-lane wave owns a strip of 64 columns for all rows; the prefix-max is 6 DPP steps
// (row_shr 1/2/4/8, row_bcast15, row_bcast31), the diagonal/left neighbours are one wave_shr DPP
// each, rows are written to HBM row-major with one coalesced 256-B store per matrix per row.
// Strips are chained left-to-right: strip s needs the right-edge column of strip s-1, handed
// over through HBM/L2 as 8-byte {tag,value} granules (tag = epoch|row, written by ONE store,
// so the data is its own flag and no fence / vmcnt drain sits on the critical path).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "sw_kernels.h"

namespace swk {

// ---- cross-lane helpers ---------------------------------------------------------------------
typedef unsigned long long u64;
typedef unsigned int u32;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) u32 gu32;

// inclusive prefix max over the 64 lanes of a wave.  All G values are >= 0, so an unsigned max
// with identity 0 is exact and lets the compiler fold each step into ONE v_max_u32_dpp
// (row_shr 1/2/4/8 inside each 16-lane row, then row_bcast15 / row_bcast31 across rows).
__device__ __forceinline__ u32 wave_prefix_max(u32 v) {
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true)); // row_shr:1
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true)); // row_shr:2
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true)); // row_shr:4
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true)); // row_shr:8
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true)); // row_bcast:15 -> rows 1,3
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true)); // row_bcast:31 -> rows 2,3
    return v;
}
// wave_shr:1 -- lane l receives lane l-1; lane 0 keeps `old`.
__device__ __forceinline__ int dpp_wave_shr1(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false);
}

__device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(u64* p, u64 v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// v_writelane_b32 with a compile-time lane select (no builtin exists on this toolchain)
template <int R>
__device__ __forceinline__ int writelane_c(int old, int sval) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(R));
    return old;
}

// Prefetch of the next block's edge granule that the compiler does not track (so it cannot
// answer with s_waitcnt vmcnt(0), which would also drain this wave's H/P stores).  The landing
// registers are the LITERAL v[126:127]: the kernel uses < 80 VGPRs, the clobber makes the kernel
// descriptor reserve them, and no compiler value ever lives there, so the asynchronously
// arriving data cannot corrupt anything (hipcc copies/reuses an "=v" output before the data
// lands -- cdna_hip_programming.md 5.7 item 1).  tools/check_isa.py verifies nobody else
// touches v126/v127.
__device__ __forceinline__ void edge_prefetch_issue(const u64* p) {
    asm volatile("global_load_dwordx2 v[126:127], %0, off sc1" ::"v"(p) : "memory", "v126", "v127");
}
// Waits until at most NYOUNGER younger vector-memory ops are outstanding, then reads the granule.
// If the hardware ever completed a store ahead of the load the granule read here is stale, its
// tag does not match and the caller falls back to polling: never wrong, only slower.
template <int NYOUNGER>
__device__ __forceinline__ u64 edge_prefetch_collect() {
    u32 lo, hi;
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, v126\n\tv_mov_b32 %1, v127"
                 : "=v"(lo), "=v"(hi) : "n"(NYOUNGER) : "memory");
    return ((u64)hi << 32) | lo;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- the fill kernel ------------------------------------------------------------------------
// Per row of a strip (K = 1 column per lane), in G-space:
//   D  = G[i-1][j-1]  (wave_shr of U, lane 0 <- carry of the previous row)
//   d  = D + (a==b ? mm : xm);  z = cz + Zi;  e = max(d, U, z);  g = prefixmax(e) v carry
//   H = g - z;  P = g==z ? NONE : d==g ? DIAGONAL : U==g ? UP : LEFT      (serial_smithW.c:204-234)
template <typename HT, int B>
__global__ void __launch_bounds__(512) sw_strip_scan(const unsigned char* __restrict__ seq_a,
                                                     const unsigned char* __restrict__ seq_b, FillParams p) {
    static_assert(B == 16, "the scalar b-window and hand-counted vmcnt assume 16-row blocks");
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    const int TW = gridDim.x * wpb;
    const int64_t M = p.M, rows = p.rows;
    HT* H = (HT*)p.H;
    int32_t* P = p.P;
    const int mm = p.mm, xm = p.xm, ngap = p.ngap;
    const u64 tag_base = p.tag_base;
    const int64_t estride = rows + 1;
    constexpr u32 OOB = 0xFFFFFF00u;   // voffset beyond num_records: the buffer store is dropped

    for (int s = gw; s < p.nstrips; s += TW) {
        const u32 jo = 1u + (u32)s * 64u + (u32)lane;       // matrix column of this lane
        const bool jvalid = (int64_t)jo < M;
        const int a_l = jvalid ? (int)seq_a[jo - 1] : 0x100;
        const int cz = ngap * (int)jo;
        const bool first = (s == 0), last = (s + 1 == p.nstrips);
        const u64* ein = p.edge + (int64_t)(s - 1) * estride; // not dereferenced when first
        u64* eout = p.edge + (int64_t)s * estride;
        const u32 voffH = jvalid ? jo * (u32)sizeof(HT) : OOB;
        const u32 voffP = jvalid ? jo * 4u : OOB;

        // row 0 (halo row): H = top or 0, P = 0
        int U = cz;
        if (jvalid) {
            const int t = p.top ? p.top[jo] : 0;
            H[jo] = (HT)t;
            P[jo] = 0;
            U = t + cz;
        }
        if (first && lane == 0) { H[0] = (HT)(p.top ? p.top[0] : 0); P[0] = 0; }
        // G[0][column left of the strip]: the first diagonal carry; the halo row needs no hand-off
        const u32 jl = (u32)s * 64u;
        int carry_prev = (p.top ? p.top[jl] : 0) + ngap * (int)jl;

        int bestv = jvalid ? 0 : 0x7fffffff, bestrow = 0;

        // Edge granules for rows [i0, i0+16) of the strip to the left, one per lane < 16.
        // Strip 0 synthesises them: column 0 has H == 0, i.e. G = -gap*i.
        auto edge_addr = [&](int64_t i0) -> const u64* {
            int64_t i = i0 + (lane & 15);
            if (i > rows) i = rows;
            return ein + i;
        };
        u64 gcur = 0;
        if (!first) gcur = granule_load(edge_addr(1));

        for (int64_t i0 = 1; i0 <= rows; i0 += B) {
            const int nb = (int)min((int64_t)B, rows - i0 + 1);
            const bool full = (nb == B);
            // --- prefetch the next block's granules with a load the compiler does not track, so
            // that it cannot answer with s_waitcnt vmcnt(0) (which would also drain our stores).
            u64 gnext = 0;
            const bool have_next = !first && (i0 + B <= rows);
            if (have_next) edge_prefetch_issue(edge_addr(i0 + B));
            // --- this block's b characters: 16 bytes through the scalar cache (b is read-only)
            u32 bw0, bw1, bw2, bw3;
            if (full) {
                const uint4 w = *reinterpret_cast<const uint4*>(seq_b + (i0 - 1));
                bw0 = w.x; bw1 = w.y; bw2 = w.z; bw3 = w.w;
            } else {
                u32 t[4] = {0, 0, 0, 0};
                for (int r = 0; r < nb; ++r) t[r >> 2] |= (u32)seq_b[i0 - 1 + r] << (8 * (r & 3));
                bw0 = t[0]; bw1 = t[1]; bw2 = t[2]; bw3 = t[3];
            }

            int Ein;
            if (first) {
                Ein = ngap * ((int)i0 + lane);
            } else {
                // every granule of this block must carry this launch's tag for its row.  Fast path:
                // the prefetched granules already do.  Slow path (peeled so that its compiler-tracked
                // loads put no s_waitcnt vmcnt(0) on the fast path): re-poll with sc1 loads.
                auto tags_ok = [&](u64 g) -> bool {
                    return ((lane & 15) >= nb) || ((g >> 32) == (tag_base | (u64)(i0 + (lane & 15))));
                };
                if (!__all(tags_ok(gcur))) {
                    unsigned spins = 0;
                    uint64_t t0 = 0;
                    for (;;) {
                        __builtin_amdgcn_s_sleep(1);
                        gcur = granule_load(edge_addr(i0));
                        if (__all(tags_ok(gcur))) break;
                        if ((++spins & 255u) == 0) {
                            const uint64_t now = __builtin_amdgcn_s_memrealtime(); // 100 MHz
                            if (t0 == 0) t0 = now;
                            const bool expired = (now - t0) > 300000000ull;        // 3 s
                            if (expired) __hip_atomic_store((gu32*)p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (expired || __hip_atomic_load((gu32*)p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                return;
                            }
                        }
                    }
                }
                Ein = (int)(u32)gcur;
            }
            int Eout = 0;
            // column 0 of this block's rows (H == 0, P == NONE), once per block by strip 0
            if (first && lane < nb) {
                H[(i0 + lane) * M] = (HT)0;
                P[(i0 + lane) * M] = 0;
            }
            // buffer descriptors for this block's rows: base = row i0, 32-bit row offsets below
            const __amdgpu_buffer_rsrc_t rH =
                __builtin_amdgcn_make_buffer_rsrc((void*)(H + i0 * M), 0, 0x7FFFFF00, 0x00020000);
            const __amdgpu_buffer_rsrc_t rP =
                __builtin_amdgcn_make_buffer_rsrc((void*)(P + i0 * M), 0, 0x7FFFFF00, 0x00020000);
            const u32 rowH = (u32)(M * (int64_t)sizeof(HT)), rowP = (u32)(M * 4);
            int Zi = ngap * (int)i0;

            auto do_row = [&](int r, auto store_edge) {
                const u32 bw = (r < 4) ? bw0 : (r < 8) ? bw1 : (r < 12) ? bw2 : bw3;
                const int b_i = (int)((bw >> (8 * (r & 3))) & 0xffu);
                const int carry_cur = __builtin_amdgcn_readlane(Ein, r);
                const int D = dpp_wave_shr1(carry_prev, U);          // G[i-1][j-1]
                const int d = D + ((a_l == b_i) ? mm : xm);
                const int z = cz + Zi;
                const u32 e = (u32)max(max(d, U), z);
                const int g = (int)max(wave_prefix_max(e), (u32)carry_cur);
                const int h = g - z;
                const int pred = (g == z) ? 0 : (d == g) ? 3 : (U == g) ? 1 : 2;
                if constexpr (sizeof(HT) == 8) {
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    v2i hv; hv.x = h; hv.y = h >> 31;
                    __builtin_amdgcn_raw_buffer_store_b64(hv, rH, voffH, (int)(rowH * (u32)r), 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(h, rH, voffH, (int)(rowH * (u32)r), 0);
                }
                __builtin_amdgcn_raw_buffer_store_b32(pred, rP, voffP, (int)(rowP * (u32)r), 0);
                if (h > bestv) { bestv = h; bestrow = (int)i0 + r; }
                store_edge(__builtin_amdgcn_readlane(g, 63));
                U = g;
                carry_prev = carry_cur;
                Zi += ngap;
            };
            if (full) {
                static_for<0, B>([&](auto R) {
                    do_row(R.value, [&](int g63) { Eout = writelane_c<R.value>(Eout, g63); });
                });
            } else {
                for (int r = 0; r < nb; ++r) do_row(r, [&](int g63) { Eout = (lane == r) ? g63 : Eout; });
            }
            // the prefetched granules were issued before this block's 2*B row stores: wait for
            // everything older than those stores (vmcnt counts loads and stores in issue order)
            if (have_next) gnext = edge_prefetch_collect<2 * B>();
            if (!last && lane < nb)
                granule_store(eout + i0 + lane, ((tag_base | (u64)(i0 + lane)) << 32) | (u64)(u32)Eout);
            gcur = gnext;
        }

        if (jvalid && bestv > 0) {
            const u64 idx = (u64)bestrow * (u64)M + (u64)jo;
            const u64 key = ((u64)(u32)bestv << 40) | (SW_KEY_IDX_MASK - idx);
            atomicMax(p.result_key, key);
        }
    }
}

template __global__ void sw_strip_scan<int32_t, 16>(const unsigned char*, const unsigned char*, FillParams);
template __global__ void sw_strip_scan<int64_t, 16>(const unsigned char*, const unsigned char*, FillParams);

// ---- small kernels --------------------------------------------------------------------------
__global__ void sw_finalize(const unsigned long long* key, const unsigned int* abort_flag, sw_result* res, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = key[i];
    res[i].max_score = (int64_t)(k >> 40);
    res[i].max_pos = k ? (int64_t)(SW_KEY_IDX_MASK - (k & SW_KEY_IDX_MASK)) : 0;
    res[i].path_len = *abort_flag ? -(int64_t)*abort_flag : 0;   // (< 0: a hand-off wait gave up; the value says which kind)
}

// compact predecessor matrix -> the reference's int32 layout (same codes, sign-extended): 16 codes per thread and pass
__global__ void sw_widen_p8(const signed char* __restrict__ P8, int32_t* __restrict__ P32, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n && (((uintptr_t)(P8 + i)) & 15) == 0 && (((uintptr_t)(P32 + i)) & 15) == 0) {
            const int4 v = *(const int4*)(P8 + i);
            const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(int4*)(P32 + i + 4 * q) = make_int4((int)(signed char)(w[q]), (int)(signed char)(w[q] >> 8), (int)(signed char)(w[q] >> 16), (int)(signed char)(w[q] >> 24));
        } else {
            for (size_t k = i; k < n && k < i + 16; ++k) P32[k] = (int32_t)P8[k];
        }
    }
}

// ---- 2-bit predecessor matrix: 4 codes per byte, cell k in bits 2 (k & 3) .. 2 (k & 3) + 1 of byte k >> 2.  Two bits
// cannot hold the sign the traceback leaves on a path (P *= PATH, serial_smithW.c:271), so a traced path is a bitmap beside it:
// bit k & 31 of word k >> 5.  One thread per 32 cells: 8 bytes of codes + one bitmap word.
template <typename PT>
__global__ void __launch_bounds__(256) sw_pack_p2(const PT* __restrict__ P, unsigned char* __restrict__ P2, unsigned int* __restrict__ bits, size_t n) {
    const size_t nw = (n + 31) / 32, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) {
        const size_t k0 = w * 32;
        u64 codes = 0;
        unsigned int neg = 0;
        if (sizeof(PT) == 1 && k0 + 32 <= n && (((uintptr_t)(P + k0)) & 15) == 0) {
            const int4 v0 = *(const int4*)(P + k0), v1 = *(const int4*)(P + k0 + 16);
            const int wv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int c = (int)(signed char)(wv[q >> 2] >> (8 * (q & 3)));
                codes |= (u64)(unsigned)((c < 0 ? -c : c) & 3) << (2 * q);
                neg |= (c < 0 ? 1u : 0u) << q;
            }
        } else {
#pragma unroll 8
            for (int q = 0; q < 32; ++q) {
                const int c = (k0 + q < n) ? (int)P[k0 + q] : 0;
                codes |= (u64)(unsigned)((c < 0 ? -c : c) & 3) << (2 * q);
                neg |= (c < 0 ? 1u : 0u) << q;
            }
        }
        *(u64*)(P2 + w * 8) = codes;
        if (bits) bits[w] = neg;
    }
}
template __global__ void sw_pack_p2<signed char>(const signed char*, unsigned char*, unsigned int*, size_t);
template __global__ void sw_pack_p2<int32_t>(const int32_t*, unsigned char*, unsigned int*, size_t);

// back to the reference's int32 layout: P32[k] = code, negated where the bitmap marks the path.  One thread per byte (4 cells, 16 bytes out).
__global__ void __launch_bounds__(256) sw_unpack_p2(const unsigned char* __restrict__ P2, const unsigned int* __restrict__ bits, int32_t* __restrict__ P32, size_t n) {
    const size_t nb = (n + 3) / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += stride) {
        const unsigned v = P2[b];
        const unsigned m = bits ? (bits[b >> 3] >> (4 * (b & 7))) & 15u : 0u;
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int code = (int)((v >> (2 * q)) & 3u);
            c[q] = ((m >> q) & 1u) ? -code : code;
        }
        const size_t k = b * 4;
        if (k + 4 <= n && (((uintptr_t)(P32 + k)) & 15) == 0) *(int4*)(P32 + k) = make_int4(c[0], c[1], c[2], c[3]);
        else
            for (int q = 0; q < 4 && k + q < n; ++q) P32[k + q] = c[q];
    }
}

// cs[i] = sum_j (u64)(u32)X[i][j] * ((j+1) * 0x9E3779B97F4A7C15); one block per row
template <typename T>
__global__ void __launch_bounds__(256) sw_row_checksums(const T* __restrict__ X, int64_t m, u64* __restrict__ cs) {
    const int64_t i = blockIdx.x;
    const T* row = X + i * m;
    u64 acc = 0;
    bool bad = false;
    for (int64_t j = threadIdx.x; j < m; j += blockDim.x) {
        const T v = row[j];
        if (sizeof(T) == 8 && (int64_t)v != (int64_t)(int32_t)v) bad = true;
        acc += (u64)(uint32_t)(int32_t)v * ((u64)(j + 1) * 0x9E3779B97F4A7C15ull);
    }
    __shared__ u64 red[256];
    __shared__ int anybad;
    if (threadIdx.x == 0) anybad = 0;
    __syncthreads();
    if (bad) anybad = 1;
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) cs[i] = anybad ? ~0ull : red[0];
}
template __global__ void sw_row_checksums<int32_t>(const int32_t*, int64_t, u64*);
template __global__ void sw_row_checksums<int64_t>(const int64_t*, int64_t, u64*);
template __global__ void sw_row_checksums<signed char>(const signed char*, int64_t, u64*);

}  // namespace swk
