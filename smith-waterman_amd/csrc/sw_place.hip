// sw_place.hip -- placement of the output matrices in HBM: probe kernel + helpers.
//
// Where the driver maps H and P in physical memory moves a store-bound fill by 25-35 %: physical HBM falls into a few coarse
// "classes" (regions of several GiB), and two store streams that go to the SAME class at the same time are slower than two
// streams into different classes.  sw_two_stream_probe reproduces the fill's store pattern -- one row segment per workgroup and
// row, the H and the P segment of a row back to back -- on two arbitrary buffers, so that a pair can be classified in tens of
// microseconds instead of with trial fills of the caller's problem.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "sw_kernels.h"

namespace swh { void set_err(const char* fmt, ...); }
using swh::set_err;

namespace swk {

// grid: nseg * nrg workgroups of 256 threads; workgroup (seg, rg) stores `seg_dw2` 8-byte elements of every row r = rg, rg + nrg, ...
// of X and (mode 0) of Y at byte offset r * pitch + seg * seg_dw2 * 8.  mode 1: X only; mode 2: X and X + half the rows (one buffer,
// two streams).  Wave v of the workgroup takes every 4th of the workgroup's rows.
__global__ void __launch_bounds__(256) sw_two_stream_probe(unsigned char* __restrict__ X, unsigned char* __restrict__ Y, int64_t rows, int64_t pitch,
                                                           int seg_dw2, int nseg, int nrg, int mode, unsigned int val) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int seg = (int)blockIdx.x % nseg, rg = (int)blockIdx.x / nseg;
    // experiments (mode bits): 16 = neighbouring segments on ONE XCD (workgroup b runs on XCD b % 8: segment = (b % 8) * per + b / 8, as the fill
    // deals its strips), 32 = segment s starts s * lag later (the strips of a fill are a hand-off apart), 64 = the first / last 8 lanes
    // (the pieces that share a 64-byte line with a neighbour) stored write-back, the others streaming
    const int mode0 = mode;
    const int lag_ns = (mode & 128) ? 0 : mode >> 8;
    const bool grouped = mode & 16, lagged = mode & 32, edges_wb = mode & 64;
    mode &= 15;
    if (grouped) {
        const int b = (int)blockIdx.x, nb = (int)gridDim.x, x = b & 7, k = b >> 3, per = (nseg + 7) / 8;
        if (nb == nseg * nrg) { const int idx = x * ((nb + 7) / 8) + k; seg = idx % nseg; rg = idx / nseg; if (idx >= nb) return; (void)per; }
    }
    if (mode0 & 128) {   // whole lines that start at lane k instead of lane 0 (k = bits 8..11; bit 12: k moves by one every second row)
        const int k0 = (mode0 >> 8) & 15;
        const bool vary = (mode0 >> 12) & 1;
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        const v2u v = {val, val + (unsigned)lane};
        for (int64_t r = rg + (int64_t)wave * nrg; r < rows; r += 4 * (int64_t)nrg) {
            const int k = vary ? (int)((k0 + (r >> 1)) & 7) : k0;
            if (lane >= k && lane < k + seg_dw2) {
                const int64_t o = r * pitch + ((int64_t)seg * seg_dw2 + lane - k) * 8;
                __builtin_nontemporal_store(v, (v2u*)(X + o));
                __builtin_nontemporal_store(v, (v2u*)(Y + o));
            }
        }
        return;
    }
    if (lane >= seg_dw2 && mode != 8) return;
    if (lagged) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        const uint64_t wait = (uint64_t)seg * (uint64_t)lag_ns / 10u;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    const int64_t col = ((int64_t)seg * seg_dw2 + lane) * 8;
    const int64_t half = (rows / 2) * pitch;
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    const v2u v = {val, val + (unsigned)lane};
    if (mode == 8) {   // the whole-line windows of overlapping strips: segment `seg` owns the lines that BEGIN inside its seg_dw2 * 8 bytes of the row
        for (int64_t r = rg + (int64_t)wave * nrg; r < rows; r += 4 * (int64_t)nrg) {
            const int64_t a0 = r * pitch + (int64_t)seg * seg_dw2 * 8 + 4, a1 = a0 + seg_dw2 * 8;
            const int64_t l0 = (a0 + 63) & ~(int64_t)63, l1 = (a1 + 63) & ~(int64_t)63;
            const int64_t o = l0 + lane * 8;
            if (o < l1) { __builtin_nontemporal_store(v, (v2u*)(X + o)); __builtin_nontemporal_store(v, (v2u*)(Y + o)); }
        }
        return;
    }
    if (edges_wb) {
        const bool edge = lane < 8 || lane >= seg_dw2 - 8;
        for (int64_t r = rg + (int64_t)wave * nrg; r < rows; r += 4 * (int64_t)nrg) {
            const int64_t o = r * pitch + col;
            if (edge) { *(v2u*)(X + o) = v; *(v2u*)(Y + o) = v; }
            else { __builtin_nontemporal_store(v, (v2u*)(X + o)); __builtin_nontemporal_store(v, (v2u*)(Y + o)); }
        }
        return;
    }
    if (mode == 4) {   // two streams, write-back stores
        for (int64_t r = rg + (int64_t)wave * nrg; r < rows; r += 4 * (int64_t)nrg) {
            const int64_t o = r * pitch + col;
            *(v2u*)(X + o) = v;
            *(v2u*)(Y + o) = v;
        }
        return;
    }
    for (int64_t r = rg + (int64_t)wave * nrg; r < (mode == 2 ? rows / 2 : rows); r += 4 * (int64_t)nrg) {
        const int64_t o = r * pitch + col;
        __builtin_nontemporal_store(v, (v2u*)(X + o));
        if (mode == 0) __builtin_nontemporal_store(v, (v2u*)(Y + o));
        else if (mode == 2) __builtin_nontemporal_store(v, (v2u*)(X + half + o));
    }
}

}  // namespace swk

extern "C" {

// (library-internal, experiments + the allocator) time `reps` launches of the probe on (d_X, d_Y): rows x pitch bytes each
int sw_probe_streams(sw_ctx* c, void* d_X, void* d_Y, int64_t rows, int64_t pitch, int seg_dw2, int nrg, int mode, int reps, float* ms) {
    if (!d_X || !ms || rows <= 0 || pitch <= 0 || seg_dw2 <= 0 || seg_dw2 > 64 || nrg <= 0 || reps <= 0) { set_err("sw_probe_streams: bad argument"); return SW_EINVAL; }
    const int nseg = (int)(pitch / (seg_dw2 * 8));
    if (nseg <= 0) { set_err("sw_probe_streams: bad argument"); return SW_EINVAL; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return SW_EDEVICE;
    hipLaunchKernelGGL(swk::sw_two_stream_probe, dim3(nseg * nrg), dim3(256), 0, nullptr, (unsigned char*)d_X, (unsigned char*)d_Y, rows, pitch, seg_dw2, nseg, nrg, mode, 1u);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(swk::sw_two_stream_probe, dim3(nseg * nrg), dim3(256), 0, nullptr, (unsigned char*)d_X, (unsigned char*)d_Y, rows, pitch, seg_dw2, nseg, nrg, mode, 2u + i);
    (void)hipEventRecord(e1, nullptr);
    const hipError_t e = hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (e != hipSuccess) { set_err("sw_probe_streams: %s", hipGetErrorString(e)); return SW_EDEVICE; }
    *ms = t / reps;
    return SW_OK;
}

// How much slower are two store streams into (X, Y) together than one stream into X alone?  ~1.3 when X and Y lie in different
// classes of the HBM, ~1.9 when they share one (profiles/r04_placement_classes_probe.log: 0.146 against 0.215 ms for 2 x 256 MiB, one
// stream alone 0.11 ms).  Samples `nwin` windows of `win` bytes at the same relative positions of the two buffers (a big matrix spans
// several classes; rows of H and P that are written together sit at the same relative position) and returns the mean ratio.  The probe
// WRITES the windows: both buffers must be fresh.  Default stream; ~0.2 ms per window.
int sw_place_pair_ratio(void* d_X, size_t xbytes, void* d_Y, size_t ybytes, float* ratio, float* ms_together) {
    if (!d_X || !d_Y || !ratio) { set_err("sw_place_pair_ratio: bad argument"); return SW_EINVAL; }
    const size_t win = std::min<size_t>(std::min(xbytes, ybytes), 128ull << 20);
    const int64_t pitch = 65540, rows = (int64_t)(win / (size_t)pitch);
    if (rows < 64) { *ratio = 1.f; if (ms_together) *ms_together = 0.f; return SW_OK; }
    const int nwin = std::max(xbytes, ybytes) > (6ull << 30) ? 8 : 1;
    float sum = 0.f, tsum = 0.f;
    for (int w = 0; w < nwin; ++w) {
        // window w starts at the fraction (w + 1/2) / nwin of each buffer (256-byte aligned), whole inside it
        auto at = [&](size_t bytes) { const size_t room = bytes - win; return ((size_t)((double)room * ((double)w + 0.5) / (double)nwin)) & ~(size_t)255; };
        unsigned char* X = (unsigned char*)d_X + (nwin == 1 ? 0 : at(xbytes));
        unsigned char* Y = (unsigned char*)d_Y + (nwin == 1 ? 0 : at(ybytes));
        float one = 0.f, two = 0.f;
        int rc = sw_probe_streams(nullptr, X, nullptr, rows, pitch, 63, 4, 1, 3, &one);
        if (rc == SW_OK) rc = sw_probe_streams(nullptr, X, Y, rows, pitch, 63, 4, 0, 3, &two);
        if (rc != SW_OK) return rc;
        sum += one > 0.f ? two / one : 1.f;
        tsum += two;
    }
    *ratio = sum / (float)nwin;
    if (ms_together) *ms_together = tsum / (float)nwin;
    return SW_OK;
}

}  // extern "C"
