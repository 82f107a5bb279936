// sw_multi.hip -- ONE DP matrix over several GPUs of one process, behind the C-ABI.
//
// The reference has no multi-GPU code (its only hint is "consider atomicCAS_system for multi GPU systems",
// simple-cuda/sw-default-discrete.cu:280).  Decomposition: contiguous ROW BANDS, one per device; every band is ONE
// band-resident launch (sw_fill_band_device), all launched at once.  The last row of band g leaves its kernel as
// {tag, H} granules plus a per-strip flag in host-pinned memory; a relay loop on the calling thread forwards finished
// column chunks with hipMemcpyPeerAsync into the granule buffer band g+1's kernel is polling (xGMI peer copies: no CU of
// either GPU is needed for the transfer), so a band's strips start the moment their halo lands.  The global arg-max is
// the maximum of the per-band packed keys; the traceback hops from band to band at the halo rows.
// (One process per GPU over RCCL: smith-waterman_amd/multi.py drives the same band entry point.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "sw_kernels.h"

namespace swh { void set_err(const char* fmt, ...); }
using swh::set_err;

#define HIP_TRYM(expr)                                                                \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SW_EDEVICE;                                                        \
        }                                                                             \
    } while (0)

struct sw_multi_band {
    int device = 0;
    sw_ctx* ctx = nullptr;
    int64_t lo = 0, hi = 0;            // global rows lo+1 .. hi
    char *d_a = nullptr, *d_b = nullptr;
    void *d_H = nullptr, *d_P = nullptr;
    uint64_t *d_top = nullptr, *d_bot = nullptr;
    bool host_gran = false;            // granule buffers in host-pinned memory (bands sharing one GPU: see sw_multi_create)
    bool placed = false;               // d_H / d_P come from sw_alloc_outputs (a band with a GPU of its own)
    uint32_t* h_done = nullptr;        // host-pinned, one flag per strip
    sw_result* d_res = nullptr;
    int64_t* d_stop = nullptr;         // where this band's part of a traceback stopped (sw_multi_traceback)
    hipStream_t stream = nullptr, copy = nullptr;
    sw_result res = {0, 0, 0};
};

struct sw_multi {
    std::vector<sw_multi_band> bands;
    int64_t cols = 0, rows = 0;
    int p_elem_bytes = 4, h_elem_bytes = 4;
    bool want_h = true;
    uint32_t tag = 0;
    sw_result result = {0, 0, 0};
    double fill_seconds = 0;
};

extern "C" {

int sw_traceback_stop_device(sw_ctx* c, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos, sw_result* d_result, int64_t* d_stop,
                             void* stream);   // sw_api.hip (library-internal)
int sw_fill_band_reserve(sw_ctx* c, int64_t cols, int64_t rows, int64_t total_rows, const sw_scores* scores, int h_elem_bytes, int p_elem_bytes, int want_h,
                         void* stream);       // sw_api.hip (library-internal)

void sw_multi_free(sw_multi* m) {
    if (!m) return;
    for (auto& b : m->bands) {
        (void)hipSetDevice(b.device);
        if (b.stream) (void)hipStreamDestroy(b.stream);
        if (b.copy) (void)hipStreamDestroy(b.copy);
        (void)hipFree(b.d_a); (void)hipFree(b.d_b);
        if (b.placed && b.ctx) (void)sw_free_outputs(b.ctx, b.d_H, b.d_P);
        else { (void)hipFree(b.d_H); (void)hipFree(b.d_P); }
        if (b.host_gran) { if (b.d_top) (void)hipHostFree(b.d_top); if (b.d_bot) (void)hipHostFree(b.d_bot); }
        else { (void)hipFree(b.d_top); (void)hipFree(b.d_bot); }
        (void)hipFree(b.d_res); (void)hipFree(b.d_stop);
        if (b.h_done) (void)hipHostFree(b.h_done);
        if (b.ctx) sw_destroy(b.ctx);
    }
    delete m;
}

// Bands over devices[0..ndev-1] (an id may repeat: several bands then share that GPU, each launch capped to its share of
// the CUs -- how the relay is tested on a one-GPU box).  a, b: HOST sequences.  Allocates the band-local matrices.
int sw_multi_create(const int* devices, int ndev, const char* a, int64_t cols, const char* b, int64_t rows, int p_elem_bytes, int want_h,
                    sw_multi** out) {
    if (!devices || ndev <= 0 || !out || cols <= 0 || rows <= 0 || !a || !b || (p_elem_bytes != 4 && p_elem_bytes != 1)) {
        set_err("sw_multi_create: bad argument");
        return SW_EINVAL;
    }
    sw_multi* m = new sw_multi();
    m->cols = cols; m->rows = rows; m->p_elem_bytes = p_elem_bytes; m->want_h = want_h != 0;
    int64_t per = (rows + ndev - 1) / ndev;
    per = (per + 15) / 16 * 16;       // band rows start on 16-byte windows of b
    const int64_t S = (cols + 62) / 63;
    int rc = SW_OK;
    for (int g = 0; g < ndev && rc == SW_OK; ++g) {
        sw_multi_band bd;
        bd.device = devices[g];
        bd.lo = std::min<int64_t>((int64_t)g * per, rows);
        bd.hi = std::min<int64_t>((int64_t)(g + 1) * per, rows);
        m->bands.push_back(bd);
    }
    while (!m->bands.empty() && m->bands.back().hi == m->bands.back().lo) m->bands.pop_back();
    const int nb = (int)m->bands.size();
    for (int g = 0; g < nb && rc == SW_OK; ++g) {
        sw_multi_band& bd = m->bands[g];
        const int64_t br = bd.hi - bd.lo;
        const size_t cells = (size_t)(br + 1) * (size_t)(cols + 1);
        auto dev_alloc = [&](void** p, size_t n) { return hipMalloc(p, n ? n : 1) == hipSuccess; };
        if (hipSetDevice(bd.device) != hipSuccess) { set_err("sw_multi_create: device %d not usable", bd.device); rc = SW_ENODEV; break; }
        if ((rc = sw_create(bd.device, &bd.ctx)) != SW_OK) break;
        // Bands that share a GPU (one-GPU test boxes) keep their granule buffers in host-pinned memory and the relay is a
        // plain memcpy: with several persistent kernels and copy streams on ONE device, an async copy can be mapped to the
        // hardware queue of a polling kernel and wait behind it.  Distinct GPUs: device memory + peer copies.
        int share0 = 0;
        for (int k = 0; k < nb; ++k) share0 += (m->bands[k].device == bd.device);
        bd.host_gran = share0 > 1;
        auto gran_alloc = [&](uint64_t** p) {
            // Distinct GPUs: the granules are written by a PEER (hipMemcpyPeerAsync over xGMI) while this GPU's kernel polls them.  A peer's
            // write does not pass through this GPU's L2, so the buffer must be fine-grained memory -- never cached there; ordinary
            // (coarse-grained) device memory is only guaranteed coherent with other agents at kernel boundaries.  The polling load
            // (s2_load_top / the importers of sw_systolic: system scope, sc0 sc1) then always goes to memory.
            if (!bd.host_gran) {
                if (hipExtMallocWithFlags((void**)p, (size_t)(cols + 1) * 8, hipDeviceMallocFinegrained) == hipSuccess) return true;
                (void)hipGetLastError();
                return dev_alloc((void**)p, (size_t)(cols + 1) * 8);
            }
            if (hipHostMalloc((void**)p, (size_t)(cols + 1) * 8, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) return false;
            memset(*p, 0, (size_t)(cols + 1) * 8);
            return true;
        };
        bool ok = dev_alloc((void**)&bd.d_a, (size_t)cols + 16) && dev_alloc((void**)&bd.d_b, (size_t)br + 16) &&
                  hipMemcpy(bd.d_a, a, (size_t)cols, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(bd.d_b, b + bd.lo, (size_t)br, hipMemcpyHostToDevice) == hipSuccess;
        // A band with a GPU of its own takes H and P from the placement-aware allocator (H and P in different classes of the HBM: where
        // the two matrices lie moves a fill by 15-30 %; candidates are classified by a store probe, no trial fills).
        // Bands sharing a GPU take plain pairs: the probe wants the device to itself.
        if (ok && want_h && share0 == 1) {
            ok = sw_alloc_outputs(bd.ctx, bd.d_a, cols, bd.d_b, br, nullptr, 4, p_elem_bytes, 0, &bd.d_H, &bd.d_P, nullptr) == SW_OK;
            bd.placed = ok;
        } else {
            ok = ok && (!want_h || dev_alloc(&bd.d_H, cells * 4)) && dev_alloc(&bd.d_P, cells * (size_t)p_elem_bytes);
        }
        ok = ok && dev_alloc((void**)&bd.d_res, sizeof(sw_result)) && dev_alloc((void**)&bd.d_stop, 8) && (g == 0 || gran_alloc(&bd.d_top)) && (g == nb - 1 || gran_alloc(&bd.d_bot));
        if (!ok) { (void)hipGetLastError(); set_err("sw_multi_create: band %d does not fit device %d", g, bd.device); rc = SW_ENOMEM; break; }
        if (g < nb - 1 && hipHostMalloc((void**)&bd.h_done, (size_t)S * 4, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
            set_err("sw_multi_create: pinned allocation failed"); rc = SW_ENOMEM; break;
        }
        if (bd.h_done) memset(bd.h_done, 0, (size_t)S * 4);
        if (hipStreamCreateWithFlags(&bd.stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&bd.copy, hipStreamNonBlocking) != hipSuccess ||
            hipMemset(bd.d_P, 0, (size_t)(cols + 1) * (size_t)p_elem_bytes) != hipSuccess ||   // row 0 of a band belongs to the band above: NONE stops a walk there
            (bd.d_top && !bd.host_gran && hipMemset(bd.d_top, 0, (size_t)(cols + 1) * 8) != hipSuccess) ||
            (bd.d_bot && !bd.host_gran && hipMemset(bd.d_bot, 0, (size_t)(cols + 1) * 8) != hipSuccess)) {
            set_err("sw_multi_create: device setup failed"); rc = SW_EDEVICE; break;
        }
        // every workspace a band launch needs exists before the first fill: no launch call may have to synchronise later
        if ((rc = sw_fill_band_reserve(bd.ctx, cols, br, rows, nullptr, 4, p_elem_bytes, want_h, bd.stream)) != SW_OK) break;
        // bands that share a GPU split its CUs (every workgroup of a launch must be resident)
        int share = 0;
        for (int k = 0; k < nb; ++k) share += (m->bands[k].device == bd.device);
        if (share > 1) sw_set_option(bd.ctx, "max_blocks", std::max<int64_t>(8, sw_get_option(bd.ctx, "num_cus") / share - 8));
        // peer access for the halo copies
        for (int k = 0; k < nb; ++k)
            if (m->bands[k].device != bd.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, bd.device, m->bands[k].device) == hipSuccess && can) {
                    hipError_t e = hipDeviceEnablePeerAccess(m->bands[k].device, 0);
                    if (e != hipSuccess) (void)hipGetLastError();   // already enabled
                }
            }
    }
    if (rc != SW_OK) { sw_multi_free(m); return rc; }
    *out = m;
    return SW_OK;
}

// One fill of the whole matrix: every band launched at once, halo rows relayed chunk by chunk.  Blocks until done.
int sw_multi_fill(sw_multi* m, const sw_scores* scores, int nchunks, sw_result* result) {
    if (!m || m->bands.empty()) { set_err("sw_multi_fill: bad argument"); return SW_EINVAL; }
    const int nb = (int)m->bands.size();
    const int64_t cols = m->cols, S = (cols + 62) / 63;
    if (nchunks <= 0) nchunks = 64;
    const int64_t per = std::max<int64_t>(1, (S + nchunks - 1) / nchunks);
    if (++m->tag == 0) m->tag = 1;
    const uint32_t tag = m->tag;
    const auto t0 = std::chrono::steady_clock::now();
    // launch in band order: a band never waits for a later one, so even if a launch call blocks (the first fill sizes the
    // per-context workspaces, and an allocation may wait for the device) the bands before it can always finish
    for (int g = 0; g < nb; ++g) {
        sw_multi_band& bd = m->bands[g];
        HIP_TRYM(hipSetDevice(bd.device));
        int share = 0;
        for (int k = 0; k < nb; ++k) share += (m->bands[k].device == bd.device);
        // a band with a GPU of its own leaves a few CUs free: a peer copy normally runs on the SDMA engines, but where the runtime falls
        // back to a blit kernel that kernel needs a CU beside the resident band kernel (multi.py reserves the same 16 for RCCL's kernels)
        const int reserve = (share == 1 && nb > 1) ? 16 : 0;
        const int rc = sw_fill_band_device(bd.ctx, bd.d_a, cols, bd.d_b, bd.hi - bd.lo, m->rows, scores, bd.d_H, 4, bd.d_P, m->p_elem_bytes,
                                           bd.d_top, bd.d_top ? tag : 0, bd.d_bot, bd.d_bot ? tag : 0, bd.h_done, reserve, share > 1, bd.d_res, bd.stream);
        if (rc != SW_OK) {   // bands already launched would wait for a halo that never comes: they give up after "band_wait_ms"
            for (int k = 0; k < g; ++k) { (void)hipSetDevice(m->bands[k].device); (void)hipStreamSynchronize(m->bands[k].stream); }
            return rc;
        }
    }
    // relay: chunk k of band g's last row goes to band g+1 as soon as all its strips have raised their flag.  A granule
    // {tag, H} is valid only as a whole, so (a) chunks are cut at EVEN granule indices -- every piece starts and ends on a
    // 16-byte boundary and no transport has a reason to move a granule in two parts -- and (b) the host-memory path (bands
    // that share a GPU) copies granule by granule with 8-byte atomic stores instead of memcpy.
    std::vector<int64_t> next(nb, 0);   // next strip to forward per band
    bool busy = nb > 1;
    auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);   // patience WITHOUT progress
    auto give_up = [&](int rc) {
        // leave no band kernel in flight: raise every context's abort flag is not reachable from here, but a band that
        // never gets its halo gives up by itself after "band_wait_ms"; wait for all of them before returning
        for (auto& bd : m->bands) { (void)hipSetDevice(bd.device); (void)hipStreamSynchronize(bd.copy); (void)hipStreamSynchronize(bd.stream); }
        return rc;
    };
    while (busy) {
        busy = false;
        bool moved = false;
        for (int g = 0; g + 1 < nb; ++g) {
            sw_multi_band& src = m->bands[g];
            sw_multi_band& dst = m->bands[g + 1];
            while (next[g] < S) {
                const int64_t s0 = next[g], s1 = std::min(S, s0 + per);
                bool ready = true;
                for (int64_t s = s0; s < s1 && ready; ++s) ready = (__atomic_load_n(&src.h_done[s], __ATOMIC_ACQUIRE) == tag);
                if (!ready) break;
                // strips s0 .. s1-1 cover columns 63 s0 + 1 .. 63 s1; cut at even indices: a chunk starts at the even index at or
                // below its first column (that granule belongs to a finished strip) and ends below the even index at or
                // below the next chunk's first column (the last chunk ends at cols + 1)
                const int64_t c0 = (s0 == 0) ? 0 : ((63 * s0 + 1) & ~1ll);
                const int64_t c1 = (s1 >= S) ? cols + 1 : ((63 * s1 + 1) & ~1ll);
                HIP_TRYM(hipSetDevice(dst.device));
                if (src.host_gran && dst.host_gran) {
                    for (int64_t c = c0; c < c1; ++c)
                        __atomic_store_n(dst.d_top + c, __atomic_load_n(src.d_bot + c, __ATOMIC_RELAXED), __ATOMIC_RELAXED);   // both in host-pinned memory
                    __atomic_thread_fence(__ATOMIC_RELEASE);
                } else if (c1 > c0) {
                    const hipError_t e = hipMemcpyPeerAsync(dst.d_top + c0, dst.device, src.d_bot + c0, src.device, (size_t)(c1 - c0) * 8, dst.copy);
                    if (e != hipSuccess) { set_err("sw_multi_fill: peer copy failed: %s", hipGetErrorString(e)); return give_up(SW_EDEVICE); }
                }
                next[g] = s1;
                moved = true;
            }
            if (next[g] < S) busy = true;
        }
        if (moved) deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
        else if (busy && std::chrono::steady_clock::now() > deadline) {
            char msg[512]; int n = snprintf(msg, sizeof msg, "sw_multi_fill: the band pipeline stalled; strips forwarded per band:");
            for (int g = 0; g + 1 < nb && n < (int)sizeof msg - 32; ++g) n += snprintf(msg + n, sizeof msg - n, " %lld/%lld", (long long)next[g], (long long)S);
            set_err("%s", msg);
            return give_up(SW_ETIMEOUT);
        }
    }
    if (getenv("SW_MULTI_DEBUG")) {
        fprintf(stderr, "sw_multi_fill: relay done after %.3f s;", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        for (int g = 0; g + 1 < nb; ++g) fprintf(stderr, " band %d forwarded %lld/%lld strips (flag[0]=%u tag=%u);", g, (long long)next[g], (long long)S, m->bands[g].h_done[0], tag);
        fprintf(stderr, "\n");
    }
    uint64_t best = 0;
    for (int g = 0; g < nb; ++g) {
        sw_multi_band& bd = m->bands[g];
        HIP_TRYM(hipSetDevice(bd.device));
        HIP_TRYM(hipStreamSynchronize(bd.copy));
        HIP_TRYM(hipStreamSynchronize(bd.stream));
        HIP_TRYM(hipMemcpy(&bd.res, bd.d_res, sizeof(sw_result), hipMemcpyDeviceToHost));
        if (bd.res.path_len < 0) { set_err("sw_multi_fill: band %d: hand-off wait timed out", g); return SW_ETIMEOUT; }
        if (bd.res.max_score > 0) {
            const int64_t row = bd.res.max_pos / (cols + 1), col = bd.res.max_pos % (cols + 1);
            const uint64_t gidx = (uint64_t)(bd.lo + row) * (uint64_t)(cols + 1) + (uint64_t)col;
            best = std::max<uint64_t>(best, ((uint64_t)bd.res.max_score << 40) | (uint64_t)(swk::SW_KEY_IDX_MASK - gidx));   // serial_smithW.c:240-242
        }
    }
    m->fill_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m->result.max_score = (int64_t)(best >> 40);
    m->result.max_pos = best ? (int64_t)(swk::SW_KEY_IDX_MASK - (best & swk::SW_KEY_IDX_MASK)) : 0;
    m->result.path_len = 0;
    if (result) *result = m->result;
    return SW_OK;
}

// backtrack() (serial_smithW.c:262-277) across the bands: the band holding max_pos walks first; when the walk reaches a
// band's halo row the band above takes over at its last row.  Negates P along the path; returns the total length.
int sw_multi_traceback(sw_multi* m, int64_t* path_len) {
    if (!m || m->bands.empty()) { set_err("sw_multi_traceback: bad argument"); return SW_EINVAL; }
    const int64_t M = m->cols + 1;
    int64_t row = m->result.max_pos / M, col = m->result.max_pos % M, total = 0;
    bool walking = m->result.max_score > 0;
    for (int g = (int)m->bands.size() - 1; g >= 0 && walking; --g) {
        sw_multi_band& bd = m->bands[g];
        if (!(bd.lo < row && row <= bd.hi)) continue;
        HIP_TRYM(hipSetDevice(bd.device));
        // the band's walk stops at the first cell with P <= 0: inside the band (the end of the path) or in the band's row 0,
        // which belongs to the band above (its P is kept NONE here) -- then that band takes over at its last row
        int rc = sw_traceback_stop_device(bd.ctx, bd.d_P, m->p_elem_bytes, m->cols, bd.hi - bd.lo, (row - bd.lo) * M + col, bd.d_res, bd.d_stop, bd.stream);
        if (rc != SW_OK) return rc;
        sw_result r = {0, 0, 0};
        int64_t stop = 0;
        HIP_TRYM(hipStreamSynchronize(bd.stream));
        HIP_TRYM(hipMemcpy(&r, bd.d_res, sizeof r, hipMemcpyDeviceToHost));
        HIP_TRYM(hipMemcpy(&stop, bd.d_stop, 8, hipMemcpyDeviceToHost));
        walking = false;
        if (r.path_len > 0) {
            total += r.path_len;
            const int64_t r2 = stop / M, c2 = stop % M;
            if (r2 == 0 && bd.lo > 0 && c2 > 0) { row = bd.lo; col = c2; walking = true; }   // crossed into the band above
        }
    }
    m->result.path_len = total;
    if (path_len) *path_len = total;
    return SW_OK;
}

int sw_multi_band_info(sw_multi* m, int g, int* device, int64_t* row_lo, int64_t* row_hi, void** d_H, void** d_P) {
    if (!m || g < 0 || g >= (int)m->bands.size()) { set_err("sw_multi_band_info: bad argument"); return SW_EINVAL; }
    const sw_multi_band& bd = m->bands[g];
    if (device) *device = bd.device;
    if (row_lo) *row_lo = bd.lo;
    if (row_hi) *row_hi = bd.hi;
    if (d_H) *d_H = bd.d_H;
    if (d_P) *d_P = bd.d_P;
    return SW_OK;
}

// Adaptive dispatch with all three executors of omp_smithW-v7-adaptive.cpp:304-396 (serial / OpenMP threads / offload, chosen there per
// anti-diagonal by its length): here the whole problem is sized once --
//   0  the host fill (sw_fill_cpu) below ~2e5 cells: launch + transfer latency would dominate;
//   1  ONE GPU (ctx, or devices[0]) while a single pair is bound by its strip chain, which more GPUs do not shorten;
//   2  row bands over ALL the given devices (sw_multi_*) from multi_min_cells cells on (0: 4e9, about 65536^2, where a fill is
//      bound by the HBM stores of one GPU), or whenever H + P do not fit the first device.
// H, P, max_pos, max_score and path_len come back exactly as serial_smithW leaves them (the traceback runs on the host P).
int sw_align_auto_multi(sw_ctx* ctx, const int* devices, int ndev, const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores,
                        int32_t* H, int32_t* P, sw_result* result, int* executor, int64_t multi_min_cells) {
    if (!result || !H || !P || cols < 0 || rows < 0 || ndev < 0 || (ndev > 0 && !devices)) { set_err("sw_align_auto_multi: bad argument"); return SW_EINVAL; }
    const double cells = (double)cols * (double)rows;
    int ex = 0;
    if ((ctx || ndev > 0) && cells >= 2.0e5) {
        ex = 1;
        if (ndev >= 2 && rows >= 16 * (int64_t)ndev) {
            if (cells >= (multi_min_cells > 0 ? (double)multi_min_cells : 4.0e9)) ex = 2;
            else {
                size_t fr = 0, tot = 0;
                if (hipSetDevice(devices[0]) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess && (double)(cols + 1) * (double)(rows + 1) * 8.0 > 0.9 * (double)fr) ex = 2;
                else (void)hipGetLastError();
            }
        }
    }
    if (executor) *executor = ex;
    if (ex < 2) {
        sw_ctx* c = ex == 1 ? ctx : nullptr;
        sw_ctx* own = nullptr;
        if (ex == 1 && !c) {
            if (int rc = sw_create(devices[0], &own)) return rc;
            c = own;
        }
        const int rc = sw_align_auto(c, a, cols, b, rows, scores, H, P, result, nullptr);
        if (own) sw_destroy(own);
        return rc;
    }
    sw_multi* m = nullptr;
    if (int rc = sw_multi_create(devices, ndev, a, cols, b, rows, 4, 1, &m)) return rc;
    int rc = sw_multi_fill(m, scores, 0, result);
    const size_t M = (size_t)cols + 1;
    memset(H, 0, M * 4); memset(P, 0, M * 4);                       // row 0 (serial_smithW.c:96-103: calloc)
    for (size_t g = 0; g < m->bands.size() && rc == SW_OK; ++g) {   // band-local row 0 is the halo row: rows lo+1 .. hi are rows 1 .. of the band
        const sw_multi_band& bd = m->bands[g];
        const size_t n = (size_t)(bd.hi - bd.lo) * M * 4;
        if (hipSetDevice(bd.device) != hipSuccess || hipMemcpy(H + (size_t)(bd.lo + 1) * M, (const int32_t*)bd.d_H + M, n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(P + (size_t)(bd.lo + 1) * M, (const int32_t*)bd.d_P + M, n, hipMemcpyDeviceToHost) != hipSuccess) {
            set_err("sw_align_auto_multi: copying band %zu back failed", g);
            rc = SW_EDEVICE;
        }
    }
    sw_multi_free(m);
    if (rc != SW_OK) return rc;
    int64_t n = 0;
    rc = sw_traceback_host(P, cols, rows, result->max_pos, nullptr, 0, &n);
    result->path_len = n;
    return rc;
}

int sw_multi_nbands(sw_multi* m) { return m ? (int)m->bands.size() : 0; }
double sw_multi_seconds(sw_multi* m) { return m ? m->fill_seconds : 0.0; }

}  // extern "C"
