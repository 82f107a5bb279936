// sw_api.hip -- C-ABI entry points that drive the HIP kernels (see include/swhip.h).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include "sw_kernels.h"

namespace swh { void set_err(const char* fmt, ...); }
using swh::set_err;

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SW_EDEVICE;                                                        \
        }                                                                             \
    } while (0)

struct sw_ctx {
    int device = 0;
    int num_cus = 256;
    unsigned epoch = 0;                 // 12-bit launch tag, see FillParams::tag_base
    unsigned long long* d_edge = nullptr;
    size_t edge_cap = 0;                // granules
    unsigned long long* d_key = nullptr; // [0] = arg-max key, [1] low word = abort flag
    unsigned long long* d_keys = nullptr; size_t keys_cap = 0;  // batch: one key per pair
    unsigned char* d_cb = nullptr;      // systolic engine: zero-padded copy of b (sw_pad_b)
    size_t cb_cap = 0;
    int64_t opt_debug = 0;
    int64_t opt_store_policy = 0;       // systolic H/P stores: 0 auto (by size), 1 write-back, 2 streaming (nt)
    int64_t opt_xcd_order = 0;          // systolic: 1 = neighbouring strip groups on one XCD
    int64_t opt_importers = 2;          // systolic, one strip per workgroup: importer waves (as far as 12 waves allow)
    int64_t opt_pace_ps = 0;            // systolic: pacing of strip 0 (ps per row; 0 = off)
    int64_t opt_dbg_ptr = 0;
    int64_t opt_engine = 0;             // 0 = systolic producer/consumer pipeline, 1 = strip_scan (row scan)
    int64_t opt_strips_per_group = 0;   // systolic: producer waves (strips) per workgroup; 0 = by problem shape
    int64_t opt_consumers = 0;          // systolic: consumer waves per strip; 0 = by problem shape
    int64_t opt_waves_per_block = 4;
    int64_t opt_max_blocks = 0;         // 0 -> 2 * CUs
    int64_t last_grid = 0, last_strips = 0;
};

extern "C" {

int sw_create(int device, sw_ctx** out) {
    if (!out) { set_err("sw_create: out is NULL"); return SW_EINVAL; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err("sw_create: no HIP device"); return SW_ENODEV; }
    if (device < 0 || device >= n) { set_err("sw_create: device %d out of range (%d)", device, n); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(device));
    sw_ctx* c = new sw_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    HIP_TRY(hipMalloc((void**)&c->d_key, 64));
    HIP_TRY(hipMemset(c->d_key, 0, 64));
    *out = c;
    return SW_OK;
}

void sw_destroy(sw_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_edge) (void)hipFree(c->d_edge);
    if (c->d_key) (void)hipFree(c->d_key);
    if (c->d_cb) (void)hipFree(c->d_cb);
    if (c->d_keys) (void)hipFree(c->d_keys);
    delete c;
}

int sw_set_option(sw_ctx* c, const char* name, int64_t v) {
    if (!c || !name) { set_err("sw_set_option: bad argument"); return SW_EINVAL; }
    if (!strcmp(name, "waves_per_block")) {
        if (v != 0 && v != 1 && v != 2 && v != 4 && v != 8) { set_err("waves_per_block must be 1,2,4,8"); return SW_EINVAL; }
        c->opt_waves_per_block = v ? v : 4;
        return SW_OK;
    }
    if (!strcmp(name, "max_blocks")) { c->opt_max_blocks = v < 0 ? 0 : v; return SW_OK; }
    if (!strcmp(name, "strips_per_group")) { c->opt_strips_per_group = v; return SW_OK; }
    if (!strcmp(name, "consumers")) { c->opt_consumers = v; return SW_OK; }
    if (!strcmp(name, "debug_flags")) { c->opt_debug = v; return SW_OK; }
    if (!strcmp(name, "pace_ps")) { c->opt_pace_ps = v; return SW_OK; }
    if (!strcmp(name, "store_policy")) { if (v < 0 || v > 2) return SW_EINVAL; c->opt_store_policy = v; return SW_OK; }
    if (!strcmp(name, "xcd_order")) { c->opt_xcd_order = v ? 1 : 0; return SW_OK; }
    if (!strcmp(name, "importers")) { if (v < 0 || v > 8) return SW_EINVAL; c->opt_importers = v ? v : 2; return SW_OK; }
    if (!strcmp(name, "debug_buf")) { c->opt_dbg_ptr = v; return SW_OK; }
    if (!strcmp(name, "engine")) {
        if (v != 0 && v != 1) { set_err("engine must be 0 (systolic) or 1 (strip_scan)"); return SW_EINVAL; }
        c->opt_engine = v;
        return SW_OK;
    }
    set_err("sw_set_option: unknown option '%s'", name);
    return SW_EINVAL;
}

int64_t sw_get_option(sw_ctx* c, const char* name) {
    if (!c || !name) return -1;
    if (!strcmp(name, "waves_per_block")) return c->opt_waves_per_block;
    if (!strcmp(name, "max_blocks")) return c->opt_max_blocks;
    if (!strcmp(name, "engine")) return c->opt_engine;
    if (!strcmp(name, "strips_per_group")) return c->opt_strips_per_group;
    if (!strcmp(name, "consumers")) return c->opt_consumers;
    if (!strcmp(name, "store_policy")) return c->opt_store_policy;
    if (!strcmp(name, "xcd_order")) return c->opt_xcd_order;
    if (!strcmp(name, "importers")) return c->opt_importers;
    if (!strcmp(name, "pace_ps")) return c->opt_pace_ps;
    if (!strcmp(name, "num_cus")) return c->num_cus;
    if (!strcmp(name, "last_grid")) return c->last_grid;
    if (!strcmp(name, "last_strips")) return c->last_strips;
    return -1;
}

static int check_dims(int64_t cols, int64_t rows, const sw_scores* sc) {
    if (cols < 0 || rows < 0 || cols > swk::SW_MAX_DIM || rows > swk::SW_MAX_DIM) {
        set_err("dimensions out of range: cols=%lld rows=%lld (max %lld)", (long long)cols, (long long)rows,
                (long long)swk::SW_MAX_DIM);
        return SW_EINVAL;
    }
    if (sc->gap > 0) { set_err("gap score must be <= 0 (got %d)", sc->gap); return SW_EINVAL; }
    if (sc->match < 0) { set_err("match score must be >= 0 (got %d)", sc->match); return SW_EINVAL; }
    const int64_t lo = std::min(cols, rows);
    const int64_t gmax = (int64_t)sc->match * lo + (int64_t)(-sc->gap) * (rows + cols + 2);
    if (gmax >= (1ll << 31) || (int64_t)sc->match * lo >= (1ll << 24)) {
        set_err("scores too large for this problem size (32-bit cell / 24-bit arg-max key)");
        return SW_EINVAL;
    }
    return SW_OK;
}

// One launch of the fill: a whole matrix, a tile of a bigger matrix (row stride, halo row/column) or a
// batch of independent problems.
struct FillJob {
    const char* d_a; int64_t cols; const char* d_b; int64_t rows;
    void* d_H; int h_elem_bytes; int32_t* d_P; int64_t stride;
    const int32_t* d_top; const int32_t* d_left; int32_t* d_right;
    int64_t npairs; int64_t a_pstride, b_pstride, hp_pstride;
    unsigned long long* d_keys;   // npairs packed arg-max keys (device)
    int p_elem_bytes = 4;         // 4: int32 P (reference layout); 1: compact int8 P
};

static int launch_fill(sw_ctx* c, const sw_scores* sc, const FillJob& j, hipStream_t stream) {
    const int64_t cols = j.cols, rows = j.rows;
    const bool systolic = (c->opt_engine == 0);
    const bool tile_features = j.d_left || j.d_right || j.stride != cols + 1 || j.npairs != 1 || !j.d_H;
    if (!systolic && tile_features) { set_err("tiles / batches need the systolic engine (engine 0)"); return SW_EINVAL; }
    const int64_t S = systolic ? (cols + 62) / 63 : (cols + 63) / 64;
    if (((uintptr_t)j.d_b & 15) != 0 || (j.b_pstride & 15) != 0) { set_err("d_b (and the batch stride of b) must be 16-byte aligned"); return SW_EINVAL; }
    const size_t need = (size_t)S * (size_t)(rows + 1) * (size_t)j.npairs;
    if (need > c->edge_cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_edge) HIP_TRY(hipFree(c->d_edge));
        c->d_edge = nullptr; c->edge_cap = 0;
        if (hipMalloc((void**)&c->d_edge, need * 8) != hipSuccess) { set_err("workspace allocation of %zu bytes failed", need * 8); return SW_ENOMEM; }
        c->edge_cap = need;
        HIP_TRY(hipMemsetAsync(c->d_edge, 0, need * 8, stream));
        c->epoch = 0;
    }
    if (++c->epoch >= 4096) {  // 12-bit tag wrapped: stale tags could match again, wipe them
        HIP_TRY(hipMemsetAsync(c->d_edge, 0, c->edge_cap * 8, stream));
        c->epoch = 1;
    }
    swk::FillParams p;
    memset(&p, 0, sizeof p);
    p.cols = cols; p.rows = rows; p.M = j.stride;
    p.H = j.d_H; p.P = j.d_P; p.top = j.d_top; p.left = j.d_left; p.right = j.d_right;
    p.mm = sc->match - 2 * sc->gap; p.xm = sc->mismatch - 2 * sc->gap; p.ngap = -sc->gap;
    p.edge = c->d_edge; p.tag_base = c->epoch << 20;
    p.result_key = j.d_keys; p.abort_flag = (unsigned int*)(c->d_key + 1);
    p.nstrips = (int)S;
    p.debug_flags = (int)c->opt_debug;
    p.pace_ps = (int)c->opt_pace_ps;
    // streaming stores pay off while the matrices are small next to what is in flight; measured cross-over between
    // 16384^2 (nt 15-18 % faster) and 32768^2 (write-back 2-20 % faster)
    p.store_nt = c->opt_store_policy == 2 || (c->opt_store_policy == 0 && (double)cols * (double)rows * (double)j.npairs <= 6.0e8);
    p.xcd_order = (int)c->opt_xcd_order;
    p.dbg = (unsigned long long*)(uintptr_t)c->opt_dbg_ptr;
    p.npairs = (int)j.npairs; p.store_hp = j.d_H ? 1 : 0;
    p.p_bytes = j.p_elem_bytes;
    if (j.p_elem_bytes == 1 && (!systolic || j.npairs != 1)) { set_err("compact (int8) P needs the systolic engine and a single pair"); return SW_EINVAL; }
    p.a_pstride = j.a_pstride; p.b_pstride = j.b_pstride; p.hp_pstride = j.hp_pstride;
    p.edge_pstride = S * (rows + 1);
    const unsigned char* ua = (const unsigned char*)j.d_a;
    const unsigned char* ub = (const unsigned char*)j.d_b;
    if (systolic) {
        // Workgroup shape.  One strip + 8 consumers per workgroup gives every producer a SIMD of its own (measured on
        // single pairs from 4096^2 to 32768^2: equal to 5 % faster than 2 + 2x4, equal at 65536^2); batches that
        // do not fit the CUs at once run two strips per workgroup, twice the work per CU (1024^2 pairs: 415 vs 194 GCUPS
        // for 20000 pairs, 203 vs 143 for 64).
        int NS = (int)c->opt_strips_per_group, NC = (int)c->opt_consumers;
        if (NS == 0) NS = (j.npairs == 1 ? (double)S <= 4.5 * c->num_cus : (double)S * (double)j.npairs <= (double)c->num_cus) ? 1 : 2;
        if (NC == 0) NC = (NS == 1) ? 8 : 4;
        // padded copies of b per problem: [front | b | tail]; front covers the fast producers' phi (< strips) + 63 lanes
        const int64_t bfront = ((S + 64 + 127) / 128) * 128;
        const int64_t per = ((rows + bfront + 512 + 15) / 16) * 16;
        const size_t ncb = (size_t)per * (size_t)j.npairs;
        if (ncb > c->cb_cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (c->d_cb) HIP_TRY(hipFree(c->d_cb));
            c->d_cb = nullptr; c->cb_cap = 0;
            if (hipMalloc((void**)&c->d_cb, ncb * 3 + 16) != hipSuccess) { set_err("workspace allocation failed"); return SW_ENOMEM; }
            c->cb_cap = ncb;
        }
        unsigned short* d_cb16 = (unsigned short*)(c->d_cb + ((c->cb_cap + 15) / 16) * 16);
        hipLaunchKernelGGL(swk::sw_pad_b, dim3((unsigned)((per + 255) / 256), (unsigned)j.npairs), dim3(256), 0, stream, ub, rows, bfront,
                           j.b_pstride, c->d_cb, d_cb16, per);
        const bool fast = (j.d_top == nullptr) && (sc->mismatch <= 0) && !(c->opt_debug & 4);
        p.phi_base = fast ? (int)S - 1 : -1;
        p.bfront = (int)bfront;
        p.bpad16 = d_cb16;
        p.bpad_pstride = per;
        const int64_t ngroups = ((S + NS - 1) / NS) * j.npairs;
        const int64_t maxb = c->opt_max_blocks > 0 ? c->opt_max_blocks : (int64_t)c->num_cus;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ngroups, maxb));
        c->last_grid = grid; c->last_strips = S;
        const int base_waves = NS * (1 + NC) + 2;
        const int extra_imp = (NS == 1) ? (int)std::max<int64_t>(0, std::min<int64_t>(c->opt_importers - 1, 12 - base_waves)) : 0;
        const int threads = 64 * (base_waves + extra_imp);
        const unsigned char* cbp = c->d_cb;
        bool launched = false;
#define SW_LAUNCH(ns, nc)                                                                                                        \
    if (!launched && NS == ns && NC == nc) {                                                                                      \
        launched = true;                                                                                                          \
        if (j.h_elem_bytes == 4)                                                                                                  \
            hipLaunchKernelGGL((swk::sw_systolic<int32_t, ns, nc>), dim3(grid), dim3(threads), 0, stream, ua, ub, cbp, p);       \
        else                                                                                                                      \
            hipLaunchKernelGGL((swk::sw_systolic<int64_t, ns, nc>), dim3(grid), dim3(threads), 0, stream, ua, ub, cbp, p);       \
    }
        SW_LAUNCH(2, 2) SW_LAUNCH(2, 3) SW_LAUNCH(2, 4) SW_LAUNCH(1, 2) SW_LAUNCH(1, 3) SW_LAUNCH(1, 4) SW_LAUNCH(1, 6) SW_LAUNCH(1, 8)
#undef SW_LAUNCH
        if (!launched) { set_err("unsupported strips_per_group/consumers combination %d/%d", NS, NC); return SW_EINVAL; }
    } else {
        const int wpb = (int)c->opt_waves_per_block;
        const int64_t maxb = c->opt_max_blocks > 0 ? c->opt_max_blocks : 2ll * c->num_cus;
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((S + wpb - 1) / wpb, maxb));
        c->last_grid = grid; c->last_strips = S;
        if (j.h_elem_bytes == 4)
            hipLaunchKernelGGL((swk::sw_strip_scan<int32_t, 16>), dim3(grid), dim3(64 * wpb), 0, stream, ua, ub, p);
        else
            hipLaunchKernelGGL((swk::sw_strip_scan<int64_t, 16>), dim3(grid), dim3(64 * wpb), 0, stream, ua, ub, p);
    }
    HIP_TRY(hipGetLastError());
    return SW_OK;
}

static int fill_tile_impl(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                          void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes, int64_t row_stride, const int32_t* d_top,
                          const int32_t* d_left, int32_t* d_right, sw_result* d_result, void* stream_) {
    static const sw_scores kDefault = {3, -3, -2};  // serial_smithW.c:59-61
    const sw_scores* sc = scores ? scores : &kDefault;
    if (!c || !d_H || !d_P || !d_result || (h_elem_bytes != 4 && h_elem_bytes != 8) || (p_elem_bytes != 4 && p_elem_bytes != 1) ||
        row_stride < cols + 1) {
        set_err("sw_fill_tile_device: bad argument");
        return SW_EINVAL;
    }
    if (int rc = check_dims(cols, rows, sc)) return rc;
    if ((cols > 0 && !d_a) || (rows > 0 && !d_b)) { set_err("sw_fill_tile_device: NULL sequence"); return SW_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->d_key, 0, 16, stream));
    if (cols == 0 || rows == 0) {
        // no interior cell: H (= halo row / zero column) and P are all boundary
        if (row_stride != cols + 1 || d_left || d_right) { set_err("empty tiles are not supported"); return SW_EINVAL; }
        const int64_t M = cols + 1;
        HIP_TRY(hipMemsetAsync(d_H, 0, (size_t)(M * (rows + 1)) * h_elem_bytes, stream));
        HIP_TRY(hipMemsetAsync(d_P, 0, (size_t)(M * (rows + 1)) * p_elem_bytes, stream));
        if (d_top && h_elem_bytes == 4) HIP_TRY(hipMemcpyAsync(d_H, d_top, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
        if (d_top && h_elem_bytes == 8) { set_err("top halo with an empty int64 band is unsupported"); return SW_EINVAL; }
    } else {
        FillJob j = {d_a, cols, d_b, rows, d_H, h_elem_bytes, (int32_t*)d_P, row_stride, d_top, d_left, d_right, 1, 0, 0, 0, c->d_key};
        j.p_elem_bytes = p_elem_bytes;
        if (int rc = launch_fill(c, sc, j, stream)) return rc;
    }
    hipLaunchKernelGGL(swk::sw_finalize, dim3(1), dim3(64), 0, stream, c->d_key, (const unsigned int*)(c->d_key + 1), d_result, 1);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}

int sw_fill_tile_device(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                        void* d_H, int h_elem_bytes, int32_t* d_P, int64_t row_stride, const int32_t* d_top,
                        const int32_t* d_left, int32_t* d_right, sw_result* d_result, void* stream_) {
    return fill_tile_impl(c, d_a, cols, d_b, rows, scores, d_H, h_elem_bytes, d_P, 4, row_stride, d_top, d_left, d_right, d_result, stream_);
}

int sw_fill_device(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                   void* d_H, int h_elem_bytes, int32_t* d_P, const int32_t* d_top, sw_result* d_result, void* stream_) {
    return fill_tile_impl(c, d_a, cols, d_b, rows, scores, d_H, h_elem_bytes, d_P, 4, cols + 1, d_top, nullptr, nullptr, d_result, stream_);
}

// compact P: one byte per predecessor code (same values 0..3, -1..-3 after the traceback), SURVEY.md 8f-2
int sw_fill_device_ex(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                      void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes, const int32_t* d_top, sw_result* d_result,
                      void* stream_) {
    return fill_tile_impl(c, d_a, cols, d_b, rows, scores, d_H, h_elem_bytes, d_P, p_elem_bytes, cols + 1, d_top, nullptr, nullptr, d_result, stream_);
}

// BASELINE config 5: npairs independent cols x rows problems; pair k reads a at d_a + k*a_stride, b at d_b + k*b_stride.
// d_H / d_P may both be NULL (score-only: max_score exact, max_pos = first row of the 16-row block that holds it).
int sw_batch_device(sw_ctx* c, const char* d_a, int64_t a_stride, int64_t cols, const char* d_b, int64_t b_stride, int64_t rows,
                    int64_t npairs, const sw_scores* scores, int32_t* d_H, int32_t* d_P, sw_result* d_results, void* stream_) {
    static const sw_scores kDefault = {3, -3, -2};
    const sw_scores* sc = scores ? scores : &kDefault;
    if (!c || !d_a || !d_b || !d_results || npairs <= 0 || cols <= 0 || rows <= 0 || ((d_H == nullptr) != (d_P == nullptr)) ||
        a_stride < cols || b_stride < rows) {
        set_err("sw_batch_device: bad argument");
        return SW_EINVAL;
    }
    if (c->opt_engine != 0) { set_err("sw_batch_device needs the systolic engine"); return SW_EINVAL; }
    if (int rc = check_dims(cols, rows, sc)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t chunk_max = 4096;   // pairs per launch: bounds the edge / padded-b workspace
    if ((size_t)std::min(npairs, chunk_max) > c->keys_cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_keys) HIP_TRY(hipFree(c->d_keys));
        c->keys_cap = (size_t)std::min(npairs, chunk_max);
        if (hipMalloc((void**)&c->d_keys, c->keys_cap * 8) != hipSuccess) { c->d_keys = nullptr; c->keys_cap = 0; set_err("workspace allocation failed"); return SW_ENOMEM; }
    }
    HIP_TRY(hipMemsetAsync(c->d_key, 0, 16, stream));
    const int64_t cells = (cols + 1) * (rows + 1);
    for (int64_t k0 = 0; k0 < npairs; k0 += chunk_max) {
        const int64_t n = std::min(chunk_max, npairs - k0);
        HIP_TRY(hipMemsetAsync(c->d_keys, 0, (size_t)n * 8, stream));
        FillJob j = {d_a + k0 * a_stride, cols, d_b + k0 * b_stride, rows, d_H ? (void*)(d_H + k0 * cells) : nullptr, 4,
                     d_P ? d_P + k0 * cells : nullptr, cols + 1, nullptr, nullptr, nullptr, n, a_stride, b_stride, cells, c->d_keys};
        if (int rc = launch_fill(c, sc, j, stream)) return rc;
        hipLaunchKernelGGL(swk::sw_finalize, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, c->d_keys,
                           (const unsigned int*)(c->d_key + 1), d_results + k0, (int)n);
        HIP_TRY(hipGetLastError());
    }
    return SW_OK;
}

int sw_fill_host(sw_ctx* c, const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores,
                 int32_t* H, int32_t* P, sw_result* result) {
    if (!c || !result) { set_err("sw_fill_host: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t cells = (size_t)(cols + 1) * (size_t)(rows + 1);
    char *d_a = nullptr, *d_b = nullptr; int32_t *d_H = nullptr, *d_P = nullptr; sw_result* d_r = nullptr;
    int rc = SW_OK;
    auto cleanup = [&]() { (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_H); (void)hipFree(d_P); (void)hipFree(d_r); };
    if (hipMalloc((void**)&d_a, (size_t)cols + 16) != hipSuccess || hipMalloc((void**)&d_b, (size_t)rows + 16) != hipSuccess ||
        hipMalloc((void**)&d_H, cells * 4) != hipSuccess || hipMalloc((void**)&d_P, cells * 4) != hipSuccess ||
        hipMalloc((void**)&d_r, sizeof(sw_result)) != hipSuccess) {
        cleanup(); set_err("sw_fill_host: device allocation failed"); return SW_ENOMEM;
    }
    if (cols) (void)hipMemcpy(d_a, a, (size_t)cols, hipMemcpyHostToDevice);
    if (rows) (void)hipMemcpy(d_b, b, (size_t)rows, hipMemcpyHostToDevice);
    rc = sw_fill_device(c, d_a, cols, d_b, rows, scores, d_H, 4, d_P, nullptr, d_r, nullptr);
    if (rc == SW_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { set_err("fill kernel failed: %s", hipGetErrorString(e)); rc = SW_EDEVICE; }
    }
    if (rc == SW_OK) {
        (void)hipMemcpy(result, d_r, sizeof(sw_result), hipMemcpyDeviceToHost);
        if (result->path_len < 0) { set_err("fill kernel: hand-off wait timed out"); rc = SW_ETIMEOUT; }
        if (H) (void)hipMemcpy(H, d_H, cells * 4, hipMemcpyDeviceToHost);
        if (P) (void)hipMemcpy(P, d_P, cells * 4, hipMemcpyDeviceToHost);
    }
    cleanup();
    return rc;
}

int sw_traceback_device_ex(sw_ctx* c, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos, int64_t* d_path,
                           int64_t path_cap, sw_result* d_result, void* stream_) {
    if (!c || !d_P || !d_result || cols < 0 || rows < 0 || max_pos < 0 || max_pos >= (cols + 1) * (rows + 1) ||
        (p_elem_bytes != 4 && p_elem_bytes != 1)) {
        set_err("sw_traceback_device: bad argument");
        return SW_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    if (p_elem_bytes == 4)
        hipLaunchKernelGGL(swk::sw_traceback<int32_t>, dim3(1), dim3(64), 0, (hipStream_t)stream_, (int32_t*)d_P, cols + 1, max_pos, d_path,
                           d_path ? path_cap : 0, d_result);
    else
        hipLaunchKernelGGL(swk::sw_traceback<signed char>, dim3(1), dim3(64), 0, (hipStream_t)stream_, (signed char*)d_P, cols + 1, max_pos,
                           d_path, d_path ? path_cap : 0, d_result);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}
int sw_traceback_device(sw_ctx* c, int32_t* d_P, int64_t cols, int64_t rows, int64_t max_pos, int64_t* d_path,
                        int64_t path_cap, sw_result* d_result, void* stream_) {
    return sw_traceback_device_ex(c, d_P, 4, cols, rows, max_pos, d_path, path_cap, d_result, stream_);
}

int sw_device_malloc(sw_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) { set_err("sw_device_malloc: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) { set_err("sw_device_malloc: %zu bytes failed", bytes); return SW_ENOMEM; }
    return SW_OK;
}
int sw_device_free(sw_ctx* c, void* d_ptr) {
    if (!c) { set_err("sw_device_free: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return SW_OK;
}
int sw_memcpy_h2d(sw_ctx* c, void* d_dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !src))) { set_err("sw_memcpy_h2d: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return SW_OK;
}
int sw_memcpy_d2h(sw_ctx* c, void* dst, const void* d_src, size_t bytes) {
    if (!c || (bytes && (!dst || !d_src))) { set_err("sw_memcpy_d2h: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SW_OK;
}
int sw_synchronize(sw_ctx* c, void* stream_) {
    if (!c) { set_err("sw_synchronize: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    return SW_OK;
}

int sw_row_checksums_device(sw_ctx* c, const void* d_X, int elem_bytes, int64_t rows1, int64_t m, uint64_t* d_cs,
                            void* stream_) {
    if (!c || !d_X || !d_cs || rows1 <= 0 || m <= 0 || (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 1) || rows1 > 0x7fffffff) {
        set_err("sw_row_checksums_device: bad argument");
        return SW_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)stream_;
    if (elem_bytes == 4)
        hipLaunchKernelGGL((swk::sw_row_checksums<int32_t>), dim3((unsigned)rows1), dim3(256), 0, stream,
                           (const int32_t*)d_X, m, (unsigned long long*)d_cs);
    else if (elem_bytes == 1)
        hipLaunchKernelGGL((swk::sw_row_checksums<signed char>), dim3((unsigned)rows1), dim3(256), 0, stream,
                           (const signed char*)d_X, m, (unsigned long long*)d_cs);
    else
        hipLaunchKernelGGL((swk::sw_row_checksums<int64_t>), dim3((unsigned)rows1), dim3(256), 0, stream,
                           (const int64_t*)d_X, m, (unsigned long long*)d_cs);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}

}  // extern "C"
