// sw_api.hip -- C-ABI entry points that drive the HIP kernels (see include/swhip.h).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
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

// Fills of one device are serialised.  The systolic kernel's workgroups spin on hand-offs from other workgroups, so
// every workgroup of a launch must be resident; two fills in flight on one device (two contexts, or one context on
// two streams) could each hold part of the CUs and wait for the rest forever.  Launches therefore happen under a
// per-device lock, and a fill enqueued on a different stream than the previous one first waits (on the device, not
// the host) for everything enqueued on that previous stream.
struct DevState {
    std::mutex mu;
    bool any = false;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev = nullptr;
    int concurrent_ok = 0;   // set while band launches that partition the CUs explicitly are being enqueued
};
static DevState g_dev[64];

struct sw_ctx {
    int device = 0;
    int num_cus = 256;
    unsigned epoch = 0;                 // 12-bit launch tag, see FillParams::tag_base
    unsigned long long* d_edge = nullptr;
    size_t edge_cap = 0;                // granules
    unsigned long long* d_key = nullptr; // [0] = arg-max key, [1] low word = abort flag
    unsigned long long* d_keys = nullptr; size_t keys_cap = 0;  // batch: one key per pair
    unsigned char* d_cb = nullptr;      // systolic engine: padded copies of b (bytes, 16-bit, letter codes; sw_pad_b)
    size_t cb_cap = 0;
    unsigned int* d_edge4 = nullptr;    // perm producer: lane-63 columns as self-tagged 4-byte values
    size_t edge4_cap = 0;               // elements
    unsigned epoch8 = 0;                // 8-bit launch tag of those values
    unsigned char* d_alpha = nullptr;   // [64..323] letter code table + letter count; [512..1535] XCD of every workgroup of the running launch (sw_systolic2, xcd_mode)
    unsigned int* d_part = nullptr;     // sw_prep_scan: one 256-bit presence map of byte values per block (up to 2048 blocks)
    unsigned int* d_sync = nullptr;     // one-launch fills (sw_systolic2's prologue / epilogue): barrier and exit counters, presence map; zero between launches
    unsigned char* d_priv = nullptr; size_t priv_cap = 0;   // ... and every workgroup's own padded copy of b + letter codes
    int64_t opt_s2w = 0;                // two-column kernel: strips every 126 or 110 columns (overlapping strips, whole-line stores); 0: the library chooses
    int64_t opt_split_blk = 0, opt_split_from = 0;   // split strips: forced split block / first strip (0: chosen by the library; tests)
    int64_t opt_probe_foreign = 0;      // fills: probe an output pair the library did not allocate once, at its first fill (the probe writes and synchronises)
    int64_t opt_place_hold_gib = 0;     // sw_alloc_outputs: GiB a pair of small matrices may hold beside itself where no plain candidate is good (0: none)
    int64_t opt_place_budget_ms = 1500; // sw_alloc_outputs: time the search for a P in another class of the HBM may take
    int place_spacer_gib = 0;           // ... the spacer that led to one last time
    int64_t last_place_held_gib = 0;
    float last_place_ratio = 0.f;       // ... two-stream / one-stream time of the pair handed out last (~1.3-1.45: different classes, ~2: one class)
    bool key_dirty = false;             // d_key was left non-zero by a launch that does not re-arm it (everything but the one-launch fill)
    bool last_fused = false;            // the last launch_fill reports by itself (no sw_finalize behind it)
    int64_t opt_debug = 0;
    int64_t opt_xcd_chain = 0;          // two-column kernel without scouts: strips dealt per XCD (0 auto, 1 on, 2 off)
    int64_t opt_filler_hop_ps = 2400000, opt_filler_tau_ps = 25000, opt_filler_bw_gbs = 4200;   // pacing of the fillers behind scouts (sw_systolic2.inc)
    int64_t opt_store_policy = 0;       // systolic H/P stores: 0 auto (by size), 1 write-back, 2 streaming (nt)
    int64_t opt_xcd_order = 0;          // systolic: 1 = neighbouring strip groups on one XCD
    int64_t opt_importers = 0;          // systolic, one strip per workgroup: importer waves (as far as 12 waves allow); 0 = by problem size
    int64_t opt_pace_ps = 0;            // systolic: pacing of strip 0 (ps per row; 0 = off)
    int64_t opt_dbg_ptr = 0;
    int64_t opt_band_wait_ms = 20000;   // band-resident launch: patience of the top-halo poll
    int64_t opt_engine = 0;             // 0 = systolic producer/consumer pipeline, 1 = strip_scan (row scan)
    int64_t opt_strips_per_group = 0;   // systolic: producer waves (strips) per workgroup; 0 = by problem shape
    int64_t opt_consumers = 0;          // systolic: consumer waves per strip; 0 = by problem shape
    int64_t opt_waves_per_block = 4;
    int64_t opt_max_blocks = 0;         // 0 -> 2 * CUs
    unsigned char* d_bcodes = nullptr; size_t bcodes_cap = 0;   // batch kernel: padded letter codes of every pair's b
    int* d_bnd = nullptr; size_t bnd_cap = 0;                   // batch kernel: boundary columns between strips (ints)
    int64_t opt_batch_lds = 0;          // batch kernel: dynamic LDS bytes per workgroup (caps the waves per CU; experiments)
    int64_t last_batch_kernel = 0;      // 1: the last sw_batch_device call ran on sw_batch_wave (one pair per wave)
    int64_t last_grid = 0, last_strips = 0;
    int64_t last_strips2 = 0;           // strips of the two-column kernel in the last launch (0: not launched)
    int64_t last_scouts = 0;            // scout workgroups of that launch
    int64_t last_xcd_mode = 0;          // that launch dealt its roles per XCD
    int64_t last_tiles = 1;             // column tiles (launches of the two-column kernel) of the last fill
    int64_t last_split_from = 0;        // first strip whose scout also fills (split strips), 0: none
    bool xcd_round_robin = false;       // sw_xcc_probe saw workgroup i on XCD i % 8 (8 XCDs of 32 CUs)
    std::map<void*, void*> out_base;    // sw_alloc_outputs: pointer handed out -> allocation to free
    std::map<void*, float> pair_ratio;  // ... P handed out -> the store probe's ratio of its pair (~1.4: two classes of the HBM, ~2: one)
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
    HIP_TRY(hipMalloc((void**)&c->d_alpha, 2048));
    HIP_TRY(hipMemset(c->d_alpha, 0, 2048));
    HIP_TRY(hipMalloc((void**)&c->d_part, 2048 * 32));
    HIP_TRY(hipMalloc((void**)&c->d_sync, 256));
    HIP_TRY(hipMemset(c->d_sync, 0, 256));
    // Are the workgroups of a launch dealt round-robin to 8 XCDs of 32 CUs (workgroup i on XCD i % 8)?  The two-column kernel then
    // places every scout on the XCD of the workgroups that read its edge column (sw_systolic2.inc).
    if (c->num_cus == 256) {
        unsigned int h[256];
        hipLaunchKernelGGL(swk::sw_xcc_probe, dim3(256), dim3(768), 0, nullptr, c->d_part);
        if (hipMemcpy(h, c->d_part, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            c->xcd_round_robin = true;
            for (int i = 0; i < 256; ++i) c->xcd_round_robin = c->xcd_round_robin && h[i] == (unsigned)(i & 7);
        } else {
            (void)hipGetLastError();
        }
    }
    *out = c;
    return SW_OK;
}

void sw_destroy(sw_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_edge) (void)hipFree(c->d_edge);
    if (c->d_key) (void)hipFree(c->d_key);
    if (c->d_cb) (void)hipFree(c->d_cb);
    if (c->d_edge4) (void)hipFree(c->d_edge4);
    if (c->d_alpha) (void)hipFree(c->d_alpha);
    if (c->d_part) (void)hipFree(c->d_part);
    if (c->d_sync) (void)hipFree(c->d_sync);
    if (c->d_priv) (void)hipFree(c->d_priv);
    if (c->d_keys) (void)hipFree(c->d_keys);
    if (c->d_bcodes) (void)hipFree(c->d_bcodes);
    if (c->d_bnd) (void)hipFree(c->d_bnd);
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
    if (!strcmp(name, "xcd_chain")) { c->opt_xcd_chain = v; return SW_OK; }
    if (!strcmp(name, "filler_hop_ps")) { c->opt_filler_hop_ps = v < 0 ? 0 : v; return SW_OK; }
    if (!strcmp(name, "filler_tau_ps")) { c->opt_filler_tau_ps = v < 1000 ? 1000 : v; return SW_OK; }
    if (!strcmp(name, "filler_bw_gbs")) { c->opt_filler_bw_gbs = v < 100 ? 100 : v; return SW_OK; }
    if (!strcmp(name, "pace_ps")) { c->opt_pace_ps = v; return SW_OK; }
    if (!strcmp(name, "store_policy")) { if (v < 0 || v > 2) return SW_EINVAL; c->opt_store_policy = v; return SW_OK; }
    if (!strcmp(name, "xcd_order")) { c->opt_xcd_order = v ? 1 : 0; return SW_OK; }
    if (!strcmp(name, "importers")) { if (v < 0 || v > 8) return SW_EINVAL; c->opt_importers = v; return SW_OK; }
    if (!strcmp(name, "debug_buf")) { c->opt_dbg_ptr = v; return SW_OK; }
    if (!strcmp(name, "batch_lds")) { c->opt_batch_lds = v < 0 ? 0 : v; return SW_OK; }
    if (!strcmp(name, "band_wait_ms")) { c->opt_band_wait_ms = v > 0 ? v : 20000; return SW_OK; }
    if (!strcmp(name, "placement_budget_ms")) { c->opt_place_budget_ms = v > 0 ? v : 1500; return SW_OK; }
    if (!strcmp(name, "placement_hold_gib")) { c->opt_place_hold_gib = v < 0 ? 0 : (v > 128 ? 128 : v); return SW_OK; }
    if (!strcmp(name, "probe_foreign_pairs")) { c->opt_probe_foreign = v ? 1 : 0; return SW_OK; }
    if (!strcmp(name, "s2w")) { if (v != 0 && v != 126 && v != 110) return SW_EINVAL; c->opt_s2w = v; return SW_OK; }
    if (!strcmp(name, "split_blk")) { c->opt_split_blk = v > 0 ? v : 0; return SW_OK; }
    if (!strcmp(name, "split_from")) { c->opt_split_from = v > 0 ? v : 0; return SW_OK; }
    if (!strcmp(name, "debug_epoch8")) { c->epoch8 = (unsigned)(v & 255); return SW_OK; }   // development aid: next launch tag = v + 1
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
    if (!strcmp(name, "band_wait_ms")) return c->opt_band_wait_ms;
    if (!strcmp(name, "num_cus")) return c->num_cus;
    if (!strcmp(name, "debug_edge4_ptr")) return (int64_t)(uintptr_t)c->d_edge4;
    if (!strcmp(name, "debug_edge4_cap")) return (int64_t)c->edge4_cap;
    if (!strcmp(name, "last_grid")) return c->last_grid;
    if (!strcmp(name, "last_strips")) return c->last_strips;
    if (!strcmp(name, "last_strips2")) return c->last_strips2;
    if (!strcmp(name, "last_scouts")) return c->last_scouts;
    if (!strcmp(name, "last_xcd_mode")) return c->last_xcd_mode;
    if (!strcmp(name, "last_tiles")) return c->last_tiles;
    if (!strcmp(name, "last_split_from")) return c->last_split_from;
    if (!strcmp(name, "xcd_round_robin")) return c->xcd_round_robin ? 1 : 0;
    if (!strcmp(name, "last_batch_kernel")) return c->last_batch_kernel;
    if (!strcmp(name, "placement_budget_ms")) return c->opt_place_budget_ms;
    if (!strcmp(name, "placement_hold_gib")) return c->opt_place_hold_gib;
    if (!strcmp(name, "probe_foreign_pairs")) return c->opt_probe_foreign;
    if (!strcmp(name, "last_placement_held_gib")) return c->last_place_held_gib;
    if (!strcmp(name, "last_placement_ratio_x1000")) return (int64_t)(c->last_place_ratio * 1000.f);
    return -1;
}

// gr, gc: extent (rows, cols) of the WHOLE matrix the values may come from (== rows, cols unless this is a tile or a
// band whose halo carries scores accumulated outside it)
static int check_dims(int64_t cols, int64_t rows, const sw_scores* sc, int64_t gc = -1, int64_t gr = -1) {
    if (gc < cols) gc = cols;
    if (gr < rows) gr = rows;
    if (cols < 0 || rows < 0 || cols > swk::SW_MAX_DIM || rows > swk::SW_MAX_DIM || gc > swk::SW_MAX_DIM || gr > swk::SW_MAX_DIM) {
        set_err("dimensions out of range: cols=%lld rows=%lld (max %lld)", (long long)cols, (long long)rows,
                (long long)swk::SW_MAX_DIM);
        return SW_EINVAL;
    }
    if (sc->gap > 0) { set_err("gap score must be <= 0 (got %d)", sc->gap); return SW_EINVAL; }
    if (sc->match < 0) { set_err("match score must be >= 0 (got %d)", sc->match); return SW_EINVAL; }
    if (sc->mismatch > sc->match) { set_err("mismatch score must not exceed the match score"); return SW_EINVAL; }
    const int64_t lo = std::min(gc, gr);
    // largest G-space magnitude: H <= match*min(dims) plus -gap*(row+col); the per-step constants ride on top
    const int64_t gmax = (int64_t)sc->match * lo + (int64_t)(-sc->gap) * (rows + cols + 2);
    const int64_t step = std::max<int64_t>(std::llabs((int64_t)sc->mismatch), (int64_t)sc->match) + 2 * (int64_t)(-sc->gap);
    if (gmax + step >= (1ll << 31) || step >= (1ll << 24) || (int64_t)sc->match * lo >= (1ll << 24)) {
        set_err("scores too large for this problem size (32-bit cell / 24-bit arg-max key)");
        return SW_EINVAL;
    }
    return SW_OK;
}

// One launch of the fill: a whole matrix, a tile of a bigger matrix (row stride, halo row/column) or a
// batch of independent problems.
struct FillJob {
    const char* d_a; int64_t cols; const char* d_b; int64_t rows;
    void* d_H; int h_elem_bytes; void* d_P; int64_t stride;     // d_H / d_P may be NULL: that matrix is not written
    const int32_t* d_top; const int32_t* d_left; int32_t* d_right;
    int64_t npairs; int64_t a_pstride, b_pstride, hp_pstride;
    unsigned long long* d_keys;   // npairs packed arg-max keys (device)
    int p_elem_bytes = 4;         // 4: int32 P (reference layout); 1: compact int8 P
    // band-resident launch (sw_fill_band_device)
    const unsigned long long* d_top_gran = nullptr; unsigned long long* d_bot_gran = nullptr; unsigned int* d_bot_done = nullptr;
    unsigned int top_tag = 0, bot_tag = 0;
    int reserve_cus = 0;          // CUs left free for other kernels (halo transfers)
    bool concurrent = false;      // do not order this launch behind fills on other streams (the caller partitions the CUs)
    int64_t total_rows = 0;       // band: rows of the whole matrix (bounds the scores a halo can carry)
    bool reserve_only = false;    // size the per-context workspaces for this job and return: nothing is launched
    bool zero_key = false;        // the preparation kernel also zeroes d_keys[0..1] (fill_one leaves that to it)
    sw_result* d_result = nullptr;   // fill_one: where the result goes (a one-launch fill writes it by itself)
};

// called with g_dev[device].mu held: make `stream` wait for the fill enqueued last on another stream of this device.  The
// event is recorded on a fill's OWN stream when the fill has been enqueued (DevOrder's destructor), never on the previous
// stream later on: that stream may have been destroyed by then.
static int order_after_previous_fill(DevState& d, hipStream_t stream, bool allow_concurrent) {
    if (d.any && d.last_stream != stream && !allow_concurrent && d.ev) HIP_TRY(hipStreamWaitEvent(stream, d.ev, 0));
    return SW_OK;
}

struct DevOrder {   // RAII: device lock + stream ordering for one fill call
    std::unique_lock<std::mutex> lk;
    DevState& d;
    hipStream_t stream;
    int rc;
    DevOrder(sw_ctx* c, hipStream_t st, bool concurrent) : lk(g_dev[c->device & 63].mu), d(g_dev[c->device & 63]), stream(st) {
        rc = order_after_previous_fill(d, stream, concurrent);
    }
    ~DevOrder() {
        if (!d.ev && hipEventCreateWithFlags(&d.ev, hipEventDisableTiming) != hipSuccess) { d.ev = nullptr; (void)hipGetLastError(); }
        if (d.ev && hipEventRecord(d.ev, stream) == hipSuccess) { d.any = true; d.last_stream = stream; }
        else { (void)hipGetLastError(); d.any = false; }
    }
};

static int launch_fill(sw_ctx* c, const sw_scores* sc, const FillJob& j, hipStream_t stream) {
    const int64_t cols = j.cols, rows = j.rows;
    const bool systolic = (c->opt_engine == 0);
    c->last_fused = false;
    const bool tile_features = j.d_left || j.d_right || j.stride != cols + 1 || j.npairs != 1 || !j.d_H || !j.d_P || j.d_top_gran || j.d_bot_gran;
    if (!systolic && tile_features) { set_err("tiles / batches / bands / matrix-less fills need the systolic engine (engine 0)"); return SW_EINVAL; }
    // (the caller holds the device lock and has ordered `stream` behind earlier fills: DevOrder)
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
    p.H = j.d_H; p.P = (int32_t*)j.d_P; p.top = j.d_top; p.left = j.d_left; p.right = j.d_right;
    p.top_gran = j.d_top_gran; p.bot_gran = j.d_bot_gran; p.bot_done = j.d_bot_done; p.top_tag = j.top_tag; p.bot_tag = j.bot_tag;
    p.top_wait_ticks = (unsigned)std::min<int64_t>(0x7fffffff, c->opt_band_wait_ms * 100000 >> 10);
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
    p.npairs = (int)j.npairs; p.store_hp = (j.d_H || j.d_P) ? 1 : 0;
    p.p_bytes = j.p_elem_bytes;
    if (j.p_elem_bytes == 1 && !systolic) { set_err("compact (int8) P needs the systolic engine"); return SW_EINVAL; }
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
        // NS == 1: nine waves on the three SIMDs the producer leaves.  A chain-bound fill (one pair, up to ~3.5e8 cells: 16384^2)
        // wants the hand-off found early -- 4 consumers + 5 importer waves: 240 vs 232 GCUPS at 16384^2, +7 % at 8192^2; bigger
        // fills are bound by the stores and want 6 consumers + 3 importer waves (65536^2: 389 vs 340 GCUPS).
        const bool chain_bound = NS == 1 && j.npairs == 1 && (double)cols * (double)rows <= 3.5e8;
        if (NC == 0) NC = (NS == 1) ? (chain_bound ? 4 : 6) : 4;
        const int importers = c->opt_importers > 0 ? (int)c->opt_importers : (chain_bound && NC <= 4 ? 4 : 2);
        if (NS == 1 && NC > 7) NC = 7;         // nine waves off the producer's SIMD: at most 7 consumers + exporter + importer
        if (NS == 1 && NC == 5) NC = 4;        // (the one-column kernel has no five-consumer form: option "consumers" = 5 is for the two-column kernel)
        // padded copies of b per problem: [front | b | tail]; front covers the fast producers' phi (< strips) + 63 lanes
        // (+ one 16-step block: the perm producer's first score window ends at step 0)
        const int64_t bfront = ((S + 64 + 32 + 127) / 128) * 128;
        const int64_t per = ((rows + bfront + 512 + 15) / 16) * 16;
        const size_t ncb = (size_t)per * (size_t)j.npairs;
        if (ncb > c->cb_cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (c->d_cb) HIP_TRY(hipFree(c->d_cb));
            c->d_cb = nullptr; c->cb_cap = 0;
            if (hipMalloc((void**)&c->d_cb, ncb * 4 + 64) != hipSuccess) { set_err("workspace allocation failed"); return SW_ENOMEM; }
            c->cb_cap = ncb;
        }
        const size_t cb16 = ((c->cb_cap + 15) / 16) * 16;
        unsigned short* d_cb16 = (unsigned short*)(c->d_cb + cb16);
        unsigned char* d_cbc = c->d_cb + cb16 + ((2 * c->cb_cap + 15) / 16) * 16;
        // perm producer (alphabets of up to 7 letters): eligible when the scores fit a signed byte and every G value,
        // with the 2^16 bias, stays below 2^24 (the top byte carries the launch tag)
        const int64_t lo = std::min(cols, rows);
        const int64_t gmax = (int64_t)sc->match * std::max<int64_t>(lo, std::min(cols, j.total_rows)) + (int64_t)(-sc->gap) * (rows + cols + 2);
        const bool halo_unbounded = (j.d_top || j.d_left) && j.total_rows == 0;   // a tile whose halo magnitudes are unknown here
        const bool perm_ok = !halo_unbounded && p.mm <= 127 && p.mm >= -127 && p.xm <= 127 && p.xm >= -127 && gmax + 0x10000 + 1024 < (1ll << 24) &&
                             !(c->opt_debug & 16);
        if (perm_ok) {
            const int64_t e4stride = ((rows + S + 160 + 31) / 32) * 32;   // (whole 128-byte lines: strips written on different XCDs share none)
            const size_t need4 = (size_t)S * (size_t)e4stride * (size_t)j.npairs;
            if (need4 > c->edge4_cap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (c->d_edge4) HIP_TRY(hipFree(c->d_edge4));
                c->d_edge4 = nullptr; c->edge4_cap = 0;
                if (hipMalloc((void**)&c->d_edge4, need4 * 4) != hipSuccess) { set_err("workspace allocation of %zu bytes failed", need4 * 4); return SW_ENOMEM; }
                c->edge4_cap = need4;
                c->epoch8 = 255;   // fresh memory: wipe it below
            }
            if (++c->epoch8 >= (unsigned)((c->opt_debug & 1024) ? 4 : 256)) {   // 8-bit tag wrapped: wipe stale values (debug bit 10: wrap early)
                const unsigned nb = (unsigned)std::max<size_t>(1, std::min<size_t>((c->edge4_cap + 255) / 256, 2048));
                hipLaunchKernelGGL(swk::sw_wipe_u32, dim3(nb), dim3(256), 0, stream, c->d_edge4, c->edge4_cap);
                c->epoch8 = 1;
            }
            p.edge4 = c->d_edge4; p.e4stride = e4stride; p.edge4_pstride = (int64_t)S * e4stride;
            p.gbias = (c->epoch8 << 24) | 0x10000u;
        }
        // Two matrix columns per lane (sw_systolic2.inc): half as many strips -- and row segments of 504 bytes per store -- for the
        // same work: 16384^2 +4 %, 8192^2 +11 %, 24576^2 +29 %, 32768^2 +32 %, 65536^2 +9 % over one column per lane.  Whole
        // matrix of one pair, int32 H and P both stored, rows a multiple of 16; the alphabet (found on the device) must allow the
        // perm path -- so both kernels are enqueued and each checks for itself which of them has to work.  (debug bit 14: off)
        // Also: int8 P, either matrix left out, and band-resident launches (halo row in, last row out as granules).
        const bool base_mode = j.d_H && j.d_P && j.h_elem_bytes == 4 && j.p_elem_bytes == 4 && !j.d_top && !j.d_top_gran && !j.d_bot_gran;   // int32 H + P, whole matrix
        // Where it pays: always for int32 H + P (8-byte stores of both matrices); in the other output formats while the strip chain
        // (~3.1 us per 63-column strip) rather than the output volume (~3.2 TB/s) bounds the fill -- measured: 262144 x 32768 with
        // int8 P +18 %, 131072^2 with int8 P -3 %, 262144^2 P-only -21 % (two byte stores per row and the in-block arg-max).
        const double est_chain = (double)S * 3.1e-6;
        const double est_hbm = (double)(cols + 1) * (double)(rows + 1) * ((j.d_H ? (double)j.h_elem_bytes : 0.0) + (j.d_P ? (double)j.p_elem_bytes : 0.0)) / 3.2e12;
        // (... and int32 H + int8 P beyond the reach of the scouts: the H of overlapping strips goes out in streamed whole lines -- 65536^2 597 GCUPS
        //  against 429 on the one-column kernel and 494 with 126-column strips)
        const bool pays = (j.d_H && j.d_P && j.p_elem_bytes == 4) || (j.d_H && j.h_elem_bytes == 8) || est_chain >= (j.d_H ? 0.5 : 2.0) * est_hbm || (c->opt_debug & 32768) ||
                          (j.d_H && j.h_elem_bytes == 4 && (!j.d_P || j.p_elem_bytes == 1) && cols % 2 == 0 && cols > 126 * 170);
        int per_cu2 = 0;
        bool two_cols = pays && perm_ok && c->opt_strips_per_group == 0 && (c->opt_consumers == 0 || c->opt_consumers >= 4) && j.npairs == 1 &&
                              !j.d_left && !j.d_right && j.stride == cols + 1 && (rows % 16 == 0 || !j.d_bot_gran) && rows >= 1 && cols >= 1 &&   // (a band's last row leaves from a full block)
                             
                              (base_mode || (cols % 2 == 0)) &&   // (an odd column count leaves one lane with a single column: only the base mode handles it)
                              !(c->opt_debug & (2 | 8 | 64 | 128 | 512 | 16384));
        // The two-column kernel is ONE launch per fill: its prologue does what sw_prep_scan / sw_prep_code do for the other kernels (every
        // workgroup keeps its own padded copy of b and of its letter codes) and its last workgroup out writes the result (sw_systolic2.inc).
        // The strict one-launch path needs a result to write (a batch keeps its own keys and never comes here).
        const int64_t priv_stride = ((2 * per + 255) / 256) * 256;
        if (two_cols) {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, swk::sw_systolic2<6, false>, 768, 0));
            if (per_cu2 < 1 || !j.d_result) two_cols = false;
        }
        if (two_cols) {
            const size_t needp = (size_t)priv_stride * (size_t)per_cu2 * (size_t)c->num_cus;
            if (needp > c->priv_cap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (c->d_priv) HIP_TRY(hipFree(c->d_priv));
                c->d_priv = nullptr; c->priv_cap = 0;
                if (hipMalloc((void**)&c->d_priv, needp) != hipSuccess) { set_err("workspace allocation of %zu bytes failed", needp); return SW_ENOMEM; }
                c->priv_cap = needp;
            }
        }
        if (j.reserve_only) {   // every workspace of this job exists now (and is wiped where fresh): wait for that, launch nothing
            HIP_TRY(hipStreamSynchronize(stream));
            return SW_OK;
        }
        // input preparation, two dispatches (sw_systolic.hip): presence maps of the letters, then codes / padded copies of b -- and, for
        // the two-column kernel, row 0 and column 0 of the matrices and the arg-max key
        auto prepare = [&](void* zH, void* zP, bool skip_row0) {
            const int64_t total = (cols + rows) * j.npairs;
            const unsigned nscan = (unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 1023) / 1024, 2048));
            hipLaunchKernelGGL(swk::sw_prep_scan, dim3(nscan), dim3(256), 0, stream, ua, cols, j.a_pstride, ub, rows, j.b_pstride, j.npairs, c->d_part);
            const unsigned npad = (unsigned)((per + 255) / 256);
            const unsigned nzero = (zH || zP) ? (unsigned)((cols + 1 + rows + 255) / 256) : 0u;
            hipLaunchKernelGGL(swk::sw_prep_code, dim3(npad + nzero, (unsigned)j.npairs), dim3(256), 0, stream, ub, rows, bfront, j.b_pstride, c->d_cb, d_cb16,
                               d_cbc, (const unsigned int*)c->d_part, (int)nscan, c->d_alpha + 64, per, (int)npad, zH, j.h_elem_bytes, zP, j.p_elem_bytes,
                               cols + 1, rows + 1, skip_row0 ? 1 : 0, j.zero_key ? j.d_keys : nullptr);
        };
        p.bcode = d_cbc;
        p.atab = c->d_alpha + 64;
        const bool fast = (j.d_top == nullptr) && (j.d_top_gran == nullptr) && (sc->mismatch <= 0) && !(c->opt_debug & 4);
        p.phi_base = fast ? (int)S - 1 : -1;
        p.bfront = (int)bfront;
        p.bpad16 = d_cb16;
        p.bpad8 = c->d_cb;
        p.bpad_pstride = per;
        const int64_t ngroups = ((S + NS - 1) / NS) * j.npairs;
        const int64_t maxb = c->opt_max_blocks > 0 ? c->opt_max_blocks : std::max<int64_t>(8, (int64_t)c->num_cus - j.reserve_cus);
        int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ngroups, maxb));
        int threads;
        if (NS == 1) {
            // wave 0 (the producer) owns SIMD 0: waves 4, 8, 12 idle; consumers, the exporter and the importers are the
            // K waves off SIMD 0 (wave id of ordinal k: k + 1 + k/3)
            const int K = std::min<int>(9, NC + 1 + std::max(1, importers));   // 12 waves: 3 per SIMD
            threads = 64 * (K + (K - 1) / 3 + 1);
        } else {
            threads = 64 * (NS * (1 + NC) + 2);
        }
        const unsigned char* cbp = c->d_cb;
        bool launched = false;
        c->last_strips2 = 0; c->last_scouts = 0; c->last_xcd_mode = 0; c->last_tiles = 1;
        if (two_cols) {
            // strip geometry: every 126 columns (the strips tile the matrix), or every 110 with 16 columns of overlap -- whole-line stores
            // (sw_systolic2.inc); option "s2w" forces one of the two (tests, A/B runs)
            const bool band_io = j.d_top || j.d_top_gran || j.d_bot_gran;
            int64_t W2 = 126;
            // Overlapping strips store whole 64-byte lines, and whole lines can be STREAMED (nt): together that is worth 1.2 - 1.5x wherever the strips
            // do not leave room for scouts (GCUPS, strips every 126 write-back / every 110 streaming, same buffers: int32 H 21760^2 278 (tiles) / 363,
            // 24576^2 314 (tiles) / 409, 32768^2 382 (tiles) / 481, 40000^2 385 / 540, 49152^2 403 / 595, 65536^2 436 / 624, 81920^2 474 / 559; int64 H
            // 20480^2 264 (scouts) / 330, 24576^2 238 / 369, 32768^2 265 / 409, 49152^2 282 / 403, 65536^2 382 / 451).  Streaming PARTIAL lines is what
            // round 2 measured as harmful; write-back whole lines are what the first version of the overlap did (+14 % for int64 H only).  Behind
            // scouts (up to 170 strips) the 126-column geometry stays: there the chain bounds the fill and 14 % more strips cost more than the
            // stores gain (int32 16384^2: 335 / 322) -- except for an int64 H whose 110-column strips no longer fit beside scouts (18 700 - 21 400
            // columns), which is faster as a plain chain of overlapping strips than behind scouts.
            const bool wl_fmt = !band_io && j.d_H && (!j.d_P || j.p_elem_bytes == 4 || (j.h_elem_bytes == 4 && cols % 2 == 0)) && (j.d_P || cols % 2 == 0) && j.stride == cols + 1 &&
                                ((uintptr_t)j.d_H & (j.h_elem_bytes == 8 ? 15u : 7u)) == 0 && ((uintptr_t)j.d_P & 7u) == 0;
            const int64_t S126 = cols <= 126 ? 1 : (cols - 126 + 125) / 126 + 1, S110 = cols <= 126 ? 1 : (cols - 126 + 109) / 110 + 1;
            // A pair that lies in ONE class of the HBM (the allocator's probe said so: its search ran out of budget, or the caller asked for a
            // plain pair) is slowed much less when its lines are streamed whole: 16384^2 307 against 232 GCUPS with 126-column strips -- 7 % behind a
            // pair in two classes instead of 30 %.  (Pairs the library did not allocate are not probed -- the probe writes -- and keep 126.)
            bool one_class = false;
            if (j.d_P && (double)cols * (double)rows >= 2.0e8) {
                auto it = c->pair_ratio.find(j.d_P);
                if (it == c->pair_ratio.end() && c->opt_probe_foreign && c->opt_s2w == 0 && wl_fmt && S126 <= 170 && !j.reserve_only) {
                    // option "probe_foreign_pairs": a pair the library did not allocate is probed once, at its first fill -- the probe WRITES both
                    // buffers (this fill overwrites them anyway) and synchronises the stream (~0.3 ms); remembered by the address of P (at most 64)
                    float r = 0.f, ms = 0.f;
                    if (c->pair_ratio.size() >= 64) c->pair_ratio.clear();
                    if (hipStreamSynchronize(stream) == hipSuccess &&
                        sw_place_pair_ratio(j.d_H, (size_t)(cols + 1) * (size_t)(rows + 1) * (size_t)j.h_elem_bytes, j.d_P, (size_t)(cols + 1) * (size_t)(rows + 1) * 4, &r, &ms) == SW_OK)
                        it = c->pair_ratio.emplace(j.d_P, r).first;
                }
                one_class = it != c->pair_ratio.end() && it->second >= 1.7f;
            }
            if (c->opt_s2w == 110 ? (!band_io && (cols % 2 == 0 || wl_fmt)) : (c->opt_s2w == 0 && wl_fmt && (S126 > 170 || (j.h_elem_bytes == 8 && S110 > 170) || one_class))) W2 = 110;
            const bool ov_auto = W2 == 110 && c->opt_s2w == 0;   // (the library's own choice: one launch, streaming stores)
            auto strips_of = [&](int64_t ncols) { return ncols <= 126 ? (int64_t)1 : (ncols - 126 + W2 - 1) / W2 + 1; };
            const int64_t S2all = strips_of(cols);
            // Column tiles.  Scout workgroups beside one filler per strip (sw_systolic2.inc) need 1.5 .. 2 workgroups per strip: up to ~170
            // strips (21 000 columns) on 256 CUs.  A wider matrix used to run the classic chain, fillers handing over to fillers at 7 us per
            // strip (32768^2: 349 GCUPS).  Now it is cut into column tiles of at most 160 strips, ONE LAUNCH EACH, every one with scouts,
            // roles per XCD and paced fillers; a tile's left halo is the previous tile's last column, read from H itself (kernel boundary:
            // no flags), the arg-max accumulates in the key across the launches and the last one reports.  Taken where the estimate says
            // it pays: not where the stores bound the fill anyway (int64 H at 65536^2), not for bands (their halo row arrives while they
            // run).  (debug bit 19: off)
            int64_t ntile = 1, tstrips = S2all;
            if (S2all > 170 && base_mode && j.stride == cols + 1 && j.d_result && !ov_auto && !(c->opt_debug & 524288)) {
                const int64_t nt = (S2all + 159) / 160, st = (S2all + nt - 1) / nt;
                // measured (one box, GCUPS tiled / untiled): 24576^2 314 / 277, 32768^2 376 / 330, 40000^2 328 / 394, 49152^2 308 / 406 -- a tile
                // ramps up and drains its chain with the stores idle (3.0 TB/s on average where the untiled fill of a big matrix keeps 3.3),
                // so tiles pay while the untiled fill is bound by its 7 us hand-offs, up to ~36 000 columns
                const double bytes = (double)(cols + 1) * (double)(rows + 1) * 8.0;
                const double t_tiles = (double)nt * std::max((double)st * 2.4e-6 + (double)(rows + 200) * 26e-9, bytes / (double)nt / 3.0e12);
                const double t_classic = std::max((double)S2all * 7.6e-6 + (double)rows * 35e-9, bytes / 3.3e12);
                if (t_tiles < 0.98 * t_classic) { ntile = nt; tstrips = st; }
            }
            c->last_tiles = ntile;
            for (int64_t tile = 0; tile < ntile; ++tile) {
            const int64_t c0 = tile * tstrips * W2, tcols = tile + 1 == ntile ? cols - c0 : tstrips * W2;   // (a tile owns tstrips * W2 columns; the last one the rest)
            const int64_t S2 = strips_of(tcols);
            swk::FillParams p2 = p;
            p2.nstrips = (int)S2;
            p2.h_bytes = j.h_elem_bytes;
            p2.cols = tcols;
            p2.s2w = (int)W2;
            if (ntile > 1) p2.store_nt = c->opt_store_policy == 2 || (c->opt_store_policy == 0 && (double)tcols * (double)rows <= 6.0e8);   // (per launch, as for a matrix of the tile's size)
            if (W2 == 110 && wl_fmt && c->opt_store_policy == 0) p2.store_nt = 1;   // (whole lines: streamed)
            p2.alpha_a = ua; p2.alpha_cols = cols;
            p2.idx_off = c0; p2.final_launch = tile + 1 == ntile ? 1 : 0;
            if (ntile > 1) {
                p2.H = (char*)j.d_H + c0 * 4; p2.P = (int32_t*)((char*)j.d_P + c0 * 4);
                p2.tile_left = tile ? (const int32_t*)j.d_H + c0 : nullptr;
                if (tile) {   // the edge values of this context are self-tagged with the launch tag: every tile launch has its own
                    if (++c->epoch8 >= (unsigned)((c->opt_debug & 1024) ? 4 : 256)) {
                        const unsigned nb = (unsigned)std::max<size_t>(1, std::min<size_t>((c->edge4_cap + 255) / 256, 2048));
                        hipLaunchKernelGGL(swk::sw_wipe_u32, dim3(nb), dim3(256), 0, stream, c->d_edge4, c->edge4_cap);
                        c->epoch8 = 1;
                    }
                    p2.gbias = (c->epoch8 << 24) | 0x10000u;
                }
            }
            const unsigned char* ua_t = ua + c0;
            const int per_cu = per_cu2;
            {
                int grid2 = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(S2, maxb), (int64_t)per_cu * c->num_cus));
                // Scouts (sw_systolic2.inc): while every strip has a workgroup of its own and half as many more fit the device, the chain
                // of strips runs in extra workgroups that keep nothing but the edge columns, and the workgroups that write the matrices
                // follow them instead of each other.  (debug bit 17: off)
                // One scout strip per workgroup where the device has the workgroups for it (S2 fillers + S2 scouts), two in as many
                // scout workgroups as it takes to fit; at least 1.5 S2 workgroups in all.
                const int64_t avail = std::min<int64_t>(maxb, (int64_t)per_cu * c->num_cus);
                const int64_t ndouble = std::max<int64_t>(0, 2 * S2 - 1 - avail);   // (the last strip needs no scout: nobody reads its edge)
                const int64_t nsc = S2 - 1 - ndouble;
                const bool scouts = S2 >= 4 && 2 * ndouble <= S2 - 1 && nsc >= 1 && !(c->opt_debug & 131072);
                p2.nscout = scouts ? (int)nsc : 0;
                p2.scout_double = scouts ? (int)ndouble : 0;
                if (scouts) grid2 = (int)(S2 + nsc);
                // Roles dealt per XCD (sw_systolic2.inc): an eighth of the fillers and the scouts that feed them on every XCD, so that
                // an edge column is read on the XCD that wrote it -- out of its L2, without the trip through the memory fabric.  Needs the
                // whole device (256 workgroups, one per CU, workgroup i on XCD i % 8).  (debug bit 23: off)
                p2.xcd_mode = 0;
                // measured (same buffers, classic against dealt per XCD): 24576^2 -1 %, 32768^2 -0.4 %, 65536^2 int32 +0.3 %, int64 H +2.3 %,
                // 262144 x 32768 with int8 P +4.4 %: the classic hand-off is not the trip through memory (the polls of a filler queue behind
                // its own H / P stores in the CU), so the gain is small and only where a workgroup runs several strips
                // (with overlapping strips and streaming stores the dealing per XCD costs instead: 65536^2 int64 H 9.52 against 8.59 ms, int32 7.00 / 6.89, 24576^2 1.60 / 1.48)
                const bool xcd_chain_pays = S2 >= 384 && W2 != 110;
                if (scouts && c->xcd_round_robin && avail >= 256 && S2 >= 16 && !(c->opt_debug & 8388608)) {
                    bool fits = true;
                    int wg = 0, dbl = 0;
                    for (int x = 0; x < 8 && fits; ++x) {
                        const int nf = (int)(S2 / 8) + (x < (int)(S2 % 8) ? 1 : 0), ns = x ? nf : nf - 1;
                        const int ndx = std::max(0, ns - (32 - nf));
                        fits = nf <= 31 && 2 * ndx <= ns;
                        wg += ns - ndx; dbl += ndx;
                    }
                    if (fits) { p2.xcd_mode = 1; p2.nscout = wg; p2.scout_double = dbl; grid2 = 256; }
                }
                // Split strips (sw_systolic2.inc): from strip split_from on -- those that would end after everybody else -- the strip's scout
                // (a workgroup with rings and consumers then) writes the blocks from split_blk on and the filler only those before; the
                // last strip gets a scout for it (one more workgroup on XCD 7).  The split point equalises the two ends: the filler needs
                // tau_f per row, the scout ~21.5 ns before its consumers start and tau_f after.  Needs the per-XCD dealing and pacing (the
                // common end every filler is paced to moves with it).  (debug bit 20: off)
                p2.split_blk = 0; p2.split_from = 0; p2.split_extra = 0; p2.filler_end_steps = 0; p2.filler_full_steps = 0;
                if (p2.xcd_mode == 1 && (rows >= 4096 || c->opt_split_blk > 0) && !j.d_top && !j.d_top_gran && !j.d_bot_gran && c->opt_filler_hop_ps > 0 &&
                    !(c->opt_debug & (134217728 | 1048576))) {
                    bool fits = true;
                    int wg = 0, dbl = 0;
                    for (int x = 0; x < 8 && fits; ++x) {
                        const int nf = (int)(S2 / 8) + (x < (int)(S2 % 8) ? 1 : 0), ns = (x ? nf : nf - 1) + (x == 7 ? 1 : 0);
                        const int ndx = std::max(0, ns - (32 - nf));
                        fits = nf <= 31 && 2 * ndx <= ns;
                        wg += ns - ndx; dbl += ndx;
                    }
                    const double hop = (double)c->opt_filler_hop_ps * 1e-12, tf = (double)c->opt_filler_tau_ps * 1e-12, ts = 21.5e-9;
                    const int64_t nblk = (rows + 15) / 16;
                    const int64_t sblk = (int64_t)((double)nblk * tf / (2.0 * tf - std::min(ts, tf)));
                    const bool forced = c->opt_split_blk > 0;   // (tests: any split point, any first strip)
                    if (fits && forced && c->opt_split_blk < nblk) {
                        p2.split_blk = (int)c->opt_split_blk; p2.split_from = (int)std::max<int64_t>(1, c->opt_split_from); p2.split_extra = 1;
                        p2.filler_end_steps = (int)((c->opt_split_blk * 16 + 126) / 64 * 64 + 64); p2.filler_full_steps = (int)((nblk * 16 + 126) / 64 * 64 + 64);
                        p2.nscout = wg; p2.scout_double = dbl;
                    } else if (fits && sblk >= 16 && sblk < nblk && rows >= 4096) {
                        const int64_t full_steps = (nblk * 16 + 126) / 64 * 64 + 64, end_steps = (sblk * 16 + 126) / 64 * 64 + 64;
                        // strip s, unsplit, would end s hand-offs + a whole strip after the start; the split ones end (S2 - 1) hand-offs + end_steps
                        const double lead = ((double)full_steps - (double)end_steps) * tf / hop;
                        const int64_t from = std::max<int64_t>(1, (int64_t)((double)(S2 - 1) - lead) + 1);
                        if (from < S2) {
                            p2.split_blk = (int)sblk; p2.split_from = (int)from; p2.split_extra = 1;
                            p2.filler_end_steps = (int)end_steps; p2.filler_full_steps = (int)full_steps;
                            p2.nscout = wg; p2.scout_double = dbl;
                        }
                    }
                }
                c->last_split_from = p2.split_blk ? p2.split_from : 0;
                // The classic chain (no room for scouts), dealt per XCD the same way: neighbouring strips on one XCD, edge columns through
                // its L2.  (option "xcd_chain": 0 auto, 1 on, 2 off)
                if (!scouts && c->xcd_round_robin && grid2 >= 64 && !(c->opt_debug & 8388608) &&
                    (c->opt_xcd_chain == 1 || (c->opt_xcd_chain == 0 && xcd_chain_pays)))
                    p2.xcd_mode = 2;
                // pacing of the fillers behind scouts (sw_systolic2.inc): estimates on the low side, so that nobody is held back more
                // than the last filler's best case allows.  (options "filler_hop_ps" / "filler_tau_ps"; debug bit 27: off)
                p2.filler_hop_ps = (scouts && !(c->opt_debug & 134217728)) ? (int)c->opt_filler_hop_ps : 0;
                p2.filler_tau_ps = (int)c->opt_filler_tau_ps;
                p2.filler_bw_gbs = (int)c->opt_filler_bw_gbs;
                c->last_xcd_mode = p2.xcd_mode;
                c->last_scouts = p2.nscout;
                // one launch per fill: the kernel's prologue prepares (letter codes, every workgroup's padded copy of b, zeros in row 0 /
                // column 0 -- except a band's halo row: its H comes from the row above, written by the kernel; its P belongs to the band
                // above), its last workgroup out reports and re-arms key / abort flag / sync words -- which therefore are zero here, unless
                // another kind of launch has used the key since
                if (c->key_dirty) { HIP_TRY(hipMemsetAsync(c->d_key, 0, 16, stream)); c->key_dirty = false; }
                p2.sync = c->d_sync; p2.priv = c->d_priv; p2.priv_stride = priv_stride;
                p2.bpad16_w = d_cb16; p2.bpad8_w = c->d_cb; p2.bcode_w = d_cbc; p2.atab_w = c->d_alpha + 64;
                p2.result = j.d_result; p2.skip_row0 = (j.d_top || j.d_top_gran) ? 1 : 0;
                // consumer waves (+ 9 - nc2 importers).  Behind scouts a filler is never the one a hand-off waits for: two importers do, and seven
                // consumers keep more stores in flight (16384^2 -1.5 %, 12288^2 -2.5 %, 20480^2 +-0 against five)
                // (overlapping strips, whose consumers are dearer: seven as well -- 32768^2 485 against 453-466 GCUPS, 65536^2 633 / 607, int64 H 502 / 495)
                const int nc2 = c->opt_consumers == 0 ? ((scouts || W2 == 110) ? 7 : (chain_bound ? 5 : 6)) : (int)std::min<int64_t>(7, c->opt_consumers);
                auto launch2 = [&](auto nc, auto ov) { hipLaunchKernelGGL((swk::sw_systolic2<decltype(nc)::value, decltype(ov)::value>), dim3(grid2), dim3(768), 0, stream, ua_t, ub, p2); };
                auto launch_nc = [&](auto ov) {
                    if (nc2 == 7) launch2(std::integral_constant<int, 7>{}, ov);
                    else if (nc2 == 6) launch2(std::integral_constant<int, 6>{}, ov);
                    else if (nc2 == 5) launch2(std::integral_constant<int, 5>{}, ov);
                    else launch2(std::integral_constant<int, 4>{}, ov);
                };
                if (W2 == 110) launch_nc(std::true_type{}); else launch_nc(std::false_type{});   // (the strip geometry is compiled in)
                c->last_strips2 = S2;
            }
            }   // (tiles)
            // the fall-back (an alphabet of more than 7 letters, known on the device only): enqueued behind, leaves at once otherwise;
            // it fills the whole matrix by itself, whatever the tiling
            p.skip_if_perm = 1;
            p.sync = c->d_sync; p.atab_w = c->d_alpha + 64; p.result = j.d_result; p.final_launch = 1;
            c->last_fused = true;
        }
        if (!two_cols) { prepare(nullptr, nullptr, false); c->key_dirty = true; }
#define SW_LAUNCH(ns, nc)                                                                                                        \
    if (!launched && NS == ns && NC == nc) {                                                                                      \
        launched = true;                                                                                                          \
        int per_cu = 0;                                                                                                           \
        if (j.h_elem_bytes == 4) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, swk::sw_systolic<int32_t, ns, nc>, threads, 0)); \
        else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, swk::sw_systolic<int64_t, ns, nc>, threads, 0));     \
        if (per_cu < 1) { set_err("the fill kernel does not fit a CU on this device"); return SW_EDEVICE; }                      \
        grid = (int)std::min<int64_t>(grid, (int64_t)per_cu * c->num_cus);                                                        \
        c->last_grid = grid; c->last_strips = S;                                                                                  \
        if (j.h_elem_bytes == 4)                                                                                                  \
            hipLaunchKernelGGL((swk::sw_systolic<int32_t, ns, nc>), dim3(grid), dim3(threads), 0, stream, ua, ub, cbp, p);       \
        else                                                                                                                      \
            hipLaunchKernelGGL((swk::sw_systolic<int64_t, ns, nc>), dim3(grid), dim3(threads), 0, stream, ua, ub, cbp, p);       \
    }
        SW_LAUNCH(2, 2) SW_LAUNCH(2, 3) SW_LAUNCH(2, 4) SW_LAUNCH(1, 2) SW_LAUNCH(1, 3) SW_LAUNCH(1, 4) SW_LAUNCH(1, 6) SW_LAUNCH(1, 7)
#undef SW_LAUNCH
        if (!launched) { set_err("unsupported strips_per_group/consumers combination %d/%d", NS, NC); return SW_EINVAL; }
    } else {
        c->key_dirty = true;
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

static const sw_scores kDefaultScores = {3, -3, -2};  // serial_smithW.c:59-61

// one matrix / tile / band: validation, empty shapes, launch, finalize
static int fill_one(sw_ctx* c, const sw_scores* scores, FillJob j, int64_t gcols, int64_t grows, sw_result* d_result, void* stream_,
                    const char* who) {
    const sw_scores* sc = scores ? scores : &kDefaultScores;
    const int64_t cols = j.cols, rows = j.rows;
    if (!c || !d_result || (j.h_elem_bytes != 4 && j.h_elem_bytes != 8) || (j.p_elem_bytes != 4 && j.p_elem_bytes != 1) ||
        j.stride < cols + 1) {
        set_err("%s: bad argument", who);
        return SW_EINVAL;
    }
    if (int rc = check_dims(cols, rows, sc, gcols, grows)) return rc;
    if ((cols > 0 && !j.d_a) || (rows > 0 && !j.d_b)) { set_err("%s: NULL sequence", who); return SW_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    DevOrder order(c, stream, j.concurrent);
    if (order.rc) return order.rc;
    j.d_keys = c->d_key;
    j.d_result = d_result;
    j.zero_key = c->opt_engine == 0 && cols > 0 && rows > 0;   // (the systolic engine's preparation zeroes the key and the abort flag -- or finds them zero)
    if (!j.zero_key) { HIP_TRY(hipMemsetAsync(c->d_key, 0, 16, stream)); c->key_dirty = true; }
    c->last_fused = false;
    if (cols == 0 || rows == 0) {
        // no interior cell: H (= halo row / zero column) and P are all boundary
        if (j.stride != cols + 1 || j.d_left || j.d_right || j.d_top_gran || j.d_bot_gran) { set_err("%s: empty tiles / bands are not supported", who); return SW_EINVAL; }
        const int64_t M = cols + 1;
        if (j.d_H) HIP_TRY(hipMemsetAsync(j.d_H, 0, (size_t)(M * (rows + 1)) * j.h_elem_bytes, stream));
        if (j.d_P) HIP_TRY(hipMemsetAsync(j.d_P, 0, (size_t)(M * (rows + 1)) * j.p_elem_bytes, stream));
        if (j.d_H && j.d_top && j.h_elem_bytes == 4) HIP_TRY(hipMemcpyAsync(j.d_H, j.d_top, (size_t)M * 4, hipMemcpyDeviceToDevice, stream));
        if (j.d_H && j.d_top && j.h_elem_bytes == 8) { set_err("top halo with an empty int64 band is unsupported"); return SW_EINVAL; }
    } else {
        if (int rc = launch_fill(c, sc, j, stream)) return rc;
    }
    // (a one-launch fill has written the result by itself: sw_systolic2.inc)
    if (!c->last_fused) hipLaunchKernelGGL(swk::sw_finalize, dim3(1), dim3(64), 0, stream, c->d_key, (const unsigned int*)(c->d_key + 1), d_result, 1);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}

static FillJob make_job(const char* d_a, int64_t cols, const char* d_b, int64_t rows, void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes,
                        int64_t row_stride, const int32_t* d_top, const int32_t* d_left, int32_t* d_right) {
    FillJob j = {d_a, cols, d_b, rows, d_H, h_elem_bytes, d_P, row_stride, d_top, d_left, d_right, 1, 0, 0, 0, nullptr};
    j.p_elem_bytes = p_elem_bytes;
    return j;
}

int sw_fill_tile_device(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                        void* d_H, int h_elem_bytes, int32_t* d_P, int64_t row_stride, const int32_t* d_top,
                        const int32_t* d_left, int32_t* d_right, sw_result* d_result, void* stream_) {
    if (!d_H || !d_P) { set_err("sw_fill_tile_device: bad argument"); return SW_EINVAL; }
    return fill_one(c, scores, make_job(d_a, cols, d_b, rows, d_H, h_elem_bytes, d_P, 4, row_stride, d_top, d_left, d_right), -1, -1, d_result,
                    stream_, "sw_fill_tile_device");
}

int sw_fill_device(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                   void* d_H, int h_elem_bytes, int32_t* d_P, const int32_t* d_top, sw_result* d_result, void* stream_) {
    if (!d_H || !d_P) { set_err("sw_fill_device: bad argument"); return SW_EINVAL; }
    return fill_one(c, scores, make_job(d_a, cols, d_b, rows, d_H, h_elem_bytes, d_P, 4, cols + 1, d_top, nullptr, nullptr), -1, -1, d_result,
                    stream_, "sw_fill_device");
}

// compact P (one byte per predecessor code, same values 0..3, -1..-3 after the traceback) and matrix-less fills
// (d_H and/or d_P NULL: that matrix is not written; arg-max stays exact), SURVEY.md 8f-2
int sw_fill_device_ex(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                      void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes, const int32_t* d_top, sw_result* d_result,
                      void* stream_) {
    return fill_one(c, scores, make_job(d_a, cols, d_b, rows, d_H, h_elem_bytes, d_P, p_elem_bytes, cols + 1, d_top, nullptr, nullptr), -1, -1,
                    d_result, stream_, "sw_fill_device_ex");
}

// One row band of a (total_rows+1) x (cols+1) matrix as ONE persistent launch (multi-GPU): the halo row
// arrives and leaves as {tag, H} granules while the kernel runs.
int sw_fill_band_device(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, int64_t total_rows,
                        const sw_scores* scores, void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes,
                        const uint64_t* d_top_gran, uint32_t top_tag, uint64_t* d_bot_gran, uint32_t bot_tag, uint32_t* d_bot_done,
                        int reserve_cus, int concurrent, sw_result* d_result, void* stream_) {
    if ((d_top_gran && top_tag == 0) || (d_bot_gran && bot_tag == 0) || (d_bot_done && !d_bot_gran) || reserve_cus < 0 || total_rows < rows) {
        set_err("sw_fill_band_device: bad argument");
        return SW_EINVAL;
    }
    if (c && c->opt_engine != 0) { set_err("sw_fill_band_device needs the systolic engine"); return SW_EINVAL; }
    FillJob j = make_job(d_a, cols, d_b, rows, d_H, h_elem_bytes, d_P, p_elem_bytes, cols + 1, nullptr, nullptr, nullptr);
    j.d_top_gran = (const unsigned long long*)d_top_gran; j.d_bot_gran = (unsigned long long*)d_bot_gran; j.d_bot_done = d_bot_done;
    j.top_tag = top_tag; j.bot_tag = bot_tag; j.reserve_cus = reserve_cus; j.concurrent = concurrent != 0; j.total_rows = total_rows;
    return fill_one(c, scores, j, cols, total_rows, d_result, stream_, "sw_fill_band_device");
}

// The batch kernel proper (csrc/sw_batch.hip): one pair per wave, no inter-workgroup traffic.  Returns 1 when the batch is not
// eligible (the caller then runs it on the single-pair machinery): more than 8 distinct letters, scores that do not fit a signed
// byte, or a pair whose matrix does not fit a 2 GiB buffer descriptor.
static int batch_one_pair_per_wave(sw_ctx* c, const char* d_a, int64_t a_stride, int64_t cols, const char* d_b, int64_t b_stride, int64_t rows,
                                   int64_t npairs, const sw_scores* sc, int32_t* d_H, void* d_P, int p_elem_bytes, sw_result* d_results,
                                   hipStream_t stream) {
    if (sc->match > 127 || sc->match < -127 || sc->mismatch > 127 || sc->mismatch < -127) return 1;
    if ((double)(rows + 132) * (double)(cols + 1) * 4.0 >= 2147483648.0) return 1;
    const unsigned char* ua = (const unsigned char*)d_a;
    const unsigned char* ub = (const unsigned char*)d_b;
    // alphabet of the whole batch -> letter codes; the count decides whether the profile look-up applies
    const int64_t total_letters = (cols + rows) * npairs;
    const unsigned nscan = (unsigned)std::max<int64_t>(1, std::min<int64_t>((total_letters + 4095) / 4096, 2048));
    hipLaunchKernelGGL(swk::sw_prep_scan, dim3(nscan), dim3(256), 0, stream, ua, cols, a_stride, ub, rows, b_stride, npairs, c->d_part);
    // (one map for the half million blocks of sw_batch_codes: every one of them ORing 2048 maps by itself cost 29 ms per 100 000 pairs)
    hipLaunchKernelGGL(swk::sw_prep_reduce, dim3(1), dim3(256), 0, stream, c->d_part, (int)nscan);
    const int front = 64;
    const int64_t per = ((rows + front + 80 + 72 + 15) / 16) * 16;   // (+40: the drain steps of the delayed int8 P stores read on)
    const int C = cols <= 256 ? 4 : cols <= 512 ? 8 : 16;
    const int64_t nstrips = (cols + 64 * C - 1) / (64 * C);
    const int64_t bnd_per = nstrips > 1 ? ((rows + 160 + 3) / 4) * 4 : 0;
    // pairs per launch: bounds the workspace (codes: ~1.2 KB per 1024-row pair)
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(npairs, std::min<int64_t>((1ll << 30) / per, bnd_per ? (1ll << 30) / (bnd_per * 4) : npairs)));
    if ((size_t)(chunk * per) > c->bcodes_cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_bcodes) HIP_TRY(hipFree(c->d_bcodes));
        c->d_bcodes = nullptr; c->bcodes_cap = 0;
        if (hipMalloc((void**)&c->d_bcodes, (size_t)(chunk * per) + 64) != hipSuccess) { set_err("workspace allocation failed"); return SW_ENOMEM; }
        c->bcodes_cap = (size_t)(chunk * per);
    }
    if (bnd_per && (size_t)(chunk * bnd_per) > c->bnd_cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_bnd) HIP_TRY(hipFree(c->d_bnd));
        c->d_bnd = nullptr; c->bnd_cap = 0;
        if (hipMalloc((void**)&c->d_bnd, (size_t)(chunk * bnd_per) * 4) != hipSuccess) { set_err("workspace allocation failed"); return SW_ENOMEM; }
        c->bnd_cap = (size_t)(chunk * bnd_per);
    }
    // lane 0 of a later strip also reads boundary entries below the matrix that no strip of THIS call writes: they must not hold
    // an earlier call's scores (a cell outside the matrix may never exceed the cells of the matrix, see the arg-max in sw_batch.hip)
    if (bnd_per) HIP_TRY(hipMemsetAsync(c->d_bnd, 0, (size_t)(chunk * bnd_per) * 4, stream));
    const int64_t cells = (cols + 1) * (rows + 1);
    unsigned int nletters = 0;
    // Score-only batches whose scores fit 15 bits run two pairs per wave on packed 16-bit lanes (sw_batch_wave16: 5 VALU per two cells
    // instead of 8).  (debug bit 18: off, A/B runs)
    const bool fits16 = npairs >= 2 && C == 16 && (int64_t)sc->match * std::min(cols, rows) < 32000 && -sc->gap < 32000 && rows < 65000 && !(c->opt_debug & 262144);
    const bool k12 = (int64_t)sc->match * std::min(cols, rows) < 4096;   // (scores of 12 bits: the arg-max runs on score * 16 + column keys)
    // ... and with an int8 P as the only matrix, P codes from packed arithmetic (debug bit 21: off)
    const bool packed16 = fits16 && !d_H && (!d_P || (p_elem_bytes == 1 && !(c->opt_debug & 2097152)));
    for (int64_t k0 = 0; k0 < npairs; k0 += chunk) {
        const int64_t n = std::min(chunk, npairs - k0);
        hipLaunchKernelGGL(swk::sw_batch_codes, dim3((unsigned)std::min<int64_t>((per + 255) / 256, 64), (unsigned)std::min<int64_t>(n, 65535)), dim3(256), 0, stream,
                           ub + k0 * b_stride, rows, b_stride, c->d_bcodes, per, front, (const unsigned int*)c->d_part, 1, c->d_alpha + 64, n);
        HIP_TRY(hipGetLastError());
        if (k0 == 0) {   // the letter count (4 bytes) decides the path: the one host round trip of a batch call
            HIP_TRY(hipMemcpyAsync(&nletters, c->d_alpha + 64 + 256, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (nletters > 8u) return 1;
        }
        swk::BatchParams bp;
        memset(&bp, 0, sizeof bp);
        bp.a = ua + k0 * a_stride; bp.a_pstride = a_stride; bp.cols = cols;
        bp.bcode = c->d_bcodes; bp.bcode_pstride = per; bp.bfront = front;
        bp.rows = rows; bp.npairs = n; bp.atab = c->d_alpha + 64;
        bp.H = d_H ? d_H + k0 * cells : nullptr;
        bp.P = d_P ? (void*)((char*)d_P + k0 * cells * p_elem_bytes) : nullptr;
        bp.hp_pstride = cells;
        bp.match = sc->match; bp.mismatch = sc->mismatch; bp.ngap = -sc->gap;
        bp.bnd = c->d_bnd; bp.bnd_pstride = bnd_per;
        bp.results = d_results + k0;
        bp.debug = (int)(c->opt_debug & 7);   // bit 0: no matrix stores, bit 1: nt stores, bit 2: sc1 stores (experiments)
        const int pb = d_P ? p_elem_bytes : 0;
        if (packed16 && n >= 2) {
            const dim3 grid16((unsigned)(((n + 1) / 2 + 3) / 4)), block16(256);   // 4 waves = 8 pairs per workgroup
            if (d_P) {
                if (nletters <= 4u) { if (k12) hipLaunchKernelGGL((swk::sw_batch_wave16<true, true, true>), grid16, block16, 0, stream, bp); else hipLaunchKernelGGL((swk::sw_batch_wave16<true, false, true>), grid16, block16, 0, stream, bp); }
                else { if (k12) hipLaunchKernelGGL((swk::sw_batch_wave16<false, true, true>), grid16, block16, 0, stream, bp); else hipLaunchKernelGGL((swk::sw_batch_wave16<false, false, true>), grid16, block16, 0, stream, bp); }
            } else if (nletters <= 4u) { if (k12) hipLaunchKernelGGL((swk::sw_batch_wave16<true, true, false>), grid16, block16, 0, stream, bp); else hipLaunchKernelGGL((swk::sw_batch_wave16<true, false, false>), grid16, block16, 0, stream, bp); }
            else { if (k12) hipLaunchKernelGGL((swk::sw_batch_wave16<false, true, false>), grid16, block16, 0, stream, bp); else hipLaunchKernelGGL((swk::sw_batch_wave16<false, false, false>), grid16, block16, 0, stream, bp); }
            HIP_TRY(hipGetLastError());
            c->last_batch_kernel = 2;
            continue;
        }
        const dim3 grid((unsigned)((n + 3) / 4)), block(256);   // 4 pairs (waves) per workgroup
#define SB_LAUNCH(CC, PP) if (C == CC && pb == PP) hipLaunchKernelGGL((swk::sw_batch_wave<CC, PP>), grid, block, (size_t)c->opt_batch_lds, stream, bp);
        SB_LAUNCH(4, 0) SB_LAUNCH(4, 1) SB_LAUNCH(4, 4) SB_LAUNCH(8, 0) SB_LAUNCH(8, 1) SB_LAUNCH(8, 4) SB_LAUNCH(16, 0) SB_LAUNCH(16, 1) SB_LAUNCH(16, 4)
#undef SB_LAUNCH
        HIP_TRY(hipGetLastError());
    }
    if (c->last_batch_kernel != 2) c->last_batch_kernel = 1;
    c->last_grid = (chunk + 3) / 4; c->last_strips = nstrips;
    return SW_OK;
}

// (library-internal) Sizes ctx's workspaces for band-resident launches of this shape without launching anything.  sw_multi_create
// calls it for every band: a launch that has to allocate synchronises its stream, and with several persistent band kernels on
// one GPU that stream can share a hardware queue with a kernel that is still polling for its halo -- which only arrives once the
// host is past the launches (seen with 8 bands on one GPU: band 1 gave up after "band_wait_ms" and the relay stalled).
int sw_fill_band_reserve(sw_ctx* c, int64_t cols, int64_t rows, int64_t total_rows, const sw_scores* scores, int h_elem_bytes, int p_elem_bytes, int want_h,
                         void* stream_) {
    const sw_scores* sc = scores ? scores : &kDefaultScores;
    if (!c || cols <= 0 || rows <= 0) { set_err("sw_fill_band_reserve: bad argument"); return SW_EINVAL; }
    if (c->opt_engine != 0) { set_err("sw_fill_band_reserve needs the systolic engine"); return SW_EINVAL; }   // (no placeholder pointer may reach a launch)
    if (int rc = check_dims(cols, rows, sc, cols, total_rows)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    FillJob j = make_job((const char*)16, cols, (const char*)16, rows, want_h ? (void*)16 : nullptr, h_elem_bytes, (void*)16, p_elem_bytes, cols + 1, nullptr, nullptr, nullptr);
    j.d_top_gran = (const unsigned long long*)16; j.d_bot_gran = (unsigned long long*)16;   // (placeholders: only their presence matters)
    j.top_tag = j.bot_tag = 1; j.total_rows = total_rows; j.concurrent = true; j.reserve_only = true;
    j.d_keys = c->d_key; j.d_result = (sw_result*)16;
    std::unique_lock<std::mutex> lk(g_dev[c->device & 63].mu);
    return launch_fill(c, sc, j, (hipStream_t)stream_);
}

// BASELINE config 5: npairs independent cols x rows problems; pair k reads a at d_a + k*a_stride, b at d_b + k*b_stride.
// d_H and/or d_P may be NULL (that matrix is not written); the arg-max is exact in every mode.
int sw_batch_device_ex(sw_ctx* c, const char* d_a, int64_t a_stride, int64_t cols, const char* d_b, int64_t b_stride, int64_t rows,
                       int64_t npairs, const sw_scores* scores, int32_t* d_H, void* d_P, int p_elem_bytes, sw_result* d_results,
                       void* stream_) {
    const sw_scores* sc = scores ? scores : &kDefaultScores;
    if (!c || !d_a || !d_b || !d_results || npairs <= 0 || cols <= 0 || rows <= 0 || a_stride < cols || b_stride < rows ||
        (p_elem_bytes != 4 && p_elem_bytes != 1)) {
        set_err("sw_batch_device: bad argument");
        return SW_EINVAL;
    }
    if (c->opt_engine != 0) { set_err("sw_batch_device needs the systolic engine"); return SW_EINVAL; }
    if (int rc = check_dims(cols, rows, sc)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(c->device));
    DevOrder order(c, stream, false);
    if (order.rc) return order.rc;
    c->last_batch_kernel = 0;
    if (!(c->opt_debug & 65536)) {   // (debug bit 16: keep the batch on the single-pair machinery, A/B runs)
        int rc = batch_one_pair_per_wave(c, d_a, a_stride, cols, d_b, b_stride, rows, npairs, sc, d_H, d_P, p_elem_bytes, d_results, stream);
        if (rc != 1) return rc;      // 1: not eligible (alphabet of more than 8 letters, scores beyond a byte, huge pairs)
    }
    const int64_t chunk_max = 4096;   // pairs per launch: bounds the edge / padded-b workspace
    if ((size_t)std::min(npairs, chunk_max) > c->keys_cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c->d_keys) HIP_TRY(hipFree(c->d_keys));
        c->keys_cap = (size_t)std::min(npairs, chunk_max);
        if (hipMalloc((void**)&c->d_keys, c->keys_cap * 8) != hipSuccess) { c->d_keys = nullptr; c->keys_cap = 0; set_err("workspace allocation failed"); return SW_ENOMEM; }
    }
    HIP_TRY(hipMemsetAsync(c->d_key, 0, 16, stream));
    c->key_dirty = true;
    const int64_t cells = (cols + 1) * (rows + 1);
    for (int64_t k0 = 0; k0 < npairs; k0 += chunk_max) {
        const int64_t n = std::min(chunk_max, npairs - k0);
        HIP_TRY(hipMemsetAsync(c->d_keys, 0, (size_t)n * 8, stream));
        FillJob j = {d_a + k0 * a_stride, cols, d_b + k0 * b_stride, rows, d_H ? (void*)(d_H + k0 * cells) : nullptr, 4,
                     d_P ? (void*)((char*)d_P + k0 * cells * p_elem_bytes) : nullptr, cols + 1, nullptr, nullptr, nullptr, n, a_stride, b_stride,
                     cells, c->d_keys};
        j.p_elem_bytes = p_elem_bytes;
        if (int rc = launch_fill(c, sc, j, stream)) return rc;
        hipLaunchKernelGGL(swk::sw_finalize, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, c->d_keys,
                           (const unsigned int*)(c->d_key + 1), d_results + k0, (int)n);
        HIP_TRY(hipGetLastError());
    }
    return SW_OK;
}
int sw_batch_device(sw_ctx* c, const char* d_a, int64_t a_stride, int64_t cols, const char* d_b, int64_t b_stride, int64_t rows,
                    int64_t npairs, const sw_scores* scores, int32_t* d_H, int32_t* d_P, sw_result* d_results, void* stream_) {
    return sw_batch_device_ex(c, d_a, a_stride, cols, d_b, b_stride, rows, npairs, scores, d_H, d_P, 4, d_results, stream_);
}

// backtrack() of every pair of a batch (serial_smithW.c:262-277 per pair): one lane per pair walks its P from
// d_results[k].max_pos, negates the path and sets d_results[k].path_len; d_paths (optional) receives the visited
// pair-local indices, path_cap per pair.
int sw_batch_traceback_device(sw_ctx* c, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t npairs, int64_t* d_paths,
                              int64_t path_cap, sw_result* d_results, void* stream_) {
    if (!c || !d_P || !d_results || cols < 0 || rows < 0 || npairs <= 0 || (p_elem_bytes != 4 && p_elem_bytes != 1) || (d_paths && path_cap <= 0)) {
        set_err("sw_batch_traceback_device: bad argument");
        return SW_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    const int64_t cells = (cols + 1) * (rows + 1);
    const dim3 grid((unsigned)npairs), block(64);   // one wave per pair (csrc/sw_traceback.hip)
    if (p_elem_bytes == 4)
        hipLaunchKernelGGL(swk::sw_traceback_wave<int32_t>, grid, block, 0, (hipStream_t)stream_, (int32_t*)d_P, cols + 1, rows + 1, cells, (int64_t)-1, d_paths,
                           d_paths ? path_cap : 0, d_results, (int64_t*)nullptr, (unsigned int*)nullptr);
    else
        hipLaunchKernelGGL(swk::sw_traceback_wave<signed char>, grid, block, 0, (hipStream_t)stream_, (signed char*)d_P, cols + 1, rows + 1, cells, (int64_t)-1,
                           d_paths, d_paths ? path_cap : 0, d_results, (int64_t*)nullptr, (unsigned int*)nullptr);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}

int sw_fill_host(sw_ctx* c, const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores,
                 int32_t* H, int32_t* P, sw_result* result) {
    if (!c || !result || cols < 0 || rows < 0 || (cols > 0 && !a) || (rows > 0 && !b)) { set_err("sw_fill_host: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t cells = (size_t)(cols + 1) * (size_t)(rows + 1);
    char *d_a = nullptr, *d_b = nullptr; void *d_H = nullptr, *d_P = nullptr; sw_result* d_r = nullptr;
    int rc = SW_OK;
    auto cleanup = [&]() { (void)hipFree(d_a); (void)hipFree(d_b); if (d_H || d_P) (void)sw_free_outputs(c, d_H, d_P); (void)hipFree(d_r); };
    if (hipMalloc((void**)&d_a, (size_t)cols + 16) != hipSuccess || hipMalloc((void**)&d_b, (size_t)rows + 16) != hipSuccess ||
        hipMalloc((void**)&d_r, sizeof(sw_result)) != hipSuccess) {
        cleanup(); set_err("sw_fill_host: device allocation failed"); return SW_ENOMEM;
    }
    // (H and P from the placement-aware allocator: different classes of the HBM, classified by its store probe -- no trial fills)
    if ((rc = sw_alloc_outputs(c, nullptr, cols, nullptr, rows, scores, 4, 4, 0, &d_H, &d_P, nullptr)) != SW_OK) { cleanup(); return rc; }
    auto copy = [&](void* dst, const void* src, size_t n, hipMemcpyKind kind, const char* what) {
        if (rc != SW_OK || n == 0) return;
        const hipError_t e = hipMemcpy(dst, src, n, kind);
        if (e != hipSuccess) { set_err("sw_fill_host: copying %s failed: %s", what, hipGetErrorString(e)); rc = SW_EDEVICE; }
    };
    copy(d_a, a, (size_t)cols, hipMemcpyHostToDevice, "a");
    copy(d_b, b, (size_t)rows, hipMemcpyHostToDevice, "b");
    if (rc == SW_OK) rc = sw_fill_device(c, d_a, cols, d_b, rows, scores, d_H, 4, (int32_t*)d_P, nullptr, d_r, nullptr);
    if (rc == SW_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { set_err("fill kernel failed: %s", hipGetErrorString(e)); rc = SW_EDEVICE; }
    }
    copy(result, d_r, sizeof(sw_result), hipMemcpyDeviceToHost, "the result");
    if (rc == SW_OK && result->path_len < 0) { set_err("fill kernel: hand-off wait timed out"); rc = SW_ETIMEOUT; }
    // The copy-out is what a host-buffer caller pays: 2 x 4 B per cell over PCIe (16384^2: 2.1 GB, ~40 ms at 55 GB/s against a 0.8 ms
    // fill).  A pageable destination goes through the runtime's staging buffers at a fraction of that: pin the caller's matrices for the
    // duration of the copies where the platform allows it, and run the two copies on two streams.
    if (rc == SW_OK && (H || P)) {
        // Matrices fresh from calloc (what the reference's main hands over, serial_smithW.c:96-103) have no pages yet: whoever writes them first
        // pays 2.1 GB of page faults at 16384^2 -- one thread ~80 ms.  Every byte is about to be overwritten, so the pages are touched first,
        // by several threads (one write per 4 KiB page).
        if (cells * 4 >= (64u << 20)) {
            const unsigned nt = std::max(1u, std::min(std::min(16u, std::thread::hardware_concurrency()), (unsigned)(cells * 4 / (128u << 20))));
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([=]() {
                    const size_t n = cells * 4, lo = n / nt * t, hi = t + 1 == nt ? n : n / nt * (t + 1);
                    for (int32_t* M : {H, P})
                        if (M) for (size_t o = (lo + 4095) & ~(size_t)4095; o < hi; o += 4096) ((volatile char*)M)[o] = 0;
                });
            for (auto& x : th) x.join();
        }
        const bool pinH = H && cells * 4 >= (64u << 20) && hipHostRegister(H, cells * 4, hipHostRegisterDefault) == hipSuccess;
        const bool pinP = P && cells * 4 >= (64u << 20) && hipHostRegister(P, cells * 4, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
        hipStream_t s2 = nullptr;
        if (pinH && pinP && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { s2 = nullptr; (void)hipGetLastError(); }
        hipError_t e = hipSuccess;
        if (H) e = pinH ? hipMemcpyAsync(H, d_H, cells * 4, hipMemcpyDeviceToHost, nullptr) : hipMemcpy(H, d_H, cells * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && P) e = pinP ? hipMemcpyAsync(P, d_P, cells * 4, hipMemcpyDeviceToHost, s2) : hipMemcpy(P, d_P, cells * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (s2) (void)hipStreamDestroy(s2);
        if (pinH) (void)hipHostUnregister(H);
        if (pinP) (void)hipHostUnregister(P);
        if (e != hipSuccess) { set_err("sw_fill_host: copying the matrices back failed: %s", hipGetErrorString(e)); rc = SW_EDEVICE; }
    }
    cleanup();
    return rc;
}

// Adaptive dispatch in the spirit of omp_smithW-v7-adaptive.cpp:304-396 (serial / OpenMP / offload chosen per diagonal by
// its length): here the whole problem is sized once.  Below `SW_AUTO_CPU_CELLS` cells the host fill (sw_fill_cpu) wins
// against launch + transfer latency; everything else goes to the GPU of `ctx`.  (Several GPUs: sw_multi_*, the caller
// decides -- a single pair only scales once it is HBM-bound, about 65536^2 and up.)  path_len is set by the traceback.
int sw_align_auto(sw_ctx* c, const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores, int32_t* H, int32_t* P,
                  sw_result* result, int* used_gpu) {
    if (!result || !H || !P) { set_err("sw_align_auto: bad argument"); return SW_EINVAL; }
    const bool gpu = c && (double)cols * (double)rows >= 2.0e5;   // measured: a 512 x 512 host fill takes ~1.3 ms, launch + copies ~0.3 ms
    if (used_gpu) *used_gpu = gpu ? 1 : 0;
    int rc = gpu ? sw_fill_host(c, a, cols, b, rows, scores, H, P, result) : sw_fill_cpu(a, cols, b, rows, scores, H, P, result);
    if (rc != SW_OK) return rc;
    int64_t n = 0;
    rc = sw_traceback_host(P, cols, rows, result->max_pos, nullptr, 0, &n);
    result->path_len = n;
    return rc;
}

static int traceback_launch(sw_ctx* c, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos, int64_t* d_path, int64_t path_cap,
                            sw_result* d_result, int64_t* d_stop, hipStream_t stream, unsigned int* d_pathbits = nullptr) {
    // a big matrix is in no cache when the walk starts: a second wave reads ahead of the walking one (csrc/sw_traceback.hip)
    // (p_elem_bytes 0: a 2-bit matrix, four cells per byte)
    const dim3 block((double)(cols + 1) * (double)(rows + 1) * (p_elem_bytes ? (double)p_elem_bytes : 0.25) > 64.0e6 ? 128 : 64);
    if (p_elem_bytes == 0)
        hipLaunchKernelGGL(swk::sw_traceback_wave<swk::P2Cells>, dim3(1), block, 0, stream, (swk::P2Cells*)d_P, cols + 1, rows + 1, (int64_t)0, max_pos, d_path,
                           d_path ? path_cap : 0, d_result, d_stop, d_pathbits);
    else if (p_elem_bytes == 4)
        hipLaunchKernelGGL(swk::sw_traceback_wave<int32_t>, dim3(1), block, 0, stream, (int32_t*)d_P, cols + 1, rows + 1, (int64_t)0, max_pos, d_path,
                           d_path ? path_cap : 0, d_result, d_stop, (unsigned int*)nullptr);
    else
        hipLaunchKernelGGL(swk::sw_traceback_wave<signed char>, dim3(1), block, 0, stream, (signed char*)d_P, cols + 1, rows + 1, (int64_t)0, max_pos, d_path,
                           d_path ? path_cap : 0, d_result, d_stop, (unsigned int*)nullptr);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}
// (library-internal: the traceback that also reports where the walk stopped -- sw_multi_traceback hops bands with it)
int sw_traceback_stop_device(sw_ctx* c, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos, sw_result* d_result, int64_t* d_stop,
                             void* stream_) {
    if (!c || !d_P || !d_result || !d_stop || max_pos < 0 || max_pos >= (cols + 1) * (rows + 1)) { set_err("sw_traceback_stop_device: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    return traceback_launch(c, d_P, p_elem_bytes, cols, rows, max_pos, nullptr, 0, d_result, d_stop, (hipStream_t)stream_);
}

int sw_traceback_device_ex(sw_ctx* c, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos, int64_t* d_path,
                           int64_t path_cap, sw_result* d_result, void* stream_) {
    if (!c || !d_P || !d_result || cols < 0 || rows < 0 || max_pos < 0 || max_pos >= (cols + 1) * (rows + 1) ||
        (p_elem_bytes != 4 && p_elem_bytes != 1)) {
        set_err("sw_traceback_device: bad argument");
        return SW_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    return traceback_launch(c, d_P, p_elem_bytes, cols, rows, max_pos, d_path, path_cap, d_result, nullptr, (hipStream_t)stream_);
}
int sw_traceback_device(sw_ctx* c, int32_t* d_P, int64_t cols, int64_t rows, int64_t max_pos, int64_t* d_path,
                        int64_t path_cap, sw_result* d_result, void* stream_) {
    return sw_traceback_device_ex(c, d_P, 4, cols, rows, max_pos, d_path, path_cap, d_result, stream_);
}

// Output matrices placed for speed.  Physical HBM falls into a few coarse classes (regions of tens of GiB), and two store streams into
// the SAME class run ~1.4x slower than into different ones; a fill stores H[r][c] and P[r][c] together, so a 16384^2 fill takes 0.79 ms
// with H and P in different classes and 1.05 ms with both in one (profiles/r04_placement_classes_probe.log).  Two
// back-to-back hipMallocs land in one class.  trials <= 0 (the default): candidates for P -- from the second one on behind a temporary
// spacer allocation, so that they come from elsewhere in the HBM -- are CLASSIFIED against H with the two-stream store probe of
// csrc/sw_place.hip (~0.3 ms per candidate, no fill of the caller's problem), the first one in another class is kept.  trials == 1: a plain
// pair.  trials > 1: round 3's search with trial fills of the caller's problem (kept for A/B runs).
int sw_place_pair_ratio(void* d_X, size_t xbytes, void* d_Y, size_t ybytes, float* ratio, float* ms_together);   // sw_place.hip

static int alloc_outputs_probed(sw_ctx* c, size_t hbytes, size_t pbytes, void** d_H, void** d_P, float* trial_ms, int ntrial_ms) {
    const size_t phase = 4u << 20;
    void* H = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    if (hipMalloc(&H, hbytes ? hbytes : 1) != hipSuccess) { (void)hipGetLastError(); set_err("sw_alloc_outputs: %zu bytes do not fit", hbytes); return SW_ENOMEM; }
    // what a spacer can cost: 0.3 ms flat on fresh memory, 30-50 ms per GiB where the driver first wipes memory that was in use before (seen:
    // 32 GiB in 0.2 ms, the next 64 GiB in 3.3 s) -- nothing tells beforehand, so a spacer is only tried while its worst case fits the budget
    double ms_per_gib = 40.0;
    struct Cand { void* base; void* P; float ratio; };
    std::vector<Cand> cands;
    // Candidates for P: three plain ones (a class boundary may be right here), then behind temporary spacer allocations (taken only while
    // 8 GiB of head room remain, released as soon as the candidate behind them exists: the next spacer, of another size, then lands
    // elsewhere).  The classes are regions of 16 .. 120 GiB in the order the driver hands memory out.  What a spacer costs depends on the
    // box: memory that was in use before (by this or an earlier process) is wiped by the driver at ~30 GiB/s when it changes hands, fresh
    // memory costs 0.3 ms per allocation -- so the search runs against a time budget (option "placement_budget_ms", default 1500 ms; a
    // caller that fills many times into the pair raises it) and settles for the best candidate seen when that is spent; a spacer whose
    // allocation alone could overrun the budget (at the wipe rate) is not tried: the default admits spacers up to 37 GiB.  The spacer that worked is
    // remembered per context and tried first next time.
    static const int kSpacerGiB[14] = {0, 0, 0, 32, 64, 16, 96, 48, 128, 24, 80, 8, 160, 112};
    const bool debug = getenv("SW_PLACE_DEBUG") != nullptr;
    // (a matrix of many GiB spans several classes itself: more sample windows, and the best of a few candidates rather than the first good one)
    const bool big = std::max(hbytes, pbytes) > (6ull << 30);
    const float accept = big ? 1.40f : 1.5f;   // another class: ~1.3-1.45; the same class: ~2.0
    int best = -1, rc = SW_OK;
    for (int i = 0; i < 14; ++i) {
        int gib = kSpacerGiB[i];
        if (i == 3 && c->place_spacer_gib > 0) gib = c->place_spacer_gib;                 // what worked last time, first
        else if (i > 3 && gib == c->place_spacer_gib) continue;
        size_t sp = (size_t)gib << 30;
        if (sp && elapsed_ms() > (double)c->opt_place_budget_ms) break;
        if (sp && elapsed_ms() + ms_per_gib * (double)gib > (double)c->opt_place_budget_ms) continue;   // (this spacer alone would overrun the budget: a smaller one may not)
        if (sp) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < sp + pbytes + (8ull << 30)) continue;   // not enough head room for this one
        }
        void* spacer = nullptr;
        const double ts = elapsed_ms();
        if (sp && hipMalloc(&spacer, sp) != hipSuccess) { (void)hipGetLastError(); spacer = nullptr; continue; }
        const double ts1 = elapsed_ms();
        Cand k = {nullptr, nullptr, 0.f};
        const hipError_t e = hipMalloc(&k.base, pbytes + phase);
        const double ts2 = elapsed_ms();
        if (spacer) (void)hipFree(spacer);   // (it only steered where P landed)
        if (sp) ms_per_gib = std::max(ms_per_gib, (elapsed_ms() - ts) / (double)gib);
        if (debug) fprintf(stderr, "sw_alloc_outputs: spacer %d GiB: malloc %.1f ms, candidate malloc %.1f ms, free %.1f ms\n", gib, ts1 - ts, ts2 - ts1, elapsed_ms() - ts2);
        if (e != hipSuccess) { (void)hipGetLastError(); break; }
        // P two MiB out of phase with H modulo 4 MiB (round 1: neighbouring 2 MiB pages of the two streams)
        const uintptr_t want = ((uintptr_t)H + (2u << 20)) % phase;
        k.P = (char*)k.base + (want + phase - ((uintptr_t)k.base % phase)) % phase;
        float ms = 0.f;
        rc = sw_place_pair_ratio(H, hbytes, k.P, pbytes, &k.ratio, &ms);
        cands.push_back(k);
        if (rc != SW_OK) break;
        if (debug) fprintf(stderr, "sw_alloc_outputs: candidate %d (spacer %d GiB): H %p P %p ratio %.3f (%.3f ms), %.1f ms so far\n", i, gib, H, k.P, k.ratio, ms, elapsed_ms());
        if (trial_ms && (int)cands.size() <= ntrial_ms) trial_ms[cands.size() - 1] = ms;
        const int prev_best = best;
        if (best < 0 || k.ratio < cands[best].ratio) best = (int)cands.size() - 1;
        if (big) {   // candidates of tens of GiB: only the best so far stays allocated
            const int loser = best == (int)cands.size() - 1 ? prev_best : (int)cands.size() - 1;
            if (loser >= 0 && cands[loser].base) { (void)hipFree(cands[loser].base); cands[loser].base = nullptr; }
        }
        if (k.ratio < accept) { if (gib) c->place_spacer_gib = gib; break; }
        if (big && cands.size() >= 5) break;   // (... and the best of five)
    }
    // The slide.  Where no candidate is good -- matrices of many GiB span classes themselves, and so does every candidate; on a box whose
    // memory was in use before, the driver hands out the little clean memory it has, all of one class, whatever the spacers (seen: twelve
    // candidates in a row at ratio 2.0) -- P is allocated with slack and SLID inside its own allocation in steps of 4 GiB: one allocation is
    // backed by whatever memory there is, dirty regions of the other classes included, and the classes are regions of 8 .. 120 GiB, so the
    // slide changes which parts of H and P meet.  The best offset is kept; the slack stays allocated while the pair lives.  Many-GiB pairs
    // take up to 32 GiB of slack by themselves (a few per cent of a 288 GB part for fills that are ~25 % faster); smaller pairs only what
    // option "placement_hold_gib" allows (default 0: bench.py, which fills thousands of times into the pair, allows 48).  Only while the
    // budget covers the worst case of that allocation.
    c->last_place_held_gib = 0;
    const size_t hold_max = big ? (32ull << 30) : ((size_t)c->opt_place_hold_gib << 30);
    const bool force_slide = getenv("SW_PLACE_FORCE_SLIDE") != nullptr;   // (tests: take the slide whatever the candidates were)
    if (hold_max >= (8ull << 30) && rc == SW_OK && best >= 0 && (cands[best].ratio >= accept || force_slide)) {
        size_t fr = 0, tot = 0;
        size_t slack = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > pbytes + (24ull << 30)) slack = std::min<size_t>(hold_max, (fr - pbytes - (16ull << 30)) & ~((4ull << 30) - 1));
        const double worst_ms = 40.0 * (double)((pbytes + slack) >> 30);
        if (slack >= (8ull << 30) && elapsed_ms() + worst_ms <= (double)c->opt_place_budget_ms) {
            void* blk = nullptr;
            if (hipMalloc(&blk, pbytes + slack + phase) == hipSuccess) {
                float bratio = cands[best].ratio; void* bP = nullptr;
                for (size_t off = 0; off <= slack && rc == SW_OK; off += (4ull << 30)) {
                    char* q = (char*)blk + off;
                    const uintptr_t want = ((uintptr_t)H + (2u << 20)) % phase;
                    q += (want + phase - ((uintptr_t)q % phase)) % phase;
                    float r = 0.f, ms = 0.f;
                    rc = sw_place_pair_ratio(H, hbytes, q, pbytes, &r, &ms);
                    if (debug) fprintf(stderr, "sw_alloc_outputs: slide %zu GiB: ratio %.3f, %.1f ms so far\n", off >> 30, r, elapsed_ms());
                    if (rc == SW_OK && (r < bratio || (force_slide && !bP))) { bratio = r; bP = q; }
                    if (r < accept && !force_slide) break;
                }
                if (rc == SW_OK && bP) {
                    Cand k = {blk, bP, bratio};
                    cands.push_back(k);
                    best = (int)cands.size() - 1;
                    c->last_place_held_gib = (int64_t)(slack >> 30);
                } else {
                    (void)hipFree(blk);
                }
            } else {
                (void)hipGetLastError();
            }
        }
    }
    for (int i = 0; i < (int)cands.size(); ++i)
        if ((i != best || rc != SW_OK) && cands[i].base) (void)hipFree(cands[i].base);
    if (rc != SW_OK || best < 0) {
        (void)hipFree(H);
        if (rc == SW_OK) { set_err("sw_alloc_outputs: %zu + %zu bytes do not fit", hbytes, pbytes); rc = SW_ENOMEM; }
        return rc;
    }
    c->last_place_ratio = cands[best].ratio;
    c->pair_ratio[cands[best].P] = cands[best].ratio;
    *d_H = H; *d_P = cands[best].P;
    c->out_base[cands[best].P] = cands[best].base;
    return SW_OK;
}

int sw_alloc_outputs(sw_ctx* c, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores, int h_elem_bytes,
                     int p_elem_bytes, int trials, void** d_H, void** d_P, float* trial_ms) {
    if (!c || !d_H || !d_P || cols < 0 || rows < 0 || (h_elem_bytes != 4 && h_elem_bytes != 8) || (p_elem_bytes != 4 && p_elem_bytes != 1) ||
        (trials > 1 && (!d_a || !d_b))) {
        set_err("sw_alloc_outputs: bad argument");
        return SW_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t cells = (size_t)(cols + 1) * (size_t)(rows + 1);
    const size_t hbytes = cells * (size_t)h_elem_bytes, pbytes = cells * (size_t)p_elem_bytes;
    // (below half a GiB of output the strip chain bounds a fill, not the stores: a plain pair)
    if (trials <= 0) {
        if (hbytes + pbytes >= (512ull << 20)) {
            for (int i = 0; i < 16 && trial_ms; ++i) trial_ms[i] = 0.f;
            return alloc_outputs_probed(c, hbytes, pbytes, d_H, d_P, trial_ms, 16);
        }
        trials = 1;
    }
    const size_t phase = 4u << 20;
    struct Cand { void* H; void* Pbase; void* P; void* spacer; float ms; };
    std::vector<Cand> cands;
    std::vector<void*> Hs;
    sw_result* d_res = nullptr;
    if (hipMalloc((void**)&d_res, sizeof(sw_result)) != hipSuccess) { set_err("sw_alloc_outputs: allocation failed"); return SW_ENOMEM; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int best = -1, rc = SW_OK;
    for (int i = 0; i < trials; ++i) {
        Cand k = {nullptr, nullptr, nullptr, nullptr, 0.f};
        // H stays where it is and the candidates differ in where P lands; half way through a second H is tried as well
        if (i == 0 || (i == trials / 2 && trials >= 6 && hbytes < (8ull << 30))) {
            void* h = nullptr;
            if (hipMalloc(&h, hbytes ? hbytes : 1) != hipSuccess) { (void)hipGetLastError(); if (i == 0) break; }
            else Hs.push_back(h);
        }
        if (Hs.empty()) break;
        k.H = Hs.back();
        // Measured (scripts/ab_arena.py, profiles/r02_placement_arena.log): inside one 96 GiB allocation a 16384^2 fill takes
        // 1.12 ms when H and P lie on different sides of the 64 GiB mark and 1.38-1.47 ms when they share a side, whatever
        // their distance.  So from the second candidate on a spacer of 64 GiB (then 32, 96, 48, 80) is allocated between
        // H and P -- and released again when the search ends: no memory stays held.
        // (a box where five of the first six candidates were slow has been seen: the search goes on to sixteen before it settles for a slow one)
        static const int kSpacerGiB[16] = {0, 64, 96, 32, 128, 48, 160, 80, 16, 112, 144, 24, 176, 72, 104, 56};
        size_t sp = (i > 0 && hbytes < (8ull << 30)) ? (size_t)kSpacerGiB[i % 16] << 30 : 0;
        if (sp) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < sp + pbytes + (8ull << 30)) sp = 0;   // not enough head room: plain candidate
        }
        if (!sp || hipMalloc(&k.spacer, sp) != hipSuccess) { (void)hipGetLastError(); k.spacer = nullptr; }
        if (hipMalloc(&k.Pbase, pbytes + phase) != hipSuccess) { (void)hipGetLastError(); if (k.spacer) (void)hipFree(k.spacer); break; }
        // P two MiB out of phase with H modulo 4 MiB
        const uintptr_t want = ((uintptr_t)k.H + (2u << 20)) % phase;
        const uintptr_t off = (want + phase - ((uintptr_t)k.Pbase % phase)) % phase;
        k.P = (char*)k.Pbase + off;
        if (trials > 1) {
            for (int f = 0; f < 4 && rc == SW_OK; ++f) {
                if (f == 1) (void)hipEventRecord(e0, nullptr);
                rc = sw_fill_device_ex(c, d_a, cols, d_b, rows, scores, k.H, h_elem_bytes, k.P, p_elem_bytes, nullptr, d_res, nullptr);
            }
            (void)hipEventRecord(e1, nullptr);
            if (rc == SW_OK && hipEventSynchronize(e1) != hipSuccess) { set_err("sw_alloc_outputs: trial fill failed"); rc = SW_EDEVICE; }
            if (rc == SW_OK) { (void)hipEventElapsedTime(&k.ms, e0, e1); k.ms /= 3.f; }
        }
        // the spacer only steers where P lands: release it before the next candidate is placed
        if (k.spacer) { (void)hipFree(k.spacer); k.spacer = nullptr; }
        cands.push_back(k);
        if (rc != SW_OK) break;
        if (trial_ms) trial_ms[i] = k.ms;
        if (best < 0 || k.ms < cands[best].ms) best = (int)cands.size() - 1;
        if ((int)cands.size() >= std::min(trials, 6)) {   // several placements seen (there are half-good ones) and clearly in the fast mode: stop looking
            float worst = 0.f;   // (the first candidate also pays the one-time costs of the first launches: not a placement signal)
            for (size_t x = 1; x < cands.size(); ++x) worst = std::max(worst, cands[x].ms);
            if (cands[best].ms < 0.80f * worst) {   // (fast and slow mode are 20-25 % apart; the two-column kernel also has a half-good one in between)
                for (int j = i + 1; j < trials && trial_ms; ++j) trial_ms[j] = 0.f; break; }
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d_res);
    for (int i = 0; i < (int)cands.size(); ++i) {
        if (cands[i].spacer) (void)hipFree(cands[i].spacer);
        if (i != best || rc != SW_OK) (void)hipFree(cands[i].Pbase);
    }
    for (void* h : Hs)
        if (best < 0 || rc != SW_OK || h != cands[best].H) (void)hipFree(h);
    if (rc != SW_OK) return rc;
    if (best < 0) { set_err("sw_alloc_outputs: %zu + %zu bytes do not fit", hbytes, pbytes); return SW_ENOMEM; }
    *d_H = cands[best].H; *d_P = cands[best].P;
    c->out_base[cands[best].P] = cands[best].Pbase;
    if (hbytes + pbytes >= (512ull << 20)) {   // (which kind of pair it is decides the strip geometry of fills into it: launch_fill)
        float r = 0.f, ms = 0.f;
        if (sw_place_pair_ratio(*d_H, hbytes, *d_P, pbytes, &r, &ms) == SW_OK) { c->pair_ratio[*d_P] = r; c->last_place_ratio = r; }
    }
    return SW_OK;
}

int sw_free_outputs(sw_ctx* c, void* d_H, void* d_P) {
    if (!c) { set_err("sw_free_outputs: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (d_H) HIP_TRY(hipFree(d_H));
    if (d_P) {
        auto it = c->out_base.find(d_P);
        void* base = (it != c->out_base.end()) ? it->second : d_P;
        if (it != c->out_base.end()) c->out_base.erase(it);
        c->pair_ratio.erase(d_P);
        HIP_TRY(hipFree(base));
    }
    return SW_OK;
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

int sw_p8_to_p32_device(sw_ctx* c, const void* d_P8, int32_t* d_P32, int64_t count, void* stream_) {
    if (!c || count < 0 || (count > 0 && (!d_P8 || !d_P32))) { set_err("sw_p8_to_p32_device: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (count == 0) return SW_OK;
    const unsigned nb = (unsigned)std::min<int64_t>((count + 256 * 16 - 1) / (256 * 16), 8192);
    hipLaunchKernelGGL(swk::sw_widen_p8, dim3(nb), dim3(256), 0, (hipStream_t)stream_, (const signed char*)d_P8, d_P32, (size_t)count);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}

// ---- 2-bit predecessor matrix: 4 codes per byte + a path bitmap; csrc/sw_kernels.hip, csrc/sw_traceback.hip
int sw_p_to_p2_device(sw_ctx* c, const void* d_P, int p_elem_bytes, void* d_P2, uint32_t* d_pathbits, int64_t count, void* stream_) {
    if (!c || count < 0 || (count > 0 && (!d_P || !d_P2)) || (p_elem_bytes != 1 && p_elem_bytes != 4)) { set_err("sw_p_to_p2_device: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (count == 0) return SW_OK;
    const unsigned nb = (unsigned)std::min<int64_t>(((count + 31) / 32 + 255) / 256, 16384);
    if (p_elem_bytes == 1) hipLaunchKernelGGL(swk::sw_pack_p2<signed char>, dim3(nb), dim3(256), 0, (hipStream_t)stream_, (const signed char*)d_P, (unsigned char*)d_P2, d_pathbits, (size_t)count);
    else hipLaunchKernelGGL(swk::sw_pack_p2<int32_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream_, (const int32_t*)d_P, (unsigned char*)d_P2, d_pathbits, (size_t)count);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}
int sw_p2_to_p32_device(sw_ctx* c, const void* d_P2, const uint32_t* d_pathbits, int32_t* d_P32, int64_t count, void* stream_) {
    if (!c || count < 0 || (count > 0 && (!d_P2 || !d_P32))) { set_err("sw_p2_to_p32_device: bad argument"); return SW_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (count == 0) return SW_OK;
    const unsigned nb = (unsigned)std::min<int64_t>(((count + 3) / 4 + 255) / 256, 16384);
    hipLaunchKernelGGL(swk::sw_unpack_p2, dim3(nb), dim3(256), 0, (hipStream_t)stream_, (const unsigned char*)d_P2, d_pathbits, d_P32, (size_t)count);
    HIP_TRY(hipGetLastError());
    return SW_OK;
}
int sw_traceback_p2_device(sw_ctx* c, const void* d_P2, int64_t cols, int64_t rows, int64_t max_pos, uint32_t* d_pathbits, int64_t* d_path,
                           int64_t path_cap, sw_result* d_result, void* stream_) {
    if (!c || !d_P2 || !d_result || cols < 0 || rows < 0 || max_pos < 0 || max_pos >= (cols + 1) * (rows + 1)) {
        set_err("sw_traceback_p2_device: bad argument");
        return SW_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    return traceback_launch(c, const_cast<void*>(d_P2), 0, cols, rows, max_pos, d_path, path_cap, d_result, nullptr, (hipStream_t)stream_, d_pathbits);
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
