/*
 * swhip.h -- C-ABI of the MI355X-native Smith-Waterman DP-fill engine (libswhip.so).
 *
 * Drop-in boundary for the reference's hot path.  The reference has no
 * FFI; each entry point below names the reference code it replaces (paths relative to the
 * reference repository root).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SW_E* code on failure, never exit()s
 *     (the reference's CUDA path prints and exit(0)s, simple-cuda/sw-default-discrete.cu:101-108);
 *     sw_last_error() returns a thread-local description of the last failure.
 *   - cols = len(a) = matrix columns, rows = len(b) = matrix rows (serial_smithW.c:72-77);
 *     H and P are (rows+1) x (cols+1), row-major, row stride cols+1 (serial_smithW.c:192).
 *   - "d_" pointers are DEVICE (HBM) pointers, everything else is host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef SWHIP_H
#define SWHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SW_OK 0
#define SW_EINVAL (-22)   /* bad argument */
#define SW_ENOMEM (-12)   /* host or device allocation failed */
#define SW_EDEVICE (-5)   /* HIP runtime error */
#define SW_ETIMEOUT (-62) /* an in-kernel hand-off wait gave up (never expected) */
#define SW_ENODEV (-19)   /* no usable GPU */

/* predecessor codes, serial_smithW.c:23-27 */
#define SW_PATH (-1)
#define SW_NONE 0
#define SW_UP 1
#define SW_LEFT 2
#define SW_DIAGONAL 3

/* serial_smithW.c:59-61 (matchScore, missmatchScore, gapScore); default {3,-3,-2}. gap must be <= 0. */
typedef struct { int32_t match, mismatch, gap; } sw_scores;

/* max_pos: linear index rows-major of the arg-max cell, lowest index among ties, 0 if H == 0
 * everywhere (what the scan at serial_smithW.c:240-242 yields); max_score = H[max_pos];
 * path_len: cells negated by the traceback (0 until a traceback ran). */
typedef struct { int64_t max_pos; int64_t max_score; int64_t path_len; } sw_result;

/* One per GPU; one host thread drives a ctx at a time.  Fills of one DEVICE are serialised by the library (a fill's
 * workgroups wait for each other, so two fills must not share the CUs): a fill enqueued on another stream than the
 * previous fill of that device first waits, on the device, for that stream.  Contexts of different devices are
 * independent. */
typedef struct sw_ctx sw_ctx;

const char* sw_last_error(void);
const char* sw_version(void);

/* ---- input side ---------------------------------------------------------------------------
 * sw_generate: replaces generate(), serial_smithW.c:334-361, bit-exact incl. its draw order
 * (cols+1 rand() draws for a, then rows+1 for b; glibc TYPE_3 rand() restated, no libc state).
 * seed 1 == serial_smithW.c (which never calls srand).  a holds cols+1 bytes, b rows+1. */
int sw_generate(int64_t cols, int64_t rows, uint32_t seed, char* a, char* b);

/* FASTA input, the step before the path for real sequences (the reference only has
 * generate()).  Reads record `record` (0-based) of the file: '>' starts a record, ';' lines are comments,
 * white space is dropped, letters are upper-cased; a file without '>' is one record.  *len receives the
 * sequence length; seq (may be NULL to query the length) receives at most cap bytes, no terminator. */
int sw_read_fasta(const char* path, int64_t record, char* seq, int64_t cap, int64_t* len);

/* ---- wavefront indexing: nElement / calcFirstDiagElement, omp_smithW.c:260-275, 282-291.
 * m, n are the padded sizes (cols+1, rows+1); i in [1, m+n-3]. */
int64_t sw_nelement(int64_t i, int64_t m, int64_t n);
void sw_first_diag_element(int64_t i, int64_t m, int64_t n, int64_t* si, int64_t* sj);

/* ---- context ------------------------------------------------------------------------------ */
int sw_create(int device, sw_ctx** out);
void sw_destroy(sw_ctx* ctx);

/* ---- the hot path ---------------------------------------------------------------------------
 * sw_fill_device: replaces the fill loop + similarityScore/matchMissmatchScore
 * (serial_smithW.c:141-145, 187-256; omp_smithW.c:203-216) and the rotated family's
 * smithWaterman(a,b,w,h,H,P,&maxloc) (rotated-cuda/sw-rotated-omp.cc:192-209).
 * Asynchronous on `stream`.  d_H/d_P need not be initialised (row 0 / column 0 are written).
 *   h_elem_bytes : 4 -> d_H is int32_t*, 8 -> d_H is int64_t* (same values, widened)
 *   d_top        : optional (may be NULL) int32 H values of the row above this band, cols+1
 *                  entries (multi-GPU row bands); NULL == zeros (a whole matrix)
 *   d_result     : device sw_result; max_pos/max_score valid when the stream has drained
 * Placement: H[r][c] and P[r][c] are written within a fraction of a microsecond of each other; when both buffers were
 * mapped to the same class of the physical HBM (the usual outcome of two back-to-back hipMallocs) a 16384^2 fill takes
 * 1.05 ms instead of 0.79 ms on MI355X.  sw_alloc_outputs() below hands out a pair that avoids it.  Results do not depend on it. */
int sw_fill_device(sw_ctx* ctx, const char* d_a, int64_t cols, const char* d_b, int64_t rows,
                   const sw_scores* scores, void* d_H, int h_elem_bytes, int32_t* d_P,
                   const int32_t* d_top, sw_result* d_result, void* stream);

/* Same with a compact predecessor matrix and/or without one of the matrices (the rolling-buffer
 * variants of the reference, rotated-cuda/sw-rotated-omp.cc:214-224, likewise keep only what the caller asks for):
 *   p_elem_bytes 4 -> d_P is int32_t* (identical to sw_fill_device), 1 -> d_P is int8_t* holding the same codes
 *                (0..3; -1..-3 after a traceback): a quarter of the P traffic and footprint;
 *   d_H == NULL  -> H is not written ("P-only": 1 or 4 B/cell, enough for the traceback);
 *   d_P == NULL  -> P is not written; both NULL = score-only.
 * max_pos / max_score are exact in every mode.  Systolic engine only. */
int sw_fill_device_ex(sw_ctx* ctx, const char* d_a, int64_t cols, const char* d_b, int64_t rows,
                      const sw_scores* scores, void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes,
                      const int32_t* d_top, sw_result* d_result, void* stream);

/* Tile of a bigger matrix (multi-GPU row bands x column chunks).  d_H / d_P point at the
 * tile's corner cell (row above / column left of its first cell) inside matrices of row stride
 * `row_stride` elements; d_top: cols+1 H values of the row above (NULL = zeros), d_left: rows+1 H values
 * of the column to the left (NULL = zeros; d_left[0] is the corner), d_right (optional): receives the
 * rows+1 H values of the tile's last column = the next tile's d_left.  The tile's last row is the next
 * band's d_top.  d_result->max_pos is relative to the corner with the full row stride.
 * Row 0 / column 0 of the tile are not re-written when they belong to a neighbour (P untouched). */
int sw_fill_tile_device(sw_ctx* ctx, const char* d_a, int64_t cols, const char* d_b, int64_t rows,
                        const sw_scores* scores, void* d_H, int h_elem_bytes, int32_t* d_P, int64_t row_stride,
                        const int32_t* d_top, const int32_t* d_left, int32_t* d_right, sw_result* d_result,
                        void* stream);

/* One ROW BAND of a bigger (total_rows+1) x (cols+1) matrix as ONE persistent launch: the multi-GPU decomposition of
 * SURVEY.md 8e (the reference has no multi-GPU code; the order preserved is omp_smithW.c:203-216).  The band's halo
 * row arrives and its last row leaves as 8-byte granules  (uint64)tag << 32 | (uint32)H  , one per column, WHILE the
 * kernel runs: a strip starts as soon as its 64 granules of the row above carry `top_tag`, so the launch can be
 * enqueued before the band above has produced anything, and whoever moves the halo (RCCL recv, a copy, a peer store)
 * may deliver it in any order and chunking -- the data is its own flag.
 *   d_H, d_P      : band-local (rows+1) x (cols+1) storage; row 0 is the halo row (H written, P left alone).  Either
 *                   may be NULL (not written), p_elem_bytes 4 or 1 as in sw_fill_device_ex
 *   d_top_gran    : cols+1 granules of the row above, or NULL for the first band (zeros)
 *   d_bot_gran    : receives cols+1 granules of the band's last row (tag = bot_tag), or NULL
 *   d_bot_done    : optional, one uint32 per 63-column strip (ceil(cols/63)), may be host-pinned memory: set to bot_tag,
 *                   system scope, after the strip's granules were released -- lets a host thread forward finished chunks
 *   tags          : non-zero, and different from what the buffers held before (e.g. a per-fill counter)
 *   reserve_cus   : CUs the launch leaves free for other kernels (halo transfers); 0 = use all
 *   concurrent    : non-zero = do not order this launch behind fills on other streams of the device (the caller keeps
 *                   the sum of the grids within the CUs, option "max_blocks")
 *   d_result      : band-local arg-max (max_pos relative to the band's row 0, stride cols+1)
 * total_rows bounds the scores a halo can carry (range check).  Patience of the halo poll: option "band_wait_ms". */
int sw_fill_band_device(sw_ctx* ctx, const char* d_a, int64_t cols, const char* d_b, int64_t rows, int64_t total_rows,
                        const sw_scores* scores, void* d_H, int h_elem_bytes, void* d_P, int p_elem_bytes,
                        const uint64_t* d_top_gran, uint32_t top_tag, uint64_t* d_bot_gran, uint32_t bot_tag,
                        uint32_t* d_bot_done, int reserve_cus, int concurrent, sw_result* d_result, void* stream);

/* ---- one matrix over several GPUs of ONE process (one process per GPU: smith-waterman_amd/multi.py) ----
 * Row bands, one per entry of `devices` (an id may repeat: the bands then share that GPU's CUs), every band one
 * band-resident launch, all launched at once; finished column chunks of a band's last row are forwarded to the next
 * band with peer copies over xGMI while the kernels run.  a, b: HOST sequences.
 *   sw_multi_create    allocates the band-local matrices (H int32 unless want_h == 0; P int32 or int8)
 *   sw_multi_fill      one fill of the whole matrix (blocking); result: global arg-max by the serial rule
 *   sw_multi_traceback backtrack() across the bands (negates P along the path), total path length
 *   sw_multi_band_info device, rows (lo, hi] and device pointers of band g (band-local (hi-lo+1) x (cols+1), row 0 = halo) */
typedef struct sw_multi sw_multi;
int sw_multi_create(const int* devices, int ndev, const char* a, int64_t cols, const char* b, int64_t rows,
                    int p_elem_bytes, int want_h, sw_multi** out);
int sw_multi_fill(sw_multi* m, const sw_scores* scores, int nchunks, sw_result* result);
int sw_multi_traceback(sw_multi* m, int64_t* path_len);
int sw_multi_band_info(sw_multi* m, int g, int* device, int64_t* row_lo, int64_t* row_hi, void** d_H, void** d_P);
int sw_multi_nbands(sw_multi* m);
double sw_multi_seconds(sw_multi* m);   /* wall time of the last sw_multi_fill */
void sw_multi_free(sw_multi* m);

/* Batch of npairs independent cols x rows problems (BASELINE config 5; the reference handles one pair per process,
 * serial_smithW.c:141-145 per pair).  Runs on the one-pair-per-wave kernel (csrc/sw_batch.hip) when the batch has at most 8
 * distinct letters and the scores fit a signed byte -- the letter count is read back once per call, the one host round trip --
 * and on the single-pair machinery otherwise; results are identical.  Pair k reads a at
 * d_a + k*a_stride and b at d_b + k*b_stride (b_stride a multiple of 16), writes d_results[k] (exact arg-max in
 * every mode) and, where given, its matrices at element offset k*(rows+1)*(cols+1).  d_H and/or d_P may be NULL. */
int sw_batch_device(sw_ctx* ctx, const char* d_a, int64_t a_stride, int64_t cols, const char* d_b, int64_t b_stride,
                    int64_t rows, int64_t npairs, const sw_scores* scores, int32_t* d_H, int32_t* d_P,
                    sw_result* d_results, void* stream);
/* the same with a compact predecessor matrix: p_elem_bytes 1 -> d_P is int8_t* (100 000 x 1025^2 codes = 105 GB) */
int sw_batch_device_ex(sw_ctx* ctx, const char* d_a, int64_t a_stride, int64_t cols, const char* d_b, int64_t b_stride,
                       int64_t rows, int64_t npairs, const sw_scores* scores, int32_t* d_H, void* d_P, int p_elem_bytes,
                       sw_result* d_results, void* stream);
/* backtrack() (serial_smithW.c:262-277) of every pair of a batch: walks pair k's P from d_results[k].max_pos, negates
 * the path, sets d_results[k].path_len; d_paths (optional, npairs x path_cap) receives the visited pair-local indices. */
int sw_batch_traceback_device(sw_ctx* ctx, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t npairs,
                              int64_t* d_paths, int64_t path_cap, sw_result* d_results, void* stream);

/* Host-buffer convenience wrapper around sw_fill_device (alloc, H2D, fill, D2H, sync).
 * H, P: caller-owned int32 (rows+1)*(cols+1); either may be NULL to skip its copy-out. */
int sw_fill_host(sw_ctx* ctx, const char* a, int64_t cols, const char* b, int64_t rows,
                 const sw_scores* scores, int32_t* H, int32_t* P, sw_result* result);

/* Adaptive dispatch (the reference's per-diagonal choice of serial / OpenMP / offload,
 * omp_smithW-v7-adaptive.cpp:304-396): tiny problems are filled on the host by sw_fill_cpu (the reference recurrence,
 * serial_smithW.c:141-145,187-244), everything else on ctx's GPU (ctx may be NULL: host only); the traceback runs on
 * the host P either way, so H, P, max_pos, max_score and path_len come back exactly as serial_smithW leaves them.
 * used_gpu (optional) reports the choice. */
int sw_align_auto(sw_ctx* ctx, const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores,
                  int32_t* H, int32_t* P, sw_result* result, int* used_gpu);
int sw_fill_cpu(const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores, int32_t* H, int32_t* P,
                sw_result* result);
/* The same with the reference's THREE executors (omp_smithW-v7-adaptive.cpp:304-396: serial / OpenMP / offload per diagonal):
 * executor 0 = host fill, 1 = one GPU (ctx if given, else a context made on devices[0]), 2 = row bands over all `ndev` devices
 * (sw_multi_*; an id may repeat).  N GPUs are used from multi_min_cells cells on (0 = 4e9, about 65536^2: below that a single pair
 * is bound by its strip chain, which more GPUs do not shorten) or when H + P do not fit devices[0].  Same outputs as sw_align_auto. */
int sw_align_auto_multi(sw_ctx* ctx, const int* devices, int ndev, const char* a, int64_t cols, const char* b, int64_t rows,
                        const sw_scores* scores, int32_t* H, int32_t* P, sw_result* result, int* executor, int64_t multi_min_cells);

/* ---- traceback: replaces backtrack(), serial_smithW.c:262-277.  Negates P along the path.  One wave walks 64 x 64 windows of P
 * held in registers (csrc/sw_traceback.hip): ~35 ns per step instead of a memory latency.
 * d_path (optional) receives the visited linear indices (capacity path_cap);
 * d_result->path_len is set.  P[max_pos]==NONE (UB in the reference) == empty path. */
int sw_traceback_device(sw_ctx* ctx, int32_t* d_P, int64_t cols, int64_t rows, int64_t max_pos,
                        int64_t* d_path, int64_t path_cap, sw_result* d_result, void* stream);
int sw_traceback_host(int32_t* P, int64_t cols, int64_t rows, int64_t max_pos,
                      int64_t* path, int64_t path_cap, int64_t* path_len);
/* the same on an int32 (p_elem_bytes 4) or compact int8 (1) predecessor matrix */
int sw_traceback_device_ex(sw_ctx* ctx, void* d_P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos,
                           int64_t* d_path, int64_t path_cap, sw_result* d_result, void* stream);
int sw_traceback_host_ex(void* P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos,
                         int64_t* path, int64_t path_cap, int64_t* path_len);

/* Compact predecessor matrix back to the reference's layout (a compact P "must still round-trip to the
 * int32 H/P layout"): d_P32[k] = (int32_t)d_P8[k] for k < count -- codes 0..3, and -1..-3 along a traced path.  The two
 * buffers must not overlap. */
int sw_p8_to_p32_device(sw_ctx* ctx, const void* d_P8, int32_t* d_P32, int64_t count, void* stream);

/* 2-bit predecessor matrix (the reference's own "keep less" variant is the rolling-buffer fill,
 * rotated-cuda/sw-rotated-omp.cc:214-224): the codes 0..3 of serial_smithW.c:23-27 need two bits, so 262144^2 predecessors take
 * 17 GB instead of 69 GB (int8) or 275 GB (the reference's int32).  Layout: 4 cells per byte, cell k (row-major linear index, as in
 * P) in bits 2*(k&3)..2*(k&3)+1 of byte k>>2.  Two bits cannot hold the sign backtrack() leaves on a path (P[pos] *= PATH,
 * serial_smithW.c:271): a traced path is a bitmap beside the matrix, bit k&31 of 32-bit word k>>5.
 *   SW_P2_BYTES(count) / SW_PATHBITS_BYTES(count): buffer sizes for `count` cells (whole 32-cell groups)
 *   sw_p_to_p2_device   packs an int8 (p_elem_bytes 1) or int32 (4) matrix; negative (traced) cells set their bit in d_pathbits
 *                       (optional, may be NULL; every word of it is written)
 *   sw_p2_to_p32_device the round trip to the reference layout: d_P32[k] = code, negated where d_pathbits (optional) marks it
 *   sw_traceback_p2_device  backtrack() on the packed matrix: marks the path in d_pathbits (optional; zeroed by the caller),
 *                       writes the visited indices to d_path (optional) and d_result->path_len -- same walk, same kernel family
 *                       as sw_traceback_device */
#define SW_P2_BYTES(count) ((((size_t)(count) + 31) / 32) * 8)
#define SW_PATHBITS_BYTES(count) ((((size_t)(count) + 31) / 32) * 4)
int sw_p_to_p2_device(sw_ctx* ctx, const void* d_P, int p_elem_bytes, void* d_P2, uint32_t* d_pathbits, int64_t count, void* stream);
int sw_p2_to_p32_device(sw_ctx* ctx, const void* d_P2, const uint32_t* d_pathbits, int32_t* d_P32, int64_t count, void* stream);
int sw_traceback_p2_device(sw_ctx* ctx, const void* d_P2, int64_t cols, int64_t rows, int64_t max_pos, uint32_t* d_pathbits,
                           int64_t* d_path, int64_t path_cap, sw_result* d_result, void* stream);

/* ---- verification helpers (not on the timed path) ------------------------------------------
 * Position-weighted row checksums of a device matrix with (rows1 x m) elements:
 *   cs[i] = sum_j (uint64)(uint32)X[i][j] * ((j+1) * 0x9E3779B97F4A7C15)  (mod 2^64)
 * elem_bytes 1, 4 or 8 (1: int8 values are sign-extended first, so a compact P checksums like its int32
 * widening; 8: low 32 bits are summed and the high half must be the sign extension, otherwise cs[i] is
 * forced to ~0). d_cs: rows1 uint64 on the device. */
int sw_row_checksums_device(sw_ctx* ctx, const void* d_X, int elem_bytes, int64_t rows1, int64_t m,
                            uint64_t* d_cs, void* stream);

/* ---- output buffers placed for speed -------------------------------------------------------
 * Physical HBM falls into a few coarse classes (regions of tens of GiB; three on MI355X), and two store streams into the SAME class
 * run ~1.4x slower than into different ones.  A fill stores H[r][c] and P[r][c] together: a 16384^2 fill takes 0.79 ms with H and P
 * in different classes and 1.05 ms with both in one -- the usual outcome of two back-to-back hipMallocs.  sw_alloc_outputs places the
 * pair:
 *   trials <= 0  (default) candidates for P -- three plain ones, then behind temporary spacer allocations of 8-160 GiB (taken only while
 *                8 GiB of head room remain, released at once) -- are classified against H with a two-stream store probe
 *                (csrc/sw_place.hip; ~0.3 ms per candidate, no fill of the caller's problem, d_a / d_b not needed); the first one in
 *                another class is kept (matrices of many GiB span classes themselves: 8 sample windows, the best of five candidates).
 *                Below 512 MiB of output a plain pair (such a fill is not bound by its stores).  The search runs against a time budget,
 *                option "placement_budget_ms" (default 1500): fresh memory costs 0.3 ms per allocation and the call takes 2-5 ms, but
 *                memory that was in use before is wiped by the driver (~30 GiB/s) before it is handed out again, nothing tells
 *                beforehand, and a spacer can then cost seconds -- so a spacer is tried only while its WORST case (40 ms per GiB)
 *                still fits the budget (the default admits spacers up to 37 GiB), and when the budget is spent the best
 *                candidate seen is handed out (sw_get_option "last_placement_ratio_x1000": ~1300-1450 = different classes,
 *                ~2000 = one class).  A caller that fills many times into the pair raises the budget (bench.py: 20 s).
 *   trials == 1  a plain pair.
 *   trials  > 1  round 3's search: up to `trials` candidates, three fills of the caller's problem into each on the DEFAULT stream
 *                (d_a / d_b must be ready), the fastest kept.
 * trial_ms (optional; max(trials, 16) floats) receives per candidate the time of the probe (or of a trial fill), 0 for those not
 * needed.  The contents of the returned buffers are undefined.  Release with sw_free_outputs (d_P may sit inside a larger allocation).
 * (sw_multi_create and sw_fill_host place their matrices the same way; results never depend on placement.) */
int sw_alloc_outputs(sw_ctx* ctx, const char* d_a, int64_t cols, const char* d_b, int64_t rows, const sw_scores* scores,
                     int h_elem_bytes, int p_elem_bytes, int trials, void** d_H, void** d_P, float* trial_ms);
int sw_free_outputs(sw_ctx* ctx, void* d_H, void* d_P);
/* The classifier of sw_alloc_outputs for buffers the caller allocated itself: how much slower two store streams into (d_X, d_Y) run
 * together than one stream into d_X alone -- *ratio ~1.3-1.45: the two lie in different classes of the HBM, ~2.0: in the same one (a fill
 * that writes H into one and P into the other is then ~1.3x slower).  Samples windows of up to 128 MiB (8 of them for buffers of more
 * than 6 GiB) and WRITES them: call it before the buffers hold anything.  Runs on the current device, default stream, ~0.2 ms per window;
 * ms_together (optional): the time of the two-stream probe. */
int sw_place_pair_ratio(void* d_X, size_t xbytes, void* d_Y, size_t ybytes, float* ratio, float* ms_together);

/* ---- device memory plumbing for hosts without a HIP binding (cgo / JNI / ctypes callers) --- */
int sw_device_malloc(sw_ctx* ctx, size_t bytes, void** d_ptr);
int sw_device_free(sw_ctx* ctx, void* d_ptr);
int sw_memcpy_h2d(sw_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int sw_memcpy_d2h(sw_ctx* ctx, void* dst, const void* d_src, size_t bytes);
int sw_synchronize(sw_ctx* ctx, void* stream);   /* waits for `stream`; reports kernel faults */

/* ---- tuning knobs (0 = built-in default) --------------------------------------------------
 * None of them changes a result bit; they only move time.
 *   "engine"            0 systolic (default), 1 strip_scan
 *   "strips_per_group"  systolic: strips (producer waves) per workgroup, 1 or 2 (0: 1 for a single pair with up to
 *                       4.5 strips per CU or a batch that fits the CUs at once, else 2)
 *   "consumers"         systolic: consumer waves per strip: 2, 3, 4; also 6, 7 with one strip per group (0: with one strip per
 *                       group 4 up to ~3.5e8 cells and 6 above, with two strips 4; 8 is taken as 7).  The two-columns-per-lane
 *                       kernel (whole-matrix or band fills of one pair, any number of rows; csrc/sw_systolic2.inc) takes 4..7 from
 *                       this option (0: 7 behind scout workgroups, else 5 up to ~3.5e8 cells, 6 above) and runs 9 minus that many
 *                       importer waves
 *   "store_policy"      systolic H/P stores: 0 by problem size and strip geometry (streaming up to 6e8 cells and wherever whole lines are stored),
 *                       1 write-back, 2 streaming (nt)
 *   "importers"         systolic, one strip per workgroup: waves polling the left neighbour's edge column, besides the one that
 *                       always does (0: 4 up to ~3.5e8 cells, 2 above; at most what 12 waves per workgroup leave)
 *   "xcd_order"         systolic: 1 = neighbouring strip groups run on the same XCD
 *   "pace_ps"           systolic: strip 0 releases one row per this many picoseconds (0 = unpaced)
 *   "xcd_chain"         two-column kernel without scout workgroups: the strips of a pass dealt per XCD, edge columns through that XCD's
 *                       L2 (0: from 384 strips on, 1 on, 2 off)
 *   "filler_hop_ps", "filler_tau_ps", "filler_bw_gbs"   two-column kernel behind scout workgroups: pacing of the workgroups that
 *                       write H / P -- estimate of a strip hand-off (2400000 ps; 0 = no pacing), time per row of
 *                       an unhindered strip (25000 ps), store bandwidth the strips share (4200 GB/s)
 *   "band_wait_ms"      sw_fill_band_device: how long a strip waits for its halo granules before the launch aborts
 *                       with SW_ETIMEOUT (default 20000)
 *   "s2w"               two-column kernel: strips every 126 columns, or every 110 -- 126 wide, overlapping by 16 columns, so that every
 *                       64-byte line of a matrix row lies inside one strip and is stored whole, by one instruction;
 *                       0: the library chooses (110, with streaming stores, for matrices with H written and P int32, absent, or int8 beside an int32 H -- even
 *                       widths but for int32 H + int32 P -- that
 *                       are too wide for scout workgroups: more than ~21 500 columns, 18 700 with an int64 H), 126 / 110 force one
 *   "split_blk", "split_from"   two-column kernel behind scouts: from strip `split_from` on, the strip's scout writes the matrix
 *                       blocks from `split_blk` on itself (0: the library chooses)
 *   "probe_foreign_pairs"  1: an H / P pair the library did not allocate is probed once, at its first fill (big int32 fills behind scouts only): the
 *                       probe WRITES both buffers -- the fill overwrites them anyway -- and synchronises the stream (~0.3 ms); a pair found to
 *                       lie in one class of the HBM is then filled with overlapping strips (16384^2: ~310 instead of ~235 GCUPS).  Default 0
 *                       (pairs from sw_alloc_outputs are known without it)
 *   "placement_hold_gib"  sw_alloc_outputs: where no candidate pair lies in two classes of the HBM (the usual case once most of a device's memory
 *                       has been in use: the driver then hands out the little clean memory it has, all of one class), P may be allocated with up
 *                       to this many GiB of slack and slid inside its own allocation to where the probe is good; the slack stays allocated
 *                       while the pair lives (sw_get_option "last_placement_held_gib").  Default 0; pairs of many GiB take up to 32 by themselves
 *   "placement_budget_ms"  sw_alloc_outputs: how long the search for an H / P pair in different classes of the HBM may take at worst (default 1500; 2-5 ms on memory that needs no wiping)
 *   "max_blocks"        cap of the resident grid (0 = all CUs); concurrent band launches partition the CUs with it
 *   "waves_per_block", "debug_flags", "debug_buf", "batch_lds"   development aids (debug_flags 131072: no scout workgroups,
 *                       8388608: scouts without the per-XCD dealing of the roles, 134217728: no pacing, 65536: batches on the
 *                       single-pair machinery, 16384: one column per lane)
 * sw_get_option also answers "last_grid", "last_strips", "last_strips2" (strips of the two-column kernel), "last_scouts"
 * (scout workgroups of the last fill), "last_xcd_mode" (1: that fill dealt its roles per XCD), "xcd_round_robin" (1: sw_create saw
 * workgroup i of a launch on XCD i % 8) and "last_batch_kernel" (1: the last batch ran one pair per wave). */
int sw_set_option(sw_ctx* ctx, const char* name, int64_t value);
int64_t sw_get_option(sw_ctx* ctx, const char* name);

#ifdef __cplusplus
}
#endif
#endif
