// sw_kernels.h -- shared declarations between the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/swhip.h"

namespace swk {

// packed arg-max key: (score << 40) | (2^40-1 - linear_index); atomicMax picks the highest
// score and, among equals, the LOWEST linear index (the serial scan's rule, serial_smithW.c:240-242)
constexpr unsigned long long SW_KEY_IDX_MASK = (1ull << 40) - 1;
constexpr int64_t SW_MAX_DIM = (1 << 20) - 1; // rows/cols limit: 20-bit row tags, 40-bit indices

struct FillParams {
    int64_t cols, rows, M;     // M = row stride of H/P in elements (cols + 1 for a whole matrix)
    void* H;                   // HT[(rows+1)*M]
    int32_t* P;                // int32[(rows+1)*M]
    const int32_t* top;        // optional halo row (cols+1), NULL = zeros
    const int32_t* left;       // systolic: optional halo column (rows+1 H values of the column left of the tile), NULL = zeros
    int32_t* right;            // systolic: optional output, H of the tile's last column (rows+1)
    int npairs;                // systolic: independent problems in this launch (batch), 1 otherwise
    int store_hp;              // (unused; H / P are written when their pointers are non-NULL)
    int64_t a_pstride, b_pstride, bpad_pstride, hp_pstride, edge_pstride;  // per-pair strides (elements)
    int32_t mm, xm, ngap;      // match-2*gap, mismatch-2*gap, -gap  (G-space constants)
    unsigned long long* edge;  // [nstrips][rows+1] {tag,value} granules
    unsigned int tag_base;     // epoch << 20
    unsigned long long* result_key;
    unsigned int* abort_flag;
    int phi_base;              // systolic: >= 0 -> fast producers, phi = phi_base - s; < 0 -> generic
    int bfront;                // systolic: index of b[0] inside bpad / bpad16
    const unsigned short* bpad16;
    const unsigned char* bpad8;          // sw_systolic2: zero-padded byte copy of b, same layout as bpad16 (b[0] at index bfront)
    unsigned long long* dbg;   // optional: per strip {start, end} s_memrealtime stamps of its producer (experiments)
    int p_bytes;               // bytes per P element: 4 (reference layout) or 1 (compact, systolic engine only)
    int store_nt;              // systolic: streaming (nt) H/P stores
    int xcd_order;             // systolic: neighbouring strip groups on one XCD
    int pace_ps;               // systolic: strip 0 releases one row per pace_ps picoseconds (0 = unpaced)
    int debug_flags;           // bit0: drop the H/P stores (timing experiments only)
    int nstrips;               // strip_scan: ceil(cols/64); systolic: ceil(cols/63)
    // band-resident launch (systolic; multi-GPU row bands): the halo row arrives / leaves as 8-byte
    // {tag, H value} granules, one per column, while the kernel runs
    const unsigned long long* top_gran;  // cols+1 granules of the row above (polled per strip), or NULL
    unsigned long long* bot_gran;        // cols+1 granules of the band's last row (written per strip), or NULL
    unsigned int* bot_done;              // optional [nstrips]: set to bot_tag (system scope, after a release) once the strip's bottom granules are written
    unsigned int top_tag, bot_tag;
    unsigned int top_wait_ticks;         // patience of the top-halo poll in 100 MHz ticks / 2^10
    // "perm" producer (systolic; alphabets of at most 7 letters, scores that fit a signed byte): the substitution
    // scores of 16 steps come from 4 v_perm_b32 over a per-lane profile, every G value carries `gbias` (8-bit launch
    // tag << 24 | 2^16), and lane 63 stores its own results as self-validating 4-byte values
    const unsigned char* bcode;          // padded copy of b as letter codes 0..6 (7 = outside the sequence), layout as bpad
    const unsigned char* atab;           // [256] letter code of every byte value; [256..259] = number of letters (uint32)
    unsigned int* edge4;                 // [nstrips][e4stride] lane-63 values indexed by the producer's local step - 1
    int64_t e4stride, edge4_pstride;
    unsigned int gbias;                  // 0 = perm path not eligible on the host side (score range)
    int skip_if_perm;                    // sw_systolic: leave when the perm path applies (sw_systolic2 was launched for that case)
    int h_bytes;                         // sw_systolic2: bytes per H element (4 or 8); sw_systolic carries it as a template parameter
    int scout_double;                    // sw_systolic2: the first scout_double scout workgroups carry two strips, the others one
    int nscout;                          // sw_systolic2: workgroups 0..nscout-1 only run the chain (two strips each) and leave the edge columns to the others
    int filler_hop_ps, filler_tau_ps, filler_bw_gbs;    // sw_systolic2 behind scouts: pacing of the fillers (sw_systolic2.inc), 0 = none
    int xcd_mode;                        // sw_systolic2: 256 workgroups, roles dealt per XCD (workgroup i on XCD i % 8; nscout = scout workgroups in all)
    // one launch per fill (round 4): the preparation and the finalisation of a fill are the prologue / epilogue of sw_systolic2 itself
    unsigned int* sync;                  // per context, all zero between launches: [0] arrivals at the prologue's grid barrier, [1] workgroups that have
                                         // left, [2] the letter count the prologue found (kept for the fall-back kernel), [8..15] presence map of the byte values
    unsigned char* priv;                 // per workgroup: a zero-padded copy of b and its letter codes ([bpad8 | bcode], bpad_pstride bytes each)
    int64_t priv_stride;                 // bytes between two workgroups' copies
    unsigned short* bpad16_w;            // writable views of the shared padded copies (the prologue of workgroup 0 fills them when the fall-back kernel has to run)
    unsigned char* bpad8_w; unsigned char* bcode_w; unsigned char* atab_w;
    sw_result* result;                   // != NULL: the last workgroup to leave writes the result and re-arms key / abort flag / sync for the next launch
    int skip_row0;                       // prologue: row 0 of H / P is not this launch's to clear (a band's halo row)
    // column tiles of one matrix, one launch each (round 4: matrices too wide for scout workgroups beside one filler per strip)
    const int32_t* tile_left;            // H of the column left of this tile's first column, row r at tile_left[r * M] (the previous tile wrote it); NULL: zeros
    int64_t idx_off;                     // added to the linear index of the arg-max (the tile's first column: indices of the whole matrix)
    int final_launch;                    // the epilogue reports and re-arms the key (earlier tiles only accumulate into it)
    const unsigned char* alpha_a;        // what the prologue scans for letters: the WHOLE a (every tile must decide alike who fills)
    int64_t alpha_cols;
    // split strips (round 4, xcd_mode 1): from strip split_from on, the filler of a strip writes only the 16-row blocks below split_blk and
    // the strip's SCOUT -- a workgroup with rings and consumers like a filler -- writes the rest (sw_systolic2.inc); 0 = off
    int split_blk, split_from;
    int split_extra;                     // 1: the last strip has a scout workgroup too (it writes that strip's lower blocks; nobody reads its edge)
    int filler_end_steps;                // pacing: the steps of the filler that ends last (a split one), 0 = every filler runs all steps
    int filler_full_steps;               // pacing: steps of a whole strip (what the bandwidth bound is computed with)
    // strip geometry of sw_systolic2: strip s spans the columns s2w * s + 1 .. s2w * s + 126 (63 lanes x 2); s2w = 126: the strips tile the
    // matrix; s2w = 110: neighbouring strips OVERLAP by 16 columns, so that every 64-byte line of a matrix row lies wholly inside some
    // strip and can be stored by ONE instruction of that strip (whole-line stores: sw_systolic2.inc).  0 is taken as 126.
    int s2w;
};
constexpr int SW_XTAB_OFF = 448;         // atab + 448: unsigned int[256], XCD + 1 of every workgroup of the running sw_systolic2 launch (0: not there yet)
constexpr int SW_PERM_PAD = -100;        // score of any cell outside the sequences (perm producer)

// sw_batch.hip: one pair per wave (BASELINE config 5)
struct BatchParams {
    const unsigned char* a; int64_t a_pstride, cols;
    const unsigned char* bcode; int64_t bcode_pstride; int bfront;   // padded letter codes of b (sw_batch_codes)
    int64_t rows, npairs;
    const unsigned char* atab;       // letter code of every byte value; [256..259] = number of letters
    int32_t* H; void* P;             // either may be NULL; pair k at element offset k * hp_pstride
    int64_t hp_pstride;
    int match, mismatch, ngap;       // plain scores (H-space), ngap = -gap
    int* bnd; int64_t bnd_pstride;   // boundary column between strips (only when cols > 64 * C), ints per pair
    sw_result* results;
    int debug;                       // bit 0: drop the H / P stores (timing experiments only)
};
__global__ void sw_batch_codes(const unsigned char* b, int64_t rows, int64_t b_pstride, unsigned char* bcode, int64_t per, int front,
                               const unsigned int* part, int npart, unsigned char* atab, int64_t npairs);
template <int C, int PB>
__global__ void sw_batch_wave(BatchParams p);
template <bool LE4, bool K12, bool PB1>
__global__ void sw_batch_wave16(BatchParams p);   // two pairs per wave on packed 16-bit lanes (score + exact maxPos; PB1: int8 P too)

template <typename HT, int B>
__global__ void sw_strip_scan(const unsigned char* a, const unsigned char* b, FillParams p);
template <typename HT, int NS, int NC>
__global__ void sw_systolic(const unsigned char* a, const unsigned char* b, const unsigned char* bpad, FillParams p);
__global__ void sw_wipe_u32(unsigned int* buf, size_t n);
template <int NC, bool OV>
__global__ void sw_systolic2(const unsigned char* a, const unsigned char* b, FillParams p);
__global__ void sw_prep_scan(const unsigned char* a, int64_t cols, int64_t a_pstride, const unsigned char* b, int64_t rows, int64_t b_pstride, int64_t npairs,
                             unsigned int* part);
__global__ void sw_prep_code(const unsigned char* b, int64_t rows, int64_t front, int64_t b_pstride, unsigned char* bpad, unsigned short* bpad16,
                             unsigned char* bcode, const unsigned int* part, int npart, unsigned char* atab, int64_t per, int npad, void* H, int h_bytes,
                             void* P, int p_bytes, int64_t M, int64_t rows1, int skip_row0, unsigned long long* key);
__global__ void sw_xcc_probe(unsigned int* xcc_of_block);
__global__ void sw_prep_reduce(unsigned int* part, int npart);
__global__ void sw_finalize(const unsigned long long* key, const unsigned int* abort_flag, sw_result* res, int n);
template <typename PT>
__global__ void sw_traceback_wave(PT* P, int64_t M, int64_t rows1, int64_t pstride, int64_t start_pos, int64_t* paths, int64_t cap, sw_result* res,
                                  int64_t* stop, unsigned int* pathbits);
template <typename T> __global__ void sw_row_checksums(const T* X, int64_t m, unsigned long long* cs);
__global__ void sw_widen_p8(const signed char* P8, int32_t* P32, size_t n);
// 2-bit predecessor matrix (4 codes per byte) + path bitmap (1 bit per cell): sw_kernels.hip, sw_traceback.hip
struct P2Cells { unsigned char v; };     // tag type: sw_traceback_wave<P2Cells> walks a packed matrix
template <typename PT> __global__ void sw_pack_p2(const PT* P, unsigned char* P2, unsigned int* bits, size_t n);
__global__ void sw_unpack_p2(const unsigned char* P2, const unsigned int* bits, int32_t* P32, size_t n);

}  // namespace swk
