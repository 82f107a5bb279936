/*
 * oracle/sw_oracle.h -- CPU restatement of the reference Smith-Waterman hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may call it, and only as the checker / timed CPU baseline.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle.py)
 * against fixtures generated from the real reference (serial_smithW.c compiled
 * in place by oracle/Makefile -> oracle/_ref/) and against the reference's
 * built-in known-answer test (maxPos==69, H[maxPos]==13, H[89]==7).
 *
 * All citations are relative to /root/reference.
 */
#ifndef SW_ORACLE_H
#define SW_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* serial_smithW.c:23-27 */
#define SWO_PATH (-1)
#define SWO_NONE 0
#define SWO_UP 1
#define SWO_LEFT 2
#define SWO_DIAGONAL 3

typedef struct { int32_t match, mismatch, gap; } swo_scores; /* serial_smithW.c:59-61: 3,-3,-2 */

/* glibc rand() TYPE_3 restated (the reference calls libc rand(), serial_smithW.c:338,351). */
typedef struct { uint32_t r[34]; int k; } swo_rng;
void swo_srand(swo_rng* g, uint32_t seed);
int32_t swo_rand(swo_rng* g);

/* generate(): serial_smithW.c:334-361.  Runs after m++,n++ (serial_smithW.c:91-92), so it
 * draws cols+1 values for a, then rows+1 for b.  a must hold cols+1 bytes, b rows+1. */
void swo_generate(int64_t cols, int64_t rows, uint32_t seed, char* a, char* b);

/* nElement / calcFirstDiagElement: omp_smithW.c:260-275, 282-291 (m,n are the PADDED sizes). */
int64_t swo_nelement(int64_t i, int64_t m, int64_t n);
void swo_first_diag_element(int64_t i, int64_t m, int64_t n, int64_t* si, int64_t* sj);

/* similarityScore + matchMissmatchScore: serial_smithW.c:187-244, 251-256. m = cols+1. */
void swo_similarity_score(int64_t i, int64_t j, int64_t m, const char* a, const char* b,
                          const swo_scores* sc, int32_t* H, int32_t* P, int64_t* maxPos);

/* Row-major fill (serial_smithW.c:141-145). H,P: (rows+1)*(cols+1) zero-initialised by caller.
 * Returns maxPos. */
int64_t swo_fill_rowmajor(const char* a, int64_t cols, const char* b, int64_t rows,
                          const swo_scores* sc, int32_t* H, int32_t* P);

/* Anti-diagonal wavefront fill (omp_smithW.c:203-216) with the deterministic arg-max rule
 * (lowest linear index among maxima). nthreads<=1 -> serial. */
int64_t swo_fill_wavefront(const char* a, int64_t cols, const char* b, int64_t rows,
                           const swo_scores* sc, int32_t* H, int32_t* P, int nthreads);

/* backtrack(): serial_smithW.c:262-277. Negates P along the path. Returns the path length;
 * if path!=NULL stores the visited linear indices (capacity path_cap). P[maxPos]==NONE at
 * entry (UB in the reference) is defined as an empty path. */
int64_t swo_backtrack(int32_t* P, int64_t m, int64_t maxPos, int64_t* path, int64_t path_cap);

/* FNV-1a 64 over raw bytes (SURVEY.md App. C goldens). */
uint64_t swo_fnv1a64(const void* data, size_t nbytes);

/* Position-weighted row checksums used to verify matrices too big to copy out:
 *   cs[i] = sum_j (uint64)(uint32)X[i][j] * ((j+1) * 0x9E3779B97F4A7C15)   (mod 2^64) */
void swo_row_checksums(const int32_t* X, int64_t rows1, int64_t m, uint64_t* cs);

/* Streaming two-row fill: never materialises H/P.  Emits per-row checksums of H and P
 * (rows+1 entries each, row 0 = 0), maxPos/maxScore, and optionally the bottom row of H. */
int64_t swo_fill_streaming(const char* a, int64_t cols, const char* b, int64_t rows,
                           const swo_scores* sc, uint64_t* csH, uint64_t* csP,
                           int32_t* max_score, int32_t* bottom_row);

/* Round 4: whole-matrix digests for sizes whose P does not fit host memory.  The streaming fill above plus H-row
 * checkpoints every `every` rows (ckpt holds rows/every + 1 rows of m int32; row 0 = zeros), and a traceback that
 * re-derives P block by block from them (serial_smithW.c:262-277 on serial_smithW.c:192-234's P).  csP_delta[i]
 * (rows+1 entries, caller-zeroed) receives what negating the path adds to the row checksum of P.
 * band_best / band_pos (optional, ceil(rows/band_rows) entries): arg-max of every band of band_rows rows taken alone
 * (highest score, lowest linear index of the WHOLE matrix; 0 / 0 when the band is all zero). */
int64_t swo_fill_streaming_ckpt(const char* a, int64_t cols, const char* b, int64_t rows,
                                const swo_scores* sc, uint64_t* csH, uint64_t* csP,
                                int32_t* max_score, int64_t every, int32_t* ckpt,
                                int64_t band_rows, int32_t* band_best, int64_t* band_pos);
int64_t swo_path_from_ckpt(const char* a, int64_t cols, const char* b, int64_t rows,
                           const swo_scores* sc, int64_t every, const int32_t* ckpt, int64_t maxPos,
                           int64_t* path, int64_t path_cap, uint64_t* csP_delta);

#ifdef __cplusplus
}
#endif
#endif
