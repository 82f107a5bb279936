/*
 * oracle/sw_oracle.c -- CPU restatement of the reference Smith-Waterman hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sw_oracle.h).  Parity status: PINNED against the real
 * reference (oracle/_ref, built from /root/reference/serial_smithW.c) and the reference's
 * built-in known-answer test; see tests/test_oracle.py and tests/golden/.
 *
 * Plain C99; builds with `gcc -O2 -fopenmp` (OpenMP optional: only the wavefront fill uses it).
 * Citations are relative to /root/reference.
 */
#include "sw_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- glibc rand(), TYPE_3 additive feedback generator (the reference never calls srand in
 * serial_smithW.c, i.e. seed 1; omp_smithW.c:491 seeds with time()). -------------------- */
void swo_srand(swo_rng* g, uint32_t seed) {
    int32_t word = seed ? (int32_t)seed : 1;
    g->r[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        /* Schrage: 16807 * word mod 2147483647 without overflow */
        int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        g->r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) g->r[i] = g->r[i - 31];
    g->k = 0; /* r[] now holds outputs 0..33 of the lag sequence; use as a ring of 34 */
    /* ring position p holds o_(n) with n == p (mod 34); next index to produce is 34 */
    g->k = 34;
    for (int i = 34; i < 344; i++) {
        uint32_t v = g->r[(g->k - 31) % 34] + g->r[(g->k - 3) % 34];
        g->r[g->k % 34] = v;
        g->k++;
    }
}

int32_t swo_rand(swo_rng* g) {
    uint32_t v = g->r[(g->k - 31) % 34] + g->r[(g->k - 3) % 34];
    g->r[g->k % 34] = v;
    g->k++;
    if (g->k >= 34 * 1000000) g->k -= 34 * 999999; /* keep k bounded, same residue mod 34 */
    return (int32_t)(v >> 1);
}

static char swo_letter(int aux) {
    /* serial_smithW.c:338-346: 0->A, 2->C, 3->G, else T */
    if (aux == 0) return 'A';
    if (aux == 2) return 'C';
    if (aux == 3) return 'G';
    return 'T';
}

void swo_generate(int64_t cols, int64_t rows, uint32_t seed, char* a, char* b) {
    swo_rng g;
    swo_srand(&g, seed);
    /* m and n were already incremented when generate() runs (serial_smithW.c:91-92,129) */
    for (int64_t i = 0; i < cols + 1; i++) a[i] = swo_letter(swo_rand(&g) % 4);
    for (int64_t i = 0; i < rows + 1; i++) b[i] = swo_letter(swo_rand(&g) % 4);
}

/* ---- wavefront indexing, omp_smithW.c:260-291 ---------------------------------------- */
int64_t swo_nelement(int64_t i, int64_t m, int64_t n) {
    int64_t mx = m > n ? m : n, mn = m < n ? m : n;
    if (i < m && i < n) return i;
    if (i < mx) return mn - 1;
    int64_t d = m - n; if (d < 0) d = -d;
    return 2 * mn - i + d - 2;
}

void swo_first_diag_element(int64_t i, int64_t m, int64_t n, int64_t* si, int64_t* sj) {
    (void)m;
    if (i < n) { *si = i; *sj = 1; }
    else       { *si = n - 1; *sj = i - n + 2; }
}

/* ---- the cell kernel, serial_smithW.c:187-244 ------------------------------------------ */
void swo_similarity_score(int64_t i, int64_t j, int64_t m, const char* a, const char* b,
                          const swo_scores* sc, int32_t* H, int32_t* P, int64_t* maxPos) {
    int64_t index = m * i + j;
    int32_t up   = H[index - m] + sc->gap;
    int32_t left = H[index - 1] + sc->gap;
    /* matchMissmatchScore, serial_smithW.c:251-256 */
    int32_t diag = H[index - m - 1] + ((a[j - 1] == b[i - 1]) ? sc->match : sc->mismatch);
    int32_t max = SWO_NONE, pred = SWO_NONE;
    if (diag > max) { max = diag; pred = SWO_DIAGONAL; }
    if (up > max)   { max = up;   pred = SWO_UP; }
    if (left > max) { max = left; pred = SWO_LEFT; }
    H[index] = max;
    P[index] = pred;
    if (maxPos && max > H[*maxPos]) *maxPos = index;
}

int64_t swo_fill_rowmajor(const char* a, int64_t cols, const char* b, int64_t rows,
                          const swo_scores* sc, int32_t* H, int32_t* P) {
    int64_t m = cols + 1, n = rows + 1, maxPos = 0;
    for (int64_t i = 1; i < n; i++)
        for (int64_t j = 1; j < m; j++)
            swo_similarity_score(i, j, m, a, b, sc, H, P, &maxPos);
    return maxPos;
}

int64_t swo_fill_wavefront(const char* a, int64_t cols, const char* b, int64_t rows,
                           const swo_scores* sc, int32_t* H, int32_t* P, int nthreads) {
    int64_t m = cols + 1, n = rows + 1;
    int64_t nDiag = m + n - 3; /* omp_smithW.c:182 */
    (void)nthreads;
    for (int64_t i = 1; i <= nDiag; ++i) {
        int64_t nEle = swo_nelement(i, m, n), si, sj;
        swo_first_diag_element(i, m, n, &si, &sj);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) if (nthreads > 1 && nEle >= 1024)
#endif
        for (int64_t j = 0; j < nEle; ++j)
            swo_similarity_score(si - j, sj + j, m, a, b, sc, H, P, NULL);
    }
    /* deterministic arg-max: lowest linear index among cells holding the max, 0 if all zero
     * (== what the serial scan of serial_smithW.c:240-242 yields) */
    int64_t maxPos = 0; int32_t best = 0;
    for (int64_t idx = 0; idx < m * n; idx++)
        if (H[idx] > best) { best = H[idx]; maxPos = idx; }
    return maxPos;
}

/* ---- backtrack, serial_smithW.c:262-277 -------------------------------------------------- */
int64_t swo_backtrack(int32_t* P, int64_t m, int64_t maxPos, int64_t* path, int64_t path_cap) {
    int64_t len = 0;
    if (P[maxPos] == SWO_NONE) return 0; /* UB in the reference; defined as empty path */
    do {
        int64_t predPos;
        if (P[maxPos] == SWO_DIAGONAL) predPos = maxPos - m - 1;
        else if (P[maxPos] == SWO_UP)  predPos = maxPos - m;
        else                           predPos = maxPos - 1; /* LEFT */
        P[maxPos] *= SWO_PATH;
        if (path && len < path_cap) path[len] = maxPos;
        len++;
        maxPos = predPos;
    } while (P[maxPos] != SWO_NONE);
    return len;
}

uint64_t swo_fnv1a64(const void* data, size_t nbytes) {
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 1099511628211ULL; }
    return h;
}

#define SWO_CS_MUL 0x9E3779B97F4A7C15ULL

void swo_row_checksums(const int32_t* X, int64_t rows1, int64_t m, uint64_t* cs) {
    for (int64_t i = 0; i < rows1; i++) {
        uint64_t s = 0;
        const int32_t* row = X + i * m;
        for (int64_t j = 0; j < m; j++) s += (uint64_t)(uint32_t)row[j] * ((uint64_t)(j + 1) * SWO_CS_MUL);
        cs[i] = s;
    }
}

int64_t swo_fill_streaming(const char* a, int64_t cols, const char* b, int64_t rows,
                           const swo_scores* sc, uint64_t* csH, uint64_t* csP,
                           int32_t* max_score, int32_t* bottom_row) {
    int64_t m = cols + 1;
    int32_t* prev = (int32_t*)calloc((size_t)m, sizeof(int32_t));
    int32_t* cur  = (int32_t*)calloc((size_t)m, sizeof(int32_t));
    int64_t maxPos = 0; int32_t best = 0;
    if (csH) csH[0] = 0;
    if (csP) csP[0] = 0;
    for (int64_t i = 1; i <= rows; i++) {
        uint64_t sh = 0, sp = 0;
        char bi = b[i - 1];
        cur[0] = 0;
        for (int64_t j = 1; j < m; j++) {
            int32_t up = prev[j] + sc->gap, left = cur[j - 1] + sc->gap;
            int32_t diag = prev[j - 1] + ((a[j - 1] == bi) ? sc->match : sc->mismatch);
            int32_t max = 0, pred = 0;
            if (diag > max) { max = diag; pred = SWO_DIAGONAL; }
            if (up > max)   { max = up;   pred = SWO_UP; }
            if (left > max) { max = left; pred = SWO_LEFT; }
            cur[j] = max;
            uint64_t w = (uint64_t)(j + 1) * SWO_CS_MUL;
            sh += (uint64_t)(uint32_t)max * w;
            sp += (uint64_t)(uint32_t)pred * w;
            if (max > best) { best = max; maxPos = m * i + j; }
        }
        if (csH) csH[i] = sh;
        if (csP) csP[i] = sp;
        int32_t* t = prev; prev = cur; cur = t;
    }
    if (bottom_row) memcpy(bottom_row, prev, (size_t)m * sizeof(int32_t));
    if (max_score) *max_score = best;
    free(prev); free(cur);
    return maxPos;
}

/* ---- whole-matrix digests of problems too big to materialise (round 4) --------------------
 * swo_fill_streaming + H-row checkpoints every `every` rows (ckpt: (rows/every + 1) x m int32, row k = H row k*every),
 * from which swo_path_from_ckpt() re-derives P block by block (same cell rule, serial_smithW.c:192-234) and walks it
 * exactly like backtrack() (serial_smithW.c:262-277).  Emits the path (walk order), and the per-row checksum DELTAS the
 * negation adds to csP, so a test can compare the traced-back matrix too. */
int64_t swo_fill_streaming_ckpt(const char* a, int64_t cols, const char* b, int64_t rows,
                                const swo_scores* sc, uint64_t* csH, uint64_t* csP,
                                int32_t* max_score, int64_t every, int32_t* ckpt,
                                int64_t band_rows, int32_t* band_best, int64_t* band_pos) {
    int64_t m = cols + 1;
    int32_t* prev = (int32_t*)calloc((size_t)m, sizeof(int32_t));
    int32_t* cur  = (int32_t*)calloc((size_t)m, sizeof(int32_t));
    int64_t maxPos = 0; int32_t best = 0;
    if (csH) csH[0] = 0;
    if (csP) csP[0] = 0;
    if (ckpt) memset(ckpt, 0, (size_t)m * sizeof(int32_t));
    for (int64_t i = 1; i <= rows; i++) {
        uint64_t sh = 0, sp = 0;
        char bi = b[i - 1];
        int32_t rbest = 0; int64_t rpos = 0;     /* first maximum of this row (bands: arg-max of rows lo+1..hi alone) */
        cur[0] = 0;
        for (int64_t j = 1; j < m; j++) {
            int32_t up = prev[j] + sc->gap, left = cur[j - 1] + sc->gap;
            int32_t diag = prev[j - 1] + ((a[j - 1] == bi) ? sc->match : sc->mismatch);
            int32_t max = 0, pred = 0;
            if (diag > max) { max = diag; pred = SWO_DIAGONAL; }
            if (up > max)   { max = up;   pred = SWO_UP; }
            if (left > max) { max = left; pred = SWO_LEFT; }
            cur[j] = max;
            uint64_t w = (uint64_t)(j + 1) * SWO_CS_MUL;
            sh += (uint64_t)(uint32_t)max * w;
            sp += (uint64_t)(uint32_t)pred * w;
            if (max > rbest) { rbest = max; rpos = m * i + j; }
        }
        if (rbest > best) { best = rbest; maxPos = rpos; }
        if (band_best && band_rows > 0) {
            int64_t k = (i - 1) / band_rows;
            if ((i - 1) % band_rows == 0) { band_best[k] = 0; band_pos[k] = 0; }
            if (rbest > band_best[k]) { band_best[k] = rbest; band_pos[k] = rpos; }
        }
        if (csH) csH[i] = sh;
        if (csP) csP[i] = sp;
        if (ckpt && i % every == 0) memcpy(ckpt + (i / every) * m, cur, (size_t)m * sizeof(int32_t));
        int32_t* t = prev; prev = cur; cur = t;
    }
    if (max_score) *max_score = best;
    free(prev); free(cur);
    return maxPos;
}

int64_t swo_path_from_ckpt(const char* a, int64_t cols, const char* b, int64_t rows,
                           const swo_scores* sc, int64_t every, const int32_t* ckpt, int64_t maxPos,
                           int64_t* path, int64_t path_cap, uint64_t* csP_delta) {
    int64_t m = cols + 1, len = 0;
    int64_t i = maxPos / m, j = maxPos % m;
    (void)rows;
    int8_t* Pb = (int8_t*)malloc((size_t)every * (size_t)m);
    int32_t* prev = (int32_t*)malloc((size_t)m * sizeof(int32_t));
    int32_t* cur  = (int32_t*)malloc((size_t)m * sizeof(int32_t));
    int done = 0;
    if (i == 0 || j == 0) done = 1;     /* P[maxPos] == NONE: empty path, as swo_backtrack */
    while (!done) {
        /* block of rows r0+1 .. r1 that holds row i; only columns 0..j can be visited from here on */
        int64_t k = (i - 1) / every, r0 = k * every, r1 = i, w = j + 1;
        memcpy(prev, ckpt + k * m, (size_t)w * sizeof(int32_t));
        for (int64_t r = r0 + 1; r <= r1; r++) {
            char bi = b[r - 1];
            int8_t* prow = Pb + (r - r0 - 1) * m;
            cur[0] = 0; prow[0] = 0;
            for (int64_t c = 1; c < w; c++) {
                int32_t up = prev[c] + sc->gap, left = cur[c - 1] + sc->gap;
                int32_t diag = prev[c - 1] + ((a[c - 1] == bi) ? sc->match : sc->mismatch);
                int32_t max = 0, pred = 0;
                if (diag > max) { max = diag; pred = SWO_DIAGONAL; }
                if (up > max)   { max = up;   pred = SWO_UP; }
                if (left > max) { max = left; pred = SWO_LEFT; }
                cur[c] = max; prow[c] = (int8_t)pred;
            }
            int32_t* t = prev; prev = cur; cur = t;
        }
        /* walk inside the block */
        while (i > r0) {
            int32_t p = Pb[(i - r0 - 1) * m + j];
            if (p == SWO_NONE) { done = 1; break; }
            if (path && len < path_cap) path[len] = i * m + j;
            if (csP_delta) {
                uint64_t wgt = (uint64_t)(j + 1) * SWO_CS_MUL;
                csP_delta[i] += (uint64_t)(uint32_t)(-p) * wgt - (uint64_t)(uint32_t)p * wgt;
            }
            len++;
            if (p == SWO_DIAGONAL) { i--; j--; }
            else if (p == SWO_UP)  { i--; }
            else                   { j--; }
            if (j == 0) { done = 1; break; }   /* column 0 is NONE everywhere */
        }
        if (i == 0) done = 1;                  /* row 0 is NONE everywhere */
    }
    free(Pb); free(prev); free(cur);
    return len;
}
