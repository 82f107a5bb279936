// sw_host.cpp -- host-only parts of the C-ABI (no GPU needed): the reference's input generator,
// its wavefront indexing helpers and the host traceback.  Citations: /root/reference paths.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/swhip.h"

namespace swh {
thread_local std::string g_err;
void set_err(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

// glibc random_r TYPE_3 (x^31 + x^3 + 1) as used by rand(): 31-word state seeded by the
// Lehmer generator 16807 mod (2^31-1), 310 warm-up steps, outputs are state sums >> 1.
// The reference draws from libc rand() (serial_smithW.c:338,351); restating it keeps the
// generated sequences identical on any host libc.
class GlibcRand {
public:
    explicit GlibcRand(uint32_t seed) {
        int64_t w = seed ? seed : 1;
        st_[0] = (uint32_t)w;
        for (int i = 1; i < 31; ++i) {
            w = (16807 * w) % 2147483647;      // same value as glibc's Schrage form
            st_[i] = (uint32_t)w;
        }
        f_ = 3; r_ = 0;
        for (int i = 0; i < 310; ++i) (void)next();
    }
    int32_t next() {
        st_[f_] += st_[r_];
        const uint32_t out = st_[f_] >> 1;
        f_ = (f_ + 1 == 31) ? 0 : f_ + 1;
        r_ = (r_ + 1 == 31) ? 0 : r_ + 1;
        return (int32_t)out;
    }
private:
    uint32_t st_[31];
    int f_, r_;
};

inline char letter(int v) {  // serial_smithW.c:339-346
    switch (v) { case 0: return 'A'; case 2: return 'C'; case 3: return 'G'; default: return 'T'; }
}
}  // namespace swh

template <typename PT>
static int64_t traceback_t(PT* P, int64_t m, int64_t max_pos, int64_t* path, int64_t path_cap) {  // serial_smithW.c:262-277
    int64_t len = 0, pos = max_pos;
    for (int pr = P[pos]; pr > 0; pr = P[pos]) {
        P[pos] = (PT)(pr * SW_PATH);
        if (path && len < path_cap) path[len] = pos;
        ++len;
        pos -= (pr == SW_DIAGONAL) ? m + 1 : (pr == SW_UP) ? m : 1;
    }
    return len;
}

extern "C" {

const char* sw_last_error(void) { return swh::g_err.c_str(); }
const char* sw_version(void) { return "swhip 0.1 (gfx950)"; }

int sw_generate(int64_t cols, int64_t rows, uint32_t seed, char* a, char* b) {
    if (cols < 0 || rows < 0 || !a || !b) { swh::set_err("sw_generate: bad argument"); return SW_EINVAL; }
    swh::GlibcRand rng(seed);
    // generate() runs after m++, n++ (serial_smithW.c:91-92,129): cols+1 draws, then rows+1
    for (int64_t i = 0; i <= cols; ++i) a[i] = swh::letter(rng.next() % 4);
    for (int64_t i = 0; i <= rows; ++i) b[i] = swh::letter(rng.next() % 4);
    return SW_OK;
}

int64_t sw_nelement(int64_t i, int64_t m, int64_t n) {  // omp_smithW.c:260-275
    const int64_t lo = m < n ? m : n, hi = m < n ? n : m;
    if (i < lo) return i;
    if (i < hi) return lo - 1;
    return 2 * lo - i + (hi - lo) - 2;
}

void sw_first_diag_element(int64_t i, int64_t m, int64_t n, int64_t* si, int64_t* sj) {  // omp_smithW.c:282-291
    (void)m;
    const bool left_edge = i < n;
    if (si) *si = left_edge ? i : n - 1;
    if (sj) *sj = left_edge ? 1 : i - n + 2;
}

int sw_traceback_host_ex(void* P, int p_elem_bytes, int64_t cols, int64_t rows, int64_t max_pos, int64_t* path,
                         int64_t path_cap, int64_t* path_len) {
    const int64_t m = cols + 1;
    if (!P || cols < 0 || rows < 0 || max_pos < 0 || max_pos >= m * (rows + 1) || (p_elem_bytes != 4 && p_elem_bytes != 1)) {
        swh::set_err("sw_traceback_host: bad argument");
        return SW_EINVAL;
    }
    const int64_t len = p_elem_bytes == 4 ? traceback_t((int32_t*)P, m, max_pos, path, path_cap)
                                          : traceback_t((signed char*)P, m, max_pos, path, path_cap);
    if (path_len) *path_len = len;
    return SW_OK;
}
int sw_traceback_host(int32_t* P, int64_t cols, int64_t rows, int64_t max_pos, int64_t* path,
                      int64_t path_cap, int64_t* path_len) {
    return sw_traceback_host_ex(P, 4, cols, rows, max_pos, path, path_cap, path_len);
}

// Host fill for problems too small to be worth a launch (sw_align_auto): the reference recurrence itself, row-major
// (serial_smithW.c:141-145, 187-244).  Product code -- the oracle under oracle/ is test infrastructure and is never used here.
int sw_fill_cpu(const char* a, int64_t cols, const char* b, int64_t rows, const sw_scores* scores, int32_t* H, int32_t* P, sw_result* result) {
    static const sw_scores kDefault = {3, -3, -2};
    const sw_scores* sc = scores ? scores : &kDefault;
    if (cols < 0 || rows < 0 || !H || !P || !result || (cols > 0 && !a) || (rows > 0 && !b)) { swh::set_err("sw_fill_cpu: bad argument"); return SW_EINVAL; }
    const int64_t m = cols + 1;
    int64_t best_pos = 0; int32_t best = 0;
    for (int64_t j = 0; j < m; ++j) { H[j] = 0; P[j] = 0; }
    for (int64_t i = 1; i <= rows; ++i) {
        int32_t* h = H + i * m; const int32_t* hu = h - m; int32_t* p = P + i * m;
        h[0] = 0; p[0] = 0;
        const char bi = b[i - 1];
        for (int64_t j = 1; j < m; ++j) {
            const int32_t diag = hu[j - 1] + (a[j - 1] == bi ? sc->match : sc->mismatch), up = hu[j] + sc->gap, left = h[j - 1] + sc->gap;
            int32_t mx = 0, pr = SW_NONE;
            if (diag > mx) { mx = diag; pr = SW_DIAGONAL; }
            if (up > mx) { mx = up; pr = SW_UP; }
            if (left > mx) { mx = left; pr = SW_LEFT; }
            h[j] = mx; p[j] = pr;
            if (mx > best) { best = mx; best_pos = i * m + j; }
        }
    }
    result->max_pos = best_pos; result->max_score = best; result->path_len = 0;
    return SW_OK;
}

// FASTA reader: the step before the path when the input is a real sequence instead of generate()
// (SURVEY.md 8f-1).  '>' starts a record, ';' lines are comments, white space is dropped, letters are
// upper-cased; a file without any '>' line is one record.
int sw_read_fasta(const char* path, int64_t record, char* seq, int64_t cap, int64_t* len) {
    if (!path || record < 0 || !len || (seq && cap < 0)) { swh::set_err("sw_read_fasta: bad argument"); return SW_EINVAL; }
    FILE* f = fopen(path, "rb");
    if (!f) { swh::set_err("sw_read_fasta: cannot open %s", path); return SW_EINVAL; }
    int64_t cur = -1, n = 0;       // cur: index of the record being read (-1: before the first header)
    bool at_line_start = true, skip_line = false, found = false;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const unsigned char ch = (unsigned char)buf[i];
            if (ch == '\n' || ch == '\r') { at_line_start = true; skip_line = false; continue; }
            if (at_line_start) {
                at_line_start = false;
                if (ch == '>') { ++cur; skip_line = true; if (cur == record) found = true; continue; }
                if (ch == ';') { skip_line = true; continue; }
                if (cur < 0) { cur = 0; if (record == 0) found = true; }   // headerless file: a single record
            }
            if (skip_line || cur != record || ch == ' ' || ch == '\t') continue;
            if (seq && n < cap) seq[n] = (char)((ch >= 'a' && ch <= 'z') ? ch - 32 : ch);
            ++n;
        }
        if (cur > record) break;
    }
    fclose(f);
    if (!found) { swh::set_err("sw_read_fasta: %s has no record %lld", path, (long long)record); return SW_EINVAL; }
    *len = n;
    return SW_OK;
}

}  // extern "C"
