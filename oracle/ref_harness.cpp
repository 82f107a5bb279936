/*
 * oracle/ref_harness.cpp -- drives the REAL reference functions, compiled in place.
 *
 * TEST INFRASTRUCTURE ONLY; exists only in the build container (/root/reference is not shipped).
 * The reference translation unit /root/reference/serial_smithW.c is #included where it lies
 * (no source is copied into this repo); its main() is renamed so that this driver can call
 * generate(), similarityScore() and backtrack() (serial_smithW.c:334, 187, 262) with a chosen
 * size/seed and dump raw results for tests/golden/make_goldens.py.
 *
 * usage: ref_harness <cols> <rows> <seed> <out_prefix>     random pair, srand(seed) (1 == no srand)
 *        ref_harness builtin <out_prefix>                  the built-in 8x9 example (serial_smithW.c:105-125)
 * writes <prefix>.a .b (chars), .H .P0 (P before traceback) .P1 (after) as int32 LE,
 *        .meta  "cols rows maxPos maxScore pathLen"
 *        .path  int64 LE linear indices in visiting order
 */
#define main ref_main_unused
#include "/root/reference/serial_smithW.c"
#undef main

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static void dump(const std::string& p, const void* d, size_t nbytes) {
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) { perror(p.c_str()); exit(2); }
    fwrite(d, 1, nbytes, f);
    fclose(f);
}

int main(int argc, char** argv) {
    bool builtin = (argc == 3 && !strcmp(argv[1], "builtin"));
    if (!builtin && argc != 5) { fprintf(stderr, "usage: see header\n"); return 2; }
    std::string prefix = builtin ? argv[2] : argv[4];
    long long cols, rows;
    if (builtin) { cols = 8; rows = 9; }
    else { cols = strtoll(argv[1], 0, 10); rows = strtoll(argv[2], 0, 10); }
    m = cols; n = rows;
    /* one byte more than the reference mallocs: generate() writes a[m], b[n] after m++,n++
     * (serial_smithW.c:87-92,337,350); the extra byte is written but never read. */
    a = (char*)malloc(m + 1);
    b = (char*)malloc(n + 1);
    m++; n++;
    int* H = (int*)calloc(m * n, sizeof(int));
    int* P = (int*)calloc(m * n, sizeof(int));
    if (builtin) {
        memcpy(b, "GGTTGACTA", 9);
        memcpy(a, "TGTTACGG", 8);
    } else {
        unsigned seed = (unsigned)strtoul(argv[3], 0, 10);
        if (seed != 1) srand(seed);   /* serial_smithW.c never calls srand */
        generate();
    }
    long long maxPos = 0;
    for (long long i = 1; i < n; i++)
        for (long long j = 1; j < m; j++)
            similarityScore(i, j, H, P, &maxPos);
    dump(prefix + ".a", a, cols);
    dump(prefix + ".b", b, rows);
    dump(prefix + ".H", H, sizeof(int) * m * n);
    dump(prefix + ".P0", P, sizeof(int) * m * n);
    long long pathLen = 0;
    std::vector<long long> path;
    if (P[maxPos] != NONE) {          /* reference would read an uninitialised predPos here */
        std::vector<int> before(P, P + m * n);
        backtrack(P, maxPos);
        /* recover visiting order by re-walking the (un-negated) copy */
        long long pos = maxPos;
        do {
            path.push_back(pos);
            int pr = before[pos];
            pos = pr == DIAGONAL ? pos - m - 1 : pr == UP ? pos - m : pos - 1;
        } while (before[pos] != NONE);
        pathLen = (long long)path.size();
        for (long long k = 0; k < pathLen; k++) if (P[path[k]] >= 0) { fprintf(stderr, "path mismatch\n"); return 3; }
    }
    dump(prefix + ".P1", P, sizeof(int) * m * n);
    dump(prefix + ".path", path.data(), sizeof(long long) * path.size());
    FILE* f = fopen((prefix + ".meta").c_str(), "w");
    fprintf(f, "%lld %lld %lld %d %lld\n", cols, rows, maxPos, H[maxPos], pathLen);
    fclose(f);
    return 0;
}
