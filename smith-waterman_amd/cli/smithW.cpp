// smithW -- host driver with the reference's command line (serial_smithW.c:71-180, omp_smithW.c:87-253):
//   smithW                  built-in 8x9 example (serial_smithW.c:105-125) + its known-answer checks
//   smithW <cols> <rows>    random DNA pair from the reference generator (seed 1 == serial_smithW.c)
//   smithW --fasta A.fa B.fa   real sequences: a = first record of A.fa (columns), b = first record of B.fa (rows)
// Extra flags: --seed N  --dump | --dump-labels (the header-row printers of omp_smithW.c)  --h64  --no-backtrack  --scores M X G  --record-a I  --record-b J
//   --gpus N | --devices 0,1,..   ONE matrix over several GPUs (row bands, sw_multi_*; an id may repeat)  --p8  int8 P
// The DP fill runs on the GPU through the C-ABI (include/swhip.h); stdout keeps the two
// "Elapsed time ..." lines the reference's run scripts grep for (readme.liao:12).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/swhip.h"

#define RESET "\033[0m"
#define BOLDRED "\033[1m\033[31m"

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != SW_OK) {                                                      \
            fprintf(stderr, "smithW: %s -> %d: %s\n", #call, rc_, sw_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

// printMatrix / printPredecessorMatrix, serial_smithW.c:283-328
static void print_matrix(const std::vector<int32_t>& H, long long n, long long m) {
    for (long long i = 0; i < n; i++) {
        for (long long j = 0; j < m; j++) printf("%d\t", H[m * i + j]);
        printf("\n");
    }
}
static void print_pred(const std::vector<int32_t>& P, long long n, long long m) {
    for (long long i = 0; i < n; i++) {
        for (long long j = 0; j < m; j++) {
            const int v = P[m * i + j], av = v < 0 ? -v : v;
            const char* sym = av == SW_UP ? "↑ " : av == SW_LEFT ? "← " : av == SW_DIAGONAL ? "↖ " : "- ";
            if (v < 0) printf(BOLDRED "%s" RESET, sym); else printf("%s", sym);
        }
        printf("\n");
    }
}

// the labelled variants (sequence a as a header row, the letters of b in front of the rows), omp_smithW.c:426-483
static void print_matrix_labelled(const std::vector<int32_t>& H, long long n, long long m, const char* a, const char* b) {
    printf("-\t-\t");
    for (long long j = 0; j + 1 < m; j++) printf("%c\t", a[j]);
    printf("\n-\t");
    for (long long i = 0; i < n; i++) {
        if (i > 0) printf("%c\t", b[i - 1]);
        for (long long j = 0; j < m; j++) printf("%d\t", H[m * i + j]);
        printf("\n");
    }
}
static void print_pred_labelled(const std::vector<int32_t>& P, long long n, long long m, const char* a, const char* b) {
    printf("    ");
    for (long long j = 0; j + 1 < m; j++) printf("%c ", a[j]);
    printf("\n  ");
    for (long long i = 0; i < n; i++) {
        if (i > 0) printf("%c ", b[i - 1]);
        for (long long j = 0; j < m; j++) {
            const int v = P[m * i + j], av = v < 0 ? -v : v;
            const char* sym = av == SW_UP ? "↑ " : av == SW_LEFT ? "← " : av == SW_DIAGONAL ? "↖ " : "- ";
            if (v < 0) printf(BOLDRED "%s" RESET, sym); else printf("%s", sym);
        }
        printf("\n");
    }
}

int main(int argc, char** argv) {
    long long cols = 8, rows = 9;
    bool builtin = true, dump = false, labels = false, h64 = false, backtrack = true, p8 = false;
    std::vector<int> devices;
    unsigned seed = 1;
    const char *fasta_a = nullptr, *fasta_b = nullptr;
    long long rec_a = 0, rec_b = 0;
    sw_scores sc = {3, -3, -2};
    int npos = 0;
    for (int ai = 1; ai < argc; ++ai) {
        std::string f = argv[ai];
        if (f[0] != '-' && npos < 2 && ai + (1 - npos) < argc) {   // <cols> <rows>, anywhere on the line
            (npos == 0 ? cols : rows) = strtoll(argv[ai], nullptr, 10);
            builtin = false;
            ++npos;
        }
        else if (f == "--dump") dump = true;
        else if (f == "--dump-labels") dump = labels = true;
        else if (f == "--h64") h64 = true;
        else if (f == "--no-backtrack") backtrack = false;
        else if (f == "--p8") p8 = true;
        else if (f == "--gpus" && ai + 1 < argc) { const int n = atoi(argv[++ai]); for (int g = 0; g < n; ++g) devices.push_back(g); }
        else if (f == "--devices" && ai + 1 < argc) { for (char* t = strtok(argv[++ai], ","); t; t = strtok(nullptr, ",")) devices.push_back(atoi(t)); }
        else if (f == "--fasta" && ai + 2 < argc) { fasta_a = argv[++ai]; fasta_b = argv[++ai]; builtin = false; }
        else if (f == "--record-a" && ai + 1 < argc) rec_a = strtoll(argv[++ai], nullptr, 10);
        else if (f == "--record-b" && ai + 1 < argc) rec_b = strtoll(argv[++ai], nullptr, 10);
        else if (f == "--seed" && ai + 1 < argc) seed = (unsigned)strtoul(argv[++ai], nullptr, 10);
        else if (f == "--scores" && ai + 3 < argc) { sc.match = atoi(argv[++ai]); sc.mismatch = atoi(argv[++ai]); sc.gap = atoi(argv[++ai]); }
        else { fprintf(stderr, "usage: smithW [<cols> <rows> | --fasta A.fa B.fa [--record-a I] [--record-b J]] [--seed N] [--dump | --dump-labels] [--h64] [--no-backtrack] [--scores M X G] [--gpus N | --devices 0,1,..] [--p8]\n"); return 2; }
    }
    if (npos == 1) { fprintf(stderr, "smithW: <cols> needs <rows>\n"); return 2; }
    if (fasta_a) {
        int64_t la = 0, lb = 0;
        CHECK(sw_read_fasta(fasta_a, rec_a, nullptr, 0, &la));
        CHECK(sw_read_fasta(fasta_b, rec_b, nullptr, 0, &lb));
        cols = la; rows = lb;
    }
    const long long m = cols + 1, n = rows + 1;
    std::vector<char> a(m + 1), b(n + 1);
    if (fasta_a) {
        int64_t la = 0, lb = 0;
        CHECK(sw_read_fasta(fasta_a, rec_a, a.data(), cols, &la));
        CHECK(sw_read_fasta(fasta_b, rec_b, b.data(), rows, &lb));
    } else if (builtin) { memcpy(b.data(), "GGTTGACTA", 9); memcpy(a.data(), "TGTTACGG", 8); }
    else CHECK(sw_generate(cols, rows, seed, a.data(), b.data()));
    if (dump) { if (builtin) printf("\n Using built-in data for testing .."); printf("\nMatrix[%lld][%lld]\n", rows, cols); }
    else {
        // the lines omp_smithW-v1-refinedOrig.cpp:119,138-142 print (there both matrices and both sequences are ints; here the
        // footprint is what is resident in HBM: H int32 | int64, P int32 | int8, one byte per letter)
        printf("Problem size: Matrix[%lld][%lld]\n", n, m);
        const unsigned long long sz = ((unsigned long long)(m + n) + (unsigned long long)m * n * ((h64 ? 8 : 4) + (p8 && !devices.empty() ? 1 : 4))) / 1024 / 1024;
        if (sz >= 1024) printf("Total memory footprint is:%llu GB\n", sz / 1024);
        else printf("Total memory footprint is:%llu MB\n", sz);
    }

    if (!devices.empty()) {
        // ---- one matrix over several GPUs: row bands, one band-resident launch per GPU, halo rows relayed over xGMI ----
        sw_multi* mh = nullptr;
        CHECK(sw_multi_create(devices.data(), (int)devices.size(), a.data(), cols, b.data(), rows, p8 ? 1 : 4, 1, &mh));
        sw_result res;
        CHECK(sw_multi_fill(mh, &sc, 64, &res));    // untimed: sizes the workspaces
        CHECK(sw_multi_fill(mh, &sc, 64, &res));
        printf("\nElapsed time for scoring matrix computation: %f\n\n", sw_multi_seconds(mh));
        double t0 = now_s();
        int64_t plen = 0;
        if (backtrack) CHECK(sw_multi_traceback(mh, &plen));
        printf("\nElapsed time for backtracking: %f\n\n", now_s() - t0);
        printf("maxPos = %lld, H[maxPos] = %lld, path length = %lld  (%d row bands)\n", (long long)res.max_pos, (long long)res.max_score,
               (long long)plen, sw_multi_nbands(mh));
        sw_multi_free(mh);
        return 0;
    }
    sw_ctx* ctx = nullptr;
    CHECK(sw_create(0, &ctx));
    const size_t cells = (size_t)m * (size_t)n;
    void *d_a, *d_b, *d_H, *d_P, *d_res;
    CHECK(sw_device_malloc(ctx, (size_t)cols + 16, &d_a));
    CHECK(sw_device_malloc(ctx, (size_t)rows + 16, &d_b));
    CHECK(sw_device_malloc(ctx, sizeof(sw_result), &d_res));
    CHECK(sw_memcpy_h2d(ctx, d_a, a.data(), (size_t)cols));
    CHECK(sw_memcpy_h2d(ctx, d_b, b.data(), (size_t)rows));
    // output matrices placed for speed (allocation is outside the reference's timer as well, serial_smithW.c:96-103,137)
    CHECK(sw_alloc_outputs(ctx, (const char*)d_a, cols, (const char*)d_b, rows, &sc, h64 ? 8 : 4, 4, cells >= (1u << 24) ? 0 : 1, &d_H, &d_P, nullptr));
    // one untimed call sizes the workspace (the reference's timer also excludes its allocations)
    CHECK(sw_fill_device(ctx, (const char*)d_a, cols, (const char*)d_b, rows, &sc, d_H, h64 ? 8 : 4, (int32_t*)d_P, nullptr, (sw_result*)d_res, nullptr));
    CHECK(sw_synchronize(ctx, nullptr));

    if (backtrack) {   // ... and loads the traceback kernel (first use of a kernel pays for its code object): a walk of nothing on a 1 x 1 matrix
        void *d_p1, *d_r1;
        CHECK(sw_device_malloc(ctx, 4 * sizeof(int32_t), &d_p1));
        CHECK(sw_device_malloc(ctx, sizeof(sw_result), &d_r1));
        const int32_t zeros[4] = {0, 0, 0, 0}; const sw_result r0 = {0, 0, 0};
        CHECK(sw_memcpy_h2d(ctx, d_p1, zeros, sizeof zeros));
        CHECK(sw_memcpy_h2d(ctx, d_r1, &r0, sizeof r0));
        CHECK(sw_traceback_device(ctx, (int32_t*)d_p1, 1, 1, 3, nullptr, 0, (sw_result*)d_r1, nullptr));
        CHECK(sw_synchronize(ctx, nullptr));
        sw_device_free(ctx, d_p1); sw_device_free(ctx, d_r1);
    }

    double t0 = now_s();
    CHECK(sw_fill_device(ctx, (const char*)d_a, cols, (const char*)d_b, rows, &sc, d_H, h64 ? 8 : 4, (int32_t*)d_P, nullptr, (sw_result*)d_res, nullptr));
    CHECK(sw_synchronize(ctx, nullptr));
    double t1 = now_s();
    printf("\nElapsed time for scoring matrix computation: %f\n\n", t1 - t0);
    sw_result res;
    CHECK(sw_memcpy_d2h(ctx, &res, d_res, sizeof res));
    if (res.path_len < 0) { fprintf(stderr, "smithW: device hand-off timed out\n"); return 1; }

    t0 = now_s();
    if (backtrack) {
        CHECK(sw_traceback_device(ctx, (int32_t*)d_P, cols, rows, res.max_pos, nullptr, 0, (sw_result*)d_res, nullptr));
        CHECK(sw_synchronize(ctx, nullptr));
        CHECK(sw_memcpy_d2h(ctx, &res, d_res, sizeof res));
    }
    t1 = now_s();
    printf("\nElapsed time for backtracking: %f\n\n", t1 - t0);
    printf("maxPos = %lld, H[maxPos] = %lld, path length = %lld\n", (long long)res.max_pos,
           (long long)res.max_score, (long long)res.path_len);

    int rc = 0;
    if (dump || builtin) {
        std::vector<int32_t> H(cells), P(cells);
        if (h64) {
            std::vector<int64_t> H8(cells);
            CHECK(sw_memcpy_d2h(ctx, H8.data(), d_H, cells * 8));
            for (size_t k = 0; k < cells; ++k) H[k] = (int32_t)H8[k];
        } else CHECK(sw_memcpy_d2h(ctx, H.data(), d_H, cells * 4));
        CHECK(sw_memcpy_d2h(ctx, P.data(), d_P, cells * 4));
        if (dump) { printf("\nSimilarity Matrix:\n"); if (labels) print_matrix_labelled(H, n, m, a.data(), b.data()); else print_matrix(H, n, m); }
        if (builtin) {
            // the reference's built-in checks: serial_smithW.c:162-166, omp_smithW-v1-refinedOrig.cpp:229-238
            const bool ok = H[m * n - 1] == 7 && res.max_pos == 69 && res.max_score == 13;
            printf("Verifying correctness using builtin data =%d\n", ok);
            if (!ok) rc = 1;
        }
        if (dump) { printf("\nPredecessor Matrix:\n"); if (labels) print_pred_labelled(P, n, m, a.data(), b.data()); else print_pred(P, n, m); }
    }
    sw_device_free(ctx, d_a); sw_device_free(ctx, d_b); sw_free_outputs(ctx, d_H, d_P); sw_device_free(ctx, d_res);
    sw_destroy(ctx);
    return rc;
}
