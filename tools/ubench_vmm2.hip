// Placement study, round 4 (4): physical HBM comes in three "classes" (profiles/r04_placement_classes_probe.log): two store streams into the
// same class are ~1.4x slower than into different ones.  With the HIP virtual-memory API the library could BUILD its output buffers
// from classified physical chunks.  This tool: hipMemCreate N chunks, classify them with the two-stream probe, map 1 GiB ranges
// (a) from one class, (b) X and Y from two classes, (c) both striped over all three classes at several stripe sizes, and time the
// probe patterns on each; also what mapping costs, and the same probes on plain hipMalloc memory.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) probe(unsigned char* X, unsigned char* Y, long rows, long pitch, int seg_dw2, int nseg, int nrg, int mode, unsigned val) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = (int)blockIdx.x % nseg, rg = (int)blockIdx.x / nseg;
    if (lane >= seg_dw2) return;
    const long col = ((long)seg * seg_dw2 + lane) * 8;
    const v2u v = {val, val + (unsigned)lane};
    for (long r = rg + (long)wave * nrg; r < rows; r += 4L * nrg) {
        const long o = r * pitch + col;
        __builtin_nontemporal_store(v, (v2u*)(X + o));
        if (mode == 0) __builtin_nontemporal_store(v, (v2u*)(Y + o));
    }
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static float run(void* X, void* Y, long rows, long pitch, int seg, int nrg, int mode, int reps = 3) {
    const int nseg = (int)(pitch / (seg * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe, dim3(nseg * nrg), dim3(256), 0, nullptr, (unsigned char*)X, (unsigned char*)Y, rows, pitch, seg, nseg, nrg, mode, 1u);
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe, dim3(nseg * nrg), dim3(256), 0, nullptr, (unsigned char*)X, (unsigned char*)Y, rows, pitch, seg, nseg, nrg, mode, 2u + i);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}
int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    const size_t CH = (argc > 1 ? atol(argv[1]) : 64) << 20;     // chunk size
    const int N = argc > 2 ? atoi(argv[2]) : 96;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    // spread the chunks over the HBM: a spacer allocation every 8 chunks
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    std::vector<void*> spacers;
    double t0 = now_ms();
    for (int i = 0; i < N; ++i) {
        if (i && i % std::max(1, N / 10) == 0) { void* s = nullptr; if (hipMalloc(&s, 20ull << 30) == hipSuccess) spacers.push_back(s); else (void)hipGetLastError(); }
        CK(hipMemCreate(&h[i], CH, &prop, 0));
    }
    printf("hipMemCreate %d x %zu MiB (+%zu spacers of 20 GiB): %.2f ms\n", N, CH >> 20, spacers.size(), now_ms() - t0);
    // every chunk mapped once at its own address (for the classification)
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, CH * N, 2ull << 20, nullptr, 0));
    t0 = now_ms();
    for (int i = 0; i < N; ++i) CK(hipMemMap((char*)va + i * CH, CH, 0, h[i], 0));
    double t1 = now_ms();
    CK(hipMemSetAccess(va, CH * N, &acc, 1));
    printf("map %d chunks: %.2f ms, set access: %.2f ms\n", N, t1 - t0, now_ms() - t1);
    auto chunk = [&](int i) { return (char*)va + (size_t)i * CH; };
    // classification: two-stream probe over the first 32 MiB of each chunk (rows x 64 KiB pitch, 512-byte segments, contiguous)
    const long prow = (long)(std::min<size_t>(CH, 32ull << 20) / 65536);
    for (int i = 0; i < 200; ++i) run(chunk(0), chunk(1), prow, 65536, 64, 4, 0, 1);   // clocks
    std::vector<int> cls(N, -1);
    std::vector<int> ref;
    t0 = now_ms();
    const float one = run(chunk(0), nullptr, prow, 65536, 64, 4, 1);
    for (int i = 0; i < N; ++i) {
        for (size_t c = 0; c < ref.size() && cls[i] < 0; ++c) {
            if (ref[c] == i) { cls[i] = (int)c; break; }
            const float t = run(chunk(ref[c]), chunk(i), prow, 65536, 64, 4, 0);
            if (t > 1.6f * one) cls[i] = (int)c;   // as slow as one stream twice: the same class
        }
        if (cls[i] < 0) { cls[i] = (int)ref.size(); ref.push_back(i); }
    }
    printf("classification of %d chunks: %.2f ms (one stream %.4f ms); classes found: %zu\n  ", N, now_ms() - t0, one, ref.size());
    for (int i = 0; i < N; ++i) printf("%d", cls[i]);
    printf("\n");
    // raw numbers for the first reference against every chunk
    printf("  two-stream times vs chunk 0 (ms x1000):");
    for (int i = 1; i < N; ++i) printf(" %d", (int)(1000 * run(chunk(0), chunk(i), prow, 65536, 64, 4, 0)));
    printf("\n");
    std::vector<std::vector<int>> by(ref.size());
    for (int i = 0; i < N; ++i) by[cls[i]].push_back(i);
    for (size_t c = 0; c < by.size(); ++c) printf("  class %zu: %zu chunks\n", c, by[c].size());
    CK(hipMemUnmap(va, CH * N));
    // build X and Y (1 GiB each) from lists of chunk ids, `piece` bytes of consecutive chunk content per stripe
    const size_t GB = 1ull << 30;
    const long rowsF = 16385, pitchF = 65540;
    auto build = [&](const std::vector<int>& ids, size_t stripe, char* base) {   // ids: chunks (each used whole), stripe <= CH
        // the range is cut into stripes; stripe k comes from chunk ids[k % n] at offset (k / n) * stripe
        const int n = (int)ids.size();
        size_t k = 0;
        for (size_t off = 0; off < GB + (64ull << 20); off += stripe, ++k) {
            const size_t coff = (k / n) * stripe;
            if (coff + stripe > CH) { printf("not enough chunks for stripe build\n"); return false; }
            const hipError_t e = hipMemMap(base + off, stripe, coff, h[ids[k % n]], 0);
            if (e != hipSuccess) { printf("hipMemMap(offset %zu) -> %s\n", coff, hipGetErrorString(e)); (void)hipGetLastError(); if (off) (void)hipMemUnmap(base, off); return false; }
        }
        return true;
    };
    auto test = [&](const char* name, const std::vector<int>& xi, const std::vector<int>& yi, size_t stripe) {
        const size_t span = GB + (64ull << 20);
        char* base = (char*)va;
        double a = now_ms();
        if (!build(xi, stripe, base)) return;
        if (!build(yi, stripe, base + span)) { (void)hipMemUnmap(base, span); return; }
        CK(hipMemSetAccess(base, 2 * span, &acc, 1));
        double b = now_ms();
        const float f = run(base, base + span, rowsF, pitchF, 63, 2, 0), q = run(base, base + span, rowsF, pitchF, 63, 8, 0);
        const float cg = run(base, base + span, 16384, 65536, 64, 4, 0), o1 = run(base, nullptr, rowsF, pitchF, 63, 2, 1), oc = run(base, nullptr, 16384, 65536, 64, 4, 1);
        printf("%-46s stripe %5zu KiB: map %.1f ms | 504B 2rg %.3f, 8rg %.3f, contiguous %.3f | one stream: 504B %.3f contiguous %.3f\n", name, stripe >> 10, b - a, f, q, cg, o1, oc);
        CK(hipMemUnmap(base, 2 * span));
    };
    if (ref.size() >= 2) {
        const size_t need = (GB + (64ull << 20)) / CH + 1;   // chunks per 1 GiB range when used whole
        auto take = [&](int c, size_t from, size_t n) { std::vector<int> v; for (size_t i = from; i < from + n && i < by[c].size(); ++i) v.push_back(by[c][i]); return v; };
        if (by[0].size() >= 2 * need) test("X, Y both class 0", take(0, 0, need), take(0, need, need), CH);
        if (by[0].size() >= need && by[1].size() >= need) test("X class 0, Y class 1", take(0, 0, need), take(1, 0, need), CH);
        if (ref.size() >= 3) {
            size_t m = std::min(by[0].size(), std::min(by[1].size(), by[2].size()));
            const size_t k = (need + 2) / 3 + 1;    // chunks per class and range
            if (m >= 2 * k) {
                std::vector<int> x, y;
                for (size_t i = 0; i < k; ++i) { x.push_back(by[0][i]); x.push_back(by[1][i]); x.push_back(by[2][i]); }
                for (size_t i = k; i < 2 * k; ++i) { y.push_back(by[1][i]); y.push_back(by[2][i]); y.push_back(by[0][i]); }   // Y one class out of phase with X
                for (size_t st : {CH, (size_t)(16u << 20), (size_t)(2u << 20), (size_t)(256u << 10), (size_t)(64u << 10)})
                    if (st <= CH) test("X, Y striped over 3 classes (Y out of phase)", x, y, st);
                std::vector<int> y2;
                for (size_t i = k; i < 2 * k; ++i) { y2.push_back(by[0][i]); y2.push_back(by[1][i]); y2.push_back(by[2][i]); }   // Y in phase with X
                test("X, Y striped over 3 classes (in phase)", x, y2, 2u << 20);
            } else printf("not enough chunks per class for the striped test (%zu)\n", m);
        }
    }
    // plain hipMalloc for comparison
    void *px = nullptr, *py = nullptr;
    CK(hipMalloc(&px, GB + (64ull << 20))); CK(hipMalloc(&py, GB + (64ull << 20)));
    printf("%-46s                    | 504B 2rg %.3f, 8rg %.3f, contiguous %.3f | one stream: 504B %.3f contiguous %.3f\n", "plain hipMalloc pair",
           run(px, py, rowsF, pitchF, 63, 2, 0), run(px, py, rowsF, pitchF, 63, 8, 0), run(px, py, 16384, 65536, 64, 4, 0), run(px, nullptr, rowsF, pitchF, 63, 2, 1),
           run(px, nullptr, 16384, 65536, 64, 4, 1));
    printf("done\n");
    return 0;
}
