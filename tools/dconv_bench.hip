// Development tool: times the narrow-channel direct convolutions (csrc/dconv.hip) on ContextAEReal's layer shapes at
// B = 256, 36x64, for every legal tile (TH, TW, MI), next to the automatic choice.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I imitation_from_observation_amd/csrc tools/dconv_bench.hip \
//         imitation_from_observation_amd/csrc/dconv.hip -o tools/dconv_bench.bin && tools/dconv_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "launch.h"
#include "options.h"
#include <cmath>
#include <algorithm>

namespace ctx {
void dconv_force_tile(int th, int tw, int mi);
extern int g_dc_last[4];
extern int g_dc2_last[5];
}
using namespace ctx;

struct Layer { const char* name; int kind; int CI, c1, nimg, hin, win, S, N; };   // kind 0 conv, 1 convt1, 2 convt2

static float* dalloc(size_t n, float val) {
    float* p;
    (void)hipMalloc(&p, n * 4);
    std::vector<float> h(n, val);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;      // one layer, automatic tile, three launches: for rocprofv3 --pmc
    const bool auto_only = getenv("DCB_AUTO_ONLY") != nullptr;   // no tile sweep: the automatic choice of every layer (ablation builds)
    const int B = 256;
    const Layer layers[] = {
        {"h0 fwd      F s1  3->32  36x64 x768", 0, 3, 3, 3 * B, 36, 64, 1, 32},
        {"h1 fwd      F s2 32->16  36x64 x768", 0, 32, 32, 3 * B, 36, 64, 2, 16},
        {"h2 fwd      F s1 16->16  18x32 x768", 0, 16, 16, 3 * B, 18, 32, 1, 16},
        {"h3 fwd      F s2 16->8   18x32 x768", 0, 16, 16, 3 * B, 18, 32, 2, 8},
        {"d_h4 dx     F s1  3->64  36x64 x512", 0, 3, 3, 2 * B, 36, 64, 1, 64},
        {"d_h3 dx     F s2 32->32  36x64 x512", 0, 32, 32, 2 * B, 36, 64, 2, 32},
        {"d_h2 dx     F s1 16->32  18x32 x512", 0, 16, 16, 2 * B, 18, 32, 1, 32},
        {"d_h1 dx     F s2 16->16  18x32 x512", 0, 16, 16, 2 * B, 18, 32, 2, 16},
        {"d_h1 fwd    T s2 8|8->16  9x16 x512", 2, 16, 8, 2 * B, 9, 16, 2, 16},
        {"d_h2 fwd    T s1 16|16->16 18x32x512", 1, 32, 16, 2 * B, 18, 32, 1, 16},
        {"d_h3 fwd    T s2 16|16->32 18x32x512", 2, 32, 16, 2 * B, 18, 32, 2, 32},
        {"h3 dx       T s2  8->16   9x16 x768", 2, 8, 8, 3 * B, 9, 16, 2, 16},
        {"h2 dx       T s1 16->16  18x32 x768", 1, 16, 16, 3 * B, 18, 32, 1, 16},
        {"h1 dx       T s2 16->32  18x32 x768", 2, 16, 16, 3 * B, 18, 32, 2, 32},
    };
    hipStream_t st;
    (void)hipStreamCreate(&st);
    float* x1 = dalloc((size_t)3 * B * 36 * 64 * 32, 0.f);
    float* x2 = dalloc((size_t)B * 36 * 64 * 32, 0.f);
    float* w = dalloc(25 * 64 * 128, 0.f);
    float* bias = dalloc(128, 0.f);
    float* out = dalloc((size_t)3 * B * 36 * 64 * 64, 0.f);
    float* wp = dalloc(DC_WPACK_FLOATS, 0.f);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int li = -1;
    for (const Layer& L : layers) {
        ++li;
        if (only >= 0 && li != only) continue;
        DcFwd P{};
        P.x1 = x1; P.ld1 = L.c1; P.c1 = L.c1; P.CI = L.CI;
        if (L.c1 < L.CI) { P.x2 = x2; P.ld2 = L.CI - L.c1; P.nmod2 = B; }
        P.hin = L.hin; P.win = L.win; P.nimg = L.nimg; P.w = w; P.wmode = L.kind ? 1 : 0; P.N = L.N; P.wp = wp;
        P.ep.out1 = out; P.ep.ld1 = L.N; P.ep.bias = bias; P.ep.lrelu = 1;
        auto run = [&]() {
            if (L.kind == 0) dconv_conv(st, P, L.S, L.S == 1 ? 2 : 1);
            else if (L.kind == 1) dconv_convt1(st, P);
            else dconv_convt2(st, P);
        };
        auto timeit = [&]() {
            run();
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return -1.f;
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < 10; ++i) run();
            (void)hipEventRecord(e1, st);
            (void)hipStreamSynchronize(st);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            return ms / 10;
        };
        const int hl = L.kind == 0 ? L.hin / L.S : L.hin, wl = L.kind == 0 ? L.win / L.S : L.win;
        const double flops = 2.0 * L.nimg * (L.kind == 2 ? 4.0 * hl * wl * 6.25 : (double)hl * wl * 25) * L.CI * L.N;
        dconv_force_tile(0, 0, 0);
        if (getenv("DCB_VERIFY")) {   // dconv2 against dconv_fwd_kernel on the same operands, many launches: a race detector
            const size_t nout = (size_t)L.nimg * (L.kind == 2 ? 4 : 1) * hl * wl * L.N;
            std::vector<float> ref(nout), got(nout);
            Options o1 = options_from_env(); o1.v[OPT_DCONV] = 1;
            Options o3 = options_from_env(); o3.v[OPT_DCONV] = 7;
            (void)hipMemset(out, 0, nout * 4);
            { OptScope os(&o1); run(); }
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(ref.data(), out, nout * 4, hipMemcpyDeviceToHost);
            int bad_runs = 0; double worst = 0; size_t worst_n = 0;
            const int reps = atoi(getenv("DCB_VERIFY")) > 0 ? atoi(getenv("DCB_VERIFY")) : 20;
            static hipStream_t hog = nullptr; static char* hb = nullptr;
            if (getenv("DCB_HOG") && !hog) { (void)hipStreamCreate(&hog); (void)hipMalloc(&hb, 2ull << 30); }
            for (int r = 0; r < reps; ++r) {
                (void)hipMemset(out, 0, nout * 4);
                if (hog) for (int k = 0; k < 4; ++k) (void)hipMemcpyAsync(hb, hb + (1ull << 30), 1ull << 30, hipMemcpyDeviceToDevice, hog);   // memory traffic beside the launch
                { OptScope os(&o3); run(); }
                (void)hipStreamSynchronize(st);
                if (hog) (void)hipStreamSynchronize(hog);
                (void)hipMemcpy(got.data(), out, nout * 4, hipMemcpyDeviceToHost);
                double mx = 0, sc = 0; size_t nb = 0, first = 0;
                for (size_t i = 0; i < nout; ++i) { sc = std::max(sc, (double)fabsf(ref[i])); }
                for (size_t i = 0; i < nout; ++i) { const double d = fabs((double)got[i] - ref[i]); if (d > 1e-4 * sc) { if (!nb) first = i; ++nb; } mx = std::max(mx, d); }
                if (nb) { ++bad_runs; if (nb > worst_n) worst_n = nb; printf("   run %d: %zu elements off (first at %zu = pixel %zu ch %zu), max |d| / max|ref| %.2e\n", r, nb, first, first / L.N, first % L.N, mx / sc); }
                worst = std::max(worst, mx / sc);
            }
            printf("%s  verify: %d of %d launches differ from dconv_fwd_kernel (worst %.2e, up to %zu elements)  [dconv2 tile TH %d TW %d]\n", L.name, bad_runs, reps, worst, worst_n, g_dc2_last[0], g_dc2_last[1]);
            continue;
        }
#ifdef DC_TRACE
        {   // phase stamps of blocks 0..7, all waves, the first 32 tiles each: mean duration of each phase in shader cycles
            unsigned long long* tr;
            const size_t nst = (size_t)8 * 8 * 32 * 8;
            (void)hipMalloc(&tr, nst * 8); (void)hipMemset(tr, 0, nst * 8);
            run(); (void)hipStreamSynchronize(st);
            P.trace = tr; run(); (void)hipStreamSynchronize(st); P.trace = nullptr;
            std::vector<unsigned long long> h(nst);
            (void)hipMemcpy(h.data(), tr, nst * 8, hipMemcpyDeviceToHost);
            const char* nm1[7] = {"issue", "classes(loops+mid epilogues)", "barrier A", "land", "epilogue(last)", "barrier B", "tile total"};
            const char* nm2[7] = {"pre+DMA requests", "MFMA loops+fold", "wait DMA", "barrier", "stores", "-", "LAST SLICE total"};
            const char** nm = g_dc2_last[0] ? nm2 : nm1;
            double sum[7] = {}; int n = 0;
            for (int b = 0; b < 8; ++b) for (int w = 0; w < 8; ++w) for (int i = 1; i < 31; ++i) {   // (dconv2: the compute waves)
                const unsigned long long* s = &h[(((size_t)b * 8 + w) * 32 + i) * 8];
                if (!s[0] || !s[6]) continue;
                for (int k = 0; k < 6; ++k) sum[k] += (double)(s[k + 1] - s[k]);
                sum[6] += (double)(s[6] - s[0]); ++n;
            }
            printf("%s  phases (cycles, mean over %d wave-tiles):", L.name, n);
            for (int k = 0; k < 7; ++k) printf("  %s %.0f", nm[k], n ? sum[k] / n : 0.0);
            printf("\n");
            (void)hipFree(tr);
        }
#endif
        if (only >= 0) { run(); run(); run(); (void)hipStreamSynchronize(st); continue; }
        const float t_auto = timeit();
        if (g_dc2_last[0]) printf("%s  dconv2: TH %d TW %d MI %d NB*10+occ %d slices %d  %.3f ms  %.1f TF/s\n", L.name, g_dc2_last[0], g_dc2_last[1], g_dc2_last[2], g_dc2_last[3], g_dc2_last[4], t_auto, flops / t_auto / 1e9);
        else printf("%s  auto: TH %d TW %d MI %d NB*10+occ %d  %.3f ms  %.1f TF/s\n", L.name, g_dc_last[0], g_dc_last[1], g_dc_last[2], g_dc_last[3], t_auto,
               flops / t_auto / 1e9);
        struct R { int th, tw, mi; float ms; };
        std::vector<R> rs;
        for (int tw = 16; tw <= 64 && !auto_only; tw *= 2) {
            if (tw > (wl + 15) / 16 * 16) continue;
            for (int mi = 1; mi <= 4; ++mi)
                for (int th = 1; th <= 16; ++th) {
                    const int nrb = th * tw / 16;
                    if (nrb > 8 * mi || nrb <= 8 * (mi - 1) / 2) continue;          // skip tiles a smaller MI covers as well
                    const int span = L.kind == 2 ? 3 : 5, S = L.kind == 0 ? L.S : 1;
                    const int cik = L.CI == 3 ? 4 : L.CI, cip = cik == 4 ? 4 : cik + 4;
                    const size_t tile = (size_t)(S * (th - 1) + span) * (S * (tw - 1) + span) * cip * 4;
                    dconv_force_tile(th, tw, mi);
                    const float ms = timeit();
                    if (ms > 0 && g_dc_last[0] == th && g_dc_last[1] == tw && g_dc_last[2] == mi) rs.push_back({th, tw, mi, ms});
                }
        }
        for (int k = 0; k < 6 && !rs.empty(); ++k) {
            size_t bi = 0;
            for (size_t i = 1; i < rs.size(); ++i) if (rs[i].ms < rs[bi].ms) bi = i;
            printf("      TH %2d TW %2d MI %d  %.3f ms  %.1f TF/s\n", rs[bi].th, rs[bi].tw, rs[bi].mi, rs[bi].ms, flops / rs[bi].ms / 1e9);
            rs.erase(rs.begin() + bi);
        }
        fflush(stdout);
    }
    if (only >= 0 && only < 100) return 0;
    // ---- filter gradients (automatic tile only)
    struct WL { const char* name; int CA, c1, CB, nimg, hb, wb, S; };
    const WL wl[] = {
        {"h0 dw    3 x 32   36x64 s1 x768", 3, 32, 32, 3 * B, 36, 64, 1},
        {"h1 dw   32 x 16   36x64 s2 x768", 32, 16, 16, 3 * B, 36, 64, 2},
        {"h2 dw   16 x 16   18x32 s1 x768", 16, 16, 16, 3 * B, 18, 32, 1},
        {"h3 dw   16 x 8    18x32 s2 x768", 16, 8, 8, 3 * B, 18, 32, 2},
        {"d_h1 dw 16 x 8|8  18x32 s2 x512", 16, 8, 16, 2 * B, 18, 32, 2},
        {"d_h2 dw 16 x 16|16 18x32 s1 x512", 16, 16, 32, 2 * B, 18, 32, 1},
        {"d_h3 dw 32 x 16|16 36x64 s2 x512", 32, 16, 32, 2 * B, 36, 64, 2},
        {"d_h4 dw  3 x 32|32 36x64 s1 x512", 3, 32, 64, 2 * B, 36, 64, 1},
    };
    const int64_t slab_floats = 8ll << 20;
    float* slab = dalloc((size_t)slab_floats, 0.f);
    int wi = 99;
    for (const WL& L : wl) {
        ++wi;
        if (only >= 100 && wi != only) continue;
        DcWgrad W{};
        W.big = x1; W.ldb = L.CA; W.CA = L.CA; W.s1 = out; W.ld1 = L.c1; W.c1 = L.c1; W.CB = L.CB;
        if (L.c1 < L.CB) { W.s2 = x2; W.ld2 = L.CB - L.c1; W.nmod2 = B; }
        W.hb = L.hb; W.wb = L.wb; W.hs = L.hb / L.S; W.ws = L.wb / L.S; W.nimg = L.nimg; W.S = L.S; W.pad = L.S == 1 ? 2 : 1; W.out = w;
        dconv_wgrad(st, W, slab, slab_floats);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s  FAILED\n", L.name); continue; }
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < 10; ++i) dconv_wgrad(st, W, slab, slab_floats);
        (void)hipEventRecord(e1, st);
        (void)hipStreamSynchronize(st);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        const double flops = 2.0 * L.nimg * (double)W.hs * W.ws * 25 * L.CA * L.CB;
        printf("%s  %.3f ms  %.1f TF/s\n", L.name, ms, flops / ms / 1e9);
        fflush(stdout);
    }
    return 0;
}
