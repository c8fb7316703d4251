// Development tool: times the matrix-core d_h4 forward (csrc/convt3m.hip) alone on the bench shapes and prints in-kernel cycle stamps of
// its phases (built with -DM3_TRACE; the product build carries no stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DM3_TRACE -I imitation_from_observation_amd/csrc tools/convt3m_bench.hip \
//         -L imitation_from_observation_amd -l:libctxtrans.so -Wl,-rpath,'$ORIGIN/../imitation_from_observation_amd' -o tools/convt3m_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../imitation_from_observation_amd/csrc/convt3m.hip"

using namespace ctx;

static float* dalloc(size_t n) {
    float* p;
    (void)hipMalloc(&p, n * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return p;
}

template <int S, int KQH, int NWM, int NWG>
static void run(const char* name, int nimg, int nmod, int hin, int win, int prio_g = 0, bool trace = true) {
    const int C1 = 16 * KQH;
    Ct3m A{};
    A.x1 = dalloc((size_t)nimg * hin * win * C1); A.x2 = dalloc((size_t)nmod * hin * win * C1); A.nmod2 = nmod;
    A.hin = hin; A.win = win; A.nimg = nimg; A.npix = hin * win;
    A.w = dalloc(75 * 2 * C1); A.bias = dalloc(3); A.prio_g = prio_g;
    A.out = dalloc((size_t)nimg * S * hin * S * win * 3);
    constexpr int NW = NWM + NWG;
    (void)hipMalloc(&A.trace, 4 * NW * 32 * 8 * 8);
    (void)hipMemset(A.trace, 0, 4 * NW * 32 * 8 * 8);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0, st);
        if (!launch_ct3m<S, KQH, NWM, NWG>(st, A)) { printf("%s: does not fit\n", name); return; }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double fl = 2.0 * nimg * hin * win * 75.0 * 2 * C1;
    printf("%-40s NW %2d + %d  gather prio %d  %.4f ms  %.1f TF/s\n", name, NWM, NWG, prio_g, best, fl / best * 1e-9);
    if (!trace) return;
    std::vector<unsigned long long> t(4 * NW * 32 * 8);
    (void)hipMemcpy(t.data(), A.trace, t.size() * 8, hipMemcpyDeviceToHost);
    for (int wv : {0, NWM - 1, NWM}) {
        printf("  block 0 wave %d: per step cycles  matrix wave [mfma | wait B1 | P write | wait B2 | - | -], gather wave [gather | wait B1 | (wait B2 ->) | .. ]   (prologue -> first step %llu)\n", wv,
               t[(wv * 32 + 0) * 8 + 0] - t[(wv * 32 + 0) * 8 + 7]);
        for (int s = 0; s < 20; ++s) {
            const unsigned long long* q = &t[(wv * 32 + s) * 8];
            if (!q[4]) break;
            printf("    step %2d: %6llu %6llu %6llu %6llu   total %6llu\n", s, q[1] - q[0], q[2] - q[1], q[3] > q[2] ? q[3] - q[2] : 0ull, q[4] - (q[3] > q[2] ? q[3] : q[2]),
                   s ? q[4] - t[(wv * 32 + s - 1) * 8 + 4] : q[4] - q[7]);
        }
    }
    (void)hipFree((void*)A.x1); (void)hipFree((void*)A.x2); (void)hipFree(A.out);
}

int main() {
    for (int pr = 0; pr < 4; ++pr) run<2, 4, 8, 4>("ContextSkipNew d_h4 32x32 64|64 x512", 512, 256, 32, 32, pr, pr == 3);
    for (int pr = 0; pr < 4; pr += 3) run<2, 4, 8, 8>("ContextSkipNew d_h4 32x32 64|64 x512", 512, 256, 32, 32, pr, false);
    for (int pr = 0; pr < 4; pr += 3) run<2, 4, 8, 2>("ContextSkipNew d_h4 32x32 64|64 x512", 512, 256, 32, 32, pr, false);
    for (int pr = 0; pr < 4; pr += 3) run<1, 2, 8, 4>("ContextAEReal d_h4 36x64 32|32 x512", 512, 256, 36, 64, pr, pr == 3);
    for (int pr = 0; pr < 4; pr += 3) run<1, 2, 8, 8>("ContextAEReal d_h4 36x64 32|32 x512", 512, 256, 36, 64, pr, false);
    return 0;
}
