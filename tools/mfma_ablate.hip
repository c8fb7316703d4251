// tools/mfma_ablate.hip -- ablation of the igemm main loop (development tool, not shipped).
// Builds variants of the 128x128x32 f32-MFMA block loop with pieces removed and prints TF/s each,
// to see where the gap between the measured ~114 TF and the ~152 TF pipe peak goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I imitation_from_observation_amd/csrc tools/mfma_ablate.hip -o /tmp/mfma_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "igemm.h"
using namespace ctx;

enum { F_NOGLOBAL = 1, F_NOLDSSTORE = 2, F_NOBARRIER = 4, F_NOFRAG = 8, F_DBUF = 16, F_SAMEADDR = 32, F_LATEUSE = 64 };

template <int FLAGS>
__global__ __launch_bounds__(NTHREADS) void k(const KmPlain la, const NmPlain lb, float* out, int M, int N, int nch, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    constexpr int MI = 2, NI = 2, TM = 128, TN = 128;
    using TA = Tile<true, TM>;
    using TB = Tile<false, TN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = TA::FLOATS + TB::FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1, l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    Fetch<KmPlain, TM> fa; Fetch<NmPlain, TN> fb;
    fa.init(la, 0, m0, tid); fb.init(lb, 0, n0, tid);
    f32x16 acc[MI][NI];
    for (int mi = 0; mi < MI; ++mi) for (int ni = 0; ni < NI; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    float4 ra[4], rb[4];
    for (int p = 0; p < 4; ++p) { ra[p] = make_float4(1.f, 2.f, 3.f, 4.f); rb[p] = make_float4(.1f, .2f, .3f, .4f); }
    if (!(FLAGS & F_NOGLOBAL)) { for (int p = 0; p < 4; ++p) { ra[p] = fa.load1(la, 0, m0, 0, tid, p); rb[p] = fb.load1(lb, 0, n0, 0, tid, p); } }
    if (FLAGS & F_DBUF) {
        float* sA = smem; float* sB = smem + TA::FLOATS;
        for (int p = 0; p < 4; ++p) { TA::store(sA, tid, p, ra[p]); TB::store(sB, tid, p, rb[p]); }
        __syncthreads();
        if (!(FLAGS & F_NOGLOBAL)) { for (int p = 0; p < 4; ++p) { ra[p] = fa.load1(la, 0, m0, 1, tid, p); rb[p] = fb.load1(lb, 0, n0, 1, tid, p); } }
    }
    for (int c = 0; c < nch; ++c) {
        float4 ta[4], tb[4];
        float* sA = smem + ((FLAGS & F_DBUF) ? (c & 1) * STAGE : 0);
        float* sB = sA + TA::FLOATS;
        if (FLAGS & F_DBUF) {
            float* nA = smem + ((c + 1) & 1) * STAGE; float* nB = nA + TA::FLOATS;
            for (int p = 0; p < 4; ++p) { TA::store(nA, tid, p, ra[p]); TB::store(nB, tid, p, rb[p]); }
            if (!(FLAGS & F_NOGLOBAL) && c + 2 < nch) { for (int p = 0; p < 4; ++p) { ra[p] = fa.load1(la, 0, m0, c + 2, tid, p); rb[p] = fb.load1(lb, 0, n0, c + 2, tid, p); } }
        } else {
            if (!(FLAGS & F_NOLDSSTORE)) for (int p = 0; p < 4; ++p) { TA::store(sA, tid, p, ra[p]); TB::store(sB, tid, p, rb[p]); }
            if (!(FLAGS & F_NOBARRIER)) __syncthreads();
            if (!(FLAGS & F_NOGLOBAL) && c + 1 < nch) {
                const int cc = (FLAGS & F_SAMEADDR) ? 0 : c + 1;
                for (int p = 0; p < 4; ++p) {
                    if (FLAGS & F_LATEUSE) { ta[p] = fa.load1(la, 0, m0, cc, tid, p); tb[p] = fb.load1(lb, 0, n0, cc, tid, p); }
                    else { ra[p] = fa.load1(la, 0, m0, cc, tid, p); rb[p] = fb.load1(lb, 0, n0, cc, tid, p); }
                }
            }
        }
        float a[2][MI][4], b[2][NI][4];
        if (FLAGS & F_NOFRAG) {
            for (int mi = 0; mi < MI; ++mi) for (int t = 0; t < 4; ++t) { a[0][mi][t] = a[1][mi][t] = ra[mi].x + t; }
            for (int ni = 0; ni < NI; ++ni) for (int t = 0; t < 4; ++t) { b[0][ni][t] = b[1][ni][t] = rb[ni].y + t; }
        } else {
            for (int mi = 0; mi < MI; ++mi) TA::frag(sA, wm * 64 + mi * 32 + l31, 0, h, a[0][mi]);
            for (int ni = 0; ni < NI; ++ni) TB::frag(sB, wn * 64 + ni * 32 + l31, 0, h, b[0][ni]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3 && !(FLAGS & F_NOFRAG)) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) TA::frag(sA, wm * 64 + mi * 32 + l31, q + 1, h, a[(q + 1) & 1][mi]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) TB::frag(sB, wn * 64 + ni * 32 + l31, q + 1, h, b[(q + 1) & 1][ni]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][mi][t], b[q & 1][ni][t], acc[mi][ni], 0, 0, 0);
        }
        if ((FLAGS & F_LATEUSE) && !(FLAGS & F_DBUF) && c + 1 < nch)
            for (int p = 0; p < 4; ++p) { asm volatile("" :: "v"(ta[p].x), "v"(ta[p].y), "v"(ta[p].z), "v"(ta[p].w), "v"(tb[p].x), "v"(tb[p].y), "v"(tb[p].z), "v"(tb[p].w)); }
        if (!(FLAGS & F_NOBARRIER)) __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        clk[2 * b] = __builtin_readcyclecounter() - t0;
        clk[2 * b + 1] = wall_clock64() - w0;
    }
    for (int mi = 0; mi < MI; ++mi) for (int r = 0; r < 16; ++r) for (int ni = 0; ni < NI; ++ni) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + wn * 64 + ni * 32 + l31;
        out[(int64_t)m * N + n] = acc[mi][ni][r];
    }
}

static unsigned long long* g_clk = nullptr;
template <int FLAGS>
void run(const char* name, const KmPlain& a, const NmPlain& b, float* out, int M, int N, int K) {
    const size_t lds = (size_t)(Tile<true, 128>::FLOATS + Tile<false, 128>::FLOATS) * 4 * ((FLAGS & F_DBUF) ? 2 : 1);
    hipFuncSetAttribute((const void*)k<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(M / 128, N / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<FLAGS>, grid, dim3(256), lds, 0, a, b, out, M, N, K / 32, g_clk);
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k<FLAGS>, grid, dim3(256), lds, 0, a, b, out, M, N, K / 32, g_clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    std::vector<unsigned long long> hc(2 * grid.x * grid.y);
    hipMemcpy(hc.data(), g_clk, hc.size() * 8, hipMemcpyDeviceToHost);
    double sc = 0, sw = 0;
    for (size_t i = 0; i < hc.size(); i += 2) { sc += (double)hc[i]; sw += (double)hc[i + 1]; }
    printf("%-52s %8.3f ms %7.1f TF/s  shader clock %.3f GHz (s_memtime/wall_clock64 @100MHz)  block time %.1f us\n", name, ms,
           2.0 * M * N * K / ms / 1e9, sc / sw * 0.1, sw / (hc.size() / 2) * 0.01);
}

int main() {
    hipMalloc(&g_clk, 2 * 8 * 65536);
    struct Shape { int M, N, K; } shapes[] = {{8192, 512, 6400}, {131072, 128, 1600}, {32768, 256, 3200}};
    for (auto s : shapes) {
        float *A, *B, *C;
        hipMalloc(&A, (size_t)s.M * s.K * 4); hipMalloc(&B, (size_t)s.K * s.N * 4); hipMalloc(&C, (size_t)s.M * s.N * 4);
        std::vector<float> hA((size_t)s.M * s.K), hB((size_t)s.K * s.N);
        unsigned x = 1234567;
        for (auto& v : hA) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
        for (auto& v : hB) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        float* Z; hipMalloc(&Z, 256); hipMemset(Z, 0, 256);
        KmPlain a{A, s.K, nullptr, 0, s.K, s.M, s.K / 32, Z};
        NmPlain b{B, s.N, nullptr, 0, s.N, s.N, s.K, Z};
        printf("--- M=%d N=%d K=%d\n", s.M, s.N, s.K);
        run<0>("full (as shipped)", a, b, C, s.M, s.N, s.K);
        run<F_DBUF>("double-buffered LDS, 1 barrier", a, b, C, s.M, s.N, s.K);
        run<F_SAMEADDR>("loads always from chunk 0 (cache-resident)", a, b, C, s.M, s.N, s.K);
        run<F_LATEUSE>("loads consumed only at chunk end (LDS gets stale regs)", a, b, C, s.M, s.N, s.K);
        run<F_LATEUSE | F_SAMEADDR>("late use + cache-resident", a, b, C, s.M, s.N, s.K);
        run<F_NOGLOBAL>("no global loads", a, b, C, s.M, s.N, s.K);
        run<F_NOGLOBAL | F_NOLDSSTORE>("no global, no LDS store", a, b, C, s.M, s.N, s.K);
        run<F_NOGLOBAL | F_NOLDSSTORE | F_NOBARRIER>("no global, no LDS store, no barrier", a, b, C, s.M, s.N, s.K);
        run<F_NOGLOBAL | F_NOLDSSTORE | F_NOBARRIER | F_NOFRAG>("MFMA only", a, b, C, s.M, s.N, s.K);
        run<F_NOGLOBAL | F_DBUF>("dbuf, no global loads", a, b, C, s.M, s.N, s.K);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
