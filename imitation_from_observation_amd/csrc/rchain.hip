// rchain.hip -- the fully connected middle of ContextAEReal (arm_shaping.py:1634-1661: h4_lin, hz_lin, translate/trans_h0,
// translate/trans_z, deconv/d_h0_lin) as THREE launches instead of about forty.  Round 3.
//
// With featsize 100 these layers are 0.35 GFLOP forward -- nothing -- but each one is an implicit-GEMM launch plus a split-K
// combine (and, backward, two column-sum launches): 114 us of the forward's critical path and 245 us of the backward's on a
// 2.8 ms step, every launch a 15-20 us floor (profiles/archive/round3_b_timeline_context_ae_real.txt).  Every FC layer acts on a ROW, and
// the rows of one (target, source, context) triple never meet another triple's before the filter gradients, so:
//   rchain_fwd   a block takes TB = 2 triples (6 encoder rows) through all five layers, activations in LDS, weights read
//                straight from L2 (1.2 MB, shared by all blocks);
//   rchain_bwd   the same for the input-gradient chain d_h0_lin -> trans_z -> trans_h0 -> hz_lin -> h4_lin (transposed
//                weights staged through LDS in 32-column slices), with the simloss seed, the lrelu' masks and the skip gradients;
//   rchain_dw    the five Matrix gradients and the five bias gradients in ONE grouped launch (64 x 64 output tiles, rows summed in
//                a fixed order: deterministic).
// Plain f32 FMAs on the vector ALUs (exact f32 like the MFMA path; the summation order differs, results agree to rounding) -- the
// work is latency, not throughput.  Used when rchain_ok(): padded code width 128, one encoder, no dropout, exact f32.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "launch.h"

namespace ctx {

namespace {

constexpr int RC_T = 256;                        // threads per block
constexpr int RC_F = 128;                        // padded code width
constexpr int RC_TB = 2;                         // triples per block
constexpr int RC_RE = 3 * RC_TB;                 // encoder rows per block: [tgt | src | ctx] x TB

__device__ __forceinline__ float lrelu_f(float v) { return fmaxf(v, LEAK * v); }
__device__ __forceinline__ float dlrelu_f(float y) { return y >= 0.f ? 1.f : LEAK; }

// acc[r] += sum_k A[r][k] * W[k][j0 + jt], k in [0, K): A in LDS (row pitch ap floats, 16-byte aligned rows); W global [K][ldw] goes
// through the two-stage LDS ring Wl [2][KC * NJ] in chunks of KC = 8192 / NJ rows (NJ = 128 | 256 columns, 32 KB a stage): every
// thread has its eight float4 of the NEXT chunk in flight (registers) while it multiplies the current one -- a direct read of
// W[k][n] per FMA group left each of the K / 4 trips waiting a full L2 latency (0.15 ms for the forward chain).  All threads of
// the block call it together; one barrier per chunk.
template <int R, int NJ>
__device__ __forceinline__ void fma_cols(const float* As, int ap, const float* __restrict__ W, int ldw, int j0, int J, int K, float* Wl, int jt,
                                         float (&acc)[R]) {
    constexpr int KC = 8192 / NJ, NL = KC * NJ / 4 / RC_T;                      // rows per chunk; float4 per thread per chunk (8)
    const int tid = threadIdx.x;
    float4 pfa[NL], pfb[NL];                                                   // chunks c + 1 and c + 2 in flight: one chunk of FMAs is shorter than an L2 round trip
    auto issue = [&](int k0, float4 (&pf)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = tid + RC_T * u, kr = i / (NJ / 4), q = i - kr * (NJ / 4), k = k0 + kr, j = j0 + 4 * q;
            pf[u] = (k < K && j < J) ? ldg4(W + (int64_t)k * ldw + j) : zero4();
        }
    };
    auto land = [&](float* dst, const float4 (&pf)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) *reinterpret_cast<float4*>(&dst[(tid + RC_T * u) * 4]) = pf[u];
    };
    auto mul = [&](int k0, const float* stage) {
        const float* wl = stage + jt;
        const int kn = K - k0 < KC ? K - k0 : KC;
#pragma unroll 2
        for (int k = 0; k < kn; k += 4) {
            const float w0 = wl[(k + 0) * NJ], w1 = wl[(k + 1) * NJ], w2 = wl[(k + 2) * NJ], w3 = wl[(k + 3) * NJ];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 a = *reinterpret_cast<const float4*>(&As[r * ap + k0 + k]);
                acc[r] = fmaf(a.x, w0, acc[r]); acc[r] = fmaf(a.y, w1, acc[r]); acc[r] = fmaf(a.z, w2, acc[r]); acc[r] = fmaf(a.w, w3, acc[r]);
            }
        }
    };
    float* const s0 = Wl;
    float* const s1 = Wl + KC * NJ;
    issue(0, pfa);
    if (KC < K) issue(KC, pfb);
    __syncthreads();                                                            // the ring's previous user is done with it
    land(s0, pfa);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 2 * KC) {
        if (k0 + 2 * KC < K) issue(k0 + 2 * KC, pfa);
        mul(k0, s0);
        if (k0 + KC < K) land(s1, pfb);
        __syncthreads();
        if (k0 + KC >= K) break;
        if (k0 + 3 * KC < K) issue(k0 + 3 * KC, pfb);
        mul(k0 + KC, s1);
        if (k0 + 2 * KC < K) land(s0, pfa);
        __syncthreads();
    }
}

struct RcF {                                     // forward
    int B, D0p;
    const float* a3;                             // [3B][D0p]   flatten(h3) of [tgt | src | ctx]
    float* a4;                                   // [3B][128]   lrelu(h4_lin)
    float* Z;                                    // [4B][128]   rows [trans_z | tgt_z | src_z | ctx_z]
    float* th0;                                  // [B][128]
    float* dz;                                   // [2B][D0p]   lrelu(d_h0_lin([trans_z | tgt_z]))
    const float *W4, *b4, *Wz, *bz, *Wt0, *bt0, *Wtz, *btz, *Wd0, *bd0;
};

__global__ __launch_bounds__(RC_T) void rchain_fwd_kernel(const RcF P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ap = P.D0p + 4;
    float* As = smem;                            // [RE][D0p + 4]
    float* a4s = As + RC_RE * ap;                // [RE][132]
    float* zs = a4s + RC_RE * 132;               // [RE][132]
    float* th0s = zs + RC_RE * 132;              // [TB][132]
    float* dins = th0s + RC_TB * 132;            // [2 TB][132]: [trans_z | tgt_z]
    float* Wl = dins + 2 * RC_TB * 132;          // [2][8192] weight ring
    const int tid = threadIdx.x, n = tid & 127, hf = tid >> 7, b0 = blockIdx.x * RC_TB;
    auto img_of = [&](int e) { return (e / RC_TB) * P.B + b0 + (e % RC_TB); };        // encoder row e -> image of the stacked batch
    auto valid = [&](int e) { return b0 + (e % RC_TB) < P.B; };

    for (int i = tid; i < RC_RE * (P.D0p / 4); i += RC_T) {
        const int e = i / (P.D0p / 4), k4 = i - e * (P.D0p / 4);
        *reinterpret_cast<float4*>(&As[e * ap + 4 * k4]) = valid(e) ? ldg4(P.a3 + (int64_t)img_of(e) * P.D0p + 4 * k4) : zero4();
    }
    // h4_lin: thread (column n, half hf) takes rows 3 hf .. 3 hf + 2   (fma_cols' first barrier orders the tile above)
    {
        float acc[RC_RE / 2] = {};
        fma_cols<RC_RE / 2, 128>(As + hf * (RC_RE / 2) * ap, ap, P.W4, RC_F, 0, RC_F, P.D0p, Wl, n, acc);
        const float bb = P.b4[n];
#pragma unroll
        for (int r = 0; r < RC_RE / 2; ++r) {
            const int e = hf * (RC_RE / 2) + r;
            const float v = lrelu_f(acc[r] + bb);
            a4s[e * 132 + n] = v;
            if (valid(e)) P.a4[(int64_t)img_of(e) * RC_F + n] = v;
        }
    }
    // hz_lin (lrelu on every z): Z row B + image
    {
        float acc[RC_RE / 2] = {};
        fma_cols<RC_RE / 2, 128>(a4s + hf * (RC_RE / 2) * 132, 132, P.Wz, RC_F, 0, RC_F, RC_F, Wl, n, acc);
        const float bb = P.bz[n];
#pragma unroll
        for (int r = 0; r < RC_RE / 2; ++r) {
            const int e = hf * (RC_RE / 2) + r;
            const float v = lrelu_f(acc[r] + bb);
            zs[e * 132 + n] = v;
            if (valid(e)) P.Z[(int64_t)(P.B + img_of(e)) * RC_F + n] = v;
        }
    }
    // translate/trans_h0 on concat([src_z, ctx_z]): half hf = triple hf of the block
    {
        float acc[1] = {};
        fma_cols<1, 128>(zs + (RC_TB + hf) * 132, 132, P.Wt0, RC_F, 0, RC_F, RC_F, Wl, n, acc);
        fma_cols<1, 128>(zs + (2 * RC_TB + hf) * 132, 132, P.Wt0 + (int64_t)RC_F * RC_F, RC_F, 0, RC_F, RC_F, Wl, n, acc);
        const float v = lrelu_f(acc[0] + P.bt0[n]);
        th0s[hf * 132 + n] = v;
        if (b0 + hf < P.B) P.th0[(int64_t)(b0 + hf) * RC_F + n] = v;
    }
    // translate/trans_z (linear) -> Z row b; the decoder's inputs [trans_z | tgt_z]
    {
        float acc[1] = {};
        fma_cols<1, 128>(th0s + hf * 132, 132, P.Wtz, RC_F, 0, RC_F, RC_F, Wl, n, acc);
        const float v = acc[0] + P.btz[n];
        dins[hf * 132 + n] = v;
        dins[(RC_TB + hf) * 132 + n] = zs[hf * 132 + n];
        if (b0 + hf < P.B) P.Z[(int64_t)(b0 + hf) * RC_F + n] = v;
    }
    // deconv/d_h0_lin: 2 TB rows x D0p columns in groups of 256, column c0 + tid
    for (int c0 = 0; c0 < P.D0p; c0 += RC_T) {
        float acc[2 * RC_TB] = {};
        fma_cols<2 * RC_TB, 256>(dins, 132, P.Wd0, P.D0p, c0, P.D0p, RC_F, Wl, tid, acc);
        const int c = c0 + tid;
        if (c < P.D0p) {
            const float bb = P.bd0[c];
#pragma unroll
            for (int r = 0; r < 2 * RC_TB; ++r) {
                const int b = b0 + (r % RC_TB);
                if (b < P.B) P.dz[(int64_t)((r / RC_TB) * P.B + b) * P.D0p + c] = lrelu_f(acc[r] + bb);
            }
        }
    }
}

struct RcB {                                     // input-gradient chain
    int B, D0p;
    const float* dDz;                            // [2B][D0p]   gradient at d_h0_lin's output (its lrelu' already applied)
    const float* dsim2;                          // [2B][128]   simloss seed of [trans_z | tgt_z]
    float* dZ;                                   // [4B][128]   gradients of [trans_z | tgt_z | src_z | ctx_z]; rows >= B masked by lrelu'(Z)
    float* dth0;                                 // [B][128]
    float* dA4;                                  // [3B][128]
    float* dA3;                                  // [3B][D0p]
    const float *Z, *th0, *a4, *a3;
    const float* dSk3;                           // [2B][D0p] skip gradients of the two decoder passes (added to the ctx rows)
    const float *W4, *Wz, *Wt0, *Wtz, *Wd0;
};

// acc[r] += sum_c Y[r][c] * W[j0 + jt][c], c in [0, C): W global [.][ldw] (a row = one output), staged through the two-stage ring
// Wl [2][NJ][33] in slices of 32 columns with the next slice's float4s in flight in registers (as fma_cols).  All threads of the
// block call it together; rows of Y are in LDS (pitch yp).
template <int R, int NJ>
__device__ __forceinline__ void fma_rows_t(const float* Ys, int yp, const float* __restrict__ W, int ldw, int j0, int J, int C, float* Wl, int jt,
                                           float (&acc)[R]) {
    constexpr int NL = NJ * 8 / RC_T;                                            // float4 per thread per slice (4 | 8)
    const int tid = threadIdx.x;
    float4 pfa[NL], pfb[NL];
    auto issue = [&](int c0, float4 (&pf)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = tid + RC_T * u, jr = i >> 3, q = i & 7, j = j0 + jr;
            pf[u] = j < J ? ldg4(W + (int64_t)j * ldw + c0 + 4 * q) : zero4();
        }
    };
    auto land = [&](float* dst, const float4 (&pf)[NL]) {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int i = tid + RC_T * u, jr = i >> 3, q = i & 7;
            float* d = &dst[jr * 33 + 4 * q];
            d[0] = pf[u].x; d[1] = pf[u].y; d[2] = pf[u].z; d[3] = pf[u].w;
        }
    };
    auto mul = [&](int c0, const float* stage) {
        const float* wr = stage + jt * 33;
#pragma unroll 2
        for (int cc = 0; cc < 32; cc += 4) {
            const float w0 = wr[cc], w1 = wr[cc + 1], w2 = wr[cc + 2], w3 = wr[cc + 3];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 y = *reinterpret_cast<const float4*>(&Ys[r * yp + c0 + cc]);
                acc[r] = fmaf(y.x, w0, acc[r]); acc[r] = fmaf(y.y, w1, acc[r]); acc[r] = fmaf(y.z, w2, acc[r]); acc[r] = fmaf(y.w, w3, acc[r]);
            }
        }
    };
    float* const s0 = Wl;
    float* const s1 = Wl + NJ * 33;
    issue(0, pfa);
    if (32 < C) issue(32, pfb);
    __syncthreads();                                                            // the ring's previous user is done with it (and Ys is written)
    land(s0, pfa);
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 64) {
        if (c0 + 64 < C) issue(c0 + 64, pfa);
        mul(c0, s0);
        if (c0 + 32 < C) land(s1, pfb);
        __syncthreads();
        if (c0 + 32 >= C) break;
        if (c0 + 96 < C) issue(c0 + 96, pfb);
        mul(c0 + 32, s1);
        if (c0 + 64 < C) land(s0, pfa);
        __syncthreads();
    }
}

__global__ __launch_bounds__(RC_T) void rchain_bwd_kernel(const RcB P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int yp = P.D0p + 4;
    float* Ys = smem;                            // [2 TB][D0p + 4]   dDz rows [pass 1 | pass 2]
    float* Wl = Ys + 2 * RC_TB * yp;             // [2][256][33] weight ring
    float* dzs = Wl + 2 * 256 * 33;              // [2 TB][132]       d [trans_z | tgt_z]
    float* dts = dzs + 2 * RC_TB * 132;          // [TB][132]         d th0
    float* des = dts + RC_TB * 132;              // [RE][132]         d [tgt_z | src_z | ctx_z], masked
    float* d4s = des + RC_RE * 132;              // [RE][132]         d h4
    const int tid = threadIdx.x, n = tid & 127, hf = tid >> 7, b0 = blockIdx.x * RC_TB;
    auto img_of = [&](int e) { return (e / RC_TB) * P.B + b0 + (e % RC_TB); };
    auto valid = [&](int e) { return b0 + (e % RC_TB) < P.B; };

    for (int i = tid; i < 2 * RC_TB * (P.D0p / 4); i += RC_T) {
        const int r = i / (P.D0p / 4), k4 = i - r * (P.D0p / 4), b = b0 + (r % RC_TB);
        *reinterpret_cast<float4*>(&Ys[r * yp + 4 * k4]) = b < P.B ? ldg4(P.dDz + (int64_t)((r / RC_TB) * P.B + b) * P.D0p + 4 * k4) : zero4();
    }
    // d_h0_lin: d [trans_z | tgt_z] = dDz Wd0^T + simloss seed; thread (column n, half hf) takes rows 2 hf, 2 hf + 1 of the 2 TB
    {
        float acc[RC_TB] = {};
        fma_rows_t<RC_TB, 128>(Ys + hf * RC_TB * yp, yp, P.Wd0, P.D0p, 0, RC_F, P.D0p, Wl, n, acc);
#pragma unroll
        for (int r = 0; r < RC_TB; ++r) {
            const int row = hf * RC_TB + r, b = b0 + (row % RC_TB);
            float v = acc[r];
            if (b < P.B) v += P.dsim2[(int64_t)((row / RC_TB) * P.B + b) * RC_F + n];
            dzs[row * 132 + n] = v;
            if (b < P.B && row < RC_TB) P.dZ[(int64_t)b * RC_F + n] = v;        // d trans_z (the tgt_z rows leave masked, below)
        }
    }
    // trans_z: d th0 = (d trans_z Wtz^T) * lrelu'(th0); half hf = triple hf
    {
        float acc[1] = {};
        fma_rows_t<1, 128>(dzs + hf * 132, 132, P.Wtz, RC_F, 0, RC_F, RC_F, Wl, n, acc);
        const int b = b0 + hf;
        const float v = b < P.B ? acc[0] * dlrelu_f(P.th0[(int64_t)b * RC_F + n]) : 0.f;
        dts[hf * 132 + n] = v;
        if (b < P.B) P.dth0[(int64_t)b * RC_F + n] = v;
    }
    // trans_h0: d concat([src_z, ctx_z]) = d th0 Wt0^T: output j = tid of 256, both triples
    {
        float acc[RC_TB] = {};
        fma_rows_t<RC_TB, 256>(dts, 132, P.Wt0, RC_F, 0, 2 * RC_F, RC_F, Wl, tid, acc);
        // j < 128: d src_z (encoder row TB + i), else d ctx_z (row 2 TB + i); masked by lrelu'(z) like the tgt_z rows below
#pragma unroll
        for (int i = 0; i < RC_TB; ++i) {
            const int e = (1 + (tid >> 7)) * RC_TB + i, b = b0 + i;
            float v = 0.f;
            if (b < P.B) {
                v = acc[i] * dlrelu_f(P.Z[(int64_t)(P.B + img_of(e)) * RC_F + n]);
                P.dZ[(int64_t)(P.B + img_of(e)) * RC_F + n] = v;
            }
            des[e * 132 + n] = v;
        }
    }
    // tgt_z rows: the decoder's gradient + seed, masked
    if (tid < RC_TB * RC_F) {
        const int i = tid >> 7, b = b0 + i;
        float v = 0.f;
        if (b < P.B) {
            v = dzs[(RC_TB + i) * 132 + n] * dlrelu_f(P.Z[(int64_t)(P.B + b) * RC_F + n]);
            P.dZ[(int64_t)(P.B + b) * RC_F + n] = v;
        }
        des[i * 132 + n] = v;
    }
    // hz_lin: d h4 = (d z Wz^T) * lrelu'(h4); rows 3 hf .. 3 hf + 2   (fma_rows_t's first barrier orders the writes above)
    {
        float acc[RC_RE / 2] = {};
        fma_rows_t<RC_RE / 2, 128>(des + hf * (RC_RE / 2) * 132, 132, P.Wz, RC_F, 0, RC_F, RC_F, Wl, n, acc);
#pragma unroll
        for (int r = 0; r < RC_RE / 2; ++r) {
            const int e = hf * (RC_RE / 2) + r;
            float v = 0.f;
            if (valid(e)) {
                v = acc[r] * dlrelu_f(P.a4[(int64_t)img_of(e) * RC_F + n]);
                P.dA4[(int64_t)img_of(e) * RC_F + n] = v;
            }
            d4s[e * 132 + n] = v;
        }
    }
    // h4_lin: d flatten(h3) = (d h4 W4^T + skip gradients on the ctx rows) * lrelu'(h3): D0p outputs in groups of 256
    for (int j0 = 0; j0 < P.D0p; j0 += 256) {
        float acc[RC_RE] = {};
        fma_rows_t<RC_RE, 256>(d4s, 132, P.W4, RC_F, j0, P.D0p, RC_F, Wl, tid, acc);
        const int j = j0 + tid;
        if (j < P.D0p) {
#pragma unroll
            for (int e = 0; e < RC_RE; ++e) {
                if (!valid(e)) continue;
                float v = acc[e];
                if (e / RC_TB == 2) {
                    const int b = b0 + (e % RC_TB);
                    v += P.dSk3[(int64_t)b * P.D0p + j];
                    v += P.dSk3[(int64_t)(P.B + b) * P.D0p + j];
                }
                const int64_t o = (int64_t)img_of(e) * P.D0p + j;
                P.dA3[o] = v * dlrelu_f(P.a3[o]);
            }
        }
    }
}

// ---- grouped Matrix / bias gradients: dW[k][n] = sum_r X[r][k] dY[r][n], db[n] = sum_r dY[r][n]
struct RcProb {
    const float* X; int ldx;                     // [rows][K] (columns < ksplit), then X2 [rows][K - ksplit]
    const float* X2; int ldx2; int ksplit;
    const float* dY; int ldy;
    int rows, K, N;
    float* dW;                                   // [K][N]
    float* db;                                   // [N]
    int tile0;                                   // first block of this problem
};
struct RcW { RcProb p[5]; int nprob; };

__global__ __launch_bounds__(RC_T) void rchain_dw_kernel(const RcW A) {
    __shared__ __attribute__((aligned(16))) float Xs[32][68];
    __shared__ __attribute__((aligned(16))) float Ys[32][68];
    __shared__ float cs[4][64];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i) if (i < A.nprob && (int)blockIdx.x >= A.p[i].tile0) pi = i;
    RcProb P;                                    // (static indices: no scratch copy of the argument struct)
    switch (pi) { case 0: P = A.p[0]; break; case 1: P = A.p[1]; break; case 2: P = A.p[2]; break; case 3: P = A.p[3]; break; default: P = A.p[4]; break; }
    const int t = blockIdx.x - P.tile0, gn = (P.N + 63) / 64, tk = t / gn, tn = t - tk * gn;
    const int k0 = tk * 64, n0 = tn * 64;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;                   // micro-tile rows k0 + 4 ty .., columns n0 + 4 tx ..
    const int lr = tid >> 4, lq = tid & 15;                                      // loader: row lr (+ 16), float4 lq of the 64 columns
    float acc[4][4] = {};
    float csum = 0.f;
    for (int r0 = 0; r0 < P.rows; r0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = r0 + lr + 16 * u, k = k0 + 4 * lq, c = n0 + 4 * lq;
            float4 xv = zero4(), yv = zero4();
            if (r < P.rows) {
                if (k < P.K) xv = k < P.ksplit ? ldg4(P.X + (int64_t)r * P.ldx + k) : ldg4(P.X2 + (int64_t)r * P.ldx2 + (k - P.ksplit));
                if (c < P.N) yv = ldg4(P.dY + (int64_t)r * P.ldy + c);
            }
            *reinterpret_cast<float4*>(&Xs[lr + 16 * u][4 * lq]) = xv;
            *reinterpret_cast<float4*>(&Ys[lr + 16 * u][4 * lq]) = yv;
        }
        __syncthreads();
#pragma unroll 4
        for (int rr = 0; rr < 32; ++rr) {
            const float4 x = *reinterpret_cast<const float4*>(&Xs[rr][4 * ty]);
            const float4 y = *reinterpret_cast<const float4*>(&Ys[rr][4 * tx]);
            const float xa[4] = {x.x, x.y, x.z, x.w}, ya[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xa[i], ya[j], acc[i][j]);
        }
        if (tk == 0) {                                                           // bias gradient: thread = (column tid % 64, row phase tid / 64)
            const int c = tid & 63, ph = tid >> 6;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) csum += Ys[ph * 8 + rr][c];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + 4 * ty + i, c = n0 + 4 * tx;
        if (k < P.K && c < P.N) *reinterpret_cast<float4*>(P.dW + (int64_t)k * P.N + c) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    if (tk == 0) {
        cs[tid >> 6][tid & 63] = csum;
        __syncthreads();
        if (tid < 64 && n0 + tid < P.N) P.db[n0 + tid] = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
    }
}

}  // namespace

bool rchain_ok(int Fp, int64_t D0p, int nset, bool drop, int prec) {
    return opt(OPT_RCHAIN) && Fp == RC_F && nset == 1 && !drop && prec == 0 && D0p % 64 == 0 && D0p >= 64 && D0p <= 4096;
}

static size_t rc_fwd_lds(int D0p) { return (size_t)(RC_RE * (D0p + 4) + (2 * RC_RE + RC_TB + 2 * RC_TB) * 132 + 2 * 8192) * sizeof(float); }
static size_t rc_bwd_lds(int D0p) { return (size_t)(2 * RC_TB * (D0p + 4) + 2 * 256 * 33 + (2 * RC_TB + RC_TB + 2 * RC_RE) * 132) * sizeof(float); }

void rchain_fwd(hipStream_t s, int B, int D0p, const float* a3, float* a4, float* Z, float* th0, float* dz, const float* const W[10]) {
    RcF P{B, D0p, a3, a4, Z, th0, dz, W[0], W[1], W[2], W[3], W[4], W[5], W[6], W[7], W[8], W[9]};
    const size_t lds = rc_fwd_lds(D0p);
    ensure_dyn_lds((const void*)rchain_fwd_kernel, lds);
    hipLaunchKernelGGL(rchain_fwd_kernel, dim3((unsigned)((B + RC_TB - 1) / RC_TB)), dim3(RC_T), lds, s, P);
}

void rchain_bwd(hipStream_t s, int B, int D0p, const float* dDz, const float* dsim2, float* dZ, float* dth0, float* dA4, float* dA3, const float* Z,
                const float* th0, const float* a4, const float* a3, const float* dSk3, const float* const W[10]) {
    RcB P{B, D0p, dDz, dsim2, dZ, dth0, dA4, dA3, Z, th0, a4, a3, dSk3, W[0], W[2], W[4], W[6], W[8]};
    const size_t lds = rc_bwd_lds(D0p);
    ensure_dyn_lds((const void*)rchain_bwd_kernel, lds);
    hipLaunchKernelGGL(rchain_bwd_kernel, dim3((unsigned)((B + RC_TB - 1) / RC_TB)), dim3(RC_T), lds, s, P);
}

// G[2 i] / G[2 i + 1]: Matrix / bias gradient of layer i in the order h4_lin, hz_lin, trans_h0, trans_z, d_h0_lin
void rchain_dw(hipStream_t s, int B, int D0p, const float* a3, const float* a4, const float* Z, const float* th0, const float* dA4, const float* dZ,
               const float* dth0, const float* dDz, float* const G[10]) {
    RcW A{};
    A.nprob = 5;
    const float* dSz = dZ + (int64_t)B * RC_F;                                   // d [tgt_z | src_z | ctx_z], masked
    A.p[0] = RcProb{a3, D0p, nullptr, 0, D0p, dA4, RC_F, 3 * B, D0p, RC_F, G[0], G[1], 0};
    A.p[1] = RcProb{a4, RC_F, nullptr, 0, RC_F, dSz, RC_F, 3 * B, RC_F, RC_F, G[2], G[3], 0};
    A.p[2] = RcProb{Z + 2ll * B * RC_F, RC_F, Z + 3ll * B * RC_F, RC_F, RC_F, dth0, RC_F, B, 2 * RC_F, RC_F, G[4], G[5], 0};
    A.p[3] = RcProb{th0, RC_F, nullptr, 0, RC_F, dZ, RC_F, B, RC_F, RC_F, G[6], G[7], 0};
    A.p[4] = RcProb{Z, RC_F, nullptr, 0, RC_F, dDz, D0p, 2 * B, RC_F, D0p, G[8], G[9], 0};
    int tiles = 0;
    for (int i = 0; i < 5; ++i) {
        A.p[i].tile0 = tiles;
        tiles += ((A.p[i].K + 63) / 64) * ((A.p[i].N + 63) / 64);
    }
    hipLaunchKernelGGL(rchain_dw_kernel, dim3((unsigned)tiles), dim3(RC_T), 0, s, A);
}

}  // namespace ctx
