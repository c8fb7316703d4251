// convt3m.hip -- conv2d_transpose 5x5 (stride 1 | 2, SAME) from the wide concat [decoder | ctx skip] to the THREE image channels on the
// MATRIX cores: d_h4 forward of every translator (gym/envs/mujoco/arm_shaping.py:1329-1330 ContextSkipNew, :1671-1672 ContextAEReal;
// deconv2d :62-85) at training batch sizes.
//
// Three output channels fill 3 of an MFMA tile's 16 columns -- but a transposed convolution is a SCATTER: input pixel (i, j) adds
// in[i][j][:] . w[ky][kx][c][:] to output (S i + ky - pad, S j + kx - pad, c).  So the product that fills the tile is
//       P[pixel][(ky, kx, c)] = sum_k in[pixel][k] * w[ky][kx][c][k]          M = pixels, K = c1 + c2, N = 25 taps x 3 = 75
// with the columns ordered (ky)(kx, c): one 16-column MFMA block per filter ROW, 15 of 16 columns used (94 %).  convt3.hip computes
// the same layer on the vector ALUs from an LDS halo tile in 8-channel slices: 0.169 ms for the 64x64 ContextSkipNew launch, 959 MB of
// HBM traffic for 293 MB of tensors (every 128-B pixel line fetched once per slice).  Here:
//   * A operand straight from global memory into registers, ONE pass over the input: v_mfma_f32_16x16x4_f32 takes A[row = lane % 16]
//     [k = lane / 16]; the K order of a product is free as long as A and B agree, so lane (p, g) loads the float4 in[pixel p][16 q + 4 g
//     .. + 3] (whole 64-B segments of each pixel line per instruction) and MFMA (q, t) contracts k = 16 q + 4 g + t.  The next step's
//     loads are in flight under the current step's MFMAs; no LDS traffic for A at all;
//   * B operand (the whole filter, K x 80 floats) resident in LDS for the block's lifetime in exactly that (q, ky, lane, t) order: one
//     conflict-free ds_read_b128 per (q, ky) feeds four MFMAs;
//   * P never leaves the CU: a wave writes its 16 pixels x 75 products into an LDS ring of pixels; once an output row has all its
//     contributing input rows in the ring, the block sums the 4 / 6 / 9 (stride 1: 25) taps that land on each output pixel in a FIXED
//     order (deterministic, no atomics), adds the bias and stores full-width NHWC rows;
//   * a block owns whole images (no halo: nothing outside the image contributes) and is persistent over images, so the filter is
//     staged once and the prefetch runs across image boundaries.
// Exact f32 (the MFMA is an fmaf chain per output; the tap sum is f32 adds in a fixed order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "launch.h"
#include "options.h"

namespace ctx {

namespace {

typedef float m3_f4 __attribute__((ext_vector_type(4)));

constexpr int M3_PST = 75;                         // floats per pixel of the P ring: column ky * 15 + kx * 3 + c (odd: lanes on consecutive pixels hit distinct banks)

struct Ct3m {
    const float* x1;                               // decoder stream [nimg][hin][win][C1]
    const float* x2; int nmod2;                    // ctx skip [nmod2][hin][win][C1], image index img % nmod2
    int hin, win, nimg, npix;
    const float* w;                                // [25][3][2 * C1]  (the reference's [5, 5, out, in])
    const float* bias;                             // [3]
    float* out;                                    // [nimg][S hin][S win][3]
    int RP;                                        // pixels in the P ring
    int steps;                                     // steps per image = ceil(npix / (16 NW))
#ifdef M3_TRACE
    int prio_g;                                    // experiment: s_setprio of the gather waves
    unsigned long long* trace;                     // tools/convt3m_bench.hip: s_memtime stamps [block < 4][wave][step < 32][8] (never in the product build)
#endif
};

#ifdef M3_TRACE
#define M3_STAMP(k) do { if (blockIdx.x < 4 && lane == 0 && nstep < 32) A.trace[((blockIdx.x * (NWM + NWG) + wave) * 32 + nstep) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define M3_STAMP(k) do {} while (0)
#endif

// KQH: 16-channel groups per input tensor (C1 = 16 KQH).  NWM matrix waves, each owning 16 consecutive pixels of a step of 16 NWM
// pixels, + NWG gather waves: while the matrix waves form step s + 1, the gather waves sum step s's products out of the ring into
// output rows (the matrix pipe and the vector / LDS pipes of a CU run side by side).  Two barriers per step:
//     matrix waves:  prefetch A(s + 1) | MFMA(s) | B1 | write P(s) into the ring | B2
//     gather waves:  gather(s - 1)               | B1 |                          | B2          (+ gather(last) behind the loop)
// B1: every read of gather(s - 1) is done, so the ring slots of P(s) (which alias rows gather(s - 1) needed) may be rewritten;
// B2: P(s) is complete.
template <int S, int KQH, int NWM, int NWG>
__global__ __launch_bounds__((NWM + NWG) * 64) void convt3m_kernel(const Ct3m A) {
    constexpr int KQ = 2 * KQH, C1 = 16 * KQH, CI = 2 * C1, SP = NWM * 16, NT = (NWM + NWG) * 64, NG = NWG * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    m3_f4* Wl = reinterpret_cast<m3_f4*>(smem);                // [KQ][5][64 lanes] float4: w[(ky * 15 + n)][k = 16 q + 4 g + t], lane = 16 g + n
    float* P = smem + KQ * 5 * 256;                            // [RP][M3_PST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nstep = 0; (void)nstep;
    M3_STAMP(7);

    for (int i = tid; i < KQ * 5 * 64; i += NT) {              // the filter in MFMA order (once per block): 16 contiguous bytes of a filter row per entry
        const int l = i & 63, qk = i >> 6, ky = qk % 5, q = qk / 5, n = l & 15;
        Wl[i] = n < 15 ? *reinterpret_cast<const m3_f4*>(A.w + (int64_t)(ky * 15 + n) * CI + 16 * q + 4 * (l >> 4)) : m3_f4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < M3_PST) P[A.RP * M3_PST + tid] = 0.f;            // the zero pixel behind the ring
    __syncthreads();                                           // Wl complete

    if (wave < NWM) {
        // ================================================================================================ matrix waves
        const int p = lane & 15, g = lane >> 4;
        m3_f4 a[KQ], an[KQ];
        auto issue = [&](int img, int st, m3_f4* dst) {
            int pi = st * SP + wave * 16 + p;
            pi = pi < A.npix ? pi : A.npix - 1;                // rows past the image: a pixel that exists (their products are never gathered)
            const float* s1 = A.x1 + ((int64_t)img * A.npix + pi) * C1 + 4 * g;
            const float* s2 = A.x2 + ((int64_t)(img % A.nmod2) * A.npix + pi) * C1 + 4 * g;
#pragma unroll
            for (int q = 0; q < KQH; ++q) {
                dst[q] = *reinterpret_cast<const m3_f4*>(s1 + 16 * q);
                dst[KQH + q] = *reinterpret_cast<const m3_f4*>(s2 + 16 * q);
            }
        };
        int img = blockIdx.x, st = 0;
        if (img < A.nimg) issue(img, st, a);
        while (img < A.nimg) {
            int img2 = img, st2 = st + 1;
            if (st2 == A.steps) { st2 = 0; img2 += gridDim.x; }
            // UNCONDITIONAL prefetch (past the block's last step: the current one again, out of L2): with the loads under a branch the
            // compiler's wait counts must cover the not-taken path, and the first MFMA waited for the prefetch it should run under
            issue(img2 < A.nimg ? img2 : img, img2 < A.nimg ? st2 : st, an);
            __builtin_amdgcn_sched_barrier(0);                 // (the scheduler otherwise sinks the loads below the MFMA loop to shorten their live ranges)
            M3_STAMP(0);

            m3_f4 acc[5];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) acc[ky] = m3_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                m3_f4 b[5];
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) b[ky] = Wl[(q * 5 + ky) * 64 + lane];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) acc[ky] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][t], b[ky][t], acc[ky], 0, 0, 0);
            }
            M3_STAMP(1);
            __syncthreads();                                   // B1
            M3_STAMP(2);
            {   // lane (g, n): acc[ky][r] = P[pixel 4 g + r of the wave's 16][ky * 15 + n]
                const int pb = st * SP + wave * 16 + 4 * g;
                int slot = pb % A.RP;
                if (p < 15) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (pb + r < A.npix) {
                            float* dst = P + slot * M3_PST + p;
#pragma unroll
                            for (int ky = 0; ky < 5; ++ky) dst[ky * 15] = acc[ky][r];
                        }
                        slot = slot + 1 == A.RP ? 0 : slot + 1;
                    }
                }
            }
            M3_STAMP(3);
            __syncthreads();                                   // B2
            M3_STAMP(4);
#pragma unroll
            for (int q = 0; q < KQ; ++q) a[q] = an[q];
            ++nstep;
            img = img2; st = st2;
        }
    } else {
        // ================================================================================================ gather waves
        const int gt = tid - NWM * 64;
#ifdef M3_TRACE
        if (A.prio_g == 1) __builtin_amdgcn_s_setprio(1); else if (A.prio_g == 2) __builtin_amdgcn_s_setprio(2); else if (A.prio_g == 3) __builtin_amdgcn_s_setprio(3);
#else
        __builtin_amdgcn_s_setprio(1);                       // (tools/convt3m_bench.hip: 0.1176 ms at 0, 0.1154-0.1160 at 1..3)
#endif
        const int wout = S * A.win, hout = S * A.hin;
        // one thread = one output row y of one INPUT column j (stride 2: the pixels x = 2 j, 2 j + 1; stride 1: x = j); NG / win rows per pass
        const int RPP = NG / A.win;                            // (win <= NG: convt3_mfma_ok)
        const int rofs = gt / A.win, j = gt - rofs * A.win;
        const float bias0 = A.bias[0], bias1 = A.bias[1], bias2 = A.bias[2];
        const int ZS = A.RP * M3_PST;                          // the all-zero pixel behind the ring: where taps outside the image (or outside the
                                                               // row's parity class) read, so the sums below carry no selects
        // output rows of (img, st): those whose contributing input rows are all in the ring once step st has landed
        auto gather = [&](int img, int st) {
            const int done = (st + 1) * SP < A.npix ? (st + 1) * SP : A.npix;
            const int Ra = done == A.npix ? A.hin : done / A.win;          // complete input rows
            const int Rp = st == 0 ? 0 : (st * SP) / A.win;                // ... after the previous step
            // stride 2: out row 2 i + py needs input rows <= i + py
            const int y0 = st == 0 ? 0 : (Rp >= 1 ? 2 * (Rp - 1) : 0);
            const int y1 = Ra == A.hin ? hout : (Ra >= 1 ? 2 * (Ra - 1) : 0);
            for (int yb = y0; yb < y1; yb += RPP) {            // (uniform: the ring slot of the pass's first input row comes from the scalar unit)
                const int y = yb + rofs;
                if (rofs >= RPP || y >= y1) continue;
                {
                    // out(2 i + py, 2 j + px): py 0 <- (i, ky 1), (i - 1, ky 3);  py 1 <- (i + 1, ky 0), (i, ky 2), (i - 1, ky 4)
                    //                          px 0 <- (j, kx 1), (j - 1, kx 3);  px 1 <- (j + 1, kx 0), (j, kx 2), (j - 1, kx 4)
                    // -- over the two pixels of a thread every (row term, kx) is used exactly once: 5 taps x 3 channels per row term
                    const int ib = yb >> 1, rbase = (ib * A.win) % A.RP;
                    const int i = y >> 1, py = y & 1;
                    float e0 = bias0, e1 = bias1, e2 = bias2, o0 = bias0, o1 = bias1, o2 = bias2;      // even x, odd x
                    int off[3][3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int ii = py ? i + 1 - u : i - u, ky = py ? 2 * u : 2 * u + 1;
                        const bool rok = (py || u < 2) && ii >= 0 && ii < A.hin;
                        int rs = rbase + (ii - ib) * A.win;
                        rs = rs < 0 ? rs + A.RP : rs;
                        while (rs >= A.RP) rs -= A.RP;
#pragma unroll
                        for (int e = 0; e < 3; ++e) {          // neighbour column j - 1 + e
                            const int jj = j - 1 + e;
                            int slot = rs + jj;
                            slot = slot >= A.RP ? slot - A.RP : slot;
                            off[u][e] = rok && jj >= 0 && jj < A.win ? slot * M3_PST + ky * 15 : ZS;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const float* pm = P + off[u][0];       // column j - 1: kx 3 -> even x, kx 4 -> odd x
                        const float* pc = P + off[u][1];       // column j:     kx 1 -> even x, kx 2 -> odd x
                        const float* pq = P + off[u][2];       // column j + 1: kx 0 -> odd x
                        const float m9 = pm[9], m10 = pm[10], m11 = pm[11], m12 = pm[12], m13 = pm[13], m14 = pm[14];
                        const float c3 = pc[3], c4 = pc[4], c5 = pc[5], c6 = pc[6], c7 = pc[7], c8 = pc[8];
                        const float q0 = pq[0], q1 = pq[1], q2 = pq[2];
                        e0 += c3; e1 += c4; e2 += c5; e0 += m9; e1 += m10; e2 += m11;
                        o0 += q0; o1 += q1; o2 += q2; o0 += c6; o1 += c7; o2 += c8; o0 += m12; o1 += m13; o2 += m14;
                    }
                    float2* o = reinterpret_cast<float2*>(A.out + (((int64_t)img * hout + y) * wout + 2 * j) * 3);
                    o[0] = float2{e0, e1}; o[1] = float2{e2, o0}; o[2] = float2{o1, o2};
                }
            }
        };
        // stride 1: out(y, x) = sum_{ky, kx} P[(y + 2 - ky, x + 2 - kx)][(ky, kx, c)] -- 25 taps per pixel and only 16 NWM pixels per step, so
        // each pixel is shared by the two lanes l, l + 32 of a wave (filter rows 0..2 | 3..4) and the halves meet through one shuffle
        auto gather1 = [&](int img, int st) {
            const int done = (st + 1) * SP < A.npix ? (st + 1) * SP : A.npix;
            const int Ra = done == A.npix ? A.hin : done / A.win;
            const int Rp = st == 0 ? 0 : (st * SP) / A.win;
            const int y0 = st == 0 ? 0 : (Rp >= 2 ? Rp - 2 : 0);
            const int y1 = Ra == A.hin ? hout : (Ra >= 2 ? Ra - 2 : 0);
            const int half = (gt >> 5) & 1, it = (gt >> 6) * 32 + (gt & 31), PPP = NG / 2;      // pixels per pass
            const int total = (y1 - y0) * A.win;
            for (int base = 0; base < total; base += PPP) {
                const int e = base + it, yy = e / A.win, jx = e - yy * A.win, y = y0 + yy;
                const bool live = e < total;
                const int yc = live ? y : y0;
                const int rbase = (y0 * A.win) % A.RP;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 1
                for (int ky = half ? 3 : 0; ky < (half ? 5 : 3); ++ky) {
                    const int ii = yc + 2 - ky;
                    const bool rok = live && ii >= 0 && ii < A.hin;
                    int rs = rbase + (ii - y0) * A.win;
                    rs = rs < 0 ? rs + A.RP : rs;
                    while (rs >= A.RP) rs -= A.RP;
                    int off[5];
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) {
                        const int jj = jx + 2 - kx;
                        int slot = rs + jj;
                        slot = slot >= A.RP ? slot - A.RP : slot;
                        off[kx] = rok && jj >= 0 && jj < A.win ? slot * M3_PST + ky * 15 + kx * 3 : ZS;
                    }
                    float v[5][3];
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) { v[kx][0] = P[off[kx]]; v[kx][1] = P[off[kx] + 1]; v[kx][2] = P[off[kx] + 2]; }
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) { s0 += v[kx][0]; s1 += v[kx][1]; s2 += v[kx][2]; }
                }
                const float t0 = __shfl_xor(s0, 32), t1 = __shfl_xor(s1, 32), t2 = __shfl_xor(s2, 32);
                if (live && half == 0) {
                    float* o = A.out + (((int64_t)img * hout + y) * wout + jx) * 3;
                    o[0] = bias0 + s0 + t0; o[1] = bias1 + s1 + t1; o[2] = bias2 + s2 + t2;
                }
            }
        };
        int img = blockIdx.x, st = 0, pimg = -1, pst = 0;
        while (img < A.nimg) {
            M3_STAMP(0);
            if (pimg >= 0) { if constexpr (S == 2) gather(pimg, pst); else gather1(pimg, pst); }
            M3_STAMP(1);
            __syncthreads();                                   // B1
            M3_STAMP(2);
            __syncthreads();                                   // B2
            M3_STAMP(4);
            ++nstep;
            pimg = img; pst = st;
            if (++st == A.steps) { st = 0; img += gridDim.x; }
        }
        if (pimg >= 0) { if constexpr (S == 2) gather(pimg, pst); else gather1(pimg, pst); }
    }
}

template <int S, int KQH, int NWM, int NWG>
bool launch_ct3m(hipStream_t s, Ct3m A) {
    constexpr int KQ = 2 * KQH, SP = NWM * 16;
    A.RP = SP + (S == 2 ? 3 : 5) * A.win;
    A.steps = (A.npix + SP - 1) / SP;
    const size_t lds = ((size_t)KQ * 5 * 256 + (size_t)(A.RP + 1) * M3_PST) * 4;
    if (lds > (size_t)dev_info().lds_per_cu || A.win > NWG * 64) return false;
    ensure_dyn_lds((const void*)convt3m_kernel<S, KQH, NWM, NWG>, lds);
    int grid = dev_info().cus;
    if (grid > A.nimg) grid = A.nimg;
    { const int rounds = (A.nimg + grid - 1) / grid; grid = (A.nimg + rounds - 1) / rounds; }
    hipLaunchKernelGGL((convt3m_kernel<S, KQH, NWM, NWG>), dim3((unsigned)grid), dim3((NWM + NWG) * 64), lds, s, A);
    return true;
}

}  // namespace

// The shapes the matrix-core d_h4 forward is built for: both input tensors 16 / 32 / 64 channels wide, enough images to give every
// CU whole images (a starved launch keeps convt3.hip's small tiles / the product + gather route), a ring that fits LDS.
bool convt3_mfma_ok(int c1, int c2, int hin, int win, int stride, int nimg) {
    if (!(opt(OPT_DIRECT3) & 16)) return false;
    if (stride != 1 && stride != 2) return false;
    if (c1 != c2 || (c1 != 16 && c1 != 32 && c1 != 64)) return false;
    if (nimg < 128 || win < 4 || hin < 2 || win > 256) return false;
    const size_t lds = ((size_t)(c1 / 8) * 5 * 256 + (size_t)(128 + (stride == 2 ? 3 : 5) * win + 1) * M3_PST) * 4;
    return lds <= (size_t)dev_info().lds_per_cu;
}

void convt3_mfma(hipStream_t s, const float* x1, int c1, const float* x2, int nmod2, int nimg, int hin, int win, int stride,
                 const float* w, const float* bias, float* out) {
    Ct3m A{};
    A.x1 = x1; A.x2 = x2; A.nmod2 = nmod2; A.hin = hin; A.win = win; A.nimg = nimg; A.npix = hin * win;
    A.w = w; A.bias = bias; A.out = out;
    bool ok = false;
#define M3_GO(S_, KQH_) ok = launch_ct3m<S_, KQH_, 8, 4>(s, A)
    if (stride == 2) { if (c1 == 64) M3_GO(2, 4); else if (c1 == 32) M3_GO(2, 2); else M3_GO(2, 1); }
    else { if (c1 == 64) M3_GO(1, 4); else if (c1 == 32) M3_GO(1, 2); else M3_GO(1, 1); }
#undef M3_GO
    if (!ok) set_launch_error("convt3_mfma: %d x %d x (%d + %d) does not fit LDS (convt3_mfma_ok was not asked)", hin, win, c1, c1);
}

}  // namespace ctx
