// dconv.h -- direct convolutions for NARROW channel counts (3 ... 64 per pixel): ContextAEReal's layers (filters 32/16/16/8,
// gym/envs/mujoco/arm_shaping.py:1622-1629, 1659-1672) and the 3-channel edge layers of ContextSkipNew (h0_conv :1283, the
// input gradient of d_h4 :1329).
//
// Why not the implicit GEMM of igemm.h: with <= 32 channels per tap every 16-byte gather feeds a single 32-column MFMA
// block, so a tile is fed from L2 at ~20 B/clk/CU and the matrix pipe idles at 50 % (measured: ContextAEReal 0.18 of its
// roofline).  Here a block stages the INPUT HALO TILE of its output tile in LDS once (coalesced NHWC rows, zero outside the
// image) and walks the 25 taps over it: every input element is fetched from HBM/L2 once per tile instead of once per tap,
// and the operands of v_mfma_f32_16x16x4_f32 (16-wide tiles: no padding of 16-channel layers to 32) come out of LDS with
// one ds_read_b128 per four MFMAs.
//
//   dconv_fwd    out[p][n] = epilogue( sum_taps sum_k in[S*p + tap][k] * w[tap][k][n] )      k = channel
//                one kernel for  conv2d stride 1 / 2  (= input gradient of conv2d_transpose),
//                                conv2d_transpose stride 1 (flipped taps) and stride 2 (four output parity classes, each
//                                with its own tap subset over the SMALL grid)  (= input gradient of conv2d)
//   dconv_wgrad  dw[tap][a][b] = sum_pixels big[S*p + tap][a] * small[p][b]                 k = pixel
//                rows m = tap * CA + a packed densely (cin = 3: 75 rows, not 25 x 32), persistent blocks that keep their
//                partial sums in registers over all their tiles, fixed-order slab reduction (deterministic, no atomics)
//
// K order inside a 16-wide MFMA step: lane (row = l & 15, kg = l >> 4) supplies k = 16*chunk + 4*kg + t to step t -- the
// same permutation for A and B, so one float4 per operand feeds four MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "igemm.h"

namespace ctx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DC_MAXTAPS = 25;
constexpr int DC_THREADS = 512;          // 8 waves: one or two 16-row blocks each
constexpr int DC_NW = DC_THREADS / 64;

struct DcTap { int16_t dy, dx, wt, pad; };        // input-tile pixel offset of the tap; filter tap index ky * 5 + kx
struct DcClass {
    int tap0, ntaps, oy, ox, pslot0;   // taps [tap0, tap0 + ntaps); physical output pixel = (osc*y + oy, osc*x + ox);
                                       // pslot0: first slot of the class in the packed filter (classes padded to whole chunks)
    int ntx, mdiv, dy0, dx0, sgn;      // tap e of the class sits at input-tile offset (dy0 + sgn * (e / ntx), dx0 + sgn * (e % ntx));
                                       // e / ntx = (e * mdiv) >> 8 for e < 25 -- pure ALU, no table read in the MFMA loop
};

struct DcFwd {
    const float* x1; int ld1; int c1;              // input channels [0, c1)
    const float* x2; int ld2; int nmod2;           // input channels [c1, CI): tensor 2, image index img % nmod2 (the ctx skip)
    int CI;                                        // c1 + c2 : 3, or a multiple of 8 up to 64
    int hin, win, nimg;
    int S;                                         // input pixels per logical output pixel (1 | 2)
    int y_org, x_org;                              // input pixel of logical output (0,0) at tap offset (0,0)
    int hlog, wlog;                                // logical output grid = the GEMM's row space (transposed stride 2: the small grid)
    int TH, TW;                                    // tile of logical outputs: TH * TW / 16 row blocks <= DC_NW * MI
    int IH, IW;                                    // input tile incl. halo
    const float* w; int wmode;                     // 0: w[tap][k][n]   1: w[tap][n][k]
    int wres, GT;                                  // wres 1: the whole packed filter stays in LDS for the block's lifetime; 0: GT tap slots are
                                                   // staged at a time inside the tile loop (filters too big to sit beside the tile)
    float* wp;                                     // the filter re-packed for the LDS image: wp[slot][k / 4][n (NP)][k & 3], slots in tap-list
                                                   // order (dconv_pack_kernel, one tiny launch before the convolution): a block's staging
                                                   // is then a straight float4 copy instead of 25 * CIK * NP strided scalar gathers
    int N;                                         // real output channels (<= 16 * NB)
    int ncls; DcClass cls[4]; DcTap taps[DC_MAXTAPS];
    int osc, hout, wout;                           // physical output grid
    int tiles_y, tiles_x;
    Epi ep;
};

__host__ __device__ inline int dc_cip(int cik) { return cik + 4 - (cik == 4 ? 4 : 0); }      // LDS pixel stride: CIK + 4 (bank spread), 4 for the 3-channel tile

// wp[slot][k / 4][n][k & 3] for every class-padded tap slot (zeros in the padding: slots past a class's taps, k >= CI, n >= N)
template <int CIK>
__global__ __launch_bounds__(256) void dconv_pack_kernel(const DcFwd P, int NP, int nslots) {
    constexpr int TPC = CIK >= 16 ? 1 : 16 / CIK;
    const int total = nslots * CIK * NP;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int n = i % NP, ek = i / NP, slot = ek / CIK, k = ek - slot * CIK;
        float v = 0.f;
        int ci = 0;
        while (ci + 1 < P.ncls && slot >= P.cls[ci + 1].pslot0) ++ci;
        const int e = slot - P.cls[ci].pslot0;
        if (e < P.cls[ci].ntaps && n < P.N && k < P.CI) {
            const int wt = P.taps[P.cls[ci].tap0 + e].wt;
            v = P.wmode ? P.w[((int64_t)wt * P.N + n) * P.CI + k] : P.w[((int64_t)wt * P.CI + k) * P.N + n];
        }
        P.wp[((int64_t)(ek >> 2) * NP + n) * 4 + (ek & 3)] = v;
    }
    (void)TPC;
}

// LDS: tile[IH*IW][CIP] | W4[wslots*CIK/4][NP][4]   (wslots = all class-padded tap slots when they fit beside the tile -- the
// usual case: ONE barrier per block -- else GT slots staged at a time)
// Row block rb = wv + DC_NW * mi: the 8 waves all stay busy (latency hiding) and, waves w and w + 4 sharing SIMD w % 4, the
// matrix pipes are evenly loaded whenever the tile has a multiple of 4 row blocks.
template <int CIK, int MI, int NB>
__global__ __launch_bounds__(DC_THREADS, 2) void dconv_fwd_kernel(const DcFwd P, int ntiles, int nslots) {
    constexpr int CIP = CIK == 4 ? 4 : CIK + 4;
    constexpr int NP = NB * 16;
    constexpr int TPC = CIK >= 16 ? 1 : 16 / CIK;          // taps per 16-k chunk
    constexpr int CPT = CIK >= 16 ? CIK / 16 : 1;          // chunks per tap
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    float* tile = smem;
    const int tile_floats = (P.IH * P.IW * CIP + 3) & ~3;
    float* W4 = smem + tile_floats;
    auto stage = [&](int s0, int n) {                        // slots [s0, s0 + n) of the packed filter -> W4[0 ...)
        const float4* src = reinterpret_cast<const float4*>(P.wp) + (int64_t)s0 * (CIK / 4) * NP;
        float4* dst = reinterpret_cast<float4*>(W4);
        for (int i = tid; i < n * (CIK / 4) * NP; i += DC_THREADS) dst[i] = src[i];
    };

    int b = blockIdx.x;
    const int txi = b % P.tiles_x; b /= P.tiles_x;
    const int tyi = b % P.tiles_y;
    const int img = b / P.tiles_y;
    const int ty0 = tyi * P.TH, tx0 = txi * P.TW;
    const int iy0 = P.S * ty0 + P.y_org, ix0 = P.S * tx0 + P.x_org;

    // ---- filter (when resident) and input halo tile -> LDS, all loads in flight together
    if (P.wres) stage(0, nslots);
    if constexpr (CIK == 4) {
        const float* src = P.x1 + (int64_t)img * P.hin * P.win * 3;
        const int rowf = P.IW * 3;
        for (int i = tid; i < P.IH * rowf; i += DC_THREADS) {
            const int iy = i / rowf, f = i - iy * rowf, ix = f / 3, ch = f - ix * 3;
            const int gy = iy0 + iy, gx = ix0 + ix;
            float v = 0.f;
            if ((unsigned)gy < (unsigned)P.hin && (unsigned)gx < (unsigned)P.win) v = src[((int64_t)gy * P.win + gx) * 3 + ch];
            tile[(iy * P.IW + ix) * 4 + ch] = v;
        }
        for (int i = tid; i < P.IH * P.IW; i += DC_THREADS) tile[i * 4 + 3] = 0.f;
    } else {
        constexpr int C4 = CIK / 4;
        const float* s1 = P.x1 + (int64_t)img * P.hin * P.win * P.ld1;
        const float* s2 = P.x2 ? P.x2 + (int64_t)(img % P.nmod2) * P.hin * P.win * P.ld2 : nullptr;
        for (int i = tid; i < P.IH * P.IW * C4; i += DC_THREADS) {
            const int pi = i / C4, c = (i - pi * C4) * 4;
            const int iy = pi / P.IW, ix = pi - iy * P.IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            float4 v = zero4();
            if ((unsigned)gy < (unsigned)P.hin && (unsigned)gx < (unsigned)P.win) {
                const int64_t pix = (int64_t)gy * P.win + gx;
                v = c < P.c1 ? ldg4(s1 + pix * P.ld1 + c) : ldg4(s2 + pix * P.ld2 + (c - P.c1));
            }
            *reinterpret_cast<float4*>(&tile[pi * CIP + c]) = v;
        }
    }

    const int rbw = P.TW >> 4;                               // row blocks per tile row
    const int nrb = P.TH * rbw;                              // row blocks of the tile (<= DC_NW * MI)
    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int rb = wv + DC_NW * mi, ty = rb / rbw, tx = (rb - ty * rbw) * 16 + l15;
        abase[mi] = rb < nrb ? ((P.S * ty) * P.IW + P.S * tx) * CIP : 0;
    }
    const bool active = wv < nrb;
    if (P.wres) __syncthreads();                             // tile + filter visible

    for (int ci = 0; ci < P.ncls; ++ci) {
        DcClass cl;                                          // (static indices: a dynamic one would put the argument struct in scratch)
        switch (ci) { case 0: cl = P.cls[0]; break; case 1: cl = P.cls[1]; break; case 2: cl = P.cls[2]; break; default: cl = P.cls[3]; break; }
        f32x4 acc[MI][NB];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ntp_all = (cl.ntaps + TPC - 1) / TPC * TPC;
        const int gts = P.wres ? ntp_all : P.GT;
        for (int t0 = 0; t0 < ntp_all; t0 += gts) {
            const int ntp = ntp_all - t0 < gts ? ntp_all - t0 : gts;
            if (!P.wres) {
                __syncthreads();                             // tile complete / the previous stage's fragments consumed
                stage(cl.pslot0 + t0, ntp);
                __syncthreads();
            }
            const int nchunks = active ? ntp / TPC * CPT : 0;
            const int kW = P.wres ? (cl.pslot0 + t0) * CIK : 0;          // LDS-K coordinate of this stage's first slot
            const int k0 = t0 * CIK + 4 * kg;                            // class-K coordinate of this lane's first k
            float4 a4[2][MI], b4[2][NB];
            auto fetch = [&](int c, int buf) {
                const int k16 = k0 + 16 * c;
                int e = k16 / CIK;
                const int kin = k16 - e * CIK;
                e = e < cl.ntaps ? e : cl.ntaps - 1;                     // padded slots: any valid address (their filter rows are zero)
                const int q = (e * cl.mdiv) >> 8, r = e - q * cl.ntx;
                const int to = ((cl.dy0 + cl.sgn * q) * P.IW + cl.dx0 + cl.sgn * r) * CIP + kin;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a4[buf][mi] = *reinterpret_cast<const float4*>(&tile[abase[mi] + to]);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    b4[buf][nb] = *reinterpret_cast<const float4*>(&W4[((kW / 4 + 4 * c + kg) * NP + nb * 16 + l15) * 4]);
            };
            auto mma = [&](int buf) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const float av = tt == 0 ? a4[buf][mi].x : tt == 1 ? a4[buf][mi].y : tt == 2 ? a4[buf][mi].z : a4[buf][mi].w;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const float bv = tt == 0 ? b4[buf][nb].x : tt == 1 ? b4[buf][nb].y : tt == 2 ? b4[buf][nb].z : b4[buf][nb].w;
                            acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[mi][nb], 0, 0, 0);
                        }
                    }
                }
            };
            // software pipeline, branch-free in the steady state: fragments of chunk c + 1 are read before the MFMAs of chunk c
            if (nchunks > 0) {
                fetch(0, 0);
                int c = 0;
                for (; c + 2 < nchunks; c += 2) {
                    fetch(c + 1, 1);
                    mma(0);
                    fetch(c + 2, 0);
                    mma(1);
                }
                if (c + 1 < nchunks) { fetch(c + 1, 1); mma(0); mma(1); }
                else mma(0);
            }
        }
        // ---- epilogue of this class.  D: col = l15, row = 4 * kg + r
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int rb = wv + DC_NW * mi, ty = rb / rbw, txb = (rb - ty * rbw) * 16;
            const int y = ty0 + ty;
            if (rb >= nrb || y >= P.hlog) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = tx0 + txb + 4 * kg + r;
                if (x >= P.wlog) continue;
                const int64_t pix = ((int64_t)img * P.hout + (P.osc * y + cl.oy)) * P.wout + (P.osc * x + cl.ox);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int n = nb * 16 + l15;
                    if (n < P.N) epi_store(P.ep, 0, pix, n, acc[mi][nb][r]);
                }
            }
        }
    }
    (void)ntiles;
}

// ------------------------------------------------------------------------------------------------
// Filter gradient.  dw[tap][a][b] = sum over images and small-grid pixels p of big[S*p + tap - pad][a] * small[p][b].
// GEMM rows m = tap * CA + a (M = 25 * CA, dense), columns b, K = pixels.  A block stages the small tile and the big halo
// tile of one (image, tile) in LDS, runs all its row blocks over the tile's pixels, moves to its next tile (persistent,
// accumulators stay in registers) and finally writes ONE partial [M][CB] to its slab; dconv_wgrad_reduce adds the slabs.
// ------------------------------------------------------------------------------------------------
struct DcWgrad {
    const float* big; int ldb; int CA;              // big-grid tensor (channels a); CA == 3: read as [pixel][3]
    const float* s1; int ld1; int c1;              // small-grid tensor, channels [0, c1)
    const float* s2; int ld2; int nmod2;           // channels [c1, CB) from tensor 2 (ctx skip, image % nmod2)
    int CB;                                        // c1 + c2, a multiple of 8
    int hb, wb, hs, ws, nimg;
    int S, pad;
    int TH, TW, tw_sh;                             // small-grid tile, TW = 1 << tw_sh in {16, 32, 64}, so TH*TW is a multiple of 16
    int IH, IW;                                    // big tile incl. halo: S*(TH-1)+5, S*(TW-1)+5
    int tiles_y, tiles_x, ntiles;                  // per image; ntiles = nimg * tiles_y * tiles_x
    int M;                                         // 25 * CA
    int RBW;                                       // row blocks (of 16 rows of M) per wave
    float* slab;                                   // [gridDim.x][M][CBP]
    float* out;                                    // dw [25][CA][CB]
};

template <int CAK /* big-tile pixel stride class: 4 (CA = 3), 8, 16, 32 */, int RBW, int NB>
__global__ __launch_bounds__(DC_THREADS) void dconv_wgrad_kernel(const DcWgrad P) {
    constexpr int CAP = CAK == 4 ? 4 : CAK + 4;             // pixel strides = 4 mod 16 dwords: the four pixel groups (kg) of a wave read
    constexpr int NP = NB * 16;                             // 16-bank windows that do not overlap (4 pixels * stride = 16 mod 64)
    constexpr int CBP = NP + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    float* bigt = smem;                                      // [IH*IW][CAP]
    float* smallt = smem + ((P.IH * P.IW * CAP + 3) & ~3);   // [TH*TW][CBP]
    const int npix = P.TH * P.TW;

    // this lane's A rows: m = (wv * RBW + rb) * 16 + l15 -> (tap, a) -> offset of the tap inside the big tile + channel
    int aoff[RBW];
    bool aok[RBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        const int m = (wv * RBW + rb) * 16 + l15;
        aok[rb] = m < P.M;
        const int mm = aok[rb] ? m : 0;
        const int tap = mm / P.CA, a = mm - tap * P.CA, ky = tap / 5, kx = tap - 5 * ky;
        aoff[rb] = (ky * P.IW + kx) * CAP + a;
    }
    f32x4 acc[RBW][NB];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = blockIdx.x; t < P.ntiles; t += gridDim.x) {
        int b = t;
        const int txi = b % P.tiles_x; b /= P.tiles_x;
        const int tyi = b % P.tiles_y;
        const int img = b / P.tiles_y;
        const int ty0 = tyi * P.TH, tx0 = txi * P.TW;
        const int iy0 = P.S * ty0 - P.pad, ix0 = P.S * tx0 - P.pad;
        __syncthreads();                                     // previous tile consumed
        // ---- big halo tile
        if constexpr (CAK == 4) {
            const float* src = P.big + (int64_t)img * P.hb * P.wb * 3;
            const int rowf = P.IW * 3;
            for (int i = tid; i < P.IH * rowf; i += DC_THREADS) {
                const int iy = i / rowf, f = i - iy * rowf, ix = f / 3, ch = f - ix * 3;
                const int gy = iy0 + iy, gx = ix0 + ix;
                float v = 0.f;
                if ((unsigned)gy < (unsigned)P.hb && (unsigned)gx < (unsigned)P.wb) v = src[((int64_t)gy * P.wb + gx) * 3 + ch];
                bigt[(iy * P.IW + ix) * 4 + ch] = v;
            }
        } else {
            constexpr int C4 = CAK / 4;
            const float* src = P.big + (int64_t)img * P.hb * P.wb * P.ldb;
            for (int i = tid; i < P.IH * P.IW * C4; i += DC_THREADS) {
                const int pi = i / C4, c = (i - pi * C4) * 4;
                const int iy = pi / P.IW, ix = pi - iy * P.IW;
                const int gy = iy0 + iy, gx = ix0 + ix;
                float4 v = zero4();
                if (c < P.CA && (unsigned)gy < (unsigned)P.hb && (unsigned)gx < (unsigned)P.wb) v = ldg4(src + ((int64_t)gy * P.wb + gx) * P.ldb + c);
                *reinterpret_cast<float4*>(&bigt[pi * CAP + c]) = v;
            }
        }
        // ---- small tile (zero outside the grid: those pixels contribute nothing)
        {
            const int C4 = NP / 4;
            const float* p1 = P.s1 + (int64_t)img * P.hs * P.ws * P.ld1;
            const float* p2 = P.s2 ? P.s2 + (int64_t)(img % P.nmod2) * P.hs * P.ws * P.ld2 : nullptr;
            for (int i = tid; i < npix * C4; i += DC_THREADS) {
                const int pi = i / C4, c = (i - pi * C4) * 4;
                const int ty = pi / P.TW, tx = pi - ty * P.TW;
                const int gy = ty0 + ty, gx = tx0 + tx;
                float4 v = zero4();
                if (c < P.CB && gy < P.hs && gx < P.ws) {
                    const int64_t pix = (int64_t)gy * P.ws + gx;
                    v = c < P.c1 ? ldg4(p1 + pix * P.ld1 + c) : ldg4(p2 + pix * P.ld2 + (c - P.c1));
                }
                *reinterpret_cast<float4*>(&smallt[pi * CBP + c]) = v;
            }
        }
        __syncthreads();
        // ---- K loop over the tile's pixels, 16 per chunk: lane kg takes pixels 16c + 4kg + t.  Software-pipelined: the LDS
        // reads of chunk c + 1 are issued before the MFMAs of chunk c.
        float av[2][RBW][4], bv[2][NB][4];
        auto fetch = [&](int c, int buf) {
            const int p0 = c + 4 * kg;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int p = p0 + tt, ty = p >> P.tw_sh, tx = p - (ty << P.tw_sh);
                const int pbig = ((P.S * ty) * P.IW + P.S * tx) * CAP, psm = p * CBP;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[buf][nb][tt] = smallt[psm + nb * 16 + l15];
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) av[buf][rb][tt] = bigt[pbig + aoff[rb]];
            }
        };
        auto mma = [&](int buf) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][rb][tt], bv[buf][nb][tt], acc[rb][nb], 0, 0, 0);
        };
        fetch(0, 0);
        for (int c = 0; c < npix; c += 32) {
            if (c + 16 < npix) fetch(c + 16, 1);
            mma(0);
            if (c + 16 < npix) {
                if (c + 32 < npix) fetch(c + 32, 0);
                mma(1);
            }
        }
    }
    // ---- partial -> slab[block][m][n]
    float* sl = P.slab + (int64_t)blockIdx.x * P.M * NP;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (wv * RBW + rb) * 16 + 4 * kg + r;
            if (m >= P.M) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) sl[(int64_t)m * NP + nb * 16 + l15] = acc[rb][nb][r];
        }
}

}  // namespace ctx
