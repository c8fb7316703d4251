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

#include <type_traits>

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

struct DcPackCache;

struct DcFwd {
    const float* x1; int ld1; int c1;              // input channels [0, c1)
    const float* x2; int ld2; int nmod2;           // input channels [c1, CI): tensor 2, image index img % nmod2 (the ctx skip)
    int CI;                                        // c1 + c2 : 3, or a multiple of 8 up to 64
    int hin, win, nimg;
    int S;                                         // input pixels per logical output pixel (1 | 2)
    int y_org, x_org;                              // input pixel of logical output (0,0) at tap offset (0,0)
    int hlog, wlog;                                // logical output grid = the GEMM's row space (transposed stride 2: the small grid)
    int TH, TW;                                    // tile of logical outputs: TH * TW / 16 row blocks <= DC_NW * MI
    int IH, IW, CIP;                               // input tile incl. halo; its pixel stride in LDS (dc_cip)
    const float* w; int wmode;                     // 0: w[tap][k][n]   1: w[tap][n][k]
    int n0, NPT;                                   // this launch computes output columns [n0, n0 + 16 * NB) (filters too big to sit in LDS beside
                                                   // a tile are run as several column slices); NPT = column count of the packed image
    float* wp;                                     // the filter re-packed for the LDS image: wp[slot][k / 4][n (NP)][k & 3], slots in tap-list
                                                   // order (dconv_pack_kernel, one tiny launch before the convolution): a block's staging
                                                   // is then a straight float4 copy instead of 25 * CIK * NP strided scalar gathers
    int N;                                         // real output channels (<= 16 * NB)
    int ncls; DcClass cls[4]; DcTap taps[DC_MAXTAPS];
    int osc, hout, wout;                           // physical output grid
    int tiles_y, tiles_x;
    Epi ep;
    unsigned long long* trace = nullptr;           // -DDC_TRACE builds of tools/dconv_bench.hip: s_memtime stamps [block][wave][tile][8] (never set by the product)
    DcPackCache* pc = nullptr;                     // host side only: where packed filters are kept between launches (null: pack into wp every time)
};

// Packed-filter cache of one handle.  The LDS image of a layer's filter (dconv_pack_kernel) depends on the parameters and on the layer's
// shape only, so it is packed ONCE per parameter version into a slot of `arena` instead of by a tiny launch in front of every
// convolution (four launches in a 0.18 ms ContextAEReal encode call; same-box A/B at 25 frames: encode 0.182 -> 0.175 ms, translate
// 0.357 -> 0.342, 64x64 0.390 -> 0.373).  The owner
// bumps `version` whenever a parameter may have changed (Adam, set_params, a broadcast), re-captures its inference graphs after a
// bump, and never hands the cache to a handle whose arena a caller may write behind its back.  An entry is repacked when its version
// is old or when another stream asks for it.  A training step repacks everything at first use, as before: redoing all entries ahead of
// the step on the idle filter-gradient lane was measured and is no faster (2.60 against 2.59 ms; the 5 us packs already hide).
struct DcPackCache {
    float* arena = nullptr;
    int64_t floats = 0, used = 0;
    uint64_t version = 1;
    uint64_t hits = 0;               // launches that skipped their pack (the owner tells a capture with all its pack nodes from one without)
    bool external = false;           // the caller holds a writable pointer to the parameters (ctx_dev_params): never trust an entry
    struct Ent { const float* w; int wmode, N, CI, CIK, NPT, nslots, sig; int64_t off; uint64_t version; void* stream; };
    static constexpr int MAXE = 64;
    Ent ent[MAXE];
    int n = 0;
};


// LDS pixel stride of the input tile.  A fragment read is a ds_read_b128 at (S * CIP) * l15 + 4 * kg dwords; its four 16-lane
// groups are conflict-free exactly when S * CIP = 8 mod 16 (or 4 / 8 outright) -- enumerated over the hardware's lane groups;
// CIK + 4 for every stride cost 2x on the stride-1 layers (SQ_LDS_BANK_CONFLICT = a third of the LDS cycles).
__host__ __device__ inline int dc_cip(int cik, int S) { return cik == 4 ? 4 : cik == 8 ? (S == 1 ? 8 : 12) : cik + (S == 1 ? 8 : 4); }

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
    if (blockIdx.x == 0 && threadIdx.x < 16) P.wp[total + threadIdx.x] = 0.f;      // 64 bytes of zeros behind the image (dconv2.h: where halo lanes fetch from)
    (void)TPC;
}

// LDS: tile[IH*IW][CIP] | W4[nslots*CIK/4][NP][4] (the whole class-padded filter, resident for the block's lifetime).
//
// PERSISTENT blocks with a REGISTER-STAGED PREFETCH: a block walks tiles t = blockIdx.x, + gridDim.x, ...; while it runs the
// MFMA loop of tile t out of LDS, the global loads of tile t + gridDim.x are already in flight into DC_PF registers per
// thread (one float4 -- one float of the [pixel][3] tensors -- per slot), and are written to LDS after the loop.  The first
// version staged each tile with a dependent load -> ds_write loop and one tile per block: 11 exposed HBM round trips per
// tile, 24 us per block for 2.7 us of MFMA work (h1_conv of ContextAEReal: 0.43 ms for a 0.07 ms layer).
// Out-of-image pixels are raw_buffer_loads at the out-of-range marker (the hardware returns zeros, no branch, no traffic).
// Row block rb = wv + DC_NW * mi: the 8 waves all stay busy (latency hiding) and, waves w and w + 4 sharing SIMD w % 4, the
// matrix pipes are evenly loaded whenever the tile has a multiple of 4 row blocks.
constexpr int DC_PF = 12;           // prefetch slots per thread: the big-tile kernels (one block of 8 waves per CU, up to 256 registers)
constexpr int DC_PF_SMALL = 4;      // ... and the kernels built for TWO blocks per CU (<= 128 registers): tiles of <= 2048 elements, so that one
                                    // block's landing / epilogue / barriers run under the other's MFMA loop
typedef unsigned dc_u32x4 __attribute__((ext_vector_type(4)));

template <int CIK, int MI, int NB, int PF = DC_PF>
__global__ __launch_bounds__(DC_THREADS, (PF <= DC_PF_SMALL ? 4 : 2)) void dconv_fwd_kernel(const DcFwd P, int ntiles, int nslots) {
    const int CIP = P.CIP;                                 // (only in address set-up: the MFMA loop reads through abase[] + tab[])
    constexpr int NP = NB * 16;
    constexpr int TPC = CIK >= 16 ? 1 : 16 / CIK;          // taps per 16-k chunk
    constexpr int CPT = CIK >= 16 ? CIK / 16 : 1;          // chunks per tap
    constexpr int C4 = CIK / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    float* tile = smem;
    const int tile_floats = (P.IH * P.IW * CIP + 3) & ~3;
    float* W4 = smem + tile_floats;

    // ---- resident filter: columns [n0, n0 + NP) of the packed image wp[slot * CIK / 4 + kq][NPT][4]
    {
        const float4* src = reinterpret_cast<const float4*>(P.wp);
        float4* dst = reinterpret_cast<float4*>(W4);
        const int total = nslots * C4 * NP;
        for (int i0 = tid; i0 < total; i0 += 4 * DC_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * DC_THREADS, row = i / NP, n = i - row * NP;
                v[u] = i < total ? src[(int64_t)row * P.NPT + P.n0 + n] : zero4();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + u * DC_THREADS < total) dst[i0 + u * DC_THREADS] = v[u];
        }
    }
    if constexpr (CIK == 4) for (int i = tid; i < P.IH * P.IW; i += DC_THREADS) tile[i * 4 + 3] = 0.f;   // the 4th channel is never loaded
    // tab[g][kg]: byte offset, inside the tile, of (tap, first channel) that lane group kg multiplies in chunk g of the packed K
    // order (tile-invariant: computed once here instead of ~40 ALU instructions per chunk in front of every ds_read)
    int* tab = reinterpret_cast<int*>(W4 + (size_t)nslots * CIK * NP);
    for (int ci = 0; ci < P.ncls; ++ci) {
        DcClass cl;                                          // (static indices: a dynamic one would put the argument struct in scratch)
        switch (ci) { case 0: cl = P.cls[0]; break; case 1: cl = P.cls[1]; break; case 2: cl = P.cls[2]; break; default: cl = P.cls[3]; break; }
        const int g0 = cl.pslot0 * CIK / 16, ng = (cl.ntaps + TPC - 1) / TPC * CPT;
        // (+ 1 behind the LAST class: the entry the pipeline reads ahead and never uses.  Behind the other classes that slot IS the next
        // class's first entry -- writing a placeholder there raced with the thread that writes the real one: harmless while the
        // placeholder's thread sat in wave 0 (classes of 4 / 6 taps first), wrong results on a cold box now and then once round 5 put
        // the 9-tap class first and its placeholder into wave 1)
        for (int i = tid; i < (ng + (ci == P.ncls - 1 ? 1 : 0)) * 4; i += DC_THREADS) {
            const int k16 = (i >> 2) * 16 + 4 * (i & 3);
            int e = k16 / CIK;
            const int kin = k16 - e * CIK;
            e = e < cl.ntaps ? e : cl.ntaps - 1;             // padded slots: any valid address (their filter rows are zero)
            const int q = (e * cl.mdiv) >> 8, r = e - q * cl.ntx;
            tab[g0 * 4 + i] = (((cl.dy0 + cl.sgn * q) * P.IW + cl.dx0 + cl.sgn * r) * CIP + kin) * 4;
        }
    }

    const int rowf = P.IW * 3;
    dc_u32x4 pf[CIK == 4 ? 1 : PF];
    unsigned pf1[CIK == 4 ? PF : 1];
    auto tile_org = [&](int t, int& img, int& ty0, int& tx0) {
        const int txi = t % P.tiles_x; t /= P.tiles_x;
        const int tyi = t % P.tiles_y;
        img = t / P.tiles_y; ty0 = tyi * P.TH; tx0 = txi * P.TW;
    };
    // Slot j of a thread is element tid + 512 j of the tile.  Its tile coordinates follow from the thread's first element by a
    // recurrence (a fixed step of rows and columns, one wrap) -- NOT by dividing every index by the runtime tile width: with the
    // divisions, address arithmetic was 4.5 VALU instructions per MFMA over the whole kernel (PMC) and took as long as the MFMA loop.
    constexpr int SPX = CIK == 4 ? DC_THREADS : DC_THREADS / C4;             // floats (CIK 4) | pixels a slot step advances
    const int roww = CIK == 4 ? rowf : P.IW;                                // floats | pixels per tile row
    const int qy = SPX / roww, rx = SPX - qy * roww;
    const int e0 = CIK == 4 ? tid : tid / C4, cth = CIK == 4 ? 0 : (tid % C4) * 4;
    const int iyb = e0 / roww, ixb = e0 - iyb * roww;
    auto issue = [&](int t) {                                // global loads of tile t -> prefetch registers
        int img, ty0, tx0;
        tile_org(t, img, ty0, tx0);
        const int iy0 = P.S * ty0 + P.y_org, ix0 = P.S * tx0 + P.x_org;
        int iy = iyb, ix = ixb;
        if constexpr (CIK == 4) {
            const rsrc_t rs = make_rsrc(P.x1 + (int64_t)img * P.hin * P.win * 3);
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int gy = iy0 + iy, gx3 = ix0 * 3 + ix;      // (ix counts floats of the row here)
                const bool ok = iy < P.IH && (unsigned)gy < (unsigned)P.hin && (unsigned)gx3 < (unsigned)(P.win * 3);
                pf1[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (uint32_t)((gy * P.win * 3 + gx3) * 4) : OOB, 0, 0);
                ix += rx; iy += qy;
                if (ix >= roww) { ix -= roww; ++iy; }
            }
        } else {
            // one source: buffer loads, out-of-image lanes at the out-of-range marker.  Two sources ([decoder | ctx skip]): the
            // descriptor would be lane-dependent (a waterfall loop per load), so plain loads through a selected pointer, halo
            // lanes at the image's first pixel and zeroed when they land.
            const float* s1 = P.x1 + (int64_t)img * P.hin * P.win * P.ld1;
            if (!P.x2) {
                const rsrc_t rs1 = make_rsrc(s1);
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int gy = iy0 + iy, gx = ix0 + ix;
                    const bool ok = iy < P.IH && (unsigned)gy < (unsigned)P.hin && (unsigned)gx < (unsigned)P.win;
                    pf[j] = __builtin_amdgcn_raw_buffer_load_b128(rs1, ok ? (uint32_t)(((gy * P.win + gx) * P.ld1 + cth) * 4) : OOB, 0, 0);
                    ix += rx; iy += qy;
                    if (ix >= roww) { ix -= roww; ++iy; }
                }
            } else {
                const float* s2 = P.x2 + (int64_t)(img % P.nmod2) * P.hin * P.win * P.ld2;
                const float* sp = cth < P.c1 ? s1 + cth : s2 + (cth - P.c1);
                const int ld = cth < P.c1 ? P.ld1 : P.ld2;
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int gy = iy0 + iy, gx = ix0 + ix;
                    const bool ok = iy < P.IH && (unsigned)gy < (unsigned)P.hin && (unsigned)gx < (unsigned)P.win;
                    const int pix = ok ? gy * P.win + gx : 0;
                    pf[j] = *reinterpret_cast<const dc_u32x4*>(sp + (int64_t)pix * ld);
                    ix += rx; iy += qy;
                    if (ix >= roww) { ix -= roww; ++iy; }
                }
            }
        }
    };
    auto land = [&](int t) {                                  // prefetch registers -> LDS tile
        // (the empty asm pins the FIRST use of the prefetch registers here: land() sits inside the class loop, and loop-invariant code
        // motion otherwise hoists the halo select -- and with it the s_waitcnt for the loads -- in front of the MFMA loops)
        if constexpr (CIK == 4) {
#pragma unroll
            for (int j = 0; j < PF; ++j) asm volatile("" : "+v"(pf1[j]));
        } else {
#pragma unroll
            for (int j = 0; j < PF; ++j) asm volatile("" : "+v"(pf[j]));
        }
        int iy = iyb, ix = ixb;
        if constexpr (CIK == 4) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int px = ix / 3, ch = ix - px * 3;
                if (iy < P.IH) tile[(iy * P.IW + px) * 4 + ch] = __uint_as_float(pf1[j]);
                ix += rx; iy += qy;
                if (ix >= roww) { ix -= roww; ++iy; }
            }
        } else {
            int img, ty0, tx0;
            tile_org(t, img, ty0, tx0);
            const int iy0 = P.S * ty0 + P.y_org, ix0 = P.S * tx0 + P.x_org;
            const int lds0 = e0 * CIP + cth;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                // (two-source tiles: halo lanes outside the image were loaded from a pixel that exists and become zeros here)
                const bool halo = P.x2 && !((unsigned)(iy0 + iy) < (unsigned)P.hin && (unsigned)(ix0 + ix) < (unsigned)P.win);
                if (iy < P.IH) *reinterpret_cast<dc_u32x4*>(&tile[lds0 + j * (SPX * CIP)]) = halo ? dc_u32x4{0u, 0u, 0u, 0u} : pf[j];
                ix += rx; iy += qy;
                if (ix >= roww) { ix -= roww; ++iy; }
            }
        }
    };

    const int rbw = P.TW >> 4;                               // row blocks per tile row
    const int nrb = P.TH * rbw;                              // row blocks of the tile (<= DC_NW * MI)
    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int rb = wv + DC_NW * mi, ty = rb / rbw, tx = (rb - ty * rbw) * 16 + l15;
        abase[mi] = rb < nrb ? ((P.S * ty) * P.IW + P.S * tx) * CIP * 4 : 0;      // bytes
    }
    const bool active = wv < nrb;

    // The products are formed TRANSPOSED (filter fragment as the MFMA's A operand, pixels as B): a lane then holds, per 16 x 16
    // block, FOUR CONSECUTIVE CHANNELS n = n0 + 16 nb + 4 kg + {0..3} of ONE pixel (x = l15) -- one 16-byte store (and one 16-byte
    // load per epilogue term) per lane and block instead of four 4-byte ones; the scalar stores of the first version were
    // issue-bound (~7 B/clk/CU) on the wide, output-heavy layers.
    int ncol[NB];
    float4 bias_v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        ncol[nb] = P.n0 + nb * 16 + 4 * kg;
        bias_v[nb] = (P.ep.bias && ncol[nb] < P.N) ? ldg4(P.ep.bias + ncol[nb]) : zero4();
    }
    // Order of a tile's phases (round 5):  issue(next) | MFMA loops of the classes | barrier | land(next) | stores of the last class |
    // barrier.  The NEXT tile is written to LDS before this tile's results are stored: land() consumes the prefetch registers, i.e.
    // waits for vmcnt(0) -- which also waits for every store issued before it.  With the stores in front of it (round 2-4: epilogue,
    // drain, barrier, land) each tile paid a full store round trip (~1.4 us of 11: h1_conv forward 21 us of 170) with all eight
    // waves idle; behind it, the stores retire under the next tile's MFMA loop.
    int t = blockIdx.x;
    if (t < ntiles) issue(t);
    __syncthreads();                                         // (4th channel zeroed)
    if (t < ntiles) land(t);
    __syncthreads();                                         // tile, filter and offset table visible
#ifdef DC_TRACE
    int trace_i = 0;
#define DC_STAMP(k) do { if (P.trace && lane == 0 && blockIdx.x < 8 && trace_i < 32) P.trace[(((size_t)blockIdx.x * DC_NW + wv) * 32 + trace_i) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define DC_STAMP(k) do {} while (0)
#endif
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + (int)gridDim.x;
        DC_STAMP(0);
#ifndef DC_ABL_NOLDG                                         // (ablation builds of tools/dconv_bench.hip only: wrong results, right amount of the OTHER work)
        if (tn < ntiles) issue(tn);
#endif
        int img, ty0, tx0;
        tile_org(t, img, ty0, tx0);
        DC_STAMP(1);

        for (int ci = 0; ci < P.ncls; ++ci) {
            DcClass cl;                                      // (static indices: a dynamic one would put the argument struct in scratch)
            switch (ci) { case 0: cl = P.cls[0]; break; case 1: cl = P.cls[1]; break; case 2: cl = P.cls[2]; break; default: cl = P.cls[3]; break; }
            f32x4 acc[MI][NB];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            // chunks [g0, g1) of the packed filter = this class; per chunk a lane needs tab[g][kg] (byte offset of its tap / channel
            // group inside the tile, built once per block) and filter rows 4 g + kg
            const int g0 = cl.pslot0 * CIK / 16;
            const int g1 = g0 + (cl.ntaps + TPC - 1) / TPC * CPT;
            auto run = [&](auto nmi_c) {
                constexpr int NMI = decltype(nmi_c)::value;  // row blocks this wave really has (MI, or MI - 1 in a tile whose last round of row blocks is partial)
                float4 a4[2][NMI > 0 ? NMI : 1], b4[2][NB];
                int tnext = tab[g0 * 4 + kg];
                const char* bp = reinterpret_cast<const char*>(W4) + ((size_t)(4 * g0 + kg) * NP + l15) * 16;
                auto fetch = [&](int buf) {
                    const int to = tnext;
#pragma unroll
                    for (int mi = 0; mi < NMI; ++mi) a4[buf][mi] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tile) + abase[mi] + to);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) b4[buf][nb] = *reinterpret_cast<const float4*>(bp + nb * 256);
                    bp += 64 * NP;
                };
                auto mma = [&](int buf) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                        for (int mi = 0; mi < NMI; ++mi) {
                            const float av = tt == 0 ? a4[buf][mi].x : tt == 1 ? a4[buf][mi].y : tt == 2 ? a4[buf][mi].z : a4[buf][mi].w;
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                const float bv = tt == 0 ? b4[buf][nb].x : tt == 1 ? b4[buf][nb].y : tt == 2 ? b4[buf][nb].z : b4[buf][nb].w;
#ifdef DC_ABL_NOMFMA
                                asm volatile("" :: "v"(bv), "v"(av));
#else
                                acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[mi][nb], 0, 0, 0);      // D^T: rows = channels, cols = pixels
#endif
                            }
                        }
                    }
                };
                // software pipeline: the fragments of chunk g + 1 are read before the MFMAs of chunk g, and the table entry of chunk
                // g + 2 before them -- a fetch never waits for the table read in front of it (LDS returns in order: waiting for
                // the newest read waits for all), and the scheduling barriers keep the compiler from sinking a fetch next to its use
                // (it did, to save registers: the ISA had an s_waitcnt lgkmcnt(0) straight behind each ds_read_b128 -- the whole LDS
                // latency exposed once per chunk, the MFMA loop at 65 % of its instruction count's time)
                fetch(0);
                int g = g0;
                int tahead = tab[(g0 + 1) * 4 + kg];         // (the table has one entry past the class's last chunk)
                for (; g + 2 < g1; g += 2) {
                    tnext = tahead;
                    tahead = tab[(g + 2) * 4 + kg];
                    fetch(1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(0);
                    __builtin_amdgcn_sched_barrier(0);
                    tnext = tahead;
                    tahead = tab[(g + 3) * 4 + kg];
                    fetch(0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g + 1 < g1) { tnext = tahead; fetch(1); __builtin_amdgcn_sched_barrier(0); mma(0); mma(1); }
                else mma(0);
            };
            // The terms an element's epilogue needs from memory (skip-gradient adds, the saved activation behind lrelu') are requested
            // HERE, in front of the class's MFMA loop, for all of the wave's row blocks: their latency runs under the MFMAs, and no
            // load sits between the stores below (a load's wait also waits for the stores issued before it: the per-row-block
            // load -> wait -> store chain costs one store round trip per row block and class).  MI * NB > 4 -- 64-column tiles with
            // two or three row blocks per wave -- has no registers for that and keeps the per-row-block batches.
            constexpr bool PRE = MI * NB <= 4;
            constexpr int PM = PRE ? MI : 1, PN = PRE ? NB : 1;
            int64_t ppix[PM];
            bool pok[PM];
            float4 p1[PM][PN], p2[PM][PN], pm[PM][PN];
            if constexpr (PRE) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int rb = wv + DC_NW * mi, ty = rb / rbw, x = tx0 + (rb - ty * rbw) * 16 + l15, y = ty0 + ty;
                    pok[mi] = rb < nrb && y < P.hlog && x < P.wlog;
                    ppix[mi] = pok[mi] ? ((int64_t)img * P.hout + (P.osc * y + cl.oy)) * P.wout + (P.osc * x + cl.ox) : 0;   // (a pixel that exists)
                    if (P.ep.add1) {
                        const int64_t pa = (P.ep.add1_mod && ppix[mi] >= P.ep.add1_mod) ? ppix[mi] - P.ep.add1_mod : ppix[mi];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) p1[mi][nb] = ldg4(P.ep.add1 + pa * P.ep.lda1 + (ncol[nb] < P.N ? ncol[nb] : 0));
                    }
                    if (P.ep.add2) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) p2[mi][nb] = ldg4(P.ep.add2 + ppix[mi] * P.ep.lda2 + (ncol[nb] < P.N ? ncol[nb] : 0));
                    }
                    if (P.ep.mask) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            pm[mi][nb] = ldg4(P.ep.mask + ppix[mi] * P.ep.ldm + (ncol[nb] < P.N && ncol[nb] < P.ep.nsplit ? ncol[nb] : 0));
                    }
                }
            }
#ifndef DC_ABL_NOLOOP
            if (wv + DC_NW * (MI - 1) < nrb) run(std::integral_constant<int, MI>{});
            else if constexpr (MI > 1) { if (active) run(std::integral_constant<int, MI - 1>{}); }
#endif
            if (ci == P.ncls - 1) {                          // the tile's last MFMA loop is done: every wave has read its last fragment
                DC_STAMP(2);
                __syncthreads();
                DC_STAMP(3);
#ifdef DC_ABL_NOLAND
                if (false)
#endif
                if (tn < ntiles) land(tn);
                DC_STAMP(4);
            }
            // ---- epilogue of this class.  D^T: a lane has channels ncol[nb] .. + 3 of pixel x = l15 of each of its row blocks.  The
            // terms an element needs from memory (skip-gradient adds, the saved activation behind lrelu') are loaded for a whole row
            // block first, branch-free, and only then applied (epi_store's load -> wait -> store chain per element took
            // longer than the MFMA loop).
#ifdef DC_ABL_NOEPI
            if (acc[0][0][0] == 12345.678f) P.ep.out1[0] = acc[0][0][1];
            if constexpr (false) {
#else
            if constexpr (PRE) {
#endif
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int n = ncol[nb];
                        // (every loaded term is CONSUMED by every lane, whether it stores or not: a load whose use sits behind a branch
                        // counts as possibly pending at the loop back-edge, and the compiler then puts s_waitcnt vmcnt(0) in front of the
                        // next class's loads into the same registers -- i.e. waits for this class's stores and the next tile's prefetch)
                        float v[4] = {acc[mi][nb][0] + bias_v[nb].x, acc[mi][nb][1] + bias_v[nb].y, acc[mi][nb][2] + bias_v[nb].z, acc[mi][nb][3] + bias_v[nb].w};
                        if (P.ep.add1) { v[0] += p1[mi][nb].x; v[1] += p1[mi][nb].y; v[2] += p1[mi][nb].z; v[3] += p1[mi][nb].w; }
                        if (P.ep.add2) { v[0] += p2[mi][nb].x; v[1] += p2[mi][nb].y; v[2] += p2[mi][nb].z; v[3] += p2[mi][nb].w; }
                        if (P.ep.lrelu) {
                            const float lk = P.ep.lrelu == 2 ? 0.f : LEAK;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lk * v[r]);
                        }
                        if (P.ep.mask) {
                            const bool m1 = n < P.ep.nsplit;
                            v[0] *= (m1 && pm[mi][nb].x < 0.f) ? LEAK : 1.f; v[1] *= (m1 && pm[mi][nb].y < 0.f) ? LEAK : 1.f;
                            v[2] *= (m1 && pm[mi][nb].z < 0.f) ? LEAK : 1.f; v[3] *= (m1 && pm[mi][nb].w < 0.f) ? LEAK : 1.f;
                        }
                        if (!pok[mi] || n >= P.N) continue;
                        if (n < P.ep.nsplit) {
                            *reinterpret_cast<float4*>(P.ep.out1 + ppix[mi] * P.ep.ld1 + n) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            *reinterpret_cast<float4*>(P.ep.out2 + ppix[mi] * P.ep.ld2 + (n - P.ep.nsplit)) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            } else
#ifdef DC_ABL_NOEPI
            if constexpr (false)
#endif
            {
                int64_t pix[MI];
                bool okp[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int rb = wv + DC_NW * mi, ty = rb / rbw, x = tx0 + (rb - ty * rbw) * 16 + l15, y = ty0 + ty;
                    okp[mi] = rb < nrb && y < P.hlog && x < P.wlog;
                    pix[mi] = okp[mi] ? ((int64_t)img * P.hout + (P.osc * y + cl.oy)) * P.wout + (P.osc * x + cl.ox) : 0;   // (a pixel that exists)
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    float4 t1[NB], t2[NB], tm[NB];
                    if (P.ep.add1) {
                        const int64_t pa = (P.ep.add1_mod && pix[mi] >= P.ep.add1_mod) ? pix[mi] - P.ep.add1_mod : pix[mi];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) t1[nb] = ldg4(P.ep.add1 + pa * P.ep.lda1 + (ncol[nb] < P.N ? ncol[nb] : 0));
                    }
                    if (P.ep.add2) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) t2[nb] = ldg4(P.ep.add2 + pix[mi] * P.ep.lda2 + (ncol[nb] < P.N ? ncol[nb] : 0));
                    }
                    if (P.ep.mask) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            tm[nb] = ldg4(P.ep.mask + pix[mi] * P.ep.ldm + (ncol[nb] < P.N && ncol[nb] < P.ep.nsplit ? ncol[nb] : 0));
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int n = ncol[nb];
                        float v[4] = {acc[mi][nb][0] + bias_v[nb].x, acc[mi][nb][1] + bias_v[nb].y, acc[mi][nb][2] + bias_v[nb].z, acc[mi][nb][3] + bias_v[nb].w};
                        if (P.ep.add1) { v[0] += t1[nb].x; v[1] += t1[nb].y; v[2] += t1[nb].z; v[3] += t1[nb].w; }
                        if (P.ep.add2) { v[0] += t2[nb].x; v[1] += t2[nb].y; v[2] += t2[nb].z; v[3] += t2[nb].w; }
                        if (P.ep.lrelu) {
                            const float lk = P.ep.lrelu == 2 ? 0.f : LEAK;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lk * v[r]);
                        }
                        if (P.ep.mask) {                    // (every loaded term consumed by every lane: see the PRE form above)
                            const bool m1 = n < P.ep.nsplit;
                            v[0] *= (m1 && tm[nb].x < 0.f) ? LEAK : 1.f; v[1] *= (m1 && tm[nb].y < 0.f) ? LEAK : 1.f;
                            v[2] *= (m1 && tm[nb].z < 0.f) ? LEAK : 1.f; v[3] *= (m1 && tm[nb].w < 0.f) ? LEAK : 1.f;
                        }
                        if (!okp[mi] || n >= P.N) continue;
                        if (n < P.ep.nsplit) {
                            *reinterpret_cast<float4*>(P.ep.out1 + pix[mi] * P.ep.ld1 + n) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            *reinterpret_cast<float4*>(P.ep.out2 + pix[mi] * P.ep.ld2 + (n - P.ep.nsplit)) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            }
#ifdef DC_DRAIN                                          // (rounds 2-4 ended every class with s_waitcnt vmcnt(0): A/B build of tools/dconv_bench.hip)
            __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
        }
        DC_STAMP(5);
        __syncthreads();                                     // the next tile is visible
        DC_STAMP(6);
#ifdef DC_TRACE
        ++trace_i;
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// Filter gradient.  dw[tap][a][b] = sum over images and small-grid pixels p of big[S*p + tap - pad][a] * small[p][b].
// GEMM rows m = tap * CA + a (M = 25 * CA, dense), columns b, K = pixels.  A block stages the small tile and the big halo
// tile of one (image, tile) in LDS, runs its row blocks over the tile's pixels, moves to its next tile (persistent,
// accumulators stay in registers) and finally writes ONE partial [M][NP] to its slab; dconv_wgrad_reduce adds the slabs.
// The 8 waves are WM groups along M x WK groups along K: a 3-channel `big` has only 5 row blocks (M = 75), so every wave
// takes all of them for an eighth of the tile's pixels (WM 1 x WK 8, B fragments shared by five row blocks) and the eight
// partials are added in a fixed tree through LDS when the block retires; 32 channels are 50 row blocks = 8 waves x 7.
// Like the forward kernel, the next tile's global loads are in flight (registers) under the current tile's MFMA loop.
// ------------------------------------------------------------------------------------------------
struct DcWgrad {
    const float* big; int ldb; int CA;              // big-grid tensor (channels a); CA == 3: read as [pixel][3]
    const float* s1; int ld1; int c1;              // small-grid tensor, channels [0, c1)
    const float* s2; int ld2; int nmod2;           // channels [c1, CB) from tensor 2 (ctx skip, image % nmod2)
    int CB;                                        // c1 + c2, a multiple of 8
    int n0;                                        // this launch: columns [n0, n0 + 16 * NB) of CB
    int hb, wb, hs, ws, nimg;
    int S, pad;
    int TH, TW, tw_sh;                             // small-grid tile, TW = 1 << tw_sh in {16, 32, 64}, so TH*TW is a multiple of 16
    int IH, IW;                                    // big tile incl. halo: S*(TH-1)+5, S*(TW-1)+5
    int tiles_y, tiles_x, ntiles;                  // per image; ntiles = nimg * tiles_y * tiles_x
    int M;                                         // 25 * CA
    float* slab;                                   // [gridDim.x][M][NP]
    float* out;                                    // dw [25][CA][CB]
    float* db;                                     // != nullptr: also the column sums of the SMALL tensor (= the bias gradient of a conv layer,
                                                   // whose dy is the small operand) -- the tile is in LDS anyway; saves a pass over dy
    float* dbslab;                                 // [gridDim.x][NP] partial column sums (behind the dw slabs)
};

constexpr int DC_PFB = 14;                         // prefetch slots per thread, big tile (float4s; floats of a [pixel][3] tensor)
__host__ __device__ constexpr int dc_cbp(int np, int S) { return np + (S == 2 ? 8 : 4); }   // small-tile pixel stride in LDS
__host__ __device__ constexpr int dc_pfs(int nb) { return nb >= 4 ? 8 : 2 * nb; }   // small tile: TH*TW*NP/4 float4s <= 512 * this

template <int CAK /* big-tile pixel stride class: 4 (CA = 3), 8, 16, 32 */, int RBW, int WM, int NB, int SS /* stride */>
__global__ __launch_bounds__(DC_THREADS) void dconv_wgrad_kernel(const DcWgrad P) {
    // ds_read_b32 is served 32 lanes (two pixel groups kg) at a time over 32 banks, 16 consecutive dwords per group: the two
    // groups must sit 16 banks apart.  Lane group kg takes pixels p0 = c + PD * kg + {0..3 | 0, 1, 8, 9}: PD = 4 pixels apart at
    // stride 1 (4 * CAP = 16 mod 32 with CAP = CAK + 4; small tile 4 * (NP + 4)), PD = 2 at stride 2 (2 * 2 * CAP = 16 mod 32;
    // small tile 2 * (NP + 8)) -- with PD = 4 at stride 2 both groups hit the same banks (measured: 43 % of the LDS cycles).
    constexpr int CAP = CAK == 4 ? 4 : CAK + 4;
    constexpr int NP = NB * 16;
    constexpr int CBP = dc_cbp(NP, SS);
    constexpr int PD = SS == 2 ? 2 : 4;
    constexpr int WK = DC_NW / WM;
    constexpr int C4 = CAK / 4, S4 = NP / 4, PFS = dc_pfs(NB);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int wm = wv % WM, wk = wv / WM;
    float* bigt = smem;                                      // [IH*IW][CAP]
    float* smallt = smem + ((P.IH * P.IW * CAP + 3) & ~3);   // [TH*TW][CBP]
    const int npix = P.TH * P.TW;

    // this lane's A rows: m = (wm * RBW + rb) * 16 + l15 -> (tap, a) -> byte offset of the tap inside the big tile + channel
    int aoff[RBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
        const int m = (wm * RBW + rb) * 16 + l15;
        const int mm = m < P.M ? m : 0;
        const int tap = mm / P.CA, a = mm - tap * P.CA, ky = tap / 5, kx = tap - 5 * ky;
        aoff[rb] = ((ky * P.IW + kx) * CAP + a) * 4;
    }
    f32x4 acc[RBW][NB];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (CAK == 4) for (int i = tid; i < P.IH * P.IW; i += DC_THREADS) bigt[i * 4 + 3] = 0.f;   // never loaded, never read as data (a < 3)
    float dbacc = 0.f;

    const int nsm_e = npix * S4;
    const int rowf = P.IW * 3;
    dc_u32x4 pb[CAK == 4 ? 1 : DC_PFB];
    unsigned pb1[CAK == 4 ? DC_PFB : 1];
    dc_u32x4 ps[PFS];
    auto tile_org = [&](int t, int& img, int& ty0, int& tx0) {
        const int txi = t % P.tiles_x; t /= P.tiles_x;
        const int tyi = t % P.tiles_y;
        img = t / P.tiles_y; ty0 = tyi * P.TH; tx0 = txi * P.TW;
    };
    // big-tile slot coordinates by recurrence from the thread's first element (see dconv_fwd_kernel: no division per slot)
    constexpr int SPX = CAK == 4 ? DC_THREADS : DC_THREADS / C4;
    const int roww = CAK == 4 ? rowf : P.IW;
    const int qy = SPX / roww, rx = SPX - qy * roww;
    const int e0 = CAK == 4 ? tid : tid / C4, cth = CAK == 4 ? 0 : (tid % C4) * 4;
    const int iyb = e0 / roww, ixb = e0 - iyb * roww;
    auto issue = [&](int t) {
        int img, ty0, tx0;
        tile_org(t, img, ty0, tx0);
        const int iy0 = SS * ty0 - P.pad, ix0 = SS * tx0 - P.pad;
        int iy = iyb, ix = ixb;
        if constexpr (CAK == 4) {
            const rsrc_t rs = make_rsrc(P.big + (int64_t)img * P.hb * P.wb * 3);
#pragma unroll
            for (int j = 0; j < DC_PFB; ++j) {
                const int gy = iy0 + iy, gx3 = ix0 * 3 + ix;
                const bool ok = iy < P.IH && (unsigned)gy < (unsigned)P.hb && (unsigned)gx3 < (unsigned)(P.wb * 3);
                pb1[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (uint32_t)((gy * P.wb * 3 + gx3) * 4) : OOB, 0, 0);
                ix += rx; iy += qy;
                if (ix >= roww) { ix -= roww; ++iy; }
            }
        } else {
            const rsrc_t rs = make_rsrc(P.big + (int64_t)img * P.hb * P.wb * P.ldb);
#pragma unroll
            for (int j = 0; j < DC_PFB; ++j) {
                const int gy = iy0 + iy, gx = ix0 + ix;
                const bool ok = iy < P.IH && cth < P.CA && (unsigned)gy < (unsigned)P.hb && (unsigned)gx < (unsigned)P.wb;
                pb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (uint32_t)(((gy * P.wb + gx) * P.ldb + cth) * 4) : OOB, 0, 0);
                ix += rx; iy += qy;
                if (ix >= roww) { ix -= roww; ++iy; }
            }
        }
        // small tile: zero outside the grid and past CB (those pixels / columns contribute nothing).  Two sources -> plain
        // loads through a selected pointer (a lane-dependent buffer descriptor would cost a waterfall loop per load).
        const float* p1 = P.s1 + (int64_t)img * P.hs * P.ws * P.ld1;
        const float* p2 = P.s2 ? P.s2 + (int64_t)(img % P.nmod2) * P.hs * P.ws * P.ld2 : p1;
#pragma unroll
        for (int j = 0; j < PFS; ++j) {
            const int i = tid + j * DC_THREADS;
            const int pi = i / S4, c = P.n0 + (i - pi * S4) * 4;
            const int ty = pi >> P.tw_sh, tx = pi - (ty << P.tw_sh);
            const int gy = ty0 + ty, gx = tx0 + tx;
            const bool ok = i < nsm_e && c < P.CB && gy < P.hs && gx < P.ws;
            const int pix = ok ? gy * P.ws + gx : 0;
            const int cc = ok ? c : 0;
            const float* p = cc < P.c1 ? p1 + (int64_t)pix * P.ld1 + cc : p2 + (int64_t)pix * P.ld2 + (cc - P.c1);
            ps[j] = *reinterpret_cast<const dc_u32x4*>(p);
        }
    };
    auto land = [&](int t) {
        int img, ty0, tx0;
        tile_org(t, img, ty0, tx0);
        {
            int iy = iyb, ix = ixb;
            if constexpr (CAK == 4) {
#pragma unroll
                for (int j = 0; j < DC_PFB; ++j) {
                    const int px = ix / 3, ch = ix - px * 3;
                    if (iy < P.IH) bigt[(iy * P.IW + px) * 4 + ch] = __uint_as_float(pb1[j]);
                    ix += rx; iy += qy;
                    if (ix >= roww) { ix -= roww; ++iy; }
                }
            } else {
                const int lds0 = e0 * CAP + cth;
#pragma unroll
                for (int j = 0; j < DC_PFB; ++j) {
                    if (iy < P.IH) *reinterpret_cast<dc_u32x4*>(&bigt[lds0 + j * (SPX * CAP)]) = pb[j];
                    ix += rx; iy += qy;
                    if (ix >= roww) { ix -= roww; ++iy; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PFS; ++j) {
            const int i = tid + j * DC_THREADS;
            const int pi = i / S4, cl = (i - pi * S4) * 4;
            const int ty = pi >> P.tw_sh, tx = pi - (ty << P.tw_sh);
            const bool ok = P.n0 + cl < P.CB && ty0 + ty < P.hs && tx0 + tx < P.ws;
            if (i < nsm_e) *reinterpret_cast<dc_u32x4*>(&smallt[pi * CBP + cl]) = ok ? ps[j] : dc_u32x4{0u, 0u, 0u, 0u};
        }
    };

    int t = blockIdx.x;
    if (t < P.ntiles) issue(t);
    for (; t < P.ntiles; t += gridDim.x) {
        __syncthreads();                                     // previous tile consumed
        land(t);
        __syncthreads();
        if (t + (int)gridDim.x < P.ntiles) issue(t + gridDim.x);
        if (P.db) {                                          // column sums of the small tile (zeros outside the grid / past CB): thread = (column, pixel phase)
            const int col = tid % NP, seg = tid / NP;
            for (int p = seg; p < npix; p += DC_THREADS / NP) dbacc += smallt[p * CBP + col];
        }
        // ---- K loop over this wave's share of the tile's pixels, 16 per chunk: lane group kg takes pixels c + 4 kg + tt.
        // Software-pipelined: the LDS reads of the wave's next chunk are issued before the MFMAs of the current one.
        // A lane's four pixels p0 .. p0 + 3 (p0 a multiple of 4, TW >= 16) sit in one tile row: one address per operand row,
        // the pixels at compile-time offsets (ds_read with an immediate) -- not an address computation per value.
        // Pipelined per k-step: the fragment registers of step tt are re-loaded for the wave's NEXT chunk right after the
        // MFMAs of step tt have been issued (single-buffered: 4 * (RBW + NB) registers instead of twice that).
        float av[RBW][4], bv[NB][4];
        constexpr int CS = 16 * WK;                          // chunk stride of one wave
        {
            const char* pa[RBW];
            const float* psm;
            auto pofs = [](int tt) { return SS == 2 ? (tt & 1) + 8 * (tt >> 1) : tt; };      // pixel of k-step tt relative to p0 (same tile row: TW >= 16)
            auto point = [&](int c) {
                const int p0 = c + PD * kg, ty = p0 >> P.tw_sh, tx = p0 - (ty << P.tw_sh);
                const char* pbig = reinterpret_cast<const char*>(bigt) + ((SS * ty) * P.IW + SS * tx) * (CAP * 4);
                psm = smallt + p0 * CBP + l15;
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) pa[rb] = pbig + aoff[rb];
            };
            auto fetch_tt = [&](int tt) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[nb][tt] = psm[pofs(tt) * CBP + nb * 16];
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) av[rb][tt] = *reinterpret_cast<const float*>(pa[rb] + pofs(tt) * SS * CAP * 4);
            };
            auto mma_tt = [&](int tt) {
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][tt], bv[nb][tt], acc[rb][nb], 0, 0, 0);
            };
            int c = 16 * wk;
            if (c < npix) {
            point(c);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) fetch_tt(tt);
            for (; c + CS < npix; c += CS) {
                point(c + CS);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) { mma_tt(tt); fetch_tt(tt); }
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) mma_tt(tt);
            }
        }
    }
    // ---- bias gradient: the threads' partial column sums -> one per column, in a fixed order, -> this block's slab row
    if (P.db) {
        __syncthreads();                                     // tiles dead
        smem[tid] = dbacc;                                   // [pixel phase][column]
        __syncthreads();
        if (tid < NP) {
            float v = 0.f;
            for (int sgi = 0; sgi < DC_THREADS / NP; ++sgi) v += smem[sgi * NP + tid];
            P.dbslab[(int64_t)blockIdx.x * NP + tid] = v;
        }
    }
    // ---- the WK partial sums of a row block -> one, in a fixed tree through LDS (deterministic)
    if constexpr (WK > 1) {
        float* red = smem;                                   // [WK / 2][WM][RBW][NB][4][64]
#pragma unroll
        for (int half = WK / 2; half >= 1; half /= 2) {
            __syncthreads();                                 // tiles dead / previous round read
            if (wk >= half && wk < 2 * half) {
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) red[(((((wk - half) * WM + wm) * RBW + rb) * NB + nb) * 4 + r) * 64 + lane] = acc[rb][nb][r];
            }
            __syncthreads();
            if (wk < half) {
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[rb][nb][r] += red[((((wk * WM + wm) * RBW + rb) * NB + nb) * 4 + r) * 64 + lane];
            }
        }
    }
    // ---- partial -> slab[block][m][n]
    if (wk == 0) {
        float* sl = P.slab + (int64_t)blockIdx.x * P.M * NP;
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = ((wm * RBW) + rb) * 16 + 4 * kg + r;
                if (m >= P.M) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) sl[(int64_t)m * NP + nb * 16 + l15] = acc[rb][nb][r];
            }
    }
}

}  // namespace ctx
