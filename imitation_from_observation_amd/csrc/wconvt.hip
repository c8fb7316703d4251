// wconvt.hip -- conv2d_transpose 5x5 stride 2 (arm_shaping.py:62-85) for WIDE channel counts (cin, cout multiples of 32) on the
// 8x8 and 16x16 grids, image-major with the input HALO TILE resident in LDS.  Round 3.
//
// Why: the position-major implicit GEMM (igemm.h: KmConvTGatherQ) streams BOTH operands of every 128x128x32 chunk from L2 --
// 21-32 FLOP per byte fed to LDS, the measured "real streaming loads" ceiling of DESIGN.md section 6 -- and re-fetches every
// input pixel once per (tap, parity class) it serves (up to 25 times; 779 MB through the fabric per launch for ~125 MB of
// tensors).  Here a block owns IMG whole images (256 output positions per parity class) and a 32*NB-wide slice of output
// channels for ALL FOUR parity classes:
//   * a 32-channel slice of the images' input pixels (+ a one-pixel zero halo) is staged in LDS ONCE and serves all 25 taps:
//     6400*NB/2 MFMAs per 47-58 KB staged, against 128 MFMAs per 24.5 KB before;
//   * the filter goes through a two-stage LDS ring, one (tap, slice) = 32*NB x 32 floats at a time (one float4 per thread), one
//     barrier per tap = per 32*NB/2... MFMAs of every wave;
//   * wave w owns rows [32 w, 32 w + 32) of the 256 and keeps 4 classes x NB accumulators (128 registers at NB = 2); the tap ->
//     (class, pixel shift) table is compile time, so a tap's A fragment is one ds_read_b128 at a literal offset from the lane's
//     base address, and the loop has no address arithmetic at all.
// All 25 taps are formed for every pixel (the halo holds zeros): 14 % / 7 % of the products on 8x8 / 16x16 grids meet a zero;
// the 4x4 layers (28 %) stay on the class-major implicit GEMM.
// v_mfma_f32_32x32x2_f32, exact f32.  k order inside a slice: step (q, t): lanes 0-31 take channel 8q + t, lanes 32-63 8q + 4 + t
// (both operands alike).  LDS: A [halo pixel][36], B [2][32 NB][36] floats -- ds_read_b128 at a 36-dword row stride is conflict free.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "launch.h"

namespace ctx {

namespace {

constexpr int WC_THREADS = 256;                 // 4 waves, one per SIMD; two blocks share a CU (LDS 44-47 KB, <= 256 registers)
constexpr int WC_ROWS = 128;                    // output positions (per parity class) of a block
constexpr int WC_LDP = 36;                      // floats per LDS row (32 channels + 4 pad)

struct WcT {
    const float* s1; int c1;                     // input [nimg, HS, WS, c1]
    const float* s2; int c2; int nmod2;          // second channel group [nmod2, HS, WS, c2] (ctx skip, image n % nmod2), c2 = 0: none
    const float* w; int ca;                      // filter [5][5][ca][c1 + c2], ca = output channels
    int nimg;
    int gn;                                      // column tiles
    int ksplit;                                  // > 1: the channel slices are dealt to `ksplit` blocks per (tile, column tile); raw partial
    float* slab;                                 // sums go to slab[split][output pixel][ca] and splitk_reduce applies the epilogue
    Epi ep;
};

// tap t of the 25 in class order (0,0) x4, (0,1) x6, (1,0) x6, (1,1) x9.  Class (py, px): taps ky = par + 2 sy with par = (py + 1) & 1,
// input row i + oy - sy, oy = (py + 1 - par) / 2 (KmConvTGather's notation, K = 5, pb = 1).
struct TapInfo { int cls, dy, dx, ky, kx; };
__host__ __device__ constexpr TapInfo tap_info(int t) {
    const int c = t < 4 ? 0 : t < 10 ? 1 : t < 16 ? 2 : 3;
    const int u = t - (c == 0 ? 0 : c == 1 ? 4 : c == 2 ? 10 : 16);
    const int py = c >> 1, px = c & 1;
    const int pary = (py + 1) & 1, parx = (px + 1) & 1;
    const int ntx = (5 - parx + 1) >> 1;
    const int sy = u / ntx, sx = u - sy * ntx;
    const int oy = (py + 1 - pary) >> 1, ox = (px + 1 - parx) >> 1;
    return TapInfo{c, oy - sy, ox - sx, pary + 2 * sy, parx + 2 * sx};
}

// MFMA row (= lane & 31 for the A operand) -> position index inside the wave's 32 positions (row-major over grid rows of WS).
// The 16 lanes of a ds_read_b128 group must hit 16 pixel slots that are distinct mod 16:
//   WS = 16: group A lanes {0-3,12-15,20-27} take grid row 0 (columns in lane order), group B lanes {4-11,16-19,28-31} row 1;
//   WS = 8 (pitch 12): group A takes rows 0 and 2 (slots 0-7 and 24-31 = 8-15 mod 16), group B rows 1 and 3 (12-19, 36-43);
//   WS = 4: positions in lane order (the 4x4 instance is not tuned).
template <int WS>
__device__ __forceinline__ int wc_pos(int row) {
    if constexpr (WS == 4) return row;
    // k = index of the lane inside its group, in the order the hardware lists the group
    const bool ga = row < 4 || (row >= 12 && row < 16) || (row >= 20 && row < 28);
    const int k = ga ? (row < 4 ? row : row < 16 ? row - 8 : row - 12) : (row < 12 ? row - 4 : row < 20 ? row - 8 : row - 16);
    if constexpr (WS == 16) return (ga ? 0 : 16) + k;                      // one grid row per group
    else return ((ga ? 0 : 1) + (k >> 3) * 2) * 8 + (k & 7);               // WS = 8: rows {0, 2} / {1, 3}
}

// A block = TH grid rows of IMGT images (TH * WS * IMGT = 128 positions) x 32 NB output channels x the four parity classes.
//   16x16 grid: half an image (TH = 8, halo tile 10 x 18);   8x8 grid: two images (halo tile 2 x 10 x 10)
template <int HS, int WS, int NB>
__global__ __launch_bounds__(WC_THREADS, 2) void wconvt_kernel(const WcT P) {
    constexpr int TH = HS * WS >= WC_ROWS ? WC_ROWS / WS : HS, IMGT = WC_ROWS / (TH * WS), TPI = HS / TH;   // rows per tile, images per tile, tiles per image
    // halo tile in LDS: HP rows of WPL pixel slots per image (WP = WS + 2 of them used).  ds_read_b128 serves a wave in four groups of 16
    // lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32); a group is conflict free when its 16 pixel slots are distinct
    // mod 16 (36-dword slots: 9 sixteen-byte units each, 9 is odd).  16-wide grids: a group = one grid row (any pitch).  8-wide grids:
    // a group = two rows two apart, which needs the pitch = 12 slots (10 would put three rows on slot 0 mod 16); PMC before: 38 % of
    // the LDS cycles of the 8x8 launches were bank conflicts.  `wc_pos` is the lane -> position map that goes with it.
    constexpr int HP = TH + 2, WP = WS + 2, WPL = WS == 8 ? 12 : WP, TPIX = IMGT * HP * WPL;
    constexpr int NPA = (TPIX * 8 + WC_THREADS - 1) / WC_THREADS;        // float4 per thread per A slice
    constexpr int BROWS = 32 * NB, BSTAGE = BROWS * WC_LDP;              // floats per B stage
    constexpr int NPB = BROWS * 8 / WC_THREADS;                          // float4 per thread per filter tile (1 or 2)
    static_assert(NPB >= 1 && TH * WS * IMGT == WC_ROWS, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + TPIX * WC_LDP;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    // block -> (tile, column tile): the column tiles of one tile sit on ONE XCD (blocks are dealt round-robin to the 8 XCDs), next to
    // each other in dispatch order, so that the second one finds the input pixels in that XCD's L2
    int item = blockIdx.x;
    {
        const int xcd = item & 7, l = item >> 3, nt = l % P.gn, g8 = l / P.gn;
        item = (g8 * 8 + xcd) * P.gn + nt;                       // (splits are the outermost index: g8 runs through them)
    }
    const int ntile = ((P.nimg + IMGT - 1) / IMGT) * TPI;
    const int nt8 = (ntile + 7) / 8 * 8 * P.gn;                  // items per split (whole groups of 8 tiles: the XCD mapping above)
    const int split = item / nt8;
    item -= split * nt8;
    const int tile = item / P.gn, ctile = item - tile * P.gn;
    if (tile >= ntile) return;
    const int img0 = (tile / TPI) * IMGT, i0 = (tile % TPI) * TH, n0 = ctile * BROWS;
    const int cb = P.c1 + P.c2, ns1 = P.c1 >> 5;
    const int nsl_all = cb >> 5, per = (nsl_all + P.ksplit - 1) / P.ksplit;
    const int sbeg = split * per, nslice = sbeg + per < nsl_all ? sbeg + per : nsl_all;      // this block's slices [sbeg, nslice)
    if (sbeg >= nslice) return;                                    // (never with the launcher's split counts)

    const rsrc_t rs1 = make_rsrc(P.s1), rs2 = make_rsrc(P.s2 ? P.s2 : P.s1);

    // ---- loaders -----------------------------------------------------------------------------------------------
    // A: float4 f = tid + 256 p of the slice tile: halo pixel f >> 3 (tile-local (il, y, x)), channels 4 (f & 7) ..
    auto a_load = [&](int slice, int p) -> float4 {
        const int f = tid + WC_THREADS * p;
        const int hp = f >> 3, k4 = (f & 7) * 4;
        const int il = hp / (HP * WPL), r = hp - il * (HP * WPL), y = i0 + r / WPL - 1, x = r - (r / WPL) * WPL - 1;   // (slots x >= WS + 1 are padding: zeros)
        const int img = img0 + il;
        const bool ok = f < TPIX * 8 && (unsigned)y < (unsigned)HS && (unsigned)x < (unsigned)WS && img < P.nimg;
        const bool second = slice >= ns1;
        const int im = second ? img % P.nmod2 : img;
        const int ld = second ? P.c2 : P.c1, kc = (second ? slice - ns1 : slice) * 32;
        const uint32_t v = (uint32_t)(((im * HS + y) * WS + x) * ld + kc + k4) * 4u;
        return bload4(second ? rs2 : rs1, ok ? v : OOB);
    };
    auto a_store = [&](int p, float4 v) {
        const int f = tid + WC_THREADS * p;
        if (f < TPIX * 8) *reinterpret_cast<float4*>(&sA[(f >> 3) * WC_LDP + (f & 7) * 4]) = v;
    };
    // B: (tap, slice) tile = rows n0 .. n0 + 32 NB of w[ky][kx][.][slice * 32 ..]: float4 (tid + 256 p) -> row >> 3, channels 4 (. & 7)
    const int bk4 = (tid & 7) * 4;
    uint32_t bvoff[NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        const int row = (tid >> 3) + 32 * p;
        bvoff[p] = n0 + row < P.ca ? (uint32_t)((n0 + row) * cb + bk4) * 4u : OOB;
    }
    auto b_load = [&](int slice, int tapw, float4 (&v)[NPB]) {         // tapw = ky * 5 + kx: a literal at every call site
        if (slice >= nslice) slice = nslice - 1;                       // (redundant tail reloads are harmless)
        const rsrc_t r = make_rsrc(P.w + ((int64_t)tapw * P.ca) * cb + slice * 32);
#pragma unroll
        for (int p = 0; p < NPB; ++p) v[p] = bload4(r, bvoff[p]);
    };
    auto b_store = [&](int stage, const float4 (&v)[NPB]) {
#pragma unroll
        for (int p = 0; p < NPB; ++p) *reinterpret_cast<float4*>(&sB[stage * BSTAGE + ((tid >> 3) + 32 * p) * WC_LDP + bk4]) = v[p];
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------
    // row m = 32 wv + l31 of the 128: tile-local position; the lane's base = halo pixel (i, j) = input pixel (i - 1, j - 1), so that
    // tap shift (dy, dx) in {-1, 0, 1}^2 is the non-negative literal offset ((dy + 1) WP + dx + 1) rows
    const int m = 32 * wv + wc_pos<WS>(l31), il = m / (TH * WS), pp = m - il * (TH * WS), pi = pp / WS, pj = pp - pi * WS;
    const float* aBase = sA + ((il * HP + pi) * WPL + pj) * WC_LDP + 4 * h;
    const float* bBase = sB + l31 * WC_LDP + 4 * h;

    f32x16 acc[4][NB];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][nb][r] = 0.f;

    // ---- prologue: slice 0 and tap 0 into LDS, tap 1 into the registers ------------------------------------------
    float4 areg[NPA], breg[NPB];
#pragma unroll
    for (int p = 0; p < NPA; ++p) areg[p] = a_load(sbeg, p);
    b_load(sbeg, tap_info(0).ky * 5 + tap_info(0).kx, breg);
#pragma unroll
    for (int p = 0; p < NPA; ++p) a_store(p, areg[p]);
    b_store(0, breg);
    b_load(sbeg, tap_info(1).ky * 5 + tap_info(1).kx, breg);
    __syncthreads();

    for (int s = sbeg; s < nslice; ++s) {
        const int snext = s + 1 < nslice ? s + 1 : s;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const TapInfo ti = tap_info(t);
            const int g = (s - sbeg) * 25 + t;
            const int stage = g & 1;
            const float* aT = aBase + ((ti.dy + 1) * WPL + (ti.dx + 1)) * WC_LDP;
            const float* bT = bBase + stage * BSTAGE;
            float4 a[2], b[2][NB];
            a[0] = *reinterpret_cast<const float4*>(aT);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[0][nb] = *reinterpret_cast<const float4*>(bT + nb * 32 * WC_LDP);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3) {
                    a[(q + 1) & 1] = *reinterpret_cast<const float4*>(aT + 8 * (q + 1));
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) b[(q + 1) & 1][nb] = *reinterpret_cast<const float4*>(bT + nb * 32 * WC_LDP + 8 * (q + 1));
                }
                const float av[4] = {a[q & 1].x, a[q & 1].y, a[q & 1].z, a[q & 1].w};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float4 bq = b[q & 1][nb];
                        const float bv = tt == 0 ? bq.x : tt == 1 ? bq.y : tt == 2 ? bq.z : bq.w;
                        // (filter value as the MFMA's row operand: D rows = output channels, columns = positions, so that a lane ends
                        // up with 4 CONSECUTIVE CHANNELS of one position in 4 consecutive registers -- float4 epilogue)
                        acc[ti.cls][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av[tt], acc[ti.cls][nb], 0, 0, 0);
                    }
                // in the gaps: the filter tile of tap g + 1 goes to the other stage, the one of tap g + 2 is requested, and the next
                // slice's input pixels are requested one float4 per tap (they sit in registers until the slice ends)
                if (q == 0) b_store(stage ^ 1, breg);
                if (q == 1) {
                    const int t2 = t + 2 < 25 ? t + 2 : t + 2 - 25;
                    const TapInfo tj = tap_info(t2);
                    b_load(t + 2 < 25 ? s : s + 1, tj.ky * 5 + tj.kx, breg);
                }
                if (q == 2 && t < NPA) areg[t] = a_load(snext, t);
            }
            __syncthreads();
        }
        if (s + 1 < nslice) {                     // every wave has finished with this slice's pixels (the barrier above)
#pragma unroll
            for (int p = 0; p < NPA; ++p) a_store(p, areg[p]);
            __syncthreads();
        }
    }

    // ---- epilogue.  D = W x In^T: column = l31 = the lane's position, row = (r & 3) + 8 (r >> 2) + 4 h = output channel inside the
    // 32-wide column block, i.e. registers 4g .. 4g + 3 are channels 8g + 4h .. + 3: every access below is one float4 per lane.  The
    // tensors an element needs besides its own value (bias; skip gradients and the saved activation behind lrelu' on the
    // input-gradient path) are loaded for a whole class first and only then applied (a per-element load -> wait -> store chain
    // costs more than a slice of the MFMA loop).
    const Epi& e = P.ep;
    const int mm = 32 * wv + wc_pos<WS>(l31);
    if (P.slab) {                                                  // split over the channel slices: raw partial sums, [split][pixel][ca]
        const int il2 = mm / (TH * WS), p2 = mm - il2 * (TH * WS), i2 = i0 + p2 / WS, j2 = p2 % WS;
        const int img = img0 + il2;
        if (img >= P.nimg) return;
        const int64_t npix = (int64_t)P.nimg * (4 * HS * WS);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t pix = ((int64_t)img * (2 * HS) + 2 * i2 + (c >> 1)) * (2 * WS) + 2 * j2 + (c & 1);
            float* q = P.slab + ((int64_t)split * npix + pix) * P.ca + n0 + 4 * h;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(q + nb * 32 + 8 * g) = make_float4(acc[c][nb][4 * g], acc[c][nb][4 * g + 1], acc[c][nb][4 * g + 2], acc[c][nb][4 * g + 3]);
        }
        return;
    }
    const int il2 = mm / (TH * WS), p2 = mm - il2 * (TH * WS), i2 = i0 + p2 / WS, j2 = p2 % WS;
    const int img = img0 + il2;
    const bool rowok = img < P.nimg;
    const rsrc_t rm = make_rsrc(e.mask ? e.mask : P.s1), r1 = make_rsrc(e.add1 ? e.add1 : P.s1), r2 = make_rsrc(e.add2 ? e.add2 : P.s1);
    const float leak = e.lrelu == 2 ? 0.f : LEAK;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const uint32_t pix = (uint32_t)((img * (2 * HS) + 2 * i2 + py) * (2 * WS) + 2 * j2 + px);       // < 2^29: one tensor is < 2 GiB
        const uint32_t pa = (e.add1_mod && (int64_t)pix >= e.add1_mod) ? pix - (uint32_t)e.add1_mod : pix;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int nn = n0 + nb * 32 + 4 * h;
            float4 bias[4], mk[4], a1[4], a2[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[g] = e.bias ? ldg4(e.bias + nn + 8 * g) : zero4();
            // (uniform branches: a launch either has these tensors or not)
            if (e.mask) {
#pragma unroll
                for (int g = 0; g < 4; ++g) mk[g] = bload4(rm, rowok ? (pix * (uint32_t)e.ldm + nn + 8 * g) * 4u : OOB);
            }
            if (e.add1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) a1[g] = bload4(r1, rowok ? (pa * (uint32_t)e.lda1 + nn + 8 * g) * 4u : OOB);
            }
            if (e.add2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) a2[g] = bload4(r2, rowok ? (pix * (uint32_t)e.lda2 + nn + 8 * g) * 4u : OOB);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[c][nb][4 * g], acc[c][nb][4 * g + 1], acc[c][nb][4 * g + 2], acc[c][nb][4 * g + 3]};
                const float bb[4] = {bias[g].x, bias[g].y, bias[g].z, bias[g].w};
                const float m4[4] = {mk[g].x, mk[g].y, mk[g].z, mk[g].w};
                const float x1[4] = {a1[g].x, a1[g].y, a1[g].z, a1[g].w};
                const float x2[4] = {a2[g].x, a2[g].y, a2[g].z, a2[g].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] += bb[u];
                    if (e.add1) v[u] += x1[u];
                    if (e.add2) v[u] += x2[u];
                    if (e.lrelu) v[u] = fmaxf(v[u], leak * v[u]);
                    if (e.mask) v[u] *= m4[u] >= 0.f ? 1.f : LEAK;
                }
                if (rowok) *reinterpret_cast<float4*>(e.out1 + (int64_t)pix * e.ld1 + nn + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row blocks (round 4): the same layer with the y direction's products on SAME padding never formed.
//
// `wconvt_kernel` forms all 25 taps at every position; on a 4x4 / 8x8 grid 28 % / 14 % of those products read the zero halo (a
// whole-step A/B with the loop cut to 18 / 21 taps -- wrong results, right amount of work -- put the ceiling of skipping them at
// -0.37 / -0.59 ms of the 13.4 ms step; profiles/archive/round4_a_ab_upper_bounds.txt).  An MFMA's 32 columns are 32 POSITIONS, so a tap can
// only be dropped where it is invalid for all of them.  Here a block owns ONE GRID ROW r of 64 / WS images (64 positions per parity
// class) x 64 output channels, so "tap (dy, dx) reads input row r + dy" is block-uniform: a block of row 0 runs the 15 taps with
// dy >= 0, one of the last row the 20 with dy <= 0, interior rows all 25 -- (15 + 20 + 25 (HS - 2)) / (25 HS) of the products:
// 85 % on 4x4, 92.5 % on 8x8 (the exact counts are 72.25 % / 85.6 %: the x direction's halo products are still formed, its
// validity varies inside an MFMA's columns).  The tap list of a row type is compile time (three bodies behind one switch), so the
// loop keeps the literal LDS offsets and the woven filter ring of the kernel above.
//   * 4 waves = 2 position halves x 2 channel halves (32 x 32 each, 64 accumulator registers per wave);
//   * LDS: input rows r-1, r, r+1 of the images (rows outside the image are never read, the x halo holds zeros) 41 KB + filter
//     ring 2 x 64 x 36 floats 18 KB: two blocks per CU;
//   * the same FLOPs per filter byte as the 128 x 32 tiles the 4x4 launches used, twice as many per staged input byte;
//   * an image group's rows and column tiles are dealt to ONE XCD (they share input rows).
// Two tile shapes (PB = positions per block):
//   PB = 64  : 64 / WS images of the row x 64 channels; waves = 2 position halves x 2 channel halves; input rows with a zero pixel on
//              either side (pitch WS + 2; 12 on 8-wide rows for the bank map), 41 KB + filter ring 18 KB.
//   PB = 128 : 128 / WS images of the row x 32 channels; waves = 4 position groups -- the filter bytes per FLOP of wconvt_kernel's
//              128-position tiles (they go with 1 / PB; at PB = 64 the 512-image 4x4 launches ran no faster with 15 % fewer MFMAs:
//              L2 -> LDS traffic).  Input rows PACKED: pitch WS + 1, the one zero slot between two rows serves as the right halo of
//              the row before and the left halo of the row after (+ one leading zero slot): 69 KB + filter ring 9 KB, two blocks per
//              CU inside the 160 KB.
__host__ __device__ constexpr bool wr_valid(int t, int RT) { return RT == 0 ? tap_info(t).dy >= 0 : RT == 2 ? tap_info(t).dy <= 0 : true; }
__host__ __device__ constexpr int wr_count(int RT) { int n = 0; for (int t = 0; t < 25; ++t) n += wr_valid(t, RT) ? 1 : 0; return n; }
__host__ __device__ constexpr int wr_tap(int RT, int n) {          // the n-th valid tap of the row type (n taken modulo their number)
    n %= wr_count(RT);
    for (int t = 0; t < 25; ++t)
        if (wr_valid(t, RT)) { if (n == 0) return t; --n; }
    return 0;
}

template <int WS, int PB>
struct WrGeo {
    static constexpr bool PACK = PB == 128;
    static constexpr int IMGT = PB / WS, HP = 3;
    static constexpr int WPL = PACK ? WS + 1 : (WS == 8 ? 12 : WS + 2), LEAD = PACK ? 1 : 0;
    static constexpr int TPIX = LEAD + IMGT * HP * WPL;
    static constexpr int COLS = PB == 128 ? 32 : 64;                     // output channels per block
    static constexpr size_t lds = (size_t)(TPIX * WC_LDP + 2 * COLS * WC_LDP) * sizeof(float);
};

// lane -> (image of the tile, column) of the lane's position.  A ds_read_b128 lane group ({0-3,12-15,20-27} / the rest) must hit 16 pixel
// slots that are distinct mod 16:
//   PB = 64, WS = 8 (image stride 36 slots): a group = two images two apart (72 = 8 mod 16); WS = 4 (stride 18): four images two apart
//   (0, 36, 72, 108 = 0, 4, 8, 12 mod 16);  PB = 128, WS = 4 (packed, stride 15 = -1 mod 16): wave w takes images w + 4 k, a group =
//   k = 0..3 or 4..7 (slots 0, -4, -8, -12 mod 16).
template <int WS, int PB>
__device__ __forceinline__ void wr_lane_pos(int wv, int row, int& il, int& pj) {
    const bool ga = row < 4 || (row >= 12 && row < 16) || (row >= 20 && row < 28);
    const int k = ga ? (row < 4 ? row : row < 16 ? row - 8 : row - 12) : (row < 12 ? row - 4 : row < 20 ? row - 8 : row - 16);   // 0..15 inside the group
    pj = k % WS;
    if constexpr (PB == 128) il = wv + 4 * (k / WS + (ga ? 0 : 16 / WS));
    else il = (wv & 1) * (32 / WS) + (k / WS) * 2 + (ga ? 0 : 1);
}

// XS (PB = 128 only): the x direction too.  Wave w owns ONE COLUMN of the row -- the 32 images' position (r, (w + rot) & 3) -- so every
// MFMA's 32 columns are one grid position and a tap that reads column -1 or WS is skipped by that wave (a scalar branch): the exact
// (5 n - 3)^2 / (5 n)^2 of the products.  The waves of a block then carry 3/5, 1, 1, 4/5 of the x taps; they meet at the filter ring's
// barrier every tap, so the skipped time only counts if another wave uses the SIMD: the launcher hands `rot` = 0 | 2 to the two
// blocks that share a CU (blocks l and l + 32 of an XCD's run), whose waves on one SIMD are then columns (0, 2) / (1, 3) / (2, 0) / (3, 1).
template <int HS, int WS, int RT, int PB, bool XS = false>
__device__ __forceinline__ void wr_body(const WcT& P, float* smem, int img0, int r, int n0, int rot = 0) {
    using G = WrGeo<WS, PB>;
    constexpr int HP = G::HP, WPL = G::WPL, TPIX = G::TPIX, LEAD = G::LEAD, COLS = G::COLS;
    constexpr int NPA = (TPIX * 8 + WC_THREADS - 1) / WC_THREADS;
    constexpr int BSTAGE = COLS * WC_LDP, NPB = COLS * 8 / WC_THREADS;
    constexpr int NT = wr_count(RT);
    static_assert(NPA <= 2 * NT && NPB >= 1, "at most two input float4 of the next slice per tap");
    float* sA = smem;
    float* sB = smem + TPIX * WC_LDP;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int wc = PB == 64 ? wv >> 1 : 0;                                // channel half (PB = 64)
    const int cb = P.c1 + P.c2, ns1 = P.c1 >> 5, nslice = cb >> 5;
    const rsrc_t rs1 = make_rsrc(P.s1), rs2 = make_rsrc(P.s2 ? P.s2 : P.s1);

    // pixel slot hp of the tile: [LEAD zero slots] then (image, row slot 0..2 = input row r - 1 + slot, x slot); data column x sits at
    // x slot x + 1 (plain) / x (packed: x slot WS is the shared zero)
    auto a_load = [&](int slice, int p) -> float4 {
        const int f = tid + WC_THREADS * p;
        const int hp = (f >> 3) - LEAD, k4 = (f & 7) * 4;
        const int il = hp / (HP * WPL), q = hp - il * (HP * WPL), y = r + q / WPL - 1, x = q - (q / WPL) * WPL - (G::PACK ? 0 : 1);
        const int img = img0 + il;
        const bool ok = hp >= 0 && f < TPIX * 8 && (unsigned)y < (unsigned)HS && (unsigned)x < (unsigned)WS && img < P.nimg;
        const bool second = slice >= ns1;
        const int im = second ? img % P.nmod2 : img;
        const int ld = second ? P.c2 : P.c1, kc = (second ? slice - ns1 : slice) * 32;
        const uint32_t v = (uint32_t)(((im * HS + y) * WS + x) * ld + kc + k4) * 4u;
        return bload4(second ? rs2 : rs1, ok ? v : OOB);
    };
    auto a_store = [&](int p, float4 v) {
        const int f = tid + WC_THREADS * p;
        if (f < TPIX * 8) *reinterpret_cast<float4*>(&sA[(f >> 3) * WC_LDP + (f & 7) * 4]) = v;
    };
    const int bk4 = (tid & 7) * 4;
    uint32_t bvoff[NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        const int row = (tid >> 3) + 32 * p;
        bvoff[p] = n0 + row < P.ca ? (uint32_t)((n0 + row) * cb + bk4) * 4u : OOB;
    }
    auto b_load = [&](int slice, int tapw, float4 (&v)[NPB]) {
        if (slice >= nslice) slice = nslice - 1;
        const rsrc_t rr = make_rsrc(P.w + ((int64_t)tapw * P.ca) * cb + slice * 32);
#pragma unroll
        for (int p = 0; p < NPB; ++p) v[p] = bload4(rr, bvoff[p]);
    };
    auto b_store = [&](int stage, const float4 (&v)[NPB]) {
#pragma unroll
        for (int p = 0; p < NPB; ++p) *reinterpret_cast<float4*>(&sB[stage * BSTAGE + ((tid >> 3) + 32 * p) * WC_LDP + bk4]) = v[p];
    };

    // the lane's position (image il of the tile, column pj of grid row r); base = the slot of input pixel (r - 1, pj - 1), so that tap
    // (dy, dx) is the literal offset ((dy + 1) WPL + dx + 1) slots
    int il, pj;
    if constexpr (XS) { il = l31; pj = (__builtin_amdgcn_readfirstlane(wv) + rot) & (WS - 1); }      // (pj in a scalar register: uniform branches below)
    else wr_lane_pos<WS, PB>(wv, l31, il, pj);
    const float* aBase = sA + (LEAD + il * HP * WPL + pj - (G::PACK ? 1 : 0)) * WC_LDP + 4 * h;
    const float* bBase = sB + (32 * wc + l31) * WC_LDP + 4 * h;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;

    float4 areg[NPA], breg[NPB];
#pragma unroll
    for (int p = 0; p < NPA; ++p) areg[p] = a_load(0, p);
    b_load(0, tap_info(wr_tap(RT, 0)).ky * 5 + tap_info(wr_tap(RT, 0)).kx, breg);
#pragma unroll
    for (int p = 0; p < NPA; ++p) a_store(p, areg[p]);
    b_store(0, breg);
    b_load(0, tap_info(wr_tap(RT, 1)).ky * 5 + tap_info(wr_tap(RT, 1)).kx, breg);
    __syncthreads();

    for (int s = 0; s < nslice; ++s) {
        const int snext = s + 1 < nslice ? s + 1 : s;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const TapInfo ti = tap_info(wr_tap(RT, n));
            const int stage = (s * NT + n) & 1;
            const float* aT = aBase + ((ti.dy + 1) * WPL + (ti.dx + 1)) * WC_LDP;
            const float* bT = bBase + stage * BSTAGE;
            // XS: this wave's column has no pixel under the tap
            const bool live = !XS || !((ti.dx < 0 && pj == 0) || (ti.dx > 0 && pj == WS - 1));
            float4 a[2], b[2];
            if (live) {
                a[0] = *reinterpret_cast<const float4*>(aT);
                b[0] = *reinterpret_cast<const float4*>(bT);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (live) {
                    if (q < 3) {
                        a[(q + 1) & 1] = *reinterpret_cast<const float4*>(aT + 8 * (q + 1));
                        b[(q + 1) & 1] = *reinterpret_cast<const float4*>(bT + 8 * (q + 1));
                    }
                    const float av[4] = {a[q & 1].x, a[q & 1].y, a[q & 1].z, a[q & 1].w};
                    const float bv[4] = {b[q & 1].x, b[q & 1].y, b[q & 1].z, b[q & 1].w};
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[ti.cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[tt], av[tt], acc[ti.cls], 0, 0, 0);
                }
                if (q == 0) b_store(stage ^ 1, breg);
                if (q == 1) {
                    const TapInfo tj = tap_info(wr_tap(RT, n + 2));
                    b_load(n + 2 < NT ? s : s + 1, tj.ky * 5 + tj.kx, breg);
                }
                if (q == 2 && n < NPA) areg[n] = a_load(snext, n);
                if (q == 3 && n + NT < NPA) areg[n + NT] = a_load(snext, n + NT);
            }
            __syncthreads();
        }
        if (s + 1 < nslice) {
#pragma unroll
            for (int p = 0; p < NPA; ++p) a_store(p, areg[p]);
            __syncthreads();
        }
    }

    // epilogue: as in wconvt_kernel (column = the lane's position, registers 4g .. 4g + 3 = channels 8g + 4h .. + 3)
    const Epi& e = P.ep;
    const int img = img0 + il;
    const bool rowok = img < P.nimg;
    const rsrc_t rm = make_rsrc(e.mask ? e.mask : P.s1), r1 = make_rsrc(e.add1 ? e.add1 : P.s1), r2 = make_rsrc(e.add2 ? e.add2 : P.s1);
    const float leak = e.lrelu == 2 ? 0.f : LEAK;
    const int nn = n0 + 32 * wc + 4 * h;
    float4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = e.bias ? ldg4(e.bias + nn + 8 * g) : zero4();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const uint32_t pix = (uint32_t)((img * (2 * HS) + 2 * r + py) * (2 * WS) + 2 * pj + px);
        const uint32_t pa = (e.add1_mod && (int64_t)pix >= e.add1_mod) ? pix - (uint32_t)e.add1_mod : pix;
        float4 mk[4], a1[4], a2[4];
        if (e.mask) {
#pragma unroll
            for (int g = 0; g < 4; ++g) mk[g] = bload4(rm, rowok ? (pix * (uint32_t)e.ldm + nn + 8 * g) * 4u : OOB);
        }
        if (e.add1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) a1[g] = bload4(r1, rowok ? (pa * (uint32_t)e.lda1 + nn + 8 * g) * 4u : OOB);
        }
        if (e.add2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) a2[g] = bload4(r2, rowok ? (pix * (uint32_t)e.lda2 + nn + 8 * g) * 4u : OOB);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4] = {acc[c][4 * g], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]};
            const float bb[4] = {bias[g].x, bias[g].y, bias[g].z, bias[g].w};
            const float m4[4] = {mk[g].x, mk[g].y, mk[g].z, mk[g].w};
            const float x1[4] = {a1[g].x, a1[g].y, a1[g].z, a1[g].w};
            const float x2[4] = {a2[g].x, a2[g].y, a2[g].z, a2[g].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] += bb[u];
                if (e.add1) v[u] += x1[u];
                if (e.add2) v[u] += x2[u];
                if (e.lrelu) v[u] = fmaxf(v[u], leak * v[u]);
                if (e.mask) v[u] *= m4[u] >= 0.f ? 1.f : LEAK;
            }
            if (rowok) *reinterpret_cast<float4*>(e.out1 + (int64_t)pix * e.ld1 + nn + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int HS, int WS, int PB, bool XS = false>
__global__ __launch_bounds__(WC_THREADS, 2) void wconvt_row_kernel(const WcT P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using G = WrGeo<WS, PB>;
    // block -> (image group, grid row, column tile).  Blocks are dealt round-robin to the 8 XCDs; XCD x owns image groups x, x + 8, ...
    // Order inside an XCD: ROW-major -- all interior rows (25 taps) of its groups first, then the last rows (20), then the first (15).
    // A launch of these layers is about one round of resident blocks (two per CU), so its time is that of the slowest CU: with the
    // long blocks dispatched first every CU gets a long one, and the short ones fill the second slots -- a (25, 15..20)-tap pair
    // shares the matrix pipe instead of two 25-tap blocks on one CU and two short ones on another (whole step 13.32 -> 13.20 ms against
    // the group-major order, profiles/archive/round4_a_ab_row_blocks.txt).
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int nig = (P.nimg + G::IMGT - 1) / G::IMGT, nigl = (nig + 7) / 8;
    const int nt = l % P.gn, q = l / P.gn;
    const int rsel = q / nigl, ig = (q % nigl) * 8 + xcd;
    if (ig >= nig) return;
    // interior rows first, the shorter border rows behind them
    const int r = rsel < HS - 2 ? rsel + 1 : rsel == HS - 2 ? HS - 1 : 0;
    const int img0 = ig * G::IMGT, n0 = nt * G::COLS;
    const int rot = XS ? ((l >> 5) & 1) * (WS / 2) : 0;       // blocks l and l + 32 of an XCD's run share a CU: complementary columns per SIMD
    if (r == 0) wr_body<HS, WS, 0, PB, XS>(P, smem, img0, r, n0, rot);
    else if (r == HS - 1) wr_body<HS, WS, 2, PB, XS>(P, smem, img0, r, n0, rot);
    else wr_body<HS, WS, 1, PB, XS>(P, smem, img0, r, n0, rot);
}

template <int HS, int WS, int PB, bool XS = false>
void launch_wr(hipStream_t s, WcT P) {
    using G = WrGeo<WS, PB>;
    static_assert(G::lds <= 80 * 1024, "two blocks per CU");
    P.gn = P.ca / G::COLS;
    const int nig = (P.nimg + G::IMGT - 1) / G::IMGT;
    const int items = (nig + 7) / 8 * 8 * HS * P.gn;
    ensure_dyn_lds((const void*)wconvt_row_kernel<HS, WS, PB, XS>, G::lds);
    hipLaunchKernelGGL((wconvt_row_kernel<HS, WS, PB, XS>), dim3((unsigned)items), dim3(WC_THREADS), G::lds, s, P);
}

template <int HS, int WS, int NB>
void launch_wc(hipStream_t s, const WcT& P) {
    constexpr int TH = HS * WS >= WC_ROWS ? WC_ROWS / WS : HS, IMGT = WC_ROWS / (TH * WS), TPI = HS / TH;
    constexpr int TPIX = IMGT * (TH + 2) * (WS == 8 ? 12 : WS + 2);
    constexpr size_t lds = (size_t)(TPIX * WC_LDP + 2 * 32 * NB * WC_LDP) * sizeof(float);
    static_assert(lds <= 65536, "two blocks per CU");
    const int ntile = ((P.nimg + IMGT - 1) / IMGT) * TPI;
    const int items = (ntile + 7) / 8 * 8 * P.gn * P.ksplit;      // whole groups of 8 tiles: the XCD mapping above
    hipLaunchKernelGGL((wconvt_kernel<HS, WS, NB>), dim3((unsigned)items), dim3(WC_THREADS), lds, s, P);
}

}  // namespace

// the shapes this kernel is instantiated for (everything else stays on the implicit GEMM).  Option "wconvt" bit 1.
// nimg: the epilogue and the loaders address a tensor with 32-bit byte offsets ((pixel * ld + n) * 4): the largest of the launch's
// tensors -- output / mask / skip-gradient terms at 4 hs ws pixels x ca channels, inputs at hs ws x max(c1, c2) -- must stay below 4 GiB
bool wconvt_ok(int hs, int ws, int c1, int c2, int ca, int nimg) {
    const int64_t out_bytes = (int64_t)nimg * 4 * hs * ws * ca * 4, in_bytes = (int64_t)nimg * hs * ws * (c1 > c2 ? c1 : c2) * 4;
    return (opt(OPT_WCONVT) & 1) && ((hs == 4 && ws == 4) || (hs == 8 && ws == 8) || (hs == 16 && ws == 16)) && c1 > 0 && c1 % 32 == 0 && c2 % 32 == 0 &&
           ca % 32 == 0 && out_bytes < (1ll << 32) && in_bytes < (1ll << 32);
}

void splitk_reduce(hipStream_t s, const Epi& ep, int M, int N, int nprob, int nsplit);     // kernels.hip

void wconvt_fwd(hipStream_t s, const float* s1, int c1, const float* s2, int c2, int nmod2, int nimg, int hs, int ws, const float* w, int ca,
                const Epi& ep, SplitWs wsp) {
    WcT P{s1, c1, s2, c2, nmod2 > 0 ? nmod2 : 1, w, ca, nimg, 1, 1, nullptr, ep};
    const int img_per = hs * ws >= 128 ? 1 : 128 / (hs * ws), tpi = hs * ws >= 128 ? hs * ws / 128 : 1;
    const int64_t ntile = (int64_t)((nimg + img_per - 1) / img_per) * tpi;
    // 64 output channels per block (half the filter traffic of 32) where tiles x column tiles still offer >= ~400 blocks to the chip's
    // 512 slots; else 32-wide column tiles.  Splitting the channel slices of a FULL launch over 2-4 blocks (partial sums through slabs)
    // was measured on whole steps in round 3 and does not pay (13.75 vs 13.72 ms: faster alone, not beside the side lanes).
    const int nsl = (c1 + c2) / 32;
    const int64_t npix = (int64_t)nimg * 4 * hs * ws;
    bool nb2 = ca % 64 == 0;
    int ks = 1;
    const int slots400 = dev_info().cus * 2 * 25 / 32;                     // ~ 78 % of the resident block slots (400 of 512 on MI355X)
    // STARVED launches (the reward hook's batch of 25: d_h1 offers 16-32 blocks to 256 CUs and each walks 32 slices x 25 taps alone --
    // 0.45 ms of a 1.4 ms translate call, profiles/archive/round4_b_reward_trace_before.txt): 32-wide column tiles and the channel slices over
    // up to 16 blocks, whatever the grid.
    const bool starved = ntile * (ca / 64) * 2 <= dev_info().cus;
    if (starved) {
        nb2 = false;
        // (up to two blocks per CU: a block's tap steps are latency-bound -- 1.2-1.3 us each at one block per CU -- and a second block hides them)
        while (ks < 16 && ntile * (ca / 32) * ks < 2 * dev_info().cus && nsl / (2 * ks) >= 2 && wsp.slab && 2 * ks * npix * ca <= wsp.slab_floats) ks *= 2;
    }
    // row blocks (above): option "wconvt" bit 2 = the 4x4 grids, bit 4 = the 8x8 grids, bit 8 = column-uniform waves on the 4x4 grids
    const int wopt = opt(OPT_WCONVT);
    if (ks == 1 && ca % 64 == 0 && ((hs == 4 && (wopt & 2)) || (hs == 8 && (wopt & 4)))) {
        P.ksplit = 1;
        P.slab = nullptr;
        if (hs == 8) launch_wr<8, 8, 64>(s, P);
        else {
            // 128-position tiles (the filter traffic of the all-taps kernel) where they still make ~2 blocks per CU; else 64 x 64
            const int64_t blocks128 = (int64_t)((nimg + 31) / 32) * 4 * (ca / 32);
            const bool big = blocks128 >= dev_info().cus * 2 * 3 / 4;
            if (big && (wopt & 8)) launch_wr<4, 4, 128, true>(s, P); else if (big) launch_wr<4, 4, 128>(s, P); else launch_wr<4, 4, 64>(s, P);
        }
        return;
    }
    if (ks == 1 && nb2 && ntile * (ca / 64) < slots400) nb2 = false;        // narrower column tiles instead of a half-empty chip
    P.gn = nb2 ? ca / 64 : ca / 32;
    P.ksplit = ks;
    P.slab = ks > 1 ? wsp.slab : nullptr;
    if (hs == 4) { if (nb2) launch_wc<4, 4, 2>(s, P); else launch_wc<4, 4, 1>(s, P); }
    else if (hs == 8) { if (nb2) launch_wc<8, 8, 2>(s, P); else launch_wc<8, 8, 1>(s, P); }
    else { if (nb2) launch_wc<16, 16, 2>(s, P); else launch_wc<16, 16, 1>(s, P); }
    if (ks > 1) {
        Epi er = ep;                                              // rows of the reduction = output pixels in memory order
        er.slab = wsp.slab; er.rowmode = 0;
        splitk_reduce(s, er, (int)npix, ca, 1, ks);
    }
}

}  // namespace ctx
