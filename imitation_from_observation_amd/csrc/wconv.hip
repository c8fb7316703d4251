// wconv.hip -- conv2d 5x5 stride 2 SAME (arm_shaping.py:21-32) for WIDE channel counts onto the 16x16 and 8x8 output grids, image-major
// with the INPUT halo tile resident in LDS: the forward convolutions h1 / h2 of both encoders and -- the same operation -- the decoder's
// input gradients (d_h3 / d_h2 dx: a stride-2 conv of dy with the filter read [5,5,ca,cb]).  Round 4; the mirror of wconvt.hip.
//
// STATUS: option "wconv", OFF by default.  Measured (round 4, B = 256): 136-141 TF/s on all taps against the implicit GEMM's 128-139
// (h1 fwd 0.393 vs 0.411 ms, d_h3 dx 0.782 vs 0.814, h2 fwd 0.387 vs 0.397, d_h2 dx 0.760 vs 0.773) -- and the whole step 0.05-0.10 ms
// SLOWER in four A/B pairs (13.06 / 13.00 / 13.11 / 13.12 with, 12.96 / 12.99 / 13.05 / 13.07 without).  The premise was wrong: wconvt.hip's
// "0.96" is its all-taps figure on launches that skip rows; the matrix pipe's real rate in every kernel here is 0.82-0.90 of nominal,
// the sustained clock under load (notebook section 6: 2.04-2.26 of 2.4 GHz), and this kernel is at it like the others.  Kept as the
// measured alternative, covered by tests/test_gpu_parity.py (option value 2).
//
// Why (as designed): the position-major implicit GEMM (igemm.h: KmConvGatherQ) streams BOTH operands of every 128x128x32 chunk from L2 and re-fetches
// an input pixel for every tap it serves (6.25 times at stride 2); it runs these launches at 0.74-0.77 of the f32 matrix peak on the
// products it forms, where wconvt.hip -- input tile staged once per channel slice, only the filter streamed -- reaches 0.96.  Here a
// block owns 128 output positions (16x16 grid: 8 rows of one image; 8x8 grid: two whole images) x 128 output channels:
//   * a 16-channel slice of the tile's input pixels (19 rows x (2 WO + 3) columns per image, zeros outside the image: SAME pads 1 before
//     and 2 after) is staged ONCE and serves all 25 taps: 25 x 32 MFMAs per wave per 55-61 KB staged;
//   * the input columns are stored DE-INTERLEAVED -- even and odd columns of a row in two planes -- so that the stride-2 gather of a tap
//     (input column 2 j + kx - 1) reads consecutive slots for consecutive j: a ds_read_b128 lane group of 16 hits 16 slots distinct
//     mod 16 (slot pitch 20 floats = 5 sixteen-byte units, odd).  8-wide grids: a group is two output rows = four plane rows apart,
//     4 (WO + 2) = 40 = 8 mod 16 slots: conflict free as well;
//   * the filter goes through a two-stage LDS ring, one (tap, slice) = 16 x 128 floats at a time, transposed on the way in to [n][k]
//     (row pitch 20) so that its fragments are ds_read_b128 too; one barrier per tap = per 32 MFMAs of every wave;
//   * 4 waves = 2 position halves x 2 channel halves (64 x 64 each, 64 accumulator registers); the tap -> slot shift table is compile
//     time, so a tap's A fragment is one ds_read_b128 at a literal offset from the lane's base address.
// All 25 taps are formed for every position (zeros in the halo): 7.4 % / 14.4 % of the products on 16x16 / 8x8 output grids meet a
// zero; the 4x4 grids (28 %) stay on the position-major implicit GEMM, which skips them exactly.
// v_mfma_f32_32x32x2_f32, exact f32.  k order inside a slice: step (q, t): lanes 0-31 take channel 8q + t, lanes 32-63 8q + 4 + t.
// The filter value is the MFMA's row operand (D rows = output channels, columns = positions): a lane ends up with 4 consecutive
// channels of one position in 4 consecutive registers -- float4 epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "launch.h"

namespace ctx {

namespace {

constexpr int WV_THREADS = 256;                 // 4 waves; two blocks share a CU (LDS 70-77 KB)
constexpr int WV_KS = 16;                       // channels per staged slice
constexpr int WV_AP = 20;                       // floats per input slot (16 channels + 4 pad)
constexpr int WV_COLS = 128;                    // output channels per block
constexpr int WV_BP = 20;                       // floats per filter row in LDS: [n][16 k + 4 pad]
constexpr int WV_TR = 8;                        // output rows per tile

struct WvT {
    const float* x; int ci;                     // input [nimg, 2 HO, 2 WO, ci]
    const float* w; int co;                     // filter [5][5][ci][co]
    int nimg;
    int gn;                                     // column tiles
    Epi ep;                                     // bias / lrelu / mask / nsplit + out2 (block-uniform: nsplit a multiple of 128)
};

template <int HO, int WO>
struct WvGeo {
    static constexpr int HI = 2 * HO, WI = 2 * WO;
    static constexpr int IMGT = 128 / (WV_TR * WO), TPI = HO / WV_TR;     // images per tile, tiles per image
    static constexpr int YR = 2 * WV_TR + 3, WP2 = WO + 2;                 // staged input rows; slots per column plane
    static constexpr int IMS = YR * 2 * WP2, TSLOT = IMGT * IMS;          // slots per image, per tile
    static constexpr int NPA = (TSLOT * (WV_KS / 4) + WV_THREADS - 1) / WV_THREADS;
    static constexpr int BSTAGE = WV_COLS * WV_BP;
    static constexpr size_t lds = (size_t)(TSLOT * WV_AP + 2 * BSTAGE) * sizeof(float);
    static_assert(IMGT * WV_TR * WO == 128 && HO % WV_TR == 0, "tile shape");
};

template <int HO, int WO>
__global__ __launch_bounds__(WV_THREADS, 2) void wconv_kernel(const WvT P) {
    using G = WvGeo<HO, WO>;
    constexpr int HI = G::HI, WI = G::WI, IMGT = G::IMGT, TPI = G::TPI, WP2 = G::WP2, IMS = G::IMS, TSLOT = G::TSLOT, NPA = G::NPA, BSTAGE = G::BSTAGE;
    static_assert(NPA <= 25 && WV_KS * WV_COLS == 8 * WV_THREADS, "one input float4 of the next slice per tap; 8 filter values per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;
    float* sB = smem + TSLOT * WV_AP;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int wp = wv & 1, wc = wv >> 1;                                  // position half, channel half
    // block -> (tile, column tile): the column tiles of one tile sit on ONE XCD, next to each other in dispatch order (the second one
    // finds the input pixels in that XCD's L2)
    int item = blockIdx.x;
    {
        const int xcd = item & 7, l = item >> 3, nt = l % P.gn, g8 = l / P.gn;
        item = (g8 * 8 + xcd) * P.gn + nt;
    }
    const int ntile = ((P.nimg + IMGT - 1) / IMGT) * TPI;
    const int tile = item / P.gn, ctile = item - tile * P.gn;
    if (tile >= ntile) return;
    const int img0 = (tile / TPI) * IMGT, i0 = (tile % TPI) * WV_TR, n0 = ctile * WV_COLS;
    const int nslice = P.ci / WV_KS;
    const rsrc_t rsx = make_rsrc(P.x);

    // ---- loaders -----------------------------------------------------------------------------------------------
    // A: float4 f = tid + 256 p of the slice tile: slot f >> 2 = (image, tile row y', column parity, plane column c), channels 4 (f & 3) ..
    //    tile row y' <-> input row 2 i0 + y' - 1; tile column x' = 2 c + parity <-> input column x' - 1
    auto a_load = [&](int slice, int p) -> float4 {
        const int f = tid + WV_THREADS * p;
        const int slot = f >> 2, k4 = (f & 3) * 4;
        const int il = slot / IMS, rem = slot - il * IMS, rp = rem / WP2, c = rem - rp * WP2;
        const int iy = 2 * i0 + (rp >> 1) - 1, ix = 2 * c + (rp & 1) - 1;
        const int img = img0 + il;
        const bool ok = f < TSLOT * 4 && (unsigned)iy < (unsigned)HI && (unsigned)ix < (unsigned)WI && img < P.nimg;
        const uint32_t v = (uint32_t)(((img * HI + iy) * WI + ix) * P.ci + slice * WV_KS + k4) * 4u;
        return bload4(rsx, ok ? v : OOB);
    };
    auto a_store = [&](int p, float4 v) {
        const int f = tid + WV_THREADS * p;
        if (f < TSLOT * 4) *reinterpret_cast<float4*>(&sA[(f >> 2) * WV_AP + (f & 3) * 4]) = v;
    };
    // B: (tap, slice) tile = w[tap][slice * 16 + k][n0 + n], k < 16, n < 128 -- in memory n runs fastest, the MFMA wants 4 consecutive k of
    // one n in a register quad.  Thread (n = tid & 127, kh = tid >> 7) fetches its 8 values k = 8 kh .. + 7 with 8 dword loads (a wave's
    // 64 lanes read 256 contiguous bytes each time) and writes them as two float4 to sB[n][8 kh ..]: rows 20 floats apart, conflict free.
    const uint32_t bv0 = (uint32_t)((8 * (tid >> 7)) * P.co + n0 + (tid & 127)) * 4u, bstep = (uint32_t)P.co * 4u;
    auto b_load = [&](int slice, int tap, float (&v)[8]) {
        if (slice >= nslice) slice = nslice - 1;                           // (redundant tail reloads are harmless)
        const rsrc_t r = make_rsrc(P.w + ((int64_t)tap * P.ci + slice * WV_KS) * P.co);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, bv0 + u * bstep, 0, 0));
    };
    auto b_store = [&](int stage, const float (&v)[8]) {
        float* q = &sB[stage * BSTAGE + (tid & 127) * WV_BP + 8 * (tid >> 7)];
        *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------
    // MFMA column l31 of position block blk = 2 wp + mi -> position 32 blk + (group A ? 0 : 16) + k, row-major over (image, row, column):
    // the ds_read_b128 lane groups ({0-3,12-15,20-27} / the rest) are 16 consecutive positions = one 16-wide row or two 8-wide rows
    const bool ga = l31 < 4 || (l31 >= 12 && l31 < 16) || (l31 >= 20 && l31 < 28);
    const int kk = ga ? (l31 < 4 ? l31 : l31 < 16 ? l31 - 8 : l31 - 12) : (l31 < 12 ? l31 - 4 : l31 < 20 ? l31 - 8 : l31 - 16);
    int pil[2], pir[2], pj[2];
    const float* aBase[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int pos = 32 * (2 * wp + mi) + (ga ? 0 : 16) + kk;
        pil[mi] = pos / (WV_TR * WO);
        const int rem = pos - pil[mi] * (WV_TR * WO);
        pir[mi] = rem / WO; pj[mi] = rem - pir[mi] * WO;
        aBase[mi] = sA + (pil[mi] * IMS + 4 * pir[mi] * WP2 + pj[mi]) * WV_AP + 4 * h;
    }
    const float* bBase = sB + (64 * wc + l31) * WV_BP + 4 * h;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // ---- prologue: slice 0 and tap 0 into LDS, tap 1 into the registers ------------------------------------------
    float4 areg[NPA];
    float breg[8];
#pragma unroll
    for (int p = 0; p < NPA; ++p) areg[p] = a_load(0, p);
    b_load(0, 0, breg);
#pragma unroll
    for (int p = 0; p < NPA; ++p) a_store(p, areg[p]);
    b_store(0, breg);
    b_load(0, 1, breg);
    __syncthreads();

    for (int s = 0; s < nslice; ++s) {
        const int snext = s + 1 < nslice ? s + 1 : s;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            const int ky = t / 5, kx = t - ky * 5;
            const int stage = (s * 25 + t) & 1;
            const int aoff = ((2 * ky + (kx & 1)) * WP2 + (kx >> 1)) * WV_AP;      // a literal per tap
            const float* bT = bBase + stage * BSTAGE;
            float4 a[2][2], b[2][2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[0][mi] = *reinterpret_cast<const float4*>(aBase[mi] + aoff);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[0][ni] = *reinterpret_cast<const float4*>(bT + 32 * ni * WV_BP);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q < 1) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) a[1][mi] = *reinterpret_cast<const float4*>(aBase[mi] + aoff + 8);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) b[1][ni] = *reinterpret_cast<const float4*>(bT + 32 * ni * WV_BP + 8);
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        const float4 aq = a[q][mi];
                        const float av = tt == 0 ? aq.x : tt == 1 ? aq.y : tt == 2 ? aq.z : aq.w;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const float4 bq = b[q][ni];
                            const float bv = tt == 0 ? bq.x : tt == 1 ? bq.y : tt == 2 ? bq.z : bq.w;
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[mi][ni], 0, 0, 0);
                        }
                    }
                }
                // in the gaps: the filter tile of tap g + 1 goes to the other stage, the one of tap g + 2 is requested, and the next
                // slice's input pixels are requested one float4 per tap (they sit in registers until the slice ends)
                if (q == 0) {
                    b_store(stage ^ 1, breg);
                    const int t2 = t + 2 < 25 ? t + 2 : t + 2 - 25;
                    b_load(t + 2 < 25 ? s : s + 1, t2, breg);
                }
                if (q == 1 && t < NPA) areg[t] = a_load(snext, t);
            }
            __syncthreads();
        }
        if (s + 1 < nslice) {                     // every wave has finished with this slice's pixels (the barrier above)
#pragma unroll
            for (int p = 0; p < NPA; ++p) a_store(p, areg[p]);
            __syncthreads();
        }
    }

    // ---- epilogue.  D = W^T x In^T: column = l31 = the lane's position, row = (r & 3) + 8 (r >> 2) + 4 h = output channel inside the
    // 32-wide channel block: registers 4g .. 4g + 3 are channels 8g + 4h .. + 3 -- one float4 per access.
    const Epi& e = P.ep;
    const bool second = n0 >= e.nsplit;                                    // block-uniform: the skip-gradient half of a decoder dx
    float* const outp = second ? e.out2 : e.out1;
    const int ldo = (int)(second ? e.ld2 : e.ld1), nbase = second ? n0 - e.nsplit : n0;
    const bool masked = e.mask && !second;
    const rsrc_t rm = make_rsrc(masked ? e.mask : P.x);
    const float leak = e.lrelu == 2 ? 0.f : LEAK;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int img = img0 + pil[mi];
        const bool rowok = img < P.nimg;
        const uint32_t pix = (uint32_t)((img * HO + i0 + pir[mi]) * WO + pj[mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int nl = 64 * wc + 32 * ni + 4 * h;                      // channel inside the block's 128
            float4 bias[4], mk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias[g] = e.bias ? ldg4(e.bias + n0 + nl + 8 * g) : zero4();
            if (masked) {
#pragma unroll
                for (int g = 0; g < 4; ++g) mk[g] = bload4(rm, rowok ? (pix * (uint32_t)e.ldm + n0 + nl + 8 * g) * 4u : OOB);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
                const float bb[4] = {bias[g].x, bias[g].y, bias[g].z, bias[g].w};
                const float m4[4] = {mk[g].x, mk[g].y, mk[g].z, mk[g].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] += bb[u];
                    if (e.lrelu) v[u] = fmaxf(v[u], leak * v[u]);
                    if (masked) v[u] *= m4[u] >= 0.f ? 1.f : LEAK;
                }
                if (rowok) *reinterpret_cast<float4*>(outp + (int64_t)pix * ldo + nbase + nl + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int HO, int WO>
void launch_wv(hipStream_t s, WvT P) {
    using G = WvGeo<HO, WO>;
    static_assert(G::lds <= 80 * 1024, "two blocks per CU");
    P.gn = P.co / WV_COLS;
    const int ntile = ((P.nimg + G::IMGT - 1) / G::IMGT) * G::TPI;
    const int items = (ntile + 7) / 8 * 8 * P.gn;                          // whole groups of 8 tiles: the XCD mapping above
    ensure_dyn_lds((const void*)wconv_kernel<HO, WO>, G::lds);
    hipLaunchKernelGGL((wconv_kernel<HO, WO>), dim3((unsigned)items), dim3(WV_THREADS), G::lds, s, P);
}

}  // namespace

// the shapes this kernel is instantiated for (everything else stays on the implicit GEMM).  Option "wconv".
// ho, wo: OUTPUT grid.  The epilogue forms of a forward conv (bias, lrelu) and of the decoder's input gradient (mask on the columns
// below nsplit, the rest to out2) only; 32-bit byte offsets into every tensor.
bool wconv_ok(int ho, int wo, int ci, int co, int nimg, const Epi& ep) {
    const int64_t in_bytes = (int64_t)nimg * 4 * ho * wo * ci * 4, out_bytes = (int64_t)nimg * ho * wo * co * 4;
    const bool split_ok = ep.nsplit >= co || (ep.nsplit % WV_COLS == 0 && ep.out2);
    // enough tiles for about a round of the chip's two-per-CU block slots: below that the position-major launch spreads better
    const int ntile = ho == 16 ? nimg * 2 : (nimg + 1) / 2;
    return opt(OPT_WCONV) && ((ho == 16 && wo == 16) || (ho == 8 && wo == 8)) && ci % WV_KS == 0 && co % WV_COLS == 0 && !ep.add1 && !ep.add2 &&
           !ep.slab && !ep.prob_stride && ep.out1 && split_ok && in_bytes < (1ll << 32) && out_bytes < (1ll << 32) &&
           (opt(OPT_WCONV) >= 2 || (int64_t)ntile * (co / WV_COLS) >= dev_info().cus * 3 / 2);      // (2 = whatever the launch size: tests)
}

void wconv_fwd(hipStream_t s, const float* x, int ci, int nimg, int ho, int wo, const float* w, int co, const Epi& ep) {
    WvT P{x, ci, w, co, nimg, 1, ep};
    if (ho == 16) launch_wv<16, 16>(s, P); else launch_wv<8, 8>(s, P);
}

}  // namespace ctx
