// c3conv.hip -- conv2d 5x5 (stride 1 | 2, TF SAME) FROM THREE CHANNELS: the frame side of both models -- h0_conv forward
// (arm_shaping.py:1283, :1291, :1633) and the input gradient of d_h4 (conv of d loss / d out with the transposed-conv filter read
// [5,5,3,cb]; :1329, :1669).  Round 3.
//
// These layers are bound by their OUTPUT: 75 multiply-adds per output float, i.e. at the f32 matrix rate the stores alone need
// ~4 TB/s -- the matrix pipe and HBM are both close to busy, so the kernel has to run them TOGETHER.  The generic narrow-channel
// kernel (dconv.h) runs one 8-wave block per CU whose waves compute and then store in lock step (and splits N = 128 into two
// launches): 0.275 ms for ContextSkipNew's d_h4 input gradient against a ~0.09 ms floor.  Here:
//   * 4-wave blocks, two to four per CU (LDS and registers permitting): one block's epilogue stores run under another's MFMAs;
//   * the whole filter (<= 128 columns) stays in LDS for the block's lifetime (persistent blocks), one launch for all columns;
//   * K = 25 taps x 3 channels is walked as 7 chunks of 4 taps: lane group kg takes tap 4 j + kg, the three MFMA steps of a chunk
//     are the three channels -- the zero fourth channel of the [pixel][4] LDS image is never multiplied (21 MFMAs per 16 x 16
//     block instead of 25: 12 % over the 18.75 a dense K would need, was 33 %);
//   * tap offsets are literals: a chunk's A fragment is one ds_read_b128 at (per-lane-group) precomputed byte offsets;
//   * products are formed transposed (rows = output channels), so every store / epilogue load is one float4 per lane;
//   * nothing in the epilogue waits on vmcnt between stores: the bias sits in LDS, the lrelu' mask rows of a tile are loaded in
//     one batch before its first store (the first version loaded the bias per 16-column block, and the s_waitcnt vmcnt(0) in
//     front of each use also waited for the PREVIOUS block's stores -- 8 to 16 store round trips per tile, matrix pipe 42 % busy);
//   * the prefetch addresses are branch-free: per thread and slot the in-tile byte offset and column are packed once, rows outside
//     the frame fall outside the per-frame buffer descriptor by themselves, only the column test remains.
// v_mfma_f32_16x16x4_f32, exact f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "launch.h"

namespace ctx {

namespace {

constexpr int C3_THREADS = 256;
constexpr int C3_NCH = 7;                        // chunks of 4 taps (28 slots, 25 used)
constexpr int C3_PF = 10;                        // prefetch dwords per thread: tile floats (IH * IW * 3) <= 256 * 10

struct C3P {
    const float* x;                              // [nimg, hin, win, 3]
    const float* w;                              // [25][3][N]
    int nimg, hin, win, hout, wout;
    int ntiles, tiles_y, tiles_x;
    Epi ep;
};

// S: stride; NB: output channels / 16; TWB: tile width / 16 (tile = 8 row blocks of 16 pixels: TH = 8 / TWB rows)
template <int S, int NB, int TWB>
__global__ __launch_bounds__(C3_THREADS, (NB <= 4 ? 3 : 2)) void c3conv_kernel(const C3P P) {
    constexpr int N = 16 * NB, TW = 16 * TWB, TH = 8 / TWB, PAD = S == 2 ? 1 : 2;
    constexpr int IH = S * (TH - 1) + 5, IW = S * (TW - 1) + 5, TFL = IH * IW * 3;       // input tile; its floats in HBM order
    static_assert(TFL <= C3_THREADS * C3_PF, "prefetch slots");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                       // [IH * IW][4]
    float* W4 = smem + ((IH * IW * 4 + 3) & ~3);              // [C3_NCH * 4 slots][N][4]
    float* Bs = W4 + C3_NCH * 4 * N * 4;                      // [N] bias (zeros without one)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;

    // ---- filter -> LDS, once per block: slot = 4 j + kg <-> tap; (w[tap][0..2][n], 0); slots >= 25 are zeros
    for (int i = tid; i < C3_NCH * 4 * N; i += C3_THREADS) {
        const int slot = i / N, n = i - slot * N;
        float4 v = zero4();
        if (slot < 25) { const float* q = P.w + (int64_t)slot * 3 * N + n; v = make_float4(q[0], q[N], q[2 * N], 0.f); }
        *reinterpret_cast<float4*>(&W4[(size_t)i * 4]) = v;
    }
    for (int i = tid; i < IH * IW; i += C3_THREADS) tile[i * 4 + 3] = 0.f;    // the fourth channel: written once, never loaded
    for (int i = tid; i < N; i += C3_THREADS) Bs[i] = P.ep.bias ? P.ep.bias[i] : 0.f;

    // ---- input tile prefetch: float e = tid + 256 j of the tile's IH rows of IW * 3 contiguous floats
    unsigned pf[C3_PF];
    auto tile_org = [&](int t, int& img, int& y0, int& x0) {
        const int txi = t % P.tiles_x; t /= P.tiles_x;
        const int tyi = t % P.tiles_y;
        img = t / P.tiles_y; y0 = tyi * TH; x0 = txi * TW;
    };
    // slot j of this thread: float e = tid + 256 j of the tile = row r, float c of the row, walked incrementally from slot 0
    // (256 = RQ rows + RR floats); slots past the tile (e >= TFL) read some float of the tile again and are never landed
    constexpr int RD = IW * 3, RQ = C3_THREADS / RD, RR = C3_THREADS % RD;
    const int r0 = tid / RD, c0 = tid - r0 * RD;
    const unsigned frame_bytes = (unsigned)(P.hin * P.win * 3 * 4), win3 = (unsigned)(P.win * 3);
    auto issue = [&](int t) {
        int img, y0, x0;
        tile_org(t, img, y0, x0);
        // num_records = one frame: rows above / below it (negative or too large offsets) read as zeros by themselves
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x + (int64_t)img * P.hin * P.win * 3), 0, (int)frame_bytes, 0x00020000);
        const int ix0 = (S * x0 - PAD) * 3;
        unsigned roff = (unsigned)((((S * y0 - PAD) + r0) * (int)win3 + ix0) * 4);      // byte offset of (row, float 0 of the tile row)
        int c = c0;
#pragma unroll
        for (int j = 0; j < C3_PF; ++j) {
            const unsigned gx3 = (unsigned)(ix0 + c);
            pf[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, gx3 < win3 ? roff + 4u * (unsigned)c : OOB, 0, 0);
            c += RR; roff += (unsigned)RQ * win3 * 4u;
            const bool wrap = c >= RD;
            c = wrap ? c - RD : c; roff = wrap ? roff + win3 * 4u : roff;
        }
    };
    auto land = [&]() {
#pragma unroll
        for (int j = 0; j < C3_PF; ++j) {
            const int e = tid + C3_THREADS * j, r = e / (IW * 3), c = e - r * (IW * 3), px = c / 3, ch = c - px * 3;
            if (e < TFL) tile[(r * IW + px) * 4 + ch] = __uint_as_float(pf[j]);
        }
    };

    // ---- fragment addresses.  Wave wv owns row blocks rb = 2 wv, 2 wv + 1: tile row rb / TWB, columns 16 (rb % TWB) + l15.
    int abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int rb = 2 * wv + m, ty = rb / TWB, tx = rb - ty * TWB;
        abase[m] = ((S * ty) * IW + S * (16 * tx + l15)) * 16;            // bytes
    }
    int toff[C3_NCH];                                                     // tap 4 j + kg of this lane group: (ky * IW + kx) pixels
#pragma unroll
    for (int j = 0; j < C3_NCH; ++j) {
        int tap = 4 * j + kg;
        tap = tap < 25 ? tap : 24;                                        // padded slots: any valid address (their filter rows are zero)
        toff[j] = ((tap / 5) * IW + tap % 5) * 16;
    }
    const char* bbase = reinterpret_cast<const char*>(W4) + (size_t)(kg * N + l15) * 16;
    const float leak = P.ep.lrelu == 2 ? 0.f : LEAK;

    int t = blockIdx.x;
    if (t < P.ntiles) issue(t);
    for (; t < P.ntiles; t += gridDim.x) {
        __syncthreads();                                       // the previous tile's fragments are consumed
        land();
        __syncthreads();                                       // tile (first pass: and the filter) visible
        if (t + (int)gridDim.x < P.ntiles) issue(t + gridDim.x);

        // output pixels of this lane's two row blocks
        int img, y0, x0;
        tile_org(t, img, y0, x0);
        int pix[2];                                              // (32-bit element offsets: every activation tensor of a handle is < 2 GiB, ctx_create)
        bool okm[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int rb = 2 * wv + m, ty = rb / TWB, tx = rb - ty * TWB;
            const int y = y0 + ty, x = x0 + 16 * tx + l15;
            okm[m] = y < P.hout && x < P.wout;
            pix[m] = okm[m] ? (img * P.hout + y) * P.wout + x : 0;                       // (a pixel that exists)
        }

        // The columns go in passes of NP <= 4 blocks of 16 (N = 128: two passes over the same LDS tile): 32 accumulator registers
        // instead of 64, and a pass's lrelu' mask rows are requested BEFORE its MFMA loop when there are two passes
        constexpr int NP = NB > 4 ? 4 : NB, NPASS = NB / NP;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const bool masked = P.ep.mask && 16 * NP * ps < P.ep.nsplit;
            float4 tm[2][NP];
            auto fetch_mask = [&]() {
                if (masked) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const int n = 16 * (NP * ps + q) + 4 * kg;
                            tm[m][q] = ldg4(P.ep.mask + (unsigned)(pix[m] * (int)P.ep.ldm + (n < P.ep.nsplit ? n : 0)));
                        }
                }
            };
            if (NPASS > 1) fetch_mask();

            f32x4 acc[2][NP];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < NP; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            float4 a[2][2], b[2][NP];
            auto fetch = [&](int j, int buf) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[buf][m] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tile) + abase[m] + toff[j]);
#pragma unroll
                for (int q = 0; q < NP; ++q) b[buf][q] = *reinterpret_cast<const float4*>(bbase + (size_t)j * (4 * N * 16) + (NP * ps + q) * 256);
            };
            fetch(0, 0);
#pragma unroll
            for (int j = 0; j < C3_NCH; ++j) {
                if (j + 1 < C3_NCH) fetch(j + 1, (j + 1) & 1);
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)                      // the three channels; the zero fourth one is skipped
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const float4 av4 = a[j & 1][m];
                        const float av = tt == 0 ? av4.x : tt == 1 ? av4.y : av4.z;
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            const float4 bv4 = b[j & 1][q];
                            const float bv = tt == 0 ? bv4.x : tt == 1 ? bv4.y : bv4.z;
                            acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[m][q], 0, 0, 0);   // D^T: rows = channels, cols = pixels
                        }
                    }
            }

            // ---- epilogue of the pass: a lane holds channels 16 nb + 4 kg .. + 3 of pixel l15 of each of its two row blocks.
            // No global load sits between the stores; the bias comes from LDS
            if (NPASS == 1) fetch_mask();
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int n = 16 * (NP * ps + q) + 4 * kg;
                    const float4 bb = *reinterpret_cast<const float4*>(&Bs[n]);
                    float v[4] = {acc[m][q][0] + bb.x, acc[m][q][1] + bb.y, acc[m][q][2] + bb.z, acc[m][q][3] + bb.w};
                    if (P.ep.lrelu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], leak * v[r]);
                    }
                    const bool first = n < P.ep.nsplit;
                    if (masked && first) {
                        v[0] *= tm[m][q].x >= 0.f ? 1.f : LEAK; v[1] *= tm[m][q].y >= 0.f ? 1.f : LEAK;
                        v[2] *= tm[m][q].z >= 0.f ? 1.f : LEAK; v[3] *= tm[m][q].w >= 0.f ? 1.f : LEAK;
                    }
                    float* dst = first ? P.ep.out1 + (unsigned)(pix[m] * (int)P.ep.ld1 + n) : P.ep.out2 + (unsigned)(pix[m] * (int)P.ep.ld2 + (n - P.ep.nsplit));
                    if (okm[m]) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

template <int S, int NB, int TWB>
void launch_c3(hipStream_t s, C3P P) {
    constexpr int N = 16 * NB, TW = 16 * TWB, TH = 8 / TWB;
    constexpr int IH = S * (TH - 1) + 5, IW = S * (TW - 1) + 5;
    constexpr size_t lds = (size_t)(((IH * IW * 4 + 3) & ~3) + C3_NCH * 4 * N * 4 + N) * sizeof(float);
    P.tiles_y = (P.hout + TH - 1) / TH;
    P.tiles_x = (P.wout + TW - 1) / TW;
    P.ntiles = P.nimg * P.tiles_y * P.tiles_x;
    ensure_dyn_lds((const void*)c3conv_kernel<S, NB, TWB>, lds);
    int per_cu = (int)(dev_info().lds_per_cu / (lds + 512));
    const int cap = NB <= 2 ? 4 : NB <= 4 ? 3 : 2;           // waves per SIMD the registers allow (kernel-resource-usage: 109 / 151 / 243 VGPRs)
    per_cu = per_cu < 1 ? 1 : per_cu > cap ? cap : per_cu;
    int grid = dev_info().cus * per_cu;
    if (grid > P.ntiles) grid = P.ntiles;
    { const int rounds = (P.ntiles + grid - 1) / grid; grid = (P.ntiles + rounds - 1) / rounds; }     // whole rounds of tiles per block
    hipLaunchKernelGGL((c3conv_kernel<S, NB, TWB>), dim3((unsigned)grid), dim3(C3_THREADS), lds, s, P);
}

template <int S, int NB>
void launch_c3_w(hipStream_t s, const C3P& P) {
    if (P.wout % 32 == 0) launch_c3<S, NB, 2>(s, P);
    else launch_c3<S, NB, 1>(s, P);
}

}  // namespace

// the shapes this kernel is instantiated for
bool c3conv_ok(int hin, int win, int stride, int N, const Epi& ep) {
    if (!(opt(OPT_DIRECT3) & 2) || (stride != 1 && stride != 2) || hin % stride || win % stride) return false;
    if (N != 32 && N != 64 && N != 128) return false;
    if ((win / stride) % 16) return false;                               // whole 16-pixel row blocks
    if (ep.add1 || ep.add2 || ep.slab || ep.rowmode) return false;
    auto m4 = [](int64_t v) { return v % 4 == 0; };
    if (!m4(ep.ld1) || (ep.nsplit < N && (!m4(ep.nsplit) || !m4(ep.ld2))) || (ep.mask && !m4(ep.ldm))) return false;
    return (int64_t)hin * win * 3 * 4 < (1ll << 31);
}

// y = epilogue(conv2d(x[nimg, hin, win, 3], w[5][5][3][N], stride, SAME))
void c3conv(hipStream_t s, const float* x, int nimg, int hin, int win, int stride, const float* w, int N, const Epi& ep) {
    C3P P{x, w, nimg, hin, win, hin / stride, win / stride, 0, 0, 0, ep};
    if (stride == 2) {
        if (N == 128) launch_c3_w<2, 8>(s, P); else if (N == 64) launch_c3_w<2, 4>(s, P); else launch_c3_w<2, 2>(s, P);
    } else {
        if (N == 128) launch_c3_w<1, 8>(s, P); else if (N == 64) launch_c3_w<1, 4>(s, P); else launch_c3_w<1, 2>(s, P);
    }
}

}  // namespace ctx
