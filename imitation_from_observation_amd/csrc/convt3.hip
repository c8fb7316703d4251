// convt3.hip -- conv2d_transpose 5x5 (stride 1 or 2, SAME) from a wide concat [decoder | ctx skip] to the THREE image channels:
// d_h4 of every translator (gym/envs/mujoco/arm_shaping.py:1329-1330 ContextSkipNew, :1671-1672 ContextAEReal; deconv2d :62-85).
//
// Three output channels do not fill a matrix-core tile (N = 3 of 16 is 19 %), and the earlier route -- an MFMA "scatter product"
// P[pixel][tap * 3 + c] (75 columns, 168-377 MB through HBM) followed by a gather of the taps that land on each output pixel --
// moved 4-8x the bytes of the tensors involved.  This kernel computes the layer directly on the vector ALUs, ONE pass over the
// input, nothing in between:
//   * a block owns an output tile; the input halo tile is staged in LDS in slices of 16 channels (pixel stride 20 floats: the
//     ds_read_b128 of 16 consecutive pixels are conflict-free), the slice of the filter ([tap][c][16 k]) beside it; the next
//     slice's global loads are in flight (registers) under the current slice's arithmetic;
//   * a lane owns P consecutive output ROWS of one column (stride 2: the 2 x 2 output pixels of P small-grid rows x 1 column):
//     for a filter column kx it reads the P + 4 (stride 2: P + 2) input vectors of its column once and uses them for every row
//     and every ky -- 240 FMAs per 8 vector reads + 15 broadcast weight reads, so the LDS pipe stays far from the bound;
//   * bias added, NHWC store of 3 (stride 2: 6 contiguous) floats per lane: full-width coalesced rows.
// Exact f32 (two fmaf chains per output element -- even and odd channels, one v_pk_fma_f32 each -- in slice order: deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>

#include "launch.h"

namespace ctx {

namespace {

// Channels per staged slice, by stride.  16 at stride 1 (ContextAEReal: 64 B of a pixel's 128-B line per pass; with 8 the kernel
// fetched 4x its tensors -- every line once per slice, gone from L2 by the next -- and was bound by that: 0.257 -> 0.227 ms).
// 8 at stride 2 (ContextSkipNew: 48 accumulators per lane; the wider slice's prefetch registers cost more than the re-fetch:
// 0.19 ms with 8, 0.27 ms with 16).  LDS pixel stride KS + 4 floats: 16 consecutive pixels' ds_read_b128 hit 16 distinct 4-bank slots.
__host__ __device__ constexpr int ct3_ks(int S) { return S == 1 ? 16 : 8; }
__host__ __device__ constexpr int ct3_pf(int S) { return S == 1 ? 16 : 6; }   // prefetch slots (float4) per thread: IH * IW * KS / 4 <= threads * this

struct Ct3 {
    const float* x1; int c1;                       // decoder stream [nimg][hin][win][c1]
    const float* x2; int c2; int nmod2;            // ctx skip [nmod2][hin][win][c2], image index img % nmod2
    int CI;                                        // c1 + c2; c1 and c2 multiples of KS
    int hin, win, nimg;
    const float* w;                                // [25][3][CI]  (the reference's [5,5,out,in])
    const float* bias;                             // [3]
    float* out;                                    // [nimg][S*hin][S*win][3]
    int TW, TR;                                    // tile of the input-resolution grid: TW columns x TR rows
    int IH, IW;                                    // staged input tile: TR + 2*HALO, TW + 2*HALO
    int tiles_y, tiles_x, ntiles;
};

typedef unsigned ct_u32x4 __attribute__((ext_vector_type(4)));
typedef float ct_f2 __attribute__((ext_vector_type(2)));     // one v_pk_fma_f32 = two FMAs: an accumulator is (even k, odd k) partial sums

template <int S /* stride */, int P /* rows per lane */, int NT /* threads */>
__global__ __launch_bounds__(NT) void convt3_kernel(const Ct3 A) {
    constexpr int CT3_PF = ct3_pf(S);
    constexpr int KS = ct3_ks(S), PS = KS + 4, KQ = KS / 4, WQ = 25 * 3 * KQ;      // slice channels, LDS pixel stride, float4s per pixel / per filter slice
    constexpr int HALO = S == 1 ? 2 : 1;           // stride 1: rows y - 2 .. y + 2; stride 2 (small grid): i - 1 .. i + 1
    constexpr int NR = P + 2 * HALO;               // input rows a lane touches per column
    constexpr int NACC = S == 1 ? P * 3 : 4 * P * 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                        // [IH][IW][PS]
    float* wsl = smem + ((A.IH * A.IW * PS + 3) & ~3);         // [25][4][KS]  (c = 3: unused row)
    const int tid = threadIdx.x;
    const int xl = tid % A.TW, rg = tid / A.TW;                // column in the tile, row group (P rows each)
    // slot j of a thread = float4 tid + NT j of the slice's tile; its (row, column) by recurrence from the thread's first slot (a
    // fixed step of rows / columns and one wrap) instead of two divisions per slot and slice
    constexpr int SPX = NT / KQ;                               // pixels a slot step advances (NT and KQ powers of two or 384 / {2, 4})
    const int qy = SPX / A.IW, rx = SPX - qy * A.IW;
    const int e0 = tid / KQ, cth = (tid % KQ) * 4;
    const int iyb = e0 / A.IW, ixb = e0 - iyb * A.IW;
    const float bias0 = A.bias[0], bias1 = A.bias[1], bias2 = A.bias[2];

    for (int t = blockIdx.x; t < A.ntiles; t += gridDim.x) {
        int q = t;
        const int txi = q % A.tiles_x; q /= A.tiles_x;
        const int tyi = q % A.tiles_y;
        const int img = q / A.tiles_y;
        const int y0 = tyi * A.TR, x0 = txi * A.TW;            // tile origin on the input-resolution grid
        const float* s1 = A.x1 + (int64_t)img * A.hin * A.win * A.c1;
        const float* s2 = A.x2 + (int64_t)(img % A.nmod2) * A.hin * A.win * A.c2;

        ct_u32x4 pf[CT3_PF];
        ct_u32x4 wpf[(WQ + NT - 1) / NT];
        auto issue = [&](int k0) {                             // slice [k0, k0 + KS) of the input tile and of the filter -> registers
            const bool first = k0 < A.c1;
            const float* sp = first ? s1 + k0 : s2 + (k0 - A.c1);
            const int ld = first ? A.c1 : A.c2;
            int iy = iyb, ix = ixb;
#pragma unroll
            for (int j = 0; j < CT3_PF; ++j) {
                const int gy = y0 - HALO + iy, gx = x0 - HALO + ix;
                const bool ok = iy < A.IH && (unsigned)gy < (unsigned)A.hin && (unsigned)gx < (unsigned)A.win;
                const int pix = ok ? gy * A.win + gx : 0;      // halo lanes read a pixel that exists and are zeroed when they land
                pf[j] = *reinterpret_cast<const ct_u32x4*>(sp + (int64_t)pix * ld + cth);
                ix += rx; iy += qy;
                if (ix >= A.IW) { ix -= A.IW; ++iy; }
            }
            // filter slice: 25 taps x 3 channels x KS k
#pragma unroll
            for (int u = 0; u < (WQ + NT - 1) / NT; ++u) {
                const int i = tid + u * NT, tc = i / KQ, c4 = (i % KQ) * 4;
                wpf[u] = i < WQ ? *reinterpret_cast<const ct_u32x4*>(A.w + (int64_t)tc * A.CI + k0 + c4) : ct_u32x4{0u, 0u, 0u, 0u};
            }
        };
        auto land = [&]() {
            int iy = iyb, ix = ixb;
#pragma unroll
            for (int j = 0; j < CT3_PF; ++j) {
                const int gy = y0 - HALO + iy, gx = x0 - HALO + ix;
                const bool ok = (unsigned)gy < (unsigned)A.hin && (unsigned)gx < (unsigned)A.win;
                if (iy < A.IH) *reinterpret_cast<ct_u32x4*>(&tile[(e0 + j * SPX) * PS + cth]) = ok ? pf[j] : ct_u32x4{0u, 0u, 0u, 0u};
                ix += rx; iy += qy;
                if (ix >= A.IW) { ix -= A.IW; ++iy; }
            }
#pragma unroll
            for (int u = 0; u < (WQ + NT - 1) / NT; ++u) {
                const int i = tid + u * NT, tc = i / KQ, tap = tc / 3, c = tc - tap * 3;
                if (i < WQ) *reinterpret_cast<ct_u32x4*>(&wsl[(tap * 4 + c) * KS + (i % KQ) * 4]) = wpf[u];
            }
        };

        ct_f2 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = ct_f2{0.f, 0.f};

        issue(0);
        for (int k0 = 0; k0 < A.CI; k0 += KS) {
            __syncthreads();                                   // previous slice consumed
            land();
            __syncthreads();
            if (k0 + KS < A.CI) issue(k0 + KS);
            const float* col = tile + ((rg * P) * A.IW + xl) * PS;     // the lane's column, first input row of its row group
            // (k4 and the filter column stay rolled: fully unrolled, the scheduler hoists all 150 weight vectors of the slice into registers)
#pragma unroll 1
            for (int k4 = 0; k4 < KS / 4; ++k4) {
                if constexpr (S == 1) {
                    // out(y, x) = sum in(y + 2 - ky, x + 2 - kx) w[ky][kx]: tile-local input row (r + 4 - ky), column (xl + 4 - kx)
#pragma unroll 1
                    for (int kx = 0; kx < 5; ++kx) {
                        float4 in[NR];
#pragma unroll
                        for (int r = 0; r < NR; ++r) in[r] = *reinterpret_cast<const float4*>(col + (r * A.IW + (4 - kx)) * PS + k4 * 4);
#pragma unroll
                        for (int ky = 0; ky < 5; ++ky) {
                            float4 w3[3];
#pragma unroll
                            for (int c = 0; c < 3; ++c) w3[c] = *reinterpret_cast<const float4*>(&wsl[((ky * 5 + kx) * 4 + c) * KS + k4 * 4]);
#pragma unroll
                            for (int r = 0; r < P; ++r) {
                                const float4 v = in[r + 4 - ky];
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    ct_f2 a = acc[r * 3 + c];
                                    a = __builtin_elementwise_fma(ct_f2{v.x, v.y}, ct_f2{w3[c].x, w3[c].y}, a);
                                    a = __builtin_elementwise_fma(ct_f2{v.z, v.w}, ct_f2{w3[c].z, w3[c].w}, a);
                                    acc[r * 3 + c] = a;
                                }
                            }
                        }
                    }
                } else {
                    // out(2i + py, 2j + px) = sum_{sy, sx} in(i + oy - sy, j + ox - sx) w[pary + 2 sy][parx + 2 sx],
                    //   par = (p + 1) & 1, o = (p + 1 - par) / 2: py 0 -> ky {1, 3} rows {i, i - 1};  py 1 -> ky {0, 2, 4} rows {i + 1, i, i - 1}
                    // tile-local input row (r + 1 + oy - sy), column (xl + 1 + ox - sx)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int parx = (px + 1) & 1, ntx = px ? 3 : 2, ox = px ? 1 : 0;
#pragma unroll 1
                        for (int sx = 0; sx < ntx; ++sx) {
                            float4 in[NR];
#pragma unroll
                            for (int r = 0; r < NR; ++r) in[r] = *reinterpret_cast<const float4*>(col + (r * A.IW + (1 + ox - sx)) * PS + k4 * 4);
#pragma unroll
                            for (int py = 0; py < 2; ++py) {
                                const int pary = (py + 1) & 1, nty = py ? 3 : 2, oy = py ? 1 : 0;
#pragma unroll
                                for (int sy = 0; sy < 3; ++sy) {
                                    if (sy >= nty) continue;
                                    const int tap = (pary + 2 * sy) * 5 + parx + 2 * sx;
                                    float4 w3[3];
#pragma unroll
                                    for (int c = 0; c < 3; ++c) w3[c] = *reinterpret_cast<const float4*>(&wsl[(tap * 4 + c) * KS + k4 * 4]);
#pragma unroll
                                    for (int r = 0; r < P; ++r) {
                                        const float4 v = in[r + 1 + oy - sy];
#pragma unroll
                                        for (int c = 0; c < 3; ++c) {
                                            ct_f2 a = acc[((py * 2 + px) * P + r) * 3 + c];
                                            a = __builtin_elementwise_fma(ct_f2{v.x, v.y}, ct_f2{w3[c].x, w3[c].y}, a);
                                            a = __builtin_elementwise_fma(ct_f2{v.z, v.w}, ct_f2{w3[c].z, w3[c].w}, a);
                                            acc[((py * 2 + px) * P + r) * 3 + c] = a;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        // ---- store (+ bias).  Lanes of a row group are consecutive columns: 12 (stride 2: 24) contiguous bytes per lane
        const int gx = x0 + xl;
        if (gx < A.win) {
            if constexpr (S == 1) {
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const int gy = y0 + rg * P + r;
                    if (gy >= A.hin) continue;
                    float* o = A.out + (((int64_t)img * A.hin + gy) * A.win + gx) * 3;
                    o[0] = acc[r * 3 + 0].x + acc[r * 3 + 0].y + bias0; o[1] = acc[r * 3 + 1].x + acc[r * 3 + 1].y + bias1; o[2] = acc[r * 3 + 2].x + acc[r * 3 + 2].y + bias2;
                }
            } else {
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const int gi = y0 + rg * P + r;
                    if (gi >= A.hin) continue;
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        float* o = A.out + (((int64_t)img * 2 * A.hin + 2 * gi + py) * (2 * A.win) + 2 * gx) * 3;
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            o[px * 3 + 0] = acc[((py * 2 + px) * P + r) * 3 + 0].x + acc[((py * 2 + px) * P + r) * 3 + 0].y + bias0;
                            o[px * 3 + 1] = acc[((py * 2 + px) * P + r) * 3 + 1].x + acc[((py * 2 + px) * P + r) * 3 + 1].y + bias1;
                            o[px * 3 + 2] = acc[((py * 2 + px) * P + r) * 3 + 2].x + acc[((py * 2 + px) * P + r) * 3 + 2].y + bias2;
                        }
                    }
                }
            }
        }
    }
}

template <int S, int P, int NT>
void launch_ct3(hipStream_t s, Ct3 A, int TW) {
    constexpr int HALO = S == 1 ? 2 : 1;
    constexpr int KS = ct3_ks(S), PS = KS + 4, KQ = KS / 4, CT3_PF = ct3_pf(S);
    A.TW = TW;
    A.TR = NT / TW * P;
    A.IH = A.TR + 2 * HALO;
    A.IW = A.TW + 2 * HALO;
    A.tiles_y = (A.hin + A.TR - 1) / A.TR;
    A.tiles_x = (A.win + A.TW - 1) / A.TW;
    A.ntiles = A.nimg * A.tiles_y * A.tiles_x;
    const size_t lds = (size_t)((A.IH * A.IW * PS + 3) & ~3) * 4 + 25 * 4 * KS * 4;
    if (A.IH * A.IW * KQ > NT * CT3_PF || lds > (size_t)dev_info().lds_per_cu) { set_launch_error("convt3: input tile %d x %d does not fit the prefetch slots / LDS", A.IH, A.IW); return; }
    ensure_dyn_lds((const void*)convt3_kernel<S, P, NT>, lds);
    int per_cu = (int)(dev_info().lds_per_cu / (lds + 1024));
    per_cu = per_cu < 1 ? 1 : per_cu > 2048 / NT ? 2048 / NT : per_cu;
    int grid = 256 * per_cu;
    if (grid > A.ntiles) grid = A.ntiles;
    { const int rounds = (A.ntiles + grid - 1) / grid; grid = (A.ntiles + rounds - 1) / rounds; }
    hipLaunchKernelGGL((convt3_kernel<S, P, NT>), dim3((unsigned)grid), dim3(NT), lds, s, A);
}

}  // namespace

bool convt3_direct_ok(int c1, int c2, int hin, int win, int stride) {
    if ((stride != 1 && stride != 2) || c1 % ct3_ks(stride) || c2 % ct3_ks(stride)) return false;
    // prefetch slots: the chosen tile's IH * IW * 2 float4s must fit threads x CT3_PF (checked against the tiles used below)
    return win >= 16 && hin >= 4;
}

// out[nimg][S*hin][S*win][3] = conv2d_transpose(concat(x1, x2[img % nmod2]), w[5][5][3][c1 + c2], stride S, SAME) + bias
void convt3_direct(hipStream_t s, const float* x1, int c1, const float* x2, int c2, int nmod2, int nimg, int hin, int win, int stride,
                   const float* w, const float* bias, float* out) {
    if (convt3_mfma_ok(c1, c2, hin, win, stride, nimg)) { convt3_mfma(s, x1, c1, x2, nmod2, nimg, hin, win, stride, w, bias, out); return; }
    Ct3 A{};
    A.x1 = x1; A.c1 = c1; A.x2 = x2; A.c2 = c2; A.nmod2 = nmod2; A.CI = c1 + c2; A.hin = hin; A.win = win; A.nimg = nimg;
    A.w = w; A.bias = bias; A.out = out;
    const int TW = win >= 64 ? 64 : win >= 32 ? 32 : 16;
    if (stride == 1) {
        launch_ct3<1, 3, 384>(s, A, TW);      // rows per tile = 384 / TW * 3: 18 at TW 64 (36 x 64 frames: two tiles), 36 at TW 32
    } else {
        // tiles of 256 / TW * 2 small-grid rows: 8 at TW 64, 16 at TW 32 (a 32 x 32 grid in two tiles), 32 at TW 16.  A STARVED launch -- the
        // reward hook's batch of 25: 50 tiles for 256 CUs, each walking all 16 channel slices alone, 63 us of a 0.8 ms translate call --
        // takes quarter-height tiles of two waves instead (200 tiles)
        const int tr_big = 256 / TW * 2;
        const int64_t tiles_big = (int64_t)nimg * ((hin + tr_big - 1) / tr_big) * ((win + TW - 1) / TW);
        if (tiles_big * 2 <= dev_info().cus && hin >= 128 / TW) launch_ct3<2, 1, 128>(s, A, TW);
        else launch_ct3<2, 2, 256>(s, A, TW);
    }
}

}  // namespace ctx
