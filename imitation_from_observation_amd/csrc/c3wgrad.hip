// c3wgrad.hip -- filter gradient of a 5x5 conv (stride 1 | 2, TF SAME) whose BIG-grid side has THREE channels: h0_conv's filter
// gradient (frames x d act; arm_shaping.py:1283, :1291, :1633) and d_h4's (d loss / d out x [decoder | ctx skip]; :1329, :1669).
// Round 3; replaces dconv_wgrad_kernel<4, ...> for the shapes c3wgrad_ok accepts.
//
//   dw[tap][a][n] = sum over images and small-grid pixels p of  big[S p + tap - pad][a] * small[p][n]
//
// GEMM rows m = 3 tap + a (75 of 80 used), columns n, K = pixels.  The generic narrow-channel kernel spends 4.8 vector-ALU
// instructions per MFMA on run-time tile geometry and keeps the matrix pipe 44 % busy (PMC, profiles/archive/round3_b_pmc_*): 0.20 ms for
// ContextSkipNew's d_h4 against ~0.08 ms of either matrix or HBM time.  Here everything about a tile is a compile-time constant:
//   * a tile is 128 small-grid pixels (4 rows x 32 or 8 x 16); wave w owns the pixels of row(s) w and ALL 16 NB columns, so its
//     K loop is 32 fully unrolled steps of four x-adjacent pixels: 5 + NB ds_read_b32 at literal offsets from two per-lane base
//     addresses, 5 NB MFMAs, no address arithmetic at all;
//   * small-tile pixel stride NP + 16 floats: the two pixel groups a ds_read_b32 serves together sit 16 banks apart;
//   * 4-wave blocks, two per CU, persistent over their tiles (accumulators stay in registers), next tile's global loads in
//     flight under the current tile's MFMAs; columns beyond 64 are a second set of blocks in the same launch;
//   * the bias gradient (column sums of the small tensor) is accumulated from the registers that carry the tile to LDS;
//   * one partial [80][NP] per block, waves combined in a fixed order; dconv_wgrad_reduce adds the blocks in a fixed order.
// v_mfma_f32_16x16x4_f32, exact f32, deterministic.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "launch.h"

namespace ctx {

namespace {

constexpr int W3_THREADS = 256;
constexpr int W3_PF = 10;                        // big tile: prefetch dwords per thread (IH * IW * 3 <= 256 * 10)
constexpr int W3_RB = 5;                         // row blocks of 16: 80 >= 75

struct W3P {
    const float* big;                            // [nimg, hb, wb, 3]
    const float* s1; int ld1; int c1;            // small-grid tensor, channels [0, c1)
    const float* s2; int ld2; int nmod2;         // channels [c1, CB): image img % nmod2
    int nimg, hb, wb, hs, ws;
    int tiles_y, tiles_x, ntiles;
    int nbl;                                     // blocks per column group (grid = nbl * ngroups)
    float* slab;                                 // [group][nbl][80][NP]
    float* dbslab;                               // [group][nbl][NP] or nullptr
};

// S: stride; NB: columns / 16 per block; TWB: tile width / 16 (TH = 8 / TWB rows, 128 pixels)
template <int S, int NB, int TWB>
__global__ __launch_bounds__(W3_THREADS, 2) void c3wgrad_kernel(const W3P P) {
    constexpr int NP = 16 * NB, CBP = NP + 16, TW = 16 * TWB, TH = 8 / TWB, PAD = S == 2 ? 1 : 2;
    constexpr int IH = S * (TH - 1) + 5, IW = S * (TW - 1) + 5, TFL = IH * IW * 3;
    constexpr int S4 = NP / 4, PFS = 128 * S4 / W3_THREADS;                  // small tile: float4s per pixel, per thread
    constexpr int RPW = TH / 4;                                              // tile rows per wave (1 | 2)
    static_assert(TFL <= W3_THREADS * W3_PF, "prefetch slots");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bigt = smem;                                       // [IH * IW][4]
    float* smallt = smem + ((IH * IW * 4 + 3) & ~3);          // [128][CBP]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int grp = blockIdx.x / P.nbl, blk = blockIdx.x - grp * P.nbl;
    const int n0 = grp * NP;                                                 // this block's columns of [s1 | s2]

    for (int i = tid; i < IH * IW; i += W3_THREADS) bigt[i * 4 + 3] = 0.f;   // the fourth channel: written once, never read as data

    // ---- big-tile prefetch (as c3conv.hip): float e = tid + 256 j of the tile's IH rows of IW * 3 contiguous floats
    unsigned pf[W3_PF];
    auto tile_org = [&](int t, int& img, int& y0, int& x0) {
        const int txi = t % P.tiles_x; t /= P.tiles_x;
        const int tyi = t % P.tiles_y;
        img = t / P.tiles_y; y0 = tyi * TH; x0 = txi * TW;
    };
    constexpr int RD = IW * 3, RQ = W3_THREADS / RD, RR = W3_THREADS % RD;
    const int r0 = tid / RD, c0 = tid - r0 * RD;
    const unsigned frame_bytes = (unsigned)(P.hb * P.wb * 3 * 4), wb3 = (unsigned)(P.wb * 3);
    // ---- small-tile prefetch: float4 i = tid + 256 j = (pixel tid / S4 + j * 256 / S4, column group tid % S4).  The thread's column
    // group picks its source tensor once
    const int cg = tid % S4, sp0 = tid / S4;
    const int col = n0 + 4 * cg;
    const bool second = P.s2 && col >= P.c1;
    const float* const sbase = second ? P.s2 + (col - P.c1) : P.s1 + col;
    const int sld = second ? P.ld2 : P.ld1;
    float4 ps[PFS];
    auto issue = [&](int t) {
        int img, y0, x0;
        tile_org(t, img, y0, x0);
        {   // num_records = one frame: rows above / below it read as zeros by themselves
            const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.big + (int64_t)img * P.hb * P.wb * 3), 0, (int)frame_bytes, 0x00020000);
            const int ix0 = (S * x0 - PAD) * 3;
            unsigned roff = (unsigned)((((S * y0 - PAD) + r0) * (int)wb3 + ix0) * 4);
            int c = c0;
#pragma unroll
            for (int j = 0; j < W3_PF; ++j) {
                const unsigned gx3 = (unsigned)(ix0 + c);
                pf[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, gx3 < wb3 ? roff + 4u * (unsigned)c : OOB, 0, 0);
                c += RR; roff += (unsigned)RQ * wb3 * 4u;
                const bool wrap = c >= RD;
                c = wrap ? c - RD : c; roff = wrap ? roff + wb3 * 4u : roff;
            }
        }
        const float* p = sbase + (int64_t)((second ? img % P.nmod2 : img) * P.hs + y0) * P.ws * sld + (int64_t)x0 * sld;
#pragma unroll
        for (int j = 0; j < PFS; ++j) {
            const int pi = sp0 + j * (W3_THREADS / S4), ty = pi / TW, tx = pi - ty * TW;      // (compile-time steps: 256 / S4 pixels per slot)
            ps[j] = ldg4(p + (int64_t)(ty * P.ws + tx) * sld);
        }
    };
    float4 dbacc = zero4();
    auto land = [&]() {
#pragma unroll
        for (int j = 0; j < W3_PF; ++j) {
            const int e = tid + W3_THREADS * j, r = e / RD, c = e - r * RD, px = c / 3, ch = c - px * 3;
            if (e < TFL) bigt[(r * IW + px) * 4 + ch] = __uint_as_float(pf[j]);
        }
#pragma unroll
        for (int j = 0; j < PFS; ++j) {
            const int pi = sp0 + j * (W3_THREADS / S4);
            *reinterpret_cast<float4*>(&smallt[pi * CBP + 4 * cg]) = ps[j];
            dbacc.x += ps[j].x; dbacc.y += ps[j].y; dbacc.z += ps[j].z; dbacc.w += ps[j].w;
        }
    };

    // ---- per-lane base addresses.  A: row m = 16 rb + l15 = 3 tap + a -> tap (ky, kx), channel a; pixel x + kg.  B: pixel kg, column l15
    int abase[W3_RB];
#pragma unroll
    for (int rb = 0; rb < W3_RB; ++rb) {
        int m = 16 * rb + l15;
        m = m < 75 ? m : 74;                                                 // rows 75..79: some valid address (never stored)
        const int tap = m / 3, a = m - 3 * tap, ky = tap / 5, kx = tap - 5 * ky;
        abase[rb] = (((S * RPW * wv + ky) * IW + kx + S * kg) * 4 + a) * 4;  // bytes; wave wv starts at tile row RPW * wv
    }
    const int bbase = ((RPW * wv * TW + kg) * CBP + l15) * 4;

    f32x4 acc[W3_RB][NB];
#pragma unroll
    for (int rb = 0; rb < W3_RB; ++rb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    int t = blk;
    if (t < P.ntiles) issue(t);
    for (; t < P.ntiles; t += P.nbl) {
        __syncthreads();                                       // the previous tile's fragments are consumed
        land();
        __syncthreads();
        if (t + P.nbl < P.ntiles) issue(t + P.nbl);
        // K loop: RPW rows x TW / 4 steps of four x-adjacent pixels, every offset a literal
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int cx = 0; cx < TW / 4; ++cx) {
                float av[W3_RB], bv[NB];
#pragma unroll
                for (int rb = 0; rb < W3_RB; ++rb)
                    av[rb] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bigt) + abase[rb] + (S * rr * IW + S * 4 * cx) * 16);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    bv[nb] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(smallt) + bbase + ((rr * TW + 4 * cx) * CBP + 16 * nb) * 4);
#pragma unroll
                for (int rb = 0; rb < W3_RB; ++rb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb], bv[nb], acc[rb][nb], 0, 0, 0);
            }
    }

    // ---- bias gradient partial of the block: 256 / S4 threads share a column group -> summed in a fixed order
    if (P.dbslab) {
        __syncthreads();                                       // tiles dead
        *reinterpret_cast<float4*>(&smem[tid * 4]) = dbacc;    // [pixel phase][column group][4]
        __syncthreads();
        if (tid < S4) {
            float4 v = zero4();
            for (int sg = 0; sg < W3_THREADS / S4; ++sg) {
                const float4 q = *reinterpret_cast<const float4*>(&smem[(sg * S4 + tid) * 4]);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            *reinterpret_cast<float4*>(P.dbslab + (int64_t)blockIdx.x * NP + 4 * tid) = v;
        }
    }
    // ---- the four waves' partial sums -> one, in a fixed tree through LDS: (0 + 2), (1 + 3), then (0 + 1)
    float* red = smem;                                         // [2][W3_RB][NB][4][64]
#pragma unroll
    for (int half = 2; half >= 1; half /= 2) {
        __syncthreads();                                       // tiles dead / previous round read
        if (wv >= half && wv < 2 * half) {
#pragma unroll
            for (int rb = 0; rb < W3_RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((((wv - half) * W3_RB + rb) * NB + nb) * 4 + r) * 64 + lane] = acc[rb][nb][r];
        }
        __syncthreads();
        if (wv < half) {
#pragma unroll
            for (int rb = 0; rb < W3_RB; ++rb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rb][nb][r] += red[(((wv * W3_RB + rb) * NB + nb) * 4 + r) * 64 + lane];
        }
    }
    // ---- partial -> slab[block][m][n]: a lane holds rows 16 rb + 4 kg + r of column 16 nb + l15
    if (wv == 0) {
        float* sl = P.slab + (int64_t)blockIdx.x * 75 * NP;
#pragma unroll
        for (int rb = 0; rb < W3_RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * rb + 4 * kg + r;
                if (m >= 75) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) sl[m * NP + 16 * nb + l15] = acc[rb][nb][r];
            }
    }
}

template <int S, int NB, int TWB>
void launch_w3(hipStream_t s, W3P P, int ngroups) {
    constexpr int NP = 16 * NB, CBP = NP + 16, TW = 16 * TWB, TH = 8 / TWB;
    constexpr int IH = S * (TH - 1) + 5, IW = S * (TW - 1) + 5;
    constexpr size_t tiles_b = (size_t)(((IH * IW * 4 + 3) & ~3) + 128 * CBP) * sizeof(float);
    constexpr size_t red_b = (size_t)2 * W3_RB * NB * 256 * sizeof(float);
    constexpr size_t lds = tiles_b > red_b ? tiles_b : red_b;
    ensure_dyn_lds((const void*)c3wgrad_kernel<S, NB, TWB>, lds);
    hipLaunchKernelGGL((c3wgrad_kernel<S, NB, TWB>), dim3((unsigned)(P.nbl * ngroups)), dim3(W3_THREADS), lds, s, P);
}

}  // namespace

bool c3wgrad_ok(const DcWgrad& P) {
    if (!(opt(OPT_DIRECT3) & 4) || P.CA != 3 || P.ldb != 3 || (P.S != 1 && P.S != 2) || P.pad != (P.S == 2 ? 1 : 2)) return false;
    if (P.hb != P.S * P.hs || P.wb != P.S * P.ws) return false;
    if (P.ws % 16 || P.hs % (P.ws % 32 == 0 ? 4 : 8)) return false;                      // whole 128-pixel tiles
    const int NP = P.CB >= 64 ? 64 : 32;
    if (P.CB % NP || P.CB > 256) return false;
    if (P.s2 && (P.c1 % 4 || P.ld2 % 4 || P.nmod2 < 1)) return false;
    if (P.ld1 % 4 || (!P.s2 && P.c1 < P.CB)) return false;
    return (int64_t)P.hb * P.wb * 3 * 4 < (1ll << 31);
}

// dw (and db when P.db) of DcWgrad P with CA = 3; slab: scratch of slab_floats floats
void c3wgrad(hipStream_t s, const DcWgrad& D, float* slab, int64_t slab_floats) {
    const int NP = D.CB >= 64 ? 64 : 32, ngroups = D.CB / NP;
    const bool tw32 = D.ws % 32 == 0;
    const int TW = tw32 ? 32 : 16, TH = tw32 ? 4 : 8;
    W3P P{D.big, D.s1, D.ld1, D.c1, D.s2, D.ld2, D.nmod2 > 0 ? D.nmod2 : 1, D.nimg, D.hb, D.wb, D.hs, D.ws, D.hs / TH, D.ws / TW, 0, 0, slab, nullptr};
    P.ntiles = P.nimg * P.tiles_y * P.tiles_x;
    // persistent blocks: two per CU over all column groups, whole rounds of tiles per block, inside the slab
    int64_t nbl = 2 * dev_info().cus / ngroups;
    const int64_t cap = slab_floats / ((int64_t)ngroups * 76 * NP);
    if (nbl > cap) nbl = cap;
    if (nbl > P.ntiles) nbl = P.ntiles;
    if (nbl < 1) { set_launch_error("c3wgrad: slab of %lld floats holds no partial", (long long)slab_floats); return; }
    { const int64_t rounds = (P.ntiles + nbl - 1) / nbl; nbl = (P.ntiles + rounds - 1) / rounds; }
    P.nbl = (int)nbl;
    P.dbslab = D.db ? slab + (int64_t)ngroups * nbl * 75 * NP : nullptr;
#define W3_GO(S_, NB_)  do { if (tw32) launch_w3<S_, NB_, 2>(s, P, ngroups); else launch_w3<S_, NB_, 1>(s, P, ngroups); } while (0)
    if (D.S == 2) { if (NP == 64) W3_GO(2, 4); else W3_GO(2, 2); }
    else { if (NP == 64) W3_GO(1, 4); else W3_GO(1, 2); }
#undef W3_GO
    for (int g = 0; g < ngroups; ++g) {
        dconv_wgrad_reduce(s, slab + (int64_t)g * nbl * 75 * NP, (int)nbl, 75, NP, NP, g * NP, D.CB, D.out);
        if (D.db) dconv_wgrad_reduce(s, P.dbslab + (int64_t)g * nbl * NP, (int)nbl, 1, NP, NP, g * NP, D.CB, D.db);
    }
}

}  // namespace ctx
