// dconv.hip -- launchers of the narrow-channel direct convolutions (dconv.h): tap tables, tile choice, LDS budget, dispatch.
#include <cstdio>
#include <cstdlib>

#include "dconv.h"
#include "launch.h"

namespace ctx {

namespace {
constexpr int LDS_BUDGET = 150 * 1024;      // of the 160 KiB per CU: one block of 8 waves

int cik_of(int CI) { return CI == 3 ? 4 : CI <= 8 ? 8 : CI <= 16 ? 16 : CI <= 32 ? 32 : 64; }

template <int CIK, int MI>
void launch_fwd_nb(hipStream_t s, const DcFwd& P, int NB, dim3 grid, size_t lds, int ntiles, int nslots) {
#define DC_CASE(nb)                                                                                                          \
    case nb: {                                                                                                               \
        static bool raised = false;                                                                                          \
        if (!raised) { (void)hipFuncSetAttribute((const void*)dconv_fwd_kernel<CIK, MI, nb>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET + 8192); raised = true; } \
        hipLaunchKernelGGL((dconv_fwd_kernel<CIK, MI, nb>), grid, dim3(DC_THREADS), lds, s, P, ntiles, nslots);                              \
        break;                                                                                                               \
    }
    if constexpr (MI <= 2) {
        switch (NB) { DC_CASE(1) DC_CASE(2) DC_CASE(4) DC_CASE(8) default: break; }
    } else {
        switch (NB) { DC_CASE(1) DC_CASE(2) DC_CASE(4) default: break; }      // MI 3 / 4 with 8 column blocks would spill; never chosen
    }
#undef DC_CASE
}

template <int CIK>
void launch_fwd_mi(hipStream_t s, const DcFwd& P, int MI, int NB, dim3 grid, size_t lds, int ntiles, int nslots) {
    if (MI == 4) launch_fwd_nb<CIK, 4>(s, P, NB, grid, lds, ntiles, nslots);
    else if (MI == 3) launch_fwd_nb<CIK, 3>(s, P, NB, grid, lds, ntiles, nslots);
    else if (MI == 2) launch_fwd_nb<CIK, 2>(s, P, NB, grid, lds, ntiles, nslots);
    else launch_fwd_nb<CIK, 1>(s, P, NB, grid, lds, ntiles, nslots);
}
}  // namespace

// measurement hook (tools/dconv_bench.hip): force the tile of the next launches; 0 = automatic
int g_dc_force[3] = {0, 0, 0};       // TH, TW, MI
void dconv_force_tile(int th, int tw, int mi) { g_dc_force[0] = th; g_dc_force[1] = tw; g_dc_force[2] = mi; }
int g_dc_last[4] = {0, 0, 0, 0};     // TH, TW, MI, GT of the last forward launch

bool dconv_ok(int CI, int N) { return (CI == 3 || CI == 8 || CI == 16 || CI == 32 || CI == 64) && N >= 1 && N <= 128; }

// Fills tiles / LDS split and launches.  `span` = extent of the tap offsets (5 for the 5x5 taps over the input, 3 for a
// stride-2 transposed conv over its small grid).
static void dconv_launch(hipStream_t s, DcFwd P, int span) {
    const int CIK = cik_of(P.CI), CIP = dc_cip(CIK);
    int NB = (P.N + 15) / 16;
    NB = NB <= 1 ? 1 : NB <= 2 ? 2 : NB <= 4 ? 4 : 8;
    const int NP = NB * 16, TPC = CIK >= 16 ? 1 : 16 / CIK;
    // the filter in the LDS image's order: classes padded to whole 16-k chunks
    int nslots = 0, maxt = 0;
    for (int c = 0; c < P.ncls; ++c) {
        P.cls[c].pslot0 = nslots;
        const int ntp = (P.cls[c].ntaps + TPC - 1) / TPC * TPC;
        nslots += ntp;
        maxt = ntp > maxt ? ntp : maxt;
    }
    const size_t wall = (size_t)nslots * CIK * NP * 4 + (size_t)nslots * 4 + 64;
    const size_t wmin = (size_t)4 * TPC * CIK * NP * 4 + 256;     // staged mode: at least four chunks of taps at a time
    // tile: TW in {16, 32, 64} (<= the logical row), TH rows; MI = row blocks per wave.  Cost model: the matrix pipes are
    // per SIMD and wave w sits on SIMD w % 4, so a tile takes the time of its busiest SIMD; scored as useful row blocks per
    // (4 x busiest SIMD), times the share of real rows in ragged last tiles, times the halo overhead of the input tile,
    // with a penalty when the filter cannot stay resident beside the tile (it is then re-staged per tile, with barriers).
    int best_th = 1, best_mi = 1, best_tw = 16, best_res = 1;
    double best = -1;
    const int mi_max = NB >= 8 ? 2 : 4;                       // accumulators: MI * NB * 4 registers
    for (int tw = 16; tw <= 64 && tw <= (P.wlog + 15) / 16 * 16; tw *= 2)
        for (int mi = 1; mi <= mi_max; ++mi)
            for (int th = 1; th <= 16; ++th) {
                const int nrb = th * (tw / 16);
                if (nrb > DC_NW * mi) break;
                const int ih = P.S * (th - 1) + span, iw = P.S * (tw - 1) + span;
                const size_t tile = (size_t)((ih * iw * CIP + 3) & ~3) * 4;
                const bool res = tile + wall <= (size_t)LDS_BUDGET;
                if (!res && tile + wmin > (size_t)LDS_BUDGET) break;
                int load[4] = {0, 0, 0, 0};
                for (int rb = 0; rb < nrb; ++rb) load[rb & 3] += 1;                   // row block rb runs on wave rb % 8, i.e. SIMD rb % 4
                int busiest = 1;
                for (int q = 0; q < 4; ++q) busiest = load[q] > busiest ? load[q] : busiest;
                const int tiles_y = (P.hlog + th - 1) / th, tiles_x = (P.wlog + tw - 1) / tw;
                const double eff = (double)nrb / (4.0 * busiest) * ((double)P.hlog * P.wlog / ((double)tiles_y * th * tiles_x * tw)) *
                                   ((double)(th * tw * P.S * P.S) / (ih * iw)) * (mi >= 2 ? 1.0 : 0.9) * (res ? 1.0 : 0.7) *
                                   (nrb >= 8 ? 1.0 : 0.8) *                         // few active waves hide little latency
                                   (2 * (tile + (res ? wall : wmin)) <= (size_t)LDS_BUDGET ? 1.0 : 0.85);  // two blocks per CU: one's loads under the other's MFMAs
                if (eff > best) { best = eff; best_th = th; best_mi = mi; best_tw = tw; best_res = res; }
            }
    if (g_dc_force[0]) {
        best_th = g_dc_force[0]; best_tw = g_dc_force[1]; best_mi = g_dc_force[2];
        const int ih = P.S * (best_th - 1) + span, iw = P.S * (best_tw - 1) + span;
        best_res = (size_t)((ih * iw * CIP + 3) & ~3) * 4 + wall <= (size_t)LDS_BUDGET;
    }
    P.TW = best_tw;
    P.TH = best_th;
    const int MI = best_mi;
    P.IH = P.S * (P.TH - 1) + span;
    P.IW = P.S * (P.TW - 1) + span;
    P.tiles_y = (P.hlog + P.TH - 1) / P.TH;
    P.tiles_x = (P.wlog + P.TW - 1) / P.TW;
    const size_t tile = (size_t)((P.IH * P.IW * CIP + 3) & ~3) * 4;
    P.wres = best_res;
    int gt = nslots;
    if (!P.wres) {
        gt = (int)(((size_t)LDS_BUDGET - tile - 256) / ((size_t)CIK * NP * 4)) / TPC * TPC;
        if (gt > maxt) gt = maxt;
        if (gt < TPC) gt = TPC;
    }
    P.GT = gt;
    g_dc_last[0] = P.TH; g_dc_last[1] = P.TW; g_dc_last[2] = MI; g_dc_last[3] = P.wres ? -nslots : gt;
    const size_t lds = tile + (size_t)(P.wres ? nslots : gt) * CIK * NP * 4 + (size_t)(nslots + 4) * 4;
    const int ntiles = P.nimg * P.tiles_y * P.tiles_x;
    const dim3 grid((unsigned)ntiles);
    {
        const int total = nslots * CIK * NP;
        const dim3 pg((unsigned)((total + 255) / 256 > 64 ? 64 : (total + 255) / 256));
        switch (CIK) {
            case 4: hipLaunchKernelGGL((dconv_pack_kernel<4>), pg, dim3(256), 0, s, P, NP, nslots); break;
            case 8: hipLaunchKernelGGL((dconv_pack_kernel<8>), pg, dim3(256), 0, s, P, NP, nslots); break;
            case 16: hipLaunchKernelGGL((dconv_pack_kernel<16>), pg, dim3(256), 0, s, P, NP, nslots); break;
            case 32: hipLaunchKernelGGL((dconv_pack_kernel<32>), pg, dim3(256), 0, s, P, NP, nslots); break;
            default: hipLaunchKernelGGL((dconv_pack_kernel<64>), pg, dim3(256), 0, s, P, NP, nslots); break;
        }
    }
    switch (CIK) {
        case 4: launch_fwd_mi<4>(s, P, MI, NB, grid, lds, ntiles, nslots); break;
        case 8: launch_fwd_mi<8>(s, P, MI, NB, grid, lds, ntiles, nslots); break;
        case 16: launch_fwd_mi<16>(s, P, MI, NB, grid, lds, ntiles, nslots); break;
        case 32: launch_fwd_mi<32>(s, P, MI, NB, grid, lds, ntiles, nslots); break;
        default: launch_fwd_mi<64>(s, P, MI, NB, grid, lds, ntiles, nslots); break;
    }
}

// conv2d 5x5, stride s, TF SAME (pad_before = pad): out [nimg, hin/s, win/s, N]; w[tap][k][n] (wmode 0) or [tap][n][k] (1)
void dconv_conv(hipStream_t s, DcFwd P, int stride, int pad) {
    P.S = stride; P.y_org = -pad; P.x_org = -pad;
    P.hlog = P.hout = P.hin / stride; P.wlog = P.wout = P.win / stride; P.osc = 1;
    P.ncls = 1; P.cls[0] = DcClass{0, 25, 0, 0, 0, 5, 52, 0, 0, 1};
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx) P.taps[ky * 5 + kx] = DcTap{(int16_t)ky, (int16_t)kx, (int16_t)(ky * 5 + kx), 0};
    dconv_launch(s, P, 5);
}

// conv2d_transpose 5x5 stride 1 (SAME pad 2): out[q] = sum_taps in[q + 2 - tap] * w[tap]  -- a correlation with the mirrored offsets
void dconv_convt1(hipStream_t s, DcFwd P) {
    P.S = 1; P.y_org = -2; P.x_org = -2;
    P.hlog = P.hout = P.hin; P.wlog = P.wout = P.win; P.osc = 1;
    P.ncls = 1; P.cls[0] = DcClass{0, 25, 0, 0, 0, 5, 52, 4, 4, -1};
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx) P.taps[ky * 5 + kx] = DcTap{(int16_t)(4 - ky), (int16_t)(4 - kx), (int16_t)(ky * 5 + kx), 0};
    dconv_launch(s, P, 5);
}

// conv2d_transpose 5x5 stride 2 (the input gradient of the SAME stride-2 conv, pad_before 1): output pixel (2i'+py, 2j'+px)
// takes taps ky = par + 2 sy (par = (py + 1) & 1) from input row i' + off - sy (off = (py + 1 - par) / 2): KmConvTGather's classes
void dconv_convt2(hipStream_t s, DcFwd P) {
    P.S = 1; P.y_org = -1; P.x_org = -1;
    P.hlog = P.hin; P.wlog = P.win; P.hout = 2 * P.hin; P.wout = 2 * P.win; P.osc = 2;
    P.ncls = 4;
    int nt = 0;
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const int pary = (py + 1) & 1, parx = (px + 1) & 1, nty = (5 - pary + 1) / 2, ntx = (5 - parx + 1) / 2;
        const int oy = (py + 1 - pary) / 2, ox = (px + 1 - parx) / 2;
        P.cls[c] = DcClass{nt, nty * ntx, py, px, 0, ntx, ntx == 2 ? 128 : 86, oy + 1, ox + 1, -1};
        for (int sy = 0; sy < nty; ++sy)
            for (int sx = 0; sx < ntx; ++sx)
                P.taps[nt++] = DcTap{(int16_t)(oy - sy + 1), (int16_t)(ox - sx + 1), (int16_t)((pary + 2 * sy) * 5 + parx + 2 * sx), 0};
    }
    dconv_launch(s, P, 3);
}

// ---- filter gradient ------------------------------------------------------------------------------------------
// out[m][n] = sum over slabs in fixed order
__global__ __launch_bounds__(256) void dconv_wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int M, int NP, int CB, float* __restrict__ out) {
    const int64_t total = (int64_t)M * CB;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / CB), n = (int)(i - (int64_t)m * CB);
        const float* p = slab + (int64_t)m * NP + n;
        float v = 0.f;
        for (int s = 0; s < nslab; ++s) v += p[(int64_t)s * M * NP];
        out[i] = v;
    }
}

namespace {
template <int CAK, int RBW>
void launch_wg_nb(hipStream_t s, const DcWgrad& P, int NB, dim3 grid, size_t lds) {
#define DC_CASE(nb)                                                                                                          \
    case nb: {                                                                                                               \
        static bool raised = false;                                                                                          \
        if (!raised) { (void)hipFuncSetAttribute((const void*)dconv_wgrad_kernel<CAK, RBW, nb>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET + 8192); raised = true; } \
        hipLaunchKernelGGL((dconv_wgrad_kernel<CAK, RBW, nb>), grid, dim3(DC_THREADS), lds, s, P);                           \
        break;                                                                                                               \
    }
    switch (NB) { DC_CASE(1) DC_CASE(2) DC_CASE(4) DC_CASE(8) default: break; }
#undef DC_CASE
}
}  // namespace

// dw[tap][a][b] (a: channels of `big`, b: channels of [s1 | s2]); slab: scratch of slab_floats floats
void dconv_wgrad(hipStream_t s, DcWgrad P, float* slab, int64_t slab_floats) {
    const int CAK = P.CA == 3 ? 4 : P.CA <= 8 ? 8 : P.CA <= 16 ? 16 : 32;
    const int CAP = CAK == 4 ? 4 : CAK + 4;
    int NB = (P.CB + 15) / 16;
    NB = NB <= 1 ? 1 : NB <= 2 ? 2 : NB <= 4 ? 4 : 8;
    const int NP = NB * 16, CBP = NP + 4;
    P.M = 25 * P.CA;
    const int nrb = (P.M + 15) / 16;
    const int RBW = (nrb + DC_NW - 1) / DC_NW;            // 1 (CA 3), 2 (8), 4 (16), 7 (32)
    P.RBW = RBW;
    P.TW = P.ws >= 64 ? 64 : P.ws >= 32 ? 32 : 16;
    P.tw_sh = P.TW == 64 ? 6 : P.TW == 32 ? 5 : 4;
    // TH: the divisor-like height that keeps big halo + small tile inside the budget with the least halo overhead
    int best_th = 1;
    double best = -1;
    for (int th = 1; th <= 32; ++th) {
        const int ih = P.S * (th - 1) + 5, iw = P.S * (P.TW - 1) + 5;
        const size_t lds = (size_t)((ih * iw * CAP + 3) & ~3) * 4 + (size_t)th * P.TW * CBP * 4;
        if (lds > (size_t)LDS_BUDGET) break;
        const int tiles = (P.hs + th - 1) / th;
        const double eff = (double)P.hs / (tiles * th) * ((double)th / (th + 4.0 / P.S));
        if (eff > best) { best = eff; best_th = th; }
    }
    P.TH = best_th;
    P.IH = P.S * (P.TH - 1) + 5;
    P.IW = P.S * (P.TW - 1) + 5;
    P.tiles_y = (P.hs + P.TH - 1) / P.TH;
    P.tiles_x = (P.ws + P.TW - 1) / P.TW;
    P.ntiles = P.nimg * P.tiles_y * P.tiles_x;
    const size_t lds = (size_t)((P.IH * P.IW * CAP + 3) & ~3) * 4 + (size_t)P.TH * P.TW * CBP * 4;
    // persistent blocks, as many per CU as LDS admits (up to 3): one block's tile loads run under another's MFMA loop
    int occ = (int)((size_t)LDS_BUDGET / lds);
    occ = occ < 1 ? 1 : occ > 3 ? 3 : occ;
    int64_t nblk = 256 * occ;
    const int64_t cap = slab_floats / ((int64_t)P.M * NP);
    if (nblk > cap) nblk = cap;
    if (nblk > P.ntiles) nblk = P.ntiles;
    if (nblk < 1) nblk = 1;
    P.slab = slab;
    const dim3 grid((unsigned)nblk);
#define DC_WG(cak, rbw) launch_wg_nb<cak, rbw>(s, P, NB, grid, lds)
    if (CAK == 4) DC_WG(4, 1);
    else if (CAK == 8) DC_WG(8, 2);
    else if (CAK == 16) DC_WG(16, 4);
    else DC_WG(32, 7);
#undef DC_WG
    const int64_t total = (int64_t)P.M * P.CB;
    int64_t rb = (total + 255) / 256;
    if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, (const float*)slab, (int)nblk, P.M, NP, P.CB, P.out);
}

}  // namespace ctx
