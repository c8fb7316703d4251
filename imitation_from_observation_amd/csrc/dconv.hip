// dconv.hip -- launchers of the narrow-channel direct convolutions (dconv.h): tap tables, tile choice, LDS budget, dispatch.
#include <cstdio>
#include <cstdlib>

#include "dconv.h"
#include "dconv2.h"
#include "launch.h"
#include "options.h"

namespace ctx {

namespace {
// per CU / one block of 8 waves / CUs: from the device (dev_info), MI355X: 160 KiB, 156 KiB, 256
#define LDS_TOTAL (dev_info().lds_per_cu)
#define LDS_BUDGET (dev_info().lds_per_cu - 4096)
#define NUM_CU (dev_info().cus)

int cik_of(int CI) { return CI == 3 ? 4 : CI <= 8 ? 8 : CI <= 16 ? 16 : CI <= 32 ? 32 : 64; }

// instantiated (MI, NB): accumulators + double-buffered fragments + the 12 prefetch slots stay inside 256 registers
// (4 x 2, 4 x 4, 3 x 4 and anything x 8 spill)
bool fwd_cfg_ok(int mi, int nb) { return nb == 1 ? mi <= 4 : nb == 2 ? mi <= 3 : nb == 4 ? mi <= 2 : false; }

template <int CIK, int MI, int NB>
void launch_fwd_one(hipStream_t s, const DcFwd& P, int occ, dim3 grid, size_t lds, int ntiles, int nslots) {
    if constexpr ((NB == 1 && MI <= 4) || (NB == 2 && MI <= 3) || (NB == 4 && MI <= 2)) {
        if constexpr (MI * NB <= 2) {          // (MI x NB = 4 spills at 128 registers)
            if (occ == 2) {                                   // the two-blocks-per-CU build (<= 128 registers, 4 prefetch slots)
                ensure_dyn_lds((const void*)dconv_fwd_kernel<CIK, MI, NB, DC_PF_SMALL>, (size_t)LDS_TOTAL / 2);
                hipLaunchKernelGGL((dconv_fwd_kernel<CIK, MI, NB, DC_PF_SMALL>), grid, dim3(DC_THREADS), lds, s, P, ntiles, nslots);
                return;
            }
        }
        ensure_dyn_lds((const void*)dconv_fwd_kernel<CIK, MI, NB>, (size_t)LDS_TOTAL);
        hipLaunchKernelGGL((dconv_fwd_kernel<CIK, MI, NB>), grid, dim3(DC_THREADS), lds, s, P, ntiles, nslots);
    }
}

template <int CIK, int MI>
void launch_fwd_nb(hipStream_t s, const DcFwd& P, int NB, int occ, dim3 grid, size_t lds, int ntiles, int nslots) {
    if (NB == 4) launch_fwd_one<CIK, MI, 4>(s, P, occ, grid, lds, ntiles, nslots);
    else if (NB == 2) launch_fwd_one<CIK, MI, 2>(s, P, occ, grid, lds, ntiles, nslots);
    else launch_fwd_one<CIK, MI, 1>(s, P, occ, grid, lds, ntiles, nslots);
}

template <int CIK>
void launch_fwd_mi(hipStream_t s, const DcFwd& P, int MI, int NB, int occ, dim3 grid, size_t lds, int ntiles, int nslots) {
    if (MI == 4) launch_fwd_nb<CIK, 4>(s, P, NB, occ, grid, lds, ntiles, nslots);
    else if (MI == 3) launch_fwd_nb<CIK, 3>(s, P, NB, occ, grid, lds, ntiles, nslots);
    else if (MI == 2) launch_fwd_nb<CIK, 2>(s, P, NB, occ, grid, lds, ntiles, nslots);
    else launch_fwd_nb<CIK, 1>(s, P, NB, occ, grid, lds, ntiles, nslots);
}
}  // namespace

// measurement hook (tools/dconv_bench.hip): force the tile of the next launches; 0 = automatic
int g_dc_force[3] = {0, 0, 0};       // TH, TW, MI
int g_dc_occ = 0;   // 1: never two blocks per CU (the round-2 A/B switch: two blocks are +8 % on the h2 / h3 / d_h1 layers)
void dconv_force_tile(int th, int tw, int mi) { g_dc_force[0] = th; g_dc_force[1] = tw; g_dc_force[2] = mi; }
int g_dc_last[4] = {0, 0, 0, 0};     // TH, TW, MI, GT of the last forward launch

// (N, the column split and every row stride of the epilogue's tensors in multiples of 4: the epilogue moves float4s)
bool dconv_ok(int CI, int N) { return (CI == 3 || CI == 8 || CI == 16 || CI == 32 || CI == 64) && N >= 4 && N <= 128 && N % 4 == 0; }

// Fills tiles / LDS split and launches.  `span` = extent of the tap offsets (5 for the 5x5 taps over the input, 3 for a
// stride-2 transposed conv over its small grid).
//
// Tile choice by a small time model of the persistent kernel (cycles per tile on one CU):
//   compute = MFMAs of the busiest SIMD (row block rb sits on wave rb % 8, i.e. SIMD rb % 4) x 32
//   load    = tile bytes / 16 B per clock  (in flight under the previous tile's MFMA loop: only max(compute, load) shows)
//   land    = tile bytes / 79 B per clock  (ds_write_b128 of the prefetch registers), fixed = barriers + epilogue
//   T = max(compute, load) + (land + fixed) / blocks per CU,   score = real output pixels per tile / T
// subject to: tile + resident filter <= LDS, tile elements <= 512 x DC_PF prefetch slots.  A filter too big to stay
// resident beside any tile is run as several launches over column slices of 16 * NB (d_h3's input gradient in
// ContextAEReal: 25 x 32 x 32 floats = 100 KB -> two slices of 16 columns).
namespace {
template <int NSL, int MI, bool NB2, int NCLS, int OCC>
void launch_fwd2_one(hipStream_t s, const DcFwd& P, const DcFwd2& Q, dim3 grid, size_t lds, int ntiles, int nslots) {
    ensure_dyn_lds((const void*)dconv_fwd2_kernel<NSL, MI, NB2, NCLS, OCC>, (size_t)LDS_TOTAL / OCC);
    hipLaunchKernelGGL((dconv_fwd2_kernel<NSL, MI, NB2, NCLS, OCC>), grid, dim3(DC2_THREADS), lds, s, P, Q, ntiles, nslots);
}
constexpr int dc2_max_mi(int ncls) { return ncls == 4 ? 3 : 4; }        // units per wave: accumulators of every class live across the slices
template <int NSL, bool NB2, int NCLS>
bool launch_fwd2(hipStream_t s, const DcFwd& P, const DcFwd2& Q, int MI, int occ, dim3 grid, size_t lds, int ntiles, int nslots) {
#define DC2_CASE(mi)                                                                                                     \
    if constexpr (mi <= dc2_max_mi(NCLS)) {                                                                              \
        if (MI == mi) {                                                                                                  \
            (void)occ;                                                                                                   \
            launch_fwd2_one<NSL, mi, NB2, NCLS, 1>(s, P, Q, grid, lds, ntiles, nslots);                                  \
            return true;                                                                                                 \
        }                                                                                                                \
    }
    DC2_CASE(1) DC2_CASE(2) DC2_CASE(3) DC2_CASE(4)
#undef DC2_CASE
    return false;
}
}  // namespace

int g_dc2_last[5] = {0, 0, 0, 0, 0};   // TH, TW, units per wave, column blocks * 10 + occ, slices of the last dconv2 launch (0: the launch went to dconv_fwd_kernel)

static void launch_pack(hipStream_t s, const DcFwd& P, int CIK, int nslots) {
    const int total = nslots * CIK * P.NPT;
    const dim3 pg((unsigned)((total + 255) / 256 > 64 ? 64 : (total + 255) / 256));
    switch (CIK) {
        case 4: hipLaunchKernelGGL((dconv_pack_kernel<4>), pg, dim3(256), 0, s, P, P.NPT, nslots); break;
        case 8: hipLaunchKernelGGL((dconv_pack_kernel<8>), pg, dim3(256), 0, s, P, P.NPT, nslots); break;
        case 16: hipLaunchKernelGGL((dconv_pack_kernel<16>), pg, dim3(256), 0, s, P, P.NPT, nslots); break;
        case 32: hipLaunchKernelGGL((dconv_pack_kernel<32>), pg, dim3(256), 0, s, P, P.NPT, nslots); break;
        default: hipLaunchKernelGGL((dconv_pack_kernel<64>), pg, dim3(256), 0, s, P, P.NPT, nslots); break;
    }
}
// Points P.wp at the cache slot of this (filter, shape) and says whether its packed image is current (no pack launch needed).
// No cache / cache full: P.wp stays the caller's scratch image and the pack runs as before.
static bool pack_cached(DcFwd& P, int CIK, int nslots, hipStream_t s) {
    DcPackCache* pc = P.pc;
    if (!pc || !pc->arena) return false;
    int sig = P.ncls * 131 + P.osc * 17 + P.S;
    for (int c = 0; c < P.ncls; ++c) {
        sig = sig * 31 + P.cls[c].ntaps * 7 + P.cls[c].pslot0;
        for (int e = 0; e < P.cls[c].ntaps; ++e) sig = sig * 31 + P.taps[P.cls[c].tap0 + e].wt;
    }
    DcPackCache::Ent* ent = nullptr;
    for (int i = 0; i < pc->n; ++i) {
        DcPackCache::Ent& q = pc->ent[i];
        if (q.w == P.w && q.wmode == P.wmode && q.N == P.N && q.CI == P.CI && q.CIK == CIK && q.NPT == P.NPT && q.nslots == nslots && q.sig == sig) { ent = &q; break; }
    }
    if (!ent) {
        const int64_t need = ((int64_t)nslots * CIK * P.NPT + 64 + 63) / 64 * 64;          // the image + the 16 zeros behind it
        if (pc->n >= DcPackCache::MAXE || pc->used + need > pc->floats) return false;
        ent = &pc->ent[pc->n++];
        *ent = DcPackCache::Ent{P.w, P.wmode, P.N, P.CI, CIK, P.NPT, nslots, sig, pc->used, 0, nullptr};
        pc->used += need;
    }
    P.wp = pc->arena + ent->off;
    if (!pc->external && ent->version == pc->version && ent->stream == (void*)s) { pc->hits++; return true; }
    ent->version = pc->version;
    ent->stream = (void*)s;
    return false;
}

// The K-sliced, double-buffered LDS-DMA kernel of dconv2.h, for 16 / 32 input channels (either source split on a multiple of 4),
// up to 32 filter columns in LDS beside two slice buffers, 1 or 4 tap classes.  false: not for this layer (the caller falls back).
// Tile choice: a time model per tile on one CU --
//   compute = MFMAs of the busiest SIMD (unit u -> compute wave u % 8 -> SIMD u % 4) x 37 cycles + fold and stores
//   req     = DMA requests of a loader wave x 420 cycles;   dma = slice bytes / 24 B per clock
//   T = max(compute, req, dma) + per-slice barrier and skew;   score = real output pixels per tile / T
static bool dconv_launch2(hipStream_t s, DcFwd P, int span) {
    if (!(opt(OPT_DCONV) & 2)) return false;
    if ((P.CI != 16 && P.CI != 32) || P.N > 32 || (P.ncls != 1 && P.ncls != 4) || (P.c1 & 3) || (P.x2 && ((P.CI - P.c1) & 3))) return false;
    // stride-2 transposed convs (four tap classes) stay on dconv_fwd_kernel unless bit 4 asks: their tiles store 4 x the pixels they read, the
    // store path takes ~10 B / clk / CU whoever issues, and here the stores of ALL classes sit behind the closing barrier with the matrix
    // pipe idle (measured: d_h3 forward 0.169 -> 0.200 ms, h1_conv dx 0.157 -> 0.195)
    if (P.ncls == 4 && !(opt(OPT_DCONV) & 4)) return false;
    if (P.ep.nsplit < P.N && (P.ep.nsplit & 3)) return false;
    const int NSL = P.CI / 16, NBT = P.N > 16 ? 2 : 1, NP = NBT * 16;
    int nslots = 0;
    for (int c = 0; c < P.ncls; ++c) { P.cls[c].pslot0 = nslots; nslots += P.cls[c].ntaps; }
    const size_t wall = (size_t)nslots * P.CI * NP * 4 + (size_t)(nslots + 1) * 4 + 16;
    auto geo = [&](int th, int tw, int& ih, int& iw, int& iwp, int& iwh, size_t& slice) {
        ih = P.S * (th - 1) + span; iw = P.S * (tw - 1) + span;
        iwh = P.S == 2 ? (iw + 1) / 2 : 0;
        iwp = P.S == 2 ? 2 * iwh : iw;
        slice = ((size_t)ih * iwp * 16 + 255) / 256 * 256 * 4;            // bytes, whole 1 KB pieces
    };
    const int max_mi = dc2_max_mi(P.ncls);
    int best_th = 0, best_tw = 0, best_mi = 0, best_occ = 1;
    double best = -1;
    for (int tw = 16; tw <= 64 && tw <= (P.wlog + 15) / 16 * 16; tw *= 2)
        for (int th = 1; th <= 32; ++th) {
            const int nunits = th * (tw / 16) * NBT, mi = (nunits + DC2_NCW - 1) / DC2_NCW;
            if (mi > max_mi) break;
            int ih, iw, iwp, iwh; size_t slice;
            geo(th, tw, ih, iw, iwp, iwh, slice);
            const size_t lds = 2 * slice + wall;
            if (lds > (size_t)LDS_BUDGET) break;
            int load[4] = {0, 0, 0, 0};
            for (int u = 0; u < nunits; ++u) load[u & 3] += 1;             // unit u -> compute wave u % 8 -> SIMD u % 4
            int busiest = 1;
            for (int q = 0; q < 4; ++q) busiest = load[q] > busiest ? load[q] : busiest;
            const int occ = 1;
            const int tiles_y = (P.hlog + th - 1) / th, tiles_x = (P.wlog + tw - 1) / tw;
            const double rows = (double)P.hlog * P.wlog / ((double)tiles_y * tiles_x);
            // MFMA loops of a SIMD's two compute waves (85 % busy; a little less with one unit per wave: four MFMAs per LDS round trip) + fold and stores
            const double compute = (double)busiest * 4.0 * nslots * NSL * (mi == 1 ? 40.0 : 37.0) + 250.0 * P.ncls * mi;
            const double req = ((double)(slice / 1024 + 3) / 4 * 300.0 + 1500.0) * NSL;                  // DMA requests of a loader wave + the last one's latency
            // ... and their bytes: a slice's DMA is LATENCY first (measured: ~7 k cycles for 30 KB with every CU loading -- the loaders
            // have one slice in flight and nothing behind it), then ~11 B / clk / CU
            const double dma = (5000.0 + (double)slice / 11.0) * NSL;
            double T = compute > dma ? compute : dma;
            if (req > T) T = req;
            T += NSL * 700.0 / occ;                                                                       // barrier + skew per slice
            const double score = rows / T;
            if (score > best) { best = score; best_th = th; best_tw = tw; best_mi = mi; best_occ = occ; }
        }
    if (!best_th) return false;
    if (const char* e = getenv("DC2_TILE")) {                                // measurement hook: "th,tw" for every dconv2 launch it fits
        int th = 0, tw = 0;
        if (sscanf(e, "%d,%d", &th, &tw) == 2 && th > 0 && (tw == 16 || tw == 32 || tw == 64)) {
            int ih, iw, iwp, iwh; size_t slice;
            geo(th, tw, ih, iw, iwp, iwh, slice);
            const int mi = (th * (tw / 16) * NBT + DC2_NCW - 1) / DC2_NCW;
            if (mi <= max_mi && 2 * slice + wall <= (size_t)LDS_BUDGET && tw <= (P.wlog + 15) / 16 * 16) { best_th = th; best_tw = tw; best_mi = mi; }
        }
    }
    if (g_dc_force[0]) {
        best_th = g_dc_force[0]; best_tw = g_dc_force[1];
        int ih, iw, iwp, iwh; size_t slice;
        geo(best_th, best_tw, ih, iw, iwp, iwh, slice);
        const int nunits = best_th * (best_tw / 16) * NBT;
        best_mi = (nunits + DC2_NCW - 1) / DC2_NCW;
        g_dc2_last[0] = 0;
        if (best_mi != g_dc_force[2] || best_mi > max_mi || 2 * slice + wall > (size_t)LDS_BUDGET) return true;   // (tile sweep of the bench: skip)
        best_occ = 1;
    }
    P.TH = best_th; P.TW = best_tw;
    DcFwd2 Q{};
    size_t slice;
    geo(P.TH, P.TW, P.IH, P.IW, Q.IWP, Q.IWH, slice);
    Q.slice_floats = (int)(slice / 4);
    Q.div_mul = (unsigned)(((1u << 20) + Q.IWP - 1) / Q.IWP);
    for (int pi = 0; pi < Q.slice_floats / 16; ++pi)                       // the multiply-shift division must be exact on every pixel index
        if ((int)(((unsigned)pi * Q.div_mul) >> 20) != pi / Q.IWP) return false;
    P.CIP = 16;
    P.tiles_y = (P.hlog + P.TH - 1) / P.TH;
    P.tiles_x = (P.wlog + P.TW - 1) / P.TW;
    P.NPT = NP; P.n0 = 0;
    const bool packed = pack_cached(P, P.CI, nslots, s);
    Q.zeros = P.wp + (size_t)nslots * P.CI * P.NPT;
    const int ntiles = P.nimg * P.tiles_y * P.tiles_x;
    const int slots = NUM_CU * best_occ;
    const int rounds = (ntiles + slots - 1) / slots;
    const dim3 grid((unsigned)((ntiles + rounds - 1) / rounds));
    const size_t lds = 2 * slice + wall;
    if (!packed) launch_pack(s, P, P.CI, nslots);
    bool ok;
#define DC2_GO(nsl, nb2, ncls) launch_fwd2<nsl, nb2, ncls>(s, P, Q, best_mi, best_occ, grid, lds, ntiles, nslots)
    if (NSL == 1) ok = NBT == 1 ? (P.ncls == 1 ? DC2_GO(1, false, 1) : DC2_GO(1, false, 4)) : (P.ncls == 1 ? DC2_GO(1, true, 1) : DC2_GO(1, true, 4));
    else ok = NBT == 1 ? (P.ncls == 1 ? DC2_GO(2, false, 1) : DC2_GO(2, false, 4)) : (P.ncls == 1 ? DC2_GO(2, true, 1) : DC2_GO(2, true, 4));
#undef DC2_GO
    if (ok) { g_dc2_last[0] = P.TH; g_dc2_last[1] = P.TW; g_dc2_last[2] = best_mi; g_dc2_last[3] = NBT * 10 + best_occ; g_dc2_last[4] = NSL; }
    return ok;
}

static void dconv_launch(hipStream_t s, DcFwd P, int span) {
    g_dc2_last[0] = 0;
    if (dconv_launch2(s, P, span)) return;
    const int CIK = cik_of(P.CI), CIP = dc_cip(CIK, P.S);
    P.CIP = CIP;
    int NBT = (P.N + 15) / 16;
    NBT = NBT <= 1 ? 1 : NBT <= 2 ? 2 : NBT <= 4 ? 4 : 8;
    const int TPC = CIK >= 16 ? 1 : 16 / CIK, CPT = CIK >= 16 ? CIK / 16 : 1;
    // the filter in the LDS image's order: classes padded to whole 16-k chunks
    int nslots = 0, nchunks = 0;
    for (int c = 0; c < P.ncls; ++c) {
        P.cls[c].pslot0 = nslots;
        const int ntp = (P.cls[c].ntaps + TPC - 1) / TPC * TPC;
        nslots += ntp;
        nchunks += ntp / TPC * CPT;
    }
    int best_th = 0, best_mi = 1, best_tw = 16, best_nb = 1, best_occ = 1;
    double best = -1;
    for (int nb = NBT > 4 ? 4 : NBT; nb >= 1; nb /= 2) {
        const size_t wall = (size_t)nslots * CIK * nb * 16 * 4 + (size_t)(nslots * CIK / 16 + 1) * 16;
        for (int tw = 16; tw <= 64 && tw <= (P.wlog + 15) / 16 * 16; tw *= 2)
            for (int mi = 1; mi <= 4 && fwd_cfg_ok(mi, nb); ++mi)
                for (int th = 1; th <= 32; ++th) {
                    const int nrb = th * (tw / 16);
                    if (nrb > DC_NW * mi) break;
                    if (mi > 1 && nrb <= DC_NW * (mi - 1)) continue;      // a smaller MI covers this tile
                    const int ih = P.S * (th - 1) + span, iw = P.S * (tw - 1) + span;
                    const size_t tile = (size_t)((ih * iw * CIP + 3) & ~3) * 4;
                    if (tile + wall > (size_t)LDS_BUDGET) break;
                    if ((int64_t)ih * iw * (CIK == 4 ? 3 : CIK / 4) > (int64_t)DC_THREADS * DC_PF) break;
                    int load[4] = {0, 0, 0, 0};
                    for (int rb = 0; rb < nrb; ++rb) load[rb & 3] += 1;
                    int busiest = 1;
                    for (int q = 0; q < 4; ++q) busiest = load[q] > busiest ? load[q] : busiest;
                    // two blocks per CU: both fit LDS, the tile fits the 4 prefetch slots of the <= 128-register build, few accumulators
                    // (never at the price of extra column slices: measured slower on the 32-column layers)
                    const int occ = (g_dc_occ != 1 && nb == (NBT > 4 ? 4 : NBT) && 2 * (tile + wall) + 1024 <= (size_t)LDS_TOTAL && mi * nb <= 2 &&
                                     (int64_t)ih * iw * (CIK == 4 ? 3 : CIK / 4) <= (int64_t)DC_THREADS * DC_PF_SMALL) ? 2 : 1;
                    const int tiles_y = (P.hlog + th - 1) / th, tiles_x = (P.wlog + tw - 1) / tw;
                    const double rows = (double)P.hlog * P.wlog / ((double)tiles_y * tiles_x);          // real output pixels per tile
                    const double compute = (double)busiest * nb * 4.0 * nchunks * 32.0;
                    const double loadc = (double)tile / 16.0, land = (double)tile / 79.0, fixed = 3000.0 + 40.0 * mi * nb * P.ncls;   // (barriers, prefetch issue, epilogue: calibrated on the tile sweeps of tools/dconv_bench.hip)
                    const double T = (compute > loadc ? compute : loadc) + (land + fixed) / occ;
                    const double score = rows / (T * (NBT / nb));
                    if (score > best) { best = score; best_th = th; best_mi = mi; best_tw = tw; best_nb = nb; best_occ = occ; }
                }
    }
    if (!best_th) { set_launch_error("dconv: no tile fits LDS for CI %d, N %d", P.CI, P.N); return; }
    if (g_dc_force[0]) {
        best_th = g_dc_force[0]; best_tw = g_dc_force[1]; best_mi = g_dc_force[2];
        const int ih = P.S * (best_th - 1) + span, iw = P.S * (best_tw - 1) + span;
        const size_t tile = (size_t)((ih * iw * CIP + 3) & ~3) * 4, wall = (size_t)nslots * CIK * best_nb * 16 * 4 + (size_t)(nslots * CIK / 16 + 1) * 16;
        g_dc_last[0] = 0;
        if (tile + wall > (size_t)LDS_BUDGET || (int64_t)ih * iw * (CIK == 4 ? 3 : CIK / 4) > (int64_t)DC_THREADS * DC_PF) return;
        best_occ = (g_dc_occ != 1 && best_nb == (NBT > 4 ? 4 : NBT) && 2 * (tile + wall) + 1024 <= (size_t)LDS_TOTAL && best_mi * best_nb <= 2 &&
                    (int64_t)ih * iw * (CIK == 4 ? 3 : CIK / 4) <= (int64_t)DC_THREADS * DC_PF_SMALL) ? 2 : 1;
        if (!fwd_cfg_ok(best_mi, best_nb)) return;
    }
    P.TW = best_tw;
    P.TH = best_th;
    const int MI = best_mi, NB = best_nb;
    P.IH = P.S * (P.TH - 1) + span;
    P.IW = P.S * (P.TW - 1) + span;
    P.tiles_y = (P.hlog + P.TH - 1) / P.TH;
    P.tiles_x = (P.wlog + P.TW - 1) / P.TW;
    P.NPT = NBT * 16;
    const size_t tile = (size_t)((P.IH * P.IW * CIP + 3) & ~3) * 4;
    g_dc_last[0] = P.TH; g_dc_last[1] = P.TW; g_dc_last[2] = MI; g_dc_last[3] = NB * 10 + best_occ;
    const size_t lds = tile + (size_t)nslots * CIK * NB * 16 * 4 + (size_t)(nslots * CIK / 16 + 1) * 16;      // tile | filter | offset table
    const int ntiles = P.nimg * P.tiles_y * P.tiles_x;
    // persistent grid: whole rounds over the CUs' block slots
    const int slots = NUM_CU * best_occ;
    const int rounds = (ntiles + slots - 1) / slots;
    const dim3 grid((unsigned)((ntiles + rounds - 1) / rounds));
    if (!pack_cached(P, CIK, nslots, s)) launch_pack(s, P, CIK, nslots);
    for (int n0 = 0; n0 < P.N; n0 += NB * 16) {
        P.n0 = n0;
        switch (CIK) {
            case 4: launch_fwd_mi<4>(s, P, MI, NB, best_occ, grid, lds, ntiles, nslots); break;
            case 8: launch_fwd_mi<8>(s, P, MI, NB, best_occ, grid, lds, ntiles, nslots); break;
            case 16: launch_fwd_mi<16>(s, P, MI, NB, best_occ, grid, lds, ntiles, nslots); break;
            case 32: launch_fwd_mi<32>(s, P, MI, NB, best_occ, grid, lds, ntiles, nslots); break;
            default: launch_fwd_mi<64>(s, P, MI, NB, best_occ, grid, lds, ntiles, nslots); break;
        }
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

// conv2d kh x kw (<= 25 taps, kh, kw <= 7), stride s, pad_before (pady, padx), output grid hout x wout given (SAME or VALID): the
// Inception front end's narrow stem layers (nets/inception_v3.py:101-116: 3x3 32->32 VALID, 3x3 32->64 SAME, 1x1 64->80).  w[tap][k][n].
void dconv_conv_k(hipStream_t s, DcFwd P, int kh, int kw, int stride, int pady, int padx, int hout, int wout) {
    P.S = stride; P.y_org = -pady; P.x_org = -padx;
    P.hlog = P.hout = hout; P.wlog = P.wout = wout; P.osc = 1;
    const int mdiv = kw == 1 ? 256 : kw == 2 ? 128 : kw == 3 ? 86 : kw == 4 ? 64 : kw == 5 ? 52 : kw == 6 ? 43 : 37;
    P.ncls = 1; P.cls[0] = DcClass{0, kh * kw, 0, 0, 0, kw, mdiv, 0, 0, 1};
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx) P.taps[ky * kw + kx] = DcTap{(int16_t)ky, (int16_t)kx, (int16_t)(ky * kw + kx), 0};
    dconv_launch(s, P, kh > kw ? kh : kw);
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
    for (int k = 0; k < 4; ++k) {
        const int c = 3 - k;                                 // parity classes (1,1), (1,0), (0,1), (0,0): 9, 6, 6, 4 taps -- longest first (dconv2.h)
        const int py = c >> 1, px = c & 1;
        const int pary = (py + 1) & 1, parx = (px + 1) & 1, nty = (5 - pary + 1) / 2, ntx = (5 - parx + 1) / 2;
        const int oy = (py + 1 - pary) / 2, ox = (px + 1 - parx) / 2;
        P.cls[k] = DcClass{nt, nty * ntx, py, px, 0, ntx, ntx == 2 ? 128 : 86, oy + 1, ox + 1, -1};
        for (int sy = 0; sy < nty; ++sy)
            for (int sx = 0; sx < ntx; ++sx)
                P.taps[nt++] = DcTap{(int16_t)(oy - sy + 1), (int16_t)(ox - sx + 1), (int16_t)((pary + 2 * sy) * 5 + parx + 2 * sx), 0};
    }
    dconv_launch(s, P, 3);
}

// ---- filter gradient ------------------------------------------------------------------------------------------
// out[m][n0 + n] = sum over the blocks' slabs, in a fixed order.  A block of 16 x 16 threads covers 16 float4s of the [M][NP]
// image; thread (e, g) adds slabs g, g + 16, ... (independent loads, in flight together), then the 16 partials of an element
// are added in order through LDS.  (One thread walking all 256 slabs of its element took 60 us per filter gradient -- more
// than the gradient kernel of the small layers.)
__global__ __launch_bounds__(256) void dconv_wgrad_reduce_kernel(const float4* __restrict__ slab, int nslab, int M, int NP, int ncols, int n0, int CB, float* __restrict__ out) {
    __shared__ float4 part[16][16];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int E4 = M * NP / 4, q = blockIdx.x * 16 + e;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < E4) {
        const float4* p = slab + q;
        for (int s0 = g; s0 < nslab; s0 += 64) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = s0 + 16 * u < nslab ? p[(int64_t)(s0 + 16 * u) * E4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
        }
    }
    part[g][e] = v;
    __syncthreads();
    if (g == 0 && q < E4) {
        float4 r = part[0][e];
        for (int k = 1; k < 16; ++k) { r.x += part[k][e].x; r.y += part[k][e].y; r.z += part[k][e].z; r.w += part[k][e].w; }
        const int m = q / (NP / 4), n = (q - m * (NP / 4)) * 4;
        float* o = out + (int64_t)m * CB + n0 + n;
        if (n + 0 < ncols) o[0] = r.x;
        if (n + 1 < ncols) o[1] = r.y;
        if (n + 2 < ncols) o[2] = r.z;
        if (n + 3 < ncols) o[3] = r.w;
    }
}

void dconv_wgrad_reduce(hipStream_t s, const float* slab, int nslab, int M, int NP, int ncols, int n0, int CB, float* out) {
    const int rb = (M * NP / 4 + 15) / 16;
    hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, (const float4*)slab, nslab, M, NP, ncols, n0, CB, out);
}

namespace {
template <int CAK, int RBW, int WM, int SS>
void launch_wg_nb(hipStream_t s, const DcWgrad& P, int NB, dim3 grid, size_t lds) {
#define DC_CASE(nb)                                                                                                          \
    case nb: {                                                                                                               \
        ensure_dyn_lds((const void*)dconv_wgrad_kernel<CAK, RBW, WM, nb, SS>, (size_t)LDS_TOTAL);                                \
        hipLaunchKernelGGL((dconv_wgrad_kernel<CAK, RBW, WM, nb, SS>), grid, dim3(DC_THREADS), lds, s, P);                       \
        break;                                                                                                               \
    }
    if constexpr (CAK == 4) {
        switch (NB) { DC_CASE(1) DC_CASE(2) DC_CASE(4) default: break; }
    } else {
        switch (NB) { DC_CASE(1) DC_CASE(2) default: break; }
    }
#undef DC_CASE
}
}  // namespace

// dw[tap][a][b] (a: channels of `big`, b: channels of [s1 | s2]); slab: scratch of slab_floats floats
void dconv_wgrad(hipStream_t s, DcWgrad P, float* slab, int64_t slab_floats) {
    if (c3wgrad_ok(P)) { c3wgrad(s, P, slab, slab_floats); return; }      // three big-grid channels: the compile-time-geometry kernel (c3wgrad.hip)
    const int CAK = P.CA == 3 ? 4 : P.CA <= 8 ? 8 : P.CA <= 16 ? 16 : 32;
    const int CAP = CAK == 4 ? 4 : CAK + 4;
    // waves: WM groups along M x WK along K (the instantiations below); RBW row blocks of 16 per wave
    const int RBW = CAK == 4 ? 5 : 7, WM = CAK == 4 ? 1 : CAK == 8 ? 2 : CAK == 16 ? 4 : 8, WK = DC_NW / WM;
    int NBT = (P.CB + 15) / 16;
    NBT = NBT <= 1 ? 1 : NBT <= 2 ? 2 : NBT <= 4 ? 4 : 8;
    const int nb_max = CAK == 4 ? 4 : 2;                   // accumulators RBW * NB * 4 + fragments + prefetch slots inside 256 registers
    const int NB = NBT < nb_max ? NBT : nb_max;
    const int NP = NB * 16, CBP = dc_cbp(NP, P.S);
    P.M = 25 * P.CA;
    P.TW = P.ws >= 64 ? 64 : P.ws >= 32 ? 32 : 16;
    P.tw_sh = P.TW == 64 ? 6 : P.TW == 32 ? 5 : 4;
    // TH: the height that keeps big halo + small tile inside LDS and the prefetch slots with the least halo / ragged overhead
    const size_t red = WK > 1 ? (size_t)(WK / 2) * WM * RBW * NB * 256 * 4 : 0;
    int best_th = 1;
    double best = -1;
    for (int th = 1; th <= 32; ++th) {
        const int ih = P.S * (th - 1) + 5, iw = P.S * (P.TW - 1) + 5;
        const size_t lds = (size_t)((ih * iw * CAP + 3) & ~3) * 4 + (size_t)th * P.TW * CBP * 4;
        if (lds > (size_t)LDS_BUDGET) break;
        if ((int64_t)ih * iw * (CAK == 4 ? 3 : CAK / 4) > (int64_t)DC_THREADS * DC_PFB) break;
        if ((int64_t)th * P.TW * (NP / 4) > (int64_t)DC_THREADS * dc_pfs(NB)) break;
        const int tiles = (P.hs + th - 1) / th;
        const int chunks = th * P.TW / 16, per_wave = (chunks + WK - 1) / WK;       // K chunks per wave: whole rounds over the WK groups
        const double eff = (double)P.hs / (tiles * th) * ((double)th / (th + 4.0 / P.S)) * ((double)chunks / (per_wave * WK));
        if (eff > best) { best = eff; best_th = th; }
    }
    P.TH = best_th;
    P.IH = P.S * (P.TH - 1) + 5;
    P.IW = P.S * (P.TW - 1) + 5;
    P.tiles_y = (P.hs + P.TH - 1) / P.TH;
    P.tiles_x = (P.ws + P.TW - 1) / P.TW;
    P.ntiles = P.nimg * P.tiles_y * P.tiles_x;
    size_t lds = (size_t)((P.IH * P.IW * CAP + 3) & ~3) * 4 + (size_t)P.TH * P.TW * CBP * 4;
    if (lds < red) lds = red;
    // persistent blocks, one per CU (8 waves, 150-250 registers), in whole rounds over the tiles
    int64_t nblk = NUM_CU;
    const int64_t cap = slab_floats / ((int64_t)(P.M + 1) * NP);         // (+ 1 row: the bias-gradient partials)
    if (nblk > cap) nblk = cap;
    if (nblk > P.ntiles) nblk = P.ntiles;
    if (nblk < 1) nblk = 1;
    {
        const int64_t rounds = (P.ntiles + nblk - 1) / nblk;
        nblk = (P.ntiles + rounds - 1) / rounds;
    }
    P.slab = slab;
    P.dbslab = slab + nblk * (int64_t)P.M * NP;
    const dim3 grid((unsigned)nblk);
    for (int n0 = 0; n0 < P.CB; n0 += NP) {
        P.n0 = n0;
#define DC_WG(ss)                                                            \
        if (CAK == 4) launch_wg_nb<4, 5, 1, ss>(s, P, NB, grid, lds);        \
        else if (CAK == 8) launch_wg_nb<8, 7, 2, ss>(s, P, NB, grid, lds);   \
        else if (CAK == 16) launch_wg_nb<16, 7, 4, ss>(s, P, NB, grid, lds); \
        else launch_wg_nb<32, 7, 8, ss>(s, P, NB, grid, lds);
        if (P.S == 1) { DC_WG(1) } else { DC_WG(2) }
#undef DC_WG
        const int ncols = P.CB - n0 < NP ? P.CB - n0 : NP;
        const int rb = (P.M * NP / 4 + 15) / 16;
        hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, s, (const float4*)slab, (int)nblk, P.M, NP, ncols, n0, P.CB, P.out);
        if (P.db)                                            // the same fixed-order sum over the blocks' partial column sums (one row of NP)
            hipLaunchKernelGGL(dconv_wgrad_reduce_kernel, dim3(1), dim3(256), 0, s, (const float4*)P.dbslab, (int)nblk, 1, NP, ncols, n0, P.CB, P.db);
    }
}

}  // namespace ctx
