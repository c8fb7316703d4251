// gemm_launch.h -- tile-shape dispatch and split-K policy for igemm_kernel (included by gemm_*.hip)
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <typeinfo>

#include "launch.h"
#include "igemm_split.h"

namespace ctx {

void splitk_reduce(hipStream_t s, const Epi& ep, int M, int N, int nprob, int nsplit);

template <class LA, class LB, int MI, int NI, int WM, int WN>
static void launch_tile_split(hipStream_t s, const LA& a, const LB& b, Epi ep, int M, int N, int nprob, int nsplit) {
    constexpr int NT = 64 * WM * WN, TM = 32 * MI * WM, TN = 32 * NI * WN;
    constexpr size_t lds = 2 * (size_t)(STile<LA::KM, TM, NT>::FLOATS + STile<LB::KM, TN, NT>::FLOATS) * sizeof(float);
    if (lds > 65536) ensure_dyn_lds((const void*)igemm_split_kernel<LA, LB, MI, NI, WM, WN>, lds);
    const int gm = (M + TM - 1) / TM, gn = (N + TN - 1) / TN;
    int64_t nblk = (int64_t)gm * gn * nprob * nsplit;
    if (nblk < 64) ep.xcd_swizzle = 0;
    if (ep.xcd_swizzle && ep.swz_group) {                 // grouped: one group = one (split, parity class); needs whole groups of 8
        const int64_t grp = (int64_t)gm * gn * (nprob / 4);
        if (grp % 8 == 0) ep.swz_group = (int)grp; else ep.xcd_swizzle = 0;
    }
    if (ep.xcd_swizzle && !ep.swz_group) nblk = (nblk + 7) / 8 * 8;
    dim3 grid((unsigned)nblk);
    hipLaunchKernelGGL((igemm_split_kernel<LA, LB, MI, NI, WM, WN>), grid, dim3(NT), lds, s, a, b, ep, M, N, nprob, nsplit, gm, gn);
}

template <class LA, class LB, int MI, int NI, int WM, int WN>
static void launch_tile_f32(hipStream_t s, const LA& a, const LB& b, Epi ep, int M, int N, int nprob, int nsplit) {
    constexpr int NT = 64 * WM * WN, TM = 32 * MI * WM, TN = 32 * NI * WN;
    // two LDS stages of [A tile | B tile]; above the 64 KiB default the limit is raised once per kernel
    constexpr size_t lds = 2 * (size_t)(Tile<LA::KM, TM, NT>::FLOATS + Tile<LB::KM, TN, NT>::FLOATS) * sizeof(float);
    if (lds > 65536) ensure_dyn_lds((const void*)igemm_kernel<LA, LB, MI, NI, WM, WN>, lds);
    const int gm = (M + TM - 1) / TM, gn = (N + TN - 1) / TN;
    int64_t nblk = (int64_t)gm * gn * nprob * nsplit;
    if (nblk < 64) ep.xcd_swizzle = 0;
    if (ep.xcd_swizzle && ep.swz_group) {                 // grouped: one group = one (split, parity class); needs whole groups of 8
        const int64_t grp = (int64_t)gm * gn * (nprob / 4);
        if (grp % 8 == 0) ep.swz_group = (int)grp; else ep.xcd_swizzle = 0;
    }
    if (ep.xcd_swizzle && !ep.swz_group) nblk = (nblk + 7) / 8 * 8;
    dim3 grid((unsigned)nblk);
    hipLaunchKernelGGL((igemm_kernel<LA, LB, MI, NI, WM, WN>), grid, dim3(NT), lds, s, a, b, ep, M, N, nprob, nsplit, gm, gn);
}

// BIG = the 8-wave 256x256 tile is instantiated for this loader pair
template <class LA, class LB, int MI, int NI, int WM, int WN>
static void launch_tile(hipStream_t s, const LA& a, const LB& b, const Epi& ep, int M, int N, int nprob, int nsplit, int prec) {
    if (prec) launch_tile_split<LA, LB, MI, NI, WM, WN>(s, a, b, ep, M, N, nprob, nsplit);
    else launch_tile_f32<LA, LB, MI, NI, WM, WN>(s, a, b, ep, M, N, nprob, nsplit);
}

// The 128x128 block tile comes in three wave layouts: W = 0: 4 waves of 64x64; 1: 8 waves of 64x32; 2: 8 waves of 32x64.
// (a 16-wave 256x128 / 128x256 tile, one block per CU, was measured too: +3..+14 % time; not kept.)
// Eight waves (4 per SIMD at two blocks per CU) hide more of the load latency: measured per loader pair and precision at
// B = 256 (f32: -6..-20 % everywhere; bf16x3: the conv gather prefers 4 waves), so each launcher names its pair's choice.
template <class LA, class LB, int W, bool SPLIT>
static void launch_128(hipStream_t s, const LA& a, const LB& b, const Epi& ep, int M, int N, int nprob, int nsplit) {
    if constexpr (SPLIT) {
        if constexpr (W == 1) launch_tile_split<LA, LB, 2, 1, 2, 4>(s, a, b, ep, M, N, nprob, nsplit);
        else if constexpr (W == 2) launch_tile_split<LA, LB, 1, 2, 4, 2>(s, a, b, ep, M, N, nprob, nsplit);
        else launch_tile_split<LA, LB, 2, 2, 2, 2>(s, a, b, ep, M, N, nprob, nsplit);
    } else {
        if constexpr (W == 1) launch_tile_f32<LA, LB, 2, 1, 2, 4>(s, a, b, ep, M, N, nprob, nsplit);
        else if constexpr (W == 2) launch_tile_f32<LA, LB, 1, 2, 4, 2>(s, a, b, ep, M, N, nprob, nsplit);
        else launch_tile_f32<LA, LB, 2, 2, 2, 2>(s, a, b, ep, M, N, nprob, nsplit);
    }
}

template <class L> struct is_plain_loader : std::integral_constant<bool, std::is_same<L, KmPlain>::value || std::is_same<L, NmPlain>::value || std::is_same<L, NmPlain2>::value> {};

// prob_weight (optional, nprob entries): relative length of each problem -- the launcher then runs them in balanced_order
template <class LA, class LB, bool BIG = false, int W32 = 1, int WSP = 0>
static void launch_igemm(hipStream_t s, const LA& a, const LB& b, Epi ep, int M, int N, int nprob, int min_chunks,
                         SplitWs ws, int max_chunks = 0, const int* prob_weight = nullptr) {
    if constexpr (BIG) {
        const int64_t big_tiles = (int64_t)((M + 255) / 256) * ((N + 255) / 256) * nprob;
        // (since the 128x128 tile runs eight waves the 256x256 tile only pays in the split-bf16 mode: f32 conv gather -2.4 % without it)
        if (M >= 256 && N >= 256 && big_tiles >= 192 && ws.prec) {
            ep.slab = nullptr;
            launch_tile<LA, LB, 2, 4, 4, 2>(s, a, b, ep, M, N, nprob, 1, ws.prec);
            return;
        }
    }
    // (a 256x64 tile of four 64x64 waves for N <= 64 was measured: 1 block/CU, -15..-25 % vs 128x64; not kept)
    // 128-wide tiles unless they would be >= 20 % padding where 64-wide ones are not (192 images or 192 channels: 3 x 64, not 2 x 128)
    auto wide = [](int X) { if (X <= 64) return 1; const int w128 = (X + 127) / 128 * 128, w64 = (X + 63) / 64 * 64; return (w128 - w64) * 5 >= w128 ? 1 : 2; };
    int MI = wide(M), NI = wide(N);
    // FC layers (one problem, plain operands) on 64x64 tiles: four times the tiles for a quarter of the split-K -- shorter slabs and
    // combines, and blocks that fit beside the conv launches of the other lanes.  Whole-step A/B on one box, two runs each (round 5):
    // all 128x128: 13.00 / 12.94 ms; launches of <= 64 tiles: 12.87 / 12.91; <= 128: 12.90 / 12.87; all: 12.85 / 12.83 -- with the FC filter
    // gradients (NmPlain x NmPlain) slower in the last (0.313 -> 0.324 ms), so those switch only up to 128 tiles.  Exact f32 only.
    // The other one-problem launches (image-major convs and the transposed-conv products of small-batch inference calls, the front end's
    // small maps): 64x64 tiles up to 256 tiles' worth -- translate at 25 frames 0.63 -> 0.60 ms, encode 0.281 -> 0.272, config 4's step
    // 6.445 -> 6.395 ms; without the cap the Inception front end loses 1 % (3.92 -> 3.96 ms).
    if (nprob == 1 && !ws.prec && !(is_plain_loader<LA>::value && is_plain_loader<LB>::value) && M >= 128 && N >= 128 &&
        (int64_t)((M + 127) / 128) * ((N + 127) / 128) <= 256) MI = NI = 1;
    if (nprob == 1 && !ws.prec && is_plain_loader<LA>::value && is_plain_loader<LB>::value && M >= 128 && N >= 128) {
        const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
        const bool dw = !LA::KM && !LB::KM;
        if (!dw || t128 <= 128) MI = NI = 1;
    }
    const int64_t tiles = (int64_t)((M + 64 * MI - 1) / (64 * MI)) * ((N + 64 * NI - 1) / (64 * NI)) * nprob;
    // Split-K by a cost model, not a block-count target: a launch takes `rounds` passes over the resident
    // slots (256 CUs x blocks/CU allowed by LDS), so 400 or 800 blocks on 512 slots run at 78 % -- the split
    // count is chosen to minimise  rounds * (chunks per block + fixed cost)  + slab write/read time.
    int nsplit = 1;
    if (min_chunks >= 8 && ws.slab) {
        const size_t lds = ws.prec ? 2 * (size_t)(64 * MI + 64 * NI) * SLDR * sizeof(float)
                                   : 2 * (size_t)((LA::KM ? 64 * MI * LDK : KC * 64 * MI) + (LB::KM ? 64 * NI * LDK : KC * 64 * NI)) * sizeof(float);
        int occ = (int)(dev_info().lds_per_cu / lds);
        if (occ > (MI * NI == 4 ? 2 : 4)) occ = MI * NI == 4 ? 2 : 4;            // register-file limit
        const double slots = (double)dev_info().cus * occ;
        // `occ` waves share each SIMD's matrix pipe; the split-bf16 chunk is 6 x 32 cycles per 32x32 tile but
        // runs at about half the pipe rate (conversion + LDS traffic), so ~1/3 of the f32 chunk
        const double t_chunk = 16.0 * MI * NI * 64.0 * occ / 2.3e9 * (ws.prec ? 0.35 : 1.0);
        const int64_t cap_ws = ws.slab_floats / ((int64_t)nprob * M * N);
        double best = 1e30;
        // (at least 4 chunks per block; 8 / 16 for the FC layers' 64x64 tiles measured: no difference, round 5)
        for (int n = 1; n <= 256 && n <= min_chunks / 4 && (n == 1 || n <= cap_ws); ++n) {
            const double rounds = std::ceil(tiles * n / slots);
            double len = rounds * ((min_chunks + n - 1) / n + 6);
            // (a longest-problem term -- a launch cannot end before its longest block -- made d_h1's filter gradient 0.87 -> 0.70 ms alone and
            // the STEP 0.1 ms longer in three A/B pairs, round 3: the unsplit launch's idle CUs are not idle in a step; not kept)
            double t = len * t_chunk;
            // slab out + in, + the combine launch.  Weight: in the exact-f32 mode a quarter of the estimate -- measured on whole steps, not
            // launches (round 3): 1, 0.7, 0.5 and 0.25 give the same step while 0.25 takes 0.19 ms off the filter-gradient launches run
            // alone; 2, 4, 8 cost +0.05, +0.15, +0.3 ms of step.  The split-bf16 mode keeps 1 (0.25: +0.1 ms).
            const double slabw = ws.prec ? 1.0 : 0.25;
            if (n > 1) t += slabw * (2.0 * n * nprob * (double)M * N * 4 / 3e12 + 4e-6);
            if (t < best * 0.97) { best = t; nsplit = n; }                        // prefer fewer splits on near-ties
        }
    }
    ep.slab = nsplit > 1 ? ws.slab : nullptr;
    {   // option "trace_launch": one stderr line per distinct launch shape (diagnostics)
        if (opt(OPT_TRACE_LAUNCH)) {
            static std::mutex mu;
            static std::set<std::string> seen;
            char buf[256];
            snprintf(buf, sizeof buf, "igemm %s | M %d N %d nprob %d chunks %d..%d tile %dx%d tiles %lld nsplit %d swz %d/%d", __PRETTY_FUNCTION__ + 0 ? "" : "", M, N, nprob, min_chunks, max_chunks, 64 * MI, 64 * NI,
                     (long long)tiles, nsplit, ep.xcd_swizzle, ep.swz_group);
            std::lock_guard<std::mutex> g(mu);
            if (seen.insert(std::string(buf) + typeid(LA).name()).second) fprintf(stderr, "%s  [%s x %s]\n", buf, typeid(LA).name(), typeid(LB).name());
        }
    }
    {   // problems of unequal length behind the XCD swizzle: balanced runs per XCD (launch.h: balanced_order).  The 8 XCDs share the
        // nsplit x nprob slots in order, so one split's problems are spread over 8 / nsplit of them.  Option "balance".
        if (prob_weight && ep.xcd_swizzle && !ep.swz_group && nprob >= 8) {
            const int nbins = nsplit <= 1 ? 8 : 8 % nsplit == 0 ? 8 / nsplit : 1;
            // (the heavy / light pairing distance only matters for a launch of about one round of resident blocks; a longer one gets
            // the plain alternation, which keeps every window of the slot sequence near the mean)
            const bool one_round = tiles * nsplit <= 3 * (int64_t)dev_info().cus;
            ep.perm = balanced_order(prob_weight, nprob, nbins, one_round ? (int)(tiles / nprob) : 32, s);
        }
    }
    // narrow operands (ContextAEReal's 32-channel layers): tiles that do not multiply zeros.  f32 only.
    if (!ws.prec && N <= 32 && M > 64) launch_tile_f32<LA, LB, 1, 1, 4, 1>(s, a, b, ep, M, N, nprob, nsplit);          // 128 x 32
    else if (!ws.prec && N <= 32 && M <= 32) launch_tile_f32<LA, LB, 1, 1, 1, 1>(s, a, b, ep, M, N, nprob, nsplit);    //  32 x 32
    else if (!ws.prec && M <= 32 && N <= 64) launch_tile_f32<LA, LB, 1, 1, 1, 2>(s, a, b, ep, M, N, nprob, nsplit);    //  32 x 64
    else if (MI == 2 && NI == 2) {
        if (ws.prec) launch_128<LA, LB, WSP, true>(s, a, b, ep, M, N, nprob, nsplit);
        else launch_128<LA, LB, W32, false>(s, a, b, ep, M, N, nprob, nsplit);
    }
    // 128x64 / 64x128: eight 32x32 waves in f32 (-3 % of a step); the split tile needs >= 2 loads per thread, so 4 waves there
    else if (MI == 2 && !ws.prec) launch_tile_f32<LA, LB, 1, 1, 4, 2>(s, a, b, ep, M, N, nprob, nsplit);
    else if (NI == 2 && !ws.prec) launch_tile_f32<LA, LB, 1, 1, 2, 4>(s, a, b, ep, M, N, nprob, nsplit);
    else if (MI == 2) launch_tile<LA, LB, 2, 1, 2, 2>(s, a, b, ep, M, N, nprob, nsplit, ws.prec);
    else if (NI == 2) launch_tile<LA, LB, 1, 2, 2, 2>(s, a, b, ep, M, N, nprob, nsplit, ws.prec);
    else launch_tile<LA, LB, 1, 1, 2, 2>(s, a, b, ep, M, N, nprob, nsplit, ws.prec);
    if (nsplit > 1) splitk_reduce(s, ep, M, N, nprob, nsplit);
}

}  // namespace ctx
