// conv2d (arm_shaping.py:21-32) as an implicit GEMM, image-major and position-major; it is also the input gradient of
// conv2d_transpose.  The transposed conv lives in gemm_convt.hip, the 3-channel layers in gemm_c3.hip.
#include <vector>

#include "gemm_launch.h"
namespace ctx {
// XCD swizzle bits: 1 position-major conv, 2 position-major transposed conv, 4 rectangle-ordered filter gradient.  Workgroups
// are dealt round-robin to the 8 XCDs, each with its own L2; the swizzle gives an XCD a CONTIGUOUS run of work items, so the
// blocks that re-read the same pixels (neighbouring positions, the n-tiles of a position, the 25 taps of a filter gradient)
// share one L2.  Measured fetch bytes per launch at unchanged time: conv 652 -> 416 MB, filter gradient 1354 -> 606 MB,
// transposed conv 851 -> 699 MB.  The transposed conv is swizzled PER PARITY CLASS (Epi::swz_group): its classes differ in
// length, and one contiguous run per XCD over the whole launch (= one class per XCD) cost 20 % time.
int xcd_swz() { return opt(OPT_XCD_SWIZZLE); }
void conv_fwd(hipStream_t s, const KmConvGather& a, const NmPlain& b_, Epi ep, int M, int N, SplitWs ws) {
    NmPlain b = b_;
    b.seglen = a.tap_outer ? 0 : a.cps * KC;     // filter rows in KmConvGather's K order (A/B measured: no difference)
    b.ntap = a.ntaps();
    launch_igemm<KmConvGather, NmPlain, true, 1, 0>(s, a, b, ep, M, N, 1, a.ntaps() * a.cps, ws);
}
void conv_fwd_q(hipStream_t s, const KmConvGatherQ& a, const NmConvWeightsQ& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 4; ep.hs = a.g.hs; ep.ws = a.g.ws; ep.xcd_swizzle = xcd_swz() & ws.swz & 1;
    // taps per position (the problems' lengths): the launcher deals them to the XCDs in balanced runs
    const PosGeo& g = a.g;
    auto nt = [](int si, int nbig, int pad, int K) { const int k0 = pad - si > 0 ? pad - si : 0, k1 = nbig + pad - si < K ? nbig + pad - si : K; return k1 - k0; };
    std::vector<int> wt((size_t)g.hs * g.ws);
    for (int i = 0; i < g.hs; ++i)
        for (int j = 0; j < g.ws; ++j) wt[(size_t)i * g.ws + j] = nt(g.s * i, g.hb, g.pad, g.K) * nt(g.s * j, g.wb, g.px(), g.kw());
    const bool bal = ((balance_bits() & 2) && !(balance_bits() & 8)) || ((balance_bits() & 1) && g.hs * g.ws <= 16);
    if ((balance_bits() & 8) && g.hs * g.ws > 16) ep.perm = morton_order(g.hs, g.ws, s);
    launch_igemm<KmConvGatherQ, NmConvWeightsQ, true, 1, 0>(s, a, b, ep, a.nimg, N, a.g.hs * a.g.ws, posgeo_min_chunks(a.g), ws, 0, bal ? wt.data() : nullptr);
}
}  // namespace ctx
