// conv2d (arm_shaping.py:21-32) and conv2d_transpose (arm_shaping.py:62-85) as implicit GEMMs.
// The same two kernels also compute each other's input gradient.
#include "gemm_launch.h"
namespace ctx {
// XCD swizzle bits: 1 position-major conv, 2 position-major transposed conv, 4 rectangle-ordered filter gradient.  Workgroups
// are dealt round-robin to the 8 XCDs, each with its own L2; the swizzle gives an XCD a CONTIGUOUS run of work items, so the
// blocks that re-read the same pixels (neighbouring positions, the n-tiles of a position, the 25 taps of a filter gradient)
// share one L2.  Measured fetch bytes per launch at unchanged time: conv 652 -> 416 MB, filter gradient 1354 -> 606 MB,
// transposed conv 851 -> 699 MB.  The transposed conv is swizzled PER PARITY CLASS (Epi::swz_group): its classes differ in
// length, and one contiguous run per XCD over the whole launch (= one class per XCD) cost 20 % time.
int xcd_swz() { static const int v = [] { const char* e = getenv("CTX_XCD_SWIZZLE"); return e ? atoi(e) : 7; }(); return v; }
void conv_fwd(hipStream_t s, const KmConvGather& a, const NmPlain& b_, Epi ep, int M, int N, SplitWs ws) {
    NmPlain b = b_;
    b.seglen = a.tap_outer ? 0 : a.cps * KC;     // filter rows in KmConvGather's K order (A/B measured: no difference)
    b.ntap = a.ntaps();
    launch_igemm<KmConvGather, NmPlain, true, 1, 0>(s, a, b, ep, M, N, 1, a.ntaps() * a.cps, ws);
}
void convt_fwd(hipStream_t s, const KmConvTGather& a, const KmConvTWeights& b, Epi ep, int M, int N, SplitWs ws) {
    ep.rowmode = 1; ep.hs = a.hs; ep.ws = a.ws;
    // the four parity classes have different K extents (4/6/6/9 taps for k 5); each class is split into the same number of
    // parts, and the cost model sees the shortest class (small grids -- the 4x4 and 8x8 layers -- do not fill the chip otherwise)
    const int par = a.pb & 1, tmin = ((a.K - (1 - par) + 1) / 2) * ((a.K - (1 - par) + 1) / 2);
    static const bool sk = [] { const char* e = getenv("CTX_CONVT_SPLITK"); return !(e && e[0] == '0'); }();
    launch_igemm<KmConvTGather, KmConvTWeights, true, 2, 2>(s, a, b, ep, M, N, 4, sk ? tmin * a.cps : 0, ws);
}
void convt3_product(hipStream_t s, const KmCat2& a, const float* w, int cb, float* P, int M, SplitWs ws) {
    // B[k][n] = w[ky,kx,c,k] with n = (ky*5+kx)*3+c: the filter itself, rows n contiguous in k
    KmPlain b{w, cb, nullptr, 0, cb, 75, cb / KC, a.zeros};
    Epi ep;
    ep.out1 = P; ep.ld1 = P3_LD;
    launch_igemm<KmCat2, KmPlain, false, 1, 2>(s, a, b, ep, M, 75, 1, 0, ws);
}
void convt3_product_t(hipStream_t s, const KmCat2& b, const float* w, int cb, float* PT, int M, SplitWs ws) {
    // rows = the 75 filter rows (tap, c), columns = pixels: D[t][pixel] = sum_k w[t][k] * cat[pixel][k]
    KmPlain a{w, cb, nullptr, 0, cb, 75, cb / KC, b.zeros};
    Epi ep;
    ep.out1 = PT; ep.ld1 = M;
    launch_igemm<KmPlain, KmCat2, false, 1, 2>(s, a, b, ep, 75, M, 1, 0, ws);
}
void convt1_fwd(hipStream_t s, const KmConvGather& a, const KmConvTWeights& b, Epi ep, int M, int N, SplitWs ws) {
    launch_igemm(s, a, b, ep, M, N, 1, a.ntaps() * a.cps, ws);
}
void conv_fwd_q(hipStream_t s, const KmConvGatherQ& a, const NmConvWeightsQ& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 4; ep.hs = a.g.hs; ep.ws = a.g.ws; ep.xcd_swizzle = xcd_swz() & ws.swz & 1;
    launch_igemm<KmConvGatherQ, NmConvWeightsQ, true, 1, 0>(s, a, b, ep, a.nimg, N, a.g.hs * a.g.ws, posgeo_min_chunks(a.g), ws);
}
void convt_fwd_q(hipStream_t s, const KmConvTGatherQ& a, const KmConvTWeightsQ& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 5; ep.hs = a.g.hs; ep.ws = a.g.ws; ep.xcd_swizzle = ((xcd_swz() & ws.swz) >> 1) & 1; ep.swz_group = 1;   // grouped by parity class (set to the group size by the launcher)
    const int t = (a.g.K + 1) / 2 - 1;                     // a corner position of the densest class still has this many taps per axis
    launch_igemm<KmConvTGatherQ, KmConvTWeightsQ, true, 2, 2>(s, a, b, ep, a.nimg, N, 4 * a.g.hs * a.g.ws, (t > 0 ? t * t : 1) * a.g.cps, ws);
}
void conv3_fwd(hipStream_t s, const KmC3Gather& a, const NmC3Weights& b, Epi ep, int M, int N, SplitWs ws) {
    launch_igemm(s, a, b, ep, M, N, 1, 4, ws);
}
}  // namespace ctx
