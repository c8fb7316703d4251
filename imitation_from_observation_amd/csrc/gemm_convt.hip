// conv2d_transpose (arm_shaping.py:62-85) as implicit GEMMs: class-major, position-major and the stride-1 form
// (separate translation unit: the instantiations of gemm_conv.hip and these compile in parallel)
#include "gemm_launch.h"
namespace ctx {
int xcd_swz();
void convt_fwd(hipStream_t s, const KmConvTGather& a, const KmConvTWeights& b, Epi ep, int M, int N, SplitWs ws) {
    ep.rowmode = 1; ep.hs = a.hs; ep.ws = a.ws;
    // the four parity classes have different K extents (4/6/6/9 taps for k 5); each class is split into the same number of
    // parts, and the cost model sees the shortest class (small grids -- the 4x4 and 8x8 layers -- do not fill the chip otherwise)
    const int par = a.pb & 1, tmin = ((a.K - (1 - par) + 1) / 2) * ((a.K - (1 - par) + 1) / 2);
    launch_igemm<KmConvTGather, KmConvTWeights, true, 2, 2>(s, a, b, ep, M, N, 4, tmin * a.cps, ws);
}
void convt_fwd_q(hipStream_t s, const KmConvTGatherQ& a, const KmConvTWeightsQ& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 5; ep.hs = a.g.hs; ep.ws = a.g.ws; ep.xcd_swizzle = ((xcd_swz() & ws.swz) >> 1) & 1; ep.swz_group = 1;   // grouped by parity class (set to the group size by the launcher)
    const int t = (a.g.K + 1) / 2 - 1;                     // a corner position of the densest class still has this many taps per axis
    launch_igemm<KmConvTGatherQ, KmConvTWeightsQ, true, 2, 2>(s, a, b, ep, a.nimg, N, 4 * a.g.hs * a.g.ws, (t > 0 ? t * t : 1) * a.g.cps, ws);
}
void convt1_fwd(hipStream_t s, const KmConvGather& a, const KmConvTWeights& b, Epi ep, int M, int N, SplitWs ws) {
    launch_igemm(s, a, b, ep, M, N, 1, a.ntaps() * a.cps, ws);
}
void convt1_fwd_q(hipStream_t s, const KmConvT1GatherQ& a, const KmConvT1WeightsQ& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 4; ep.hs = a.g.hs; ep.ws = a.g.ws; ep.xcd_swizzle = xcd_swz() & ws.swz & 1;
    launch_igemm<KmConvT1GatherQ, KmConvT1WeightsQ, true, 2, 2>(s, a, b, ep, a.nimg, N, a.g.hs * a.g.ws, posgeo_min_chunks(a.g), ws);
}
}  // namespace ctx
