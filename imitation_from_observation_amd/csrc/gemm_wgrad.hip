// filter gradients of conv2d / conv2d_transpose: 25 taps = 25 problems, reduction over pixels
#include <type_traits>

#include "gemm_launch.h"
namespace ctx {
void conv_wgrad(hipStream_t s, const NmWgradBig& a, const NmWgradSmall& b, Epi ep, int M, int N, SplitWs ws) {
    ep.prob_stride = (int64_t)M * N;
    launch_igemm<NmWgradBig, std::remove_cv_t<std::remove_reference_t<decltype(b)>>, false, 1, 2>(s, a, b, ep, M, N, a.K * a.K, (a.npix + KC - 1) / KC, ws);
}
void conv_wgrad_p(hipStream_t s, const NmWgradBigP& a, const NmWgradSmallP& b, Epi ep, int M, int N, SplitWs ws) {
    ep.prob_stride = (int64_t)M * N;
    launch_igemm<NmWgradBigP, std::remove_cv_t<std::remove_reference_t<decltype(b)>>, false, 1, 2>(s, a, b, ep, M, N, 25, ((a.g.rows_total + a.g.R - 1) / a.g.R) << a.g.ncol_sh, ws);
}
void conv_wgrad2_p(hipStream_t s, const NmWgradBigP& a, const NmWgradSmall2P& b, Epi ep, int M, int N, SplitWs ws) {
    ep.prob_stride = (int64_t)M * N;
    launch_igemm<NmWgradBigP, std::remove_cv_t<std::remove_reference_t<decltype(b)>>, false, 1, 2>(s, a, b, ep, M, N, 25, ((a.g.rows_total + a.g.R - 1) / a.g.R) << a.g.ncol_sh, ws);
}
static int rect_avg_chunks(const RectGeo& g) {
    int64_t t = 0;
    for (int ky = 0; ky < g.K; ++ky)
        for (int kx = 0; kx < g.K; ++kx) t += (int64_t)g.rh[ky] * g.rw[kx] * g.ipc;
    return (int)(t / (g.K * g.K));
}
static int rect_max_chunks(const RectGeo& g) {
    int64_t t = 0;
    for (int ky = 0; ky < g.K; ++ky)
        for (int kx = 0; kx < g.K; ++kx) { const int64_t c = (int64_t)g.rh[ky] * g.rw[kx] * g.ipc; if (c > t) t = c; }
    return (int)t;
}
// positions per tap: the lengths of the 25 problems (launch.h: balanced_order)
static void rect_weights(const RectGeo& g, int wt[25]) {
    for (int ky = 0; ky < g.K; ++ky)
        for (int kx = 0; kx < g.K; ++kx) wt[ky * g.K + kx] = g.rh[ky] * g.rw[kx];
}
int xcd_swz();
void conv_wgrad_r(hipStream_t s, const NmWgradBigR& a, const NmWgradSmallR& b, Epi ep, int M, int N, SplitWs ws) {
    ep.prob_stride = (int64_t)M * N; ep.xcd_swizzle = ((xcd_swz() & ws.swz) >> 2) & 1;
    int wt[25];
    rect_weights(a.g, wt);
    launch_igemm<NmWgradBigR, NmWgradSmallR, false, 1, 2>(s, a, b, ep, M, N, a.g.K * a.g.K, rect_avg_chunks(a.g), ws, rect_max_chunks(a.g), (balance_bits() & 4) ? wt : nullptr);
}
void conv_wgrad2_r(hipStream_t s, const NmWgradBigR& a, const NmWgradSmall2R& b, Epi ep, int M, int N, SplitWs ws) {
    ep.prob_stride = (int64_t)M * N; ep.xcd_swizzle = ((xcd_swz() & ws.swz) >> 2) & 1;
    int wt[25];
    rect_weights(a.g, wt);
    launch_igemm<NmWgradBigR, NmWgradSmall2R, false, 1, 2>(s, a, b, ep, M, N, a.g.K * a.g.K, rect_avg_chunks(a.g), ws, rect_max_chunks(a.g), (balance_bits() & 4) ? wt : nullptr);
}
void conv3_wgrad(hipStream_t s, const NmC3WgradBig& a, const NmWgradSmall& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 2;
    launch_igemm(s, a, b, ep, 100, N, 1, (a.npix + KC - 1) / KC, ws);
}
void conv_wgrad2(hipStream_t s, const NmWgradBig& a, const NmWgradSmall2& b, Epi ep, int M, int N, SplitWs ws) {
    ep.prob_stride = (int64_t)M * N;
    launch_igemm<NmWgradBig, std::remove_cv_t<std::remove_reference_t<decltype(b)>>, false, 1, 2>(s, a, b, ep, M, N, a.K * a.K, (a.npix + KC - 1) / KC, ws);
}
void conv3_wgrad2(hipStream_t s, const NmC3WgradBig& a, const NmWgradSmall2& b, Epi ep, int N, SplitWs ws) {
    ep.rowmode = 2;
    launch_igemm(s, a, b, ep, 100, N, 1, (a.npix + KC - 1) / KC, ws);
}
}  // namespace ctx
