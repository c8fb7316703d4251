// the 3-channel edge layers: cin = 3 conv and the cout = 3 transposed conv as a product + gather
#include "gemm_launch.h"
namespace ctx {
void convt3_product(hipStream_t s, const KmCat2& a, const float* w, int cb, float* P, int M, SplitWs ws) {
    // B[k][n] = w[ky,kx,c,k] with n = (ky*5+kx)*3+c: the filter itself, rows n contiguous in k
    KmPlain b{w, cb, nullptr, 0, cb, 75, cb / KC, a.zeros};
    Epi ep;
    ep.out1 = P; ep.ld1 = P3_LD;
    launch_igemm<KmCat2, KmPlain, false, 1, 2>(s, a, b, ep, M, 75, 1, 0, ws);
}
// the same product for a transposed conv to `ca` output channels: P[pixel][(ky*5+kx)*ca + c] (convt_gather adds the taps of an output pixel)
void convt_product(hipStream_t s, const KmCat2& a, const float* w, int cb, int ca, float* P, int M, SplitWs ws) {
    KmPlain b{w, cb, nullptr, 0, cb, 25 * ca, cb / KC, a.zeros};
    Epi ep;
    ep.out1 = P; ep.ld1 = 25 * ca;
    launch_igemm<KmCat2, KmPlain, false, 1, 2>(s, a, b, ep, M, 25 * ca, 1, cb / KC, ws);
}
void convt3_product_t(hipStream_t s, const KmCat2& b, const float* w, int cb, float* PT, int M, SplitWs ws) {
    // rows = the 75 filter rows (tap, c), columns = pixels: D[t][pixel] = sum_k w[t][k] * cat[pixel][k]
    KmPlain a{w, cb, nullptr, 0, cb, 75, cb / KC, b.zeros};
    Epi ep;
    ep.out1 = PT; ep.ld1 = M;
    launch_igemm<KmPlain, KmCat2, false, 1, 2>(s, a, b, ep, 75, M, 1, 0, ws);
}
void conv3_fwd(hipStream_t s, const KmC3Gather& a, const NmC3Weights& b, Epi ep, int M, int N, SplitWs ws) {
    launch_igemm(s, a, b, ep, M, N, 1, 4, ws);
}
}  // namespace ctx
