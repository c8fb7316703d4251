// linear (arm_shaping.py:48-59): forward, input gradient, Matrix gradient
#include "gemm_launch.h"
namespace ctx {
void gemm_fc_fwd(hipStream_t s, const KmPlain& a, const NmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws) {
    launch_igemm<KmPlain, NmPlain, false, 2, 2>(s, a, b, ep, M, N, 1, nchunks, ws);
}
void gemm_fc_dx(hipStream_t s, const KmPlain& a, const KmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws) {
    launch_igemm<KmPlain, KmPlain, false, 2, 2>(s, a, b, ep, M, N, 1, nchunks, ws);
}
void gemm_fc_dw(hipStream_t s, const NmPlain& a, const NmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws) {
    launch_igemm<NmPlain, NmPlain, false, 2, 2>(s, a, b, ep, M, N, 1, nchunks, ws);
}
void gemm_fc_dw2(hipStream_t s, const NmPlain2& a, const NmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws) {
    launch_igemm<NmPlain2, NmPlain, false, 2, 2>(s, a, b, ep, M, N, 1, nchunks, ws);
}
}  // namespace ctx
