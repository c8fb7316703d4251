// launch.h -- host-callable launchers of every HIP kernel in libctxtrans (all enqueue on `s`, none
// synchronise).  Definitions: gemm_*.hip (implicit-GEMM instantiations) and kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "igemm.h"
#include "dconv.h"

#include "options.h"

namespace ctx {

// A launcher that cannot run its kernel (no tile fits, ...) records the reason here instead of printing or aborting inside a shared
// library; the C ABI entry points turn a pending message into CTX_E_DEVICE when they finish (ctx_engine.cpp: finish / HIP checks).
void set_launch_error(const char* fmt, ...);
bool take_launch_error(char* buf, size_t n);       // true (and the message, cleared) if one is pending on this thread

// The current device's CU count and LDS per CU (hipDeviceProp, cached per device id): what the launchers size persistent grids and
// tiles with.  ensure_dyn_lds raises a kernel's dynamic-LDS limit once per (kernel, device) -- and again if a later launch needs more.
struct DevInfo { int cus; int lds_per_cu; };
const DevInfo& dev_info();
void ensure_dyn_lds(const void* kernel, size_t bytes);
// Load-balanced problem order for launches whose problems differ in length (position-major convolutions: 9 .. 25 taps per position on
// a 4x4 grid; rectangle-ordered filter gradients: 9/16 .. 16/16 of the positions per tap).  With the XCD swizzle an XCD runs a
// CONTIGUOUS run of problem slots and a launch of about one round of resident blocks ends with its slowest XCD: in row-major order
// the 4x4 conv's XCDs get 24 .. 45 taps (mean 36), i.e. the launch runs at 80 %.  balanced_order deals the problems to `nbins` runs
// of (nearly) equal length by longest-first greedy, heavy and light problems placed inside a run so that the two blocks that share
// a CU (blocks l and l + 32 of an XCD's run) are a heavy and a light one, and returns the order as a DEVICE array of nprob uint16 (cached per weight vector and device;
// created on first use -- a synchronous 2 * nprob-byte copy).  Same problems, same arithmetic per problem: results do not change.
// A first use that lands inside a stream capture (a caller capturing its own stream around a handle made by ctx_create_ex) must not
// allocate or copy synchronously -- that would invalidate the capture: it returns nullptr (the kernel's plain order) and the next
// un-captured launch of that shape creates the entry.
const uint16_t* balanced_order(const int* weight, int nprob, int nbins, int tiles_per_problem, hipStream_t stream);
const uint16_t* morton_order(int gh, int gw, hipStream_t stream);      // Z-order walk of a grid of problems (kernels.hip)
// CTX_BALANCE bits (whole-step A/B, profiles/archive/round4_a_ab_balance.txt): 1 = position-major conv on grids of <= 16 positions (the 4x4
// layers: d_h1's input gradient 0.675 -> 0.59 ms, h3_conv forward 0.352 -> 0.316); 2 = on larger grids too (with the pair-aware order
// inside a run: -0.03 ms of step; before it the scattered positions of an 8x8 grid cost more L2 misses than the 7.5 % imbalance);
// 4 = the rectangle-ordered filter gradient's 25 taps (+0.05 ms of step: its launches run several rounds and balance themselves);
// 8 (round 5) = larger grids in Z-order instead of 2: an XCD's run, and the 8-16 problems of it that are resident together, is a 2-D
// compact patch of positions -- 560 -> 431 MB of HBM traffic per launch (row-major runs: 462) at the same step time as the balanced
// order (13.69 / 13.68 against 13.70 / 13.81 ms on one box; row-major 13.91), profiles/round5_d_conv_gather_order.txt.
// Default 9.
int balance_bits();

// Split-K policy shared by all launchers: `slab` is scratch of `slab_floats` floats.
struct SplitWs {
    float* slab;
    int64_t slab_floats;
    int prec = 0;   // CTX_PREC_*: 0 exact-f32 MFMA, 1 split-bf16 (igemm_split.h)
    int swz = 0;    // XCD-swizzle bits the caller allows (gemm_conv.hip: xcd_swz()); measured to pay only on ContextSkipNew's launches
};

// Each launcher computes D = A*B through igemm_kernel with the given loader pair; nprob problems
// share M, N (transposed-conv parity classes, filter-gradient taps).  min_chunks = smallest K-chunk
// count over the problems (bounds the split).
void gemm_fc_fwd(hipStream_t s, const KmPlain& a, const NmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws);
void gemm_fc_dx(hipStream_t s, const KmPlain& a, const KmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws);
void gemm_fc_dw(hipStream_t s, const NmPlain& a, const NmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws);
void gemm_fc_dw2(hipStream_t s, const NmPlain2& a, const NmPlain& b, Epi ep, int M, int N, int nchunks, SplitWs ws);
void conv_fwd(hipStream_t s, const KmConvGather& a, const NmPlain& b, Epi ep, int M, int N, SplitWs ws);
void convt_fwd(hipStream_t s, const KmConvTGather& a, const KmConvTWeights& b, Epi ep, int M, int N, SplitWs ws);
void conv_wgrad(hipStream_t s, const NmWgradBig& a, const NmWgradSmall& b, Epi ep, int M, int N, SplitWs ws);
void conv_wgrad2(hipStream_t s, const NmWgradBig& a, const NmWgradSmall2& b, Epi ep, int M, int N, SplitWs ws);
// power-of-two grids: patch-ordered K (see igemm.h PatchGeo)
// rectangle-ordered K (nimg % 32 == 0): no SAME-padding zeros are multiplied
void conv_wgrad_r(hipStream_t s, const NmWgradBigR& a, const NmWgradSmallR& b, Epi ep, int M, int N, SplitWs ws);
void conv_wgrad2_r(hipStream_t s, const NmWgradBigR& a, const NmWgradSmall2R& b, Epi ep, int M, int N, SplitWs ws);
void conv_wgrad_p(hipStream_t s, const NmWgradBigP& a, const NmWgradSmallP& b, Epi ep, int M, int N, SplitWs ws);
void conv_wgrad2_p(hipStream_t s, const NmWgradBigP& a, const NmWgradSmall2P& b, Epi ep, int M, int N, SplitWs ws);
// stride-1 conv2d_transpose as a flipped stride-1 correlation (a.flip = 1, b.flip25 = 1); filter rows as B
void convt1_fwd(hipStream_t s, const KmConvGather& a, const KmConvTWeights& b, Epi ep, int M, int N, SplitWs ws);
void convt1_fwd_q(hipStream_t s, const KmConvT1GatherQ& a, const KmConvT1WeightsQ& b, Epi ep, int N, SplitWs ws);
// position-major variants (one problem per output position, rows = images): only the taps that land inside the grid
void conv_fwd_q(hipStream_t s, const KmConvGatherQ& a, const NmConvWeightsQ& b, Epi ep, int N, SplitWs ws);
void convt_fwd_q(hipStream_t s, const KmConvTGatherQ& a, const KmConvTWeightsQ& b, Epi ep, int N, SplitWs ws);
void conv3_fwd(hipStream_t s, const KmC3Gather& a, const NmC3Weights& b, Epi ep, int M, int N, SplitWs ws);
void conv3_wgrad(hipStream_t s, const NmC3WgradBig& a, const NmWgradSmall& b, Epi ep, int N, SplitWs ws);
void conv3_wgrad2(hipStream_t s, const NmC3WgradBig& a, const NmWgradSmall2& b, Epi ep, int N, SplitWs ws);

// narrow-channel direct convolutions (dconv.h / dconv.hip).  P carries tensors, channel counts, filter and epilogue; the
// launchers fill the tap tables, tiles and LDS split.  dconv_ok: the channel counts the kernels are instantiated for.
bool dconv_ok(int CI, int N);
constexpr int64_t DC_WPACK_FLOATS = 40ll * 64 * 128 + 64;   // P.wp: <= 37 class-padded tap slots x 64 k x 128 n, then their tile offsets
void dconv_conv(hipStream_t s, DcFwd P, int stride, int pad);        // conv2d 5x5 SAME (also: input gradient of conv2d_transpose)
void dconv_conv_k(hipStream_t s, DcFwd P, int kh, int kw, int stride, int pady, int padx, int hout, int wout);   // conv2d kh x kw, given output grid (SAME or VALID)
void dconv_convt1(hipStream_t s, DcFwd P);                           // conv2d_transpose 5x5 stride 1 (also: input gradient of a stride-1 conv2d)
void dconv_convt2(hipStream_t s, DcFwd P);                           // conv2d_transpose 5x5 stride 2 (also: input gradient of a stride-2 conv2d)
void dconv_wgrad(hipStream_t s, DcWgrad P, float* slab, int64_t slab_floats);   // filter gradient of either, 25 taps in one launch

// the fixed-order sum of nslab partials [M][NP] into out[m][n0 + n] (row stride CB), columns < ncols
void dconv_wgrad_reduce(hipStream_t s, const float* slab, int nslab, int M, int NP, int ncols, int n0, int CB, float* out);
// filter gradient with a 3-channel big-grid side on whole 128-pixel tiles (c3wgrad.hip); dconv_wgrad routes to it when c3wgrad_ok
bool c3wgrad_ok(const DcWgrad& P);
void c3wgrad(hipStream_t s, const DcWgrad& P, float* slab, int64_t slab_floats);

// ContextAEReal's fully connected middle (h4_lin, hz_lin, trans_h0, trans_z, d_h0_lin) in three launches (rchain.hip): rows of one
// triple stay in one block.  W[10] = {W4, b4, Wz, bz, Wt0, bt0, Wtz, btz, Wd0, bd0} (padded layouts of the arena); G likewise.
bool rchain_ok(int Fp, int64_t D0p, int nset, bool drop, int prec);
void rchain_fwd(hipStream_t s, int B, int D0p, const float* a3, float* a4, float* Z, float* th0, float* dz, const float* const W[10]);
void rchain_bwd(hipStream_t s, int B, int D0p, const float* dDz, const float* dsim2, float* dZ, float* dth0, float* dA4, float* dA3, const float* Z,
                const float* th0, const float* a4, const float* a3, const float* dSk3, const float* const W[10]);
void rchain_dw(hipStream_t s, int B, int D0p, const float* a3, const float* a4, const float* Z, const float* th0, const float* dA4, const float* dZ,
               const float* dth0, const float* dDz, float* const G[10]);

// conv2d 5x5 (stride 1 | 2, SAME) from a 3-channel tensor to N = 32 | 64 | 128 channels (c3conv.hip): h0_conv forward and the input
// gradient of d_h4 in both models; epilogue = bias / lrelu / lrelu' mask / column split.  c3conv_ok: the shapes it is built for.
bool c3conv_ok(int hin, int win, int stride, int N, const Epi& ep);
void c3conv(hipStream_t s, const float* x, int nimg, int hin, int win, int stride, const float* w, int N, const Epi& ep);

// conv2d_transpose 5x5 stride 2 for wide channel counts on the 8x8 / 16x16 grids: image-major, input halo tile resident in LDS
// (wconvt.hip).  in = [s1 | s2] (s2 = ctx skip with image index img % nmod2; c2 = 0: none), filter w[5][5][ca][c1 + c2].
bool wconvt_ok(int hs, int ws, int c1, int c2, int ca, int nimg);
void wconvt_fwd(hipStream_t s, const float* s1, int c1, const float* s2, int c2, int nmod2, int nimg, int hs, int ws, const float* w, int ca,
                const Epi& ep, SplitWs ws_);

// conv2d_transpose to 3 output channels (d_h4, arm_shaping.py:1329-1330) in two steps: the scatter
// product P[pixel][(ky,kx,c)] = sum_k in[pixel][k] * w[ky,kx,c,k] as an MFMA GEMM (N = 75), then a
// gather of the <= 9 taps that land on each output pixel (deterministic, no atomics).
// The same layer in ONE pass on the vector ALUs (convt3.hip): input halo tile in LDS by 8-channel slices, a lane owns a column of
// output rows; nothing between input and output.  The exact-f32 path of every 3-channel d_h4; the product + gather pair below
// remains for the split-bf16 mode and for shapes convt3_direct_ok refuses.
bool convt3_direct_ok(int c1, int c2, int hin, int win, int stride);
void convt3_direct(hipStream_t s, const float* x1, int c1, const float* x2, int c2, int nmod2, int nimg, int hin, int win, int stride,
                   const float* w, const float* bias, float* out);
// The same layer on the MATRIX cores (convt3m.hip): P[pixel][(ky)(kx, c)] as 16-column MFMA blocks per filter row, A operand straight
// from global memory, P kept in an LDS ring and gathered into output rows inside the block -- one pass over the input, for launches of
// >= 128 images (training batches).  convt3_direct dispatches to it where convt3_mfma_ok (option direct3 bit 16).
bool convt3_mfma_ok(int c1, int c2, int hin, int win, int stride, int nimg);
void convt3_mfma(hipStream_t s, const float* x1, int c1, const float* x2, int nmod2, int nimg, int hin, int win, int stride,
                 const float* w, const float* bias, float* out);
constexpr int P3_LD = 80;                   // row stride of P (75 used)
constexpr int PP_IMG = 32;                  // most images of a launch that takes the product + gather route of convt_product
void convt3_product(hipStream_t s, const KmCat2& a, const float* w, int cb, float* P, int M, SplitWs ws);
// conv2d_transpose 5x5 s2 as ONE plain product + a gather, for STARVED inference launches (the reward hook's 25 frames: a handful of
// images cannot fill 256 CUs with image-major tiles, a [pixels x cb] x [cb x 25 ca] product can): P[pixel][(ky*5+kx)*ca + c], then
// out[n, y, x, c] = act(b[c] + the 4-9 taps of (y, x)) in a fixed order.  All 25 taps of every input pixel are formed.
void convt_product(hipStream_t s, const KmCat2& a, const float* w, int cb, int ca, float* P, int M, SplitWs ws);
void convt_gather(hipStream_t s, const float* P, const float* bias, float* out, int nimg, int hs, int ws, int ca, int lrelu);
void convt3_gather(hipStream_t s, const float* P, const float* bias, float* out, int nimg, int hs, int ws);
// stride-1 variant: out[n,y,x,c] = b[c] + sum_{ky,kx} P[(n, y+2-ky, x+2-kx)][(ky*5+kx)*3+c]
// tap-major variant: PT[(tap*3+c)][M pixels] = w (75 x cb) times the concat input, then a coalesced gather
void convt3_product_t(hipStream_t s, const KmCat2& b, const float* w, int cb, float* PT, int M, SplitWs ws);
void convt3_gather_s1_t(hipStream_t s, const float* PT, const float* bias, float* out, int nimg, int hs, int ws);

// (x * 1/255 - 0.5) * 2 in unfused f32 ops (rllab/sampler/base.py:116-119)
void u8_to_f32(hipStream_t s, const uint8_t* in, float* out, int64_t n);
void pack3to4(hipStream_t s, const float* in3, float* out4, int64_t npix);   // the 4-channel copy the cin = 3 loaders read
// Inception front end: 3-channel frames into a channel-padded buffer; 3x3 max (stride 2, VALID) / avg (stride 1, SAME) pooling
void pad_channels_u8(hipStream_t s, const uint8_t* in, float* out, int64_t npix, int cpad);
void pad_channels_f32(hipStream_t s, const float* in, float* out, int64_t npix, int cpad);
void maxpool3x3s2(hipStream_t s, const float* in, float* out, int nimg, int hi, int wi, int c, int ldo);
void avgpool3x3s1(hipStream_t s, const float* in, float* out, int nimg, int hi, int wi, int c, int ldo);

// device batch sampler (scripts/train_script.py:153-159); lut[256] = f32(x / 127.5 - 1)
void gather_triples(hipStream_t s, const uint8_t* vdata, int T, int N, int64_t npi, const int* csrc, const int* ctgt, int B, int b0,
                    const float* lut, float* img);

// Losses (arm_shaping.py:1345,1352-1354) and their seeds of the backward pass.
//   out [2B, npi] (rows < B: translated pass, rows >= B: truth pass), tgt [B, npi]
//   dout (nullable) = out - tgt[n % B]
//   tz, tgt_z [B, F]; dsim2 (nullable) [2B, F] = [+c (tz - tgt_z); -c (tz - tgt_z)], c = 2e3/(sim_batch F)
//   scalars[4] = {loss, simloss, recon1, recon2};  scratch: >= 4 * LOSS_BLOCKS floats
constexpr int LOSS_BLOCKS = 512;
//   terms: CTX_LOSS_* mask of the terms that make up `loss` (ablations_code/ablations.py:175-182): an excluded term is still
//   reported in scalars[1..3] but contributes neither to scalars[0] nor to the backward seeds
void losses(hipStream_t s, const float* out, const float* tgt, float* dout, int64_t npi, int B, const float* tz,
            const float* tgt_z, float* dsim2, int F, int sim_batch, float* scratch, float* scalars, int F_real = 0, int terms = 7);

// per-frame reward cost (rllab/sampler/base.py:243-249): costs[j] = |means[j % bs] - feat[j]|^2 + scale * |imgs[j % bs] - x[j]|^2
// (ablation 1: image term only, 2: feature term only)
void reward_costs(hipStream_t s, const float* feat, int ldf, int F, const float* x, int64_t npi, const float* means, const float* imgs,
                  int bs, int nframes, float scale, int ablation, float* costs);

// db[c] = sum_rows x[row][c], deterministic two-stage; scratch >= COLSUM_SPLITS * max(C, 4) floats
constexpr int COLSUM_SPLITS = 512;
void colsum(hipStream_t s, const float* x, int64_t rows, int C, float* scratch, float* out);

// tf.nn.dropout of ContextAEReal's training graph (kernels.hip): factors mask / keep_prob from a hash of (seed, step, site, element),
// an elementwise product with own row strides, and  out = (raw * M + add1 + add2) * lrelu'(act)  (add1 / add2 / act nullable)
void drop_factors(hipStream_t s, float* M, int rows, int ld, int gp, int gr, int ncols, float keep_prob, uint32_t seed, uint32_t step, int site);
void ew_mul(hipStream_t s, float* out, int ldo, const float* x, int ldx, const float* M, int ldm, int rows, int cols);
void drop_fin(hipStream_t s, float* out, const float* raw, const float* M, const float* add1, const float* add2, const float* act, int64_t n);

// g *= (act >= 0 ? 1 : 0.2)
void lrelu_mask(hipStream_t s, float* g, const float* act, int64_t n);

// TF Adam (scripts/train_script.py:124-128): lr_t precomputed on the host
void adam(hipStream_t s, float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
          float eps);

}  // namespace ctx
