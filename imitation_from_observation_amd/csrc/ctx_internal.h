// ctx_internal.h -- what the translation units of libctxtrans.so share: the handle (HBM layout below), the table-driven engine's state,
// and the entry points of the launch sequences (ctx_engine.cpp) that the C ABI (ctx_abi.cpp) and the RCCL client (ctx_dp.cpp) call.
// Nothing here is part of the public interface (include/ctxtrans.h).
//
// Batching (what makes each filter gradient a single launch): the `conv` encoder runs once on the
// stacked [tgt | src] frames (2B), the decoder once on the stacked [translated | truth] codes (2B)
// with the ctx skips indexed img % B; `conv_context` runs on B.
//
// HBM layout per handle
//   arena   [params | grads | adam_m | adam_v], each Ppad floats (P rounded up to 64)
//   img     [tgt | src | ctx] f32 frames, 3B x H x W x 3
//   Z       [trans_z | tgt_z | src_z] codes, 3B x F  -> decoder input = first 2B rows,
//           `conv` encoder output = last 2B rows, no copies
//   dZ      same rows for the code gradients
//   one buffer per activation and per activation gradient (NHWC), sized for max_batch
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library itself is dlopen()ed by ctx_dp_init (no link-time dependency)

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../include/ctxtrans.h"
#include "launch.h"

using namespace ctx;

namespace ctxi {

extern thread_local std::string g_create_error;

struct ParamInfo {
    std::string name;
    int ndim;
    int64_t shape[4];
    int64_t offset;
    int64_t size;
};

struct GenState {
    // ---- architecture
    int Kk[4] = {5, 5, 5, 5};    // kernel size (k x k) of encoder layer h0..h3; d_h1..d_h4 mirror them (k4, k3, k2, k1)
    int S[4] = {1, 2, 1, 2};     // strides of h0..h3; d_h1..d_h4 mirror them (s4, s3, s2, s1)
    int nf[4] = {};              // filter counts as in the reference
    int cp[4] = {};              // the same rounded up to 32: channel count of the stored activations
    int C0 = 3;                  // frame channels: 3 (cin = 3 kernels) or a multiple of 32
    int nset = 1;                // 1: one encoder for src, tgt, ctx; 2: `conv` (set 0) + `conv_context` (set 1)
    bool residual = false;       // out = decode(.) + ctx frame
    bool narrow = false;         // ContextAEReal in f32: activations and filters at their REAL widths (32/16/16/8), every conv / deconv and
                                 // filter gradient on the direct kernels of dconv.h (no zero padding to 32 channels anywhere but the codes)
    // ---- derived geometry
    int Fp = 0;                  // padded featsize
    int gh[4], gw[4];            // grid after encoder layer k
    int se[4], pd[4];            // effective stride and SAME pad_before of encoder layer k
    int64_t D0p = 0;             // padded flatten: gh[3]*gw[3]*cp[3]
    // ---- TF-shaped flat vector <-> padded arena
    struct Seg { int64_t roff, poff, size, map0; };   // map0 < 0: contiguous copy; else real2pad[map0 + i]
    std::vector<Seg> segs;
    std::vector<int32_t> real2pad;
    // padded parameter offsets (floats into the arena), per encoder set
    int64_t w[2][4], b[2][4], w4[2], b4[2], wz[2], bz[2], th0w, th0b, tzw, tzb, d0w, d0b, dw[5], db[5];
    // ---- activations (NHWC, cp[k] channels) and their gradients
    float *a[5] = {}, *dA[5] = {};           // a[0..3] conv outputs over 3B images [tgt | src | ctx], a[4] = h4 [3B, Fp]
    float *th0 = nullptr, *dth0 = nullptr, *dz = nullptr, *dDz = nullptr;
    float *e[4] = {}, *dE[4] = {};           // e[1..3] decoder outputs over 2B
    float* dSk[4] = {};                      // d loss / d skip h_k, both decoder passes (2B)
    float* dsim2 = nullptr;
    // tf.nn.dropout of ContextAEReal's training graph (keep_prob < 1 only; arm_shaping.py:1637-1661): per site the factors
    // mask / keep_prob (dM[1..6]: sites 1 flatten(h3), 2 h4, 3 concat([src_z, ctx_z]), 4 trans_h0, 5 z, 6 reshape(z_)) and the dropped
    // copy the next layer reads
    float* dM[7] = {};
    float *x1d = nullptr, *x2d = nullptr, *xcat = nullptr, *th0d = nullptr, *zd = nullptr, *dzd = nullptr, *graw = nullptr;
    int in_grid_h(int k, int H) const { return k ? gh[k - 1] : H; }
    int in_grid_w(int k, int W) const { return k ? gw[k - 1] : W; }
    int in_ch(int k) const { return k ? cp[k - 1] : C0; }
};

}  // namespace ctxi
using ctxi::GenState;
using ctxi::ParamInfo;

struct ctx_handle {
    ctx_config cfg{};
    Options opt{};               // this handle's switches (options.h; ctx_set_option): the process defaults (environment) at ctx_create
    GenState* gen = nullptr;     // CTX_VARIANT_REAL / CTX_VARIANT_INCEPTION2 state (ctxtrans_gen.inc)
    int Fp = 0;                  // row stride of the code buffers Z / dZ (featsize, or featsize padded to 32 for REAL)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float* arena = nullptr;
    bool own_arena = false;
    int64_t P = 0, Ppad = 0;
    std::vector<ParamInfo> params;
    std::vector<void*> allocs;
    std::string err;
    int64_t adam_t = 0;
    int last_B = 0;
    bool have_grads = false;

    // dims
    int H, W, d, F, Bm;
    int hh[5], ww[5];   // hh[k] = H >> k
    int64_t npi;        // H*W*3
    int64_t D0;         // d_h0_lin width = 8d * h16 * w16

    // buffers (see header comment)
    uint8_t* u8 = nullptr;
    float *img = nullptr, *Z = nullptr, *dZ = nullptr;
    float *img4 = nullptr, *dout4 = nullptr;   // 4-channel copies of img / dout for the cin = 3 loaders (C == 3 only)
    float *s[5] = {}, *c[5] = {}, *cz = nullptr, *th0 = nullptr;     // s[k], c[k]: h0..h3 conv outputs, [4] = h4
    float *dz = nullptr, *e[4] = {}, *out = nullptr;                 // e[1..3] decoder activations
    float *dout = nullptr, *dE[4] = {}, *dSk[4] = {}, *dDz = nullptr, *dsim2 = nullptr;
    float *dth0 = nullptr, *dcz = nullptr, *dS[5] = {}, *dC[5] = {};
    float *scratch = nullptr, *slab = nullptr, *scalars = nullptr;
    float* zeros = nullptr;   // 256 B of zeros for the branch-free loaders
    // second lane: the conv_context encoder (forward and backward) is independent of the `conv` encoder chain
    // and runs on its own stream with its own split-K slab / reduction scratch, so its half-size launches fill
    // the tails of the other chain's launches
    // lane 0 = conv_context chain; lane 1 = filter / bias gradients (off the backward critical path: only the
    // input gradients feed the next layer)
    static constexpr int NLANE = 2;
    hipStream_t aux[NLANE] = {};
    hipEvent_t ev_fork[NLANE] = {}, ev_join[NLANE] = {};
    float *slabL[NLANE] = {}, *scratchL[NLANE] = {};
    float *wpack = nullptr, *wpackL[NLANE] = {};   // dconv's re-packed filters, one buffer per stream lane (concurrent launches)
    bool overlap = true;
    // hipGraph cache of the two inference forwards (reward hook: batch-25 calls are launch-bound): key = mode * 2^20 + B
    struct GraphSlot { int calls = 0; hipGraphExec_t exec = nullptr; uint64_t pack_version = 0; bool self_packing = false; };
    DcPackCache pack;                // packed filters of the direct kernels, valid for pack.version (dconv.h); bumped wherever parameters change
    bool ctx_single = false;         // MODE_TRANSLATE with ONE context frame for the whole batch (`[context] * batch_size`, base.py:217-218):
                                     // `conv_context` runs on that one frame and its outputs are read by every row (forward)
    std::map<int, GraphSlot> graphs;
    bool use_graphs = true, capturing = false;
    // data-parallel overlap: called from inside backward once the translate/* and deconv/* gradients are complete in the
    // handle's stream order, so the caller can start their all-reduce while the encoders' backward is still being enqueued
    ctx_bucket_fn bucket_fn = nullptr;
    void* bucket_user = nullptr;
    // resident demo tensor (ctx_demos_upload): uint8 vdata[T][N][H*W*3], the x/127.5-1 table, index staging
    uint8_t* vdata = nullptr;
    int vT = 0, vN = 0;
    float* lut = nullptr;
    int* choice = nullptr;    // [2 * max_batch]: choicesrc | choicetgt
    // reward hook on the device (ctx_reward_*): per viewpoint the cached demo means [bs, F] and mean translated frames [bs, H, W, 3]
    struct RewardCache { float* means = nullptr; float* imgs = nullptr; int bs = 0; };
    std::vector<RewardCache> rcache;
    float* rcosts = nullptr;
    float* P3 = nullptr;   // d_h4 scatter product [2B * H/2 * W/2][P3_LD]
    float* PP = nullptr;   // transposed-conv product of the starved inference launches (<= PP_IMG images): [images * hs * ws][25 ca], largest layer
    int64_t slab_floats = 0;
    // data parallel over RCCL (ctx_dp_*): communicator, a stream for the collectives (they overlap the encoders' backward),
    // the events that order it with the compute stream, a device buffer for the global scalars
    ncclComm_t dp_comm = nullptr;
    int dp_rank = 0, dp_world = 1;
    hipStream_t dp_stream = nullptr;
    hipEvent_t dp_ev_ready = nullptr, dp_ev_done = nullptr;
    float* dp_scal = nullptr;
    double* dp_host_buf = nullptr;    // device staging of ctx_dp_allreduce_host_f64, grown on demand (not per call)
    size_t dp_host_cap = 0;
    bool dp_in_step = false;      // inside ctx_dp_train_step: fire_bucket starts the tail bucket's all-reduce itself
    int64_t dp_split = -1;        // first float of the tail bucket once it has been started in this step
    int dp_rc = 0;                // result of the collectives started from inside backward
    std::vector<std::pair<int64_t, int64_t>> dp_done;   // [first, end) of the HEAD of the arena already sent in this step (the encoders' FC slices)
    // Adam beside the backward (fused training steps only: adam_begin / adam_early / adam_end): a slice of the arena is updated on
    // its own stream as soon as its gradients are final and its parameters have been read for the last time in this step
    hipStream_t adam_stream = nullptr;
    hipEvent_t adam_ev[2] = {}, adam_ev_done = nullptr;
    bool adam_early_on = false;
    float adam_lr_t = 0.f;
    std::vector<std::pair<int64_t, int64_t>> adam_done;   // [first, end) slices already enqueued in this step
    // tf.nn.dropout (CTX_VARIANT_REAL with keep_prob < 1): on only while a TRAINING step (forward + backward) is being enqueued
    bool drop_on = false;
    uint64_t drop_seed = 0;

    // per-op profiling (ctx_profile_step): HIP events around every launch group
    bool prof_on = false;
    int prof_cursor = 0;
    std::vector<hipEvent_t> prof_ev;
    std::vector<ctx_prof_entry> prof_entries;
    std::vector<double> prof_ms;

    float* Wp(const char* name) const { return arena + find(name); }
    float* Gp(const char* name) const { return arena + Ppad + find(name); }
    int64_t find(const char* name) const {
        for (auto& p : params)
            if (p.name == name) return p.offset;
        return -1;
    }
};

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return ctxi::fail(h, CTX_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define TRY(expr)              \
    do {                       \
        int r_ = (expr);       \
        if (r_ != CTX_OK) return r_; \
    } while (0)

namespace ctxi {
enum Mode { MODE_TRAIN, MODE_TRANSLATE, MODE_ENCODE };
constexpr int LANE_CTX = 0, LANE_DW = 1;
int fail(ctx_handle* h, int code, const char* fmt, ...);
int check_B(ctx_handle* h, int B);
int finish(ctx_handle* h);
int copy_d2h(ctx_handle* h, void* dst, const void* src, size_t bytes);
int forward_inference(ctx_handle* h, int B, Mode mode);
void forward(ctx_handle* h, int B, Mode mode);
int fused_step(ctx_handle* h, int B, float lr);
int loss_terms_of(const ctx_handle* h);
int gen_layout(const ctx_config& c, GenState& r, std::vector<ParamInfo>& real, int64_t& P, int64_t& Ppadded);
int check_cfg(const ctx_config* c, ctx_handle* h);
int upload_f32(ctx_handle* h, const float* src, const float* ctxf, const float* tgt, int B);
int64_t round_up(int64_t x, int64_t m);
void build_params(const ctx_config& c, std::vector<ParamInfo>& out, int64_t& total);
void backward(ctx_handle* h, int B, int sim_batch);
int adam_step(ctx_handle* h, float lr);
int gen_alloc(ctx_handle* h);
int alloc_buffers(ctx_handle* h);
bool use_lanes(const ctx_handle* h);
void adam_begin(ctx_handle* h, float lr);
void adam_end(ctx_handle* h);
bool d_h4_direct(const ctx_handle* h, int c1, int c2, int hs, int ws, int stride);
void join(ctx_handle* h, int lane);
void fork(ctx_handle* h, int lane);
void fire_bucket(ctx_handle* h, int64_t first);
int dp_reduce_range(ctx_handle* h, int64_t first, int64_t count);    // ctx_dp.cpp
void dp_teardown(ctx_handle* h);                                      // ctx_dp.cpp
int stage_frames(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B);   // ctx_abi.cpp

template <class T>
int dev_alloc(ctx_handle* h, T** p, int64_t count, bool whole_tensor = true) {
    // the loaders address a tensor with 32-bit byte offsets (buffer descriptors, 0x80000000 = out-of-range marker)
    if (whole_tensor && count * (int64_t)sizeof(T) >= (1ll << 31))
        return fail(h, CTX_E_INVALID, "a %lld-byte activation buffer exceeds the 2 GiB the kernels address per tensor: lower max_batch",
                    (long long)(count * sizeof(T)));
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (size_t)count * sizeof(T));
    if (e != hipSuccess) return fail(h, CTX_E_NOMEM, "hipMalloc(%lld bytes): %s", (long long)(count * sizeof(T)), hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T*)q;
    // debugging aid: CTX_DEBUG_POISON=1 fills every fresh buffer with 0xFF bytes (float NaN) so that a kernel reading memory nothing has
    // written shows up as NaN on every run instead of as a rare mismatch that depends on what the allocator handed back
    static const bool poison = getenv("CTX_DEBUG_POISON") && atoi(getenv("CTX_DEBUG_POISON"));
    if (poison) {      // (the fill runs on the null stream, which the handle's non-blocking streams do not wait for: finish it here)
        (void)hipMemset(q, 0xFF, (size_t)count * sizeof(T));
        (void)hipDeviceSynchronize();
    }
    return CTX_OK;
}

}  // namespace ctxi
