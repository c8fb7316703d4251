// ctxtrans.cpp -- host side of libctxtrans.so: handle, HBM layout, the forward / backward / Adam
// launch sequence of ContextSkipNew (gym/envs/mujoco/arm_shaping.py:1272-1354) and the C ABI of
// include/ctxtrans.h.  There is no CPU code path: every contraction runs in the HIP kernels of
// igemm.h / kernels.hip; without a device ctx_create fails.
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

namespace {

thread_local std::string g_create_error;

struct ParamInfo {
    std::string name;
    int ndim;
    int64_t shape[4];
    int64_t offset;
    int64_t size;
};

}  // namespace

namespace { struct GenState; }

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

namespace {

int fail(ctx_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

// RAII timer of one launch group; a no-op unless ctx_profile_step is running
struct ProfScope {
    ctx_handle* h;
    int idx = -1;
    // useful: share of `flops` whose product meets two data operands (tap_frac for SAME-padded convolutions; 1 elsewhere)
    ProfScope(ctx_handle* h_, const std::string& name, const char* kernel, double flops, double useful = 1.0) : h(h_) {
        if (!h->prof_on) return;
        idx = h->prof_cursor++;
        if ((int)h->prof_entries.size() <= idx) {
            ctx_prof_entry e{};
            snprintf(e.name, sizeof e.name, "%s", name.c_str());
            snprintf(e.kernel, sizeof e.kernel, "%s", kernel);
            e.flops = flops;
            e.useful_frac = (float)useful;
            h->prof_entries.push_back(e);
            h->prof_ms.push_back(0.0);
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            h->prof_ev.push_back(a);
            h->prof_ev.push_back(b);
        }
        (void)hipEventRecord(h->prof_ev[2 * idx], h->stream);
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(h->prof_ev[2 * idx + 1], h->stream);
    }
};

// Share of a SAME-padded K x K stride-s layer's (position, tap) pairs whose tap lies INSIDE the image: the products of the conv, of
// its transposed conv and of its filter gradient that multiply data and not padding zeros.  n_big = the layer's large grid (conv
// input = transposed-conv output); TF's rule: out = ceil(n / s), pad_total = max((out - 1) s + K - n, 0), before = total / 2.
// 5x5 stride 2 on an even grid: (5 n_small - 3) / (5 n_small) per axis -- 92.6 / 85.6 / 72.3 % in 2-D on 16x16 / 8x8 / 4x4 grids.
double tap_frac1(int n_big, int K, int s) {
    const int n_small = (n_big + s - 1) / s;
    const int total = std::max((n_small - 1) * s + K - n_big, 0), before = total / 2;
    int64_t valid = 0;
    for (int i = 0; i < n_small; ++i)
        for (int k = 0; k < K; ++k) {
            const int y = s * i + k - before;
            valid += y >= 0 && y < n_big;
        }
    return (double)valid / ((double)K * n_small);
}
double tap_frac(int hb, int wb, int K, int s) { return tap_frac1(hb, K, s) * tap_frac1(wb, K, s); }
// the same count for a layer given by its own (stride, pad_before) -- the table-driven models' parameterisation
double tap_frac_p1(int n_big, int n_small, int K, int s, int pad) {
    int64_t valid = 0;
    for (int i = 0; i < n_small; ++i)
        for (int k = 0; k < K; ++k) {
            const int y = s * i + k - pad;
            valid += y >= 0 && y < n_big;
        }
    return (double)valid / ((double)K * n_small);
}
double tap_frac_p(int hb, int wb, int hs, int ws, int K, int s, int pad) { return tap_frac_p1(hb, hs, K, s, pad) * tap_frac_p1(wb, ws, K, s, pad); }

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(h, CTX_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int check_cfg(const ctx_config* c, ctx_handle* h) {
    if (!c) return fail(h, CTX_E_INVALID, "cfg is NULL");
    if (c->variant != CTX_VARIANT_SKIPNEW && c->variant != CTX_VARIANT_REAL && c->variant != CTX_VARIANT_INCEPTION2)
        return fail(h, CTX_E_INVALID, "unsupported variant %d", c->variant);
    if (c->loss_terms < 0 || c->loss_terms > 7) return fail(h, CTX_E_INVALID, "loss_terms must be a mask of CTX_LOSS_RECON1 | CTX_LOSS_RECON2 | CTX_LOSS_SIM (0 = all)");
    if (!(c->keep_prob >= 0.f && c->keep_prob <= 1.f)) return fail(h, CTX_E_INVALID, "keep_prob must lie in [0, 1] (0 or 1: no dropout)");
    if (c->keep_prob > 0.f && c->keep_prob < 1.f && c->variant != CTX_VARIANT_REAL)
        return fail(h, CTX_E_INVALID, "keep_prob: only ContextAEReal has dropout in its graph (arm_shaping.py:1637-1661)");
    if (c->variant == CTX_VARIANT_INCEPTION2) {   // feature maps [h, w, C]; ContextAEInception2(strides, kernels, filters)
        if (c->C <= 0 || c->C % 32) return fail(h, CTX_E_INVALID, "C (feature channels) must be a positive multiple of 32");
        bool any_f = false, all_f = true;
        for (int k = 0; k < 4; ++k) { any_f = any_f || c->filters[k]; all_f = all_f && c->filters[k]; }
        if (any_f != all_f) return fail(h, CTX_E_INVALID, "filters: give all four counts or none");
        if (!all_f && (c->df_dim <= 0 || c->df_dim % 4)) return fail(h, CTX_E_INVALID, "df_dim must be a multiple of 4 (default filters 16d/16d/8d/8d)");
        for (int k = 0; k < 4; ++k) {
            if (c->filters[k] < 0 || c->filters[k] % 32) return fail(h, CTX_E_INVALID, "filters[%d] = %d: filter counts must be multiples of 32", k, c->filters[k]);
            if (c->kernels[k] < 0 || c->kernels[k] > 5) return fail(h, CTX_E_INVALID, "kernels[%d] = %d: kernel sizes 1..5 (k x k) are built", k, c->kernels[k]);
            if (c->strides[k] < 0 || c->strides[k] > 2) return fail(h, CTX_E_INVALID, "strides[%d] = %d: strides 1 and 2 are built", k, c->strides[k]);
        }
        if (c->featsize <= 0 || c->featsize % 32) return fail(h, CTX_E_INVALID, "featsize must be a multiple of 32");
        if (c->H <= 0 || c->W <= 0 || c->max_batch <= 0) return fail(h, CTX_E_INVALID, "H, W, max_batch must be positive");
        int hc = c->H, wc = c->W;
        for (int k = 0; k < 4; ++k) {
            const int sk = c->strides[k] ? c->strides[k] : ((k & 1) ? 2 : 1);
            const int s = sk == 2 && !(hc == 1 && wc == 1) ? 2 : 1;
            if (hc % s || wc % s) return fail(h, CTX_E_INVALID, "feature grid %dx%d: a stride-2 layer meets an odd grid larger than 1x1", c->H, c->W);
            hc /= s; wc /= s;
        }
        if (c->precision != CTX_PREC_F32 && c->precision != CTX_PREC_BF16X3) return fail(h, CTX_E_INVALID, "unsupported precision %d", c->precision);
        return CTX_OK;
    }
    if (c->C != 3) return fail(h, CTX_E_INVALID, "C must be 3");
    if (c->precision != CTX_PREC_F32 && c->precision != CTX_PREC_BF16X3) return fail(h, CTX_E_INVALID, "unsupported precision %d", c->precision);
    if (c->variant == CTX_VARIANT_REAL) {   // ContextAEReal: two stride-2 layers, fixed filters 32/16/16/8
        if (c->H <= 0 || c->W <= 0 || c->H % 4 || c->W % 4) return fail(h, CTX_E_INVALID, "H, W must be positive multiples of 4 (got %dx%d)", c->H, c->W);
        if (c->featsize <= 0 || c->featsize % 4) return fail(h, CTX_E_INVALID, "featsize must be a multiple of 4");
        if (c->max_batch <= 0) return fail(h, CTX_E_INVALID, "max_batch must be positive");
        return CTX_OK;
    }
    if (c->H <= 0 || c->W <= 0 || c->H % 16 || c->W % 16)
        return fail(h, CTX_E_INVALID, "H, W must be positive multiples of 16 (got %dx%d)", c->H, c->W);
    if (c->df_dim <= 0 || c->df_dim % 32) return fail(h, CTX_E_INVALID, "df_dim must be a multiple of 32");
    if (c->featsize <= 0 || c->featsize % 32) return fail(h, CTX_E_INVALID, "featsize must be a multiple of 32");
    if (c->max_batch <= 0) return fail(h, CTX_E_INVALID, "max_batch must be positive");
    return CTX_OK;
}

// TF variable inventory in arena order (names: SURVEY.md section 5; shapes: arm_shaping.py:24-29,
// 51-55, 66-79, 1282-1343)
void build_params(const ctx_config& c, std::vector<ParamInfo>& out, int64_t& total) {
    const int64_t d = c.df_dim, F = c.featsize, h16 = c.H / 16, w16 = c.W / 16;
    int64_t off = 0;
    auto add = [&](const std::string& name, std::vector<int64_t> shp) {
        ParamInfo p;
        p.name = name;
        p.ndim = (int)shp.size();
        p.size = 1;
        for (int i = 0; i < 4; ++i) {
            p.shape[i] = i < p.ndim ? shp[i] : 1;
            p.size *= p.shape[i];
        }
        p.offset = off;
        off += p.size;
        out.push_back(p);
    };
    auto enc = [&](const std::string& sc) {
        int64_t cin = c.C;
        const int64_t couts[4] = {d, 2 * d, 4 * d, 8 * d};
        for (int k = 0; k < 4; ++k) {
            add(sc + "/h" + std::to_string(k) + "_conv/w", {5, 5, cin, couts[k]});
            add(sc + "/h" + std::to_string(k) + "_conv/biases", {couts[k]});
            cin = couts[k];
        }
        add(sc + "/h4_lin/Matrix", {h16 * w16 * 8 * d, F});
        add(sc + "/h4_lin/bias", {F});
        add(sc + "/hz_lin/Matrix", {F, F});
        add(sc + "/hz_lin/bias", {F});
    };
    enc("conv_context");
    enc("conv");
    add("translate/trans_h0/Matrix", {2 * F, F});
    add("translate/trans_h0/bias", {F});
    add("translate/trans_z/Matrix", {F, F});
    add("translate/trans_z/bias", {F});
    add("deconv/d_h0_lin/Matrix", {F, 8 * d * h16 * w16});
    add("deconv/d_h0_lin/bias", {8 * d * h16 * w16});
    add("deconv/d_h1/w", {5, 5, 4 * d, 16 * d});
    add("deconv/d_h1/biases", {4 * d});
    add("deconv/d_h2/w", {5, 5, 2 * d, 8 * d});
    add("deconv/d_h2/biases", {2 * d});
    add("deconv/d_h3/w", {5, 5, d, 4 * d});
    add("deconv/d_h3/biases", {d});
    add("deconv/d_h4/w", {5, 5, c.C, 2 * d});
    add("deconv/d_h4/biases", {c.C});
    total = off;
}

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

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

#define TRY(expr)              \
    do {                       \
        int r_ = (expr);       \
        if (r_ != CTX_OK) return r_; \
    } while (0)

bool d_h4_direct(const ctx_handle* h, int c1, int c2, int hs, int ws, int stride);
int alloc_buffers(ctx_handle* h) {
    const int64_t B = h->Bm, d = h->d, F = h->F;
    TRY(dev_alloc(h, &h->u8, 3 * B * h->npi));
    TRY(dev_alloc(h, &h->img, 3 * B * h->npi));
    TRY(dev_alloc(h, &h->img4, 3 * B * h->npi / 3 * 4));
    TRY(dev_alloc(h, &h->Z, 3 * B * F));
    TRY(dev_alloc(h, &h->dZ, 3 * B * F));
    for (int k = 0; k < 4; ++k) {
        const int64_t pix = (int64_t)h->hh[k + 1] * h->ww[k + 1], ch = d << k;
        TRY(dev_alloc(h, &h->s[k], 2 * B * pix * ch));
        TRY(dev_alloc(h, &h->dS[k], 2 * B * pix * ch));
        TRY(dev_alloc(h, &h->c[k], B * pix * ch));
        TRY(dev_alloc(h, &h->dC[k], B * pix * ch));
        TRY(dev_alloc(h, &h->dSk[k], 2 * B * pix * ch));   // d loss / d (ctx skip h_k), per decoder pass
    }
    TRY(dev_alloc(h, &h->s[4], 2 * B * F));
    TRY(dev_alloc(h, &h->dS[4], 2 * B * F));
    TRY(dev_alloc(h, &h->c[4], B * F));
    TRY(dev_alloc(h, &h->dC[4], B * F));
    TRY(dev_alloc(h, &h->cz, B * F));
    TRY(dev_alloc(h, &h->dcz, B * F));
    TRY(dev_alloc(h, &h->th0, B * F));
    TRY(dev_alloc(h, &h->dth0, B * F));
    TRY(dev_alloc(h, &h->dsim2, 2 * B * F));
    TRY(dev_alloc(h, &h->dz, 2 * B * h->D0));
    TRY(dev_alloc(h, &h->dDz, 2 * B * h->D0));
    for (int k = 1; k <= 3; ++k) {   // e[k]: output of d_hk, spatial H >> (4-k), channels 8d >> k
        const int64_t pix = (int64_t)h->hh[4 - k] * h->ww[4 - k], ch = (8 * d) >> k;
        TRY(dev_alloc(h, &h->e[k], 2 * B * pix * ch));
        TRY(dev_alloc(h, &h->dE[k], 2 * B * pix * ch));
    }
    TRY(dev_alloc(h, &h->out, 2 * B * h->npi));
    // d_h4's scatter product exists only where the direct 3-channel kernel (convt3.hip) does not run: the split-bf16 mode / odd shapes
    // (a handle on the direct kernel keeps a small one for the starved inference launches, which take the product + gather route: forward)
    TRY(dev_alloc(h, &h->P3, (d_h4_direct(h, d, d, h->hh[1], h->ww[1], 2) ? std::min<int64_t>(B, PP_IMG) : 2 * B) * h->hh[1] * h->ww[1] * P3_LD, false));   // written by an epilogue, read by the gather: 64-bit indexing
    {   // (option "wconvt" bit 16)
        int64_t per = 0;
        for (int k = 1; k <= 3; ++k)     // only input grids of <= 16 positions take the product route (forward: `prod`)
            if (h->hh[5 - k] * h->ww[5 - k] <= 16) per = std::max<int64_t>(per, (int64_t)h->hh[5 - k] * h->ww[5 - k] * 25 * ((8 * d) >> k));
        if (per) TRY(dev_alloc(h, &h->PP, std::min<int64_t>(B, PP_IMG) * per, false));
    }
    TRY(dev_alloc(h, &h->dout, 2 * B * h->npi));
    TRY(dev_alloc(h, &h->dout4, 2 * B * h->npi / 3 * 4));
    int64_t maxc = std::max<int64_t>(h->D0, F);
    maxc = std::max<int64_t>(maxc, 16 * d);
    TRY(dev_alloc(h, &h->scratch, std::max<int64_t>(4 * LOSS_BLOCKS, (int64_t)COLSUM_SPLITS * maxc)));
    h->slab_floats = 32ll << 20;
    TRY(dev_alloc(h, &h->slab, h->slab_floats));
    TRY(dev_alloc(h, &h->wpack, DC_WPACK_FLOATS));
    for (int l = 0; l < ctx_handle::NLANE; ++l) {
        TRY(dev_alloc(h, &h->slabL[l], h->slab_floats));
        TRY(dev_alloc(h, &h->scratchL[l], std::max<int64_t>(4 * LOSS_BLOCKS, (int64_t)COLSUM_SPLITS * maxc)));
        TRY(dev_alloc(h, &h->wpackL[l], DC_WPACK_FLOATS));
    }
    TRY(dev_alloc(h, &h->scalars, 4));
    TRY(dev_alloc(h, &h->zeros, 64));
    if (hipMemset(h->zeros, 0, 64 * sizeof(float)) != hipSuccess) return fail(h, CTX_E_DEVICE, "hipMemset(zeros)");
    return CTX_OK;
}

// the 4-channel copy of a pointer into img / dout (cin = 3 loaders); pack_c4 refreshes `npix` pixels of it
const float* c4of(const ctx_handle* h, const float* p3) {
    const int64_t ni = 3 * (int64_t)h->Bm * h->npi;
    if (p3 >= h->img && p3 < h->img + ni) return h->img4 + (p3 - h->img) / 3 * 4;
    return h->dout4 + (p3 - h->dout) / 3 * 4;
}
void pack_c4(ctx_handle* h, const float* p3, int64_t npix) { pack3to4(h->stream, p3, const_cast<float*>(c4of(h, p3)), npix); }

// terms of `loss` (ctx_config.loss_terms; 0 = all)
int loss_terms_of(const ctx_handle* h) { return h->cfg.loss_terms ? h->cfg.loss_terms : 7; }

SplitWs ws_of(ctx_handle* h) { return SplitWs{h->slab, h->slab_floats, h->cfg.precision, h->gen ? 0 : 7}; }

// Everything below enqueues on h->stream with h->slab / h->scratch; LaneSwap points those at the second lane
// for the lifetime of a scope.  fork(): the second lane starts after everything enqueued so far on the
// main stream; join(): the main stream continues after everything enqueued so far on the second lane.
struct LaneSwap {
    ctx_handle* h;
    hipStream_t s0;
    float *sl0, *sc0, *wp0;
    LaneSwap(ctx_handle* h_, int lane) : h(h_), s0(h_->stream), sl0(h_->slab), sc0(h_->scratch), wp0(h_->wpack) {
        h->stream = h->aux[lane]; h->slab = h->slabL[lane]; h->scratch = h->scratchL[lane]; h->wpack = h->wpackL[lane];
    }
    ~LaneSwap() { h->stream = s0; h->slab = sl0; h->scratch = sc0; h->wpack = wp0; }
};
// (inside a graph capture the lanes are captured as branches -- fork / join are event record + wait, which stream capture follows --
// when option graph_lanes is set: the two encoders of a translate call at batch 25 then run side by side)
bool use_lanes(const ctx_handle* h) { return h->overlap && h->aux[0] && !h->prof_on && (!h->capturing || h->opt.v[OPT_GRAPH_LANES]); }
// fork: `lane` starts after everything enqueued so far on the CURRENT stream; join: the current stream
// continues after everything enqueued so far on `lane`
void fork(ctx_handle* h, int lane) {
    (void)hipEventRecord(h->ev_fork[lane], h->stream);
    (void)hipStreamWaitEvent(h->aux[lane], h->ev_fork[lane], 0);
}
void join(ctx_handle* h, int lane) {
    (void)hipEventRecord(h->ev_join[lane], h->aux[lane]);
    (void)hipStreamWaitEvent(h->stream, h->ev_join[lane], 0);
}
// Side(h, lane): run the enclosed launches on `lane`, ordered after what the current stream has queued;
// a no-op (stays on the current stream) when lanes are off
struct Side {
    ctx_handle* h;
    bool on;
    hipStream_t s0 = nullptr;
    float *sl0 = nullptr, *sc0 = nullptr, *wp0 = nullptr;
    Side(ctx_handle* h_, int lane) : h(h_), on(lane >= 0 && use_lanes(h_)) {
        if (!on) return;
        fork(h, lane);
        s0 = h->stream; sl0 = h->slab; sc0 = h->scratch; wp0 = h->wpack;
        h->stream = h->aux[lane]; h->slab = h->slabL[lane]; h->scratch = h->scratchL[lane]; h->wpack = h->wpackL[lane];
    }
    ~Side() { if (on) { h->stream = s0; h->slab = sl0; h->scratch = sc0; h->wpack = wp0; } }
};
constexpr int LANE_CTX = 0, LANE_DW = 1;

// ---- Adam beside the backward ------------------------------------------------------------------------
// Adam is 7 arena passes of HBM traffic (0.24 ms for ContextSkipNew's 47.6 M parameters) and nothing else in the step is HBM-bound,
// so the fused training entry points run it in slices on `adam_stream` while the matrix-core kernels of the remaining backward run:
// a slice may go as soon as (1) its gradients are final and (2) nothing later in this step reads its parameters.  backward() marks
// those points with adam_early(); adam_end() updates what is left on the compute stream and joins.  The arithmetic per element is
// that of adam_step (same kernel, same lr_t): results are bit-identical to the unsliced update (tests/test_gpu_parity.py).
// MEASURED: round 3 (six back-to-back bench runs) no gain, 13.79 / 13.81 ms with the slices against 13.79 / 13.79 without.  Round 4, with
// the nontemporal Adam kernel: -0.06..-0.08 ms in four A/B pairs (13.205 -> 13.134, 13.247 -> 13.167, 13.185 -> 13.124) -- and the kernel
// trace shows why it is not more: the runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES = 4 hardware queues, the process's null
// stream holds one, compute stream and two lanes the other three, and `adam_stream` lands on the filter-gradient lane's queue, so the
// slices run beside the dx chain but in turn with the filter gradients.  With GPU_MAX_HW_QUEUES=8 it has its own queue and the step
// loses another 0.05 ms, but ContextAEReal's small launches then run truly side by side and get slower (2.71 -> 3.11 ms), so the
// library does not ask for it (profiles/archive/round4_e_early_adam_queues.txt).  ON by default (option "early_adam"; read at every step).
void adam_launch(ctx_handle* h, hipStream_t s, int64_t first, int64_t end) {
    adam(s, h->arena + first, h->arena + h->Ppad + first, h->arena + 2 * h->Ppad + first, h->arena + 3 * h->Ppad + first, end - first,
         h->adam_lr_t, 0.9f, 0.999f, 1e-8f);
}
void adam_begin(ctx_handle* h, float lr) {
    const bool env_on = h->opt.v[OPT_EARLY_ADAM] != 0;
    const double b1 = 0.9, b2 = 0.999;
    h->adam_t += 1;
    h->adam_lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow(b2, (double)h->adam_t)) / (1.0 - std::pow(b1, (double)h->adam_t)));
    h->adam_done.clear();
    h->pack.version++;               // the parameters change in this step: packed filters are stale from here on
    // (only where the update is worth hiding: ContextAEReal's 1.2 M parameters are a 7 us update, and the slices' events and queue hops
    // among its 5-30 us launches cost 0.4 ms of a 2.6 ms step -- bench.py's secondary leg against forward_backward + adam on one box: 3.00 -> 2.58 ms, round 5)
    h->adam_early_on = env_on && h->adam_stream && use_lanes(h) && h->P >= (4ll << 20);
}
// [first, end) is final in the order of the CURRENT stream plus (lane >= 0) of that side lane
void adam_early(ctx_handle* h, int64_t first, int64_t end, int lane) {
    if (!h->adam_early_on || first < 0 || end <= first || (first & 3) || (end & 3)) return;
    (void)hipEventRecord(h->adam_ev[0], h->stream);
    (void)hipStreamWaitEvent(h->adam_stream, h->adam_ev[0], 0);
    if (lane >= 0) {
        (void)hipEventRecord(h->adam_ev[1], h->aux[lane]);
        (void)hipStreamWaitEvent(h->adam_stream, h->adam_ev[1], 0);
    }
    adam_launch(h, h->adam_stream, first, end);
    h->adam_done.emplace_back(first, end);
}
void adam_end(ctx_handle* h) {
    std::sort(h->adam_done.begin(), h->adam_done.end());
    int64_t at = 0;
    ProfScope ps(h, "adam", "adam", 0.0);
    for (size_t i = 0; i <= h->adam_done.size(); ++i) {
        const int64_t stop = i < h->adam_done.size() ? h->adam_done[i].first : h->Ppad;
        if (stop > at) adam_launch(h, h->stream, at, stop);
        if (i < h->adam_done.size()) at = h->adam_done[i].second;
    }
    if (!h->adam_done.empty()) {
        (void)hipEventRecord(h->adam_ev_done, h->adam_stream);
        (void)hipStreamWaitEvent(h->stream, h->adam_ev_done, 0);
    }
    h->adam_early_on = false;
    h->adam_done.clear();
    h->pack.version++;               // nothing packed while the update was in flight (adam_begin .. here) may pass as current afterwards
}

// the tail of the gradient arena [first, Ppad) (translate/*, deconv/*: arena order is conv_context, conv, translate, deconv) is final
int dp_reduce_range(ctx_handle* h, int64_t first, int64_t count);
void fire_bucket(ctx_handle* h, int64_t first) {
    if (!h->bucket_fn && !h->dp_in_step) {       // plain fused step: nothing after this point reads translate/* or deconv/* parameters
        adam_early(h, first, h->Ppad, LANE_DW);
        return;
    }
    if (h->dp_in_step) {                         // ctx_dp_train_step: the tail bucket goes out while the encoders' backward is enqueued
        // its filter / bias gradients ran on the side lane: the COLLECTIVE's stream waits for that lane, the compute stream does not
        // (joining the lane into the compute stream here cost 0.3 ms per step: the encoders' backward then queued behind the decoder's
        // filter gradients instead of running beside them)
        if (use_lanes(h)) {
            (void)hipEventRecord(h->ev_join[LANE_DW], h->aux[LANE_DW]);
            (void)hipStreamWaitEvent(h->dp_stream, h->ev_join[LANE_DW], 0);
        }
        h->dp_rc = dp_reduce_range(h, first, h->Ppad - first);      // (checked by ctx_dp_train_step after backward returns)
        h->dp_split = first;
        return;
    }
    if (use_lanes(h)) join(h, LANE_DW);          // a host callback expects the bucket final in the compute stream's order
    h->bucket_fn(h->bucket_user, 0, first, h->Ppad - first);
}

// ctx_dp_train_step, inside the encoders' backward: gradients [first, end) -- an encoder's h4_lin / hz_lin, two thirds of its
// parameters -- are final in the order of the CURRENT stream plus (lane >= 0) that side lane: their all-reduce starts now, behind
// the tail bucket on the collective stream, instead of waiting for the convolutions' filter gradients.  What is left for the end
// of the step are the encoders' conv filters (2 x 4.3 M of 47.6 M floats).  Same order of collectives on every rank (program order).
void dp_bucket(ctx_handle* h, int64_t first, int64_t end, int lane) {
    if (!h->dp_in_step || h->dp_rc != CTX_OK || first < 0 || end <= first) return;
    if (lane >= 0 && use_lanes(h)) {
        (void)hipEventRecord(h->ev_join[lane], h->aux[lane]);
        (void)hipStreamWaitEvent(h->dp_stream, h->ev_join[lane], 0);
    }
    h->dp_rc = dp_reduce_range(h, first, end - first);
    h->dp_done.emplace_back(first, end);
}

const char* const K_CONV = "igemm<ConvGather,Plain>";
const char* const K_CONVT = "igemm<ConvTGather,ConvTWeights>";
const char* const K_CONVT1 = "igemm<ConvGather,ConvTWeights>";   // stride-1 conv2d_transpose as a flipped correlation (convt1_fwd)
const char* const K_WGRAD = "igemm<WgradBig,WgradSmall>";
const char* const K_C3FWD = "igemm<C3Gather,C3Weights>";
const char* const K_C3WGRAD = "igemm<C3WgradBig,WgradSmall>";
const char* const K_FCFWD = "igemm<KmPlain,NmPlain>";
const char* const K_FCDX = "igemm<KmPlain,KmPlain>";
const char* const K_FCDW = "igemm<NmPlain,NmPlain>";
const char* const K_CONVT3 = "convt3_gather";
const char* const K_CONVT3P = "igemm<Cat2,KmPlain>";
const char* const K_CONVT3D = "convt3_kernel";
// the narrow-channel direct kernels of dconv.h (ContextSkipNew's 3-channel edge layers in f32, all of ContextAEReal's narrow path):
// a launch group is labelled with the kernel that actually runs it
const char* const K_WCONVT = "wconvt_kernel";       // wide-channel transposed conv with the input halo tile in LDS (wconvt.hip)
const char* const K_C3CONV = "c3conv_kernel";       // conv from 3 channels, 4-wave blocks (c3conv.hip)
const char* const K_DCFWD = "dconv_fwd_kernel";
const char* const K_DCWGRAD = "dconv_wgrad_kernel";
const char* const K_C3WGRADK = "c3wgrad_kernel";    // filter gradient with a 3-channel big-grid side on whole 128-pixel tiles (c3wgrad.hip)
const char* dw_label(const DcWgrad& W) { return c3wgrad_ok(W) ? K_C3WGRADK : K_DCWGRAD; }
const char* const K_RCHAIN = "rchain";               // ContextAEReal's FC middle in three launches (rchain.hip)
const char* const K_COLSUM = "colsum";
const char* const K_EW = "elementwise";

// ---- layer launch helpers -------------------------------------------------------------------------
thread_local const float* g_zeros = nullptr;   // 256 B of device zeros: where the loaders send out-of-range lanes (set per call)
NmPlain nm(const float* p, int64_t ld, int R, int K) { return NmPlain{p, ld, nullptr, 0, R, R, K, g_zeros}; }
KmPlain km(const float* p, int64_t ld, int R, int K) { return KmPlain{p, ld, nullptr, 0, K, R, K / KC, g_zeros}; }

// ContextSkipNew's 3-channel edge layers (h0_conv forward / filter gradient, d_h4's input and filter gradients) on the direct
// kernels of dconv.h: the frames and d loss / d out are read as they are ([pixel][3]), so the 4-channel copies and their pack
// passes go away.  ON by default, in both precisions since the end of round 3 (the split-bf16 mode used to keep the implicit GEMM on 4-channel copies):
// measured on the persistent / prefetching dconv kernels 0.99 -> 0.71 ms of layer time per step.  CTX_DCONV_C3=0 restores the
// implicit GEMM.  The direct forward kernel packs at most 128 filter columns (dconv_ok): d_h4's input gradient has N = 2 * df_dim
// columns, so a handle with df_dim > 64 stays on the implicit GEMM for all of its 3-channel layers (decided per handle, because
// the implicit GEMM needs the 4-channel copies refreshed by forward / backward).
bool use_dc3(const ctx_handle* h) {
    const bool on = (h->opt.v[OPT_DIRECT3] & 1) != 0;
    // (both precisions: the seven 3-channel launches are 1 % of the step's FLOPs, and their exact-f32 direct kernels are faster than the
    // split-bf16 implicit GEMM on 4-channel copies -- 0.8 ms against 1.5 ms of the split-bf16 step -- and more accurate)
    return on && dconv_ok(3, h->d) && dconv_ok(3, 2 * h->d);
}

// d_h4 (conv2d_transpose to the 3 image channels) in one pass on the vector ALUs (convt3.hip) instead of scatter product + gather.
// ContextSkipNew: both precisions (exact f32 arithmetic either way); the table-driven models: exact-f32 mode only.
// CTX_CONVT3_DIRECT=0 restores the two-step route (and its P3 buffer).
bool d_h4_direct(const ctx_handle* h, int c1, int c2, int hs, int ws, int stride) {
    const bool on = (h->opt.v[OPT_DIRECT3] & 8) != 0;
    return on && (h->cfg.precision == CTX_PREC_F32 || !h->gen) && convt3_direct_ok(c1, c2, hs, ws, stride);
}
bool use_q(int nimg) { return opt(OPT_POSMAJOR) && nimg >= 64; }

// smallest grid (positions) whose transposed conv runs position-major: in f32 the 4x4 grids keep the class-major launch (64 problems of
// 1 .. 9 taps leave a tail)
int q_minpos(const ctx_handle* h) { return h->cfg.precision ? 0 : 64; }

// y = lrelu(conv2d(x) + b): x [nimg, hb, wb, ca] -> y [nimg, hb/2, wb/2, cb]
void conv_layer(ctx_handle* h, const std::string& name, const float* x, int nimg, int hb, int wb, int ca, const float* w,
                const float* b, float* y, int cb) {
    const int hs = hb / 2, ws = wb / 2, R = nimg * hs * ws;
    Epi ep;
    ep.out1 = y; ep.ld1 = cb; ep.bias = b; ep.lrelu = 1;
    const bool c3 = ca == 3 && use_dc3(h) && c3conv_ok(hb, wb, 2, cb, ep);
    ProfScope ps(h, name + " fwd", ca == 3 ? (c3 ? K_C3CONV : use_dc3(h) ? K_DCFWD : K_C3FWD) : K_CONV, 2.0 * R * 25 * ca * cb, tap_frac(hb, wb, 5, 2));
    if (c3) c3conv(h->stream, x, nimg, hb, wb, 2, w, cb, ep);
    else if (ca == 3 && use_dc3(h)) {
        DcFwd P{};
        P.x1 = x; P.ld1 = 3; P.c1 = 3; P.CI = 3; P.hin = hb; P.win = wb; P.nimg = nimg; P.w = w; P.wmode = 0; P.N = cb; P.ep = ep; P.wp = h->wpack; P.pc = &h->pack;
        dconv_conv(h->stream, P, 2, 1);
    } else if (ca == 3) conv3_fwd(h->stream, KmC3Gather{c4of(h, x), hb, wb, hs, ws, R, g_zeros}, NmC3Weights{w, cb, g_zeros}, ep, R, cb, ws_of(h));
    else if (use_q(nimg)) conv_fwd_q(h->stream, KmConvGatherQ{x, ca, make_posgeo(hs, ws, hb, wb, 2, 1, 5, ca / KC), nimg, g_zeros}, NmConvWeightsQ{w, ca, cb, 5, g_zeros}, ep, cb, ws_of(h));
    else conv_fwd(h->stream, KmConvGather{x, ca, hb, wb, hs, ws, ca / KC, R, g_zeros}, nm(w, cb, cb, 25 * ca), ep, R, cb, ws_of(h));
}

// y = act(x W + b), x possibly [x0 | x1] along K
void fc_layer(ctx_handle* h, const std::string& name, const KmPlain& a, int M, int K, const float* w, const float* b, int N,
              int lrelu, float* y) {
    ProfScope ps(h, name + " fwd", K_FCFWD, 2.0 * M * K * N);
    Epi ep;
    ep.out1 = y; ep.ld1 = N; ep.bias = b; ep.lrelu = lrelu;
    gemm_fc_fwd(h->stream, a, nm(w, N, N, K), ep, M, N, K / KC, ws_of(h));
}

// dx = dy W^T (+ epilogue): dy [M, N], W [K, N] -> dx [M, K]
void fc_dx(ctx_handle* h, const std::string& name, const float* dy, int M, int N, const float* w, int K, Epi ep) {
    ProfScope ps(h, name + " dx", K_FCDX, 2.0 * M * K * N);
    gemm_fc_dx(h->stream, km(dy, N, M, N), km(w, N, K, N), ep, M, K, N / KC, ws_of(h));
}

// dW = x^T dy, db = colsum(dy): x [M rows] possibly [x0 | x1] along features
void bias_grad(ctx_handle* h, const std::string& name, const float* dy, int64_t rows, int C, float* db) {
    ProfScope ps(h, name + " db", K_COLSUM, 0.0);
    colsum(h->stream, dy, rows, C, h->scratch, db);
}

void fc_dw_launch(hipStream_t s, const NmPlain& x, const NmPlain& dy, Epi ep, int K, int N, int nch, SplitWs ws) { gemm_fc_dw(s, x, dy, ep, K, N, nch, ws); }
void fc_dw_launch(hipStream_t s, const NmPlain2& x, const NmPlain& dy, Epi ep, int K, int N, int nch, SplitWs ws) { gemm_fc_dw2(s, x, dy, ep, K, N, nch, ws); }

template <class XL>
void fc_dw(ctx_handle* h, const std::string& name, const XL& x, int K, const float* dy, int M, int N, float* dw, float* db) {
    {
        ProfScope ps(h, name + " dw", K_FCDW, 2.0 * M * K * N);
        Epi ep;
        ep.out1 = dw; ep.ld1 = N;
        fc_dw_launch(h->stream, x, nm(dy, N, N, M), ep, K, N, (M + KC - 1) / KC, ws_of(h));
    }
    bias_grad(h, name, dy, M, N, db);
}

struct Scope {
    float *w[4], *b[4], *w4, *b4, *wz, *bz;
    float *gw[4], *gb[4], *gw4, *gb4, *gwz, *gbz;
};

Scope scope_of(ctx_handle* h, const std::string& sc) {
    Scope s;
    for (int k = 0; k < 4; ++k) {
        const std::string base = sc + "/h" + std::to_string(k) + "_conv/";
        s.w[k] = h->Wp((base + "w").c_str()); s.b[k] = h->Wp((base + "biases").c_str());
        s.gw[k] = h->Gp((base + "w").c_str()); s.gb[k] = h->Gp((base + "biases").c_str());
    }
    s.w4 = h->Wp((sc + "/h4_lin/Matrix").c_str()); s.b4 = h->Wp((sc + "/h4_lin/bias").c_str());
    s.wz = h->Wp((sc + "/hz_lin/Matrix").c_str()); s.bz = h->Wp((sc + "/hz_lin/bias").c_str());
    s.gw4 = h->Gp((sc + "/h4_lin/Matrix").c_str()); s.gb4 = h->Gp((sc + "/h4_lin/bias").c_str());
    s.gwz = h->Gp((sc + "/hz_lin/Matrix").c_str()); s.gbz = h->Gp((sc + "/hz_lin/bias").c_str());
    return s;
}

// arm_shaping.py:1282-1288 / :1290-1307: four conv+lrelu, h4_lin+lrelu, hz_lin (+lrelu for `conv`)
void encoder_fwd(ctx_handle* h, const std::string& scn, const Scope& sc, const float* x, int nimg, float* const act[5], float* z,
                 int z_lrelu) {
    const int d = h->d, F = h->F;
    const float* in = x;
    int ca = 3;
    for (int k = 0; k < 4; ++k) {
        conv_layer(h, scn + "/h" + std::to_string(k) + "_conv", in, nimg, h->hh[k], h->ww[k], ca, sc.w[k], sc.b[k], act[k], d << k);
        in = act[k];
        ca = d << k;
    }
    const int K3 = h->hh[4] * h->ww[4] * 8 * d;   // NHWC flatten, arm_shaping.py:1287
    fc_layer(h, scn + "/h4_lin", km(act[3], K3, nimg, K3), nimg, K3, sc.w4, sc.b4, F, 1, act[4]);
    fc_layer(h, scn + "/hz_lin", km(act[4], F, nimg, F), nimg, F, sc.wz, sc.bz, F, z_lrelu, z);
}

enum Mode { MODE_TRAIN, MODE_TRANSLATE, MODE_ENCODE };

}  // namespace
#include "ctxtrans_gen.inc"
namespace {

// Forward.  TRAIN/EVAL: st = [tgt | src] (2B), decoder = [translated | truth] (2B).
// TRANSLATE: only what translated_z / out depend on (src encoder, ctx encoder, translate, decoder
// pass 1) -- the subgraph TF would run for base.py:216-218.  ENCODE: `conv` encoder on src only.
void forward(ctx_handle* h, int B, Mode mode) {
    OptScope os(&h->opt);
    if (h->gen) { gen_forward(h, B, mode); return; }
    g_zeros = h->zeros;
    const int d = h->d, F = h->F;
    const int64_t npi = h->npi;
    const Scope st = scope_of(h, "conv"), cx = scope_of(h, "conv_context");
    float* src_z = h->Z + 2ll * B * F;
    const bool lanes = use_lanes(h) && mode != MODE_ENCODE;
    // images through `conv_context`: B, or the ONE frame every row shares (its code and skip activations are then read with row stride 0
    // / image index n % 1 by their consumers -- same values as B copies, 1 / B of the work)
    const int nc = mode == MODE_TRANSLATE && h->ctx_single ? 1 : B;
    // refresh the 4-channel copy of the frames in use (what the cin = 3 loaders read)
    if (use_dc3(h)) {}
    else if (mode == MODE_TRAIN) pack_c4(h, h->img, 3ll * B * h->H * h->W);
    else pack_c4(h, h->img + B * npi, (mode == MODE_ENCODE ? 1ll : 2ll) * B * h->H * h->W);
    // (inside a captured translate the second branch starts ~70 us behind the first whichever chain is issued first, or layer by layer --
    // measured, profiles/round5_c_reward_latency.txt; with ONE context frame its chain still ends before the 25-frame `conv` chain needs it)
    if (lanes) {
        fork(h, LANE_CTX);
        LaneSwap sw(h, LANE_CTX);
        encoder_fwd(h, "conv_context", cx, h->img + 2 * B * npi, nc, h->c, h->cz, 0);
    }
    if (mode == MODE_TRAIN) encoder_fwd(h, "conv", st, h->img, 2 * B, h->s, h->Z + (int64_t)B * F, 1);
    else encoder_fwd(h, "conv", st, h->img + B * npi, B, h->s, src_z, 1);
    if (mode == MODE_ENCODE) return;
    if (lanes) join(h, LANE_CTX);
    else encoder_fwd(h, "conv_context", cx, h->img + 2 * B * npi, nc, h->c, h->cz, 0);
    // translate (arm_shaping.py:1309-1312): trans_h0 on concat([src_z, ctx_z], 1), then trans_z
    KmPlain tcat{src_z, F, h->cz, nc == 1 && B > 1 ? 0 : F, F, B, 2 * F / KC, g_zeros};
    fc_layer(h, "translate/trans_h0", tcat, B, 2 * F, h->Wp("translate/trans_h0/Matrix"), h->Wp("translate/trans_h0/bias"), F, 1, h->th0);
    fc_layer(h, "translate/trans_z", km(h->th0, F, B, F), B, F, h->Wp("translate/trans_z/Matrix"), h->Wp("translate/trans_z/bias"), F, 0, h->Z);
    // decoder (arm_shaping.py:1321-1330, :1334-1343)
    const int nd = mode == MODE_TRAIN ? 2 * B : B;
    fc_layer(h, "deconv/d_h0_lin", km(h->Z, F, nd, F), nd, F, h->Wp("deconv/d_h0_lin/Matrix"), h->Wp("deconv/d_h0_lin/bias"), (int)h->D0, 1, h->dz);
    const float* dec = h->dz;
    for (int k = 1; k <= 4; ++k) {
        const int hs = h->hh[5 - k], ws = h->ww[5 - k];      // input grid of d_hk
        const int c1 = (16 * d) >> k, c2 = c1;                // decoder stream | ctx skip h_{4-k}
        const int ca = k < 4 ? (8 * d) >> k : 3;
        const std::string nm_ = "deconv/d_h" + std::to_string(k);
        const float* w = h->Wp((nm_ + "/w").c_str());
        const float* b = h->Wp((nm_ + "/biases").c_str());
        const float* skip = h->c[4 - k];
        const double fl = 2.0 * nd * hs * ws * 25 * (c1 + c2) * ca, uf = tap_frac(2 * hs, 2 * ws, 5, 2);
        if (k < 4) {
            const int R = nd * hs * ws;
            const bool wide = h->cfg.precision == CTX_PREC_F32 && wconvt_ok(hs, ws, c1, c2, ca, nd);
            // starved inference launches (the reward hook's 25 frames): one plain product + a gather (launch.h: convt_product)
            // (measured at 25 frames: 4x4 grid 130 -> 87 us; the 8x8 / 16x16 grids 75 / 73 -> 87 / 85 us, so those stay on the tiles)
            const bool prod = mode != MODE_TRAIN && nd <= PP_IMG && hs * ws <= 16 && h->PP && (h->opt.v[OPT_WCONVT] & 16) && ca % 4 == 0 && (c1 + c2) % KC == 0 && c1 % KC == 0;
            ProfScope ps(h, nm_ + " fwd", prod ? K_CONVT3P : wide ? K_WCONVT : K_CONVT, fl, uf);
            Epi ep;
            ep.out1 = h->e[k]; ep.ld1 = ca; ep.bias = b; ep.lrelu = 1;
            if (prod) {
                convt_product(h->stream, KmCat2{dec, c1, c1, skip, c2, nc, hs * ws, R, (c1 + c2) / KC, g_zeros}, w, c1 + c2, ca, h->PP, R, ws_of(h));
                convt_gather(h->stream, h->PP, b, h->e[k], nd, hs, ws, ca, 1);
            } else if (wide) wconvt_fwd(h->stream, dec, c1, skip, c2, nc, nd, hs, ws, w, ca, ep, ws_of(h));
            else if (use_q(nd) && hs * ws >= q_minpos(h)) convt_fwd_q(h->stream, KmConvTGatherQ{dec, c1, c1, skip, c2, nc, make_tposgeo(hs, ws, 5, 1, (c1 + c2) / KC), nd, g_zeros},
                                       KmConvTWeightsQ{w, ca, c1 + c2, 5, g_zeros}, ep, ca, ws_of(h));
            else convt_fwd(h->stream, KmConvTGather{dec, c1, c1, skip, c2, nc, hs, ws, (c1 + c2) / KC, R, g_zeros},
                           KmConvTWeights{w, ca, c1 + c2, (c1 + c2) / KC, g_zeros}, ep, R, ca, ws_of(h));
            dec = h->e[k];
        } else {
            const int R = nd * hs * ws;
            // (the same for d_h4 at <= PP_IMG images: the direct kernel offers 200 two-wave tiles to 256 CUs there, 54 us at 25 frames)
            const bool prod3 = mode != MODE_TRAIN && nd <= PP_IMG && (h->opt.v[OPT_WCONVT] & 16) && (c1 + c2) % KC == 0 && c1 % KC == 0;
            if (d_h4_direct(h, c1, c2, hs, ws, 2) && !prod3) {
                ProfScope ps(h, nm_ + " fwd", K_CONVT3D, fl, uf);
                convt3_direct(h->stream, dec, c1, skip, c2, nc, nd, hs, ws, 2, w, b, h->out);
            } else {
                { ProfScope ps(h, nm_ + " fwd product", K_CONVT3P, fl, uf);
                  convt3_product(h->stream, KmCat2{dec, c1, c1, skip, c2, nc, hs * ws, R, (c1 + c2) / KC, g_zeros}, w, c1 + c2, h->P3, R, ws_of(h)); }
                { ProfScope ps(h, nm_ + " fwd gather", K_CONVT3, 0.0);
                  convt3_gather(h->stream, h->P3, b, h->out, nd, hs, ws); }
            }
        }
    }
}

// d loss / d params into the grad arena (what AdamOptimizer.minimize differentiates,
// scripts/train_script.py:128).  Every gradient tensor is written exactly once.
void backward(ctx_handle* h, int B, int sim_batch) {
    OptScope os(&h->opt);
    if (h->gen) { gen_backward(h, B, sim_batch); return; }
    g_zeros = h->zeros;
    const int d = h->d, F = h->F;
    const int64_t npi = h->npi;
    float* tgt_z = h->Z + (int64_t)B * F;
    float* src_z = h->Z + 2ll * B * F;
    {
        ProfScope ps(h, "losses", K_EW, 0.0);
        losses(h->stream, h->out, h->img, h->dout, npi, B, h->Z, tgt_z, h->dsim2, F, sim_batch, h->scratch, h->scalars, 0, loss_terms_of(h));
        if (!use_dc3(h)) pack_c4(h, h->dout, 2ll * B * h->H * h->W);
    }

    // ---- decoder, both passes at once (batch 2B)
    const float* dy = h->dout;
    for (int k = 4; k >= 1; --k) {
        const int hs = h->hh[5 - k], wsm = h->ww[5 - k], hb = 2 * hs, wb = 2 * wsm;
        const int c1 = (16 * d) >> k, c2 = c1, cb = c1 + c2;
        const int ca = k < 4 ? (8 * d) >> k : 3;
        const int R = 2 * B * hs * wsm;
        const std::string nm_ = "deconv/d_h" + std::to_string(k);
        const float* w = h->Wp((nm_ + "/w").c_str());
        const float* dec_in = k > 1 ? h->e[k - 1] : h->dz;      // decoder half of the concat input
        float* d_dec = k > 1 ? h->dE[k - 1] : h->dDz;
        const double fl = 2.0 * R * 25 * cb * ca, uf = tap_frac(hb, wb, 5, 2);
        NmWgradSmall2 small{dec_in, c1, c1, h->c[4 - k], c2, B, cb, hs * wsm, make_pixdiv(1, hs * wsm).ws_sh, R, g_zeros};
        Epi eg;
        eg.out1 = h->Gp((nm_ + "/w").c_str()); eg.ld1 = cb;
        // input gradient = SAME stride-2 conv of dy with the same filter read as [5,5,ca,cb]; cols < c1
        // are the decoder stream (masked by its lrelu), cols >= c1 the ctx skip of this pass
        Epi ed;
        ed.out1 = d_dec; ed.ld1 = c1; ed.nsplit = c1; ed.mask = dec_in; ed.ldm = c1;
        ed.out2 = h->dSk[4 - k]; ed.ld2 = c2;
        if (ca == 3 && use_dc3(h)) {
            { Side sd(h, LANE_DW);
              bias_grad(h, nm_, dy, (int64_t)2 * B * hb * wb, ca, h->Gp((nm_ + "/biases").c_str()));
              DcWgrad Wg{};
              Wg.big = dy; Wg.ldb = 3; Wg.CA = 3; Wg.s1 = dec_in; Wg.ld1 = c1; Wg.c1 = c1; Wg.s2 = h->c[4 - k]; Wg.ld2 = c2; Wg.nmod2 = B; Wg.CB = cb;
              Wg.hb = hb; Wg.wb = wb; Wg.hs = hs; Wg.ws = wsm; Wg.nimg = 2 * B; Wg.S = 2; Wg.pad = 1; Wg.out = eg.out1;
              ProfScope ps(h, nm_ + " dw", dw_label(Wg), fl, uf);
              dconv_wgrad(h->stream, Wg, h->slab, h->slab_floats); }
            if (c3conv_ok(hb, wb, 2, cb, ed)) {
                ProfScope ps(h, nm_ + " dx", K_C3CONV, fl, uf);
                c3conv(h->stream, dy, 2 * B, hb, wb, 2, w, cb, ed);
            } else {
              ProfScope ps(h, nm_ + " dx", K_DCFWD, fl, uf);
              DcFwd D{};
              D.x1 = dy; D.ld1 = 3; D.c1 = 3; D.CI = 3; D.hin = hb; D.win = wb; D.nimg = 2 * B; D.w = w; D.wmode = 0; D.N = cb; D.ep = ed; D.wp = h->wpack; D.pc = &h->pack;
              dconv_conv(h->stream, D, 2, 1); }
        } else if (ca == 3) {
            { Side sd(h, LANE_DW);
              bias_grad(h, nm_, dy, (int64_t)2 * B * hb * wb, ca, h->Gp((nm_ + "/biases").c_str()));
              ProfScope ps(h, nm_ + " dw", K_C3WGRAD, fl, uf); conv3_wgrad2(h->stream, NmC3WgradBig{c4of(h, dy), hb, wb, make_pixdiv(hs, wsm), R, g_zeros}, small, eg, cb, ws_of(h)); }
            { ProfScope ps(h, nm_ + " dx", K_C3FWD, fl, uf); conv3_fwd(h->stream, KmC3Gather{c4of(h, dy), hb, wb, hs, wsm, R, g_zeros}, NmC3Weights{w, cb, g_zeros}, ed, R, cb, ws_of(h)); }
        } else {
            { Side sd(h, LANE_DW);
              bias_grad(h, nm_, dy, (int64_t)2 * B * hb * wb, ca, h->Gp((nm_ + "/biases").c_str()));
              ProfScope ps(h, nm_ + " dw", K_WGRAD, fl, uf);
              if (rect_ok(2 * B) && rect_ok(B)) {
                  const RectGeo rg = make_rect(2 * B, hs, wsm, hb, wb, 2, 1, 5);
                  conv_wgrad2_r(h->stream, NmWgradBigR{dy, ca, ca, rg, g_zeros}, NmWgradSmall2R{dec_in, c1, c1, h->c[4 - k], c2, B, cb, rg, g_zeros}, eg, ca, cb, ws_of(h));
              } else if (patch_ok(hs, wsm)) {
                  const PatchGeo pg = make_patch(2 * B, hs, wsm);
                  conv_wgrad2_p(h->stream, NmWgradBigP{dy, ca, ca, wb, pg, g_zeros}, NmWgradSmall2P{dec_in, c1, c1, h->c[4 - k], c2, B, cb, pg, g_zeros}, eg, ca, cb, ws_of(h));
              } else conv_wgrad2(h->stream, NmWgradBig{dy, ca, ca, hb, wb, make_pixdiv(hs, wsm), R, g_zeros}, small, eg, ca, cb, ws_of(h)); }
            { ProfScope ps(h, nm_ + " dx", K_CONV, fl, uf);
              if (use_q(2 * B)) conv_fwd_q(h->stream, KmConvGatherQ{dy, ca, make_posgeo(hs, wsm, hb, wb, 2, 1, 5, ca / KC), 2 * B, g_zeros}, NmConvWeightsQ{w, ca, cb, 5, g_zeros}, ed, cb, ws_of(h));
              else conv_fwd(h->stream, KmConvGather{dy, ca, hb, wb, hs, wsm, ca / KC, R, g_zeros}, nm(w, cb, cb, 25 * ca), ed, R, cb, ws_of(h)); }
        }
        dy = d_dec;
    }
    // d_h0_lin: input Z[0:2B] = [trans_z | tgt_z]; simloss adds +-c(trans_z - tgt_z) to its gradient
    {
        const int D0 = (int)h->D0;
        { Side sd(h, LANE_DW);
          fc_dw(h, "deconv/d_h0_lin", nm(h->Z, F, F, 2 * B), F, h->dDz, 2 * B, D0, h->Gp("deconv/d_h0_lin/Matrix"), h->Gp("deconv/d_h0_lin/bias")); }
        Epi ep;
        ep.out1 = h->dZ; ep.ld1 = F; ep.add1 = h->dsim2; ep.lda1 = F;
        fc_dx(h, "deconv/d_h0_lin", h->dDz, 2 * B, D0, h->Wp("deconv/d_h0_lin/Matrix"), F, ep);
    }
    // ---- translate MLP: d trans_z = dZ[0:B]
    {
        { Side sd(h, LANE_DW);
          fc_dw(h, "translate/trans_z", nm(h->th0, F, F, B), F, h->dZ, B, F, h->Gp("translate/trans_z/Matrix"), h->Gp("translate/trans_z/bias")); }
        Epi e1;
        e1.out1 = h->dth0; e1.ld1 = F; e1.mask = h->th0; e1.ldm = F;
        fc_dx(h, "translate/trans_z", h->dZ, B, F, h->Wp("translate/trans_z/Matrix"), F, e1);
        NmPlain2 tcat{src_z, F, h->cz, F, F, 2 * F, B, g_zeros};
        { Side sd(h, LANE_DW);
          fc_dw(h, "translate/trans_h0", tcat, 2 * F, h->dth0, B, F, h->Gp("translate/trans_h0/Matrix"), h->Gp("translate/trans_h0/bias")); }
        Epi e2;   // d concat: cols < F -> d src_z (row block 2 of dZ), cols >= F -> d ctx_z
        e2.out1 = h->dZ + 2ll * B * F; e2.ld1 = F; e2.nsplit = F; e2.out2 = h->dcz; e2.ld2 = F;
        fc_dx(h, "translate/trans_h0", h->dth0, B, F, h->Wp("translate/trans_h0/Matrix"), 2 * F, e2);
    }
    fire_bucket(h, h->find("translate/trans_h0/Matrix"));
    // ---- encoders
    auto encoder_bwd = [&](const std::string& scn, const Scope& sc, const float* x, int nimg, float* const act[5], float* dzp, float* const dA[5],
                           bool with_skips, int dw_lane) {
        const int K3 = h->hh[4] * h->ww[4] * 8 * d;
        { Side sd(h, dw_lane); fc_dw(h, scn + "/hz_lin", nm(act[4], F, F, nimg), F, dzp, nimg, F, sc.gwz, sc.gbz); }
        Epi e4;
        e4.out1 = dA[4]; e4.ld1 = F; e4.mask = act[4]; e4.ldm = F;
        fc_dx(h, scn + "/hz_lin", dzp, nimg, F, sc.wz, F, e4);
        { Side sd(h, dw_lane); fc_dw(h, scn + "/h4_lin", nm(act[3], K3, K3, nimg), K3, dA[4], nimg, F, sc.gw4, sc.gb4); }
        Epi e3;
        e3.out1 = dA[3]; e3.ld1 = K3; e3.mask = act[3]; e3.ldm = K3;
        if (with_skips) { e3.add1 = h->dSk[3]; e3.lda1 = K3; e3.add2 = h->dSk[3] + (int64_t)B * K3; e3.lda2 = K3; }
        fc_dx(h, scn + "/h4_lin", dA[4], nimg, F, sc.w4, K3, e3);
        {   // h4_lin / hz_lin of this encoder (2/3 of its parameters) are done with
            const int64_t lin0 = h->find((scn + "/h4_lin/Matrix").c_str()), lin1 = lin0 + (int64_t)K3 * F + F + (int64_t)F * F + F;
            if (h->dp_in_step) dp_bucket(h, lin0, lin1, dw_lane);
            else if (!h->bucket_fn) adam_early(h, lin0, lin1, dw_lane);
        }
        for (int k = 3; k >= 0; --k) {
            const int hb = h->hh[k], wb = h->ww[k], hs = hb / 2, wsm = wb / 2;
            const int ca = k ? d << (k - 1) : 3, cb = d << k;
            const int R = nimg * hs * wsm;
            const float* xin = k ? act[k - 1] : x;
            const std::string ln = scn + "/h" + std::to_string(k) + "_conv";
            const double fl = 2.0 * R * 25 * ca * cb, uf = tap_frac(hb, wb, 5, 2);
            NmWgradSmall small{dA[k], cb, cb, nullptr, 0, 1, cb, hs * wsm, make_pixdiv(1, hs * wsm).ws_sh, R, g_zeros};
            Epi eg;
            eg.out1 = sc.gw[k]; eg.ld1 = cb;
            if (k == 0) {
                Side sd(h, dw_lane);
                if (!use_dc3(h)) bias_grad(h, ln, dA[k], R, cb, sc.gb[k]);       // (dconv_wgrad returns the column sums of its small operand too)
                if (use_dc3(h)) {
                    DcWgrad Wg{};
                    Wg.big = xin; Wg.ldb = 3; Wg.CA = 3; Wg.s1 = dA[k]; Wg.ld1 = cb; Wg.c1 = cb; Wg.CB = cb;
                    Wg.hb = hb; Wg.wb = wb; Wg.hs = hs; Wg.ws = wsm; Wg.nimg = nimg; Wg.S = 2; Wg.pad = 1; Wg.out = eg.out1;
                    Wg.db = sc.gb[k];
                    ProfScope ps(h, ln + " dw", dw_label(Wg), fl, uf);
                    dconv_wgrad(h->stream, Wg, h->slab, h->slab_floats);
                } else {
                    ProfScope ps(h, ln + " dw", K_C3WGRAD, fl, uf);
                    conv3_wgrad(h->stream, NmC3WgradBig{c4of(h, xin), hb, wb, make_pixdiv(hs, wsm), R, g_zeros}, small, eg, cb, ws_of(h));
                }
                break;   // no gradient w.r.t. the frame
            }
            { Side sd(h, dw_lane);
              bias_grad(h, ln, dA[k], R, cb, sc.gb[k]);
              ProfScope ps(h, ln + " dw", K_WGRAD, fl, uf);
              if (rect_ok(nimg)) {
                  const RectGeo rg = make_rect(nimg, hs, wsm, hb, wb, 2, 1, 5);
                  conv_wgrad_r(h->stream, NmWgradBigR{xin, ca, ca, rg, g_zeros}, NmWgradSmallR{dA[k], cb, cb, rg, g_zeros}, eg, ca, cb, ws_of(h));
              } else if (patch_ok(hs, wsm)) {
                  const PatchGeo pg = make_patch(nimg, hs, wsm);
                  conv_wgrad_p(h->stream, NmWgradBigP{xin, ca, ca, wb, pg, g_zeros}, NmWgradSmallP{dA[k], cb, cb, pg, g_zeros}, eg, ca, cb, ws_of(h));
              } else conv_wgrad(h->stream, NmWgradBig{xin, ca, ca, hb, wb, make_pixdiv(hs, wsm), R, g_zeros}, small, eg, ca, cb, ws_of(h)); }
            // input gradient = conv2d_transpose of dA[k] with the same filter read as [5,5,ca,cb]
            Epi ed;
            ed.out1 = dA[k - 1]; ed.ld1 = ca; ed.mask = act[k - 1]; ed.ldm = ca;
            if (with_skips) {
                ed.add1 = h->dSk[k - 1]; ed.lda1 = ca;
                ed.add2 = h->dSk[k - 1] + (int64_t)B * hb * wb * ca; ed.lda2 = ca;
            }
            // (split-bf16 mode: the exact-f32 kernel only where it is the faster one -- the 16x16 grids' few-channel input gradients)
            const bool wide = (h->cfg.precision == CTX_PREC_F32 || hs == 16) && wconvt_ok(hs, wsm, cb, 0, ca, nimg);
            ProfScope ps(h, ln + " dx", wide ? K_WCONVT : K_CONVT, fl, uf);
            if (wide) wconvt_fwd(h->stream, dA[k], cb, nullptr, 0, 1, nimg, hs, wsm, sc.w[k], ca, ed, ws_of(h));
            else if (use_q(nimg) && hs * wsm >= q_minpos(h)) convt_fwd_q(h->stream, KmConvTGatherQ{dA[k], cb, cb, nullptr, 0, 1, make_tposgeo(hs, wsm, 5, 1, cb / KC), nimg, g_zeros},
                                         KmConvTWeightsQ{sc.w[k], ca, cb, 5, g_zeros}, ed, ca, ws_of(h));
            else convt_fwd(h->stream, KmConvTGather{dA[k], cb, cb, nullptr, 0, 1, hs, wsm, cb / KC, R, g_zeros}, KmConvTWeights{sc.w[k], ca, cb, cb / KC, g_zeros},
                           ed, R, ca, ws_of(h));
        }
    };
    // `conv` on [tgt | src]: code gradients are rows [B, 3B) of dZ; hz_lin has an lrelu
    float* dSz = h->dZ + (int64_t)B * F;
    const bool lanes = use_lanes(h);
    if (lanes) {
        // `conv_context` (linear hz_lin; its h0..h3 also fed both decoder passes as skips) on the second lane:
        // everything it reads (dcz, dSk[*], c[*]) was produced before this point
        fork(h, LANE_CTX);
        LaneSwap sw(h, LANE_CTX);
        encoder_bwd("conv_context", scope_of(h, "conv_context"), h->img + 2 * B * npi, B, h->c, h->dcz, h->dC, true, -1);
    }
    { ProfScope ps(h, "conv/hz_lin lrelu'", K_EW, 0.0); lrelu_mask(h->stream, dSz, tgt_z, 2ll * B * F); }
    encoder_bwd("conv", scope_of(h, "conv"), h->img, 2 * B, h->s, dSz, h->dS, false, LANE_DW);
    if (lanes) { join(h, LANE_CTX); join(h, LANE_DW); }
    else encoder_bwd("conv_context", scope_of(h, "conv_context"), h->img + 2 * B * npi, B, h->c, h->dcz, h->dC, true, -1);
    h->have_grads = true;

}

// The reward hook's fetches at small batch are launch-bound (9-40 launches for well under 100 us of GPU work at B = 25):
// the forward of a given (mode, B) is captured into a hipGraph on its second call and replayed afterwards (translate at
// B = 25: 1.7 -> 0.9 ms per call).  All buffers are owned by the handle, so the captured pointers stay valid; parameters are
// read through the arena pointer at replay.  CTX_GRAPHS=0 keeps plain launches.
int forward_inference(ctx_handle* h, int B, Mode mode) {
    if (!h->use_graphs || h->prof_on || B > 64) { forward(h, B, mode); return CTX_OK; }
    ctx_handle::GraphSlot& g = h->graphs[(int)mode * (1 << 20) + (mode == MODE_TRANSLATE && h->ctx_single ? 1 << 19 : 0) + B];
    // A graph captured while every packed filter it uses was stale holds all its pack nodes ("self-packing": right after a training step)
    // and is valid for any later parameters -- they are read through the arena pointer at replay.  One captured on current entries
    // skips the packs: after a parameter change it is dropped and re-captured AT ONCE (the entries are stale now, so the new graph is
    // self-packing): a loop that alternates training steps and reward calls replays graphs instead of falling back to plain launches.
    if (g.exec && !g.self_packing && h->pack.n && g.pack_version != h->pack.version) {
        (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        g.calls = 1;
    }
    if (g.calls++ == 0) { forward(h, B, mode); return CTX_OK; }      // first call: plain (code objects, LDS limits, filter packs)
    if (!g.exec) {
        hipGraph_t graph = nullptr;
        const uint64_t hits0 = h->pack.hits;
        h->capturing = true;                                         // (lanes inside the capture: option graph_lanes)
        hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            forward(h, B, mode);
            e = hipStreamEndCapture(h->stream, &graph);
        }
        h->capturing = false;
        if (e == hipSuccess && graph) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        g.pack_version = h->pack.version;
        g.self_packing = h->pack.hits == hits0;
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess || !g.exec) {                            // capture not possible here: stay on plain launches
            (void)hipGetLastError();
            g.exec = nullptr;
            h->use_graphs = false;
            h->pack.version++;                                       // entries the failed capture stamped "packed" exist as dropped graph nodes only
            forward(h, B, mode);
            return CTX_OK;
        }
    }
    HIP_TRY(h, hipGraphLaunch(g.exec, h->stream));
    return CTX_OK;
}

int check_B(ctx_handle* h, int B) {
    if (!h) return CTX_E_INVALID;
    if (B <= 0 || B > h->Bm) return fail(h, CTX_E_INVALID, "B=%d outside [1, max_batch=%d]", B, h->Bm);
    return CTX_OK;
}

// Device -> pageable host, in pieces of 16 MiB so that no single transfer leaves the runtime's staged path.  (The "B = 1000
// cliff" of ctx_encode -- 49 MB of float frames handed back -- turned out NOT to be this copy: it was the caller's fresh > 32 MB
// numpy array faulting its pages in while the copy landed; profiles/archive/round2_b_encode_cliff.txt, Translator.encode(out=...).)
int copy_d2h(ctx_handle* h, void* dst, const void* src, size_t bytes) {
    constexpr size_t PIECE = 16u << 20;
    for (size_t o = 0; o < bytes; o += PIECE)
        HIP_TRY(h, hipMemcpyAsync((char*)dst + o, (const char*)src + o, bytes - o < PIECE ? bytes - o : PIECE, hipMemcpyDeviceToHost, h->stream));
    return CTX_OK;
}

int finish(ctx_handle* h) {
    { char msg[256]; if (take_launch_error(msg, sizeof msg)) return fail(h, CTX_E_DEVICE, "%s", msg); }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CTX_OK;
}

int adam_step(ctx_handle* h, float lr) {
    if (!h->have_grads) return fail(h, CTX_E_STATE, "ctx_dev_adam before any backward");
    adam_begin(h, lr);
    h->adam_early_on = false;      // (called after the backward: one launch over the whole arena)
    adam_end(h);
    return CTX_OK;
}

// forward + backward + Adam on the frames in h->img, Adam sliced beside the backward (adam_early)
int fused_step(ctx_handle* h, int B, float lr) {
    h->drop_on = true;      // (dropout belongs to the training graph only)
    forward(h, B, MODE_TRAIN);
    adam_begin(h, lr);
    backward(h, B, B);
    h->drop_on = false;
    adam_end(h);
    return CTX_OK;
}

// host f32 frames -> img slots [tgt | src | ctx]
int upload_f32(ctx_handle* h, const float* src, const float* ctxf, const float* tgt, int B) {
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    HIP_TRY(h, hipMemcpyAsync(h->img, tgt, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->img + B * h->npi, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->img + 2 * B * h->npi, ctxf, bytes, hipMemcpyHostToDevice, h->stream));
    return CTX_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
// ================================================================================================
// RCCL behind the C ABI (ctx_dp_init / ctx_dp_allreduce_grads).  The reference has no multi-GPU path; this is
// the path's one exchange step (8e): SUM all-reduce of the flat f32 gradient arena between backward and Adam.  librccl is
// dlopen()ed on first use -- in a process that already holds one (PyTorch ships its own librccl.so.1) the SAME copy is
// shared, a plain C/C++ host gets the system's -- so libctxtrans.so keeps loading on boxes without RCCL.
// ================================================================================================
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi& rccl() { static RcclApi api; return api; }

bool rccl_load() {
    RcclApi& a = rccl();
    if (a.lib) return true;
    std::vector<std::string> names;
    if (const char* e = getenv("CTX_RCCL_LIB")) names.push_back(e);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
    for (const std::string& n : names) {
        a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
        a.err = dlerror();
    }
    if (!a.lib) return false;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p) { ok = false; a.err = std::string("missing symbol ") + n; } return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(a.lib); a.lib = nullptr; }
    return ok;
}

#define RCCL_TRY(h, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(h, CTX_E_DEVICE, "%s: %s", #expr, rccl().GetErrorString(r_));    \
    } while (0)

void dp_teardown(ctx_handle* h) {
    if (h->dp_stream) (void)hipStreamSynchronize(h->dp_stream);      // no collective may still be in flight when its communicator goes
    if (h->dp_comm && rccl().lib) (void)rccl().CommDestroy(h->dp_comm);
    h->dp_comm = nullptr;
    if (h->dp_stream) { (void)hipStreamDestroy(h->dp_stream); h->dp_stream = nullptr; }
    if (h->dp_ev_ready) { (void)hipEventDestroy(h->dp_ev_ready); h->dp_ev_ready = nullptr; }
    if (h->dp_ev_done) { (void)hipEventDestroy(h->dp_ev_done); h->dp_ev_done = nullptr; }
}

// SUM all-reduce of grads[first, first + count) on the collective stream, ordered after everything the compute stream has
// queued so far.  The compute stream is NOT made to wait here (dp_wait does that), so the collective overlaps what follows.
// Returns CTX_OK or a CTX_E_* code (message in h->err): a failed collective must never let Adam run on un-reduced gradients.
int dp_reduce_range(ctx_handle* h, int64_t first, int64_t count) {
    if (count <= 0) return CTX_OK;
    float* g = h->arena + h->Ppad + first;
    HIP_TRY(h, hipEventRecord(h->dp_ev_ready, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->dp_stream, h->dp_ev_ready, 0));
    RCCL_TRY(h, rccl().AllReduce(g, g, (size_t)count, ncclFloat, ncclSum, h->dp_comm, h->dp_stream));
    return CTX_OK;
}
int dp_wait(ctx_handle* h) {
    HIP_TRY(h, hipEventRecord(h->dp_ev_done, h->dp_stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->dp_ev_done, 0));
    return CTX_OK;
}
}  // namespace

extern "C" {

int ctx_abi_version(void) { return CTX_ABI_VERSION; }

int64_t ctx_param_total_for(const ctx_config* cfg) {
    if (check_cfg(cfg, nullptr) != CTX_OK) return CTX_E_INVALID;
    std::vector<ParamInfo> ps;
    int64_t total = 0;
    if (cfg->variant != CTX_VARIANT_SKIPNEW) {
        GenState r;
        int64_t pp = 0;
        gen_layout(*cfg, r, ps, total, pp);
    } else build_params(*cfg, ps, total);
    return total;
}

int64_t ctx_arena_bytes(const ctx_config* cfg) {
    if (check_cfg(cfg, nullptr) != CTX_OK) return CTX_E_INVALID;
    if (cfg->variant != CTX_VARIANT_SKIPNEW) {   // the arena holds the zero-padded parameters
        GenState r;
        std::vector<ParamInfo> ps;
        int64_t total = 0, pp = 0;
        gen_layout(*cfg, r, ps, total, pp);
        return 4 * pp * (int64_t)sizeof(float);
    }
    const int64_t p = ctx_param_total_for(cfg);
    return p < 0 ? p : 4 * round_up(p, 64) * (int64_t)sizeof(float);
}

int ctx_create_ex(const ctx_config* cfg, int device, void* stream, void* arena, ctx_handle** out) {
    if (!out) return fail(nullptr, CTX_E_INVALID, "out is NULL");
    *out = nullptr;
    TRY(check_cfg(cfg, nullptr));
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CTX_E_DEVICE, "no HIP device available (%s); libctxtrans has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, CTX_E_INVALID, "device %d out of range [0,%d)", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, CTX_E_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    ctx_handle* h = new ctx_handle();
    h->cfg = *cfg;
    h->opt = options_from_env();             // CTX_<NAME> in the environment = this handle's defaults; ctx_set_option changes them
    OptScope os(&h->opt);
    h->device = device;
    h->H = cfg->H; h->W = cfg->W; h->d = cfg->df_dim; h->F = cfg->featsize; h->Bm = cfg->max_batch;
    for (int k = 0; k < 5; ++k) { h->hh[k] = cfg->H >> k; h->ww[k] = cfg->W >> k; }
    h->npi = (int64_t)cfg->H * cfg->W * cfg->C;
    h->D0 = (int64_t)8 * h->d * h->hh[4] * h->ww[4];
    h->Fp = h->F;
    if (cfg->variant != CTX_VARIANT_SKIPNEW) {
        h->gen = new GenState();
        gen_layout(*cfg, *h->gen, h->params, h->P, h->Ppad);
        h->Fp = h->gen->Fp;
    } else {
        build_params(*cfg, h->params, h->P);
        h->Ppad = round_up(h->P, 64);
        for (auto& p : h->params)
            if (p.offset % 4) { delete h; return fail(nullptr, CTX_E_INVALID, "parameter %s not 16-byte aligned in the arena", p.name.c_str()); }
    }
    int rc = CTX_OK;
    if (stream) h->stream = (hipStream_t)stream;
    else {
        e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { rc = fail(nullptr, CTX_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    if (rc == CTX_OK) {
        if (arena) h->arena = (float*)arena;
        else {
            rc = dev_alloc(h, &h->arena, 4 * h->Ppad, false);     // addressed per parameter tensor, not as a whole
            h->own_arena = true;
        }
    }
    if (rc == CTX_OK) rc = h->gen ? gen_alloc(h) : alloc_buffers(h);
    // packed-filter cache (dconv.h: DcPackCache): only where the library owns the parameters -- a caller-owned arena (ctx_create_ex) may be
    // written behind the handle's back, there every launch packs as before
    if (rc == CTX_OK && h->own_arena) {
        h->pack.floats = 4ll << 20;
        rc = dev_alloc(h, &h->pack.arena, h->pack.floats, false);
    }
    if (rc == CTX_OK) {
        // -1 = by size: the lanes pay where the launches are long enough to hide a cross-queue hop (measured 11.5 us each; a step has ~25
        // of them).  On ContextAEInception2's 2x2 maps they gain 0.07 ms of 2.7 alone and LOSE 0.23 ms of 6.85 behind the front end on a
        // caller's stream, where lane and compute stream came to share a hardware queue (profiles/archive/round4_e_config4_lanes.txt).
        if (h->opt.v[OPT_OVERLAP] < 0) h->opt.v[OPT_OVERLAP] = !(h->gen && h->H * h->W < 64);
        h->overlap = h->opt.v[OPT_OVERLAP] != 0;
        h->use_graphs = h->opt.v[OPT_GRAPHS] != 0;
        // Side-lane stream priority: NORMAL.  (Lowest was -0.03 ms on the ContextSkipNew step and -0.7 ms on the split-bf16 config-4
        // step, but the f32 config-4 step -- front end chained on the same stream -- went from 7.8 to 17.4 ms with it; highest +0.08 ms.)
        const int lane_prio = 0;
        for (int l = 0; l < ctx_handle::NLANE && rc == CTX_OK; ++l)
            if (hipStreamCreateWithPriority(&h->aux[l], hipStreamNonBlocking, lane_prio) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_fork[l], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming) != hipSuccess)
                rc = fail(h, CTX_E_DEVICE, "side-lane stream/event creation failed");
    }
    // The Adam stream in its own PRIORITY class (option adam_prio, default 2 = low for exact-f32 handles, normal for split-bf16 ones, whose
    // step -- in a process that also holds an f32 handle -- went from 8.0 to 9.2 ms with it: which streams share a queue is the runtime's choice): a priority class has its own hardware queues, so the
    // slices of the early update no longer take turns with the filter-gradient lane on a shared queue (step -0.03..-0.05 ms in two A/B
    // pairs, profiles/archive/round4_e_early_adam_queues.txt).  Its launches are 80-230 us HBM-bound kernels: the slowdown seen with prioritised
    // LANES (5 us kernels beside another class's) does not apply.
    if (h->opt.v[OPT_ADAM_PRIO] == 2) h->opt.v[OPT_ADAM_PRIO] = h->cfg.precision == CTX_PREC_F32 ? 1 : 0;     // (2 = by precision; reads back resolved)
    if (rc == CTX_OK && (hipStreamCreateWithPriority(&h->adam_stream, hipStreamNonBlocking, h->opt.v[OPT_ADAM_PRIO]) != hipSuccess ||
                         hipEventCreateWithFlags(&h->adam_ev[0], hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&h->adam_ev[1], hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&h->adam_ev_done, hipEventDisableTiming) != hipSuccess))
        rc = fail(h, CTX_E_DEVICE, "Adam stream/event creation failed");
    if (rc == CTX_OK) {
        e = hipMemsetAsync(h->arena + h->Ppad, 0, 3 * h->Ppad * sizeof(float), h->stream);   // grads, m, v
        if (e == hipSuccess && h->own_arena) e = hipMemsetAsync(h->arena, 0, h->Ppad * sizeof(float), h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, CTX_E_DEVICE, "arena init: %s", hipGetErrorString(e));
    }
    if (rc != CTX_OK) {
        g_create_error = h->err.empty() ? g_create_error : h->err;
        ctx_destroy(h);
        return rc;
    }
    *out = h;
    return CTX_OK;
}

int ctx_create(const ctx_config* cfg, int device, ctx_handle** out) { return ctx_create_ex(cfg, device, nullptr, nullptr, out); }

void ctx_destroy(ctx_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto& rc : h->rcache) { if (rc.means) (void)hipFree(rc.means); if (rc.imgs) (void)hipFree(rc.imgs); }
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->vdata) (void)hipFree(h->vdata);
    if (h->dp_host_buf) (void)hipFree(h->dp_host_buf);
    for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    delete h->gen;
    dp_teardown(h);
    for (int l = 0; l < ctx_handle::NLANE; ++l) {
        if (h->aux[l]) { (void)hipStreamSynchronize(h->aux[l]); (void)hipStreamDestroy(h->aux[l]); }
        if (h->ev_fork[l]) (void)hipEventDestroy(h->ev_fork[l]);
        if (h->ev_join[l]) (void)hipEventDestroy(h->ev_join[l]);
    }
    if (h->adam_stream) { (void)hipStreamSynchronize(h->adam_stream); (void)hipStreamDestroy(h->adam_stream); }
    for (hipEvent_t e : {h->adam_ev[0], h->adam_ev[1], h->adam_ev_done}) if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* ctx_last_error(const ctx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int64_t ctx_param_total(const ctx_handle* h) { return h ? h->P : CTX_E_INVALID; }
int ctx_param_count(const ctx_handle* h) { return h ? (int)h->params.size() : CTX_E_INVALID; }

int ctx_param_info(const ctx_handle* h, int index, const char** name, int* ndim, int64_t shape[4], int64_t* offset) {
    if (!h || index < 0 || index >= (int)h->params.size()) return CTX_E_INVALID;
    const ParamInfo& p = h->params[index];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (shape) memcpy(shape, p.shape, sizeof p.shape);
    if (offset) *offset = p.offset;
    return CTX_OK;
}

static int arena_io(ctx_handle* h, int slot, float* host, const float* chost, size_t n) {
    if (!h) return CTX_E_INVALID;
    if (chost && slot == 0) h->pack.version++;
    if ((int64_t)n != h->P) return fail(h, CTX_E_INVALID, "expected %lld floats, got %zu", (long long)h->P, n);
    HIP_TRY(h, hipSetDevice(h->device));
    float* dev = h->arena + slot * h->Ppad;
    if (h->gen) {   // scatter / gather between the TF-shaped vector and the (possibly zero-padded) arena
        std::vector<float> padded((size_t)h->Ppad, 0.f);
        const std::vector<int32_t>& map = h->gen->real2pad;
        if (!chost) {
            HIP_TRY(h, hipMemcpyAsync(padded.data(), dev, padded.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
            TRY(finish(h));
        }
        for (const GenState::Seg& sg : h->gen->segs) {
            if (sg.map0 < 0) {
                if (chost) memcpy(padded.data() + sg.poff, chost + sg.roff, (size_t)sg.size * sizeof(float));
                else memcpy(host + sg.roff, padded.data() + sg.poff, (size_t)sg.size * sizeof(float));
            } else if (chost) {
                for (int64_t i = 0; i < sg.size; ++i) padded[(size_t)map[(size_t)(sg.map0 + i)]] = chost[sg.roff + i];
            } else {
                for (int64_t i = 0; i < sg.size; ++i) host[sg.roff + i] = padded[(size_t)map[(size_t)(sg.map0 + i)]];
            }
        }
        if (chost) {
            HIP_TRY(h, hipMemcpyAsync(dev, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
            return finish(h);
        }
        return CTX_OK;
    }
    if (chost) HIP_TRY(h, hipMemcpyAsync(dev, chost, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    else TRY(copy_d2h(h, host, dev, n * sizeof(float)));
    return finish(h);
}

int ctx_set_params(ctx_handle* h, const float* flat, size_t n) { return flat ? arena_io(h, 0, nullptr, flat, n) : CTX_E_INVALID; }
int ctx_get_params(ctx_handle* h, float* flat, size_t n) { return flat ? arena_io(h, 0, flat, nullptr, n) : CTX_E_INVALID; }
int ctx_get_grads(ctx_handle* h, float* flat, size_t n) { return flat ? arena_io(h, 1, flat, nullptr, n) : CTX_E_INVALID; }

int ctx_set_adam_state(ctx_handle* h, const float* m, const float* v, size_t n, int64_t step) {
    if (!h || !m || !v || step < 0) return CTX_E_INVALID;
    TRY(arena_io(h, 2, nullptr, m, n));
    TRY(arena_io(h, 3, nullptr, v, n));
    h->adam_t = step;
    return CTX_OK;
}

int ctx_get_adam_state(ctx_handle* h, float* m, float* v, size_t n, int64_t* step) {
    if (!h) return CTX_E_INVALID;
    if (m) TRY(arena_io(h, 2, m, nullptr, n));
    if (v) TRY(arena_io(h, 3, v, nullptr, n));
    if (step) *step = h->adam_t;
    return CTX_OK;
}

int ctx_init_params(ctx_handle* h, uint64_t seed) {
    if (!h) return CTX_E_INVALID;
    std::vector<float> host((size_t)h->P, 0.f);
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> nd(0.0, 1.0);
    for (auto& p : h->params) {
        const bool bias = p.ndim == 1;
        const bool truncated = p.name.find("_conv/w") != std::string::npos;   // arm_shaping.py:25-26
        if (bias) continue;
        for (int64_t i = 0; i < p.size; ++i) {
            double x = nd(rng);
            if (truncated) while (std::fabs(x) > 2.0) x = nd(rng);
            host[(size_t)(p.offset + i)] = (float)(0.02 * x);
        }
    }
    TRY(ctx_set_params(h, host.data(), host.size()));
    HIP_TRY(h, hipMemsetAsync(h->arena + 2 * h->Ppad, 0, 2 * h->Ppad * sizeof(float), h->stream));
    h->adam_t = 0;
    return finish(h);
}

// ContextAEInception2's `out = decode + tgtctx` (arm_shaping.py:1890-1891)
static bool residual_out(const ctx_handle* h) { return h->gen && h->gen->residual; }

static int translate_tail(ctx_handle* h, int B, float* pred, float* feat) {
    TRY(forward_inference(h, B, MODE_TRANSLATE));
    if (pred) TRY(copy_d2h(h, pred, h->out, (size_t)B * h->npi * sizeof(float)));
    if (feat) HIP_TRY(h, hipMemcpy2DAsync(feat, h->F * sizeof(float), h->Z, h->Fp * sizeof(float), h->F * sizeof(float), B,
                                         hipMemcpyDeviceToHost, h->stream));
    h->last_B = 0;
    return finish(h);
}

// the same two fetches on float inputs: frames already in [-1, 1], or Inception feature maps (CTX_VARIANT_INCEPTION2)
int ctx_translate_f32(ctx_handle* h, const float* src, const float* ctx0, int ctx_batched, int B, float* pred, float* feat) {
    TRY(check_B(h, B));
    if (!src || !ctx0) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t npi = h->npi;
    HIP_TRY(h, hipMemcpyAsync(h->img + B * npi, src, (size_t)B * npi * sizeof(float), hipMemcpyHostToDevice, h->stream));
    // one context frame: encoded once, read by every row (forward); B frames: one per row.  (translate reads nothing of the tgt slot)
    h->ctx_single = !ctx_batched;
    HIP_TRY(h, hipMemcpyAsync(h->img + 2ll * B * npi, ctx0, (size_t)(ctx_batched ? B : 1) * npi * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (!ctx_batched && residual_out(h))           // out = decode + tgtctx reads the context frame of every row
        for (int b = 1; b < B; ++b) HIP_TRY(h, hipMemcpyAsync(h->img + (2ll * B + b) * npi, ctx0, (size_t)npi * sizeof(float), hipMemcpyHostToDevice, h->stream));
    return translate_tail(h, B, pred, feat);
}

// The same two fetches with the inputs already on the DEVICE (the Inception front end's output buffer): results to the host.
int ctx_translate_dev(ctx_handle* h, const float* d_src, const float* d_ctx0, int ctx_batched, int B, float* pred, float* feat) {
    TRY(check_B(h, B));
    if (!d_src || !d_ctx0) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t npi = h->npi;
    HIP_TRY(h, hipMemcpyAsync(h->img + B * npi, d_src, (size_t)B * npi * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    h->ctx_single = !ctx_batched;
    HIP_TRY(h, hipMemcpyAsync(h->img + 2ll * B * npi, d_ctx0, (size_t)(ctx_batched ? B : 1) * npi * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (!ctx_batched && residual_out(h))
        for (int b = 1; b < B; ++b) HIP_TRY(h, hipMemcpyAsync(h->img + (2ll * B + b) * npi, d_ctx0, (size_t)npi * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return translate_tail(h, B, pred, feat);
}

int ctx_encode_dev(ctx_handle* h, const float* d_frames, int B, float* feat) {
    TRY(check_B(h, B));
    if (!d_frames) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->img + B * h->npi, d_frames, (size_t)B * h->npi * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    TRY(forward_inference(h, B, MODE_ENCODE));
    if (feat) HIP_TRY(h, hipMemcpy2DAsync(feat, h->F * sizeof(float), h->Z + 2ll * B * h->Fp, h->Fp * sizeof(float), h->F * sizeof(float), B,
                                         hipMemcpyDeviceToHost, h->stream));
    h->last_B = 0;
    return finish(h);
}

int ctx_encode_f32(ctx_handle* h, const float* frames, int B, float* feat) {
    TRY(check_B(h, B));
    if (!frames) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->img + B * h->npi, frames, (size_t)B * h->npi * sizeof(float), hipMemcpyHostToDevice, h->stream));
    TRY(forward_inference(h, B, MODE_ENCODE));
    if (feat) HIP_TRY(h, hipMemcpy2DAsync(feat, h->F * sizeof(float), h->Z + 2ll * B * h->Fp, h->Fp * sizeof(float), h->F * sizeof(float), B,
                                         hipMemcpyDeviceToHost, h->stream));
    h->last_B = 0;
    return finish(h);
}

static int need_frames(ctx_handle* h) {
    if (h && h->cfg.variant == CTX_VARIANT_INCEPTION2)
        return fail(h, CTX_E_INVALID, "uint8 frames need the Inception-v3 front end (not built): pass Mixed_7c feature maps to the _f32 entry points");
    return CTX_OK;
}

int ctx_translate(ctx_handle* h, const uint8_t* src, const uint8_t* ctx0, int ctx_batched, int B, float* pred, float* feat) {
    TRY(check_B(h, B));
    TRY(need_frames(h));
    if (!src || !ctx0) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t npi = h->npi;
    uint8_t* u_src = h->u8;
    uint8_t* u_ctx = h->u8 + B * npi;
    HIP_TRY(h, hipMemcpyAsync(u_src, src, (size_t)B * npi, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(u_ctx, ctx0, (size_t)(ctx_batched ? B : 1) * npi, hipMemcpyHostToDevice, h->stream));
    u8_to_f32(h->stream, u_src, h->img + B * npi, B * npi);
    h->ctx_single = !ctx_batched;                 // one context frame: encoded once, read by every row (forward)
    u8_to_f32(h->stream, u_ctx, h->img + 2 * B * npi, (ctx_batched ? B : 1) * npi);
    return translate_tail(h, B, pred, feat);
}

int ctx_encode(ctx_handle* h, const uint8_t* frames, int B, float* feat, float* frames_f32) {
    TRY(check_B(h, B));
    TRY(need_frames(h));
    if (!frames) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t npi = h->npi;
    HIP_TRY(h, hipMemcpyAsync(h->u8, frames, (size_t)B * npi, hipMemcpyHostToDevice, h->stream));
    u8_to_f32(h->stream, h->u8, h->img + B * npi, B * npi);
    TRY(forward_inference(h, B, MODE_ENCODE));
    if (feat) HIP_TRY(h, hipMemcpy2DAsync(feat, h->F * sizeof(float), h->Z + 2ll * B * h->Fp, h->Fp * sizeof(float), h->F * sizeof(float), B,
                                         hipMemcpyDeviceToHost, h->stream));
    if (frames_f32) TRY(copy_d2h(h, frames_f32, h->img + B * npi, (size_t)B * npi * sizeof(float)));
    h->last_B = 0;
    return finish(h);
}

int ctx_reward_set_cache(ctx_handle* h, int vp, const float* means, const float* imgs, int bs) {
    if (!h) return CTX_E_INVALID;
    if (vp < 0 || vp >= 64 || !means || !imgs || bs <= 0 || bs > h->Bm) return fail(h, CTX_E_INVALID, "bad viewpoint / batch_size");
    if (h->cfg.variant == CTX_VARIANT_INCEPTION2) return fail(h, CTX_E_INVALID, "the device reward path takes frames, not feature maps");
    HIP_TRY(h, hipSetDevice(h->device));
    if ((int)h->rcache.size() <= vp) h->rcache.resize(vp + 1);
    ctx_handle::RewardCache& rc = h->rcache[vp];
    if (rc.means) { (void)hipFree(rc.means); (void)hipFree(rc.imgs); rc.means = rc.imgs = nullptr; }
    const size_t nm = (size_t)bs * h->F * sizeof(float), ni = (size_t)bs * h->npi * sizeof(float);
    if (hipMalloc((void**)&rc.means, nm) != hipSuccess || hipMalloc((void**)&rc.imgs, ni) != hipSuccess) return fail(h, CTX_E_NOMEM, "reward cache");
    rc.bs = bs;
    HIP_TRY(h, hipMemcpyAsync(rc.means, means, nm, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(rc.imgs, imgs, ni, hipMemcpyHostToDevice, h->stream));
    if (!h->rcosts) TRY(dev_alloc(h, &h->rcosts, h->Bm));
    return finish(h);
}

int ctx_reward_costs(ctx_handle* h, int vp, const uint8_t* frames, int npaths, float scale, int ablation, float* costs) {
    if (!h) return CTX_E_INVALID;
    if (vp < 0 || vp >= (int)h->rcache.size() || !h->rcache[vp].means) return fail(h, CTX_E_STATE, "ctx_reward_set_cache(vp = %d) first", vp);
    const ctx_handle::RewardCache& rc = h->rcache[vp];
    if (!frames || !costs || npaths <= 0 || ablation < 0 || ablation > 2) return fail(h, CTX_E_INVALID, "bad argument");
    const int B = npaths * rc.bs;
    TRY(check_B(h, B));
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t npi = h->npi;
    HIP_TRY(h, hipMemcpyAsync(h->u8, frames, (size_t)B * npi, hipMemcpyHostToDevice, h->stream));
    u8_to_f32(h->stream, h->u8, h->img + B * npi, B * npi);               // image_trans[0], base.py:116-119
    if (ablation != 1) TRY(forward_inference(h, B, MODE_ENCODE));         // input_z: the `conv` encoder on the frames (base.py:234-235)
    reward_costs(h->stream, h->Z + 2ll * B * h->Fp, h->Fp, h->F, h->img + B * npi, npi, rc.means, rc.imgs, rc.bs, B, scale, ablation, h->rcosts);
    HIP_TRY(h, hipMemcpyAsync(costs, h->rcosts, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    h->last_B = 0;
    return finish(h);
}

// d_src / d_ctx / d_tgt -> the handle's frame buffer [tgt | src | ctx]; a slot the caller filled IN PLACE (pointers of ctx_dev_frames) is not copied
static int stage_frames(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B) {
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    const float* from[3] = {d_tgt, d_src, d_ctx};
    for (int k = 0; k < 3; ++k) {
        float* slot = h->img + (int64_t)k * B * h->npi;
        if (from[k] != slot) HIP_TRY(h, hipMemcpyAsync(slot, from[k], bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    return CTX_OK;
}

int ctx_dev_forward(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B) {
    TRY(check_B(h, B));
    if (!d_src || !d_ctx || !d_tgt) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(stage_frames(h, d_src, d_ctx, d_tgt, B));
    forward(h, B, MODE_TRAIN);
    losses(h->stream, h->out, h->img, nullptr, h->npi, B, h->Z, h->Z + (int64_t)B * h->Fp, nullptr, h->Fp, B, h->scratch, h->scalars, h->F, loss_terms_of(h));
    h->last_B = B;
    HIP_TRY(h, hipGetLastError());
    return CTX_OK;
}

int ctx_dev_frames(ctx_handle* h, int B, float** d_src, float** d_ctx, float** d_tgt) {
    TRY(check_B(h, B));
    if (d_tgt) *d_tgt = h->img;
    if (d_src) *d_src = h->img + (int64_t)B * h->npi;
    if (d_ctx) *d_ctx = h->img + 2ll * B * h->npi;
    return CTX_OK;
}

int ctx_dev_forward_backward(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B, int sim_batch) {
    TRY(check_B(h, B));
    if (!d_src || !d_ctx || !d_tgt) return fail(h, CTX_E_INVALID, "NULL input");
    if (sim_batch < 0) return fail(h, CTX_E_INVALID, "sim_batch < 0");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(stage_frames(h, d_src, d_ctx, d_tgt, B));
    h->drop_on = true;      // (dropout belongs to the training graph only)
    forward(h, B, MODE_TRAIN);
    backward(h, B, sim_batch ? sim_batch : B);
    h->drop_on = false;
    h->last_B = B;
    HIP_TRY(h, hipGetLastError());
    return CTX_OK;
}

int ctx_dev_train_step(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B, float lr) {
    TRY(check_B(h, B));
    if (!d_src || !d_ctx || !d_tgt) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(stage_frames(h, d_src, d_ctx, d_tgt, B));
    TRY(fused_step(h, B, lr));
    h->last_B = B;
    { char msg[256]; if (take_launch_error(msg, sizeof msg)) return fail(h, CTX_E_DEVICE, "%s", msg); }
    HIP_TRY(h, hipGetLastError());
    return CTX_OK;
}

int ctx_set_dropout_seed(ctx_handle* h, uint64_t seed) {
    if (!h) return CTX_E_INVALID;
    h->drop_seed = seed;
    return CTX_OK;
}

int ctx_set_grad_bucket_callback(ctx_handle* h, ctx_bucket_fn fn, void* user) {
    if (!h) return CTX_E_INVALID;
    h->bucket_fn = fn; h->bucket_user = user;
    return CTX_OK;
}

int ctx_dev_adam(ctx_handle* h, float lr) {
    if (!h) return CTX_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(adam_step(h, lr));
    HIP_TRY(h, hipGetLastError());
    return CTX_OK;
}

int ctx_dev_scalars(ctx_handle* h, float scalars[4]) {
    if (!h || !scalars) return CTX_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(scalars, h->scalars, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return finish(h);
}

// The caller may WRITE through this pointer (a custom optimiser, a torch-side broadcast) without the library seeing it: from here on
// the handle packs its direct kernels' filters in front of every launch again (DcPackCache::external) and graphs captured before are dropped.
void* ctx_dev_params(ctx_handle* h) {
    if (!h) return nullptr;
    if (!h->pack.external) { h->pack.external = true; h->pack.version++; }
    return h->arena;
}
void* ctx_dev_grads(ctx_handle* h) { return h ? h->arena + h->Ppad : nullptr; }
void* ctx_dev_scalar_buf(ctx_handle* h) { return h ? h->scalars : nullptr; }
void* ctx_stream(ctx_handle* h) { return h ? (void*)h->stream : nullptr; }

int ctx_sync(ctx_handle* h) {
    if (!h) return CTX_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    return finish(h);
}

int ctx_dev_outputs(ctx_handle* h, const float** out, const float** out2, const float** input_z, const float** translated_z) {
    if (!h) return CTX_E_INVALID;
    if (h->last_B <= 0) return fail(h, CTX_E_STATE, "no training-mode forward has run");
    const int B = h->last_B;
    if (out) *out = h->out;
    if (out2) *out2 = h->out + B * h->npi;
    if (input_z) *input_z = h->Z + 2ll * B * h->Fp;   // row stride Fp (== featsize except for the padded REAL variant)
    if (translated_z) *translated_z = h->Z;
    return CTX_OK;
}

int ctx_last_codes(ctx_handle* h, float* input_z, float* translated_z, int* Bout) {
    if (!h) return CTX_E_INVALID;
    if (h->last_B <= 0) return fail(h, CTX_E_STATE, "no training-mode forward has run");
    const int B = h->last_B;
    if (Bout) *Bout = B;
    if (!input_z && !translated_z) return CTX_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t row = (size_t)h->F * sizeof(float), pitch = (size_t)h->Fp * sizeof(float);
    if (input_z) HIP_TRY(h, hipMemcpy2DAsync(input_z, row, h->Z + 2ll * B * h->Fp, pitch, row, B, hipMemcpyDeviceToHost, h->stream));
    if (translated_z) HIP_TRY(h, hipMemcpy2DAsync(translated_z, row, h->Z, pitch, row, B, hipMemcpyDeviceToHost, h->stream));
    return finish(h);
}

int ctx_train_step(ctx_handle* h, const float* src, const float* ctxf, const float* tgt, int B, float lr, float scalars[4]) {
    TRY(check_B(h, B));
    if (!src || !ctxf || !tgt) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(upload_f32(h, src, ctxf, tgt, B));
    TRY(fused_step(h, B, lr));
    h->last_B = B;
    if (scalars) HIP_TRY(h, hipMemcpyAsync(scalars, h->scalars, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return finish(h);
}

int ctx_train_step_u8(ctx_handle* h, const uint8_t* src, const uint8_t* ctx8, const uint8_t* tgt, int B, float lr, float scalars[4]) {
    TRY(check_B(h, B));
    if (!src || !ctx8 || !tgt) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nb = (size_t)B * h->npi;
    HIP_TRY(h, hipMemcpyAsync(h->u8, tgt, nb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->u8 + nb, src, nb, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->u8 + 2 * nb, ctx8, nb, hipMemcpyHostToDevice, h->stream));
    u8_to_f32(h->stream, h->u8, h->img, 3 * (int64_t)nb);
    TRY(fused_step(h, B, lr));
    h->last_B = B;
    if (scalars) HIP_TRY(h, hipMemcpyAsync(scalars, h->scalars, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return finish(h);
}

int ctx_eval(ctx_handle* h, const float* src, const float* ctxf, const float* tgt, int B, float scalars[4], float* out, float* out2) {
    TRY(check_B(h, B));
    if (!src || !ctxf || !tgt) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(upload_f32(h, src, ctxf, tgt, B));
    forward(h, B, MODE_TRAIN);
    losses(h->stream, h->out, h->img, nullptr, h->npi, B, h->Z, h->Z + (int64_t)B * h->Fp, nullptr, h->Fp, B, h->scratch, h->scalars, h->F, loss_terms_of(h));
    h->last_B = B;
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    if (scalars) HIP_TRY(h, hipMemcpyAsync(scalars, h->scalars, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (out) TRY(copy_d2h(h, out, h->out, bytes));
    if (out2) TRY(copy_d2h(h, out2, h->out + B * h->npi, bytes));
    return finish(h);
}

int ctx_profile_step(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B, float lr, int iters,
                     ctx_prof_entry* entries, int max_entries, int* n_entries) {
    TRY(check_B(h, B));
    if (!d_src || !d_ctx || !d_tgt || iters <= 0 || !n_entries) return fail(h, CTX_E_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    for (auto& m : h->prof_ms) m = 0.0;
    for (int it = 0; it < iters; ++it) {
        HIP_TRY(h, hipMemcpyAsync(h->img, d_tgt, bytes, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->img + B * h->npi, d_src, bytes, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->img + 2 * B * h->npi, d_ctx, bytes, hipMemcpyDeviceToDevice, h->stream));
        h->prof_on = true;
        h->prof_cursor = 0;
        h->drop_on = true;      // (dropout belongs to the training graph only)
        forward(h, B, MODE_TRAIN);
        backward(h, B, B);
        h->drop_on = false;
        const int rc = adam_step(h, lr);
        h->prof_on = false;
        if (rc != CTX_OK) return rc;
        TRY(finish(h));
        for (int i = 0; i < h->prof_cursor; ++i) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
            h->prof_ms[i] += ms;
        }
    }
    h->last_B = B;
    *n_entries = h->prof_cursor;
    for (int i = 0; i < h->prof_cursor && i < max_entries; ++i) {
        entries[i] = h->prof_entries[i];
        entries[i].ms = (float)(h->prof_ms[i] / iters);
    }
    return CTX_OK;
}

// ---- per-handle options (csrc/options.h) ---------------------------------------------------------------------------------------
int ctx_option_count(void) { return OPT_COUNT; }
const char* ctx_option_name(int index) { return opt_name(index); }
int ctx_get_option(const ctx_handle* h, const char* name, int* value) {
    if (!h || !value) return CTX_E_INVALID;
    const int i = opt_find(name);
    if (i < 0) return CTX_E_INVALID;
    *value = h->opt.v[i];
    return CTX_OK;
}
int ctx_set_option(ctx_handle* h, const char* name, int value) {
    if (!h) return CTX_E_INVALID;
    const int i = opt_find(name);
    if (i < 0) return fail(h, CTX_E_INVALID, "unknown option '%s'", name ? name : "(null)");
    // options that decided the handle's buffers, kernel parameter layouts or streams at ctx_create (and the ctx_cnn handles' switches,
    // which a translator handle never reads) cannot change afterwards: refusing beats a silent no-op that reads back as set
    const bool create_only = i == OPT_DIRECT3 || i == OPT_DCONV || i == OPT_ADAM_PRIO || i == OPT_CNN_LANES || i == OPT_CNN_DCONV || i == OPT_CNN_STEM4;
    if (create_only && value != h->opt.v[i]) {
        char up[32] = {};
        const char* nm = opt_name(i);
        for (size_t c = 0; nm[c] && c + 1 < sizeof up; ++c) up[c] = (char)toupper((unsigned char)nm[c]);
        return fail(h, CTX_E_STATE, "option '%s' is fixed at ctx_create (it decides buffers, layouts or streams): set CTX_%s in the environment before creating the handle", nm, up);
    }
    if (i == OPT_OVERLAP && value < 0) value = !(h->gen && h->H * h->W < 64);      // -1 = by size, resolved as ctx_create does; reads back 0 / 1
    h->opt.v[i] = value;
    if (i == OPT_OVERLAP) h->overlap = value != 0;
    if (i == OPT_GRAPHS) {
        h->use_graphs = value != 0;
        for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        h->graphs.clear();
    }
    if (i != OPT_TRACE_LAUNCH && i != OPT_EARLY_ADAM) {      // anything that changes which kernels a captured forward holds
        for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        h->graphs.clear();
    }
    return CTX_OK;
}

int ctx_dp_unique_id(uint8_t id[CTX_DP_UNIQUE_ID_BYTES]) {
    if (!id) return fail(nullptr, CTX_E_INVALID, "id is NULL");
    if (!rccl_load()) return fail(nullptr, CTX_E_DEVICE, "librccl could not be loaded: %s", rccl().err.c_str());
    static_assert(sizeof(ncclUniqueId) == CTX_DP_UNIQUE_ID_BYTES, "unique-id blob size");
    ncclUniqueId u;
    ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, CTX_E_DEVICE, "ncclGetUniqueId: %s", rccl().GetErrorString(r));
    memcpy(id, &u, sizeof u);
    return CTX_OK;
}

int ctx_dp_init(ctx_handle* h, const uint8_t id[CTX_DP_UNIQUE_ID_BYTES], int rank, int world) {
    if (!h) return CTX_E_INVALID;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(h, CTX_E_INVALID, "bad rank %d / world %d", rank, world);
    if (h->dp_comm) return fail(h, CTX_E_STATE, "ctx_dp_init was already called on this handle");
    if (!rccl_load()) return fail(h, CTX_E_DEVICE, "librccl could not be loaded: %s", rccl().err.c_str());
    HIP_TRY(h, hipSetDevice(h->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCL_TRY(h, rccl().CommInitRank(&h->dp_comm, world, u, rank));
    h->dp_rank = rank; h->dp_world = world;
    HIP_TRY(h, hipStreamCreateWithFlags(&h->dp_stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreateWithFlags(&h->dp_ev_ready, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->dp_ev_done, hipEventDisableTiming));
    if (!h->dp_scal) TRY(dev_alloc(h, &h->dp_scal, 4));
    // replicas start identical: rank 0's parameters and Adam slots (the step counter is host state: every rank must hold the
    // same one, which ctx_init_params / ctx_set_adam_state guarantee when called alike)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    RCCL_TRY(h, rccl().GroupStart());
    for (int slot : {0, 2, 3}) {
        float* p = h->arena + (int64_t)slot * h->Ppad;
        h->pack.version++;
        RCCL_TRY(h, rccl().Broadcast(p, p, (size_t)h->Ppad, ncclFloat, 0, h->dp_comm, h->dp_stream));
    }
    RCCL_TRY(h, rccl().GroupEnd());
    // ... and rank 0's step counter (host state behind the bias correction of Adam): two floats carry its 48 low bits exactly
    {
        float t2[2] = {(float)(h->adam_t & 0xffffff), (float)((h->adam_t >> 24) & 0xffffff)};
        HIP_TRY(h, hipMemcpyAsync(h->dp_scal, t2, sizeof t2, hipMemcpyHostToDevice, h->dp_stream));
        RCCL_TRY(h, rccl().Broadcast(h->dp_scal, h->dp_scal, 2, ncclFloat, 0, h->dp_comm, h->dp_stream));
        HIP_TRY(h, hipMemcpyAsync(t2, h->dp_scal, sizeof t2, hipMemcpyDeviceToHost, h->dp_stream));
        HIP_TRY(h, hipStreamSynchronize(h->dp_stream));
        h->adam_t = (int64_t)t2[0] + ((int64_t)t2[1] << 24);
    }
    HIP_TRY(h, hipStreamSynchronize(h->dp_stream));
    return CTX_OK;
}

int ctx_dp_world(const ctx_handle* h, int* rank, int* world) {
    if (!h) return CTX_E_INVALID;
    if (rank) *rank = h->dp_rank;
    if (world) *world = h->dp_comm ? h->dp_world : 0;
    return CTX_OK;
}

int ctx_dp_allreduce_grads(ctx_handle* h) {
    if (!h) return CTX_E_INVALID;
    if (!h->dp_comm) return fail(h, CTX_E_STATE, "ctx_dp_init first");
    if (!h->have_grads) return fail(h, CTX_E_STATE, "no backward has run");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(dp_reduce_range(h, 0, h->Ppad));
    TRY(dp_wait(h));
    HIP_TRY(h, hipGetLastError());
    return CTX_OK;
}

// the data-parallel step on the shard already in h->img = [tgt | src | ctx] (B triples): forward, backward with the simloss mean over the
// GLOBAL batch, the gradient buckets sent from inside backward, the rest after it, Adam
static int dp_step_on_img(ctx_handle* h, int B, float lr, float scalars[4]) {
    h->drop_on = true;
    forward(h, B, MODE_TRAIN);
    // two buckets: [split, Ppad) = translate/* + deconv/* leaves from inside backward (fire_bucket) and travels while the
    // encoders' backward runs; [0, split) = the encoders after it.  simloss is a mean over the GLOBAL batch (arm_shaping.py:1345).
    h->dp_in_step = true; h->dp_split = -1; h->dp_rc = CTX_OK; h->dp_done.clear();
    backward(h, B, B * h->dp_world);
    h->dp_in_step = false;
    h->drop_on = false;
    TRY(h->dp_rc);
    const int64_t split = h->dp_split >= 0 ? h->dp_split : h->Ppad;
    {   // what the buckets sent from inside backward left of the head [0, split)
        std::sort(h->dp_done.begin(), h->dp_done.end());
        int64_t at = 0;
        for (size_t i = 0; i <= h->dp_done.size(); ++i) {
            const int64_t stop = i < h->dp_done.size() ? h->dp_done[i].first : split;
            if (stop > at) TRY(dp_reduce_range(h, at, stop - at));
            if (i < h->dp_done.size()) at = h->dp_done[i].second;
        }
    }
    TRY(dp_wait(h));
    TRY(adam_step(h, lr));
    h->last_B = B;
    { char msg[256]; if (take_launch_error(msg, sizeof msg)) return fail(h, CTX_E_DEVICE, "%s", msg); }
    HIP_TRY(h, hipGetLastError());
    if (scalars) return ctx_dp_scalars(h, scalars);
    return CTX_OK;
}

int ctx_dp_train_step(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B, float lr, float scalars[4]) {
    TRY(check_B(h, B));
    if (!h->dp_comm) return fail(h, CTX_E_STATE, "ctx_dp_init first");
    if (!d_src || !d_ctx || !d_tgt) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(stage_frames(h, d_src, d_ctx, d_tgt, B));
    return dp_step_on_img(h, B, lr, scalars);
}

// This rank's shard of the trainer's batch, gathered on the device from the resident demo tensor: every rank is handed the SAME global
// index arrays (train_script.py:154-155) and takes rows [rank * B/world, (rank + 1) * B/world) with t = b % T on the GLOBAL row b.
static int dp_gather_shard(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B_global, int* B_local) {
    if (!h) return CTX_E_INVALID;
    if (!h->dp_comm) return fail(h, CTX_E_STATE, "ctx_dp_init first");
    if (!h->vdata) return fail(h, CTX_E_STATE, "ctx_demos_upload first");
    if (!choicesrc || !choicetgt) return fail(h, CTX_E_INVALID, "NULL index array");
    if (B_global <= 0 || B_global % h->dp_world) return fail(h, CTX_E_INVALID, "global batch %d is not a multiple of the %d ranks", B_global, h->dp_world);
    const int B = B_global / h->dp_world, b0 = h->dp_rank * B;
    TRY(check_B(h, B));
    for (int b = 0; b < B_global; ++b)      // (the whole array: every rank refuses the same bad call, so no rank is left waiting in a collective)
        if (choicesrc[b] < 0 || choicesrc[b] >= h->vN || choicetgt[b] < 0 || choicetgt[b] >= h->vN)
            return fail(h, CTX_E_INVALID, "video index out of range [0,%d)", h->vN);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->choice, choicesrc + b0, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->choice + h->Bm, choicetgt + b0, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    gather_triples(h->stream, h->vdata, h->vT, h->vN, h->npi, h->choice, h->choice + h->Bm, B, b0, h->lut, h->img);
    *B_local = B;
    return CTX_OK;
}

int ctx_dp_train_step_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B_global, float lr, float scalars[4]) {
    int B = 0;
    TRY(dp_gather_shard(h, choicesrc, choicetgt, B_global, &B));
    return dp_step_on_img(h, B, lr, scalars);
}

int ctx_dp_eval_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B_global, float scalars[4], float* out,
                        float* out2) {
    int B = 0;
    TRY(dp_gather_shard(h, choicesrc, choicetgt, B_global, &B));
    forward(h, B, MODE_TRAIN);
    losses(h->stream, h->out, h->img, nullptr, h->npi, B, h->Z, h->Z + (int64_t)B * h->Fp, nullptr, h->Fp, B, h->scratch, h->scalars, h->F, loss_terms_of(h));
    h->last_B = B;
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    if (out) TRY(copy_d2h(h, out, h->out, bytes));
    if (out2) TRY(copy_d2h(h, out2, h->out + B * h->npi, bytes));
    TRY(finish(h));
    if (scalars) return ctx_dp_scalars(h, scalars);
    return CTX_OK;
}

int ctx_dp_allreduce_host_f64(ctx_handle* h, double* buf, size_t n) {
    if (!h || (!buf && n)) return CTX_E_INVALID;
    if (!h->dp_comm) return fail(h, CTX_E_STATE, "ctx_dp_init first");
    if (n == 0) return CTX_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (n > h->dp_host_cap) {                               // staging buffer owned by the handle, grown when a larger call arrives
        if (h->dp_host_buf) { (void)hipFree(h->dp_host_buf); h->dp_host_buf = nullptr; h->dp_host_cap = 0; }
        if (hipMalloc((void**)&h->dp_host_buf, n * sizeof(double)) != hipSuccess) return fail(h, CTX_E_NOMEM, "hipMalloc(%zu bytes) for the host all-reduce", n * sizeof(double));
        h->dp_host_cap = n;
    }
    double* d = h->dp_host_buf;
    int rc = CTX_OK;
    do {
        if (hipMemcpyAsync(d, buf, n * sizeof(double), hipMemcpyHostToDevice, h->dp_stream) != hipSuccess) { rc = fail(h, CTX_E_DEVICE, "host all-reduce: upload failed"); break; }
        const ncclResult_t r = rccl().AllReduce(d, d, n, ncclDouble, ncclSum, h->dp_comm, h->dp_stream);
        if (r != ncclSuccess) { rc = fail(h, CTX_E_DEVICE, "ncclAllReduce(f64): %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "error"); break; }
        if (hipMemcpyAsync(buf, d, n * sizeof(double), hipMemcpyDeviceToHost, h->dp_stream) != hipSuccess) { rc = fail(h, CTX_E_DEVICE, "host all-reduce: download failed"); break; }
    } while (0);
    const hipError_t es = hipStreamSynchronize(h->dp_stream);
    if (rc == CTX_OK && es != hipSuccess) rc = fail(h, CTX_E_DEVICE, "host all-reduce: %s", hipGetErrorString(es));
    return rc;
}

int ctx_dp_scalars(ctx_handle* h, float scalars[4]) {
    if (!h || !scalars) return CTX_E_INVALID;
    if (!h->dp_comm) return fail(h, CTX_E_STATE, "ctx_dp_init first");
    HIP_TRY(h, hipSetDevice(h->device));
    // {loss, simloss, recon1, recon2} of this rank's shard -> global: recon sums add, simloss is the mean of equal shards
    HIP_TRY(h, hipEventRecord(h->dp_ev_ready, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->dp_stream, h->dp_ev_ready, 0));
    RCCL_TRY(h, rccl().AllReduce(h->scalars, h->dp_scal, 4, ncclFloat, ncclSum, h->dp_comm, h->dp_stream));
    float s[4];
    HIP_TRY(h, hipMemcpyAsync(s, h->dp_scal, sizeof s, hipMemcpyDeviceToHost, h->dp_stream));
    HIP_TRY(h, hipStreamSynchronize(h->dp_stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->dp_world > 1) {                         // (a one-rank sum is the identity: the device's own f64-accumulated loss stands)
        s[1] /= (float)h->dp_world;
        // `loss` = what Adam minimises: only the terms ctx_config.loss_terms keeps (ablations_code/ablations.py:175-182), like the
        // loss kernel's own masking of scalars[0] on one rank
        const int terms = loss_terms_of(h);
        s[0] = (float)((terms & CTX_LOSS_SIM ? (double)s[1] : 0.0) + (terms & CTX_LOSS_RECON1 ? (double)s[2] : 0.0) +
                       (terms & CTX_LOSS_RECON2 ? (double)s[3] : 0.0));
    }
    memcpy(scalars, s, sizeof s);
    return CTX_OK;
}

int ctx_demos_upload(ctx_handle* h, const uint8_t* vdata, int T, int N) {
    if (!h || !vdata || T <= 0 || N <= 0) return h ? fail(h, CTX_E_INVALID, "bad demo tensor") : CTX_E_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->vdata) { (void)hipFree(h->vdata); h->vdata = nullptr; }
    const size_t bytes = (size_t)T * N * h->npi;
    if (hipMalloc((void**)&h->vdata, bytes) != hipSuccess) return fail(h, CTX_E_NOMEM, "hipMalloc(%zu bytes) for the demo tensor", bytes);
    if (!h->lut) {
        TRY(dev_alloc(h, &h->lut, 256));
        TRY(dev_alloc(h, &h->choice, 2 * (int64_t)h->Bm));
        float host[256];
        for (int i = 0; i < 256; ++i) host[i] = (float)((double)i / 127.5 - 1.0);   // train_script.py:16-19, then the f32 feed
        HIP_TRY(h, hipMemcpy(h->lut, host, sizeof host, hipMemcpyHostToDevice));
    }
    HIP_TRY(h, hipMemcpyAsync(h->vdata, vdata, bytes, hipMemcpyHostToDevice, h->stream));
    h->vT = T; h->vN = N;
    return finish(h);
}

int ctx_train_step_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B, float lr, float scalars[4]) {
    TRY(check_B(h, B));
    if (!h->vdata) return fail(h, CTX_E_STATE, "ctx_demos_upload first");
    if (!choicesrc || !choicetgt) return fail(h, CTX_E_INVALID, "NULL index array");
    for (int b = 0; b < B; ++b)
        if (choicesrc[b] < 0 || choicesrc[b] >= h->vN || choicetgt[b] < 0 || choicetgt[b] >= h->vN)
            return fail(h, CTX_E_INVALID, "video index out of range [0,%d)", h->vN);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->choice, choicesrc, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->choice + h->Bm, choicetgt, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    gather_triples(h->stream, h->vdata, h->vT, h->vN, h->npi, h->choice, h->choice + h->Bm, B, 0, h->lut, h->img);
    TRY(fused_step(h, B, lr));
    h->last_B = B;
    if (scalars) HIP_TRY(h, hipMemcpyAsync(scalars, h->scalars, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return finish(h);
}

int ctx_eval_sampled(ctx_handle* h, const int32_t* choicesrc, const int32_t* choicetgt, int B, float scalars[4], float* out, float* out2) {
    TRY(check_B(h, B));
    if (!h->vdata) return fail(h, CTX_E_STATE, "ctx_demos_upload first");
    if (!choicesrc || !choicetgt) return fail(h, CTX_E_INVALID, "NULL index array");
    for (int b = 0; b < B; ++b)
        if (choicesrc[b] < 0 || choicesrc[b] >= h->vN || choicetgt[b] < 0 || choicetgt[b] >= h->vN)
            return fail(h, CTX_E_INVALID, "video index out of range [0,%d)", h->vN);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->choice, choicesrc, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->choice + h->Bm, choicetgt, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    gather_triples(h->stream, h->vdata, h->vT, h->vN, h->npi, h->choice, h->choice + h->Bm, B, 0, h->lut, h->img);
    forward(h, B, MODE_TRAIN);
    losses(h->stream, h->out, h->img, nullptr, h->npi, B, h->Z, h->Z + (int64_t)B * h->Fp, nullptr, h->Fp, B, h->scratch, h->scalars, h->F, loss_terms_of(h));
    h->last_B = B;
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    if (scalars) HIP_TRY(h, hipMemcpyAsync(scalars, h->scalars, 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (out) TRY(copy_d2h(h, out, h->out, bytes));
    if (out2) TRY(copy_d2h(h, out2, h->out + B * h->npi, bytes));
    return finish(h);
}

int ctx_last_outputs(ctx_handle* h, float* out, float* out2, float* tgt) {
    if (!h) return CTX_E_INVALID;
    if (h->last_B <= 0) return fail(h, CTX_E_STATE, "no training-mode forward has run");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)h->last_B * h->npi * sizeof(float);
    if (out) TRY(copy_d2h(h, out, h->out, bytes));
    if (out2) HIP_TRY(h, hipMemcpyAsync(out2, h->out + (int64_t)h->last_B * h->npi, bytes, hipMemcpyDeviceToHost, h->stream));
    if (tgt) HIP_TRY(h, hipMemcpyAsync(tgt, h->img, bytes, hipMemcpyDeviceToHost, h->stream));   // img = [tgt | src | ctx]
    return finish(h);
}

// Test hook: copy an internal device buffer to the host (names: img Z dZ cz th0 dz out dout dDz dsim2
// dth0 dcz, s0..s4 c0..c4 dS0..dS4 dC0..dC4 dSk0..dSk3, e1..e3 dE1..dE3; ContextAEReal / ContextAEInception2: img Z dZ out dz th0
// a0..a4 e1..e3).  n = floats to copy.
int ctx_debug_read(ctx_handle* h, const char* name, float* host, size_t n) {
    if (!h || !name || !host) return CTX_E_INVALID;
    const std::string s(name);
    const float* p = nullptr;
    auto idx = [&](const char* pre, int lo, int hi) -> int {
        const size_t L = strlen(pre);
        if (s.size() == L + 1 && s.compare(0, L, pre) == 0 && s[L] >= '0' + lo && s[L] <= '0' + hi) return s[L] - '0';
        return -1;
    };
    int k;
    if (h->gen) {   // table-driven models: a0..a3 conv outputs over the stacked [tgt | src | ctx] images, a4 = h4 [3B, Fp], e1..e3, dz, th0, Z
        const GenState& r = *h->gen;
        if (s == "img") p = h->img; else if (s == "Z") p = h->Z; else if (s == "dZ") p = h->dZ; else if (s == "out") p = h->out;
        else if (s == "dz") p = r.dz; else if (s == "th0") p = r.th0;
        else if ((k = idx("a", 0, 4)) >= 0) p = r.a[k];
        else if ((k = idx("e", 1, 3)) >= 0) p = r.e[k];
        if (!p) return fail(h, CTX_E_INVALID, "unknown debug buffer '%s' (table-driven models: img Z dZ out dz th0 a0..a4 e1..e3)", name);
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(host, p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        return finish(h);
    }
    if (s == "img") p = h->img; else if (s == "Z") p = h->Z; else if (s == "dZ") p = h->dZ;
    else if (s == "cz") p = h->cz; else if (s == "th0") p = h->th0; else if (s == "dz") p = h->dz;
    else if (s == "out") p = h->out; else if (s == "dout") p = h->dout; else if (s == "dDz") p = h->dDz;
    else if (s == "dsim2") p = h->dsim2; else if (s == "dth0") p = h->dth0; else if (s == "dcz") p = h->dcz;
    else if ((k = idx("dSk", 0, 3)) >= 0) p = h->dSk[k];
    else if ((k = idx("dS", 0, 4)) >= 0) p = h->dS[k];
    else if ((k = idx("dC", 0, 4)) >= 0) p = h->dC[k];
    else if ((k = idx("dE", 1, 3)) >= 0) p = h->dE[k];
    else if ((k = idx("s", 0, 4)) >= 0) p = h->s[k];
    else if ((k = idx("c", 0, 4)) >= 0) p = h->c[k];
    else if ((k = idx("e", 1, 3)) >= 0) p = h->e[k];
    if (!p) return fail(h, CTX_E_INVALID, "unknown debug buffer '%s'", name);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(host, p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return finish(h);
}

}  // extern "C"
