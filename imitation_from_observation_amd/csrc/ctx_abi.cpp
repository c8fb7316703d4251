// ctx_abi.cpp -- the C ABI of include/ctxtrans.h (one export per TF call site of the reference: SURVEY.md 8b) except ctx_dp_* (ctx_dp.cpp).
#include "ctx_internal.h"

using namespace ctxi;

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
    h->ctx_single = !ctx_batched;                 // one context frame: encoded once, read by every row (forward)
    u8_to_f32(h->stream, u_src, h->img + B * npi, (B + (ctx_batched ? B : 1)) * npi);      // [src | ctx] are adjacent in both buffers: one launch
    return translate_tail(h, B, pred, feat);
}

// image_trans of uint8 frames ON THE HOST: the 256 values of prep_u8 (kernels.hip: x * (1/255), - 0.5, * 2.0 as three separately rounded f32
// operations, base.py:116-119) from a table -- bit-identical to what u8_to_f32_kernel writes.  ctx_encode hands `frames_f32` back this way
// at the reward hook's call sizes: the conversion runs while the device encodes, and the 1.2 MB download of the batch of 25 (a pageable
// copy, ~40 us of a 0.27 ms call) is not made.
static void host_prep_u8(const uint8_t* in, float* out, size_t n) {
    static const struct Lut {
        float v[256];
        Lut() {
#pragma clang fp contract(off)
            for (int x = 0; x < 256; ++x) {
                volatile float scaled = (float)x * (1.0f / 255.0f);
                volatile float centred = scaled - 0.5f;
                v[x] = centred * 2.0f;
            }
        }
    } lut;
    for (size_t i = 0; i < n; ++i) out[i] = lut.v[in[i]];
}
constexpr size_t HOST_PREP_MAX = 1u << 20;        // elements (64 frames of 64x64x3): beyond, the device's copy is the faster source

int ctx_encode(ctx_handle* h, const uint8_t* frames, int B, float* feat, float* frames_f32) {
    TRY(check_B(h, B));
    TRY(need_frames(h));
    if (!frames) return fail(h, CTX_E_INVALID, "NULL input");
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t npi = h->npi;
    HIP_TRY(h, hipMemcpyAsync(h->u8, frames, (size_t)B * npi, hipMemcpyHostToDevice, h->stream));
    u8_to_f32(h->stream, h->u8, h->img + B * npi, B * npi);
    TRY(forward_inference(h, B, MODE_ENCODE));
    const bool on_host = frames_f32 && (size_t)B * npi <= HOST_PREP_MAX;
    if (on_host) host_prep_u8(frames, frames_f32, (size_t)B * npi);                      // (the device is busy with the launches above)
    if (feat) HIP_TRY(h, hipMemcpy2DAsync(feat, h->F * sizeof(float), h->Z + 2ll * B * h->Fp, h->Fp * sizeof(float), h->F * sizeof(float), B,
                                         hipMemcpyDeviceToHost, h->stream));
    if (frames_f32 && !on_host) TRY(copy_d2h(h, frames_f32, h->img + B * npi, (size_t)B * npi * sizeof(float)));
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
}  // extern "C"
namespace ctxi {
int stage_frames(ctx_handle* h, const float* d_src, const float* d_ctx, const float* d_tgt, int B) {
    const size_t bytes = (size_t)B * h->npi * sizeof(float);
    const float* from[3] = {d_tgt, d_src, d_ctx};
    for (int k = 0; k < 3; ++k) {
        float* slot = h->img + (int64_t)k * B * h->npi;
        if (from[k] != slot) HIP_TRY(h, hipMemcpyAsync(slot, from[k], bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    return CTX_OK;
}
}  // namespace ctxi
extern "C" {

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

