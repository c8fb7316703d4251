// ctx_engine.cpp -- the launch sequences of libctxtrans.so: buffers, forward / backward / Adam of ContextSkipNew
// (gym/envs/mujoco/arm_shaping.py:1272-1354), the table-driven engine of ContextAEReal / ContextAEInception2 (ctxtrans_gen.inc), the
// captured inference forwards.  No CPU code path: every contraction runs in the HIP kernels of igemm.h / kernels.hip / the direct kernels.
#include "ctx_internal.h"

namespace ctxi {
thread_local std::string g_create_error;
}  // namespace ctxi

namespace ctxi {

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


}  // namespace ctxi
#include "ctxtrans_gen.inc"
namespace ctxi {

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

}  // namespace ctxi
