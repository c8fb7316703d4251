// cnn.cpp -- forward executor for a frozen conv net given as an op list: the Inception-v3 front end of mode
// 'oursinception' (nets/inception_v3.py:93-416 under nets/inception_utils.py:31-71, is_training=False), C ABI
// ctx_cnn_* in include/ctxtrans.h.  The GRAPH lives on the host side (imitation_from_observation_amd/
// inception_frontend.py mirrors the reference's Python graph builder); this file only executes it:
//   conv     slim.conv2d = conv + batch norm (moving statistics) + ReLU.  The host folds the batch norm into the
//            filter and a bias; the conv is the same implicit GEMM as the translator's (KmConvGather with a KH x KW
//            kernel, stride, SAME/VALID padding x NmPlain filter rows), epilogue bias + ReLU, written into a CHANNEL
//            SLICE of the destination tensor -- tf.concat never copies.
//   maxpool  3x3 stride 2 VALID;  avgpool  3x3 stride 1 SAME (mean over the taps inside the image).
// Activations are NHWC with the channel count rounded up to 32 (the 3-channel frames, the 80- and 48-wide tensors);
// padded channels are zero because nothing ever writes them and the matching filter rows are zero.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <vector>

#include "../../include/ctxtrans.h"
#include "launch.h"

using namespace ctx;

struct ctx_cnn {
    Options opt{};                                    // this handle's switches (options.h): the environment's at ctx_cnn_create
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int max_images = 0, precision = 0;
    std::vector<ctx_cnn_buf> bufs;
    std::vector<ctx_cnn_op> ops;
    std::vector<float*> dbuf;
    float* weights = nullptr;
    int64_t weight_floats = 0;
    uint8_t* u8 = nullptr;
    float* f32in = nullptr;
    static constexpr int NLANE = 4;                   // branch lanes: lane 0 is `stream`
    hipStream_t lane[NLANE] = {};
    float* slab[NLANE] = {};                          // one split-K workspace per lane
    float* wpack[NLANE] = {};                         // dconv's re-packed filter, per lane
    int64_t slab_floats = 0;
    std::vector<hipEvent_t> done;                     // done[i]: op i finished (recorded on its lane)
    std::vector<std::vector<int>> deps;               // deps[i]: ops on OTHER lanes that write op i's src buffer
    hipEvent_t ev_fork = nullptr;
    bool overlap = true;
    float* zeros = nullptr;
    bool stem4 = false;                               // buffer 0 is kept as [pixels][4] (see stem4_ok)
    struct GraphSlot { int calls = 0; hipGraphExec_t exec = nullptr; };
    std::map<int, GraphSlot> graphs;                  // one captured pass per image count (run_cached)
    bool use_graphs = true;
    std::string err;
};

namespace {
thread_local std::string g_cnn_create_error;

int cfail(ctx_cnn* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_cnn_create_error = buf;
    return code;
}
#define CNN_HIP(h, expr)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return cfail(h, CTX_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

void out_dims(const ctx_cnn_buf& in, const ctx_cnn_op& op, int& ho, int& wo) {
    if (op.same) { ho = (in.h + op.stride - 1) / op.stride; wo = (in.w + op.stride - 1) / op.stride; }
    else { ho = (in.h - op.kh) / op.stride + 1; wo = (in.w - op.kw) / op.stride + 1; }
}
int same_before(int n, int k, int s) {
    const int out = (n + s - 1) / s, total = (out - 1) * s + k - n;
    return total > 0 ? total / 2 : 0;
}

// Buffer 0 holds 3 real channels.  When every reader is a square 3x3 / 5x5 conv it is stored 4 channels wide and those
// convs run on the cin = 3 gather (K = 4 x taps instead of 32 x taps: Conv2d_1a_3x3 0.245 -> 0.05 ms at 192 x 125 x 125).
bool stem4_ok(const std::vector<ctx_cnn_buf>& bufs, const std::vector<ctx_cnn_op>& ops) {
    const bool on = opt(OPT_CNN_STEM4) != 0;
    bool any = false;
    for (const ctx_cnn_op& op : ops) {
        if (op.dst == 0) return false;
        if (op.src != 0) continue;
        if (op.kind != CTX_CNN_CONV || op.kh != op.kw || (op.kh != 3 && op.kh != 5)) return false;
        if (op.same && same_before(bufs[0].h, op.kh, op.stride) != same_before(bufs[0].w, op.kw, op.stride)) return false;
        any = true;
    }
    return on && any;
}

int validate(const std::vector<ctx_cnn_buf>& bufs, const std::vector<ctx_cnn_op>& ops, int64_t weight_floats, int max_images) {
    if (bufs.empty() || ops.empty()) return cfail(nullptr, CTX_E_INVALID, "empty graph");
    for (size_t i = 0; i < bufs.size(); ++i) {
        const ctx_cnn_buf& b = bufs[i];
        if (b.h <= 0 || b.w <= 0 || b.c <= 0 || b.c % 32) return cfail(nullptr, CTX_E_INVALID, "buffer %zu: %dx%dx%d (channels must be a positive multiple of 32)", i, b.h, b.w, b.c);
        if ((int64_t)max_images * b.h * b.w * b.c * 4 >= (1ll << 31))
            return cfail(nullptr, CTX_E_INVALID, "buffer %zu needs >= 2 GiB at max_images %d: lower max_images (forward() chunks the batch)", i, max_images);
    }
    for (size_t i = 0; i < ops.size(); ++i) {
        const ctx_cnn_op& op = ops[i];
        if (op.src < 0 || op.dst < 0 || op.src >= (int)bufs.size() || op.dst >= (int)bufs.size() || op.src == op.dst)
            return cfail(nullptr, CTX_E_INVALID, "op %zu: bad buffer ids", i);
        if (op.lane < 0 || op.lane >= ctx_cnn::NLANE) return cfail(nullptr, CTX_E_INVALID, "op %zu: lane %d outside [0,%d)", i, op.lane, ctx_cnn::NLANE);
        const ctx_cnn_buf &in = bufs[op.src], &out = bufs[op.dst];
        int ho, wo;
        ctx_cnn_op o = op;
        if (op.kind != CTX_CNN_CONV) { o.kh = o.kw = 3; o.stride = op.kind == CTX_CNN_MAXPOOL ? 2 : 1; o.same = op.kind == CTX_CNN_AVGPOOL; }
        out_dims(in, o, ho, wo);
        if (ho != out.h || wo != out.w) return cfail(nullptr, CTX_E_INVALID, "op %zu: output grid %dx%d but buffer %d is %dx%d", i, ho, wo, op.dst, out.h, out.w);
        const int cw = op.kind == CTX_CNN_CONV ? (op.nsplit ? op.nsplit : op.cout) : in.c;
        if (op.dst_ch0 < 0 || op.dst_ch0 % 4 || op.dst_ch0 + cw > out.c) return cfail(nullptr, CTX_E_INVALID, "op %zu: channel slice [%d,%d) outside buffer %d", i, op.dst_ch0, op.dst_ch0 + cw, op.dst);
        if (op.kind != CTX_CNN_CONV && (op.src_c || op.nsplit)) return cfail(nullptr, CTX_E_INVALID, "op %zu: channel slices / merged outputs are conv-only", i);
        if (op.kind == CTX_CNN_CONV) {
            if (op.kh < 1 || op.kw < 1 || op.kh * op.kw > 25 || op.cout <= 0 || op.cout % 4 || (op.stride != 1 && op.stride != 2))
                return cfail(nullptr, CTX_E_INVALID, "op %zu: unsupported conv %dx%d stride %d cout %d", i, op.kh, op.kw, op.stride, op.cout);
            if (op.src_c && (op.src_c % 32 || op.src_ch0 % 32 || op.src_ch0 < 0 || op.src_ch0 + op.src_c > in.c || op.src == 0))
                return cfail(nullptr, CTX_E_INVALID, "op %zu: input slice [%d,%d) of buffer %d", i, op.src_ch0, op.src_ch0 + op.src_c, op.src);
            if (op.nsplit) {
                if (op.nsplit < 0 || op.nsplit % 4 || op.nsplit >= op.cout || op.dst2 < 0 || op.dst2 >= (int)bufs.size() || op.dst2 == op.src || op.dst2 == op.dst)
                    return cfail(nullptr, CTX_E_INVALID, "op %zu: bad merged-output description", i);
                const ctx_cnn_buf& o2 = bufs[op.dst2];
                if (o2.h != out.h || o2.w != out.w || op.dst2_ch0 < 0 || op.dst2_ch0 % 4 || op.dst2_ch0 + (op.cout - op.nsplit) > o2.c)
                    return cfail(nullptr, CTX_E_INVALID, "op %zu: second output slice outside buffer %d", i, op.dst2);
            }
            const int64_t nw = (int64_t)op.kh * op.kw * (op.src_c ? op.src_c : in.c) * op.cout;
            if (op.w_off < 0 || op.w_off % 4 || op.w_off + nw > weight_floats || op.b_off < 0 || op.b_off + op.cout > weight_floats)
                return cfail(nullptr, CTX_E_INVALID, "op %zu: weights outside the blob", i);
        } else if (op.kind != CTX_CNN_MAXPOOL && op.kind != CTX_CNN_AVGPOOL) return cfail(nullptr, CTX_E_INVALID, "op %zu: unknown kind %d", i, op.kind);
    }
    return CTX_OK;
}

// One pass over the op list.  Ops of different lanes overlap (the branches of an Inception block are independent small
// GEMMs that do not fill the chip alone); each op is ordered after the ops that wrote its src buffer, and the pass ends
// with every lane joined into lane 0.  With `ev` (profiling) everything runs on lane 0, one event per op boundary.
int run(ctx_cnn* h, int n, std::vector<hipEvent_t>* ev = nullptr) {
    OptScope os(&h->opt);
    const bool par = h->overlap && !ev;
    if (par) {
        (void)hipEventRecord(h->ev_fork, h->lane[0]);
        for (int l = 1; l < ctx_cnn::NLANE; ++l) (void)hipStreamWaitEvent(h->lane[l], h->ev_fork, 0);
    }
    int last_on[ctx_cnn::NLANE] = {-1, -1, -1, -1};
    for (size_t oi = 0; oi < h->ops.size(); ++oi) {
        const ctx_cnn_op& op = h->ops[oi];
        const int L = par ? op.lane : 0;
        hipStream_t st = h->lane[L];
        if (ev) (void)hipEventRecord((*ev)[oi], st);
        if (par) for (int j : h->deps[oi]) (void)hipStreamWaitEvent(st, h->done[j], 0);
        const SplitWs ws{h->slab[L], h->slab_floats, h->precision};
        const ctx_cnn_buf &in = h->bufs[op.src], &out = h->bufs[op.dst];
        const float* x = h->dbuf[op.src] + (op.kind == CTX_CNN_CONV ? op.src_ch0 : 0);
        const int cin = op.kind == CTX_CNN_CONV && op.src_c ? op.src_c : in.c;      // channels the conv reads (row stride stays in.c)
        float* y = h->dbuf[op.dst] + op.dst_ch0;
        if (op.kind == CTX_CNN_MAXPOOL) maxpool3x3s2(st, x, y, n, in.h, in.w, in.c, out.c);
        else if (op.kind == CTX_CNN_AVGPOOL) avgpool3x3s1(st, x, y, n, in.h, in.w, in.c, out.c);
        else {
            const int R = n * out.h * out.w;
            const float* w = h->weights + op.w_off;
            Epi ep;
            ep.out1 = y; ep.ld1 = out.c; ep.bias = h->weights + op.b_off; ep.lrelu = 2;
            if (op.nsplit) { ep.nsplit = op.nsplit; ep.out2 = h->dbuf[op.dst2] + op.dst2_ch0; ep.ld2 = h->bufs[op.dst2].c; }
            const int pady = op.same ? same_before(in.h, op.kh, op.stride) : 0, padx = op.same ? same_before(in.w, op.kw, op.stride) : 0;
            if (op.src == 0 && h->stem4) {
                KmC3Gather a{x, in.h, in.w, out.h, out.w, R, h->zeros};
                a.s = op.stride; a.pad = pady; a.K = op.kh;
                NmC3Weights b{w, op.cout, h->zeros};
                b.ntap = op.kh * op.kw; b.cs = in.c;
                conv3_fwd(st, a, b, ep, R, op.cout, ws);
            } else if (h->precision == CTX_PREC_F32 && opt(OPT_CNN_DCONV) && op.kh == op.kw && cin >= 8 && (cin & (cin - 1)) == 0 && dconv_ok(cin, op.cout) &&
                       cin <= 32 && op.cout <= 32) {
                // Conv2d_2a_3x3 (32 -> 32): a 32-column GEMM wastes the implicit GEMM's tiles (57 TF/s); the direct convolution over an LDS
                // halo tile with the filter resident (dconv.h) runs it at 81 (0.223 -> 0.157 ms at 192 images of 125x125).  Wider layers lose:
                // 32 -> 64 0.271 -> 0.399 ms, 64 -> 80 1x1 0.076 -> 0.125, 64 -> 96 0.048 -> 0.092 (option value 2 runs them all)
                DcFwd D{};
                D.x1 = x; D.ld1 = in.c; D.c1 = cin; D.CI = cin; D.hin = in.h; D.win = in.w; D.nimg = n;
                D.w = w; D.wmode = 0; D.N = op.cout; D.ep = ep; D.wp = h->wpack[L];
                dconv_conv_k(st, D, op.kh, op.kw, op.stride, pady, padx, out.h, out.w);
            } else if (n >= 64 && op.same && op.kh * op.kw > 1) {   // position-major: SAME-padding taps outside the grid are never multiplied
                PosGeo g = make_posgeo(out.h, out.w, in.h, in.w, op.stride, pady, op.kh, cin / KC);
                g.KW = op.kw; g.padx = padx;
                conv_fwd_q(st, KmConvGatherQ{x, in.c, g, n, h->zeros}, NmConvWeightsQ{w, cin, op.cout, op.kw, h->zeros}, ep, op.cout, ws);
            } else {
                KmConvGather a{x, in.c, in.h, in.w, out.h, out.w, cin / KC, R, h->zeros};
                a.s = op.stride; a.K = op.kh; a.KW = op.kw; a.pad = pady; a.padx = padx;
                NmPlain b{w, op.cout, nullptr, 0, op.cout, op.cout, op.kh * op.kw * cin, h->zeros};
                conv_fwd(st, a, b, ep, R, op.cout, ws);
            }
        }
        if (par) { (void)hipEventRecord(h->done[oi], st); last_on[L] = (int)oi; }
    }
    if (par) for (int l = 1; l < ctx_cnn::NLANE; ++l) if (last_on[l] >= 0) (void)hipStreamWaitEvent(h->lane[0], h->done[last_on[l]], 0);
    if (ev) (void)hipEventRecord((*ev)[h->ops.size()], h->lane[0]);
    if (hipGetLastError() != hipSuccess) return cfail(h, CTX_E_DEVICE, "kernel launch failed");
    return CTX_OK;
}
// The ~107 launches of a pass are 20-120 us each: captured into a hipGraph on the second pass with a given image count and
// replayed afterwards (a kernel trace of the config-4 step showed the device idle ~9 % of the time between them).  Single-lane
// passes only (the branch lanes of the split-bf16 mode keep plain launches); every pointer a launch captures is owned by the
// handle.  Option "graphs" = 0 (CTX_GRAPHS=0 in the environment at create) keeps plain launches.
int run_cached(ctx_cnn* h, int n) {
    if (!h->use_graphs || h->overlap) return run(h, n);
    ctx_cnn::GraphSlot& g = h->graphs[n];
    if (g.calls++ == 0) return run(h, n);                      // first pass: plain (code objects, LDS limits)
    if (!g.exec) {
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(h->lane[0], hipStreamCaptureModeThreadLocal);
        int rc = CTX_OK;
        if (e == hipSuccess) {
            rc = run(h, n);
            e = hipStreamEndCapture(h->lane[0], &graph);
        }
        if (e == hipSuccess && rc == CTX_OK && graph) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess || rc != CTX_OK || !g.exec) {       // capture not possible here: stay on plain launches
            (void)hipGetLastError();
            g.exec = nullptr;
            h->use_graphs = false;
            return run(h, n);
        }
    }
    if (hipGraphLaunch(g.exec, h->lane[0]) != hipSuccess) return cfail(h, CTX_E_DEVICE, "hipGraphLaunch failed");
    return CTX_OK;
}
}  // namespace

extern "C" {

int ctx_cnn_create(const ctx_cnn_buf* bufs, int nbufs, const ctx_cnn_op* ops, int nops, int64_t weight_floats, int max_images,
                   int precision, int device, void* stream, ctx_cnn** out) {
    if (!out) return cfail(nullptr, CTX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!bufs || !ops || nbufs <= 0 || nops <= 0 || weight_floats <= 0 || max_images <= 0) return cfail(nullptr, CTX_E_INVALID, "bad arguments");
    if (precision != CTX_PREC_F32 && precision != CTX_PREC_BF16X3) return cfail(nullptr, CTX_E_INVALID, "unsupported precision %d", precision);
    std::vector<ctx_cnn_buf> vb(bufs, bufs + nbufs);
    std::vector<ctx_cnn_op> vo(ops, ops + nops);
    if (validate(vb, vo, weight_floats, max_images) != CTX_OK) return CTX_E_INVALID;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return cfail(nullptr, CTX_E_DEVICE, "no HIP device available; libctxtrans has no CPU path");
    if (device < 0 || device >= ndev) return cfail(nullptr, CTX_E_INVALID, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return cfail(nullptr, CTX_E_DEVICE, "hipSetDevice failed");
    ctx_cnn* h = new ctx_cnn();
    h->opt = options_from_env();
    OptScope os(&h->opt);
    h->device = device; h->max_images = max_images; h->precision = precision; h->bufs = vb; h->ops = vo; h->weight_floats = weight_floats;
    bool ok = true;
    if (stream) h->stream = (hipStream_t)stream;
    else { ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess; h->own_stream = ok; }
    auto alloc = [&](void** p, size_t bytes, bool zero) {
        if (!ok) return;
        ok = hipMalloc(p, bytes) == hipSuccess && (!zero || hipMemset(*p, 0, bytes) == hipSuccess);
    };
    h->dbuf.assign(nbufs, nullptr);
    h->stem4 = stem4_ok(vb, vo);
    for (int i = 0; i < nbufs; ++i) alloc((void**)&h->dbuf[i], (size_t)max_images * vb[i].h * vb[i].w * (i == 0 && h->stem4 ? 4 : vb[i].c) * sizeof(float), true);
    alloc((void**)&h->weights, (size_t)weight_floats * sizeof(float), true);
    const size_t npix = (size_t)max_images * vb[0].h * vb[0].w;
    alloc((void**)&h->u8, npix * 3, false);
    alloc((void**)&h->f32in, npix * 3 * sizeof(float), false);
    h->slab_floats = 32ll << 20;
    h->lane[0] = h->stream;
    for (int l = 0; l < ctx_cnn::NLANE; ++l) {
        alloc((void**)&h->slab[l], (size_t)h->slab_floats * sizeof(float), false);
        alloc((void**)&h->wpack[l], (size_t)DC_WPACK_FLOATS * sizeof(float), false);
        if (l && ok) ok = hipStreamCreateWithFlags(&h->lane[l], hipStreamNonBlocking) == hipSuccess;
    }
    h->done.assign(nops, nullptr);
    h->deps.assign(nops, {});
    for (int i = 0; i < nops && ok; ++i) {
        ok = hipEventCreateWithFlags(&h->done[i], hipEventDisableTiming) == hipSuccess;
        for (int j = 0; j < i; ++j)
            if ((vo[j].dst == vo[i].src || (vo[j].nsplit && vo[j].dst2 == vo[i].src)) && vo[j].lane != vo[i].lane) h->deps[i].push_back(j);
    }
    if (ok) ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    // measured at 192 images of 125x125: branch lanes -11 % in the split-bf16 mode, +2 % (and a slower chained train step) in f32
    h->overlap = h->opt.v[OPT_CNN_LANES] < 0 ? precision == CTX_PREC_BF16X3 : h->opt.v[OPT_CNN_LANES] != 0;
    h->use_graphs = h->opt.v[OPT_GRAPHS] != 0;
    alloc((void**)&h->zeros, 256, true);
    if (!ok) { cfail(nullptr, CTX_E_NOMEM, "device allocation failed"); ctx_cnn_destroy(h); return CTX_E_NOMEM; }
    *out = h;
    return CTX_OK;
}

void ctx_cnn_destroy(ctx_cnn* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (float* p : h->dbuf) if (p) (void)hipFree(p);
    for (void* p : {(void*)h->weights, (void*)h->u8, (void*)h->f32in, (void*)h->zeros}) if (p) (void)hipFree(p);
    for (int l = 0; l < ctx_cnn::NLANE; ++l) {
        if (h->slab[l]) (void)hipFree(h->slab[l]);
        if (h->wpack[l]) (void)hipFree(h->wpack[l]);
        if (l && h->lane[l]) { (void)hipStreamSynchronize(h->lane[l]); (void)hipStreamDestroy(h->lane[l]); }
    }
    for (auto& kv : h->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    for (hipEvent_t e : h->done) if (e) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* ctx_cnn_last_error(const ctx_cnn* h) { return h ? h->err.c_str() : g_cnn_create_error.c_str(); }

int ctx_cnn_set_weights(ctx_cnn* h, const float* blob, size_t n) {
    if (!h || !blob) return CTX_E_INVALID;
    if ((int64_t)n != h->weight_floats) return cfail(h, CTX_E_INVALID, "expected %lld floats, got %zu", (long long)h->weight_floats, n);
    CNN_HIP(h, hipSetDevice(h->device));
    CNN_HIP(h, hipMemcpyAsync(h->weights, blob, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    CNN_HIP(h, hipStreamSynchronize(h->stream));
    return CTX_OK;
}

// frames: host uint8 [n, H, W, 3] (preprocessed like base.py:116-119) -> out: host f32 [n, h, w, c] of the LAST buffer
int ctx_cnn_forward_u8(ctx_cnn* h, const uint8_t* frames, int n, float* out) {
    if (!h || !frames || n <= 0) return h ? cfail(h, CTX_E_INVALID, "bad arguments") : CTX_E_INVALID;
    CNN_HIP(h, hipSetDevice(h->device));
    const ctx_cnn_buf &b0 = h->bufs.front(), &bl = h->bufs.back();
    const int64_t pix_in = (int64_t)b0.h * b0.w, per_out = (int64_t)bl.h * bl.w * bl.c;
    for (int i0 = 0; i0 < n; i0 += h->max_images) {
        const int m = n - i0 < h->max_images ? n - i0 : h->max_images;
        CNN_HIP(h, hipMemcpyAsync(h->u8, frames + (int64_t)i0 * pix_in * 3, (size_t)m * pix_in * 3, hipMemcpyHostToDevice, h->stream));
        pad_channels_u8(h->stream, h->u8, h->dbuf[0], m * pix_in, h->stem4 ? 4 : b0.c);
        const int rc = run_cached(h, m);
        if (rc != CTX_OK) return rc;
        if (out) CNN_HIP(h, hipMemcpyAsync(out + (int64_t)i0 * per_out, h->dbuf.back(), (size_t)m * per_out * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        CNN_HIP(h, hipStreamSynchronize(h->stream));
    }
    return CTX_OK;
}

// frames: host uint8 [n, H, W, 3], n <= max_images -> *d_out: DEVICE pointer of the last buffer [n, h, w, c].  Asynchronous on the
// handle's stream (the upload is stream-ordered too; `frames` must stay valid until the stream has passed it -- callers that reuse
// the buffer sync first).  The product path of mode 'oursinception': the feature maps never visit the host.
int ctx_cnn_forward_u8_dev(ctx_cnn* h, const uint8_t* frames, int n, const float** d_out) {
    if (!h || !frames || n <= 0 || n > h->max_images) return h ? cfail(h, CTX_E_INVALID, "n must be in [1, max_images]") : CTX_E_INVALID;
    CNN_HIP(h, hipSetDevice(h->device));
    const ctx_cnn_buf& b0 = h->bufs.front();
    const int64_t pix_in = (int64_t)b0.h * b0.w;
    CNN_HIP(h, hipMemcpyAsync(h->u8, frames, (size_t)n * pix_in * 3, hipMemcpyHostToDevice, h->stream));
    pad_channels_u8(h->stream, h->u8, h->dbuf[0], n * pix_in, h->stem4 ? 4 : b0.c);
    const int rc = run_cached(h, n);
    if (rc != CTX_OK) return rc;
    if (d_out) *d_out = h->dbuf.back();
    return CTX_OK;
}

// d_frames: DEVICE f32 [n, H, W, 3] in [-1,1], n <= max_images; *d_out: device pointer of the last buffer [n, h, w, c].
// Asynchronous on the handle's stream.
int ctx_cnn_forward_dev(ctx_cnn* h, const float* d_frames, int n, const float** d_out) {
    if (!h || !d_frames || n <= 0 || n > h->max_images) return h ? cfail(h, CTX_E_INVALID, "n must be in [1, max_images]") : CTX_E_INVALID;
    CNN_HIP(h, hipSetDevice(h->device));
    const ctx_cnn_buf& b0 = h->bufs.front();
    pad_channels_f32(h->stream, d_frames, h->dbuf[0], (int64_t)n * b0.h * b0.w, h->stem4 ? 4 : b0.c);
    const int rc = run_cached(h, n);
    if (rc != CTX_OK) return rc;
    if (d_out) *d_out = h->dbuf.back();
    return CTX_OK;
}

// copies buffer `index` (n images) to the host: bring-up and the end-point tests
int ctx_cnn_read_buffer(ctx_cnn* h, int index, int n, float* out) {
    if (!h || !out || index < 0 || index >= (int)h->bufs.size() || n <= 0 || n > h->max_images) return CTX_E_INVALID;
    const ctx_cnn_buf& b = h->bufs[index];
    CNN_HIP(h, hipSetDevice(h->device));
    if (index == 0 && h->stem4) {                     // stored 4 wide; the caller sees the declared (padded) width
        CNN_HIP(h, hipStreamSynchronize(h->stream));
        memset(out, 0, (size_t)n * b.h * b.w * b.c * sizeof(float));
        CNN_HIP(h, hipMemcpy2D(out, (size_t)b.c * sizeof(float), h->dbuf[0], 16, 16, (size_t)n * b.h * b.w, hipMemcpyDeviceToHost));
        return CTX_OK;
    }
    CNN_HIP(h, hipMemcpyAsync(out, h->dbuf[index], (size_t)n * b.h * b.w * b.c * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    CNN_HIP(h, hipStreamSynchronize(h->stream));
    return CTX_OK;
}

// per-op HIP-event times (ms) of one forward over the n images currently in buffer 0, averaged over `iters` passes;
// ms[i] belongs to op i of the list given to ctx_cnn_create.  Measurement only.
int ctx_cnn_profile(ctx_cnn* h, int n, int iters, float* ms, int max_ops) {
    if (!h || !ms || n <= 0 || n > h->max_images || iters <= 0 || max_ops < (int)h->ops.size()) return CTX_E_INVALID;
    CNN_HIP(h, hipSetDevice(h->device));
    std::vector<hipEvent_t> ev(h->ops.size() + 1);
    for (auto& e : ev) CNN_HIP(h, hipEventCreate(&e));
    std::vector<double> acc(h->ops.size(), 0.0);
    int rc = run(h, n);                                        // warm-up: code objects
    for (int it = 0; it < iters && rc == CTX_OK; ++it) {
        rc = run(h, n, &ev);
        if (rc == CTX_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = cfail(h, CTX_E_DEVICE, "sync failed");
        for (size_t i = 0; i < h->ops.size() && rc == CTX_OK; ++i) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
            acc[i] += t;
        }
    }
    for (size_t i = 0; i < h->ops.size(); ++i) ms[i] = (float)(acc[i] / iters);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

void* ctx_cnn_stream(ctx_cnn* h) { return h ? (void*)h->stream : nullptr; }
int ctx_cnn_sync(ctx_cnn* h) {
    if (!h) return CTX_E_INVALID;
    CNN_HIP(h, hipSetDevice(h->device));
    CNN_HIP(h, hipStreamSynchronize(h->stream));
    return CTX_OK;
}

}  // extern "C"
