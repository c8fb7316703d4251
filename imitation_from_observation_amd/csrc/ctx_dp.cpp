// ctx_dp.cpp -- RCCL behind the C ABI (include/ctxtrans.h: ctx_dp_*): the data-parallel exchange step of SURVEY.md 8e.
#include "ctx_internal.h"

using namespace ctxi;

// RCCL behind the C ABI (ctx_dp_init / ctx_dp_allreduce_grads).  The reference has no multi-GPU path; this is
// the path's one exchange step (8e): SUM all-reduce of the flat f32 gradient arena between backward and Adam.  librccl is
// dlopen()ed on first use -- in a process that already holds one (PyTorch ships its own librccl.so.1) the SAME copy is
// shared, a plain C/C++ host gets the system's -- so libctxtrans.so keeps loading on boxes without RCCL.
// ================================================================================================
namespace ctxi {
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
}  // namespace ctxi

extern "C" {


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
}  // extern "C"
