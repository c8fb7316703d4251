// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product: a stand-in for librccl that lets 2 ... 8 PROCESSES ON ONE GPU run the
// data-parallel path of libctxtrans (ctx_dp_init / ctx_dp_train_step / ctx_dp_scalars, include/ctxtrans.h) exactly as N
// ranks on N GPUs would.  The build's gpurun boxes have one GPU; real RCCL refuses two ranks on one device, so without this
// the two-bucket / second-stream schedule of ctx_dp_train_step would first execute with N > 1 on the driver's 8-GPU node.
//
// Loaded through CTX_RCCL_LIB (ctx_dp.cpp: rccl_load), it exports the eight nccl* symbols the library binds.  Semantics kept:
// collectives are ASYNCHRONOUS and STREAM-ORDERED (device->pinned copy, a host function on the stream that meets the other
// ranks in a POSIX shared-memory segment, pinned->device copy); every rank sums the shards in rank order 0, 1, ..., so all
// replicas receive bit-identical results (what a ring all-reduce also guarantees).  Every wait is bounded (30 s, FAKE_RCCL_TIMEOUT_S): a rank that
// never arrives turns into ncclSystemError instead of a hung GPU box.
//
// Segment layout:  Header | rank 0 payload | rank 1 payload | ...   (payload capacity CAP bytes per rank, larger messages go in
// pieces).  The unique id carries the segment's name.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

constexpr size_t CAP = 8u << 20;           // bytes per rank per piece
constexpr int MAXR = 8;
// seconds a rank waits for the others at a barrier (FAKE_RCCL_TIMEOUT_S: eight processes starting on one box spread further than two)
const double TIMEOUT_S = getenv("FAKE_RCCL_TIMEOUT_S") ? atof(getenv("FAKE_RCCL_TIMEOUT_S")) : 30.0;

struct Header {
    std::atomic<int> arrived;              // barrier: arrivals of the current generation
    std::atomic<int> generation;
    std::atomic<int> failed;               // sticky: some rank timed out
    std::atomic<int> attached;
};

struct Comm {
    int rank, world;
    char name[64];
    Header* hdr;
    char* payload;                         // world * CAP bytes behind the header
    size_t map_bytes;
    char* pinned;                          // CAP bytes, hipHostMalloc
};

struct Op {                                // one piece of one collective, handed to the stream's host function
    Comm* c;
    size_t bytes;
    int kind;                              // 0 = sum all-reduce of floats, 1 = broadcast, 2 = sum all-reduce of doubles
    int root;
};

bool barrier(Comm* c) {
    Header* h = c->hdr;
    if (h->failed.load()) return false;
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == c->world) {
        h->arrived.store(0);
        h->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load() == gen) {
        if (h->failed.load()) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) {
            h->failed.store(1);
            fprintf(stderr, "fake_rccl: rank %d timed out at a barrier\n", c->rank);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return true;
}

void host_step(void* user) {               // runs on the stream (hipLaunchHostFunc): no HIP calls in here
    Op* op = (Op*)user;
    Comm* c = op->c;
    char* mine = c->payload + (size_t)c->rank * CAP;
    if (op->kind != 1 || c->rank == op->root) memcpy(mine, c->pinned, op->bytes);
    if (barrier(c)) {
        if (op->kind == 2) {
            double* out = (double*)c->pinned;
            const size_t n = op->bytes / sizeof(double);
            memcpy(out, c->payload, op->bytes);
            for (int r = 1; r < c->world; ++r) {
                const double* s = (const double*)(c->payload + (size_t)r * CAP);
                for (size_t i = 0; i < n; ++i) out[i] += s[i];
            }
        } else if (op->kind == 0) {
            float* out = (float*)c->pinned;
            const size_t n = op->bytes / sizeof(float);
            memcpy(out, c->payload, op->bytes);                                  // rank 0's shard first, then 1, 2, ... : same order everywhere
            for (int r = 1; r < c->world; ++r) {
                const float* s = (const float*)(c->payload + (size_t)r * CAP);
                for (size_t i = 0; i < n; ++i) out[i] += s[i];
            }
        } else {
            memcpy(c->pinned, c->payload + (size_t)op->root * CAP, op->bytes);
        }
        barrier(c);                                                              // nobody overwrites a payload another rank still reads
    }
    delete op;
}

ncclResult_t run(Comm* c, const void* send, void* recv, size_t bytes, int kind, int root, hipStream_t s) {
    if (!c || c->hdr->failed.load()) return ncclSystemError;
    for (size_t o = 0; o < bytes; o += CAP) {
        const size_t n = bytes - o < CAP ? bytes - o : CAP;
        if (hipMemcpyAsync(c->pinned, (const char*)send + o, n, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
        if (hipLaunchHostFunc(s, host_step, new Op{c, n, kind, root}) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpyAsync((char*)recv + o, c->pinned, n, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/ctxfake_%d_%ld", (int)getpid(),
             (long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->map_bytes = 4096 + (size_t)nranks * CAP;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        struct stat st;
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < c->map_bytes) {
            if (fd >= 0) { close(fd); fd = -1; }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) { delete c; return ncclSystemError; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr = (Header*)m;                   // a fresh segment is zero-filled: counters start at 0
    c->payload = (char*)m + 4096;
    if (hipHostMalloc((void**)&c->pinned, CAP, hipHostMallocDefault) != hipSuccess) { munmap(m, c->map_bytes); delete c; return ncclUnhandledCudaError; }
    c->hdr->attached.fetch_add(1);
    if (!barrier(c)) { delete c; return ncclSystemError; }      // everyone is attached before anyone proceeds
    if (rank == 0) shm_unlink(c->name);                         // the mappings keep it alive; nothing is left behind in /dev/shm
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    if (c->pinned) (void)hipHostFree(c->pinned);
    munmap((void*)c->hdr, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    if (op != ncclSum || (dt != ncclFloat && dt != ncclDouble)) return ncclInvalidArgument;
    if (dt == ncclDouble) return run((Comm*)comm, send, recv, count * sizeof(double), 2, 0, s);
    return run((Comm*)comm, send, recv, count * sizeof(float), 0, 0, s);
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t s) {
    if (dt != ncclFloat) return ncclInvalidArgument;
    return run((Comm*)comm, send, recv, count * sizeof(float), 1, root, s);
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }     // pieces are enqueued in call order on one stream: nothing to batch
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: shared-memory rendezvous failed or a rank timed out";
        case ncclInvalidArgument: return "fake_rccl: unsupported argument (f32 / f64 sum, f32 broadcast only)";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"
