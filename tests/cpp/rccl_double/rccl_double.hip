// rccl_double.hip -- TEST DOUBLE of RCCL for libmipt_multitest.so (see rccl/rccl.h in this directory).  Test infrastructure only.
//
// N logical ranks, any of them on the same HIP device.  Collectives are only accepted the way mipt_multi.cpp issues them: every
// rank's call inside ONE ncclGroupStart / ncclGroupEnd on one host thread.  ncclGroupEnd then does what the real library's kernels
// do, in stream order:
//   every rank's stream records "my send buffer is ready";
//   the root's stream waits for all of them, then moves the data --
//       ncclGather: one hipMemcpyAsync per rank into recvbuff + rank * count (the rank-major layout of the real call),
//       ncclReduce: one kernel summing the send buffers in RANK ORDER ((s0 + s1) + s2 ...), so tests can compare bit for bit
//                   with the oracle's rank-ordered sum (the real library's ring / tree order differs);
//   the root's stream records "done"; every other rank's stream waits for it (its send buffer may be reused after that).
#include "rccl/rccl.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>

struct Clique;
struct ncclComm {
    Clique *clique = nullptr;
    int rank = 0, n = 0, device = 0;
    hipEvent_t ready = nullptr, done = nullptr;
    ncclResult_t async = ncclSuccess;
    int fail_next = 0;
};
struct Clique {
    std::vector<ncclComm *> comms;
    int alive = 0;
};

namespace {

constexpr int kMaxRanks = 64;
struct Ptrs { const float *p[kMaxRanks]; };

__global__ void sum_ranks(Ptrs src, int n, size_t count, float *dst) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (size_t)gridDim.x * blockDim.x) {
        float acc = src.p[0][j];
        for (int i = 1; i < n; i++) acc = acc + src.p[i][j];
        dst[j] = acc;
    }
}

struct Op { int kind; const void *send; void *recv; size_t count; int root; ncclComm *comm; hipStream_t stream; };   // kind 0 gather, 1 reduce
thread_local int g_depth = 0;
thread_local bool g_failed = false;
thread_local std::vector<Op> g_ops;
std::mutex g_mu;                       // guards the injection table and the clique registry
std::atomic<long long> g_completed{0};
int g_inject_call[kMaxRanks] = {0}, g_inject_async[kMaxRanks] = {0};

ncclResult_t run_clique(std::vector<Op> &ops) {          // ops: one per rank of one clique, sorted by rank
    const int n = (int)ops.size(), root = ops[0].root;
    ncclComm *rc = ops[(size_t)root].comm;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return ncclUnhandledCudaError;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; i++) {
        e = hipSetDevice(ops[(size_t)i].comm->device);
        if (e == hipSuccess) e = hipEventRecord(ops[(size_t)i].comm->ready, ops[(size_t)i].stream);
    }
    hipStream_t rs = ops[(size_t)root].stream;
    if (e == hipSuccess) e = hipSetDevice(rc->device);
    for (int i = 0; i < n && e == hipSuccess; i++) e = hipStreamWaitEvent(rs, ops[(size_t)i].comm->ready, 0);
    if (e == hipSuccess && ops[0].count > 0) {
        if (ops[0].kind == 0) {
            for (int i = 0; i < n && e == hipSuccess; i++)
                e = hipMemcpyAsync((char *)ops[(size_t)root].recv + (size_t)i * ops[0].count * sizeof(float), ops[(size_t)i].send,
                                   ops[0].count * sizeof(float), hipMemcpyDefault, rs);
        } else {
            Ptrs src{};
            for (int i = 0; i < n; i++) src.p[i] = (const float *)ops[(size_t)i].send;
            size_t blocks = (ops[0].count + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(sum_ranks, dim3((unsigned)blocks), dim3(256), 0, rs, src, n, ops[0].count, (float *)ops[(size_t)root].recv);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipEventRecord(rc->done, rs);
    for (int i = 0; i < n && e == hipSuccess; i++) {
        if (i == root) continue;
        e = hipSetDevice(ops[(size_t)i].comm->device);
        if (e == hipSuccess) e = hipStreamWaitEvent(ops[(size_t)i].stream, rc->done, 0);
    }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return ncclUnhandledCudaError;
    g_completed.fetch_add(1);
    return ncclSuccess;
}

ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    const bool failed = g_failed;
    g_failed = false;
    if (failed) return ncclInternalError;                // a call of the group failed: the group moves no data
    while (!ops.empty()) {
        Clique *c = ops[0].comm->clique;
        const int n = (int)c->comms.size();
        std::vector<Op> mine((size_t)n, Op{-1, nullptr, nullptr, 0, 0, nullptr, nullptr}), rest;
        for (const Op &o : ops) {
            if (o.comm->clique != c) { rest.push_back(o); continue; }
            if (mine[(size_t)o.comm->rank].kind != -1) return ncclInvalidUsage;          // a rank twice in one group
            mine[(size_t)o.comm->rank] = o;
        }
        for (const Op &o : mine)
            if (o.kind == -1 || o.kind != mine[0].kind || o.count != mine[0].count || o.root != mine[0].root) return ncclInvalidUsage;   // a rank is missing or disagrees
        if (mine[0].root < 0 || mine[0].root >= n || !mine[(size_t)mine[0].root].recv) return ncclInvalidArgument;
        const ncclResult_t r = run_clique(mine);
        if (r != ncclSuccess) return r;
        ops.swap(rest);
    }
    return ncclSuccess;
}

ncclResult_t enqueue(int kind, const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !send || dt != ncclFloat) { if (g_depth) g_failed = true; return ncclInvalidArgument; }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (comm->rank < kMaxRanks && g_inject_call[comm->rank]) { g_inject_call[comm->rank] = 0; if (g_depth) g_failed = true; return ncclInternalError; }
    }
    g_ops.push_back(Op{kind, send, recv, count, root, comm, stream});
    if (g_depth == 0) return flush();                    // outside a group: only a one-rank communicator can complete
    return ncclSuccess;
}

} // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
    if (!comm || ndev < 1 || ndev > kMaxRanks) return ncclInvalidArgument;
    int visible = 0, prev = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || hipGetDevice(&prev) != hipSuccess) return ncclUnhandledCudaError;
    Clique *c = new Clique();
    c->alive = ndev;
    for (int i = 0; i < ndev; i++) {
        ncclComm *m = new ncclComm();
        m->clique = c; m->rank = i; m->n = ndev; m->device = devlist ? devlist[i] : i;
        c->comms.push_back(m);
        comm[i] = m;
    }
    for (ncclComm *m : c->comms) {
        if (m->device < 0 || m->device >= visible || hipSetDevice(m->device) != hipSuccess ||
            hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess) {
            for (ncclComm *x : c->comms) { if (x->ready) (void)hipEventDestroy(x->ready); if (x->done) (void)hipEventDestroy(x->done); delete x; }
            delete c;
            for (int i = 0; i < ndev; i++) comm[i] = nullptr;
            (void)hipSetDevice(prev);
            return ncclInvalidArgument;
        }
    }
    (void)hipSetDevice(prev);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclInvalidArgument;
    Clique *c = comm->clique;
    if (comm->ready) (void)hipEventDestroy(comm->ready);
    if (comm->done) (void)hipEventDestroy(comm->done);
    bool last;
    { std::lock_guard<std::mutex> lk(g_mu); last = --c->alive == 0; }
    delete comm;
    if (last) delete c;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (rccl double)";
    case ncclSystemError: return "unhandled system error (rccl double)";
    case ncclInternalError: return "internal error (rccl double)";
    case ncclInvalidArgument: return "invalid argument (rccl double)";
    case ncclInvalidUsage: return "invalid usage (rccl double)";
    case ncclRemoteError: return "remote process exited or there was a network error (rccl double)";
    default: return "unknown result code (rccl double)";
    }
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *asyncError) {
    if (!comm || !asyncError) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    *asyncError = ncclSuccess;
    if (comm->rank < kMaxRanks && g_inject_async[comm->rank]) { g_inject_async[comm->rank] = 0; *asyncError = ncclRemoteError; }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return flush();
}

ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream) {
    if (op != ncclSum) { if (g_depth) g_failed = true; return ncclInvalidArgument; }
    return enqueue(1, sendbuff, recvbuff, count, datatype, root, comm, stream);
}
ncclResult_t ncclGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm,
                        hipStream_t stream) {
    return enqueue(0, sendbuff, recvbuff, sendcount, datatype, root, comm, stream);
}

long long rccl_double_inject(int rank, int kind) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (rank >= 0 && rank < kMaxRanks) {
        if (kind == 0) g_inject_call[rank] = g_inject_async[rank] = 0;
        if (kind == 1) g_inject_call[rank] = 1;
        if (kind == 2) g_inject_async[rank] = 1;
    }
    return g_completed.load();
}

} // extern "C"
