// rccl_shim.cpp — a test double for librccl (TEST INFRASTRUCTURE: nothing in the product links, loads or names it).
//
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), and the boxes the tests run on have one GPU, so the multi-member branch
// of sp_group's RCCL gather (spectroplot-js_amd/csrc/sp_group.hip: comms[r], stage_slot, the re-tiling, the receive addresses of both
// layouts) cannot run against the real library there.  This library exports the entry points that branch binds - ncclCommInitAll,
// ncclGroupStart / ncclGroupEnd, ncclSend / ncclRecv, ncclCommDestroy / ncclCommAbort, ncclGetVersion, ncclGetErrorString - with the
// point-to-point semantics of rccl.h, and accepts any device list:
//   * a send and a receive are only QUEUED between GroupStart and GroupEnd; GroupEnd matches them: the k-th send of rank a to peer b
//     with the k-th receive of rank b from peer a (the order in which each side posted them), byte counts equal, or the call fails;
//   * a matched pair becomes one device copy: the receiver's stream waits for everything the sender's stream had queued before the
//     exchange, copies (hipMemcpyPeerAsync between devices), and the sender's stream in turn waits for the copy - a send buffer is free
//     again, and a receive buffer filled, exactly where real RCCL says so (stream order), and nowhere earlier;
//   * outside a group a call is its own group.
// Point SPECTROPLOT_HIP_RCCL_LIB at the built .so (tests/cpp/Makefile) with SPECTROPLOT_HIP_FORCE_RCCL=1.
//
// Failure injection (environment, read at every call so a test can change it between renders):
//   RCCL_SHIM_FAIL_INIT=1         ncclCommInitAll returns ncclInvalidArgument
//   RCCL_SHIM_FAIL_SEND=k         the k-th ncclSend since the last rccl_shim_reset() returns ncclInternalError (nothing queued)
//   RCCL_SHIM_FAIL_RECV=k         the same for ncclRecv
//   RCCL_SHIM_FAIL_GROUPEND=k     the k-th outermost ncclGroupEnd fails after having started HALF of its copies
//   RCCL_SHIM_REFUSE_DUPLICATES=1 ncclCommInitAll refuses a device listed twice, as the real library does
// Counters for the tests: rccl_shim_stats(uint64_t[12]) = {comms made, comms destroyed, comms aborted, sends, receives, groups ended,
// pairs copied, bytes copied, cross-device pairs, failures injected, largest message, init calls}; rccl_shim_reset() zeroes them.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

struct World;
struct Comm {
    uint32_t magic = 0x5c0c0a11u;
    World *world = nullptr;
    int rank = 0, device = 0;
    bool alive = true;
};
struct World {
    std::vector<Comm *> comms;
    int alive = 0;
};
struct Op {
    bool send;
    Comm *comm;
    int peer;
    void *buf;
    size_t bytes;
    hipStream_t stream;
    bool matched = false;
};

std::mutex g_mu;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
uint64_t g_stats[12];
uint64_t g_sends = 0, g_recvs = 0, g_ends = 0;
std::vector<hipEvent_t> g_events;   // events of earlier groups, destroyed once they have completed

void reap_events()
{
    size_t keep = 0;
    for (hipEvent_t ev : g_events) {
        if (hipEventQuery(ev) == hipSuccess) (void)hipEventDestroy(ev);
        else g_events[keep++] = ev;
    }
    (void)hipGetLastError();
    g_events.resize(keep);
}

long env_num(const char *name)
{
    const char *v = getenv(name);
    return v && *v ? atol(v) : 0;
}

size_t type_size(int dtype)
{
    switch (dtype) {   // rccl.h ncclDataType_t
    case 0: case 1: return 1;            // int8 / char, uint8
    case 2: case 3: case 7: return 4;    // int32, uint32, float32
    case 4: case 5: case 8: return 8;    // int64, uint64, float64
    case 6: case 9: return 2;            // float16, bfloat16
    default: return 0;
    }
}

bool valid(const Comm *c) { return c && c->magic == 0x5c0c0a11u && c->alive && c->world; }

int run_group(std::vector<Op> &ops)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_stats[5]++;
    const long fail_at = env_num("RCCL_SHIM_FAIL_GROUPEND");
    const bool fail = fail_at > 0 && (long)++g_ends == fail_at;
    if (fail_at <= 0) g_ends = 0;
    // match: for every send, the first unmatched receive posted by (the communicator whose rank is the send's peer) from (the sender's rank)
    struct Pair { Op *s, *r; };
    std::vector<Pair> pairs;
    for (Op &s : ops) {
        if (!s.send) continue;
        for (Op &r : ops) {
            if (r.send || r.matched || r.comm->world != s.comm->world || r.comm->rank != s.peer || r.peer != s.comm->rank) continue;
            if (r.bytes != s.bytes) return ncclInvalidArgument;   // (real RCCL would hang or corrupt: the sizes of a pair must agree)
            r.matched = s.matched = true;
            pairs.push_back({&s, &r});
            break;
        }
        if (!s.matched) return ncclInvalidUsage;                  // a send nobody receives
    }
    for (Op &r : ops)
        if (!r.send && !r.matched) return ncclInvalidUsage;       // a receive nobody sends
    int dev0 = 0;
    (void)hipGetDevice(&dev0);
    reap_events();
    size_t done = 0;
    hipError_t e = hipSuccess;
    for (Pair &p : pairs) {
        if (fail && done >= pairs.size() / 2) break;
        hipEvent_t ready = nullptr, copied = nullptr;
        const int sd = p.s->comm->device, rd = p.r->comm->device;
        // everything the sender had queued (its render) comes first
        if (e == hipSuccess) e = hipSetDevice(sd);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ready, p.s->stream);
        if (e == hipSuccess) e = hipSetDevice(rd);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.r->stream, ready, 0);
        if (e == hipSuccess && p.s->bytes)
            e = sd == rd ? hipMemcpyAsync(p.r->buf, p.s->buf, p.s->bytes, hipMemcpyDeviceToDevice, p.r->stream)
                         : hipMemcpyPeerAsync(p.r->buf, rd, p.s->buf, sd, p.s->bytes, p.r->stream);
        // ... and the sender's stream continues once its buffer has been read
        if (e == hipSuccess) e = hipEventCreateWithFlags(&copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(copied, p.r->stream);
        if (e == hipSuccess && p.s->stream != p.r->stream) {
            e = hipSetDevice(sd);
            if (e == hipSuccess) e = hipStreamWaitEvent(p.s->stream, copied, 0);
        }
        if (ready) g_events.push_back(ready);
        if (copied) g_events.push_back(copied);
        if (e != hipSuccess) break;
        done++;
        g_stats[6]++;
        g_stats[7] += p.s->bytes;
        if (sd != rd) g_stats[8]++;
        if (p.s->bytes > g_stats[10]) g_stats[10] = p.s->bytes;
    }
    (void)hipSetDevice(dev0);
    if (e != hipSuccess) return ncclUnhandledCudaError;
    if (fail) {
        g_stats[9]++;
        return ncclInternalError;
    }
    return ncclSuccess;
}

int post(bool send, void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    if (!valid(c)) return ncclInvalidArgument;
    const size_t ts = type_size(dtype);
    if (!ts || peer < 0 || peer >= (int)c->world->comms.size() || (count && !buf)) return ncclInvalidArgument;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        g_stats[send ? 3 : 4]++;
        const long k = env_num(send ? "RCCL_SHIM_FAIL_SEND" : "RCCL_SHIM_FAIL_RECV");
        uint64_t &seen = send ? g_sends : g_recvs;
        if (k > 0 && (long)++seen == k) {
            g_stats[9]++;
            return ncclInternalError;
        }
        if (k <= 0) seen = 0;
    }
    t_ops.push_back(Op{send, c, peer, buf, count * ts, stream});
    if (t_depth == 0) {   // outside a group: a group of its own (a lone send then has no receive: usage error, as it would hang for real)
        std::vector<Op> ops;
        ops.swap(t_ops);
        return run_group(ops);
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

int ncclGetVersion(int *version)
{
    if (!version) return ncclInvalidArgument;
    *version = 29999;   // (nothing real: 2.99.99)
    return ncclSuccess;
}

const char *ncclGetErrorString(int code)
{
    switch (code) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (shim)";
    case ncclSystemError: return "unhandled system error (shim)";
    case ncclInternalError: return "internal error (shim)";
    case ncclInvalidArgument: return "invalid argument (shim)";
    case ncclInvalidUsage: return "invalid usage (shim)";
    default: return "unknown result code (shim)";
    }
}

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_stats[11]++;
    if (!comms || ndev < 1) return ncclInvalidArgument;
    if (env_num("RCCL_SHIM_FAIL_INIT")) {
        g_stats[9]++;
        return ncclInvalidArgument;
    }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) return ncclUnhandledCudaError;
    for (int i = 0; i < ndev; i++) {
        const int d = devlist ? devlist[i] : i;
        if (d < 0 || d >= have) return ncclInvalidArgument;
        if (env_num("RCCL_SHIM_REFUSE_DUPLICATES"))
            for (int k = 0; k < i; k++)
                if ((devlist ? devlist[k] : k) == d) return ncclInvalidUsage;
    }
    World *w = new World;
    for (int i = 0; i < ndev; i++) {
        Comm *c = new Comm;
        c->world = w;
        c->rank = i;
        c->device = devlist ? devlist[i] : i;
        w->comms.push_back(c);
        comms[i] = c;
    }
    w->alive = ndev;
    g_stats[0] += (uint64_t)ndev;
    return ncclSuccess;
}

static int retire(void *comm, int counter)
{
    std::lock_guard<std::mutex> lock(g_mu);
    Comm *c = (Comm *)comm;
    if (!valid(c)) return ncclInvalidArgument;
    c->alive = false;
    g_stats[counter]++;
    if (--c->world->alive == 0) {
        World *w = c->world;
        for (Comm *x : w->comms) {
            x->magic = 0;
            delete x;
        }
        delete w;
    }
    return ncclSuccess;
}

int ncclCommDestroy(void *comm) { return retire(comm, 1); }

int ncclCommAbort(void *comm) { return retire(comm, 2); }

int ncclGroupStart()
{
    t_depth++;
    return ncclSuccess;
}

int ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_group(ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    return post(true, (void *)buf, count, dtype, peer, comm, stream);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    return post(false, buf, count, dtype, peer, comm, stream);
}

void rccl_shim_stats(uint64_t *out)
{
    std::lock_guard<std::mutex> lock(g_mu);
    memcpy(out, g_stats, sizeof g_stats);
}

void rccl_shim_reset()
{
    std::lock_guard<std::mutex> lock(g_mu);
    memset(g_stats, 0, sizeof g_stats);
    g_sends = g_recvs = g_ends = 0;
}

}  // extern "C"
