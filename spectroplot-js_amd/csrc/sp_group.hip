// sp_group.hip — the caller's sliced render with the merge on the device, from ONE process (include/spectroplot_hip.h, sp_group_*).
//
// The reference cuts a capture into `workers` contiguous slices (lib/samples.js:253-258), renders each on its own Worker and merges
// histograms, dBfs range and strips on the main thread (lib/spectroplot.js:1206-1244).  A group owns one context per listed device:
// slice r is uploaded to and rendered on member r, all members at once; the strips and the side-output records then travel to the
// root member's device without touching host memory - RCCL (grouped ncclSend / ncclRecv over xGMI) when the members sit on distinct
// devices and librccl can be loaded, peer copies (hipMemcpyPeerAsync) otherwise - where sp_merge_replies and sp_place_strips do the
// caller's merge; the merged image comes back in one copy.  Built on the library's own C ABI (contexts, plans, device buffers) plus
// the HIP runtime for the copies; RCCL is loaded at run time (dlopen), so the library has no link-time dependency on it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spectroplot_hip.h"
#include "sp_formats.h"

namespace {

// the six RCCL entry points the gather needs (signatures of rccl.h; ncclUint8 = 1)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv) return true;
        dlclose(lib);
        lib = nullptr;
        return false;
    }
};
constexpr int kNcclUint8 = 1;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, bytes + 256) != hipSuccess) return SP_ERR_NOMEM;
        cap = bytes + 256;
        return SP_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Member {
    int device = 0;
    sp_context *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    sp_plan *plan = nullptr;
    DevBuf in, strip, small;
    int status = SP_OK;
    std::string error;
};

}  // namespace

struct sp_group {
    std::vector<Member> m;
    std::string error;
    // the request the members' plans were built from
    bool have_plan = false;
    sp_request req{};
    std::vector<double> window;
    std::vector<uint8_t> lut;
    // root-side gather targets
    DevBuf strips, smalls, image, merged;
    std::vector<uint8_t> host_small;
    // transport of the last render: 0 none (one member), 1 RCCL, 2 peer copies
    int transport = 0;
    bool distinct = false;
    Rccl rccl;
    std::vector<void *> comms;
};

namespace {

int gfail(sp_group *g, int code, const std::string &msg)
{
    if (g) g->error = msg;
    return code;
}

bool same_request(const sp_group *g, const sp_request *r)
{
    const sp_request &q = g->req;
    if (!g->have_plan || q.format != r->format || q.n != r->n || q.channel_mode != r->channel_mode || q.waterfall != r->waterfall
        || q.lut_len != r->lut_len)
        return false;
    if (memcmp(&q.block_norm, &r->block_norm, 8) || memcmp(&q.gain, &r->gain, 8) || memcmp(&q.range, &r->range, 8)) return false;
    if (g->window.size() != (size_t)r->n || memcmp(g->window.data(), r->windowc, sizeof(double) * (size_t)r->n)) return false;
    return g->lut.size() == 3 * (size_t)r->lut_len && memcmp(g->lut.data(), r->lut_rgb, g->lut.size()) == 0;
}

void drop_plans(sp_group *g)
{
    for (Member &mb : g->m) {
        if (mb.plan) sp_plan_destroy(mb.plan);
        mb.plan = nullptr;
    }
    g->have_plan = false;
}

}  // namespace

extern "C" int sp_group_create(const int32_t *devices, int32_t count, sp_group **out)
{
    if (!out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || count < 1 || count > 64) return SP_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SP_ERR_NO_DEVICE;
    for (int i = 0; i < count; i++)
        if (devices[i] < 0 || devices[i] >= ndev) return SP_ERR_INVALID_ARG;
    sp_group *g = new (std::nothrow) sp_group;
    if (!g) return SP_ERR_NOMEM;
    g->m.resize((size_t)count);
    g->distinct = true;
    for (int i = 0; i < count; i++)
        for (int k = 0; k < i; k++) g->distinct = g->distinct && devices[i] != devices[k];
    int rc = SP_OK;
    for (int i = 0; i < count && rc == SP_OK; i++) {
        Member &mb = g->m[(size_t)i];
        mb.device = devices[i];
        rc = sp_context_create(devices[i], &mb.ctx);
        if (rc) break;
        if (hipSetDevice(mb.device) != hipSuccess || hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking) != hipSuccess
            || hipEventCreateWithFlags(&mb.done, hipEventDisableTiming) != hipSuccess) {
            rc = SP_ERR_HIP;
            break;
        }
        rc = sp_context_set_stream(mb.ctx, mb.stream);
    }
    // peers: the root reads what the others wrote (peer copies), RCCL sets up its own paths
    if (rc == SP_OK && count > 1) {
        for (int i = 1; i < count; i++) {
            if (g->m[(size_t)i].device == g->m[0].device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->m[0].device, g->m[(size_t)i].device) == hipSuccess && can) {
                (void)hipSetDevice(g->m[0].device);
                (void)hipDeviceEnablePeerAccess(g->m[(size_t)i].device, 0);   // (already enabled: an error that does not matter)
                (void)hipSetDevice(g->m[(size_t)i].device);
                (void)hipDeviceEnablePeerAccess(g->m[0].device, 0);
                (void)hipGetLastError();
            }
        }
    }
    if (rc != SP_OK) {
        sp_group_destroy(g);
        return rc;
    }
    *out = g;
    return SP_OK;
}

extern "C" void sp_group_destroy(sp_group *g)
{
    if (!g) return;
    for (Member &mb : g->m) {
        if (mb.ctx) {
            (void)hipSetDevice(mb.device);
            if (mb.stream) (void)hipStreamSynchronize(mb.stream);
        }
    }
    for (void *c : g->comms)
        if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
    drop_plans(g);
    if (!g->m.empty()) {
        (void)hipSetDevice(g->m[0].device);
        g->strips.release();
        g->smalls.release();
        g->image.release();
        g->merged.release();
    }
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.device);
        mb.in.release();
        mb.strip.release();
        mb.small.release();
        if (mb.ctx) {
            (void)sp_context_set_stream(mb.ctx, nullptr);
            sp_context_destroy(mb.ctx);
        }
        if (mb.done) (void)hipEventDestroy(mb.done);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
    }
    delete g;
}

extern "C" int sp_group_size(const sp_group *g) { return g ? (int)g->m.size() : 0; }

extern "C" const char *sp_group_last_error(const sp_group *g) { return g ? g->error.c_str() : ""; }

extern "C" const char *sp_group_transport(const sp_group *g)
{
    if (!g) return "";
    return g->transport == 1 ? "rccl" : g->transport == 2 ? "peer" : "none";
}

extern "C" int sp_group_render(sp_group *g, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply)
{
    if (!g || !req || !reply) return SP_ERR_INVALID_ARG;
    if (width < 0) return gfail(g, SP_ERR_INVALID_ARG, "width < 0");
    if (nbytes && !bytes) return gfail(g, SP_ERR_INVALID_ARG, "bytes is null");
    if (req->format < 0 || req->format >= SP_FMT_COUNT) return gfail(g, SP_ERR_INVALID_ARG, "unknown format id");
    const int count = (int)g->m.size();
    const spfmt::Format f = spfmt::describe(req->format);
    // the reference constructs its typed view over the whole buffer before it slices (lib/spectroplot.js:1096-1100)
    if (nbytes % (size_t)f.elem) return gfail(g, SP_ERR_BYTE_LENGTH, "byte length is not a multiple of the element size");
    Member &root = g->m[0];

    // plans: one per member (its tables live on its device), kept while the request's constants repeat
    if (!same_request(g, req)) {
        drop_plans(g);
        for (Member &mb : g->m) {
            const int rc = sp_plan_create(mb.ctx, req, &mb.plan);
            if (rc) {
                const std::string msg = sp_last_error(mb.ctx);
                drop_plans(g);
                return gfail(g, rc, msg);
            }
        }
        g->req = *req;
        g->window.assign(req->windowc, req->windowc + req->n);
        g->lut.assign(req->lut_rgb, req->lut_rgb + 3 * (size_t)req->lut_len);
        g->req.windowc = nullptr;
        g->req.lut_rgb = nullptr;
        g->have_plan = true;
    }

    const size_t n = (size_t)req->n, L = (size_t)req->lut_len, W = (size_t)width;
    const size_t sw = (size_t)(width / count);                          // sliceWidth = ~~(width / workers), lib/spectroplot.js:1208
    const size_t strip_bytes = 4 * sw * n;
    const size_t rec_u64 = L + SP_CB_HIST_SIZE + 2;                     // [c_hist | cB_hist | dBfs_min, dBfs_max]
    const size_t small_bytes = rec_u64 * 8 + 3 * sw;                    // ... followed by the slice's three gauge arrays
    const size_t small_pitch = (small_bytes + 15) & ~(size_t)15;

    // ---- every member: its slice to its device, rendered there (all members at once) -------------------------------------------
    auto run_member = [&](int r) {
        Member &mb = g->m[(size_t)r];
        mb.status = SP_OK;
        size_t b0 = 0, b1 = 0;
        sp_slice_bounds(nbytes, f.width, r, count, &b0, &b1);          // lib/samples.js:253-258
        auto hip = [&](hipError_t e, const char *what) {
            if (e != hipSuccess && mb.status == SP_OK) {
                mb.status = SP_ERR_HIP;
                mb.error = std::string(what) + ": " + hipGetErrorString(e);
            }
        };
        hip(hipSetDevice(mb.device), "hipSetDevice");
        if (mb.status) return;
        int rc = SP_OK;
        if (b1 > b0) hip(hipMemcpyAsync(mb.in.p, bytes + b0, b1 - b0, hipMemcpyHostToDevice, mb.stream), "slice upload");
        if (mb.status) return;
        uint64_t *d_c = (uint64_t *)mb.small.p;
        uint8_t *d_g = (uint8_t *)(d_c + rec_u64);
        sp_reply d{};
        d.rgba = reply->rgba ? (uint8_t *)mb.strip.p : nullptr;
        d.c_hist = d_c;
        d.cb_hist = d_c + L;
        d.dbfs_minmax = (double *)(d_c + L + SP_CB_HIST_SIZE);
        d.gauge_mins = d_g;
        d.gauge_maxs = d_g + sw;
        d.gauge_amps = d_g + 2 * sw;
        rc = sp_plan_execute(mb.plan, mb.in.p, b1 - b0, (int32_t)sw, &d);
        if (rc) {
            mb.status = rc;
            mb.error = sp_last_error(mb.ctx);
            return;
        }
        hip(hipEventRecord(mb.done, mb.stream), "hipEventRecord");
    };
    // (buffers first, on this thread: growing one frees the old block, and hipFree waits for the whole device - not something to do
    // next to another member's copy in flight)
    for (int r = 0; r < count; r++) {
        Member &mb = g->m[(size_t)r];
        size_t b0 = 0, b1 = 0;
        sp_slice_bounds(nbytes, f.width, r, count, &b0, &b1);
        if (hipSetDevice(mb.device) != hipSuccess) return gfail(g, SP_ERR_HIP, "hipSetDevice");
        int rc = mb.in.reserve(b1 - b0 + 16);
        if (!rc) rc = mb.strip.reserve(strip_bytes + 16);
        if (!rc) rc = mb.small.reserve(small_pitch);
        if (rc) return gfail(g, rc, "group member: out of device memory");
    }
    // one host thread per member when the members sit on distinct devices (each thread drives its own GPU: the pageable uploads run side
    // by side); members that share a device take turns on this thread - their kernels still overlap on the device's streams
    if (count == 1 || !g->distinct) {
        for (int r = 0; r < count; r++) run_member(r);
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < count; r++) th.emplace_back(run_member, r);
        for (std::thread &t : th) t.join();
    }
    for (Member &mb : g->m)
        if (mb.status) {
            for (Member &o : g->m) {
                (void)hipSetDevice(o.device);
                (void)hipStreamSynchronize(o.stream);
            }
            return gfail(g, mb.status, mb.error);
        }

    // ---- gather to the root's device ------------------------------------------------------------------------------------------------
    hipError_t e = hipSetDevice(root.device);
    int rc = g->smalls.reserve(small_pitch * (size_t)count);
    if (!rc && reply->rgba) rc = g->strips.reserve(strip_bytes * (size_t)count + 16);
    if (!rc && reply->rgba) rc = g->image.reserve(4 * W * n + 16);
    if (!rc) rc = g->merged.reserve(rec_u64 * 8 * ((size_t)count + 1));   // the records end to end, then the merged record
    if (rc) return gfail(g, rc, "group root: out of device memory");
    g->transport = 0;
    if (count > 1) {
        bool use_rccl = g->distinct && !getenv("SPECTROPLOT_HIP_NO_RCCL");
        if (use_rccl && g->comms.empty()) {
            use_rccl = g->rccl.load();
            if (use_rccl) {
                std::vector<int> devs;
                for (Member &mb : g->m) devs.push_back(mb.device);
                g->comms.assign((size_t)count, nullptr);
                if (g->rccl.CommInitAll(g->comms.data(), count, devs.data()) != 0) {
                    g->comms.clear();
                    use_rccl = false;
                }
            }
        }
        if (use_rccl && !g->comms.empty()) {
            // one grouped exchange: every member sends its strip and its record block, the root posts the matching receives
            g->transport = 1;
            int nrc = g->rccl.GroupStart();
            for (int r = 1; r < count && nrc == 0; r++) {
                Member &mb = g->m[(size_t)r];
                if (reply->rgba && strip_bytes) {
                    nrc = g->rccl.Send(mb.strip.p, strip_bytes, kNcclUint8, 0, g->comms[(size_t)r], mb.stream);
                    if (!nrc) nrc = g->rccl.Recv((char *)g->strips.p + strip_bytes * (size_t)r, strip_bytes, kNcclUint8, r, g->comms[0], root.stream);
                }
                if (!nrc) nrc = g->rccl.Send(mb.small.p, small_pitch, kNcclUint8, 0, g->comms[(size_t)r], mb.stream);
                if (!nrc) nrc = g->rccl.Recv((char *)g->smalls.p + small_pitch * (size_t)r, small_pitch, kNcclUint8, r, g->comms[0], root.stream);
            }
            const int erc = g->rccl.GroupEnd();
            if (nrc == 0) nrc = erc;
            if (nrc != 0)
                return gfail(g, SP_ERR_HIP, std::string("RCCL gather: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(nrc) : "error"));
        } else {
            // peer copies on the sender's stream (behind its render), the root's stream waits for each
            g->transport = 2;
            for (int r = 1; r < count && e == hipSuccess; r++) {
                Member &mb = g->m[(size_t)r];
                e = hipSetDevice(mb.device);
                auto peer = [&](void *dst, const void *src, size_t nb) {
                    return mb.device == root.device ? hipMemcpyAsync(dst, src, nb, hipMemcpyDeviceToDevice, mb.stream)
                                                    : hipMemcpyPeerAsync(dst, root.device, src, mb.device, nb, mb.stream);
                };
                if (e == hipSuccess && reply->rgba && strip_bytes) e = peer((char *)g->strips.p + strip_bytes * (size_t)r, mb.strip.p, strip_bytes);
                if (e == hipSuccess) e = peer((char *)g->smalls.p + small_pitch * (size_t)r, mb.small.p, small_pitch);
                if (e == hipSuccess) e = hipEventRecord(mb.done, mb.stream);
                if (e == hipSuccess) e = hipSetDevice(root.device);
                if (e == hipSuccess) e = hipStreamWaitEvent(root.stream, mb.done, 0);
            }
            if (e == hipSuccess) e = hipSetDevice(root.device);
        }
    }
    // the root's own strip and record block (its stream: behind its render)
    if (e == hipSuccess && reply->rgba && strip_bytes) e = hipMemcpyAsync(g->strips.p, root.strip.p, strip_bytes, hipMemcpyDeviceToDevice, root.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(g->smalls.p, root.small.p, small_pitch, hipMemcpyDeviceToDevice, root.stream);
    if (e != hipSuccess) return gfail(g, SP_ERR_HIP, std::string("group gather: ") + hipGetErrorString(e));

    // ---- the caller's merge on the root (lib/spectroplot.js:1229-1244) ------------------------------------------------------------------
    // records sit small_pitch apart: sp_merge_replies takes them end to end, so they are packed first (count small copies)
    DevBuf &packed = g->merged;   // [count records] then [merged record]
    for (int r = 0; r < count && e == hipSuccess; r++)
        e = hipMemcpyAsync((char *)packed.p + rec_u64 * 8 * (size_t)r, (char *)g->smalls.p + small_pitch * (size_t)r, rec_u64 * 8,
                           hipMemcpyDeviceToDevice, root.stream);
    if (e != hipSuccess) return gfail(g, SP_ERR_HIP, std::string("group merge: ") + hipGetErrorString(e));
    uint64_t *d_merged = (uint64_t *)((char *)packed.p + rec_u64 * 8 * (size_t)count);
    rc = sp_merge_replies(root.ctx, packed.p, count, (int32_t)L, d_merged, d_merged + L, (double *)(d_merged + L + SP_CB_HIST_SIZE));
    if (rc) return gfail(g, rc, sp_last_error(root.ctx));
    if (reply->rgba && W && n) {
        if (sw * (size_t)count < W) e = hipMemsetAsync(g->image.p, 0, 4 * W * n, root.stream);   // un-rendered columns stay clear (:1208)
        if (e == hipSuccess && sw) {
            rc = sp_place_strips(root.ctx, (uint8_t *)g->image.p, (const uint8_t *)g->strips.p, count, (int32_t)n, width, (int32_t)sw,
                                 req->waterfall ? 1 : 0);
            if (rc) return gfail(g, rc, sp_last_error(root.ctx));
        }
        if (e == hipSuccess) e = hipMemcpyAsync(reply->rgba, g->image.p, 4 * W * n, hipMemcpyDeviceToHost, root.stream);
    }
    g->host_small.resize(small_pitch * (size_t)count + rec_u64 * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(g->host_small.data(), g->smalls.p, small_pitch * (size_t)count, hipMemcpyDeviceToHost, root.stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(g->host_small.data() + small_pitch * (size_t)count, d_merged, rec_u64 * 8, hipMemcpyDeviceToHost, root.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(root.stream);
    for (int r = 1; r < count; r++) {   // (the senders' streams: their part of an RCCL exchange has long finished)
        (void)hipSetDevice(g->m[(size_t)r].device);
        const hipError_t e2 = hipStreamSynchronize(g->m[(size_t)r].stream);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) return gfail(g, SP_ERR_HIP, std::string("group download: ") + hipGetErrorString(e));

    const uint8_t *hm = g->host_small.data() + small_pitch * (size_t)count;
    if (reply->c_hist) memcpy(reply->c_hist, hm, L * 8);
    if (reply->cb_hist) memcpy(reply->cb_hist, hm + L * 8, SP_CB_HIST_SIZE * 8);
    if (reply->dbfs_minmax) memcpy(reply->dbfs_minmax, hm + (L + SP_CB_HIST_SIZE) * 8, 16);
    // gauges: slice r's at columns [r * sliceWidth, (r + 1) * sliceWidth), the rest clear
    uint8_t *gs[3] = {reply->gauge_mins, reply->gauge_maxs, reply->gauge_amps};
    for (int k = 0; k < 3; k++) {
        if (!gs[k]) continue;
        memset(gs[k], 0, W);
        for (int r = 0; r < count; r++) memcpy(gs[k] + sw * (size_t)r, g->host_small.data() + small_pitch * (size_t)r + rec_u64 * 8 + sw * (size_t)k, sw);
    }
    return SP_OK;
}
