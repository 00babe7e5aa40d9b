// sp_group.hip — the caller's sliced render with the merge on the device, from ONE process (include/spectroplot_hip.h, sp_group_*).
//
// The reference cuts a capture into `workers` contiguous slices (lib/samples.js:253-258), renders each on its own Worker and merges
// histograms, dBfs range and strips on the main thread (lib/spectroplot.js:1206-1244).  A group owns one context per listed device:
// slice r is uploaded to and rendered on member r, all members at once.  Where the strips meet is the caller's choice:
//   * SP_GROUP_GATHER_DEVICE: on the root member's device, without touching host memory - RCCL (grouped ncclSend / ncclRecv over xGMI)
//     when the members sit on distinct devices and librccl can be loaded, peer copies otherwise.  Peer copies and the waterfall layout's
//     receives land in the image itself; only the spectrogram layout under RCCL receives whole strips beside the image and re-tiles
//     them.  sp_merge_replies does the side outputs' merge; the merged image comes back in one copy.
//   * SP_GROUP_GATHER_HOST: in the caller's host image - every member runs the chunked host render (sp_render_strip) on its slice and
//     writes its band over its own host link; the side outputs are merged on the host.
// Built on the library's own C ABI (contexts, plans, device buffers) plus the HIP runtime for the copies; RCCL is loaded at run time
// (dlopen), so the library has no link-time dependency on it.  An RCCL failure of any kind ends in peer copies, never in a failed render.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spectroplot_hip.h"
#include "sp_formats.h"

namespace {

// the RCCL entry points the gather needs (signatures of rccl.h; ncclUint8 = 1)
struct Rccl {
    void *lib = nullptr;
    std::string loaded_from, why;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*CommAbort)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    bool load(const char *override_name)
    {
        if (lib) return true;
        std::vector<std::string> names;
        if (override_name && *override_name) names.push_back(override_name);
        else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const std::string &name : names) {
            lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) {
                loaded_from = name;
                break;
            }
            const char *de = dlerror();
            why = std::string("dlopen(") + name + "): " + (de ? de : "failed");
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        GetVersion = (decltype(GetVersion))dlsym(lib, "ncclGetVersion");
        {
            Dl_info di;   // where the loader found it
            if (CommInitAll && dladdr((void *)CommInitAll, &di) && di.dli_fname) loaded_from = di.dli_fname;
        }
        // (ncclCommAbort is part of the set: without it a half-posted exchange could not be taken back, and giving up on RCCL would
        // have to wait for kernels that may never finish)
        if (CommInitAll && CommDestroy && CommAbort && GroupStart && GroupEnd && Send && Recv) return true;
        why = loaded_from + " lacks an entry point of the gather (ncclCommInitAll / ncclCommAbort / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd)";
        dlclose(lib);
        lib = nullptr;
        return false;
    }
    std::string describe(int code) const { return GetErrorString ? GetErrorString(code) : ("ncclResult " + std::to_string(code)); }
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
    bool peer_ok = true;            // the root can address this member's memory (same device, or peer access enabled both ways)
    sp_context *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t started = nullptr, rendered = nullptr, done = nullptr;
    sp_plan *plan = nullptr;
    DevBuf strip, small;
    int status = SP_OK;
    std::string error;
    // host mode: the slice's side outputs and the host clock of its render
    std::vector<uint64_t> h_hist;
    double h_minmax[2] = {0.0, -200.0};
    double host_ms = 0;
};

enum { kRcclUntried = 0, kRcclReady = 1, kRcclFailed = -1 };

}  // namespace

struct sp_group {
    std::vector<Member> m;
    std::string error, note;
    // the request the members' plans were built from
    bool have_plan = false;
    sp_request req{};
    std::vector<double> window;
    std::vector<uint8_t> lut;
    // root-side gather targets
    DevBuf staging, smalls, image, merged;
    hipEvent_t gathered = nullptr, downloaded = nullptr;
    std::vector<uint8_t> host_small;
    // transport of the last render: 0 none (one member), 1 RCCL, 2 peer copies, 3 host
    int transport = 0;
    bool distinct = false;
    bool no_rccl = false, force_rccl = false;
    std::string rccl_lib;
    Rccl rccl;
    int rccl_state = kRcclUntried;
    std::vector<void *> comms;
    double t_render = 0, t_gather = 0, t_download = 0;
};

namespace {

int gfail(sp_group *g, int code, const std::string &msg)
{
    if (g) g->error = msg;
    return code;
}

void add_note(sp_group *g, const std::string &msg)
{
    if (g->note.find(msg) != std::string::npos) return;
    if (!g->note.empty()) g->note += "; ";
    g->note += msg;
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

void drain(sp_group *g)
{
    for (Member &o : g->m) {
        (void)hipSetDevice(o.device);
        if (o.stream) (void)hipStreamSynchronize(o.stream);
    }
    if (!g->m.empty()) (void)hipSetDevice(g->m[0].device);
}

// RCCL is out for this group from now on: whatever its kernels were doing is aborted, the streams are drained, and the reason is kept.
void give_up_rccl(sp_group *g, const std::string &why)
{
    for (void *c : g->comms)
        if (c) {
            (void)g->rccl.CommAbort(c);
        }
    g->comms.clear();
    g->rccl_state = kRcclFailed;
    drain(g);
    (void)hipGetLastError();
    add_note(g, "RCCL not used: " + why + " (peer copies instead)");
}

// members run side by side on their own host threads when they sit on distinct devices (each thread drives its own GPU: uploads and
// downloads over N host links at once); members that share a device take turns on this thread - their kernels still overlap on the
// device's streams.  A thread that cannot be started runs its member here.
template <typename F>
void for_each_member(sp_group *g, F &&run)
{
    const int count = (int)g->m.size();
    if (count == 1 || !g->distinct) {
        for (int r = 0; r < count; r++) run(r);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)count);
    for (int r = 0; r < count; r++) {
        try {
            th.emplace_back(run, r);
        } catch (...) {
            run(r);
        }
    }
    for (std::thread &t : th) t.join();
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
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
    {
        auto on = [](const char *name) {
            const char *v = getenv(name);
            return v && *v && strcmp(v, "0") != 0;
        };
        g->no_rccl = on("SPECTROPLOT_HIP_NO_RCCL");
        g->force_rccl = on("SPECTROPLOT_HIP_FORCE_RCCL") && !g->no_rccl;
        const char *lib = getenv("SPECTROPLOT_HIP_RCCL_LIB");
        if (lib) g->rccl_lib = lib;
    }
    int rc = SP_OK;
    for (int i = 0; i < count && rc == SP_OK; i++) {
        Member &mb = g->m[(size_t)i];
        mb.device = devices[i];
        rc = sp_context_create(devices[i], &mb.ctx);
        if (rc) break;
        if (hipSetDevice(mb.device) != hipSuccess || hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking) != hipSuccess
            || hipEventCreate(&mb.started) != hipSuccess || hipEventCreate(&mb.rendered) != hipSuccess || hipEventCreate(&mb.done) != hipSuccess) {
            rc = SP_ERR_HIP;
            break;
        }
        rc = sp_context_set_stream(mb.ctx, mb.stream);
    }
    if (rc == SP_OK
        && (hipSetDevice(g->m[0].device) != hipSuccess || hipEventCreate(&g->gathered) != hipSuccess || hipEventCreate(&g->downloaded) != hipSuccess))
        rc = SP_ERR_HIP;
    // peers: the root and a member on another device address each other's memory (peer copies; RCCL sets up its own paths).  A pair
    // that cannot is served by hipMemcpyPeerAsync into a staging block, and the note says so.
    if (rc == SP_OK && count > 1) {
        const int root = g->m[0].device;
        for (int i = 1; i < count; i++) {
            Member &mb = g->m[(size_t)i];
            if (mb.device == root) continue;
            auto enable = [&](int from, int to) {
                int can = 0;
                hipError_t e = hipDeviceCanAccessPeer(&can, from, to);
                if (e == hipSuccess && !can) return std::string("device ") + std::to_string(from) + " cannot access device " + std::to_string(to);
                if (e == hipSuccess) e = hipSetDevice(from);
                if (e == hipSuccess) e = hipDeviceEnablePeerAccess(to, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    e = hipSuccess;
                }
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    return std::string("peer access ") + std::to_string(from) + " -> " + std::to_string(to) + ": " + hipGetErrorString(e);
                }
                return std::string();
            };
            const std::string a = enable(root, mb.device), b = enable(mb.device, root);
            if (!a.empty() || !b.empty()) {
                mb.peer_ok = false;
                add_note(g, (a.empty() ? b : a) + " (copies to the root go through hipMemcpyPeerAsync and a staging block)");
            }
        }
        (void)hipSetDevice(root);
        // (tests) treat every other member as one the root cannot address: its strip is staged beside the image and re-tiled
        const char *np = getenv("SPECTROPLOT_HIP_ASSUME_NO_PEER");
        if (np && *np && strcmp(np, "0") != 0) {
            for (int i = 1; i < count; i++) g->m[(size_t)i].peer_ok = false;
            add_note(g, "SPECTROPLOT_HIP_ASSUME_NO_PEER: copies to the root go through a staging block");
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
        g->staging.release();
        g->smalls.release();
        g->image.release();
        g->merged.release();
        if (g->gathered) (void)hipEventDestroy(g->gathered);
        if (g->downloaded) (void)hipEventDestroy(g->downloaded);
    }
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.device);
        mb.strip.release();
        mb.small.release();
        if (mb.ctx) {
            (void)sp_context_set_stream(mb.ctx, nullptr);
            sp_context_destroy(mb.ctx);
        }
        if (mb.started) (void)hipEventDestroy(mb.started);
        if (mb.rendered) (void)hipEventDestroy(mb.rendered);
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
    return g->transport == 1 ? "rccl" : g->transport == 2 ? "peer" : g->transport == 3 ? "host" : "none";
}

extern "C" const char *sp_group_transport_note(const sp_group *g) { return g ? g->note.c_str() : ""; }

extern "C" int sp_group_rccl_info(const sp_group *g, char *text, size_t capacity)
{
    if (!g || !text || !capacity) return SP_ERR_INVALID_ARG;
    std::string t;
    if (g->rccl.lib) {
        int v = 0;
        if (g->rccl.GetVersion) (void)g->rccl.GetVersion(&v);
        t = g->rccl.loaded_from + ", ncclGetVersion " + std::to_string(v) + ", " + std::to_string(g->comms.size()) + " communicator(s)";
    }
    snprintf(text, capacity, "%s", t.c_str());
    return SP_OK;
}

extern "C" int sp_group_last_timings(const sp_group *g, double *render_ms, double *gather_ms, double *download_ms)
{
    if (!g) return SP_ERR_INVALID_ARG;
    if (render_ms) *render_ms = g->t_render;
    if (gather_ms) *gather_ms = g->t_gather;
    if (download_ms) *download_ms = g->t_download;
    return SP_OK;
}

extern "C" int sp_group_root_bytes(const sp_group *g, size_t *image_bytes, size_t *staging_bytes)
{
    if (!g) return SP_ERR_INVALID_ARG;
    if (image_bytes) *image_bytes = g->image.cap;
    if (staging_bytes) *staging_bytes = g->staging.cap;
    return SP_OK;
}

// ---- SP_GROUP_GATHER_HOST: N host links side by side, nothing gathered on a device ----------------------------------------------------
static int render_to_host(sp_group *g, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply)
{
    const int count = (int)g->m.size();
    const spfmt::Format f = spfmt::describe(req->format);
    const size_t n = (size_t)req->n, L = (size_t)req->lut_len, W = (size_t)width;
    const size_t sw = (size_t)(width / count);                          // sliceWidth = ~~(width / workers), lib/spectroplot.js:1208
    auto run_member = [&](int r) {
        Member &mb = g->m[(size_t)r];
        mb.status = SP_OK;
        size_t b0 = 0, b1 = 0;
        sp_slice_bounds(nbytes, f.width, r, count, &b0, &b1);          // lib/samples.js:253-258
        mb.h_hist.assign(L + SP_CB_HIST_SIZE, 0);
        sp_reply hr{};
        if (reply->rgba)   // putImageData(strip, offset, 0) / (strip, 0, width - sliceWidth - offset), lib/spectroplot.js:1244
            hr.rgba = req->waterfall ? reply->rgba + 4 * n * (W - sw - sw * (size_t)r) : reply->rgba + 4 * sw * (size_t)r;
        hr.c_hist = mb.h_hist.data();
        hr.cb_hist = mb.h_hist.data() + L;
        hr.dbfs_minmax = mb.h_minmax;
        hr.gauge_mins = reply->gauge_mins ? reply->gauge_mins + sw * (size_t)r : nullptr;
        hr.gauge_maxs = reply->gauge_maxs ? reply->gauge_maxs + sw * (size_t)r : nullptr;
        hr.gauge_amps = reply->gauge_amps ? reply->gauge_amps + sw * (size_t)r : nullptr;
        const double t0 = now_ms();
        mb.status = sp_render_strip(mb.ctx, req, bytes + b0, b1 - b0, (int32_t)sw, &hr, width);
        mb.host_ms = now_ms() - t0;
        if (mb.status) mb.error = sp_last_error(mb.ctx);
    };
    for_each_member(g, run_member);
    for (Member &mb : g->m)
        if (mb.status) return gfail(g, mb.status, mb.error);
    g->transport = 3;
    g->t_render = 0;
    for (Member &mb : g->m) g->t_render = mb.host_ms > g->t_render ? mb.host_ms : g->t_render;
    g->t_gather = g->t_download = 0;

    // the caller's merge (lib/spectroplot.js:1125-1126, 1229-1238)
    double mn = 0.0, mx = -200.0;
    for (Member &mb : g->m) {
        if (mb.h_minmax[0] < mn) mn = mb.h_minmax[0];
        if (mb.h_minmax[1] > mx) mx = mb.h_minmax[1];
    }
    if (reply->dbfs_minmax) {
        reply->dbfs_minmax[0] = mn;
        reply->dbfs_minmax[1] = mx;
    }
    if (reply->c_hist)
        for (size_t i = 0; i < L; i++) {
            uint64_t s = 0;
            for (Member &mb : g->m) s += mb.h_hist[i];
            reply->c_hist[i] = s;
        }
    if (reply->cb_hist)
        for (size_t i = 0; i < SP_CB_HIST_SIZE; i++) {
            uint64_t s = 0;
            for (Member &mb : g->m) s += mb.h_hist[L + i];
            reply->cb_hist[i] = s;
        }
    // what no slice draws stays clear, as on the caller's fresh canvas (:1208: columns workers * sliceWidth ... width - 1)
    const size_t rest = W - sw * (size_t)count;
    if (rest) {
        if (reply->rgba) {
            if (req->waterfall) memset(reply->rgba, 0, 4 * n * rest);
            else
                for (size_t y = 0; y < n; y++) memset(reply->rgba + 4 * (W * y + sw * (size_t)count), 0, 4 * rest);
        }
        for (uint8_t *gp : {reply->gauge_mins, reply->gauge_maxs, reply->gauge_amps})
            if (gp) memset(gp + sw * (size_t)count, 0, rest);
    }
    return SP_OK;
}

extern "C" int sp_group_render_ex(sp_group *g, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply,
                                  int32_t gather)
{
    if (!g || !req || !reply) return SP_ERR_INVALID_ARG;
    if (gather != SP_GROUP_GATHER_DEVICE && gather != SP_GROUP_GATHER_HOST) return gfail(g, SP_ERR_INVALID_ARG, "unknown gather mode");
    if (width < 0) return gfail(g, SP_ERR_INVALID_ARG, "width < 0");
    if (nbytes && !bytes) return gfail(g, SP_ERR_INVALID_ARG, "bytes is null");
    if (req->format < 0 || req->format >= SP_FMT_COUNT) return gfail(g, SP_ERR_INVALID_ARG, "unknown format id");
    // (everything the plan cache below dereferences; the rest of the request is validated where the plans are made)
    if (req->n < 1 || (req->n & (req->n - 1))) return gfail(g, SP_ERR_NOT_POW2, "Length is not a power of 2");
    if (req->n > SP_MAX_N) return gfail(g, SP_ERR_UNSUPPORTED, "n exceeds SP_MAX_N");
    if (req->lut_len < 1 || req->lut_len > SP_MAX_LUT) return gfail(g, SP_ERR_UNSUPPORTED, "lut_len must be 1..SP_MAX_LUT");
    if (!req->windowc || !req->lut_rgb) return gfail(g, SP_ERR_INVALID_ARG, "windowc / lut_rgb is null");
    const int count = (int)g->m.size();
    const spfmt::Format f = spfmt::describe(req->format);
    // the reference constructs its typed view over the whole buffer before it slices (lib/spectroplot.js:1096-1100)
    if (nbytes % (size_t)f.elem) return gfail(g, SP_ERR_BYTE_LENGTH, "byte length is not a multiple of the element size");
    if (gather == SP_GROUP_GATHER_HOST) return render_to_host(g, req, bytes, nbytes, width, reply);
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
    const bool want_image = reply->rgba && strip_bytes;

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
        hip(hipEventRecord(mb.started, mb.stream), "hipEventRecord");
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
        // the slice travels in chunks of frames while earlier chunks are rendered (a sparse slice: only the samples its frames read)
        const int rc = sp_plan_execute_from_host(mb.plan, bytes + b0, b1 - b0, (int32_t)sw, &d);
        if (rc) {
            mb.status = rc;
            mb.error = sp_last_error(mb.ctx);
            return;
        }
        hip(hipEventRecord(mb.rendered, mb.stream), "hipEventRecord");
    };
    // (buffers first, on this thread: growing one frees the old block, and hipFree waits for the whole device - not something to do
    // next to another member's copy in flight; the capture's own staging buffer belongs to the member's context)
    for (int r = 0; r < count; r++) {
        Member &mb = g->m[(size_t)r];
        if (hipSetDevice(mb.device) != hipSuccess) return gfail(g, SP_ERR_HIP, "hipSetDevice");
        int rc = mb.strip.reserve(strip_bytes + 16);
        if (!rc) rc = mb.small.reserve(small_pitch);
        if (rc) return gfail(g, rc, "group member: out of device memory");
    }
    // ---- which transport: RCCL between distinct devices (or when forced), peer copies otherwise -------------------------------------
    bool use_rccl = !g->no_rccl && g->rccl_state != kRcclFailed && (g->force_rccl || (count > 1 && g->distinct));
    if (use_rccl && g->rccl_state == kRcclUntried) {
        if (!g->rccl.load(g->rccl_lib.c_str())) {
            give_up_rccl(g, g->rccl.why);
        } else {
            std::vector<int> devs;
            for (Member &mb : g->m) devs.push_back(mb.device);
            g->comms.assign((size_t)count, nullptr);
            const int nrc = g->rccl.CommInitAll(g->comms.data(), count, devs.data());
            if (nrc != 0) {
                for (void *&c : g->comms) c = nullptr;   // (a failed init hands out no communicators)
                give_up_rccl(g, "ncclCommInitAll: " + g->rccl.describe(nrc));
            } else {
                g->rccl_state = kRcclReady;
            }
        }
        use_rccl = g->rccl_state == kRcclReady;
    }
    // under RCCL the root is a sender like every other member only in a forced one-member group (which has nothing else to send);
    // otherwise its strip is already where the merge happens
    const int first_sender = use_rccl && g->force_rccl && count == 1 ? 0 : 1;
    // spectrogram strips are column bands: an RCCL receive is contiguous, so those strips land beside the image and are re-tiled;
    // peers that cannot address the root's memory directly need the same block
    bool stage = use_rccl && !req->waterfall;
    for (int r = 1; r < count; r++) stage = stage || !g->m[(size_t)r].peer_ok;
    const size_t staged = want_image && stage ? (size_t)(count - (use_rccl ? first_sender : 1)) : 0;
    auto stage_slot = [&](int r) { return (char *)g->staging.p + strip_bytes * (size_t)(r - (use_rccl ? first_sender : 1)); };

    hipError_t e = hipSetDevice(root.device);
    int rc = g->smalls.reserve(small_pitch * (size_t)count);
    if (!rc && staged) rc = g->staging.reserve(strip_bytes * staged + 16);
    if (!rc && reply->rgba) rc = g->image.reserve(4 * W * n + 16);
    if (!rc) rc = g->merged.reserve(rec_u64 * 8 * ((size_t)count + 1));   // the records end to end, then the merged record
    if (rc) return gfail(g, rc, "group root: out of device memory");

    for_each_member(g, run_member);
    for (Member &mb : g->m)
        if (mb.status) {
            drain(g);
            return gfail(g, mb.status, mb.error);
        }

    // ---- gather to the root's device ------------------------------------------------------------------------------------------------
    // where strip r goes in the image (lib/spectroplot.js:1244): a column band, or a row band in reverse order
    auto band = [&](int r) {
        return (char *)g->image.p + (req->waterfall ? 4 * n * (W - sw - sw * (size_t)r) : 4 * sw * (size_t)r);
    };
    // strip r from `src` into its band, on member `on`'s stream.  Row bands (waterfall) are one contiguous copy.  Column bands on the
    // image's own device go through the placement kernel (sp_place_strips; a pitched device-to-device copy of 1024 rows of 1 MiB runs at
    // ~120 GB/s here, the kernel at HBM rate: 8 GiB of config-4 strips 129 -> ~10 ms), `cnt` strips laid end to end in one launch;
    // from another device a pitched peer copy.
    std::string place_error;   // what sp_place_strips said, should it refuse
    auto place = [&](Member &on, int r, const void *src, int cnt = 1) {
        if (req->waterfall) return hipMemcpyAsync(band(r), src, strip_bytes, hipMemcpyDeviceToDevice, on.stream);
        if (on.device == root.device) {
            const int prc = sp_place_strips(on.ctx, (uint8_t *)g->image.p + 4 * sw * (size_t)r, (const uint8_t *)src, cnt, (int32_t)n, width, (int32_t)sw, 0);
            if (prc != SP_OK) place_error = std::string("sp_place_strips: ") + sp_last_error(on.ctx);
            return prc == SP_OK ? hipSuccess : hipErrorLaunchFailure;
        }
        return hipMemcpy2DAsync(band(r), 4 * W, src, 4 * sw, 4 * sw, n, hipMemcpyDeviceToDevice, on.stream);
    };
    e = hipSetDevice(root.device);
    g->transport = 0;
    // what no slice draws stays clear (:1208) - only that part is cleared: the members' copies into their bands are not ordered
    // behind the root's stream
    if (e == hipSuccess && reply->rgba && n && sw * (size_t)count < W) {
        const size_t rest = W - sw * (size_t)count;
        if (req->waterfall) e = hipMemsetAsync(g->image.p, 0, 4 * n * rest, root.stream);
        else e = hipMemset2DAsync((char *)g->image.p + 4 * sw * (size_t)count, 4 * W, 0, 4 * rest, n, root.stream);
    }
    if (e != hipSuccess) {
        drain(g);
        return gfail(g, SP_ERR_HIP, std::string("group gather: ") + hipGetErrorString(e));
    }
    bool rccl_done = false;
    if (use_rccl) {
        // one grouped exchange: every sender ships its strip and its record block, the root posts the matching receives (the waterfall
        // layout's straight into the image's row bands)
        int nrc = g->rccl.GroupStart();
        std::string where = "ncclGroupStart";
        for (int r = first_sender; r < count && nrc == 0; r++) {
            Member &mb = g->m[(size_t)r];
            if (want_image) {
                where = "ncclSend / ncclRecv of a strip";
                nrc = g->rccl.Send(mb.strip.p, strip_bytes, kNcclUint8, 0, g->comms[(size_t)r], mb.stream);
                if (!nrc) nrc = g->rccl.Recv(req->waterfall ? band(r) : stage_slot(r), strip_bytes, kNcclUint8, r, g->comms[0], root.stream);
            }
            if (!nrc) {
                where = "ncclSend / ncclRecv of a record block";
                nrc = g->rccl.Send(mb.small.p, small_pitch, kNcclUint8, 0, g->comms[(size_t)r], mb.stream);
            }
            if (!nrc) nrc = g->rccl.Recv((char *)g->smalls.p + small_pitch * (size_t)r, small_pitch, kNcclUint8, r, g->comms[0], root.stream);
        }
        const int erc = g->rccl.GroupEnd();
        if (nrc == 0 && erc != 0) {
            nrc = erc;
            where = "ncclGroupEnd";
        }
        if (nrc != 0) {
            give_up_rccl(g, where + ": " + g->rccl.describe(nrc));   // drains every stream; the renders are complete, the copies below redo the gather
        } else {
            rccl_done = true;
            g->transport = 1;
            // the strips that arrived beside the image go to their column bands (the root's stream: behind its receives), one launch
            if (want_image && !req->waterfall && count > first_sender) e = place(root, first_sender, stage_slot(first_sender), count - first_sender);
        }
    }
    if (!rccl_done && count > 1) {
        // peer copies on the sender's stream (behind its render), the root's stream waits for each
        g->transport = 2;
        for (int r = 1; r < count && e == hipSuccess; r++) {
            Member &mb = g->m[(size_t)r];
            e = hipSetDevice(mb.device);
            auto peer = [&](void *dst, const void *src, size_t nb) {
                return mb.device == root.device ? hipMemcpyAsync(dst, src, nb, hipMemcpyDeviceToDevice, mb.stream)
                                                : hipMemcpyPeerAsync(dst, root.device, src, mb.device, nb, mb.stream);
            };
            bool restage = false;
            if (e == hipSuccess && want_image) {
                if (req->waterfall) e = peer(band(r), mb.strip.p, strip_bytes);          // a contiguous band of rows
                else if (mb.peer_ok) e = place(mb, r, mb.strip.p);                       // a column band, written where it belongs
                else {
                    e = peer(stage_slot(r), mb.strip.p, strip_bytes);
                    restage = true;
                }
            }
            if (e == hipSuccess) e = peer((char *)g->smalls.p + small_pitch * (size_t)r, mb.small.p, small_pitch);
            if (e == hipSuccess) e = hipEventRecord(mb.done, mb.stream);
            if (e == hipSuccess) e = hipSetDevice(root.device);
            if (e == hipSuccess) e = hipStreamWaitEvent(root.stream, mb.done, 0);
            if (e == hipSuccess && restage) e = place(root, r, stage_slot(r));
        }
        if (e == hipSuccess) e = hipSetDevice(root.device);
    }
    // the root's own strip and record block (its stream: behind its render) unless they went through the forced exchange
    if (!(rccl_done && first_sender == 0)) {
        if (e == hipSuccess && want_image) e = place(root, 0, root.strip.p);
        if (e == hipSuccess) e = hipMemcpyAsync(g->smalls.p, root.small.p, small_pitch, hipMemcpyDeviceToDevice, root.stream);
    }
    if (e != hipSuccess) {
        drain(g);
        return gfail(g, SP_ERR_HIP, std::string("group gather: ") + (place_error.empty() ? hipGetErrorString(e) : place_error.c_str()));
    }

    // ---- the caller's merge on the root (lib/spectroplot.js:1229-1238) -----------------------------------------------------------------
    // records sit small_pitch apart: sp_merge_replies takes them end to end, so they are packed first (count small copies)
    DevBuf &packed = g->merged;   // [count records] then [merged record]
    for (int r = 0; r < count && e == hipSuccess; r++)
        e = hipMemcpyAsync((char *)packed.p + rec_u64 * 8 * (size_t)r, (char *)g->smalls.p + small_pitch * (size_t)r, rec_u64 * 8,
                           hipMemcpyDeviceToDevice, root.stream);
    uint64_t *d_merged = (uint64_t *)((char *)packed.p + rec_u64 * 8 * (size_t)count);
    if (e == hipSuccess) {
        rc = sp_merge_replies(root.ctx, packed.p, count, (int32_t)L, d_merged, d_merged + L, (double *)(d_merged + L + SP_CB_HIST_SIZE));
        if (rc) {
            drain(g);
            return gfail(g, rc, sp_last_error(root.ctx));
        }
    }
    if (e == hipSuccess) e = hipEventRecord(g->gathered, root.stream);
    if (e == hipSuccess && reply->rgba && W && n) e = hipMemcpyAsync(reply->rgba, g->image.p, 4 * W * n, hipMemcpyDeviceToHost, root.stream);
    g->host_small.resize(small_pitch * (size_t)count + rec_u64 * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(g->host_small.data(), g->smalls.p, small_pitch * (size_t)count, hipMemcpyDeviceToHost, root.stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(g->host_small.data() + small_pitch * (size_t)count, d_merged, rec_u64 * 8, hipMemcpyDeviceToHost, root.stream);
    if (e == hipSuccess) e = hipEventRecord(g->downloaded, root.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(root.stream);
    for (int r = 1; r < count; r++) {   // (the senders' streams: their part of the exchange has long finished)
        (void)hipSetDevice(g->m[(size_t)r].device);
        const hipError_t e2 = hipStreamSynchronize(g->m[(size_t)r].stream);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) {
        drain(g);
        return gfail(g, SP_ERR_HIP, std::string("group download: ") + hipGetErrorString(e));
    }

    // phase clocks (device events; a member's pair lives on its own device)
    g->t_render = g->t_gather = g->t_download = 0;
    for (Member &mb : g->m) {
        float ms = 0;
        (void)hipSetDevice(mb.device);
        if (hipEventElapsedTime(&ms, mb.started, mb.rendered) == hipSuccess && ms > g->t_render) g->t_render = ms;
    }
    (void)hipSetDevice(root.device);
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, root.rendered, g->gathered) == hipSuccess) g->t_gather = ms;
        // members that share the root's device render one after the other: the gather starts when the LAST of them has rendered
        // (events of one device can be compared; on distinct devices the members render side by side and the root's event stands for all)
        for (Member &mb : g->m)
            if (mb.device == root.device && hipEventElapsedTime(&ms, mb.rendered, g->gathered) == hipSuccess && ms < g->t_gather) g->t_gather = ms;
        if (hipEventElapsedTime(&ms, g->gathered, g->downloaded) == hipSuccess) g->t_download = ms;
        (void)hipGetLastError();
    }

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

extern "C" int sp_group_render(sp_group *g, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply)
{
    return sp_group_render_ex(g, req, bytes, nbytes, width, reply, SP_GROUP_GATHER_DEVICE);
}
