// addon.cc — N-API binding of the C ABI (include/spectroplot_hip.h) for Node.js.
//
// The JavaScript class HipWorker (js/hip_worker.js) is what the reference library receives through its
// `workerOrUrl` option (lib/spectroplot.js:100-116); this file only moves typed-array pointers across the
// language boundary.  No compute happens here and there is no fallback: every entry point that needs a device
// throws when the HIP library reports an error.
//
// Exports
//   deviceCount()                                  -> number
//   parseFormat(name)                              -> {id, sampleWidth}
//   window(name, n)                                -> {window: Float64Array, weight}
//   sliceBounds(nbytes, sampleWidth, index, count) -> [begin, end]
//   poolStats()                                    -> {fresh, recycled, recycledPinned, freeBytes, pinnedBytes, pinRefused, keepLimit,
//                                                      pinnedLimit, replyWeightCap} of the reply-image pool
//   cmap(name)                                     -> Uint8Array of r,g,b triples (the reference's map under its lookup rules) or null
//   cmapKeys()                                     -> the reference's colour-map keys in its table order
//   allocBuffer(nbytes)                            -> ArrayBuffer in page-locked host memory (sp_host_alloc): a request whose `buffer`
//                                                     is one of these goes to the device at the full rate of the host link
//   createContext(device)                          -> external handle
//   destroyContext(handle)                         -> releases the device context as soon as no render is in flight on it
//   render(handle, req, cb)   req = {format:int, buffer:ArrayBuffer, n, windowc:Float64Array, block_norm, gain, range,
//                                    lut:Uint8Array, width, channelMode, waterfall}
//       runs sp_render on a libuv worker thread and calls cb(err, {rgba, gauge_mins, gauge_maxs, gauge_amps: ArrayBuffer,
//       c_hist, cB_hist: Float64Array, dBfs_min, dBfs_max}) on the main thread
//   renderSync(handle, req)                        -> the same reply object, synchronously
//   renderNamed(handle, req, cb) / renderNamedSync(handle, req)   req = {format, window, cmap: strings, buffer, n, gain, range, width,
//                                    channelMode, waterfall}: sp_render_named - the library resolves the names as the reference's caller
//                                    does (lib/spectroplot.js:238-264, 1113-1146) and keeps the plan while they repeat
//   namedResolve(window, cmap)                     -> {window, cmap, lutLength}: what the two option names resolve to (sp_named_resolve)
//   planCreations(handle)                          -> how many plans this context has built (sp_context_plan_creations)
//   createGroup(devices: number[])                 -> external handle of an sp_group: one member context per listed device
//   destroyGroup(handle)                           -> like destroyContext
//   groupRender(handle, req, cb) / groupRenderSync(handle, req)   req as for render, `width` = frames of the WHOLE image, plus
//       `gather`: 'device' (default) | 'host' (sp_group_render_ex); the result carries transport, transportNote and timings:
//       sp_group_render - the caller's sliced render (lib/spectroplot.js:1206-1244) with the strips gathered device to device (RCCL or
//       peer copies) and merged on the root; the reply is the merged result (rgba = the whole image, gauges [width]) plus
//       {sliceWidth, members, transport: 'none' | 'rccl' | 'peer'}
// One render at a time per handle: a second render on a handle whose first is still in flight throws (HipWorker serialises its own).
#include <node_api.h>

#include <chrono>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sys/mman.h>
#include <string>
#include <vector>

#include "../../include/spectroplot_hip.h"

namespace {

#define NAPI_OK(env, call)                                                         \
    do {                                                                           \
        if ((call) != napi_ok) {                                                   \
            napi_throw_error((env), nullptr, "N-API call failed: " #call);        \
            return nullptr;                                                        \
        }                                                                          \
    } while (0)

napi_value throw_status(napi_env env, int status, const char *msg)
{
    std::string m = (msg && *msg) ? msg : sp_status_string(status);
    napi_value err, text, code;
    napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &text);
    napi_create_error(env, nullptr, text, &err);
    napi_create_int32(env, status, &code);
    napi_set_named_property(env, err, "status", code);
    napi_throw(env, err);
    return nullptr;
}

bool get_named(napi_env env, napi_value obj, const char *name, napi_value *out)
{
    return napi_get_named_property(env, obj, name, out) == napi_ok;
}

double get_double(napi_env env, napi_value obj, const char *name)
{
    napi_value v;
    double d = 0;
    if (get_named(env, obj, name, &v)) napi_get_value_double(env, v, &d);
    return d;
}

bool get_bool(napi_env env, napi_value obj, const char *name)
{
    napi_value v, b;
    bool r = false;
    if (get_named(env, obj, name, &v) && napi_coerce_to_bool(env, v, &b) == napi_ok) napi_get_value_bool(env, b, &r);
    return r;
}

// One device context as JavaScript sees it.  Renders run on libuv worker threads, so the sp_context must outlive every job that
// was queued on it: jobs count themselves in `inflight` (main thread only), destroyContext / the finalizer only mark the
// context closed, and whoever brings the count to zero on a closed context destroys it.
struct Ctx {
    sp_context *c = nullptr;
    sp_group *g = nullptr;    // a group handle instead (createGroup): the renders go to sp_group_render
    int inflight = 0;
    bool closed = false;
    bool collected = false;   // the JS handle is gone: the struct itself may be deleted
};

void ctx_release(Ctx *x)
{
    if (x->closed && x->inflight == 0 && x->c) {
        sp_context_destroy(x->c);
        x->c = nullptr;
    }
    if (x->closed && x->inflight == 0 && x->g) {
        sp_group_destroy(x->g);
        x->g = nullptr;
    }
    if (x->collected && x->inflight == 0) delete x;
}

// Image buffers of replies are recycled: a block that JavaScript has dropped (its ArrayBuffer was collected) serves a later reply
// of the same size.  Fresh memory is the slow part of a reply (the kernel zero-fills every page the copy from the device touches
// for the first time); a recycled block takes the copy at the rate of the host link.  V8 is told the size of every block it
// holds (napi_adjust_external_memory), so dropped replies are collected under memory pressure like any large ArrayBuffer.
// A block that comes round a second time is page-locked (on the worker thread, once): copies into it then run asynchronously at
// the rate of the host link, which is what lets sp_render overlap the image's way back with the samples' way in.
// Limits, each read once from the environment (MiB): SPECTROPLOT_HIP_POOL_KEEP_MB bounds the dropped blocks kept for reuse (default
// 3072), SPECTROPLOT_HIP_POOL_PINNED_MB the page-locked bytes of pool blocks, free or held by JavaScript (default 2048 - ten to twelve
// 64-MiB replies are alive between two collections, and a reply in a block that is not page-locked takes 4.0 instead of 2.7 ms; 0 = never
// page-lock), SPECTROPLOT_HIP_REPLY_WEIGHT_MB what V8 is told a reply weighs at most (below).
inline size_t env_mib(const char *name, size_t dflt_mib)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt_mib << 20;
    char *end = nullptr;
    const unsigned long long v = strtoull(e, &end, 10);
    return (end && *end == 0) ? (size_t)v << 20 : dflt_mib << 20;
}
enum PinState : int { kUnpinned = 0, kPinned = 1, kPinRefused = 2 };   // kPinRefused: hipHostRegister failed for this block; not retried
struct HostPool {
    struct Block { void *p; size_t size; int pin; };
    std::mutex m;
    std::vector<Block> free_blocks;
    size_t free_bytes = 0, pinned_bytes = 0;
    size_t n_fresh = 0, n_recycled = 0, n_recycled_pinned = 0, n_pin_refused = 0;   // poolStats()
    const size_t keep_bytes = env_mib("SPECTROPLOT_HIP_POOL_KEEP_MB", 3072);
    const size_t pinned_limit = env_mib("SPECTROPLOT_HIP_POOL_PINNED_MB", 2048);
    void *take(size_t size, int *pin)
    {
        *pin = kUnpinned;
        {
            std::unique_lock<std::mutex> g(m);
            for (size_t i = 0; i < free_blocks.size(); i++)
                if (free_blocks[i].size == size) {
                    void *p = free_blocks[i].p;
                    *pin = free_blocks[i].pin;
                    n_recycled++;
                    if (*pin == kPinned) n_recycled_pinned++;
                    free_bytes -= size;
                    free_blocks.erase(free_blocks.begin() + (long)i);
                    const bool try_pin = *pin == kUnpinned && size >= ((size_t)1 << 20) && pinned_bytes + size <= pinned_limit;
                    if (try_pin) pinned_bytes += size;   // reserved under the lock, returned below if the registration fails
                    g.unlock();
                    if (try_pin) {
                        if (sp_host_register(p, size) == SP_OK) *pin = kPinned;
                        else {
                            *pin = kPinRefused;
                            std::lock_guard<std::mutex> g2(m);
                            pinned_bytes -= size;
                            n_pin_refused++;
                        }
                    }
                    return p;
                }
        }
        {
            std::lock_guard<std::mutex> g(m);
            n_fresh++;
        }
        // large blocks on 2 MiB boundaries with transparent huge pages requested: the first touch of a fresh block then costs one
        // fault per 2 MiB instead of one per 4 KiB
        void *p = nullptr;
        const size_t align = size >= ((size_t)4 << 20) ? ((size_t)2 << 20) : 4096;
        if (posix_memalign(&p, align, size ? size : 1) != 0) return nullptr;
#ifdef MADV_HUGEPAGE
        if (align > 4096) (void)madvise(p, size, MADV_HUGEPAGE);
#endif
        return p;
    }
    void give(void *p, size_t size, int pin)
    {
        std::lock_guard<std::mutex> g(m);
        if (free_bytes + size > keep_bytes) {
            if (pin == kPinned) {
                sp_host_unregister(p);
                pinned_bytes -= size;
            }
            free(p);
            return;
        }
        free_blocks.push_back({p, size, pin});
        free_bytes += size;
    }
};
HostPool g_pool;

// What V8 is told a reply image weighs: its size up to 6 MiB.  At full weight a config-2 reply (64 MiB) reaches the collector's
// external-memory limit by itself: one mark-sweep per message (1.2 ms each, `node --trace-gc tools/js_dropin_bench.js`).  Capped at
// 16 MiB (rounds 2-4) the collector still ran about every fourth large reply, and its marking steps land in whatever allocates next:
// 0.25-0.3 ms of an average config-2 message, seen as `_request` taking 280-350 us instead of 40 (tools/js_message_stages.js cfg2,
// profiles/r05_host_path.txt).  At 6 MiB it runs about every tenth reply and the pool holds ten to twelve blocks (0.7 GiB of host
// memory for 64-MiB replies) instead of five or six; the blocks still return by collection only.
// The cap grows with the reply beyond 60 MiB - weight = max(cap, size / 10) - so that the collector still runs about every tenth reply
// whatever a reply weighs: with a flat 6 MiB, ten 2-GiB replies (config 5) would be 20 GiB of garbage before V8 looked.
// SPECTROPLOT_HIP_REPLY_WEIGHT_MB changes the cap (a caller that holds many replies at once can raise it to their real size).
const size_t g_reply_weight_cap = env_mib("SPECTROPLOT_HIP_REPLY_WEIGHT_MB", 6);
inline int64_t reply_weight(size_t size)
{
    const size_t floor_w = size < g_reply_weight_cap ? size : g_reply_weight_cap;
    return (int64_t)(size / 10 > floor_w ? size / 10 : floor_w);
}
struct PoolTag { size_t size; int pin; };
void pool_free_cb(napi_env env, void *data, void *hint)
{
    PoolTag *t = (PoolTag *)hint;
    int64_t total = 0;
    napi_adjust_external_memory(env, -reply_weight(t->size), &total);
    g_pool.give(data, t->size, t->pin);
    delete t;
}

struct Job {
    Ctx *owner = nullptr;
    sp_context *ctx = nullptr;
    sp_group *group = nullptr;   // a group handle's job
    int32_t gather = SP_GROUP_GATHER_DEVICE;   // ... and where its strips meet (req.gather: 'device' | 'host')
    std::string transport, note;
    double t_render = 0, t_gather = 0, t_download = 0;
    // where the job's time went on the worker thread (microseconds): reply buffers from the pool, the library call
    double us_take = 0, us_native = 0;
    sp_request req{};
    std::vector<double> window;
    std::vector<uint8_t> lut;
    // a request by option names (renderNamed): the library evaluates taper and colour map itself
    bool named = false;
    std::string nformat, nwindow, ncmap;
    const uint8_t *bytes = nullptr;
    size_t nbytes = 0;
    int32_t width = 0;
    // outputs: the image in a recycled block, the gauges malloc'd; handed to JS as external ArrayBuffers
    uint8_t *rgba = nullptr, *gmin = nullptr, *gmax = nullptr, *gamp = nullptr;
    size_t rgba_size = 0;
    int rgba_pin = kUnpinned;
    std::vector<uint64_t> c_hist, cb_hist;
    double minmax[2] = {0.0, -200.0};
    int status = SP_OK;
    std::string error;
    // async plumbing
    napi_async_work work = nullptr;
    napi_ref cb_ref = nullptr, buf_ref = nullptr, ctx_ref = nullptr;
};

void free_cb(napi_env, void *data, void *) { free(data); }

std::string get_string(napi_env env, napi_value obj, const char *name)
{
    napi_value v, sv;
    if (!get_named(env, obj, name, &v) || napi_coerce_to_string(env, v, &sv) != napi_ok) return "";
    size_t len = 0;
    napi_get_value_string_utf8(env, sv, nullptr, 0, &len);
    std::string out(len, '\0');
    napi_get_value_string_utf8(env, sv, &out[0], len + 1, &len);
    return out;
}

bool parse_request(napi_env env, napi_value handle, napi_value req, Job *j, bool named = false)
{
    void *p = nullptr;
    if (napi_get_value_external(env, handle, &p) != napi_ok || !p) {
        napi_throw_type_error(env, nullptr, "context handle expected");
        return false;
    }
    j->owner = (Ctx *)p;
    if (j->owner->closed || (!j->owner->c && !j->owner->g)) {
        napi_throw_error(env, nullptr, "context has been destroyed");
        return false;
    }
    j->ctx = j->owner->c;
    j->group = j->owner->g;
    napi_value v;
    int32_t i32 = 0;
    j->named = named;
    if (named) {
        j->nformat = get_string(env, req, "format");
        j->nwindow = get_string(env, req, "window");
        j->ncmap = get_string(env, req, "cmap");
    } else {
        get_named(env, req, "format", &v); napi_get_value_int32(env, v, &i32); j->req.format = i32;
    }
    get_named(env, req, "n", &v); napi_get_value_int32(env, v, &i32); j->req.n = i32;
    get_named(env, req, "width", &v); napi_get_value_int32(env, v, &i32); j->width = i32;
    if (j->group && get_string(env, req, "gather") == "host") j->gather = SP_GROUP_GATHER_HOST;
    j->req.channel_mode = get_bool(env, req, "channelMode");
    j->req.waterfall = get_bool(env, req, "waterfall");
    j->req.block_norm = get_double(env, req, "block_norm");
    j->req.gain = get_double(env, req, "gain");
    j->req.range = get_double(env, req, "range");

    void *data = nullptr;
    size_t len = 0;
    get_named(env, req, "buffer", &v);
    if (napi_get_arraybuffer_info(env, v, &data, &len) != napi_ok) {
        napi_throw_type_error(env, nullptr, "buffer must be an ArrayBuffer");
        return false;
    }
    j->bytes = (const uint8_t *)data;
    j->nbytes = len;
    if (named) {
        int32_t lut_len = 0;
        sp_named_resolve(j->nwindow.c_str(), j->ncmap.c_str(), nullptr, nullptr, &lut_len);   // sizes the reply's c_hist
        j->req.lut_len = lut_len;
        return true;
    }

    napi_typedarray_type tt;
    napi_value ab;
    size_t off = 0;
    get_named(env, req, "windowc", &v);
    if (napi_get_typedarray_info(env, v, &tt, &len, &data, &ab, &off) != napi_ok || tt != napi_float64_array) {
        napi_throw_type_error(env, nullptr, "windowc must be a Float64Array");
        return false;
    }
    j->window.assign((const double *)data, (const double *)data + len);
    get_named(env, req, "lut", &v);
    if (napi_get_typedarray_info(env, v, &tt, &len, &data, &ab, &off) != napi_ok || (tt != napi_uint8_array && tt != napi_uint8_clamped_array)) {
        napi_throw_type_error(env, nullptr, "lut must be a Uint8Array");
        return false;
    }
    j->lut.assign((const uint8_t *)data, (const uint8_t *)data + len);
    j->req.lut_len = (int32_t)(len / 3);
    if ((size_t)j->req.n != j->window.size() && j->req.n > 0 && j->window.size() < (size_t)j->req.n) {
        napi_throw_range_error(env, nullptr, "windowc is shorter than n");
        return false;
    }
    j->req.windowc = j->window.data();
    j->req.lut_rgb = j->lut.data();
    return true;
}

inline double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void run_job_inner(Job *j);
void run_job(Job *j)
{
    const double t0 = now_us();
    run_job_inner(j);
    j->us_native = now_us() - t0 - j->us_take;
}

void run_job_inner(Job *j)
{
    const double t_take0 = now_us();
    const size_t W = j->width > 0 ? (size_t)j->width : 0, n = j->req.n > 0 ? (size_t)j->req.n : 0;
    j->rgba_size = 4 * W * n + 1;
    j->rgba = (uint8_t *)g_pool.take(j->rgba_size, &j->rgba_pin);
    j->gmin = (uint8_t *)calloc(W + 1, 1);
    j->gmax = (uint8_t *)calloc(W + 1, 1);
    j->gamp = (uint8_t *)calloc(W + 1, 1);
    j->c_hist.assign(j->req.lut_len > 0 ? (size_t)j->req.lut_len : 0, 0);
    j->cb_hist.assign(SP_CB_HIST_SIZE, 0);
    j->us_take = now_us() - t_take0;
    if (!j->rgba || !j->gmin || !j->gmax || !j->gamp) {
        j->status = SP_ERR_NOMEM;
        j->error = "out of host memory";
        return;
    }
    sp_reply r{};
    r.rgba = j->rgba; r.gauge_mins = j->gmin; r.gauge_maxs = j->gmax; r.gauge_amps = j->gamp;
    r.c_hist = j->c_hist.data(); r.cb_hist = j->cb_hist.data(); r.dbfs_minmax = j->minmax;
    if (j->group) {
        j->status = sp_group_render_ex(j->group, &j->req, j->bytes, j->nbytes, j->width, &r, j->gather);
        if (j->status != SP_OK) j->error = sp_group_last_error(j->group);
        // (read here, on the worker thread that owns the group for the duration of the job)
        j->transport = sp_group_transport(j->group);
        j->note = sp_group_transport_note(j->group);
        sp_group_last_timings(j->group, &j->t_render, &j->t_gather, &j->t_download);
        return;
    }
    if (j->named) {
        sp_named_request nr{};
        nr.format = j->nformat.c_str();
        nr.window = j->nwindow.c_str();
        nr.cmap = j->ncmap.c_str();
        nr.n = j->req.n;
        nr.channel_mode = j->req.channel_mode;
        nr.waterfall = j->req.waterfall;
        nr.gain = j->req.gain;
        nr.range = j->req.range;
        j->status = sp_render_named(j->ctx, &nr, j->bytes, j->nbytes, j->width, &r);
    } else {
        j->status = sp_render(j->ctx, &j->req, j->bytes, j->nbytes, j->width, &r);
    }
    if (j->status != SP_OK) j->error = sp_last_error(j->ctx);
}

napi_value make_reply(napi_env env, Job *j)
{
    const size_t W = j->width > 0 ? (size_t)j->width : 0, n = (size_t)j->req.n;
    napi_value out, v;
    NAPI_OK(env, napi_create_object(env, &out));
    auto put_ab = [&](const char *name, uint8_t *&p, size_t len) {
        napi_value ab;
        if (napi_create_external_arraybuffer(env, p, len, free_cb, nullptr, &ab) == napi_ok) {
            p = nullptr;   // owned by the ArrayBuffer now
            napi_set_named_property(env, out, name, ab);
        }
    };
    {
        napi_value ab;
        PoolTag *tag = new PoolTag{j->rgba_size, j->rgba_pin};
        if (napi_create_external_arraybuffer(env, j->rgba, 4 * W * n, pool_free_cb, tag, &ab) == napi_ok) {
            int64_t total = 0;
            napi_adjust_external_memory(env, reply_weight(j->rgba_size), &total);
            j->rgba = nullptr;
            napi_set_named_property(env, out, "rgba", ab);
        } else {
            delete tag;
        }
    }
    put_ab("gauge_mins", j->gmin, W);
    put_ab("gauge_maxs", j->gmax, W);
    put_ab("gauge_amps", j->gamp, W);
    auto put_hist = [&](const char *name, const std::vector<uint64_t> &h) {
        napi_value ab, ta;
        void *data;
        if (napi_create_arraybuffer(env, h.size() * 8, &data, &ab) != napi_ok) return;
        for (size_t i = 0; i < h.size(); i++) ((double *)data)[i] = (double)h[i];
        napi_create_typedarray(env, napi_float64_array, h.size(), ab, 0, &ta);
        napi_set_named_property(env, out, name, ta);
    };
    put_hist("c_hist", j->c_hist);
    put_hist("cB_hist", j->cb_hist);
    {
        napi_value st;
        napi_create_object(env, &st);
        napi_create_double(env, j->us_take, &v); napi_set_named_property(env, st, "take_us", v);
        napi_create_double(env, j->us_native, &v); napi_set_named_property(env, st, "native_us", v);
        napi_set_named_property(env, out, "stages", st);
    }
    napi_create_double(env, j->minmax[0], &v); napi_set_named_property(env, out, "dBfs_min", v);
    napi_create_double(env, j->minmax[1], &v); napi_set_named_property(env, out, "dBfs_max", v);
    if (j->group) {
        const int members = sp_group_size(j->group);
        napi_create_int32(env, members, &v); napi_set_named_property(env, out, "members", v);
        napi_create_int32(env, members > 0 ? j->width / members : 0, &v); napi_set_named_property(env, out, "sliceWidth", v);
        napi_create_string_utf8(env, j->transport.c_str(), NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, out, "transport", v);
        napi_create_string_utf8(env, j->note.c_str(), NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, out, "transportNote", v);
        napi_value t;
        napi_create_object(env, &t);
        napi_create_double(env, j->t_render, &v); napi_set_named_property(env, t, "render_ms", v);
        napi_create_double(env, j->t_gather, &v); napi_set_named_property(env, t, "gather_ms", v);
        napi_create_double(env, j->t_download, &v); napi_set_named_property(env, t, "download_ms", v);
        napi_set_named_property(env, out, "timings", t);
    }
    return out;
}

void free_job(napi_env env, Job *j)
{
    if (j->rgba) g_pool.give(j->rgba, j->rgba_size, j->rgba_pin);
    free(j->gmin); free(j->gmax); free(j->gamp);
    if (j->ctx_ref) napi_delete_reference(env, j->ctx_ref);
    if (j->cb_ref) napi_delete_reference(env, j->cb_ref);
    if (j->buf_ref) napi_delete_reference(env, j->buf_ref);
    if (j->work) napi_delete_async_work(env, j->work);
    delete j;
}

napi_value make_error(napi_env env, Job *j)
{
    napi_value err, text, code;
    const std::string m = j->error.empty() ? sp_status_string(j->status) : j->error;
    napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &text);
    napi_create_error(env, nullptr, text, &err);
    napi_create_int32(env, j->status, &code);
    napi_set_named_property(env, err, "status", code);
    return err;
}

// An sp_context is one stream with one set of staging buffers: one render at a time.  HipWorker's promise queue already serialises
// its requests; a second render on a handle whose first is still on a libuv thread is refused here rather than left to race.
napi_value render_sync(napi_env env, napi_callback_info info, bool named)
{
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Job *j = new Job;
    if (!parse_request(env, argv[0], argv[1], j, named)) { free_job(env, j); return nullptr; }
    if (j->owner->inflight > 0) {
        free_job(env, j);
        napi_throw_error(env, nullptr, "a render is already in flight on this context");
        return nullptr;
    }
    j->owner->inflight++;
    run_job(j);
    j->owner->inflight--;
    ctx_release(j->owner);
    napi_value out = nullptr;
    if (j->status != SP_OK) napi_throw(env, make_error(env, j));
    else out = make_reply(env, j);
    free_job(env, j);
    return out;
}
napi_value RenderSync(napi_env env, napi_callback_info info) { return render_sync(env, info, false); }
napi_value RenderNamedSync(napi_env env, napi_callback_info info) { return render_sync(env, info, true); }

void exec_cb(napi_env, void *data) { run_job((Job *)data); }

void done_cb(napi_env env, napi_status, void *data)
{
    Job *j = (Job *)data;
    j->owner->inflight--;
    ctx_release(j->owner);
    napi_value cb, global, argv[2];
    napi_get_reference_value(env, j->cb_ref, &cb);
    napi_get_global(env, &global);
    if (j->status != SP_OK) {
        argv[0] = make_error(env, j);
        napi_get_undefined(env, &argv[1]);
    } else {
        napi_get_null(env, &argv[0]);
        argv[1] = make_reply(env, j);
    }
    napi_value ignored;
    napi_call_function(env, global, cb, 2, argv, &ignored);
    free_job(env, j);
}

napi_value render_async(napi_env env, napi_callback_info info, bool named)
{
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Job *j = new Job;
    if (!parse_request(env, argv[0], argv[1], j, named)) { free_job(env, j); return nullptr; }
    if (j->owner->inflight > 0) {
        free_job(env, j);
        napi_throw_error(env, nullptr, "a render is already in flight on this context");
        return nullptr;
    }
    napi_value buf, name;
    bool ok = napi_get_named_property(env, argv[1], "buffer", &buf) == napi_ok;
    ok = ok && napi_create_reference(env, buf, 1, &j->buf_ref) == napi_ok;       // keep the input alive while the worker thread reads it
    ok = ok && napi_create_reference(env, argv[2], 1, &j->cb_ref) == napi_ok;
    ok = ok && napi_create_reference(env, argv[0], 1, &j->ctx_ref) == napi_ok;   // the handle (and with it the Ctx) stays reachable while the job runs
    ok = ok && napi_create_string_utf8(env, "spectroplot_hip.render", NAPI_AUTO_LENGTH, &name) == napi_ok;
    ok = ok && napi_create_async_work(env, nullptr, name, exec_cb, done_cb, j, &j->work) == napi_ok;
    if (ok) {
        // counted only once the work is certain to run: done_cb is what takes the count down again
        j->owner->inflight++;
        if (napi_queue_async_work(env, j->work) != napi_ok) {
            j->owner->inflight--;
            ok = false;
        }
    }
    if (!ok) {
        Ctx *owner = j->owner;
        free_job(env, j);          // references, the work item, the buffers
        ctx_release(owner);        // a context closed meanwhile is destroyed here if nothing else holds it
        napi_throw_error(env, nullptr, "could not queue the render");
    }
    return nullptr;
}
napi_value Render(napi_env env, napi_callback_info info) { return render_async(env, info, false); }
napi_value RenderNamed(napi_env env, napi_callback_info info) { return render_async(env, info, true); }

napi_value DeviceCount(napi_env env, napi_callback_info)
{
    int32_t c = 0;
    sp_device_count(&c);
    napi_value v;
    napi_create_int32(env, c, &v);
    return v;
}

napi_value ParseFormat(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1], s;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    NAPI_OK(env, napi_coerce_to_string(env, argv[0], &s));
    char buf[64];
    size_t len = 0;
    napi_get_value_string_utf8(env, s, buf, sizeof buf, &len);
    int32_t id = 0, sw = 0;
    sp_format_parse(buf, &id, &sw);
    napi_value out, v;
    napi_create_object(env, &out);
    napi_create_int32(env, id, &v); napi_set_named_property(env, out, "id", v);
    napi_create_int32(env, sw, &v); napi_set_named_property(env, out, "sampleWidth", v);
    napi_create_int32(env, sp_format_element_size(id), &v); napi_set_named_property(env, out, "elementSize", v);
    return out;
}

napi_value Window(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    char name[64];
    size_t len = 0;
    int32_t n = 0;
    napi_get_value_string_utf8(env, argv[0], name, sizeof name, &len);
    napi_get_value_int32(env, argv[1], &n);
    if (n < 1) return throw_status(env, SP_ERR_INVALID_ARG, "n must be positive");
    napi_value ab, ta, out, w;
    void *data;
    NAPI_OK(env, napi_create_arraybuffer(env, (size_t)n * 8, &data, &ab));
    double weight = 0;
    const int rc = sp_window(name, n, (double *)data, &weight);
    if (rc) return throw_status(env, rc, "unknown window name");
    napi_create_typedarray(env, napi_float64_array, (size_t)n, ab, 0, &ta);
    napi_create_object(env, &out);
    napi_set_named_property(env, out, "window", ta);
    napi_create_double(env, weight, &w);
    napi_set_named_property(env, out, "weight", w);
    return out;
}

napi_value SliceBounds(napi_env env, napi_callback_info info)
{
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    double nbytes = 0;
    int32_t sw = 0, index = 0, count = 0;
    napi_get_value_double(env, argv[0], &nbytes);
    napi_get_value_int32(env, argv[1], &sw);
    napi_get_value_int32(env, argv[2], &index);
    napi_get_value_int32(env, argv[3], &count);
    size_t b = 0, e = 0;
    const int rc = sp_slice_bounds((size_t)nbytes, sw, index, count, &b, &e);
    if (rc) return throw_status(env, rc, nullptr);
    napi_value out, v;
    napi_create_array_with_length(env, 2, &out);
    napi_create_double(env, (double)b, &v); napi_set_element(env, out, 0, v);
    napi_create_double(env, (double)e, &v); napi_set_element(env, out, 1, v);
    return out;
}

void ctx_finalize(napi_env, void *data, void *)
{
    Ctx *x = (Ctx *)data;
    x->closed = true;
    x->collected = true;
    ctx_release(x);
}

napi_value DestroyContext(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *p = nullptr;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) {
        napi_throw_type_error(env, nullptr, "context handle expected");
        return nullptr;
    }
    Ctx *x = (Ctx *)p;
    x->closed = true;
    ctx_release(x);
    napi_value v;
    napi_get_boolean(env, x->c == nullptr && x->g == nullptr, &v);    // true: released now; false: a render is still in flight, released when it ends
    return v;
}

napi_value CreateGroup(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    uint32_t len = 0;
    bool is_array = false;
    if (argc < 1 || napi_is_array(env, argv[0], &is_array) != napi_ok || !is_array || napi_get_array_length(env, argv[0], &len) != napi_ok || len < 1) {
        napi_throw_type_error(env, nullptr, "createGroup: a non-empty array of device indices expected");
        return nullptr;
    }
    std::vector<int32_t> devices(len);
    for (uint32_t i = 0; i < len; i++) {
        napi_value e;
        napi_get_element(env, argv[0], i, &e);
        napi_get_value_int32(env, e, &devices[i]);
    }
    sp_group *g = nullptr;
    const int rc = sp_group_create(devices.data(), (int32_t)len, &g);
    if (rc) return throw_status(env, rc, rc == SP_ERR_NO_DEVICE ? "no HIP device: spectroplot-hip has no CPU fallback" : nullptr);
    Ctx *x = new Ctx;
    x->g = g;
    napi_value ext;
    if (napi_create_external(env, x, ctx_finalize, nullptr, &ext) != napi_ok) {
        sp_group_destroy(g);
        delete x;
        napi_throw_error(env, nullptr, "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

void pinned_free_cb(napi_env env, void *data, void *hint)
{
    int64_t total = 0;
    napi_adjust_external_memory(env, -(int64_t)(size_t)hint, &total);
    sp_host_free(data);
}

napi_value AllocBuffer(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    double nbytes = 0;
    napi_get_value_double(env, argv[0], &nbytes);
    if (!(nbytes >= 0) || nbytes > 4e12) return throw_status(env, SP_ERR_INVALID_ARG, "allocBuffer: bad size");
    void *p = nullptr;
    const int rc = sp_host_alloc((size_t)nbytes, &p);
    if (rc) return throw_status(env, rc, rc == SP_ERR_NO_DEVICE ? "no HIP device: spectroplot-hip has no CPU fallback" : "page-locked allocation failed");
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, (size_t)nbytes, pinned_free_cb, (void *)(size_t)nbytes, &ab) != napi_ok) {
        sp_host_free(p);
        napi_throw_error(env, nullptr, "napi_create_external_arraybuffer failed");
        return nullptr;
    }
    int64_t total = 0;
    napi_adjust_external_memory(env, (int64_t)nbytes, &total);
    return ab;
}

napi_value PoolStats(napi_env env, napi_callback_info)
{
    napi_value out, v;
    napi_create_object(env, &out);
    std::lock_guard<std::mutex> g(g_pool.m);
    napi_create_double(env, (double)g_pool.n_fresh, &v); napi_set_named_property(env, out, "fresh", v);
    napi_create_double(env, (double)g_pool.n_recycled, &v); napi_set_named_property(env, out, "recycled", v);
    napi_create_double(env, (double)g_pool.n_recycled_pinned, &v); napi_set_named_property(env, out, "recycledPinned", v);
    napi_create_double(env, (double)g_pool.free_bytes, &v); napi_set_named_property(env, out, "freeBytes", v);
    napi_create_double(env, (double)g_pool.pinned_bytes, &v); napi_set_named_property(env, out, "pinnedBytes", v);
    napi_create_double(env, (double)g_pool.n_pin_refused, &v); napi_set_named_property(env, out, "pinRefused", v);
    napi_create_double(env, (double)g_pool.keep_bytes, &v); napi_set_named_property(env, out, "keepLimit", v);
    napi_create_double(env, (double)g_pool.pinned_limit, &v); napi_set_named_property(env, out, "pinnedLimit", v);
    napi_create_double(env, (double)g_reply_weight_cap, &v); napi_set_named_property(env, out, "replyWeightCap", v);
    return out;
}

napi_value Cmap(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1], sname;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    NAPI_OK(env, napi_coerce_to_string(env, argv[0], &sname));
    char name[96];
    size_t len = 0;
    napi_get_value_string_utf8(env, sname, name, sizeof name, &len);
    int32_t entries = 0;
    napi_value out;
    if (sp_cmap(name, nullptr, 0, &entries) == SP_ERR_UNSUPPORTED) {
        napi_get_null(env, &out);
        return out;
    }
    napi_value ab;
    void *data;
    NAPI_OK(env, napi_create_arraybuffer(env, 3 * (size_t)entries, &data, &ab));
    sp_cmap(name, (uint8_t *)data, entries, &entries);
    NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, 3 * (size_t)entries, ab, 0, &out));
    return out;
}

napi_value CmapKeys(napi_env env, napi_callback_info)
{
    napi_value out, v;
    const int n = sp_cmap_count();
    napi_create_array_with_length(env, (size_t)n, &out);
    for (int i = 0; i < n; i++) {
        napi_create_string_utf8(env, sp_cmap_key(i), NAPI_AUTO_LENGTH, &v);
        napi_set_element(env, out, (uint32_t)i, v);
    }
    return out;
}

napi_value NamedResolve(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2], out, v, sv;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    std::string names[2];
    for (int k = 0; k < 2; k++) {
        size_t len = 0;
        NAPI_OK(env, napi_coerce_to_string(env, argv[k], &sv));
        napi_get_value_string_utf8(env, sv, nullptr, 0, &len);
        names[k].assign(len, '\0');
        napi_get_value_string_utf8(env, sv, &names[k][0], len + 1, &len);
    }
    const char *wname = "", *ckey = "";
    int32_t lut_len = 0;
    sp_named_resolve(names[0].c_str(), names[1].c_str(), &wname, &ckey, &lut_len);
    NAPI_OK(env, napi_create_object(env, &out));
    napi_create_string_utf8(env, wname, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, out, "window", v);
    napi_create_string_utf8(env, ckey, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, out, "cmap", v);
    napi_create_int32(env, lut_len, &v); napi_set_named_property(env, out, "lutLength", v);
    return out;
}

napi_value PlanCreations(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1], v;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *p = nullptr;
    if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p || !((Ctx *)p)->c) {
        napi_throw_type_error(env, nullptr, "context handle expected");
        return nullptr;
    }
    int64_t n = 0;
    sp_context_plan_creations(((Ctx *)p)->c, &n);
    NAPI_OK(env, napi_create_int64(env, n, &v));
    return v;
}

napi_value CreateContext(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t dev = 0;
    if (argc >= 1) napi_get_value_int32(env, argv[0], &dev);
    sp_context *ctx = nullptr;
    const int rc = sp_context_create(dev, &ctx);
    if (rc) return throw_status(env, rc, rc == SP_ERR_NO_DEVICE ? "no HIP device: spectroplot-hip has no CPU fallback" : nullptr);
    Ctx *x = new Ctx;
    x->c = ctx;
    napi_value ext;
    if (napi_create_external(env, x, ctx_finalize, nullptr, &ext) != napi_ok) {
        sp_context_destroy(ctx);
        delete x;
        napi_throw_error(env, nullptr, "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

napi_value Init(napi_env env, napi_value exports)
{
    const napi_property_descriptor props[] = {
        {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"parseFormat", nullptr, ParseFormat, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"window", nullptr, Window, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"sliceBounds", nullptr, SliceBounds, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"createContext", nullptr, CreateContext, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"destroyContext", nullptr, DestroyContext, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"allocBuffer", nullptr, AllocBuffer, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"poolStats", nullptr, PoolStats, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"cmap", nullptr, Cmap, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"cmapKeys", nullptr, CmapKeys, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"render", nullptr, Render, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"renderSync", nullptr, RenderSync, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"renderNamed", nullptr, RenderNamed, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"renderNamedSync", nullptr, RenderNamedSync, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"namedResolve", nullptr, NamedResolve, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"planCreations", nullptr, PlanCreations, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"createGroup", nullptr, CreateGroup, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"destroyGroup", nullptr, DestroyContext, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupRender", nullptr, Render, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupRenderSync", nullptr, RenderSync, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    napi_value v;
    napi_create_int32(env, sp_version(), &v);
    napi_set_named_property(env, exports, "version", v);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
