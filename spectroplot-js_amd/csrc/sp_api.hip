// sp_api.hip — C ABI of the library (include/spectroplot_hip.h): contexts, plans, the frame-loop launch.
//
// One sp_context mirrors one reference Worker instance (lib/spectroplot.js:85-116): requests run in order on
// its stream.  A plan holds everything the reference computes once per request before its frame loop
// (lib/worker.js:30-62) plus the threshold tables that stand in for the per-pixel Math.log10.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/spectroplot_hip.h"
#include "sp_host.h"
#include "sp_kernel_frames.h"
#include "sp_kernel_scratch.h"
#include "sp_synth.h"
#include "sp_cmap_tables.h"

namespace {

struct DeviceBuffer {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SP_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            if (hipMalloc(&p, bytes) != hipSuccess) return SP_ERR_NOMEM;
            want = bytes;
        }
        cap = want;
        return SP_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// page-locked host memory owned by a context (the landing place of a reply's small outputs)
struct HostBuffer {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SP_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 8 + 256;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return SP_ERR_NOMEM;
        cap = want;
        return SP_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct sp_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;
    int cu_count = 256;
    // workspace of the frame loop
    DeviceBuffer frame_minmax;   // 2 * width doubles (the scratch kernel's frame extremes, read by k_finish_frames)
    DeviceBuffer partial;        // [0,4) the number of the last request k_frames has started; the scratch kernel's {min,max} and histogram accumulators
    DeviceBuffer scratch;        // scratch kernel slabs
    // staging for sp_render (host-buffer entry point)
    DeviceBuffer in_bytes, out_rgba, render_small;
    HostBuffer host_small;
    // sp_render's copy streams and events: the image goes back to the host chunk by chunk while later chunks still arrive
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    static constexpr int kMaxChunks = 16;
    hipEvent_t ev_arrived[kMaxChunks] = {}, ev_rendered[kMaxChunks] = {};
    sp_plan *cached_plan = nullptr;
    // sp_render_named: the names and numbers the cached plan was built from (empty: the cached plan came from arrays)
    std::string named_format, named_window, named_cmap;
    int32_t named_n = 0, named_ch = 0, named_wf = 0;
    double named_gain = 0, named_range = 0;
    std::vector<double> named_windowc;
    std::vector<uint8_t> named_lut;
    double named_block_norm = 0;
    long long plans_created = 0; // sp_context_plan_creations: how many plans (table sets on the device) this context has built
    bool acc_dirty = false;      // a request failed between its launches: accumulators must be re-initialised
    size_t last_upload_bytes = 0; // what the last sp_render sent over the host link (a sparse request sends its frames only)
    uint32_t seq = 0;            // requests started on this context (k_frames publishes the number once the reply is cleared; never 0)
    // timing
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
};

struct sp_plan {
    sp_context *ctx = nullptr;
    sp_request req{};               // windowc / lut_rgb pointers are NOT kept (copied below)
    std::vector<double> window;
    std::vector<uint8_t> lut;
    int levels = 0;
    spfmt::Format fmt{};
    double block_norm_db = 0;
    float gray_a = 0, gray_b = 0, cb_a = 0, cb_b = 0;
    bool edges_in_f32 = false;      // every colour / centi-bel edge lies in [2^-100, 2^100]: the f32 first guess is enough
    bool taper_finite = false;      // no +-inf / NaN in windowc: the (1, 0) butterflies may skip their products on integer samples
    bool tw16_ok = false;           // the first-pass twiddle literals of k_frames equal this plan's table entries
    sphost::Thresholds th{};        // f32 coefficients of both frame-loop kernels (edge vectors are released after the upload)
    DeviceBuffer tables;            // one allocation, carved below
    const double *d_window = nullptr, *d_cos = nullptr, *d_sin = nullptr, *d_gray_edge = nullptr, *d_cb_edge = nullptr;
    const uint32_t *d_lut = nullptr;
    const uint16_t *d_cell_g = nullptr, *d_cell_l = nullptr;   // merged-cell ranges per colour index / level (k_frames)
    const double2 *d_stage_tw = nullptr;   // per-stage twiddle tables for k_frames
    int force_kernel = 0;           // 0 auto, 1 scratch, 3 frames
};

namespace {

int fail(sp_context *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->error = msg;
    return code;
}

int hip_fail(sp_context *ctx, hipError_t e, const char *what)
{
    return fail(ctx, SP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define SP_HIP(ctx, call)                                            \
    do {                                                             \
        hipError_t e_ = (call);                                      \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #call);     \
    } while (0)

template <typename F>
int dispatch_format(int32_t fmt, F &&f)
{
    switch (fmt) {
#define SP_CASE(X) case X: return f(std::integral_constant<int, X>{});
        SP_CASE(SP_FMT_CU4) SP_CASE(SP_FMT_CS4) SP_CASE(SP_FMT_CU8) SP_CASE(SP_FMT_CS8) SP_CASE(SP_FMT_CU12)
        SP_CASE(SP_FMT_CS12) SP_CASE(SP_FMT_CU16) SP_CASE(SP_FMT_CS16) SP_CASE(SP_FMT_CU32) SP_CASE(SP_FMT_CS32)
        SP_CASE(SP_FMT_CU64) SP_CASE(SP_FMT_CS64) SP_CASE(SP_FMT_CF32) SP_CASE(SP_FMT_CF64)
#undef SP_CASE
    default: return SP_ERR_INVALID_ARG;
    }
}

int validate_request(sp_context *ctx, const sp_request *r)
{
    if (!r) return fail(ctx, SP_ERR_INVALID_ARG, "request is null");
    if (r->format < 0 || r->format >= SP_FMT_COUNT) return fail(ctx, SP_ERR_INVALID_ARG, "unknown format id");
    if (r->n < 1 || sphost::log2_exact(r->n) < 0) return fail(ctx, SP_ERR_NOT_POW2, "Length is not a power of 2");
    if (r->n < 2) return fail(ctx, SP_ERR_UNSUPPORTED, "n = 1 is not supported (the reference writes no pixels for it)");
    if (r->n > SP_MAX_N) return fail(ctx, SP_ERR_UNSUPPORTED, "n exceeds SP_MAX_N");
    if (r->lut_len < 1 || r->lut_len > SP_MAX_LUT) return fail(ctx, SP_ERR_UNSUPPORTED, "lut_len must be 1..SP_MAX_LUT");
    if (!r->windowc || !r->lut_rgb) return fail(ctx, SP_ERR_INVALID_ARG, "windowc / lut_rgb is null");
    if (!(r->range > 0) || !std::isfinite(r->range)) return fail(ctx, SP_ERR_UNSUPPORTED, "range must be finite and > 0");
    if (!std::isfinite(r->gain)) return fail(ctx, SP_ERR_UNSUPPORTED, "gain must be finite");
    return SP_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------- host helpers

extern "C" int sp_version(void) { return SP_VERSION; }

extern "C" const char *sp_status_string(int status)
{
    switch (status) {
    case SP_OK: return "ok";
    case SP_ERR_INVALID_ARG: return "invalid argument";
    case SP_ERR_NOT_POW2: return "Length is not a power of 2";
    case SP_ERR_BYTE_LENGTH: return "byte length is not a multiple of the element size";
    case SP_ERR_UNSUPPORTED: return "unsupported by this library";
    case SP_ERR_NO_DEVICE: return "no HIP device";
    case SP_ERR_HIP: return "HIP runtime error";
    case SP_ERR_NOMEM: return "out of device memory";
    default: return "unknown status";
    }
}

extern "C" const char *sp_last_error(const sp_context *ctx) { return ctx ? ctx->error.c_str() : ""; }

extern "C" int sp_format_parse(const char *name, int32_t *format, int32_t *sample_width)
{
    const int32_t id = sphost::parse_format(name ? name : "");
    if (format) *format = id;
    if (sample_width) *sample_width = spfmt::describe(id).width;
    return SP_OK;
}

extern "C" int sp_format_element_size(int32_t format)
{
    if (format < 0 || format >= SP_FMT_COUNT) return SP_ERR_INVALID_ARG;
    return spfmt::describe(format).elem;
}

extern "C" int sp_slice_bounds(size_t nbytes, int32_t sample_width, int32_t index, int32_t count, size_t *begin, size_t *end)
{
    if (sample_width < 1 || count < 1 || index < 0 || index >= count || !begin || !end) return SP_ERR_INVALID_ARG;
    const size_t end_sample = nbytes / (size_t)sample_width;                     // ~~(byteLength / sampleWidth)
    const size_t slice_len = (size_t)sample_width * (end_sample / (size_t)count);  // sampleWidth * ~~(samples / count)
    *begin = slice_len * (size_t)index;
    *end = slice_len * ((size_t)index + 1);
    return SP_OK;
}

extern "C" int sp_window(const char *name, int32_t n, double *window, double *weight)
{
    if (!name || n < 1 || !window || !weight) return SP_ERR_INVALID_ARG;
    return sphost::window(name, n, window, weight) ? SP_OK : SP_ERR_INVALID_ARG;
}

extern "C" int sp_twiddles(int32_t n, double *cos_table, double *sin_table)
{
    if (n < 1 || sphost::log2_exact(n) < 0) return SP_ERR_NOT_POW2;
    if (!cos_table || !sin_table) return SP_ERR_INVALID_ARG;
    sphost::twiddles(n, cos_table, sin_table);
    return SP_OK;
}

extern "C" double sp_js_log10(double x) { return spjs::log10(x); }

extern "C" int sp_cmap_count(void) { return spcmap::kCount; }

extern "C" const char *sp_cmap_key(int32_t index) { return index >= 0 && index < spcmap::kCount ? spcmap::kEntries[index].key : ""; }

static int cmap_index(const char *name)
{
    const char *keys[spcmap::kCount];
    for (int i = 0; i < spcmap::kCount; i++) keys[i] = spcmap::kEntries[i].key;
    return sphost::lookup_key(keys, spcmap::kCount, name);
}

// entry `i` of the colour-map table as r, g, b bytes: the literal tables of the reference as tables, its computed maps
// (lib/soxcmap.js, lib/naivecmap.js) evaluated by their generators
static void cmap_bytes(int i, uint8_t *rgb)
{
    const spcmap::Entry &e = spcmap::kEntries[i];
    if (e.offset >= 0) memcpy(rgb, spcmap::kData + e.offset, 3 * (size_t)e.length);
    else if (!sphost::cmap_generate(e.key, e.length, rgb)) memset(rgb, 0, 3 * (size_t)e.length);   // (every offset -1 entry has a generator: test_abi_cpu)
}

extern "C" int sp_cmap(const char *name, uint8_t *rgb, int32_t capacity_entries, int32_t *lut_len)
{
    const int i = cmap_index(name);
    if (i < 0) return SP_ERR_UNSUPPORTED;
    const spcmap::Entry &e = spcmap::kEntries[i];
    if (lut_len) *lut_len = e.length;
    if (!rgb || capacity_entries < e.length) return SP_ERR_INVALID_ARG;
    cmap_bytes(i, rgb);
    return SP_OK;
}

extern "C" int sp_cmap_generate(const char *key, int32_t stops, uint8_t *rgb)
{
    if (!key || stops < 1 || stops > 65536 || !rgb) return SP_ERR_INVALID_ARG;
    return sphost::cmap_generate(key, stops, rgb) ? SP_OK : SP_ERR_UNSUPPORTED;
}

extern "C" int sp_host_alloc(size_t nbytes, void **ptr)
{
    if (!ptr) return SP_ERR_INVALID_ARG;
    *ptr = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return SP_ERR_NO_DEVICE;
    return hipHostMalloc(ptr, nbytes ? nbytes : 1, hipHostMallocDefault) == hipSuccess ? SP_OK : SP_ERR_NOMEM;
}

extern "C" void sp_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

extern "C" int sp_host_register(void *ptr, size_t nbytes)
{
    if (!ptr || !nbytes) return SP_ERR_INVALID_ARG;
    return hipHostRegister(ptr, nbytes, hipHostRegisterDefault) == hipSuccess ? SP_OK : SP_ERR_HIP;
}

extern "C" void sp_host_unregister(void *ptr)
{
    if (ptr) (void)hipHostUnregister(ptr);
}

// ------------------------------------------------------------------------------------------------- contexts

extern "C" int sp_device_count(int32_t *count)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    if (count) *count = c;
    return c > 0 ? SP_OK : SP_ERR_NO_DEVICE;
}

extern "C" int sp_context_create(int32_t device, sp_context **out)
{
    if (!out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return SP_ERR_NO_DEVICE;
    if (device < 0 || device >= c) return SP_ERR_INVALID_ARG;
    sp_context *ctx = new (std::nothrow) sp_context;
    if (!ctx) return SP_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return SP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    *out = ctx;
    return SP_OK;
}

extern "C" void sp_context_destroy(sp_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->cached_plan) sp_plan_destroy(ctx->cached_plan);
    ctx->frame_minmax.release();
    ctx->partial.release();
    ctx->scratch.release();
    ctx->in_bytes.release();
    ctx->out_rgba.release();
    ctx->render_small.release();
    ctx->host_small.release();
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (int k = 0; k < sp_context::kMaxChunks; k++) {
        if (ctx->ev_arrived[k]) (void)hipEventDestroy(ctx->ev_arrived[k]);
        if (ctx->ev_rendered[k]) (void)hipEventDestroy(ctx->ev_rendered[k]);
    }
    if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
    if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int sp_context_set_stream(sp_context *ctx, void *hip_stream)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SP_OK;
}

extern "C" int sp_context_get_stream(const sp_context *ctx, void **hip_stream)
{
    if (!ctx || !hip_stream) return SP_ERR_INVALID_ARG;
    *hip_stream = ctx->stream == ctx->own_stream ? nullptr : (void *)ctx->stream;
    return SP_OK;
}

extern "C" int sp_context_synchronize(sp_context *ctx)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

extern "C" int sp_context_enable_timing(sp_context *ctx, int32_t on)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    ctx->timing = on != 0;
    ctx->timed = false;
    return SP_OK;
}

extern "C" int sp_context_last_kernel_ms(sp_context *ctx, float *ms)
{
    if (!ctx || !ms) return SP_ERR_INVALID_ARG;
    if (!ctx->timed) return fail(ctx, SP_ERR_INVALID_ARG, "no timed execute on this context");
    SP_HIP(ctx, hipEventSynchronize(ctx->ev1));
    SP_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return SP_OK;
}

__global__ void k_noop() {}

extern "C" int sp_context_event_pair_overhead_ms(sp_context *ctx, float *ms)
{
    if (!ctx || !ms) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    float best = 1e30f;
    for (int i = 0; i < 8; i++) {
        SP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, ctx->stream);
        SP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        SP_HIP(ctx, hipEventSynchronize(ctx->ev1));
        float t = 0;
        SP_HIP(ctx, hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
        if (t < best) best = t;
    }
    ctx->timed = false;
    *ms = best;
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------- device memory

extern "C" int sp_device_alloc(sp_context *ctx, size_t nbytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(d_ptr, nbytes ? nbytes : 1) != hipSuccess) return fail(ctx, SP_ERR_NOMEM, "hipMalloc failed");
    return SP_OK;
}

extern "C" int sp_device_free(sp_context *ctx, void *d_ptr)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SP_HIP(ctx, hipFree(d_ptr));
    return SP_OK;
}

extern "C" int sp_device_upload(sp_context *ctx, void *d_dst, const void *src, size_t nbytes)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipMemcpyAsync(d_dst, src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

extern "C" int sp_device_download(sp_context *ctx, void *dst, const void *d_src, size_t nbytes)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipMemcpyAsync(dst, d_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    SP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SP_OK;
}

extern "C" int sp_device_memset(sp_context *ctx, void *d_ptr, int value, size_t nbytes)
{
    if (!ctx) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    SP_HIP(ctx, hipMemsetAsync(d_ptr, value, nbytes, ctx->stream));
    return SP_OK;
}

extern "C" int sp_synth_trinoise(sp_context *ctx, void *d_bytes, int32_t format, uint64_t t0, uint64_t count, uint32_t seed,
                                 uint32_t step, uint32_t gshift, double amp, double namp)
{
    if (!ctx || !d_bytes) return SP_ERR_INVALID_ARG;
    if (format < 0 || format >= SP_FMT_COUNT) return fail(ctx, SP_ERR_INVALID_ARG, "unknown format id");
    if (count == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const spk::SynthArgs a{(uint8_t *)d_bytes, t0, count, seed, step, gshift, amp, namp};
    const uint64_t blocks = (count + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(ctx, SP_ERR_UNSUPPORTED, "synth: too many samples for one launch");
    const int rc = dispatch_format(format, [&](auto F) {
        hipLaunchKernelGGL(spk::k_synth_trinoise<decltype(F)::value>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        return SP_OK;
    });
    if (rc) return rc;
    SP_HIP(ctx, hipGetLastError());
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------- plans

extern "C" int sp_plan_create(sp_context *ctx, const sp_request *req, sp_plan **out)
{
    if (!ctx || !out) return SP_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = validate_request(ctx, req);
    if (rc) return rc;
    SP_HIP(ctx, hipSetDevice(ctx->device));

    sp_plan *p = new (std::nothrow) sp_plan;
    if (!p) return fail(ctx, SP_ERR_NOMEM, "out of host memory");
    p->ctx = ctx;
    p->req = *req;
    p->window.assign(req->windowc, req->windowc + req->n);
    p->lut.assign(req->lut_rgb, req->lut_rgb + 3 * (size_t)req->lut_len);
    p->req.windowc = nullptr;
    p->req.lut_rgb = nullptr;
    p->levels = sphost::log2_exact(req->n);
    p->fmt = spfmt::describe(req->format);

    const int n = req->n, half = n / 2, L = req->lut_len;
    std::vector<double> ct((size_t)half), st((size_t)half);
    sphost::twiddles(n, ct.data(), st.data());
    const sphost::PixelMath pm(req->block_norm, req->gain, req->range, L);
    const sphost::Thresholds th = sphost::build_thresholds(pm, L);
    p->block_norm_db = pm.block_norm_db;
    p->gray_a = th.gray_a;
    p->gray_b = th.gray_b;
    p->cb_a = th.cb_a;
    p->cb_b = th.cb_b;
    p->edges_in_f32 = true;
    for (int g = 1; g < L; g++) p->edges_in_f32 &= th.gray_edge[(size_t)g] >= 0x1p-100 && th.gray_edge[(size_t)g] <= 0x1p100;
    for (int j = 1; j <= SP_CB_HIST_SIZE; j++) p->edges_in_f32 &= th.cb_edge[(size_t)j] >= 0x1p-100 && th.cb_edge[(size_t)j] <= 0x1p100;
    p->taper_finite = true;
    for (int i = 0; i < n; i++) p->taper_finite &= std::isfinite(p->window[(size_t)i]);
    p->tw16_ok = n >= 16;
    for (int k = 0; k < 8 && p->tw16_ok; k++)
        p->tw16_ok = ct[(size_t)k * (size_t)(n / 16)] == spk2::kTw16Host[k].c && st[(size_t)k * (size_t)(n / 16)] == spk2::kTw16Host[k].s;
    p->th = th;
    p->th.gray_edge.clear();
    p->th.cb_edge.clear();
    p->th.cell_g.clear();
    p->th.cell_l.clear();
    std::vector<uint32_t> lut32((size_t)L);
    for (int i = 0; i < L; i++)
        lut32[(size_t)i] = (uint32_t)p->lut[3 * (size_t)i] | ((uint32_t)p->lut[3 * (size_t)i + 1] << 8)
                           | ((uint32_t)p->lut[3 * (size_t)i + 2] << 16) | 0xff000000u;
    // per-stage twiddle tables for k_frames: stage s (size 2^s) has 2^(s-1) entries (cos, sin), consecutive
    std::vector<double2> stage_tw;
    if (spk::frame_parts_support(n)) {
        stage_tw.resize((size_t)n);   // entries 1 .. n-1 used: stage s starts at 2^(s-1)
        stage_tw[0] = make_double2(0, 0);
        for (int s = 1; s <= p->levels; s++) {
            const int cnt = 1 << (s - 1);
            for (int m = 0; m < cnt; m++) {
                const int k = m << (p->levels - s);
                stage_tw[(size_t)(cnt + m)] = make_double2(ct[(size_t)k], st[(size_t)k]);
            }
        }
    }

    // one device allocation: [window n][cos half][sin half][gray_edge L][cb_edge 1001][stage_tw 2n doubles][lut L u32]
    const size_t nd = (size_t)n + 2 * (size_t)half + (size_t)L + (SP_CB_HIST_SIZE + 1) + 2 * stage_tw.size();
    const size_t cell_u16 = th.cell_g.size() + th.cell_l.size();
    const size_t bytes = nd * sizeof(double) + (((size_t)L * sizeof(uint32_t) + 7) & ~(size_t)7) + cell_u16 * sizeof(uint16_t);
    rc = p->tables.reserve(bytes);
    if (rc) {
        delete p;
        return fail(ctx, rc, "plan tables: out of device memory");
    }
    std::vector<uint8_t> host(bytes);
    double *h = (double *)host.data();
    double *d = (double *)p->tables.p;
    size_t o = 0;
    auto put = [&](const void *src, size_t count) {
        if (count) memcpy(h + o, src, count * sizeof(double));
        const double *dev = d + o;
        o += count;
        return dev;
    };
    p->d_window = put(p->window.data(), (size_t)n);
    p->d_cos = put(ct.data(), (size_t)half);
    p->d_sin = put(st.data(), (size_t)half);
    p->d_gray_edge = put(th.gray_edge.data(), (size_t)L);
    p->d_cb_edge = put(th.cb_edge.data(), SP_CB_HIST_SIZE + 1);
    p->d_stage_tw = (const double2 *)put(stage_tw.data(), 2 * stage_tw.size());
    memcpy(h + o, lut32.data(), (size_t)L * sizeof(uint32_t));
    p->d_lut = (const uint32_t *)(d + o);
    {
        const size_t off = o * sizeof(double) + (((size_t)L * sizeof(uint32_t) + 7) & ~(size_t)7);
        memcpy(host.data() + off, th.cell_g.data(), th.cell_g.size() * sizeof(uint16_t));
        memcpy(host.data() + off + th.cell_g.size() * sizeof(uint16_t), th.cell_l.data(), th.cell_l.size() * sizeof(uint16_t));
        p->d_cell_g = (const uint16_t *)((const char *)p->tables.p + off);
        p->d_cell_l = p->d_cell_g + th.cell_g.size();
    }
    hipError_t e = hipMemcpyAsync(p->tables.p, host.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        p->tables.release();
        delete p;
        return hip_fail(ctx, e, "plan table upload");
    }
    ctx->plans_created++;
    *out = p;
    return SP_OK;
}

extern "C" int sp_context_plan_creations(const sp_context *ctx, int64_t *count)
{
    if (!ctx || !count) return SP_ERR_INVALID_ARG;
    *count = (int64_t)ctx->plans_created;
    return SP_OK;
}

extern "C" void sp_plan_destroy(sp_plan *plan)
{
    if (!plan) return;
    if (plan->ctx) {
        (void)hipSetDevice(plan->ctx->device);
        (void)hipStreamSynchronize(plan->ctx->stream);
        if (plan->ctx->cached_plan == plan) plan->ctx->cached_plan = nullptr;
    }
    plan->tables.release();
    delete plan;
}

// k_frames skips the products of (1, 0) butterflies on integer samples, which is exact only with a finite taper; it takes its
// indices from an f32 scale, which needs every edge inside the f32 range and usable margins (sp_host.cpp); its first-pass twiddles
// are literals, checked against this plan's table.
static bool plan_frames_capable(const sp_plan *plan)
{
    return spk2::frames_kernel_supports(plan->req.n) && plan->req.lut_len >= 2 && plan->req.lut_len <= spk::kLdsMaxLut
           && plan->gray_b <= spk::kLdsMaxGrayB && plan->edges_in_f32 && plan->taper_finite && plan->th.frames_ok && plan->tw16_ok;
}

// 1 = scratch_radix2, 3 = frames (2 was k_lds_r16, round 1's frame loop: removed in round 3)
static int plan_kernel(const sp_plan *plan)
{
    if (plan->force_kernel) return plan->force_kernel;
#ifdef SP_EXPERIMENT_KNOBS
    static const int env_kernel = getenv("SP_FORCE_KERNEL") ? atoi(getenv("SP_FORCE_KERNEL")) : 0;
    if (env_kernel == 3 && plan_frames_capable(plan)) return 3;
    if (env_kernel == 1) return 1;
#endif
    // k_frames is the fast path; the scratch kernel covers every request it does not
    if (plan_frames_capable(plan)) return 3;
    return 1;
}

extern "C" int sp_plan_force_kernel(sp_plan *plan, int32_t which)
{
    if (!plan || which < 0 || which > 3) return SP_ERR_INVALID_ARG;
    if (which == 2) return fail(plan->ctx, SP_ERR_UNSUPPORTED, "k_lds_r16 is no longer part of the library");
    if (which == 3 && !plan_frames_capable(plan)) return fail(plan->ctx, SP_ERR_UNSUPPORTED, "k_frames does not cover this request");
    plan->force_kernel = which;
    return SP_OK;
}

extern "C" const char *sp_plan_kernel_name(const sp_plan *plan)
{
    if (!plan) return "";
    switch (plan_kernel(plan)) {
    case 3: return "frames";
    default: return "scratch_radix2";
    }
}

// The frame loop over frames [x_begin, x_end) of a width-frame image.  `first` prepares the context's workspace and accumulators,
// `last` ends the request: k_frames then produces histograms and dBfs range itself (its last workgroup; the gauges it writes group by
// group in every launch), behind the scratch kernel a finish kernel is queued.  sp_plan_execute is the whole range in one launch - ONE
// kernel for every request k_frames covers; sp_render walks the image in chunks so that the copies to and from the host overlap.
// `src` (sp_render's packed upload, below): the frames [x_begin, x_end) do not lie in the capture at d_bytes but in a packed copy of it;
// the kernel is handed that copy's address, length and stride instead (every frame inside it), everything else - image geometry, frame
// numbers, reply - stays the request's.
struct PackedSource {
    const void *bytes;     // address of (virtual) sample 0 of the packed layout
    size_t nbytes;         // its (virtual) length
    double stride;         // frame x starts at sample ~~(0.5 + stride * x) of it
};

static int plan_execute_range(sp_plan *plan, const void *d_bytes, size_t nbytes, int32_t width, int32_t x_begin, int32_t x_end, bool first,
                              bool last, const sp_reply *out, const PackedSource *src = nullptr)
{
    if (!plan || !out) return SP_ERR_INVALID_ARG;
    sp_context *ctx = plan->ctx;
    if (width < 0) return fail(ctx, SP_ERR_INVALID_ARG, "width < 0");
    if (nbytes && !d_bytes) return fail(ctx, SP_ERR_INVALID_ARG, "d_bytes is null");
    const spfmt::Format f = plan->fmt;
    if (nbytes % (size_t)f.elem) return fail(ctx, SP_ERR_BYTE_LENGTH, "byte length is not a multiple of the element size");
    const int n = plan->req.n;
    const double sample_count = (double)nbytes / (double)f.width;                 // samples.js:167
    if (sample_count >= 2147483648.0 - (double)n)
        return fail(ctx, SP_ERR_UNSUPPORTED, "captures of 2^31 samples or more must be sliced (sample positions are int32)");
    if ((double)width * (double)n > 4e12) return fail(ctx, SP_ERR_UNSUPPORTED, "image too large");
    // (the kernels add to these with 64-bit device atomics)
    if ((((uintptr_t)out->c_hist | (uintptr_t)out->cb_hist | (uintptr_t)out->dbfs_minmax) & 7) != 0)
        return fail(ctx, SP_ERR_INVALID_ARG, "reply: c_hist, cb_hist and dbfs_minmax must be 8-byte aligned");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    {
        // The request's number travels in the kernel arguments, and the frame loop's workgroups wait for workgroup 0 to publish it (the
        // reply is cleared first): a captured launch replayed from a hipGraph would find the number already there.  Refused.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(ctx, SP_ERR_UNSUPPORTED, "sp_plan_execute cannot be captured into a hipGraph (every launch carries its request's number)");
        (void)hipGetLastError();
    }

    if (width == 0) {
        // nothing to draw; the reply keeps the loop's initial values (worker.js:35-36)
        if (out->c_hist) SP_HIP(ctx, hipMemsetAsync(out->c_hist, 0, (size_t)plan->req.lut_len * sizeof(uint64_t), s));
        if (out->cb_hist) SP_HIP(ctx, hipMemsetAsync(out->cb_hist, 0, SP_CB_HIST_SIZE * sizeof(uint64_t), s));
        if (out->dbfs_minmax) {
            static const double init[2] = {0.0, -200.0};
            SP_HIP(ctx, hipMemcpyAsync(out->dbfs_minmax, init, sizeof init, hipMemcpyHostToDevice, s));
            SP_HIP(ctx, hipStreamSynchronize(s));
        }
        return SP_OK;
    }

    double stride = (sample_count - (double)n) / (double)(width - 1);              // worker.js:50
    // do all frames lie inside the buffer?
    bool in_bounds = false;
    if (src) {
        d_bytes = src->bytes;
        nbytes = src->nbytes;
        stride = src->stride;
        in_bounds = true;
    } else {
        const double last_d = 0.5 + stride * (double)(width - 1);
        if (width == 1) {
            in_bounds = (size_t)n * (size_t)f.width <= nbytes;
        } else if (stride >= 0.0 && std::isfinite(stride) && last_d < 2147483647.0) {
            const int64_t last = spjs::to_int32(last_d);
            in_bounds = (size_t)(last + n) * (size_t)f.width <= nbytes;
        }
    }

    const int which = plan_kernel(plan);
    if (src && which != 3) return fail(ctx, SP_ERR_INVALID_ARG, "a packed source is for the frame-loop kernel only");
    int rc = which == 3 ? SP_OK : ctx->frame_minmax.reserve(2 * (size_t)width * sizeof(double));
    if (rc) return fail(ctx, rc, "workspace: out of device memory");
    int finish_blocks = 3 * ((width + spk::kFinishThreads - 1) / spk::kFinishThreads);   // three roles per 256 frames
    {
        const int bins = plan->req.lut_len > SP_CB_HIST_SIZE ? plan->req.lut_len : SP_CB_HIST_SIZE;   // it also moves the histograms
        const int hb = (bins + spk::kFinishThreads - 1) / spk::kFinishThreads;
        if (finish_blocks < hb) finish_blocks = hb;
        finish_blocks += 1;          // the dBfs range: a workgroup of its own, behind neither the histograms nor a gauge
    }
    // [0,4) k_frames: the number of the last request whose reply workgroup 0 has cleared; scratch kernel: [16,32) bit patterns of the
    // extreme |X|^2 of a request, [64, ...) colour and centi-bel histogram accumulators, back at their initial values when a request ends
    const size_t acc_bytes = 64 + (SP_MAX_LUT + SP_CB_HIST_SIZE) * sizeof(unsigned long long);
    const bool fresh_partial = ctx->partial.cap < acc_bytes;
    rc = ctx->partial.reserve(acc_bytes);
    if (rc) return fail(ctx, rc, "workspace: out of device memory");
    if (first && (fresh_partial || ctx->acc_dirty)) {
        static const unsigned long long mm_init[2] = {0x7ff0000000000000ull, 0ull};           // +inf, 0
        SP_HIP(ctx, hipMemsetAsync(ctx->partial.p, 0, ctx->partial.cap, s));
        SP_HIP(ctx, hipMemcpyAsync((char *)ctx->partial.p + 16, mm_init, sizeof mm_init, hipMemcpyHostToDevice, s));
        SP_HIP(ctx, hipStreamSynchronize(s));
    }
    ctx->acc_dirty = true;   // until this request's last launch (k_frames) or its finish kernel (scratch path) has been queued

    spk::FrameArgs a{};
    a.bytes = (const uint8_t *)d_bytes;
    a.nbytes = (int64_t)nbytes;
    a.nelem = (int64_t)(nbytes / (size_t)f.elem);
    a.stride = width > 1 ? stride : 0.0;   // one frame: (S - n) / 0 is an infinity or a NaN and ~~(0.5 + it * 0) = 0, as with 0
    a.n = n;
    a.levels = plan->levels;
    a.width = width;
    a.channel_mode = plan->req.channel_mode ? 1 : 0;
    a.waterfall = plan->req.waterfall ? 1 : 0;
    a.lut_len = plan->req.lut_len;
    a.in_bounds = in_bounds ? 1 : 0;
    a.frame0 = x_begin;
    a.x_end = x_end;
    a.sample_width = f.width;
    a.window = plan->d_window;
    a.cos_t = plan->d_cos;
    a.sin_t = plan->d_sin;
    a.gray_edge = plan->d_gray_edge;
    a.cb_edge = plan->d_cb_edge;
    a.lut_rgba = plan->d_lut;
    a.gray_a = plan->gray_a;
    a.gray_b = plan->gray_b;
    a.cb_a = plan->cb_a;
    a.cb_b = plan->cb_b;
    a.g2_a = plan->th.g2_a;
    a.g2_b = plan->th.g2_b;
    a.g2_thr = plan->th.g2_thr;
    a.g2_m = plan->th.g2_m;
    a.c2_a = plan->th.c2_a;
    a.c2_b = plan->th.c2_b;
    a.c2_thr = plan->th.c2_thr;
    a.c2_m = plan->th.c2_m;
    a.c2_lo = plan->th.c2_lo;
    a.c2_hi = plan->th.c2_hi;
    a.rgba = out->rgba;
    a.frame_min = (double *)ctx->frame_minmax.p;
    a.frame_max = a.frame_min + width;
    a.scratch = nullptr;
    if (first && ++ctx->seq == 0) ctx->seq = 1;
    a.first = first ? 1 : 0;
    a.seq = ctx->seq;
    a.flag = (unsigned int *)ctx->partial.p;
    a.gauge_mins = out->gauge_mins;
    a.gauge_maxs = out->gauge_maxs;
    a.gauge_amps = out->gauge_amps;
    a.block_norm_db = plan->block_norm_db;
    a.gain = plan->req.gain;
    a.range = plan->req.range;
    a.out_c = (unsigned long long *)out->c_hist;
    a.out_cb = (unsigned long long *)out->cb_hist;
    a.out_minmax = out->dbfs_minmax;
    a.cell_g = plan->d_cell_g;
    a.cell_l = plan->d_cell_l;

    // the kernels count into the context's accumulators; the finish kernel moves the counts to the reply
    a.mm_acc = (unsigned long long *)((char *)ctx->partial.p + 16);
    a.c_hist = (unsigned long long *)((char *)ctx->partial.p + 64);
    a.cb_hist = a.c_hist + SP_MAX_LUT;
    a.cells = plan->th.cells;
    a.rgba_fast = out->rgba && ((uintptr_t)out->rgba & 15) == 0 && (width & 3) == 0 && width < (1 << 24)
                  && (double)width * (double)n * 4.0 <= 4294967296.0;

    if (ctx->timing) SP_HIP(ctx, hipEventRecord(ctx->ev0, s));
    if (which == 3) {
        rc = spk2::launch_frames(a, plan->req.format, plan->d_stage_tw, ctx->cu_count, ctx->device, s);
        if (rc) return fail(ctx, rc, "k_frames launch rejected the configuration");
    } else {
        // scratch slabs: one per workgroup, capped at 256 MiB
        long long blocks = (256ll << 20) / (16ll * n);
        if (blocks > x_end - x_begin) blocks = x_end - x_begin;
        if (blocks > 4 * ctx->cu_count) blocks = 4 * ctx->cu_count;
        if (blocks < 1) blocks = 1;
        rc = ctx->scratch.reserve((size_t)blocks * 2 * (size_t)n * sizeof(double));
        if (rc) return fail(ctx, rc, "scratch: out of device memory");
        a.scratch = (double *)ctx->scratch.p;
        rc = dispatch_format(plan->req.format, [&](auto F) {
            if (n >= 4096) hipLaunchKernelGGL((spk::k_scratch_radix2<decltype(F)::value, true>), dim3((unsigned)blocks), dim3(spk::kScratchThreads), 0, s, a);
            else hipLaunchKernelGGL((spk::k_scratch_radix2<decltype(F)::value, false>), dim3((unsigned)blocks), dim3(spk::kScratchThreads), 0, s, a);
            return SP_OK;
        });
        if (rc) return fail(ctx, rc, "bad format");
    }
    SP_HIP(ctx, hipGetLastError());
    if (ctx->timing) {
        SP_HIP(ctx, hipEventRecord(ctx->ev1, s));
        ctx->timed = true;
    }

    if (!last) return SP_OK;
    if (which == 3) {   // k_frames has finished the request itself
        ctx->acc_dirty = false;
        return SP_OK;
    }

    spk::FinishArgs fa{};
    fa.bytes = a.bytes;
    fa.nbytes = a.nbytes;
    fa.nelem = a.nelem;
    fa.stride = stride;
    fa.n = n;
    fa.width = width;
    fa.format = plan->req.format;
    fa.block_norm_db = plan->block_norm_db;
    fa.gain = plan->req.gain;
    fa.range = plan->req.range;
    fa.frame_min = a.frame_min;
    fa.frame_max = a.frame_max;
    fa.gauge_mins = out->gauge_mins;
    fa.gauge_maxs = out->gauge_maxs;
    fa.gauge_amps = out->gauge_amps;
    fa.mm_acc = a.mm_acc;
    fa.out_minmax = out->dbfs_minmax;
    fa.lut_len = plan->req.lut_len;
    fa.acc_c = a.c_hist;
    fa.acc_cb = a.cb_hist;
    fa.out_c = (unsigned long long *)out->c_hist;
    fa.out_cb = (unsigned long long *)out->cb_hist;
    hipLaunchKernelGGL(spk::k_finish_frames, dim3((unsigned)finish_blocks), dim3(spk::kFinishThreads), 0, s, fa);
    SP_HIP(ctx, hipGetLastError());
    ctx->acc_dirty = false;
    return SP_OK;
}

extern "C" int sp_plan_execute(sp_plan *plan, const void *d_bytes, size_t nbytes, int32_t width, const sp_reply *out)
{
    return plan_execute_range(plan, d_bytes, nbytes, width, 0, width < 0 ? 0 : width, true, true, out);
}

// ------------------------------------------------------------------------------------------------- merge of slice replies

// `rank_stride`: 64-bit words between two ranks' records of the same render; blockIdx.y: the render of a batch (its records start
// blockIdx.y * record_len words into every rank's block, its merged record blockIdx.y * record_len words into the outputs)
__global__ void k_merge_replies(const unsigned long long *rec, int count, int lut_len, size_t rank_stride, unsigned long long *c_hist,
                                unsigned long long *cb_hist, double *minmax)
{
    const int stride = lut_len + SP_CB_HIST_SIZE + 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t shift = (size_t)blockIdx.y * (size_t)stride;
    rec += shift;
    if (i < lut_len + SP_CB_HIST_SIZE) {
        unsigned long long s = 0;
        for (int r = 0; r < count; r++) s += rec[(size_t)r * rank_stride + i];                      // spectroplot.js:1232-1238
        if (i < lut_len) {
            if (c_hist) c_hist[shift + i] = s;
        } else if (cb_hist) {
            cb_hist[shift + i - lut_len] = s;
        }
    } else if (i < stride && minmax) {
        const int k = i - (lut_len + SP_CB_HIST_SIZE);                                              // 0: min, 1: max
        double v = __longlong_as_double((long long)rec[i]);
        for (int r = 1; r < count; r++) {
            const double w = __longlong_as_double((long long)rec[(size_t)r * rank_stride + i]);
            v = k == 0 ? (w < v ? w : v) : (w > v ? w : v);                                         // the `<` / `>` updates of :1230-1231
        }
        minmax[shift + k] = v;
    }
}

extern "C" int sp_merge_replies(sp_context *ctx, const void *d_records, int32_t count, int32_t lut_len, uint64_t *d_c_hist,
                                uint64_t *d_cb_hist, double *d_dbfs_minmax)
{
    if (!ctx || !d_records || count < 1 || lut_len < 1 || lut_len > SP_MAX_LUT) return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int total = lut_len + SP_CB_HIST_SIZE + 2;
    hipLaunchKernelGGL(k_merge_replies, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)d_records, (int)count, (int)lut_len, (size_t)total, (unsigned long long *)d_c_hist,
                       (unsigned long long *)d_cb_hist, d_dbfs_minmax);
    SP_HIP(ctx, hipGetLastError());
    return SP_OK;
}

extern "C" int sp_merge_replies_batch(sp_context *ctx, const void *d_gathered, int32_t ranks, int32_t renders, int32_t lut_len, void *d_merged)
{
    if (!ctx || !d_gathered || !d_merged || ranks < 1 || renders < 1 || renders > 65535 || lut_len < 1 || lut_len > SP_MAX_LUT)
        return SP_ERR_INVALID_ARG;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const int total = lut_len + SP_CB_HIST_SIZE + 2;
    unsigned long long *const out = (unsigned long long *)d_merged;
    // the merged records keep the record layout [c_hist | cB_hist | min, max]: the three outputs are one block, `total` words per render
    hipLaunchKernelGGL(k_merge_replies, dim3((unsigned)((total + 255) / 256), (unsigned)renders), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)d_gathered, (int)ranks, (int)lut_len, (size_t)renders * (size_t)total, out, out + lut_len,
                       (double *)(out + lut_len + SP_CB_HIST_SIZE));
    SP_HIP(ctx, hipGetLastError());
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------- strip placement

// putImageData(strip, offset, 0) for every gathered strip at once (lib/spectroplot.js:1241-1244): strip r of `count`, laid end to end in
// `strips` as an all-gather / gather delivers them, goes to columns [r * slice_width, (r + 1) * slice_width) of the n x width image
// (spectrogram), or to rows [width - slice_width - r * slice_width, ...) of the width x n image (waterfall: row bands in reverse order).
// One thread moves one V (16 bytes = 4 pixels where the widths allow, else one pixel); reads and writes are whole row pieces.
template <typename V>
__global__ void k_place_strips(uint8_t *__restrict__ image, const uint8_t *__restrict__ strips, int count, int n, int width, int slice_width,
                               int waterfall)
{
    constexpr int PX = (int)sizeof(V) / 4;
    const size_t units_per_strip = (size_t)slice_width * (size_t)n / PX;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= units_per_strip * (size_t)count) return;
    const int r = (int)(i / units_per_strip);
    const size_t u = i % units_per_strip;
    size_t dst;
    if (waterfall) {
        dst = ((size_t)(width - slice_width - r * slice_width) * (size_t)n) / PX + u;        // one contiguous band of rows
    } else {
        const size_t row_units = (size_t)slice_width / PX;
        const size_t y = u / row_units, x = u % row_units;
        dst = (y * (size_t)width + (size_t)r * (size_t)slice_width) / PX + x;
    }
    ((V *)image)[dst] = ((const V *)strips)[i];
}

extern "C" int sp_place_strips(sp_context *ctx, uint8_t *d_image, const uint8_t *d_strips, int32_t count, int32_t n, int32_t width,
                               int32_t slice_width, int32_t waterfall)
{
    if (!ctx || !d_image || !d_strips || count < 1 || n < 1 || width < 1 || slice_width < 0) return SP_ERR_INVALID_ARG;
    if ((int64_t)count * slice_width > width) return fail(ctx, SP_ERR_INVALID_ARG, "strips do not fit the image");
    if (slice_width == 0) return SP_OK;
    SP_HIP(ctx, hipSetDevice(ctx->device));
    const bool wide = (slice_width & 3) == 0 && (width & 3) == 0 && (((uintptr_t)d_image | (uintptr_t)d_strips) & 15) == 0
                      && (!waterfall || (n & 3) == 0);
    const size_t units = (size_t)slice_width * (size_t)n * (size_t)count / (wide ? 4 : 1);
    const size_t blocks = (units + 255) / 256;
    if (blocks > 0x7fffffffull) return fail(ctx, SP_ERR_UNSUPPORTED, "image too large for one placement launch");
    if (wide) hipLaunchKernelGGL(k_place_strips<uint4>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_image, d_strips, count, n, width, slice_width, waterfall);
    else hipLaunchKernelGGL(k_place_strips<uint32_t>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_image, d_strips, count, n, width, slice_width, waterfall);
    SP_HIP(ctx, hipGetLastError());
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------- host-buffer render

static bool same_request(const sp_plan *p, const sp_request *r)
{
    const sp_request &q = p->req;
    if (q.format != r->format || q.n != r->n || q.channel_mode != r->channel_mode || q.waterfall != r->waterfall
        || q.lut_len != r->lut_len)
        return false;
    if (memcmp(&q.block_norm, &r->block_norm, 8) || memcmp(&q.gain, &r->gain, 8) || memcmp(&q.range, &r->range, 8)) return false;
    if (memcmp(p->window.data(), r->windowc, sizeof(double) * (size_t)r->n)) return false;
    return memcmp(p->lut.data(), r->lut_rgb, 3 * (size_t)r->lut_len) == 0;
}

// ---- sparse requests: upload only what the frames read --------------------------------------------------------------------------------
// With stride > n the reference's loop touches n samples per frame and skips the rest (lib/worker.js:50, 70-75) - its interactive shape:
// a long capture at a screen-wide `width`.  Copying the capture contiguously moves stride / n times the bytes any frame reads.  Instead
// a chunk's frames travel as the rows of pitched copies (hipMemcpy2DAsync: source rows floor(stride) samples apart) into a packed device
// buffer whose rows are P samples apart, and the kernel is launched on that buffer with the stride P + frac(stride): frame x then
// starts at ~~(0.5 + (P + frac) x) = P x + floor(0.5 + frac x), which is where the pitched copy put it, because the capture has it at
// floor(stride) x + floor(0.5 + frac x).  The identity holds in exact arithmetic; the two sides round differently in f64, so the host
// evaluates both for EVERY frame and takes the contiguous path if a single one disagrees.  The start's fractional drift within a chunk
// (d_j = start_j - start_0 - j floor(stride), 0 <= d_j <= j) is what a pitched copy cannot follow row by row: it covers a run of rows
// whose drifts differ by at most `span` samples and brings that many samples more per row (span: a few hundred samples, at most n/2,
// chosen below to balance the cost of a copy call against the extra bytes); P = n + the widest run's range.
// A frame's centre sample (gauge_amps) lies inside the frame, and nothing else of the path depends on where a frame came from.
struct PackedBlock {
    int32_t j0, j1;        // rows of the chunk (frame x0 + j)
    int32_t dmin, dmax;    // their drifts lie in [dmin, dmax]
};
struct PackedChunk {
    int32_t x0 = 0, x1 = 0;
    int64_t first = 0;     // the capture's sample where frame x0 starts
    int64_t F = 0;         // samples between the capture's rows: floor(stride)
    int64_t P = 0;         // samples between the device rows
    size_t dev_off = 0;    // the chunk's byte offset in the staging buffer
    double stride2 = 0;    // P + frac(stride): the kernel's stride
    int64_t pos2_x0 = 0;   // ~~(0.5 + stride2 * x0): the kernel's start of frame x0
    int64_t pos2_last = 0; // ... and of frame x1 - 1
    std::vector<PackedBlock> blocks;
    std::vector<int32_t> drift;   // d_j per row
};

// Cuts [0, width) at `bounds` and lays every chunk out; false = this request is not worth packing or cannot be (then nothing is used).
static bool build_packed_chunks(int n, int sample_width, size_t nbytes, int32_t width, double stride, const std::vector<int32_t> &bounds,
                                std::vector<PackedChunk> &out, size_t *dev_bytes, size_t *link_bytes)
{
    out.clear();
    if (width < 2 || !(stride > (double)n) || !std::isfinite(stride) || !(0.5 + stride * (double)(width - 1) < 2147483000.0)) return false;
    const int64_t F = (int64_t)std::floor(stride);
    const double frac = stride - (double)F;
    size_t off = 256, moved = 0;
    for (size_t c = 0; c + 1 < bounds.size(); c++) {
        PackedChunk ch;
        ch.x0 = bounds[c];
        ch.x1 = bounds[c + 1];
        if (ch.x1 <= ch.x0) continue;
        ch.first = spjs::to_int32(0.5 + stride * (double)ch.x0);                       // worker.js:72
        ch.F = F;
        const int rows = ch.x1 - ch.x0;
        ch.drift.resize((size_t)rows);
        for (int j = 0; j < rows; j++) {
            const int64_t d = (int64_t)spjs::to_int32(0.5 + stride * (double)(ch.x0 + j)) - ch.first - (int64_t)j * F;
            if (d < 0 || d > (int64_t)rows) return false;       // (0 <= d_j <= j in exact arithmetic)
            ch.drift[(size_t)j] = (int32_t)d;
        }
        // the frame must also END inside the capture (the caller established in_bounds for the request as a whole)
        if ((size_t)(ch.first + (int64_t)(rows - 1) * F + ch.drift[(size_t)rows - 1] + n) * (size_t)sample_width > nbytes) return false;
        // runs of rows whose drifts stay within `span` samples of each other: one pitched copy each, its rows widened by the run's drift
        // range.  A copy call costs the link ~11 us (tools/pcie_probe.hip: 17 MiB in 16 pitched copies 0.49 ms, in one 0.32 ms), a
        // widened row span / 2 samples on average: rows * frac / span calls against rows * span / 2 samples at ~55 GB/s balance at
        // span = sqrt(2 * 11 us * frac * 55 GB/s / bytes per sample) - 275 samples for cf32 at frac = 0.5 - kept within [16, n/2].
        int32_t span = (int32_t)std::sqrt(2.0 * 11e-6 * (frac > 1e-3 ? frac : 1e-3) * 55e9 / (double)sample_width);
        span = span > n / 2 ? n / 2 : span;
        span = span < 16 ? 16 : span;
        int32_t widest = 0;
        for (int j = 0; j < rows;) {
            PackedBlock b{j, j + 1, ch.drift[(size_t)j], ch.drift[(size_t)j]};
            while (b.j1 < rows) {
                const int32_t d = ch.drift[(size_t)b.j1];
                const int32_t lo = d < b.dmin ? d : b.dmin, hi = d > b.dmax ? d : b.dmax;
                if (hi - lo > span) break;
                b.dmin = lo;
                b.dmax = hi;
                b.j1++;
            }
            moved += (size_t)(b.j1 - b.j0) * (size_t)(n + b.dmax - b.dmin) * (size_t)sample_width;
            if (b.dmax - b.dmin > widest) widest = b.dmax - b.dmin;
            ch.blocks.push_back(b);
            j = b.j1;
        }
        // device rows P apart: wide enough that a widened row ends where the next one begins (row j of a run lands at j P + dmin and is
        // n + dmax - dmin long); the frames themselves sit at j P + d_j, their drift accumulating as it does in the capture
        ch.P = (int64_t)n + widest;
        ch.stride2 = (double)ch.P + frac;
        if (!((double)(ch.P + 1) * (double)width < 2147483000.0)) return false;       // the kernel's positions are int32
        ch.pos2_x0 = spjs::to_int32(0.5 + ch.stride2 * (double)ch.x0);
        for (int j = 0; j < rows; j++) {
            const int64_t pos2 = spjs::to_int32(0.5 + ch.stride2 * (double)(ch.x0 + j));
            if (pos2 - ch.pos2_x0 != (int64_t)j * ch.P + ch.drift[(size_t)j]) return false;   // the two sides of the identity rounded apart
            if (j == rows - 1) ch.pos2_last = pos2;
        }
        ch.dev_off = off;
        off += ((size_t)((int64_t)rows * ch.P + ch.drift[(size_t)rows - 1] + widest) * (size_t)sample_width + 255) & ~(size_t)255;
        out.push_back(std::move(ch));
    }
    *dev_bytes = off + 256;
    *link_bytes = moved;
    // worth it only if clearly fewer bytes cross the link, and not in a hail of small copies
    size_t copies = 0;
    for (const PackedChunk &ch : out) copies += ch.blocks.size();
    return !out.empty() && moved <= nbytes / 4 * 3 && copies <= 512;
}

// The pitched copies of one chunk, on `stream`.
static hipError_t upload_packed_chunk(const PackedChunk &ch, int n, int sample_width, const uint8_t *bytes, size_t nbytes, uint8_t *stage,
                                      hipStream_t stream)
{
    const size_t sw = (size_t)sample_width;
    hipError_t e = hipSuccess;
    for (const PackedBlock &b : ch.blocks) {
        int32_t j1 = b.j1;
        // the widened rows may reach past the capture's end in the request's very last rows: those travel one by one, exactly
        while (j1 > b.j0 && (size_t)(ch.first + (int64_t)(j1 - 1) * ch.F + b.dmax + n) * sw > nbytes) j1--;
        if (j1 > b.j0) {
            const size_t row_bytes = (size_t)(n + b.dmax - b.dmin) * sw;
            const uint8_t *src = bytes + (size_t)(ch.first + (int64_t)b.j0 * ch.F + b.dmin) * sw;
            uint8_t *dst = stage + ch.dev_off + (size_t)((int64_t)b.j0 * ch.P + b.dmin) * sw;
            if (j1 - b.j0 == 1) e = hipMemcpyAsync(dst, src, row_bytes, hipMemcpyHostToDevice, stream);
            else e = hipMemcpy2DAsync(dst, (size_t)ch.P * sw, src, (size_t)ch.F * sw, row_bytes, (size_t)(j1 - b.j0), hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
        }
        for (int32_t j = j1; j < b.j1; j++) {
            const int32_t d = ch.drift[(size_t)j];
            e = hipMemcpyAsync(stage + ch.dev_off + (size_t)((int64_t)j * ch.P + d) * sw, bytes + (size_t)(ch.first + (int64_t)j * ch.F + d) * sw,
                               (size_t)n * sw, hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
        }
    }
    return e;
}

// How [0, width) is cut into chunks of frames for a request that moves in_est bytes of samples in and rgba_bytes of image out.
static void chunk_bounds(int32_t width, size_t in_est, size_t rgba_bytes, bool chunkable, std::vector<int32_t> &bounds)
{
    int chunks = 1;
    // SPECTROPLOT_HIP_RENDER_CHUNKS=k overrides the count (2 .. 16), SPECTROPLOT_HIP_CHUNK_RATIO=r the size of a chunk relative to
    // its neighbour (0.2 .. 1; 1 = equal chunks); both read once
    static const int env_chunks = getenv("SPECTROPLOT_HIP_RENDER_CHUNKS") ? atoi(getenv("SPECTROPLOT_HIP_RENDER_CHUNKS")) : 0;
    static const double env_ratio = getenv("SPECTROPLOT_HIP_CHUNK_RATIO") ? atof(getenv("SPECTROPLOT_HIP_CHUNK_RATIO")) : 0.0;
    const double ratio = env_ratio >= 0.2 && env_ratio <= 1.0 ? env_ratio : 0.65;
    const bool uneven = ratio < 1.0;
    if (chunkable && width >= 1024 && in_est + rgba_bytes >= ((size_t)16 << 20)) {
        chunks = in_est + rgba_bytes >= ((size_t)64 << 20) ? 6 : 4;
        if (env_chunks >= 2 && env_chunks <= sp_context::kMaxChunks && width >= 32 * env_chunks) chunks = env_chunks;
    }
    // The busier direction of the link never pauses; what does not overlap it is one chunk's way in the other direction plus its
    // render: the LAST chunk's image when the samples are the longer transfer, the FIRST chunk's samples when the image is.  So the
    // chunks shrink (or grow) geometrically towards that end - each 0.65 of its neighbour, which also keeps the shorter direction
    // from falling behind - instead of being equal (measured, config 2: 8 equal chunks 2.73 ms, pure two-way copy 2.37 ms; every
    // additional copy call costs the link ~13 us, so few chunks).  Chunks end on multiples of 32 frames.
    bounds.assign(1, 0);
    if (chunks > 1 && uneven) {
        const bool in_heavy = in_est >= rgba_bytes;
        double w[sp_context::kMaxChunks], sum = 0, acc = 0;
        for (int k = 0; k < chunks; k++) sum += (w[k] = std::pow(ratio, in_heavy ? k : chunks - 1 - k));
        for (int k = 0; k + 1 < chunks; k++) {
            acc += w[k];
            const int32_t x = (int32_t)((int64_t)((double)width * acc / sum) & ~(int64_t)31);
            if (x > bounds.back() && x < width) bounds.push_back(x);
        }
        bounds.push_back(width);
    } else {
        for (int k = 0; k < chunks; k++) {
            const int32_t x = k + 1 == chunks ? width : (int32_t)(((int64_t)width * (k + 1) / chunks) & ~(int64_t)31);
            if (x > bounds.back()) bounds.push_back(x);
        }
        if (bounds.back() != width) bounds.push_back(width);
    }
    if (bounds.size() < 2) bounds.push_back(width);   // (width = 0: one empty chunk, so that the reply still gets its initial values)
}

// (tests) The upload plan sp_render would use for a request of this shape - pure host arithmetic, no device.  out[]: packed (0 / 1),
// chunks, device bytes, link bytes; per chunk x0, x1 and, if packed, first, F, P, dev_off, pos2_x0, pos2_last, the bits of stride2,
// blocks, then j0, j1, dmin, dmax per block.
extern "C" int sp_debug_upload_plan(int32_t format, int32_t n, size_t nbytes, int32_t width, int32_t want_image, int64_t *out, size_t capacity,
                                    size_t *used)
{
    if (format < 0 || format >= SP_FMT_COUNT || n < 2 || width < 1 || !out || !used) return SP_ERR_INVALID_ARG;
    const spfmt::Format f = spfmt::describe(format);
    const double sample_count = (double)nbytes / (double)f.width;
    const double stride = width > 1 ? (sample_count - (double)n) / (double)(width - 1) : 0.0;
    const bool stride_ok = stride >= 0.0 && std::isfinite(stride) && 0.5 + stride * (double)(width - 1) < 2147483000.0;
    const size_t rgba_bytes = 4 * (size_t)width * (size_t)n;
    bool sparse = stride_ok && width >= 2 && stride > (double)n
                  && (size_t)(spjs::to_int32(0.5 + stride * (double)(width - 1)) + (int64_t)n) * (size_t)f.width <= nbytes;
    std::vector<PackedChunk> packed;
    std::vector<int32_t> bounds;
    size_t dev = 0, link = nbytes;
    for (int attempt = sparse ? 0 : 1; attempt < 2; attempt++) {
        chunk_bounds(width, attempt == 0 ? (size_t)width * (size_t)n * (size_t)f.width : nbytes, rgba_bytes, want_image && stride_ok, bounds);
        if (attempt == 0) {
            sparse = build_packed_chunks(n, f.width, nbytes, width, stride, bounds, packed, &dev, &link) && packed.size() + 1 == bounds.size();
            if (sparse) break;
            packed.clear();
            dev = 0;
            link = nbytes;
        }
    }
    std::vector<int64_t> v{sparse ? 1 : 0, (int64_t)bounds.size() - 1, (int64_t)dev, (int64_t)link};
    for (size_t c = 0; c + 1 < bounds.size(); c++) {
        v.push_back(bounds[c]);
        v.push_back(bounds[c + 1]);
        if (!sparse) continue;
        const PackedChunk &ch = packed[c];
        int64_t bits;
        memcpy(&bits, &ch.stride2, 8);
        for (int64_t x : {ch.first, ch.F, ch.P, (int64_t)ch.dev_off, ch.pos2_x0, ch.pos2_last, bits, (int64_t)ch.blocks.size()}) v.push_back(x);
        for (const PackedBlock &b : ch.blocks)
            for (int64_t x : {(int64_t)b.j0, (int64_t)b.j1, (int64_t)b.dmin, (int64_t)b.dmax}) v.push_back(x);
    }
    *used = v.size();
    if (v.size() > capacity) return SP_ERR_INVALID_ARG;
    memcpy(out, v.data(), v.size() * 8);
    return SP_OK;
}

// sp_render / sp_render_strip / sp_plan_execute_from_host: the capture comes from HOST memory, in chunks of frames that travel while
// earlier chunks are rendered.  `image_width` is the width in frames of the image reply->rgba points into (the strip's own width for
// sp_render); it only matters for the spectrogram layout, whose rows are image_width pixels apart.
// device_out = false: the reply's pointers are host pointers; the image travels back chunk by chunk, the small outputs in one copy, and
// the call returns when everything has arrived.  device_out = true: the reply's pointers are device pointers (as for sp_plan_execute);
// nothing comes back, and the call returns once everything is queued.
static int render_core(sp_plan *plan, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply, int32_t image_width,
                       bool device_out)
{
    sp_context *ctx = plan->ctx;
    const sp_request *req = &plan->req;
    hipStream_t s = ctx->stream;

    const size_t W = (size_t)width, n = (size_t)req->n, L = (size_t)req->lut_len;
    const size_t rgba_bytes = 4 * W * n;
    const size_t host_pitch = req->waterfall ? 4 * n : 4 * (size_t)image_width;   // bytes between rows of the caller's image
    int rc = SP_OK;
    // small outputs: [c_hist L u64][cb_hist 1000 u64][minmax 2 f64][gauges 3*W u8]
    const size_t small_u64 = L + SP_CB_HIST_SIZE + 2;
    DeviceBuffer &small = ctx->render_small;
    const size_t small_bytes = small_u64 * 8 + 3 * W;
    sp_reply d = *reply;
    if (!device_out) {
        rc = ctx->out_rgba.reserve(rgba_bytes + 16);
        if (!rc) rc = small.reserve(small_bytes + 16);
        if (!rc) rc = ctx->host_small.reserve(small_bytes + 16);
        if (rc) return fail(ctx, rc, "sp_render: out of memory");
        uint64_t *d_c = (uint64_t *)small.p, *d_cb = d_c + L;
        double *d_mm = (double *)(d_cb + SP_CB_HIST_SIZE);
        uint8_t *d_g = (uint8_t *)(d_mm + 2);
        d.rgba = reply->rgba ? (uint8_t *)ctx->out_rgba.p : nullptr;
        d.gauge_mins = d_g;
        d.gauge_maxs = d_g + W;
        d.gauge_amps = d_g + 2 * W;
        d.c_hist = d_c;
        d.cb_hist = d_cb;
        d.dbfs_minmax = d_mm;
    }

    // Large requests are rendered in chunks of frames: chunk k's samples travel to the device while chunk k-1 is rendered and
    // chunk k-2's part of the image travels back (PCIe is full duplex; the kernels are a few per cent of the copies).  Chunks end
    // on multiples of 32 frames (whole write-out groups); a chunk needs the samples up to the end of its last frame.
    const spfmt::Format f = spfmt::describe(req->format);
    const double sample_count = (double)nbytes / (double)f.width;
    const double stride = width > 1 ? (sample_count - (double)req->n) / (double)(width - 1) : 0.0;
    const bool stride_ok = stride >= 0.0 && std::isfinite(stride) && 0.5 + stride * (double)(width - 1) < 2147483000.0;
    // a sparse request (stride > n, every frame inside the capture, the frame-loop kernel): only the frames' own samples are uploaded
    bool sparse = stride_ok && width >= 2 && stride > (double)req->n && plan_kernel(plan) == 3 && !getenv("SPECTROPLOT_HIP_NO_PACKED_UPLOAD")
                  && (size_t)(spjs::to_int32(0.5 + stride * (double)(width - 1)) + (int64_t)req->n) * (size_t)f.width <= nbytes;
    std::vector<PackedChunk> packed;
    std::vector<int32_t> bounds;
    int chunks = 1;
    size_t packed_dev_bytes = 0, packed_link_bytes = 0;
    for (int attempt = sparse ? 0 : 1; attempt < 2; attempt++) {
        const size_t in_est = attempt == 0 ? W * n * (size_t)f.width : nbytes;
        // (device_out: no image crosses the link, so the samples are the longer transfer whatever the image weighs)
        chunk_bounds(width, in_est, device_out ? 0 : rgba_bytes, (device_out || reply->rgba != nullptr) && stride_ok, bounds);
        chunks = (int)bounds.size() - 1;
        if (attempt == 0) {
            sparse = build_packed_chunks(req->n, f.width, nbytes, width, stride, bounds, packed, &packed_dev_bytes, &packed_link_bytes)
                     && (int)packed.size() == chunks;
            if (sparse) break;
            packed.clear();
        }
    }
    ctx->last_upload_bytes = sparse ? packed_link_bytes : nbytes;
    rc = ctx->in_bytes.reserve(sparse ? packed_dev_bytes : nbytes + 16);
    if (rc) return fail(ctx, rc, "sp_render: out of memory");
    hipError_t e = hipSuccess;
    if (chunks > 1) {
        if (!ctx->copy_in) e = hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking);
        if (e == hipSuccess && !ctx->copy_out) e = hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking);
        for (int k = 0; k < chunks && e == hipSuccess; k++) {
            if (!ctx->ev_arrived[k]) e = hipEventCreateWithFlags(&ctx->ev_arrived[k], hipEventDisableTiming);
            if (e == hipSuccess && !ctx->ev_rendered[k]) e = hipEventCreateWithFlags(&ctx->ev_rendered[k], hipEventDisableTiming);
        }
        // device_out returns without waiting: whatever the stream still holds (an earlier request reading the staging buffer, the
        // caller's own work) comes before this request's first copy
        if (e == hipSuccess && device_out) {
            e = hipEventRecord(ctx->ev_rendered[0], s);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_in, ctx->ev_rendered[0], 0);
        }
        if (e != hipSuccess) return hip_fail(ctx, e, "sp_render streams");
    }
    // a packed chunk as the kernel sees it: (virtual) sample 0 of its layout, a length that covers its last frame and one spare sample
    // (3-byte samples are fetched as dwords), its stride
    auto packed_source = [&](const PackedChunk &ch) {
        PackedSource ps;
        ps.bytes = (const uint8_t *)ctx->in_bytes.p + ch.dev_off - (size_t)ch.pos2_x0 * (size_t)f.width;
        ps.nbytes = (size_t)(ch.pos2_last + (int64_t)req->n + 1) * (size_t)f.width;
        ps.nbytes -= ps.nbytes % (size_t)f.elem;
        ps.stride = ch.stride2;
        return ps;
    };
    // (nothing to clear: the kernels overwrite every histogram count, both range values and every gauge byte)
    if (chunks == 1) {
        if (sparse) {
            e = upload_packed_chunk(packed[0], req->n, f.width, bytes, nbytes, (uint8_t *)ctx->in_bytes.p, s);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(s);
                return hip_fail(ctx, e, "sp_render packed upload");
            }
            const PackedSource ps = packed_source(packed[0]);
            rc = plan_execute_range(plan, ctx->in_bytes.p, nbytes, width, 0, width, true, true, &d, &ps);
        } else {
            if (nbytes) e = hipMemcpyAsync(ctx->in_bytes.p, bytes, nbytes, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) return hip_fail(ctx, e, "sp_render upload");
            rc = sp_plan_execute(plan, ctx->in_bytes.p, nbytes, width, &d);
        }
        if (rc) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
    } else {
        size_t sent = 0;
        for (int k = 0; k < chunks; k++) {
            const int32_t x0 = bounds[(size_t)k], x1 = bounds[(size_t)k + 1];
            PackedSource ps{};
            if (sparse) {
                e = upload_packed_chunk(packed[(size_t)k], req->n, f.width, bytes, nbytes, (uint8_t *)ctx->in_bytes.p, ctx->copy_in);
                ps = packed_source(packed[(size_t)k]);
            } else {
                size_t need = nbytes;
                if (k + 1 < chunks) {
                    const int64_t last_start = spjs::to_int32(0.5 + stride * (double)(x1 - 1));          // worker.js:72
                    need = (size_t)(last_start + req->n) * (size_t)f.width;
                    if (need > nbytes) need = nbytes;
                }
                if (need > sent) {
                    e = hipMemcpyAsync((char *)ctx->in_bytes.p + sent, bytes + sent, need - sent, hipMemcpyHostToDevice, ctx->copy_in);
                    sent = need;
                }
            }
            if (e == hipSuccess) e = hipEventRecord(ctx->ev_arrived[k], ctx->copy_in);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, ctx->ev_arrived[k], 0);
            if (e != hipSuccess) break;
            rc = plan_execute_range(plan, ctx->in_bytes.p, nbytes, width, x0, x1, k == 0, k + 1 == chunks, &d, sparse ? &ps : nullptr);
            if (rc) break;
            if (device_out) continue;
            e = hipEventRecord(ctx->ev_rendered[k], s);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_out, ctx->ev_rendered[k], 0);
            if (e != hipSuccess) break;
            if (x1 > x0) {
                if (req->waterfall) {   // rows width-1-x: the chunk is one contiguous band of rows
                    const size_t off = 4 * n * (size_t)(width - x1);
                    e = hipMemcpyAsync(reply->rgba + off, (char *)ctx->out_rgba.p + off, 4 * n * (size_t)(x1 - x0), hipMemcpyDeviceToHost,
                                       ctx->copy_out);
                } else {                // columns x0 .. x1-1 of every row
                    e = hipMemcpy2DAsync(reply->rgba + 4 * (size_t)x0, host_pitch, (char *)ctx->out_rgba.p + 4 * (size_t)x0, 4 * W,
                                         4 * (size_t)(x1 - x0), n, hipMemcpyDeviceToHost, ctx->copy_out);
                }
                if (e != hipSuccess) break;
            }
        }
        if (rc || e != hipSuccess) {
            (void)hipStreamSynchronize(ctx->copy_in);
            (void)hipStreamSynchronize(s);
            (void)hipStreamSynchronize(ctx->copy_out);
            ctx->acc_dirty = true;
            return rc ? rc : hip_fail(ctx, e, "sp_render chunk");
        }
    }
    if (device_out) return SP_OK;
    auto down = [&](void *dst, const void *src, size_t bytes_) {
        if (dst && bytes_ && e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes_, hipMemcpyDeviceToHost, s);
    };
    if (chunks == 1) {
        if (req->waterfall || host_pitch == 4 * W) down(reply->rgba, ctx->out_rgba.p, rgba_bytes);
        else if (reply->rgba && rgba_bytes && e == hipSuccess)   // a column band of a wider image
            e = hipMemcpy2DAsync(reply->rgba, host_pitch, ctx->out_rgba.p, 4 * W, 4 * W, n, hipMemcpyDeviceToHost, s);
    }
    // the six small outputs sit side by side on the device: one copy into the context's page-locked block, handed out from there
    // (six separate copies into pageable memory cost more than the kernels of a small request)
    down(ctx->host_small.p, small.p, small_bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (chunks > 1) {
        const hipError_t e2 = hipStreamSynchronize(ctx->copy_out);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) return hip_fail(ctx, e, "sp_render download");
    const uint8_t *h = (const uint8_t *)ctx->host_small.p;
    const uint8_t *h_g = h + small_u64 * 8;
    if (reply->c_hist) memcpy(reply->c_hist, h, L * 8);
    if (reply->cb_hist) memcpy(reply->cb_hist, h + L * 8, SP_CB_HIST_SIZE * 8);
    if (reply->dbfs_minmax) memcpy(reply->dbfs_minmax, h + (L + SP_CB_HIST_SIZE) * 8, 16);
    if (reply->gauge_mins) memcpy(reply->gauge_mins, h_g, W);
    if (reply->gauge_maxs) memcpy(reply->gauge_maxs, h_g + W, W);
    if (reply->gauge_amps) memcpy(reply->gauge_amps, h_g + 2 * W, W);
    return SP_OK;
}

static int render_host(sp_context *ctx, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply,
                       int32_t image_width)
{
    if (!ctx || !reply) return SP_ERR_INVALID_ARG;
    int rc = validate_request(ctx, req);
    if (rc) return rc;
    if (width < 0) return fail(ctx, SP_ERR_INVALID_ARG, "width < 0");
    if (image_width < width) return fail(ctx, SP_ERR_INVALID_ARG, "image_width < width");
    if (nbytes && !bytes) return fail(ctx, SP_ERR_INVALID_ARG, "bytes is null");
    // the reference constructs its typed view before anything else (worker.js:24)
    if (nbytes % (size_t)spfmt::describe(req->format).elem)
        return fail(ctx, SP_ERR_BYTE_LENGTH, "byte length is not a multiple of the element size");
    SP_HIP(ctx, hipSetDevice(ctx->device));

    if (!ctx->cached_plan || !same_request(ctx->cached_plan, req)) {
        if (ctx->cached_plan) sp_plan_destroy(ctx->cached_plan);
        ctx->cached_plan = nullptr;
        rc = sp_plan_create(ctx, req, &ctx->cached_plan);
        if (rc) return rc;
    }
    return render_core(ctx->cached_plan, bytes, nbytes, width, reply, image_width, false);
}

extern "C" int sp_render(sp_context *ctx, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply)
{
    return render_host(ctx, req, bytes, nbytes, width, reply, width);
}

extern "C" int sp_plan_execute_from_host(sp_plan *plan, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *d_reply)
{
    if (!plan || !d_reply) return SP_ERR_INVALID_ARG;
    sp_context *ctx = plan->ctx;
    if (width < 0) return fail(ctx, SP_ERR_INVALID_ARG, "width < 0");
    if (nbytes && !bytes) return fail(ctx, SP_ERR_INVALID_ARG, "bytes is null");
    if (nbytes % (size_t)plan->fmt.elem) return fail(ctx, SP_ERR_BYTE_LENGTH, "byte length is not a multiple of the element size");
    SP_HIP(ctx, hipSetDevice(ctx->device));
    return render_core(plan, bytes, nbytes, width, d_reply, width, true);
}

extern "C" int sp_context_last_upload_bytes(const sp_context *ctx, size_t *nbytes)
{
    if (!ctx || !nbytes) return SP_ERR_INVALID_ARG;
    *nbytes = ctx->last_upload_bytes;
    return SP_OK;
}

extern "C" int sp_render_strip(sp_context *ctx, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width,
                               const sp_reply *reply, int32_t image_width)
{
    return render_host(ctx, req, bytes, nbytes, width, reply, image_width);
}

// ------------------------------------------------------------------------------------------------- requests by name

extern "C" int sp_named_resolve(const char *window, const char *cmap, const char **window_name, const char **cmap_key, int32_t *lut_len)
{
    // lookup(windows, name) || blackmanHarrisWindow (lib/spectroplot.js:241); lookup(cmaps, name) || cube1_cmap (:252-264)
    if (window_name) *window_name = sphost::window_by_name(window ? window : "");
    int ci = cmap_index(cmap ? cmap : "");
    if (ci < 0) ci = 0;
    if (cmap_key) *cmap_key = spcmap::kEntries[ci].key;
    if (lut_len) *lut_len = spcmap::kEntries[ci].length;
    return SP_OK;
}

extern "C" int sp_render_named(sp_context *ctx, const sp_named_request *nr, const uint8_t *bytes, size_t nbytes, int32_t width,
                               const sp_reply *reply)
{
    if (!ctx || !nr || !reply) return SP_ERR_INVALID_ARG;
    if (nr->n < 1 || sphost::log2_exact(nr->n) < 0) return fail(ctx, SP_ERR_NOT_POW2, "Length is not a power of 2");
    if (nr->n > SP_MAX_N) return fail(ctx, SP_ERR_UNSUPPORTED, "n exceeds SP_MAX_N");
    const std::string f = nr->format ? nr->format : "", w = nr->window ? nr->window : "", c = nr->cmap ? nr->cmap : "";
    const bool same = ctx->cached_plan && !ctx->named_windowc.empty() && ctx->named_format == f && ctx->named_window == w && ctx->named_cmap == c
                      && ctx->named_n == nr->n && ctx->named_ch == (nr->channel_mode ? 1 : 0) && ctx->named_wf == (nr->waterfall ? 1 : 0)
                      && !memcmp(&ctx->named_gain, &nr->gain, 8) && !memcmp(&ctx->named_range, &nr->range, 8);
    if (!same) {
        // the caller's message assembly (lib/spectroplot.js:1113-1146)
        ctx->named_windowc.assign((size_t)nr->n, 0.0);
        double weight = 0.0;
        if (!sphost::window(sphost::window_by_name(w.c_str()), nr->n, ctx->named_windowc.data(), &weight))
            return fail(ctx, SP_ERR_INVALID_ARG, "window");
        ctx->named_block_norm = 1.0 / weight;
        int ci = cmap_index(c.c_str());
        if (ci < 0) ci = 0;                                              // cube1 (lib/spectroplot.js:252-264)
        const spcmap::Entry &e = spcmap::kEntries[ci];
        ctx->named_lut.assign(3 * (size_t)e.length, 0);
        cmap_bytes(ci, ctx->named_lut.data());
        for (int k = 0; k < 3; k++) {                                    // ends forced to black / white (:1129-1130)
            ctx->named_lut[(size_t)k] = 0;
            ctx->named_lut[3 * (size_t)(e.length - 1) + (size_t)k] = 255;
        }
        ctx->named_format = f;
        ctx->named_window = w;
        ctx->named_cmap = c;
        ctx->named_n = nr->n;
        ctx->named_ch = nr->channel_mode ? 1 : 0;
        ctx->named_wf = nr->waterfall ? 1 : 0;
        ctx->named_gain = nr->gain;
        ctx->named_range = nr->range;
    }
    sp_request r{};
    r.format = sphost::parse_format(f.c_str());
    r.n = nr->n;
    r.channel_mode = nr->channel_mode;
    r.waterfall = nr->waterfall;
    r.lut_len = (int32_t)(ctx->named_lut.size() / 3);
    r.block_norm = ctx->named_block_norm;
    r.gain = nr->gain;
    r.range = nr->range;
    r.windowc = ctx->named_windowc.data();
    r.lut_rgb = ctx->named_lut.data();
    // same names and numbers: sp_render finds the cached plan by value (same arrays), nothing is rebuilt or uploaded
    const int rc = sp_render(ctx, &r, bytes, nbytes, width, reply);
    if (rc) ctx->named_windowc.clear();
    return rc;
}
