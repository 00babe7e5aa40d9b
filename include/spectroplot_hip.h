/*
 * spectroplot_hip.h — C ABI of the MI355X-native I/Q STFT -> RGBA hot path.
 *
 * This library replaces the compute half of triq-org/spectroplot-js' Web Worker (lib/worker.js `renderFft`):
 * sample decode -> taper -> radix-2 DFT -> |X|^2 -> dB -> colour index -> RGBA, plus the worker's side outputs
 * (two histograms, dBfs min/max, three per-frame gauges).  One `sp_context` corresponds to one reference Worker
 * instance (lib/spectroplot.js:85-116): it owns one HIP stream on one device and processes requests in order.
 *
 * Reference interface each entry point replaces (file:line relative to the reference checkout):
 *
 *   sp_format_parse          lib/samples.js:22-162      the format-name table (aliases, unknown -> CU8)
 *   sp_slice_bounds          lib/samples.js:253-258     SampleView.slice: the caller's per-worker byte range
 *   sp_window                lib/windows.js:14-88       named taper generators (optional sugar; the wire carries arrays)
 *   sp_cmap, sp_cmap_key     lib/cube1cmap.js, lib/matplotlibcmaps.js, lib/parabolacmap.js (tables), lib/utils.js:25-40
 *   sp_cmap_generate         lib/soxcmap.js:12-49, lib/naivecmap.js:13-81   the computed maps, evaluated (sp_cmap serves them from here)
 *   sp_render_named          lib/spectroplot.js:1113-1146, 1213-1226   the caller's message assembly from option names
 *   sp_named_resolve         lib/spectroplot.js:238-264, lib/utils.js:25-40   option name -> generator / table entry, with the defaults
 *   sp_twiddles              lib/fft_nayuki.js:42-47    cos/sin tables (exposed for tests)
 *   sp_plan_create           lib/worker.js:30-62        per-request constants + the cached FFT object
 *   sp_render                lib/worker.js:23-156       renderFft(ctx) on host buffers = one postMessage -> one reply
 *   sp_render_strip          lib/worker.js:23-156 + lib/spectroplot.js:1241-1244   the same, written into the strip's band of the caller's image
 *   sp_plan_execute          lib/worker.js:68-137       the frame loop, operands resident in HBM (benchmarks, multi-GPU)
 *   sp_plan_execute_from_host  lib/worker.js:68-137     the same with the capture in host memory (uploaded in chunks under the renders),
 *                                                        outputs resident in HBM: what a group member does with its slice
 *   sp_merge_replies(_batch) lib/spectroplot.js:1229-1238   the caller's merge of the slices' histograms and dBfs range, on the device
 *   sp_place_strips          lib/spectroplot.js:1241-1244   the caller's putImageData of every slice's strip, on the device
 *   sp_group_render          lib/spectroplot.js:1206-1244, lib/samples.js:253-258   the caller's sliced render: one slice per device, the
 *                                                        strips gathered device to device (RCCL / peer copies), merged on the root
 *   sp_synth_*               (none)                     device-side synthetic I/Q for benchmarks
 *
 * The request fields are the reference message's (lib/spectroplot.js:1213-1226):
 *   block_norm, gain, range, cmap -> lut_rgb/lut_len, n, windowc, width, buffer -> bytes/nbytes, format, channelMode,
 *   waterfall; `offset` is only echoed by the worker and stays in the host wrapper.
 * The reply fields are the reference reply's (lib/worker.js:140-155):
 *   cB_hist[1000], c_hist[lut_len], dBfs_min, dBfs_max, gauge_mins/maxs/amps[width], imageData.data[4*width*n].
 *
 * Numerics: the DFT runs in IEEE f64 with the reference's exact butterfly graph and operation order (no FMA
 * contraction), so |X|^2 is bit-identical to the reference's; colour and histogram indices are taken from exact
 * threshold tables built on the host with a restatement of the engine's Math.log10.  There is no CPU fallback:
 * without a HIP device every compute entry point returns SP_ERR_NO_DEVICE.
 *
 * All functions return SP_OK (0) or a negative status; sp_last_error(ctx) gives a message for the last failure on ctx.
 */
#ifndef SPECTROPLOT_HIP_H
#define SPECTROPLOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_VERSION 100          /* 0.1.0 */
#define SP_CB_HIST_SIZE 1000    /* centi-bel histogram bins, lib/worker.js:41 */
#define SP_MAX_LUT 4096         /* largest colour map accepted */
#define SP_MAX_N (1 << 20)      /* largest FFT length accepted */

enum sp_status {
    SP_OK = 0,
    SP_ERR_INVALID_ARG = -1,
    SP_ERR_NOT_POW2 = -2,        /* reference: throw 'Length is not a power of 2' (lib/fft_nayuki.js:38-39) */
    SP_ERR_BYTE_LENGTH = -3,     /* reference: RangeError, byte length not a multiple of the view's element size */
    SP_ERR_UNSUPPORTED = -4,     /* valid for the reference but outside this library's limits (documented per call) */
    SP_ERR_NO_DEVICE = -5,
    SP_ERR_HIP = -6,
    SP_ERR_NOMEM = -7
};

/* sample formats, lib/samples.js:30-155 */
enum sp_format {
    SP_FMT_CU4 = 0, SP_FMT_CS4, SP_FMT_CU8, SP_FMT_CS8, SP_FMT_CU12, SP_FMT_CS12, SP_FMT_CU16, SP_FMT_CS16,
    SP_FMT_CU32, SP_FMT_CS32, SP_FMT_CU64, SP_FMT_CS64, SP_FMT_CF32, SP_FMT_CF64, SP_FMT_COUNT
};

typedef struct sp_context sp_context;
typedef struct sp_plan sp_plan;

/* One render request minus the sample buffer (the reference message, lib/spectroplot.js:1213-1226). Host pointers. */
typedef struct sp_request {
    int32_t format;          /* enum sp_format */
    int32_t n;               /* FFT length, power of two, 2 .. SP_MAX_N */
    int32_t channel_mode;    /* 0 = I/Q, 1 = L/R split (lib/fft_nayuki.js:103-119) */
    int32_t waterfall;       /* 0 = spectrogram (n rows x width cols), 1 = waterfall (width rows x n cols) */
    int32_t lut_len;         /* colour map entries, 1 .. SP_MAX_LUT */
    int32_t reserved;
    double block_norm;       /* 1 / sum(taper) */
    double gain;             /* dB */
    double range;            /* dB, finite and > 0 */
    const double *windowc;   /* [n] evaluated taper */
    const uint8_t *lut_rgb;  /* [3 * lut_len] r,g,b per entry (ends already forced by the caller if wanted) */
} sp_request;

/* Reply buffers.  For sp_render these are host pointers, for sp_plan_execute device pointers. Any may be NULL. */
typedef struct sp_reply {
    uint8_t *rgba;           /* [4 * width * n] */
    uint8_t *gauge_mins;     /* [width] */
    uint8_t *gauge_maxs;     /* [width] */
    uint8_t *gauge_amps;     /* [width] */
    uint64_t *c_hist;        /* [lut_len]  overwritten with the counts of this request */
    uint64_t *cb_hist;       /* [SP_CB_HIST_SIZE]  likewise */
    double *dbfs_minmax;     /* [2] = {dBfs_min, dBfs_max} */
} sp_reply;

int sp_version(void);
const char *sp_status_string(int status);
const char *sp_last_error(const sp_context *ctx);

/* ---- pure host helpers (no device needed) ------------------------------------------------------------- */

/* Maps a format name (any case, aliases, unknown -> CU8) to its id and bytes per complex sample. */
int sp_format_parse(const char *name, int32_t *format, int32_t *sample_width);
/* Element size of the typed view the reference lays over the buffer (byte length must be a multiple of it). */
int sp_format_element_size(int32_t format);
/* The caller's slice `index` of `count` over a capture of `nbytes` bytes: [*begin, *end) in bytes. */
int sp_slice_bounds(size_t nbytes, int32_t sample_width, int32_t index, int32_t count, size_t *begin, size_t *end);
/* Named tapers: "rectangular", "bartlett", "hamming", "hann", "blackman", "blackmanHarris" (exact names). */
int sp_window(const char *name, int32_t n, double *window, double *weight);
/*
 * Colour maps of the reference under its own keys ("cube1_cmap", "sox_cmap", ... "viridis_cmap", "parabola_cmap"), in the key
 * order of its merged table (lib/spectroplot.js:41).  sp_cmap resolves `name` as lib/utils.js:25-40 does (exact, then
 * case-insensitive, then case-insensitive prefix, first hit in key order) and copies the r,g,b triples as the reference's
 * modules evaluate them, i.e. BEFORE the caller's end forcing (lib/spectroplot.js:1129-1130).  *lut_len receives the entry count
 * (also when rgb is NULL or capacity_entries is too small, which returns SP_ERR_INVALID_ARG); an unknown name returns
 * SP_ERR_UNSUPPORTED.
 */
int sp_cmap_count(void);
const char *sp_cmap_key(int32_t index);
int sp_cmap(const char *name, uint8_t *rgb, int32_t capacity_entries, int32_t *lut_len);
/*
 * The reference's computed colour maps as the functions they are (lib/soxcmap.js:12-49 `sox_cmap`; lib/naivecmap.js:13-81 `naive_cmap`,
 * `grayscale_cmap`, `roentgen_cmap`, `phosphor_cmap`): `stops` entries (the reference exports them at 256) of r, g, b into rgb[3 * stops].
 * sp_cmap serves these five from the same generators; SP_ERR_UNSUPPORTED for any other key (the other maps are literal tables there too).
 */
int sp_cmap_generate(const char *key, int32_t stops, uint8_t *rgb);
/* cosTable / sinTable of the reference's FFT object, n/2 entries each. */
int sp_twiddles(int32_t n, double *cos_table, double *sin_table);
/* The engine's Math.log10 as restated by this library (exposed so tests can pin it). */
double sp_js_log10(double x);

/* ---- device ------------------------------------------------------------------------------------------- */

int sp_device_count(int32_t *count);
/* One context = one reference Worker: one device, one stream, in-order execution. */
int sp_context_create(int32_t device, sp_context **ctx);
void sp_context_destroy(sp_context *ctx);
/* Uses an existing hipStream_t instead of the context's own (e.g. the caller's framework stream). NULL restores. */
int sp_context_set_stream(sp_context *ctx, void *hip_stream);
/* The stream a caller bound with sp_context_set_stream, NULL while the context uses its own (what to hand back to set_stream to restore). */
int sp_context_get_stream(const sp_context *ctx, void **hip_stream);
int sp_context_synchronize(sp_context *ctx);

/*
 * renderFft on host buffers: copies `bytes` to the device, renders, copies the reply back, synchronously.
 * Plans are cached inside the context while n / taper / LUT / gain / range / block_norm stay the same, as the
 * reference caches its FFT object (lib/worker.js:59-62).
 */
int sp_render(sp_context *ctx, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply);
/*
 * The same render delivered where the caller's merge would put it (lib/spectroplot.js:1241-1244, putImageData(strip, offset, 0)):
 * reply->rgba points at the strip's first pixel INSIDE an image of image_width >= width frames - spectrogram layout: column `offset` of
 * row 0, rows 4 * image_width bytes apart; waterfall layout: the first of the strip's `width` contiguous rows.  Everything else as
 * sp_render (sp_render is sp_render_strip with image_width = width).
 */
int sp_render_strip(sp_context *ctx, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply,
                    int32_t image_width);
/*
 * Bytes of samples the last sp_render / sp_render_strip / sp_render_named on this context sent over the host link.  A request whose
 * frames are more than n samples apart (stride > n, lib/worker.js:50: the reference's loop skips the samples in between, :70-75) and lie
 * inside the capture uploads the frames' own samples only, as rows of pitched copies into a packed device buffer; every other request
 * uploads the capture as it is.  SPECTROPLOT_HIP_NO_PACKED_UPLOAD=1 turns the packed upload off.
 * A large request (>= 16 MiB of samples + image, width >= 1024) is pipelined in chunks of frames: 4, from 64 MiB 6, of UNEVEN size -
 * each 0.65 of its neighbour, shrinking towards the end of the longer transfer's direction.  SPECTROPLOT_HIP_RENDER_CHUNKS=2..16
 * overrides the count and SPECTROPLOT_HIP_CHUNK_RATIO=0.2..1 the ratio (1 = equal chunks); both are read once per process and neither
 * applies below that gate (the count also needs width >= 32 per chunk).
 */
int sp_context_last_upload_bytes(const sp_context *ctx, size_t *nbytes);
/*
 * (tests) The upload plan sp_render would follow for a request of this shape, without a device: how [0, width) is cut into chunks of
 * frames and, for a sparse request, the packed layout and the pitched copies of every chunk.  out[] receives int64 words: packed (0 / 1),
 * chunks, device bytes, link bytes; per chunk x0, x1 and, if packed: the capture's sample where frame x0 starts, the samples between
 * source rows (floor(stride)) and between device rows, the chunk's byte offset on the device, the kernel's position of frames x0 and
 * x1 - 1, the bit pattern of the kernel's stride, the number of pitched copies, then first row, end row, smallest and largest drift
 * per copy.  *used = words needed (SP_ERR_INVALID_ARG if capacity is smaller).  tests/test_upload_plan_cpu.py checks every frame of
 * thousands of shapes against the reference's own positions (lib/worker.js:72).
 */
int sp_debug_upload_plan(int32_t format, int32_t n, size_t nbytes, int32_t width, int32_t want_image, int64_t *out, size_t capacity,
                         size_t *used);

/*
 * The same with the request given by names, as the reference's caller assembles its message from options
 * (lib/spectroplot.js:1113-1146): window = lookup(windows, name) or blackmanHarris, block_norm = 1 / weight, cmap = lookup or
 * cube1 with its ends forced to black / white, format by sp_format_parse.  The plan (and its device tables) is kept while the
 * names and numbers repeat: nothing is re-evaluated or re-uploaded then.
 */
typedef struct sp_named_request {
    const char *format;      /* "cu8", "CF32", ... (unknown -> CU8, as the reference) */
    const char *window;      /* "hann", "blackmanHarris", ... (lookup rules of lib/utils.js:25-40; no hit -> blackmanHarris) */
    const char *cmap;        /* "viridis", "cube1_cmap", ... (same lookup; no hit -> cube1) */
    int32_t n;
    int32_t channel_mode, waterfall;
    double gain, range;
} sp_named_request;
int sp_render_named(sp_context *ctx, const sp_named_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply);
/*
 * What the two option names of a named request resolve to (no device needed): the taper's plain name as sp_window takes it, the
 * colour map's key and its entry count (the caller sizes the reply's c_hist with it).  Any output may be NULL.
 */
int sp_named_resolve(const char *window, const char *cmap, const char **window_name, const char **cmap_key, int32_t *lut_len);
/*
 * How many plans - i.e. sets of taper / twiddle / LUT / threshold tables evaluated on the host and uploaded to the device - this
 * context has built so far, through sp_plan_create, sp_render or sp_render_named.  A request that repeats the previous one's
 * constants (sp_render: the same arrays by value; sp_render_named: the same names and numbers) does not add to it.
 */
int sp_context_plan_creations(const sp_context *ctx, int64_t *count);

/* Pre-evaluated request constants resident on the device: twiddles, taper, RGBA LUT, threshold tables. */
int sp_plan_create(sp_context *ctx, const sp_request *req, sp_plan **plan);
void sp_plan_destroy(sp_plan *plan);
/*
 * The frame loop on device-resident operands, asynchronous on the context's stream.
 * d_bytes: device pointer to the raw capture (nbytes bytes, 16-byte aligned); reply: device pointers.
 * Every output of the reply is overwritten, histograms included: a reply holds the counts of its own request, as the
 * reference's worker returns fresh arrays (lib/worker.js:40-41); the caller sums the slices (lib/spectroplot.js:1229-1238).
 * The reply's arrays must live in DEVICE memory of the context's device, c_hist / cb_hist / dbfs_minmax 8-byte aligned: the frame-loop
 * kernel clears them itself and its workgroups add their shares with device atomics (one kernel per call; no separate finish launch
 * for requests the frame loop covers).
 * Not capturable: every launch carries the number of its request, which its workgroups compare with what workgroup 0 publishes once it
 * has cleared the reply, so a launch replayed from a hipGraph would not wait.  A call on a stream that is being captured returns
 * SP_ERR_UNSUPPORTED.  (The wait relies on workgroup 0 being dispatched with the first wave of workgroups, as the hardware does; it is
 * bounded - a launch that never sees the number traps instead of hanging.  A trap is a failed launch: the HIP context of the PROCESS is
 * unusable afterwards - every later call returns SP_ERR_HIP - and can only be had back in a fresh process.)
 */
int sp_plan_execute(sp_plan *plan, const void *d_bytes, size_t nbytes, int32_t width, const sp_reply *d_reply);
/*
 * The same request with the capture in HOST memory and the reply in DEVICE memory: the samples travel in chunks of frames on a copy
 * stream of the context while earlier chunks are rendered (the chunk schedule of sp_render; a sparse request uploads only the samples
 * its frames read), nothing comes back.  Returns once everything is queued: `bytes` must stay valid, and the reply is complete, when
 * the context's stream has been synchronised (page-locked `bytes`: the copies are asynchronous and at link rate).
 */
int sp_plan_execute_from_host(sp_plan *plan, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *d_reply);
/*
 * The caller's merge of slice replies (lib/spectroplot.js:1229-1238) on device-resident side outputs: `count` records of
 * [c_hist u64[lut_len] | cB_hist u64[SP_CB_HIST_SIZE] | dBfs_min f64 | dBfs_max f64] laid end to end at d_records (what an
 * all-gather of the slices' records delivers) are reduced to element-wise sums and min / max.  Outputs are device pointers;
 * any may be NULL.  Asynchronous on the context's stream.
 */
int sp_merge_replies(sp_context *ctx, const void *d_records, int32_t count, int32_t lut_len, uint64_t *d_c_hist, uint64_t *d_cb_hist,
                     double *d_dbfs_minmax);
/*
 * The same merge for a BATCH of renders in one launch: d_gathered holds what an all-gather of every rank's batch delivers -
 * [rank][render][record], `ranks` x `renders` records of lut_len + SP_CB_HIST_SIZE + 2 words - and d_merged receives `renders` merged
 * records [c_hist | cB_hist | dBfs_min, dBfs_max] end to end (device pointers).  What bench.py --gpus N runs once per collective
 * (one small launch per render between the frame loops costs a launch gap each).  Asynchronous on the context's stream.
 */
int sp_merge_replies_batch(sp_context *ctx, const void *d_gathered, int32_t ranks, int32_t renders, int32_t lut_len, void *d_merged);
/*
 * The caller's strip placement (lib/spectroplot.js:1241-1244: putImageData(strip, offset, 0), or (0, width - sliceWidth - offset) for the
 * waterfall layout) on device-resident strips: `count` strips of slice_width frames each, laid end to end at d_strips (what a gather of
 * the ranks' strips delivers, rank order), are placed in the merged image at d_image (n rows x width columns of RGBA, or width rows x n
 * columns for the waterfall layout).  Columns / rows beyond count * slice_width are left as they are (the caller's canvas keeps them
 * blank: clear the image first).  Asynchronous on the context's stream.
 */
int sp_place_strips(sp_context *ctx, uint8_t *d_image, const uint8_t *d_strips, int32_t count, int32_t n, int32_t width,
                    int32_t slice_width, int32_t waterfall);
/*
 * The caller's sliced render from one process (lib/spectroplot.js:1206-1244): a group owns one context per listed device (a device may be
 * listed more than once: every entry is a member with its own context and stream).  sp_group_render cuts the capture into
 * sp_group_size() slices as SampleView.slice does (lib/samples.js:253-258), uploads slice r to member r and renders it there - all
 * members at once - with sliceWidth = ~~(width / members) frames each; the strips and the slices' side outputs then travel to the root
 * member's device (member 0) without visiting host memory: grouped ncclSend / ncclRecv (RCCL over xGMI; librccl is loaded at run time)
 * when the members sit on distinct devices, peer copies otherwise or when SPECTROPLOT_HIP_NO_RCCL is set; there sp_merge_replies and
 * sp_place_strips do the caller's merge, and the merged image returns in ONE copy.
 * `reply` holds host pointers: rgba [4 * width * n] (columns beyond members * sliceWidth are zero, as the caller's canvas leaves them),
 * c_hist / cb_hist / dbfs_minmax merged over the slices (starting from 0 and (0, -200), :1125-1126), gauge_* [width] with slice r's
 * gauges at [r * sliceWidth, (r + 1) * sliceWidth).  Any may be NULL.  Plans are kept while the request's constants repeat.
 * sp_group_transport names what moved the strips in the last render: "none" (one member), "rccl", "peer" or "host".
 *
 * sp_group_render_ex chooses where the strips meet (sp_group_render = SP_GROUP_GATHER_DEVICE):
 *   SP_GROUP_GATHER_DEVICE  as above: the image is assembled in the root's HBM and comes back over the root's host link in one copy.
 *                           Peer copies and waterfall-layout RCCL receives land in the image itself (root memory = the image); the
 *                           spectrogram layout under RCCL receives whole strips next to it and re-tiles them on the root.  The mode
 *                           for an image that is consumed on the root GPU, and the one that exercises the xGMI gather.
 *   SP_GROUP_GATHER_HOST    the image is bound for the host anyway, so nothing is gathered on a device: every member runs the chunked
 *                           sp_render_strip pipeline on its own slice and copies its strip straight into its band of `reply->rgba`
 *                           over its OWN host link (N links side by side instead of one); histograms, dBfs range and gauges are
 *                           merged on the host.  transport = "host".  The mode to use whenever reply->rgba is host memory.
 * Environment (read when the group is created):
 *   SPECTROPLOT_HIP_NO_RCCL=1      never use RCCL
 *   SPECTROPLOT_HIP_FORCE_RCCL=1   use RCCL whatever the member list looks like: a one-member group then moves the root's own strip and
 *                                  record through a grouped self ncclSend / ncclRecv (what a one-GPU box can execute of the transport);
 *                                  members sharing a device make ncclCommInitAll fail, which falls back to peer copies like any other
 *                                  RCCL failure
 *   SPECTROPLOT_HIP_RCCL_LIB=path  the library to dlopen instead of librccl.so.1
 * An RCCL failure (library missing, init, send / receive, group end) never fails the render: the member streams are drained, the
 * communicators are aborted, the render is completed with peer copies, RCCL is not tried again on this group, and
 * sp_group_transport_note tells what happened (also: peer access that could not be enabled).
 * sp_group_last_timings: milliseconds of the last render's phases - upload + render (slowest member, device events), gather (from the
 * root's render to the assembled image on the root), download (root to host); in host mode the first is the slowest member's whole
 * sp_render_strip by the host clock and the other two are 0.
 */
typedef struct sp_group sp_group;
enum sp_group_gather { SP_GROUP_GATHER_DEVICE = 0, SP_GROUP_GATHER_HOST = 1 };
int sp_group_create(const int32_t *devices, int32_t count, sp_group **group);
void sp_group_destroy(sp_group *group);
int sp_group_size(const sp_group *group);
int sp_group_render(sp_group *group, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply);
int sp_group_render_ex(sp_group *group, const sp_request *req, const uint8_t *bytes, size_t nbytes, int32_t width, const sp_reply *reply,
                       int32_t gather);
const char *sp_group_transport(const sp_group *group);
const char *sp_group_transport_note(const sp_group *group);
int sp_group_last_timings(const sp_group *group, double *render_ms, double *gather_ms, double *download_ms);
/* Which RCCL this group has loaded: "<path of the library>, ncclGetVersion <code>, <n> communicator(s)"; empty while none is loaded. */
int sp_group_rccl_info(const sp_group *group, char *text, size_t capacity);
/* Device bytes the root member holds for the gather beyond its own strip (image + staging), after the last render. */
int sp_group_root_bytes(const sp_group *group, size_t *image_bytes, size_t *staging_bytes);
const char *sp_group_last_error(const sp_group *group);

/* Name of the kernel sp_plan_execute launches: "frames" (64 <= n <= 8192, LUT <= 256 entries) or "scratch_radix2" (everything else). */
const char *sp_plan_kernel_name(const sp_plan *plan);
/* Forces a kernel (tests compare the two device paths): 0 automatic, 1 scratch_radix2, 3 frames (2: removed, SP_ERR_UNSUPPORTED). */
int sp_plan_force_kernel(sp_plan *plan, int32_t which);

/*
 * Page-locked host memory for request / reply buffers: sp_render moves pinned buffers at the full rate of the host link, pageable
 * ones through the runtime's staging copies.  The N-API addon backs the reply ArrayBuffers it hands out with these.
 */
int sp_host_alloc(size_t nbytes, void **ptr);
void sp_host_free(void *ptr);
/* Page-locks (and later releases) memory the caller owns, e.g. a reply block that is being recycled. */
int sp_host_register(void *ptr, size_t nbytes);
void sp_host_unregister(void *ptr);

/* Device memory helpers so that non-HIP hosts (Node, ctypes) can keep operands resident. */
int sp_device_alloc(sp_context *ctx, size_t nbytes, void **d_ptr);
int sp_device_free(sp_context *ctx, void *d_ptr);
int sp_device_upload(sp_context *ctx, void *d_dst, const void *src, size_t nbytes);
int sp_device_download(sp_context *ctx, void *dst, const void *d_src, size_t nbytes);
int sp_device_memset(sp_context *ctx, void *d_ptr, int value, size_t nbytes);

/*
 * Benchmark input: fills d_bytes with `count` samples of the seeded tone + noise signal (definition: DESIGN.md,
 * tests/siggen.py 'trinoise'), starting at global sample index t0.  Bit-identical to the CPU generators.
 */
int sp_synth_trinoise(sp_context *ctx, void *d_bytes, int32_t format, uint64_t t0, uint64_t count,
                      uint32_t seed, uint32_t step, uint32_t gshift, double amp, double namp);

/* Elapsed milliseconds of the last sp_plan_execute's main kernel on this context (HIP events on its stream). */
int sp_context_last_kernel_ms(sp_context *ctx, float *ms);
/* Enables (1) / disables (0) per-execute HIP event timing; off by default. */
int sp_context_enable_timing(sp_context *ctx, int32_t on);
/*
 * Calibration for the figure above: elapsed milliseconds of the same event pair around a one-wavefront no-op kernel on the
 * context's stream (minimum of several tries).  It is the dispatch latency an event pair adds to whatever it brackets; a
 * profiler's kernel duration (rocprofv3 --kernel-trace) does not contain it.
 */
int sp_context_event_pair_overhead_ms(sp_context *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif
