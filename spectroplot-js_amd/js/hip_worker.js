'use strict'
/**
 * HipWorker — a Worker-shaped object for triq-org/spectroplot-js' `workerOrUrl` option, backed by the MI355X HIP library.
 *
 * The reference constructs its render workers with `new workerOrUrl()` when the option is not a URL string
 * (lib/spectroplot.js:100-116) and only uses two members: `postMessage(message, transfer)` and an assignable
 * `onmessage`.  This class provides exactly that, with the reference worker's contract:
 *   - a message without a `buffer` is ignored (the transferable probe, lib/worker.js:158-163, lib/spectroplot.js:118-119);
 *   - every render request produces exactly one reply, replies come back in request order (FIFO per instance);
 *   - the reply has the reference's fields (lib/worker.js:140-155): cB_hist, c_hist, dBfs_min, dBfs_max, offset,
 *     gauge_mins / gauge_maxs / gauge_amps (Uint8ClampedArray) and imageData.data (Uint8ClampedArray).
 * Where the reference worker would throw (and leave the caller's promise pending forever), this class reports an
 * `onerror({message, status})` event (or an 'error' listener) and still never emits a malformed reply.
 *
 * Instances are spread over the visible GPUs round-robin, so the reference's pool of N workers maps onto N devices
 * (its per-worker time slices become per-GPU shards).  There is no CPU fallback.
 */
const path = require('path')

let native = null
function addon() {
    if (!native) native = require(path.join(__dirname, '..', 'lib', 'spectroplot_hip.node'))
    return native
}

let nextDevice = 0

// Renders run on libuv's thread pool (4 threads unless UV_THREADPOOL_SIZE says otherwise, shared with fs / dns work), one thread
// per render in flight: a pool of N HipWorkers on N GPUs needs N of them.  The size is read when the pool first starts, so it is
// raised here, at load time, if nothing has been set; applications that have already used the pool must set it themselves.
function reserveThreads(devices) {
    const want = devices + 4
    const have = parseInt(process.env.UV_THREADPOOL_SIZE || '0', 10)
    if (!(have >= want)) process.env.UV_THREADPOOL_SIZE = String(Math.min(want, 1024))
}

/** cmap entries -> packed Uint8Array with the store semantics of the reference's Uint8ClampedArray image (worker.js:118-121). */
function packLut(cmap) {
    const lut = new Uint8ClampedArray(cmap.length * 3)
    for (let i = 0; i < cmap.length; i++) {
        lut[3 * i] = cmap[i][0]; lut[3 * i + 1] = cmap[i][1]; lut[3 * i + 2] = cmap[i][2]
    }
    return new Uint8Array(lut.buffer)
}

/** Float64Array -> Array of numbers, as the reference's reply carries its histograms (a loop: Array.from() on the two typed arrays took 46 us of a 270 us config-1 message, the loops take 6). */
function plainArray(t) {
    const a = new Array(t.length)
    for (let i = 0; i < t.length; i++) a[i] = t[i]
    return a
}

class HipWorker {
    /** @param {{device?: number}} [options] — device index; default: round-robin over the visible GPUs. */
    constructor(options) {
        this.onmessage = null
        this.onerror = null
        this._listeners = { message: [], error: [] }
        this._queue = Promise.resolve()
        this._closed = false
        const n = addon().deviceCount()
        if (nextDevice === 0) reserveThreads(n)
        if (n < 1) throw Object.assign(new Error('no HIP device: spectroplot-hip has no CPU fallback'), { status: -5 })
        this.device = options && options.device !== undefined ? options.device : (nextDevice++ % n)
        this._ctx = addon().createContext(this.device)
    }

    addEventListener(type, fn) { if (this._listeners[type]) this._listeners[type].push(fn) }
    removeEventListener(type, fn) {
        if (this._listeners[type]) this._listeners[type] = this._listeners[type].filter(f => f !== fn)
    }

    _emit(type, event) {
        const h = type === 'message' ? this.onmessage : this.onerror
        if (typeof h === 'function') h.call(this, event)
        for (const fn of this._listeners[type]) fn.call(this, event)
    }

    /** Same signature as Worker.postMessage; `transfer` is accepted and ignored (the buffer is read in place). */
    postMessage(message, transfer) { // eslint-disable-line no-unused-vars
        if (this._closed) return
        if (!(message && message.buffer)) return                        // worker.js:159
        // requests are serialised per instance, like one worker thread; the GPU work itself runs off the event loop
        // (a worker that has been terminated meanwhile neither starts queued requests nor reports anything)
        this._queue = this._queue.then(() => this._closed ? null : this._render(message)).then(
            reply => { if (reply && !this._closed) this._emit('message', { data: reply }) },
            err => { if (!this._closed) this._emit('error', { message: err.message, status: err.status, error: err }) })
    }

    _request(m) {
        const a = addon()
        const fmt = a.parseFormat(String(m.format))                     // upper-cases, aliases, unknown -> CU8
        const n = m.n
        // (the typed-array constructor converts an Array in one go; Float64Array.from() walks its iterator)
        const windowc = m.windowc instanceof Float64Array ? m.windowc : new Float64Array(m.windowc)
        let buffer = m.buffer
        if (ArrayBuffer.isView(buffer)) buffer = buffer.buffer.slice(buffer.byteOffset, buffer.byteOffset + buffer.byteLength)
        return { format: fmt.id, buffer, n, windowc, block_norm: m.block_norm, gain: m.gain, range: m.range,
            lut: packLut(m.cmap), width: m.width, channelMode: !!m.channelMode, waterfall: !!m.waterfall }
    }

    _wrap(m, r) {
        return {
            cB_hist: plainArray(r.cB_hist), c_hist: plainArray(r.c_hist), dBfs_min: r.dBfs_min, dBfs_max: r.dBfs_max,
            offset: m.offset,
            gauge_mins: new Uint8ClampedArray(r.gauge_mins), gauge_maxs: new Uint8ClampedArray(r.gauge_maxs),
            gauge_amps: new Uint8ClampedArray(r.gauge_amps), imageData: { data: new Uint8ClampedArray(r.rgba) },
        }
    }

    _render(m) {
        return new Promise((resolve, reject) => {
            let req
            try { req = this._request(m) } catch (e) { reject(e); return }
            // (Always on a libuv thread.  Rendering small requests on the calling thread was tried - the hand-off costs 0.06-0.08 ms of a
            // 0.36 ms config-1 message - and is slower: a chain of replies that resolves in microtasks never lets V8 run the finalisers
            // that return reply images to the pool, so every reply takes a fresh, unpinned block: 0.51 ms, tools/js_config1_message.js.)
            addon().render(this._ctx, req, (err, r) => err ? reject(err) : resolve(this._wrap(m, r)))
        })
    }

    /**
     * A request by option names instead of evaluated arrays - not part of the reference worker's wire contract (its messages carry the
     * evaluated taper and colour map, lib/spectroplot.js:1213-1226), but what the reference's caller starts from: the library resolves
     * `window` and `cmap` with the reference's lookup rules and defaults (lib/utils.js:25-40, lib/spectroplot.js:238-264), evaluates
     * taper, block_norm and the end-forced colour map itself (:1113-1130) and keeps the device tables while names and numbers repeat.
     * @param {{buffer: ArrayBuffer, format: string, window: string, cmap: string, n: number, width: number, gain?: number,
     *          range?: number, channelMode?: boolean, waterfall?: boolean, offset?: number}} o
     * @returns {Promise<object>} the reply, fields as in a worker reply; runs in the instance's request order
     */
    renderNamed(o) {
        const run = () => new Promise((resolve, reject) => {
            if (this._closed) { reject(new Error('worker has been terminated')); return }
            let req
            try { req = this._namedRequest(o) } catch (e) { reject(e); return }
            addon().renderNamed(this._ctx, req, (err, r) => err ? reject(err) : resolve(this._wrap(o, r)))
        })
        const p = this._queue.then(run)
        this._queue = p.then(() => null, () => null)
        return p
    }

    _namedRequest(o) {
        let buffer = o.buffer
        if (ArrayBuffer.isView(buffer)) buffer = buffer.buffer.slice(buffer.byteOffset, buffer.byteOffset + buffer.byteLength)
        return { format: String(o.format), window: String(o.window === undefined ? '' : o.window), cmap: String(o.cmap === undefined ? '' : o.cmap),
            buffer, n: o.n, width: o.width, gain: o.gain === undefined ? 6 : o.gain, range: o.range === undefined ? 30 : o.range,
            channelMode: !!o.channelMode, waterfall: !!o.waterfall }
    }

    /** Synchronous form of renderNamed (tests). */
    renderNamedSync(o) { return this._wrap(o, addon().renderNamedSync(this._ctx, this._namedRequest(o))) }

    /** How many plans (table sets on the device) this worker's context has built so far. */
    planCreations() { return addon().planCreations(this._ctx) }

    /** Synchronous render of one message (used by tests and by callers that drive the GPUs themselves). */
    renderSync(m) { return this._wrap(m, addon().renderSync(this._ctx, this._request(m))) }

    /**
     * Worker.terminate(): queued requests are dropped, nothing is emitted any more, and the device context (stream, staging and
     * workspace buffers) is released as soon as the render that may be running on a libuv thread has returned.
     */
    terminate() {
        if (this._closed) return
        this._closed = true
        const ctx = this._ctx
        this._ctx = null
        if (ctx) addon().destroyContext(ctx)
    }
}

/** A constructor bound to one device, for `workerOrUrl: HipWorker.onDevice(3)`. */
HipWorker.onDevice = (device) => class extends HipWorker { constructor() { super({ device }) } }
HipWorker.deviceCount = () => addon().deviceCount()
/**
 * An ArrayBuffer in page-locked host memory.  A message whose `buffer` is one of these is copied to the GPU at the full rate of
 * the host link (pageable memory goes through the runtime's staging copies at less than half of it); js/render_file.js cuts its
 * slices into such buffers, where the reference's SampleView.slice copies into a plain one (lib/samples.js:253-258).
 */
HipWorker.allocBuffer = (nbytes) => addon().allocBuffer(nbytes)

module.exports = { HipWorker, packLut, plainArray }
