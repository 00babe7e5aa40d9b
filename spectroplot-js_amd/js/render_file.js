'use strict'
/**
 * The data half of the reference caller's processData (lib/spectroplot.js:1113-1130, :1206-1244) for Node: evaluates the
 * taper, forces the LUT ends, cuts the capture into `workers` contiguous slices (lib/samples.js:253-258), renders each
 * slice on its own HipWorker (= its own GPU when several are visible) and merges strips, histograms and the dBfs range.
 * Un-rendered columns stay zero, as on the reference's canvas.
 */
const path = require('path')
const { HipWorker, packLut, plainArray } = require('./hip_worker.js')

function native() { return require(path.join(__dirname, '..', 'lib', 'spectroplot_hip.node')) }

/**
 * With `byName: true`, `window` and `cmap` are option names (any spelling the reference's lookup accepts) and the slices go to the
 * workers as named requests (HipWorker.renderNamed -> sp_render_named): taper, block_norm and the end-forced colour map are evaluated
 * inside the library, once per worker while the names repeat.
 * With `device: true` the whole sliced render is ONE native call (sp_group_render_ex): slice r is rendered on group member r - member r
 * on GPU r modulo the visible GPUs.  `gather` says where the strips meet:
 *   'device' (default)  on the root GPU, device to device (RCCL over xGMI between distinct GPUs, peer copies otherwise), merged there;
 *                       the merged image crosses the root's host link once.  For an image that is wanted on the root GPU, and to
 *                       exercise / measure the xGMI gather.
 *   'host'              every member copies its strip straight into its band of the result over its OWN host link (N links side by
 *                       side) and the side outputs are merged on the host: what to use whenever the image is wanted in host memory.
 * The result has the same fields; `replies` then carry the slices' offsets and gauges only (their strips never existed as separate
 * host buffers), `transport` names what moved the strips ('none' | 'rccl' | 'peer' | 'host'), `transportNote` why it was not the first
 * choice (an RCCL failure ends in peer copies, never in a failed render) and `timings` the phases' milliseconds.  Groups are kept per
 * worker count (renderSliced.closeGroups() releases them).
 * With `merge: false` the strips are not copied into one image (`data` is null; `replies` hold them): an image of 2^31 bytes or more -
 * BASELINE config 5 is exactly 2^31 - is beyond what one typed array can hold under Node 12, as it is beyond one canvas.
 * @param {{buffer: ArrayBuffer, format: string, n: number, width: number, workers?: number, window?: string|{window, weight},
 *          cmap: number[][]|string, gain?: number, range?: number, channelMode?: boolean, waterfall?: boolean, byName?: boolean,
 *          merge?: boolean, device?: boolean, gather?: 'device'|'host'}} o
 * @returns {Promise<{data: Uint8ClampedArray, width, height, c_hist, cB_hist, dBfs_min, dBfs_max, sliceWidth, replies}>}
 */
function renderSliced(o, pool) {
    const a = native()
    const workers = o.workers || Math.max(1, a.deviceCount())
    const n = o.n
    const byName = !!o.byName
    const w = byName ? null : (typeof o.window === 'object' && o.window ? o.window : a.window(o.window || 'blackmanHarris', n))
    const block_norm = byName ? 0 : 1.0 / w.weight                      // spectroplot.js:1116
    const cmap = byName ? new Array(a.namedResolve(String(o.window), String(o.cmap)).lutLength) : o.cmap.map(c => c.slice())
    if (!byName) { cmap[0] = [0, 0, 0]; cmap[cmap.length - 1] = [255, 255, 255] }   // spectroplot.js:1129-1130
    const gain = o.gain === undefined ? 6 : o.gain, range = o.range === undefined ? 30 : o.range
    const fmt = a.parseFormat(o.format)
    const width = o.width, sliceWidth = ~~(width / workers)             // spectroplot.js:1208
    if (o.device) {
        if (byName) throw new Error('renderSliced: device: true takes evaluated arrays (window, cmap), not names')
        return renderOnGroup(a, o, { workers, n, w, block_norm, cmap, gain, range, fmt, width, sliceWidth })
    }
    const own = !pool
    pool = pool || Array.from({ length: workers }, () => new HipWorker())
    const merged = o.merge === false ? null : new Uint8ClampedArray(4 * width * n)
    const c_hist = new Array(cmap.length).fill(0), cB_hist = new Array(1000).fill(0)
    let dBfs_min = 0.0, dBfs_max = -200.0
    // one FIFO of resolvers per worker, like the reference's renderCallbacks (lib/spectroplot.js:89-98, :111-115)
    const pending = pool.map(() => [])
    const saved = pool.map(wk => [wk.onmessage, wk.onerror])
    pool.forEach((wk, k) => {
        wk.onmessage = (e) => { const p = pending[k].shift(); if (p) p.resolve(e.data) }
        wk.onerror = (e) => { const p = pending[k].shift(); if (p) p.reject(e.error || new Error(e.message)) }
    })
    const jobs = []
    for (let i = 0; i < workers; i++) {
        const [b0, b1] = a.sliceBounds(o.buffer.byteLength, fmt.sampleWidth, i, workers)
        // the slice is a copy, as in the reference (SampleView.slice -> ArrayBuffer.slice); here it lands in page-locked memory,
        // from where the GPU fetches it at the full rate of the host link
        // (without a device there is nothing to lock pages for: a caller-supplied pool of other workers gets a plain copy)
        const slice = a.deviceCount() > 0 ? a.allocBuffer(b1 - b0) : new ArrayBuffer(b1 - b0)
        new Uint8Array(slice).set(new Uint8Array(o.buffer, b0, b1 - b0))
        const k = i % pool.length
        if (byName) {
            jobs.push(pool[k].renderNamed({ buffer: slice, format: o.format, window: o.window, cmap: o.cmap, n, width: sliceWidth,
                offset: i * sliceWidth, gain, range, channelMode: !!o.channelMode, waterfall: !!o.waterfall }))
            continue
        }
        const message = { block_norm, gain, range, cmap, n, windowc: w.window, width: sliceWidth, offset: i * sliceWidth,
            buffer: slice, format: o.format, channelMode: !!o.channelMode, waterfall: !!o.waterfall }
        jobs.push(new Promise((resolve, reject) => {
            pending[k].push({ resolve, reject })
            pool[k].postMessage(message, [message.buffer])
        }))
    }
    return Promise.all(jobs).then(replies => {
        for (const r of replies) {
            if (r.dBfs_min < dBfs_min) dBfs_min = r.dBfs_min
            if (r.dBfs_max > dBfs_max) dBfs_max = r.dBfs_max
            for (let k = 0; k < 1000; k++) cB_hist[k] += r.cB_hist[k]
            for (let k = 0; k < cmap.length; k++) c_hist[k] += r.c_hist[k]
            if (!merged) continue
            const img = r.imageData.data
            const [px, py, pw, ph] = stripPlacement(r, { width, sliceWidth, n, waterfall: !!o.waterfall })
            const canvasWidth = o.waterfall ? n : width
            for (let y = 0; y < ph; y++) merged.set(img.subarray(4 * y * pw, 4 * (y + 1) * pw), 4 * ((py + y) * canvasWidth + px))
        }
        pool.forEach((wk, k) => { wk.onmessage = saved[k][0]; wk.onerror = saved[k][1] })
        if (own) pool.forEach(wk => wk.terminate())
        return { data: merged, width: o.waterfall ? n : width, height: o.waterfall ? width : n, c_hist, cB_hist, dBfs_min, dBfs_max,
            sliceWidth, replies }
    })
}

const groups = new Map()   // worker count -> {handle, queue}: a group renders one request at a time, like one worker

function renderOnGroup(a, o, q) {
    let entry = groups.get(q.workers)
    if (!entry) {
        const gpus = Math.max(1, a.deviceCount())
        entry = { handle: a.createGroup(Array.from({ length: q.workers }, (_, i) => i % gpus)), queue: Promise.resolve() }
        groups.set(q.workers, entry)
    }
    const g = entry.handle
    const windowc = q.w.window instanceof Float64Array ? q.w.window : new Float64Array(q.w.window)
    const req = { format: q.fmt.id, buffer: o.buffer, n: q.n, windowc, block_norm: q.block_norm, gain: q.gain, range: q.range,
        lut: packLut(q.cmap), width: q.width, channelMode: !!o.channelMode, waterfall: !!o.waterfall, gather: o.gather === 'host' ? 'host' : 'device' }
    const run = () => new Promise((resolve, reject) => {
        a.groupRender(g, req, (err, r) => {
            if (err) { reject(err); return }
            const gm = new Uint8ClampedArray(r.gauge_mins), gx = new Uint8ClampedArray(r.gauge_maxs), ga = new Uint8ClampedArray(r.gauge_amps)
            const replies = []
            for (let i = 0; i < q.workers; i++) {
                const lo = i * q.sliceWidth, hi = lo + q.sliceWidth
                replies.push({ offset: lo, gauge_mins: gm.subarray(lo, hi), gauge_maxs: gx.subarray(lo, hi), gauge_amps: ga.subarray(lo, hi), imageData: null })
            }
            resolve({ data: new Uint8ClampedArray(r.rgba), width: o.waterfall ? q.n : q.width, height: o.waterfall ? q.width : q.n,
                c_hist: plainArray(r.c_hist), cB_hist: plainArray(r.cB_hist), dBfs_min: r.dBfs_min, dBfs_max: r.dBfs_max,
                sliceWidth: q.sliceWidth, replies, transport: r.transport, transportNote: r.transportNote, timings: r.timings, members: r.members })
        })
    })
    const p = entry.queue.then(run)
    entry.queue = p.then(() => null, () => null)
    return p
}

renderSliced.closeGroups = () => {
    for (const e of groups.values()) native().destroyGroup(e.handle)
    groups.clear()
}

/**
 * Where the caller puts a reply's strip: putImageData(newImageData(data, waterfall ? n : sliceWidth), x, y) with
 * (x, y) = (offset, 0), or (0, width - sliceWidth - offset) for the waterfall layout (lib/spectroplot.js:1241-1244).
 * @returns {[number, number, number, number]} x, y, strip width, strip height
 */
function stripPlacement(reply, o) {
    const w = o.waterfall ? o.n : o.sliceWidth
    const h = reply.imageData.data.length / 4 / w
    return o.waterfall ? [0, o.width - o.sliceWidth - reply.offset, w, h] : [reply.offset, 0, w, h]
}

module.exports = { renderSliced, stripPlacement }
