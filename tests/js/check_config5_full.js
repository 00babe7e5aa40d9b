'use strict'
/**
 * GPU: BASELINE config 5 at full size through the JavaScript boundary: 64 MSample cs12 (192 MiB), n = 8192, zoom x8 (65 536 frames,
 * 87.5 % overlap), 2 GiB of RGBA.  One reply of that size is one element beyond Node 12's typed-array limit, so the capture goes through
 * renderSliced with two HipWorker slices, as the reference's >= 2 workers would take it (SURVEY 8d): each slice has its own stride and a
 * 1 GiB strip.  Checked: histogram totals, and sampled frames of both slices (image column, the three gauges) bit for bit against the
 * JavaScript oracle, which renders a sampled frame from its own n samples (frames are independent given their start).
 */
const O = require('../../oracle/js/worker_oracle.js')
const siggen = require('../../oracle/js/siggen.js')
const { HipWorker, renderSliced, cmaps } = require('../../spectroplot-js_amd/js')

async function main() {
    const fmt = 'CS12', n = 8192, S = 1 << 26, workers = 2
    const width = S / n * 8
    const gen = { kind: 'trinoise', seed: 0x5EED0001, step: 7321, gshift: 11, amp: 0.5, namp: 0.02 }
    const t0 = Date.now()
    const bytes = siggen.generate(fmt, gen, S, 0)
    const t1 = Date.now()
    const w = O.makeWindow('blackmanHarris', n)
    const cmap = cmaps.cube1_cmap.map(c => c.slice())
    const m = await renderSliced({ buffer: bytes.buffer, format: fmt, n, width, workers, window: w, cmap, gain: 6, range: 30, merge: false })
    const t2 = Date.now()
    const failures = []
    if (m.data !== null) failures.push('merge: false still merged')
    if (m.sliceWidth !== width / workers) failures.push('slice width ' + m.sliceWidth)
    const total = m.c_hist.reduce((a, b) => a + b, 0)
    if (total !== width * n) failures.push(`c_hist total ${total} != ${width * n}`)
    const forced = cmap.map(c => c.slice()); forced[0] = [0, 0, 0]; forced[forced.length - 1] = [255, 255, 255]
    const sliceSamples = Math.floor(S / workers)
    let seed = 12345
    const rnd = (k) => { seed = (seed * 1103515245 + 12345) & 0x7fffffff; return seed % k }
    for (let s = 0; s < workers; s++) {
        const r = m.replies[s], W = m.sliceWidth
        if (r.imageData.data.length !== 4 * W * n) { failures.push(`slice ${s}: strip length`); continue }
        const stride = (sliceSamples - n) / (W - 1)                                   // worker.js:50, per slice
        for (const x of [0, W - 1, rnd(W), rnd(W), rnd(W), rnd(W)]) {
            const start = ~~(0.5 + stride * x)
            const fb = siggen.generate(fmt, gen, n, s * sliceSamples + start)
            const o = O.render({ block_norm: 1.0 / w.weight, gain: 6, range: 30, cmap: forced, n, windowc: w.window, width: 1, offset: 0,
                buffer: fb.buffer.slice(fb.byteOffset, fb.byteOffset + fb.byteLength), format: fmt, channelMode: false, waterfall: false })
            const img = r.imageData.data
            for (let y = 0; y < n; y++) {
                const a = 4 * (y * W + x), b = 4 * y
                if (img[a] !== o.imageData.data[b] || img[a + 1] !== o.imageData.data[b + 1] || img[a + 2] !== o.imageData.data[b + 2] || img[a + 3] !== 255) {
                    failures.push(`slice ${s} frame ${x}: pixel row ${y}`); break
                }
            }
            if (r.gauge_mins[x] !== o.gauge_mins[0] || r.gauge_maxs[x] !== o.gauge_maxs[0] || r.gauge_amps[x] !== o.gauge_amps[0]) failures.push(`slice ${s} frame ${x}: gauges`)
        }
    }
    if (failures.length) { console.log(failures.slice(0, 20).join('\n')); process.exit(1) }
    console.log(`config 5 at full size through renderSliced ok: ${workers} slices x ${m.sliceWidth} frames, ${(4 * width * n / 2 ** 30).toFixed(1)} GiB of RGBA, ` +
        `generate ${t1 - t0} ms, render ${t2 - t1} ms`)
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
