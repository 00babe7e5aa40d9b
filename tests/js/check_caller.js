'use strict'
/**
 * The caller's half of the seam against the reference's OWN caller: tests/golden/caller.json holds what lib/spectroplot.js'
 * startWorkers + processData did when run unmodified with 1, 2 and 8 workers (oracle/ref_harness.mjs): which members of a worker
 * instance they touch, every message they post, every putImageData on the fft canvas, every fillRect on the gauge canvases and the
 * merged histograms / dBfs range.  js/render_file.js (renderSliced, stripPlacement) and js/consumers.js (gaugeColumns) must do the
 * same.  `node check_caller.js oracle` drives them with a pool of JavaScript-oracle workers (CPU), `... hip` with HipWorker (GPU).
 */
const fs = require('fs')
const path = require('path')
const assert = require('assert')
const G = require('./golden_util.js')
const O = require('../../oracle/js/worker_oracle.js')
const siggen = require('../../oracle/js/siggen.js')
const { HipWorker, renderSliced, stripPlacement, gaugeColumns, cmaps, windows } = require('../../spectroplot-js_amd/js')

const kind = process.argv[2] || 'oracle'
const K = JSON.parse(fs.readFileSync(path.join(G.gdir, 'caller.json'), 'utf8'))
const spec = JSON.parse(fs.readFileSync(path.join(G.gdir, 'cases.json'), 'utf8')).caller_kat
const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.readBigUInt64LE().toString(16).padStart(16, '0') }
const f64bytes = (arr) => { const b = Buffer.alloc(arr.length * 8); arr.forEach((v, i) => b.writeDoubleLE(v, i * 8)); return b }

// a Worker-shaped stand-in around the JavaScript oracle, recording what it is sent (CPU runs)
class OracleWorker {
    constructor() { this.onmessage = null; this.sent = [] }
    postMessage(message) {
        if (!(message && message.buffer)) return
        this.sent.push(message)
        const reply = O.render(message)
        Promise.resolve().then(() => this.onmessage({ data: reply }))
    }
    terminate() {}
}
// HipWorker with the same recording
class RecordingHipWorker extends HipWorker {
    constructor() { super(); this.sent = [] }
    postMessage(message, transfer) { if (message && message.buffer) this.sent.push(message); super.postMessage(message, transfer) }
}

async function main() {
    // 1. the members startWorkers touches on a worker instance exist on HipWorker with the same roles
    for (const run of K.runs) {
        assert.deepStrictEqual(run.members_read, ['postMessage'])
        assert.deepStrictEqual(run.members_written, ['onmessage'])
        assert.deepStrictEqual(run.probe, [{ keys: ['transferable'], transfer: 1 }])
    }
    assert.strictEqual(typeof HipWorker.prototype.postMessage, 'function')
    assert.strictEqual(HipWorker.prototype.postMessage.length, 2)
    let checked = 0
    for (const run of K.runs) {
        for (const e of run.cases) {
            const c = spec.cases.find(x => x.name === e.name)
            const sw = siggen.SAMPLE_WIDTH[c.format]
            const input = siggen.generate(c.format, c.gen, Math.ceil(c.bytes / sw), 0).slice(0, c.bytes)
            const Cls = kind === 'hip' ? RecordingHipWorker : OracleWorker
            const pool = Array.from({ length: run.workers }, () => new Cls())
            const w = windows[c.window + 'Window'](c.n)
            const m = await renderSliced({ buffer: input.buffer.slice(input.byteOffset, input.byteOffset + input.byteLength), format: c.format, n: c.n,
                width: e.width, workers: run.workers, window: w, cmap: cmaps[c.cmap + '_cmap'], gain: c.gain, range: c.range,
                channelMode: c.channelMode, waterfall: c.waterfall }, pool)
            // canvas size and every message the reference posted, in worker order
            assert.deepStrictEqual([m.width, m.height], e.canvas.fft, e.name)
            const sent = pool.map(p => p.sent[0])
            assert.strictEqual(sent.length, e.messages.length)
            sent.forEach((s, i) => {
                const r = e.messages[i]
                assert.deepStrictEqual({ block_norm: f64hex(s.block_norm), gain: s.gain, range: s.range, n: s.n, width: s.width, offset: s.offset,
                    format: s.format, channelMode: s.channelMode, waterfall: s.waterfall, buffer_bytes: s.buffer.byteLength, cmap_len: s.cmap.length,
                    cmap_first: s.cmap[0], cmap_last: s.cmap[s.cmap.length - 1],
                    windowc_sha256: require('crypto').createHash('sha256').update(f64bytes(s.windowc)).digest('hex') },
                { block_norm: r.block_norm, gain: r.gain, range: r.range, n: r.n, width: r.width, offset: r.offset, format: r.format,
                    channelMode: r.channelMode, waterfall: r.waterfall, buffer_bytes: r.buffer_bytes, cmap_len: r.cmap_len, cmap_first: r.cmap_first,
                    cmap_last: r.cmap_last, windowc_sha256: r.windowc_sha256 }, `${e.name} message ${i}`)
            })
            // strips: where each goes and what it holds
            const placed = m.replies.map(r => ['putImageData', ...stripPlacement(r, { width: e.width, sliceWidth: m.sliceWidth, n: c.n, waterfall: c.waterfall })
                .slice(0, 2), ...stripPlacement(r, { width: e.width, sliceWidth: m.sliceWidth, n: c.n, waterfall: c.waterfall }).slice(2), G.sha256(r.imageData.data)])
            assert.deepStrictEqual(placed, e.fft_calls, e.name + ' strips')
            // gauge strips, call for call
            const mm = [], amp = []
            for (const r of m.replies) { const g = gaugeColumns(r, m.sliceWidth, c.minmaxHeight, c.ampHeight); mm.push(...g.minmax); amp.push(...g.amp) }
            assert.deepStrictEqual(mm, e.minmax_calls, e.name + ' minmax gauge')
            assert.deepStrictEqual(amp, e.amp_calls, e.name + ' amp gauge')
            // merged side outputs
            assert.deepStrictEqual(m.c_hist, e.c_hist, e.name)
            const cB = {}
            m.cB_hist.forEach((v, i) => { if (v) cB[i] = v })
            assert.deepStrictEqual(cB, e.cB_hist, e.name)
            assert.strictEqual(f64hex(m.dBfs_min), e.dBfs_min, e.name)
            assert.strictEqual(f64hex(m.dBfs_max), e.dBfs_max, e.name)
            pool.forEach(p => p.terminate())
            checked++
        }
    }
    console.log(`caller checks ok (${kind}): ${checked} processData runs of the reference reproduced`)
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
