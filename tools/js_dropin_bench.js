'use strict'
/**
 * What the drop-in buys a Node caller: round-trip time of one worker message (postMessage -> onmessage, host buffers, PCIe
 * both ways) through HipWorker versus the same message through the JavaScript restatement of the reference worker
 * (oracle/js/worker_oracle.js, bit-identical to lib/worker.js) on this host.  Replies are compared on the way.
 *   node tools/js_dropin_bench.js            (run on a machine with a HIP device)
 */
const path = require('path')
const root = path.join(__dirname, '..')
const O = require(path.join(root, 'oracle', 'js', 'worker_oracle.js'))
const siggen = require(path.join(root, 'oracle', 'js', 'siggen.js'))
const { HipWorker } = require(path.join(root, 'spectroplot-js_amd', 'js'))

const GEN = { kind: 'trinoise', seed: 0x5EED0001, step: 7321, gshift: 11, amp: 0.5, namp: 0.02 }
function message(format, S, n, windowName, pinned, width) {
    let bytes = siggen.generate(format, GEN, S, 0)
    if (pinned) {   // the same samples in a page-locked ArrayBuffer (HipWorker.allocBuffer), as js/render_file.js cuts its slices
        const p = new Uint8Array(HipWorker.allocBuffer(bytes.byteLength))
        p.set(new Uint8Array(bytes.buffer, bytes.byteOffset, bytes.byteLength))
        bytes = p
    }
    const { window: windowc, weight } = O.makeWindow(windowName, n)
    const cmap = []
    for (let i = 0; i < 256; i++) cmap.push([i, 255 - i, (i * 3) & 255])
    cmap[0] = [0, 0, 0]; cmap[255] = [255, 255, 255]
    return { block_norm: 1.0 / weight, gain: 6, range: 30, cmap, n, windowc, width: width || S / n, offset: 0, buffer: bytes.buffer, format,
        channelMode: false, waterfall: false }
}
function ask(worker, m) {
    return new Promise((resolve, reject) => {
        worker.onmessage = e => resolve(e.data)
        worker.onerror = reject
        worker.postMessage(m, [])
    })
}
async function main() {
    const worker = new HipWorker()
    const json = process.argv.includes('--json')
    const rows = []
    // (the last two: config 2's capture at a screen-wide image, 2048 frames ~ 8 n apart - the reference's interactive shape
    // (lib/worker.js:50, 70-75): sp_render uploads the frames' own samples only; with SPECTROPLOT_HIP_NO_PACKED_UPLOAD=1 the whole capture)
    for (const [name, format, log2s, n, win, pinned, width] of [['config 1', 'CU8', 20, 512, 'hann', false], ['config 2 / 4', 'CF32', 22, 1024, 'blackmanHarris', false],
        ['config 2', 'CF32', 24, 1024, 'blackmanHarris', false], ['config 2, request buffer page-locked', 'CF32', 24, 1024, 'blackmanHarris', true],
        ['sparse', 'CF32', 24, 1024, 'blackmanHarris', false, 2048], ['sparse, request buffer page-locked', 'CF32', 24, 1024, 'blackmanHarris', true, 2048]]) {
        const m = message(format, 2 ** log2s, n, win, pinned, width)
        // warm-up: plan creation, and the pool of reply images (a block is recycled once V8 has collected the reply that held it, and
        // page-locked when it comes round; a long-running viewer re-rendering at one size is in that state)
        const t_cold = process.hrtime.bigint()
        await ask(worker, m)
        const cold_ms = Number(process.hrtime.bigint() - t_cold) / 1e6
        // ... until the pool is steady: no fresh block for 24 messages in a row (small replies weigh little in V8's books, so it collects
        // them rarely and the pool needs a few dozen blocks before every reply finds a recycled one), at most 400 messages
        const addon = require(path.join(root, 'spectroplot-js_amd', 'lib', 'spectroplot_hip.node'))
        let warm = 1, calm = 0, fresh = addon.poolStats().fresh
        while (warm < 12 || (calm < 24 && warm < 400)) {
            await ask(worker, m)
            warm++
            const f = addon.poolStats().fresh
            calm = f === fresh ? calm + 1 : 0
            fresh = f
        }
        // mean AND median over enough messages that V8's collector (a pause every ~10th large reply) weighs what it weighs
        const reps = log2s >= 24 ? 96 : 48
        let t0 = process.hrtime.bigint()
        let reply
        const each = []
        for (let i = 0; i < reps; i++) { const t1 = process.hrtime.bigint(); reply = await ask(worker, m); each.push(Number(process.hrtime.bigint() - t1) / 1e6) }
        const gpu_ms = Number(process.hrtime.bigint() - t0) / 1e6 / reps
        each.sort((a, b) => a - b)
        const median_ms = each[each.length >> 1]
        let cpu_ms = null, same = null
        if (log2s <= 22 || width) {
            O.render(m)
            t0 = process.hrtime.bigint()
            const ref = O.render(m)
            cpu_ms = Number(process.hrtime.bigint() - t0) / 1e6
            same = Buffer.compare(Buffer.from(ref.imageData.data.buffer), Buffer.from(reply.imageData.data.buffer)) === 0
        }
        rows.push({ name, format, samples: 2 ** log2s, n, request_buffer: pinned ? 'page-locked' : 'pageable', ms_per_message: gpu_ms, median_ms_per_message: median_ms, messages: reps, first_message_ms: cold_ms,
            warmup_messages: warm,
            msamples_per_s: 2 ** log2s / gpu_ms / 1e3, js_worker_ms: cpu_ms, images_identical: same })
        if (!json) console.log(`${name}: ${format} 2^${log2s} samples, n=${n}: HipWorker ${gpu_ms.toFixed(2)} ms per message` +
            (cpu_ms ? `, JS worker ${cpu_ms.toFixed(0)} ms (x${(cpu_ms / gpu_ms).toFixed(0)}), images identical: ${same}` : ''))
    }
    const stats = require(path.join(root, 'spectroplot-js_amd', 'lib', 'spectroplot_hip.node')).poolStats()
    if (!json) console.log('reply-image pool:', JSON.stringify(stats))
    if (json) console.log(JSON.stringify(rows))
    worker.terminate()
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
