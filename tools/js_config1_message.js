'use strict'
/**
 * One config-1 worker message (1 MSample cu8, n = 512: 2 MiB in, 4 MiB out) through HipWorker: per-message time after a short and after a
 * long warm-up (the reply-image pool is steady once V8 has collected enough replies for every new one to find a recycled, page-locked
 * block), and the same with small requests rendered on the calling thread (tried in round 4, not kept).
 *   node tools/js_config1_message.js <repo root>
 */
const path = require('path')
const root = process.argv[2]
const O = require(path.join(root, 'oracle', 'js', 'worker_oracle.js'))
const { HipWorker } = require(path.join(root, 'spectroplot-js_amd', 'js'))
const addon = require(path.join(root, 'spectroplot-js_amd', 'lib', 'spectroplot_hip.node'))
const n = 512, width = 2048
const { window: windowc, weight } = O.makeWindow('hann', n)
const cmap = Array.from({ length: 256 }, (_, i) => [i, 255 - i, i])
const samples = new Uint8Array(2 * n * width)
for (let i = 0; i < samples.length; i++) samples[i] = (i * 2654435761) >>> 24
const msg = () => ({ block_norm: 1 / weight, gain: 6, range: 30, cmap, n, windowc, width, offset: 0, buffer: samples.buffer, format: 'CU8', channelMode: false, waterfall: false })
function ask(worker, m) { return new Promise((resolve, reject) => { worker.onmessage = e => resolve(e.data); worker.onerror = reject; worker.postMessage(m, []) }) }
async function run(label, sync, warm, reps) {
    const w = new HipWorker()
    if (sync === true) w._render = function (m) {      // the variant that was tried: small requests on the calling thread
        return new Promise((resolve, reject) => { try { resolve(this._wrap(m, addon.renderSync(this._ctx, this._request(m)))) } catch (e) { reject(e) } })
    }
    if (sync === 'turn') w._render = function (m) {    // ... the same, but the reply is handed over in a new turn of the event loop
        return new Promise((resolve, reject) => {
            let r
            try { r = this._wrap(m, addon.renderSync(this._ctx, this._request(m))) } catch (e) { reject(e); return }
            setImmediate(() => resolve(r))
        })
    }
    for (let i = 0; i < warm; i++) await ask(w, msg())
    const s0 = addon.poolStats()
    const t0 = process.hrtime.bigint()
    for (let i = 0; i < reps; i++) await ask(w, msg())
    const ms = Number(process.hrtime.bigint() - t0) / 1e6 / reps
    const s1 = addon.poolStats()
    console.log(`${label}: warm ${warm}, ${reps} messages: ${ms.toFixed(3)} ms per message; fresh blocks in the timed part ${s1.fresh - s0.fresh}, recycled ${s1.recycled - s0.recycled} (pinned ${s1.recycledPinned - s0.recycledPinned})`)
    w.terminate()
}
async function main() {
    await run('async, short warm-up', false, 12, 24)
    await run('sync,  short warm-up', true, 12, 24)
    await run('async, long warm-up', false, 200, 200)
    await run('sync,  long warm-up', true, 200, 200)
    await run('sync + turn, short warm-up', 'turn', 12, 24)
    await run('sync + turn, long warm-up', 'turn', 200, 200)
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
