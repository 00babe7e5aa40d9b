'use strict'
/**
 * Where a config-1 worker message spends its time in HipWorker: request marshalling (_request), the native render (addon.render ->
 * callback: the C ABI's sp_render plus two thread hand-offs), reply wrapping (_wrap) and the promise plumbing around them.
 *   node tools/js_message_stages.js <repo root>
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
const now = () => Number(process.hrtime.bigint()) / 1e3
async function main() {
    const w = new HipWorker()
    const T = { request: 0, native: 0, wrap: 0, total: 0, post: 0 }
    let count = 0
    const orig_request = w._request.bind(w), orig_wrap = w._wrap.bind(w)
    let t_req0 = 0, t_req1 = 0, t_cb = 0
    w._request = m => { t_req0 = now(); const r = orig_request(m); t_req1 = now(); return r }
    w._wrap = (m, r) => { t_cb = now(); const x = orig_wrap(m, r); const t = now(); T.request += t_req1 - t_req0; T.native += t_cb - t_req1; T.wrap += t - t_cb; return x }
    const ask = m => new Promise((resolve, reject) => { w.onmessage = e => resolve(e.data); w.onerror = reject; w.postMessage(m, []) })
    for (let i = 0; i < 300; i++) await ask(msg())
    T.request = T.native = T.wrap = 0
    const reps = 300
    for (let i = 0; i < reps; i++) {
        const t0 = now()
        await ask(msg())
        T.total += now() - t0
        T.post += t_req0 - t0
    }
    console.log(`per message (us): total ${(T.total / reps).toFixed(1)}, postMessage->_request ${(T.post / reps).toFixed(1)}, _request ${(T.request / reps).toFixed(1)}, addon.render -> callback ${(T.native / reps).toFixed(1)}, _wrap ${(T.wrap / reps).toFixed(1)}, rest (wrap -> onmessage -> await) ${((T.total - T.post - T.request - T.native - T.wrap) / reps).toFixed(1)}`)
    console.log('windowc is', windowc.constructor.name, 'pool', JSON.stringify(addon.poolStats()))
    w.terminate()
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
