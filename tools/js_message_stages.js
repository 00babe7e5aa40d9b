'use strict'
/**
 * Where a config-1 (or, with `cfg2`, a config-2) worker message spends its time in HipWorker: request marshalling (_request), the native render (addon.render ->
 * callback: the C ABI's sp_render plus two thread hand-offs), reply wrapping (_wrap) and the promise plumbing around them.
 *   node tools/js_message_stages.js <repo root> [cfg2] [pinned]
 */
const path = require('path')
const root = process.argv[2]
const O = require(path.join(root, 'oracle', 'js', 'worker_oracle.js'))
const { HipWorker } = require(path.join(root, 'spectroplot-js_amd', 'js'))
const addon = require(path.join(root, 'spectroplot-js_amd', 'lib', 'spectroplot_hip.node'))
// config 1 (default): 2^20 cu8 samples, n = 512, 2048 frames; `cfg2`: 2^24 cf32 samples, n = 1024, 16384 frames (128 MiB in, 64 MiB out);
// `pinned` as a further argument: the request buffer is page-locked (HipWorker.allocBuffer)
// `sparse`: config 2's capture at a screen-wide image (2048 frames: stride ~ 8 n, the reference's interactive shape)
const sparse = process.argv.includes('sparse'), big = process.argv.includes('cfg2') || sparse, pinned = process.argv.includes('pinned')
const n = big ? 1024 : 512, width = sparse ? 2048 : big ? 16384 : 2048, format = big ? 'CF32' : 'CU8', sw = big ? 8 : 2
const { window: windowc, weight } = O.makeWindow(big ? 'blackmanHarris' : 'hann', n)
const cmap = Array.from({ length: 256 }, (_, i) => [i, 255 - i, i])
const nbytes = sparse ? 8 << 24 : sw * n * width
const buffer = pinned ? HipWorker.allocBuffer(nbytes) : new ArrayBuffer(nbytes)
if (big) { const f = new Float32Array(buffer); for (let i = 0; i < f.length; i++) f[i] = ((i * 2654435761) >>> 8) / 16777216 - 0.5 }
else { const b = new Uint8Array(buffer); for (let i = 0; i < b.length; i++) b[i] = (i * 2654435761) >>> 24 }
const msg = () => ({ block_norm: 1 / weight, gain: 6, range: 30, cmap, n, windowc, width, offset: 0, buffer, format, channelMode: false, waterfall: false })
const now = () => Number(process.hrtime.bigint()) / 1e3
async function main() {
    const w = new HipWorker()
    const T = { request: 0, native: 0, wrap: 0, total: 0, post: 0, take: 0, lib: 0 }
    let count = 0
    const orig_request = w._request.bind(w), orig_wrap = w._wrap.bind(w)
    let t_req0 = 0, t_req1 = 0, t_cb = 0
    w._request = m => { t_req0 = now(); const r = orig_request(m); t_req1 = now(); return r }
    w._wrap = (m, r) => { t_cb = now(); const x = orig_wrap(m, r); const t = now(); T.request += t_req1 - t_req0; T.native += t_cb - t_req1; T.wrap += t - t_cb; T.take += r.stages.take_us; T.lib += r.stages.native_us; return x }
    const ask = m => new Promise((resolve, reject) => { w.onmessage = e => resolve(e.data); w.onerror = reject; w.postMessage(m, []) })
    for (let i = 0; i < (big ? 60 : 300); i++) await ask(msg())
    T.request = T.native = T.wrap = T.take = T.lib = 0
    const reps = big ? 80 : 300
    const worst = []
    for (let i = 0; i < reps; i++) {
        const t0 = now()
        await ask(msg())
        T.total += now() - t0
        worst.push(now() - t0)
        T.post += t_req0 - t0
    }
    worst.sort((a, b) => a - b)
    console.log(`per message (us): total ${(T.total / reps).toFixed(1)}, postMessage->_request ${(T.post / reps).toFixed(1)}, _request ${(T.request / reps).toFixed(1)}, addon.render -> callback ${(T.native / reps).toFixed(1)} (of it on the worker thread: reply buffers ${(T.take / reps).toFixed(1)}, library call ${(T.lib / reps).toFixed(1)}; the rest = two thread hand-offs + the reply object), _wrap ${(T.wrap / reps).toFixed(1)}, rest (wrap -> onmessage -> await) ${((T.total - T.post - T.request - T.native - T.wrap) / reps).toFixed(1)}`)
    console.log(`median ${worst[worst.length >> 1].toFixed(1)} us, fastest ${worst[0].toFixed(1)}, slowest ${worst[worst.length - 1].toFixed(1)}; ${sparse ? 'sparse (16 MSample cf32 at 2048 frames)' : big ? 'config 2' : 'config 1'} message, ${pinned ? 'page-locked' : 'pageable'} request buffer`)
    console.log('windowc is', windowc.constructor.name, 'pool', JSON.stringify(addon.poolStats()))
    w.terminate()
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
