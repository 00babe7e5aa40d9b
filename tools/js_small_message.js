'use strict'
const path = require('path')
const root = process.argv[2]
const O = require(path.join(root, 'oracle/js/worker_oracle.js'))
const { HipWorker } = require(path.join(root, 'spectroplot-js_amd/js'))
const addon = require(path.join(root, 'spectroplot-js_amd/lib/spectroplot_hip.node'))
const n = 512, width = 2048
const { window: windowc, weight } = O.makeWindow('hann', n)
const cmap = Array.from({ length: 256 }, (_, i) => [i, 255 - i, i])
const samples = new Uint8Array(2 * n * width)
for (let i = 0; i < samples.length; i++) samples[i] = (i * 2654435761) >>> 24
const w = new HipWorker()
const msg = () => ({ block_norm: 1 / weight, gain: 6, range: 30, cmap, n, windowc, width, offset: 0, buffer: samples.buffer, format: 'CU8', channelMode: false, waterfall: false })
function hr() { const t = process.hrtime(); return t[0] * 1e3 + t[1] / 1e6 }
async function main() {
    let tReq = 0, tRender = 0, tWrap = 0, tSync = 0
    const N = 300
    for (let k = 0; k < N + 20; k++) {
        const m = msg()
        const t0 = hr()
        const req = w._request(m)
        const t1 = hr()
        const r = await new Promise((res, rej) => addon.render(w._ctx, req, (e, r) => e ? rej(e) : res(r)))
        const t2 = hr()
        const rep = w._wrap(m, r)
        const t3 = hr()
        if (k >= 20) { tReq += t1 - t0; tRender += t2 - t1; tWrap += t3 - t2 }
        if (rep.c_hist.length !== 256) throw new Error('x')
    }
    // the synchronous entry point, if any
    if (addon.renderSync) {
        for (let k = 0; k < N + 20; k++) {
            const req = w._request(msg())
            const t1 = hr()
            addon.renderSync(w._ctx, req)
            if (k >= 20) tSync += hr() - t1
        }
    }
    console.log(`request ${(tReq / N).toFixed(3)} ms, render (async round trip) ${(tRender / N).toFixed(3)} ms, wrap ${(tWrap / N).toFixed(3)} ms, renderSync ${(tSync / N).toFixed(3)} ms`)
    w.terminate()
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
