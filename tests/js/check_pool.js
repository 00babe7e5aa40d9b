'use strict'
/**
 * GPU: the reply-image pool of the addon keeps to its limits.  Run with --expose-gc.  argv[2] = 'nopin' expects no page-locked
 * block at all (SPECTROPLOT_HIP_POOL_PINNED_MB=0 in the environment), anything else expects the page-locked bytes to stay at or
 * below the limit poolStats() reports while replies are dropped, collected and their blocks recycled.
 */
const assert = require('assert')
const path = require('path')
const O = require('../../oracle/js/worker_oracle.js')   // only for makeWindow
const { HipWorker } = require('../../spectroplot-js_amd/js')
const addon = require(path.join(__dirname, '..', '..', 'spectroplot-js_amd', 'lib', 'spectroplot_hip.node'))

function ask(worker, message) {
    return new Promise((resolve, reject) => {
        worker.onmessage = (e) => resolve(e.data)
        worker.onerror = (e) => reject(e)
        worker.postMessage(message, [message.buffer])
    })
}

async function main() {
    const mode = process.argv[2] || 'default'
    const n = 1024, width = 2048                      // 8 MiB of RGBA per reply: above the 1 MiB page-lock threshold
    const { window: windowc, weight } = O.makeWindow('blackmanHarris', n)
    const cmap = Array.from({ length: 256 }, (_, i) => [i, 255 - i, i])
    const samples = new Uint8Array(2 * n * width)
    for (let i = 0; i < samples.length; i++) samples[i] = (i * 2654435761) >>> 24
    const worker = new HipWorker()
    const limits = addon.poolStats()
    if (mode === 'nopin') assert.strictEqual(limits.pinnedLimit, 0)
    let first = null
    for (let k = 0; k < 12; k++) {
        let reply = await ask(worker, { block_norm: 1.0 / weight, gain: 6, range: 30, cmap, n, windowc, width, offset: 0,
            buffer: samples.buffer.slice(0), format: 'CU8', channelMode: false, waterfall: false })
        const sum = reply.imageData.data.reduce((a, b) => a + b, 0)
        if (first === null) first = sum
        assert.strictEqual(sum, first, 'a recycled block must hold the same image')
        reply = null
        global.gc()
        await new Promise(r => setImmediate(r))      // finalizers of collected ArrayBuffers run after the collection
        const s = addon.poolStats()
        assert(s.pinnedBytes <= s.pinnedLimit, `pinned ${s.pinnedBytes} > limit ${s.pinnedLimit}`)
        assert(s.freeBytes <= s.keepLimit)
    }
    const s = addon.poolStats()
    assert(s.recycled > 0, 'dropped replies are recycled')
    if (mode === 'nopin') {
        assert.strictEqual(s.pinnedBytes, 0)
        assert.strictEqual(s.recycledPinned, 0)
    } else if (mode === 'tight') {
        assert(s.pinnedBytes <= 8 << 20, 'one 8 MiB block at most under an 8 MiB limit')
    } else {
        assert(s.pinnedBytes > 0, 'a block that came round again is page-locked')
    }
    worker.terminate()
    console.log(`pool checks ok (${mode}): ${JSON.stringify(s)}`)
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
