'use strict'
/** CPU-only: the N-API addon loads, its host helpers agree with the reference KATs, and HipWorker fails loudly without a GPU. */
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const assert = require('assert')
const G = require('./golden_util.js')
const addon = require('../../spectroplot-js_amd/lib/spectroplot_hip.node')
const { HipWorker, packLut } = require('../../spectroplot-js_amd/js')

for (const k of ['deviceCount', 'parseFormat', 'window', 'sliceBounds', 'createContext', 'destroyContext', 'cmap', 'cmapKeys', 'render', 'renderSync']) assert.strictEqual(typeof addon[k], 'function', k)
assert.ok(addon.version >= 100)
// format table, lib/samples.js:22-162
for (const [name, sw] of [['cu8', 2], ['CFILE', 8], ['cs12', 3], ['nonsense', 2], ['CS64', 16], ['complex16s', 2]]) assert.strictEqual(addon.parseFormat(name).sampleWidth, sw, name)
// windows, lib/windows.js (golden KAT)
const idx = JSON.parse(fs.readFileSync(path.join(G.gdir, 'windows.json'), 'utf8'))
for (const e of idx) {
    const w = addon.window(e.name, e.n)
    assert.strictEqual(crypto.createHash('sha256').update(Buffer.from(w.window.buffer)).digest('hex'), e.sha256, `${e.name} ${e.n}`)
    assert.ok(G.sameF64(w.weight, e.weight))
}
assert.throws(() => addon.window('kaiser', 8))
// colour maps by name (sp_cmap): the reference's keys in its order, its lookup rules, its evaluated tables
assert.strictEqual(addon.cmapKeys().length, 14)
assert.strictEqual(addon.cmap('nosuch'), null)
assert.strictEqual(addon.cmap('VIRIDIS').length, 3 * 256)
assert.strictEqual(addon.cmap('para').length, 3 * 64)
// slice bounds, lib/samples.js:253-258
for (const c of G.spec.worker_cases) {
    const e = G.expected.find(x => x.name === c.name)
    if (!e.slices) continue
    const sw = addon.parseFormat(c.format).sampleWidth
    e.slices.forEach((s, i) => { const [b0, b1] = addon.sliceBounds(c.bytes, sw, i, c.slices); assert.strictEqual(b1 - b0, s.slice_bytes) })
}
// Uint8ClampedArray store semantics for non-integer LUT entries
assert.deepStrictEqual(Array.from(packLut([[0.5, 1.5, 2.5], [254.5, 255.5, -3], [NaN, 300, 7.49]])), [0, 2, 2, 254, 255, 0, 0, 255, 7])
if (addon.deviceCount() === 0) {
    assert.throws(() => new HipWorker(), /no HIP device/)
    assert.throws(() => addon.createContext(0), /no HIP device/)
}
console.log('addon cpu checks ok')
