'use strict'
/** CPU-only: the caller-side parameter helpers (SURVEY §8f rows f2/f3) against vectors produced by the reference's own modules. */
const fs = require('fs')
const path = require('path')
const assert = require('assert')
const G = require('./golden_util.js')
const P = require('../../spectroplot-js_amd/js/params.js')

const K = JSON.parse(fs.readFileSync(path.join(G.gdir, 'parse.json'), 'utf8'))
for (const e of K.parsed) {
    assert.strictEqual(P.parseFormat(e.name), e.format, 'parseFormat ' + e.name)
    assert.deepStrictEqual(P.parseFreqRate(e.name), e.fr, 'parseFreqRate ' + e.name)
}
assert.deepStrictEqual(Object.keys(P.windows), K.window_key_order)
assert.deepStrictEqual(P.CMAP_KEY_ORDER, K.cmap_key_order)
const wkeys = {}
for (const k of Object.keys(P.windows)) wkeys[k] = k
for (const e of K.lookups) assert.strictEqual(P.lookup(wkeys, e.key), e.hit, 'window lookup ' + e.key)
const ckeys = {}
for (const k of P.CMAP_KEY_ORDER) ckeys[k] = k
for (const e of K.clookups) assert.strictEqual(P.lookup(ckeys, e.key), e.hit, 'cmap lookup ' + e.key)
assert.strictEqual(P.lookup(wkeys, null), null)
assert.deepStrictEqual(P.lookup(wkeys, [1, 2]), [1, 2])

// every colour map of the reference, by key and through the reference's lookup rules, equals the reference's evaluated modules
const idx = JSON.parse(fs.readFileSync(path.join(G.gdir, 'cmaps.json'), 'utf8'))
const bin = fs.readFileSync(path.join(G.gdir, 'cmaps.bin'))
assert.deepStrictEqual(Object.keys(P.cmaps), K.cmap_key_order)
for (const e of idx) {
    const lut = P.cmaps[e.name]
    assert.strictEqual(lut.length, e.length, e.name)
    const flat = Buffer.from(Uint8Array.from([].concat(...lut)))
    assert.strictEqual(Buffer.compare(flat, bin.slice(e.offset, e.offset + 3 * e.length)), 0, e.name)
}
for (const e of K.clookups) assert.strictEqual(P.lookup(P.cmaps, e.key), e.hit ? P.cmaps[e.hit] : null, 'cmap table lookup ' + e.key)
assert.strictEqual(P.cmapByName('viridis'), P.cmaps.viridis_cmap)
assert.strictEqual(P.cmapByName('nosuch'), P.cmaps.cube1_cmap)
// named windows through the native library equal lib/windows.js
const widx = JSON.parse(fs.readFileSync(path.join(G.gdir, 'windows.json'), 'utf8'))
for (const e of widx.filter(x => x.n <= 1024)) {
    const w = P.windows[e.name + 'Window'](e.n)
    assert.ok(G.sameF64(w.weight, e.weight), e.name)
    assert.strictEqual(w.window.length, e.n)
}
assert.strictEqual(P.windowByName('nosuch'), P.windows.blackmanHarrisWindow)
assert.strictEqual(P.windowByName('hann'), P.windows.hannWindow)
// ES-module key order: the prefix 'blackman' meets blackmanHarrisWindow first (same in the reference run under Node)
assert.strictEqual(P.windowByName('blackman'), P.windows.blackmanHarrisWindow)
console.log('params checks ok')
