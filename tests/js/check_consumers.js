'use strict'
// js/consumers.js against the reference's drawColorRamp / drawHistograms run on a recording context (tests/golden/consumers.json)
const fs = require('fs'), path = require('path'), assert = require('assert')
const root = path.join(__dirname, '..', '..')
const C = require(path.join(root, 'spectroplot-js_amd', 'js', 'consumers.js'))
const siggen = require(path.join(root, 'oracle', 'js', 'siggen.js'))
const gold = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'consumers.json')))
const spec = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'cases.json'))).consumer_kat.cases
const cmeta = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'cmaps.json')))
const cbin = fs.readFileSync(path.join(root, 'tests', 'golden', 'cmaps.bin'))
function cmapByName(name) {
    const m = cmeta.find(e => e.name === name || e.name + '_cmap' === name || e.name === name.replace(/_cmap$/, ''))
    assert(m, 'colour map ' + name)
    const out = []
    for (let i = 0; i < m.length; i++) out.push([cbin[m.offset + 3 * i], cbin[m.offset + 3 * i + 1], cbin[m.offset + 3 * i + 2]])
    return out
}
let n = 0
for (const c of spec) {
    const g = gold.find(e => e.name === c.name)
    const cmap = cmapByName(c.cmap)
    const opts = { rampWidth: c.rampWidth, rampTop: c.rampTop, histLeft: c.histLeft }
    const r = C.rampMarkers({ gain: c.gain, range: c.range, height: c.height, cmap, histWidth: c.histWidth, opts })
    assert.deepStrictEqual(r.canvas, g.canvas, c.name + ' canvas size')
    assert.strictEqual(r.calls.length, g.ramp.length, c.name + ' ramp call count')
    for (let i = 0; i < r.calls.length; i++) {
        const mine = r.calls[i].slice(), ref = g.ramp[i]
        if (mine[0] === 'putImageData') mine[5] = Buffer.from(mine[5].data.buffer).toString('base64')
        assert.deepStrictEqual(mine, ref, `${c.name} ramp call ${i}`)
        n++
    }
    const c_hist = new Array(cmap.length), cB_hist = new Array(1000)
    for (let i = 0; i < cmap.length; i++) c_hist[i] = siggen.hash(c.seed, i) % 5000
    for (let i = 0; i < 1000; i++) cB_hist[i] = siggen.hash(c.seed ^ 0x55, i) % 70000
    const h = C.histogramOutlines({ c_hist, cB_hist, cmapLength: cmap.length, height: c.height, histWidth: c.histWidth, opts })
    assert.deepStrictEqual(h, g.hist, c.name + ' histogram outlines')
    n += h.length
}
// gaugeColumns: shape and arithmetic on a hand-made reply
const rep = { offset: 7, gauge_mins: Uint8ClampedArray.from([0, 10, 200]), gauge_maxs: Uint8ClampedArray.from([5, 250, 255]), gauge_amps: Uint8ClampedArray.from([1, 128, 255]) }
const gc = C.gaugeColumns(rep, 3, 40, 20)
assert.deepStrictEqual(gc.minmax[1], ['fillRect', 8, ~~(10 * 40 / 256), 1, ~~(240 * 40 / 256), 'rgb(5,5,5)'])
assert.deepStrictEqual(gc.amp[2], ['fillRect', 9, 0, 1, ~~(255 * 20 / 256), 'rgb(0,0,0)'])
console.log(`consumers checks ok (${n} reference draw calls)`)
