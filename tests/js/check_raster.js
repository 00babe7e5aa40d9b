'use strict'
/**
 * js/raster.js (CPU): the software surface against an independent per-pixel model, on the reference's own draw calls:
 *   - the gauge fillRect calls the reference's processData made on recording canvases (tests/golden/caller.json), replayed on a Surface,
 *     against "the last rectangle that covers a pixel wins" evaluated pixel by pixel;
 *   - the colour ramp image and tick rectangles of the reference's drawColorRamp (tests/golden/consumers.json): the ramp pixels are the
 *     recorded image's, integer-aligned ticks are theme.rampFill; text commands come back as skipped;
 *   - translucent fills blend source-over; composePlot puts strips, gauges and scale where its layout says.
 */
const fs = require('fs'), path = require('path'), assert = require('assert')
const root = path.join(__dirname, '..', '..')
const { Surface, parseColor, composePlot } = require(path.join(root, 'spectroplot-js_amd', 'js', 'raster.js'))
const C = require(path.join(root, 'spectroplot-js_amd', 'js', 'consumers.js'))
const caller = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'caller.json')))
const cons = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'consumers.json')))

assert.deepStrictEqual(parseColor('#666'), [102, 102, 102, 255])
assert.deepStrictEqual(parseColor('#b0b'), [187, 0, 187, 255])
assert.deepStrictEqual(parseColor('rgb(5,5,5)'), [5, 5, 5, 255])
assert.deepStrictEqual(parseColor('rgba(187,0,187,0.2)'), [187, 0, 187, 51])

let rects = 0
for (const run of caller.runs) {
    for (const e of run.cases) {
        for (const [calls, key] of [[e.minmax_calls, 'minmax'], [e.amp_calls, 'amp']]) {
            if (!calls || !calls.length) continue
            const W = e.width, H = 1 + Math.max(...calls.map(c => c[2] + c[4]))
            const s = new Surface(W, H)
            assert.deepStrictEqual(s.replay(calls), [], e.name + ' ' + key + ': every gauge call is a fillRect')
            for (let y = 0; y < H; y++) {
                for (let x = 0; x < W; x++) {
                    let want = [0, 0, 0, 0]
                    for (const c of calls) if (x >= c[1] && x < c[1] + c[3] && y >= c[2] && y < c[2] + c[4]) want = parseColor(c[5])
                    const o = 4 * (y * W + x)
                    if (s.data[o] !== want[0] || s.data[o + 1] !== want[1] || s.data[o + 2] !== want[2] || s.data[o + 3] !== want[3])
                        assert.fail(`${e.name} ${key}: pixel (${x}, ${y})`)
                }
            }
            rects += calls.length
        }
    }
}
assert(rects > 1000, 'gauge rectangles replayed: ' + rects)

let ticks = 0
for (const g of cons) {
    const calls = g.ramp.map(c => c[0] === 'putImageData' ? [c[0], c[1], c[2], c[3], c[4], { data: new Uint8ClampedArray(Buffer.from(c[5], 'base64')), width: c[3], height: c[4] }] : c)
    const s = new Surface(g.canvas.width, g.canvas.height)
    const skipped = s.replay(calls)
    assert(skipped.length > 0 && skipped.every(c => c[0] === 'fillText'), g.name + ': only text is skipped')
    const img = calls[0]
    for (let y = 0; y < img[4]; y++) for (let x = 0; x < img[3]; x++) for (let k = 0; k < 4; k++)
        assert.strictEqual(s.data[4 * ((y + img[2]) * s.width + x + img[1]) + k], img[5].data[4 * (y * img[3] + x) + k], g.name + ' ramp pixel')
    for (const c of calls) {
        if (c[0] !== 'fillRect' || c[2] !== Math.floor(c[2])) continue          // ticks between pixel rows are anti-aliased: not pinned
        if (c[2] >= s.height) continue
        const o = 4 * (c[2] * s.width + c[1])
        assert.deepStrictEqual(Array.from(s.data.subarray(o, o + 4)), parseColor(c[5]), g.name + ' tick')
        ticks++
    }
}
assert(ticks > 10, 'ticks checked: ' + ticks)

// source-over blending of a translucent fill, 8-bit
const b = new Surface(2, 1, '#fff')
b.fillRect(0, 0, 1, 1, 'rgba(187,0,187,0.2)')
assert.deepStrictEqual(Array.from(b.data), [241, 204, 241, 255, 255, 255, 255, 255])

// composePlot: a 2-slice spectrogram result made by hand
const n = 8, sw = 3, width = 6
const mk = (v) => ({ offset: v * sw, gauge_mins: Uint8ClampedArray.from([0, 64, 128]), gauge_maxs: Uint8ClampedArray.from([128, 192, 255]),
    gauge_amps: Uint8ClampedArray.from([255, 128, 0]), imageData: { data: new Uint8ClampedArray(4 * sw * n).fill(100 + v) } })
const data = new Uint8ClampedArray(4 * width * n)
for (let p = 0; p < data.length; p += 4) data.set([p & 255, 7, 9, 255], p)
const cmap = []
for (let i = 0; i < 256; i++) cmap.push([i, 255 - i, 0])
const plot = composePlot({ data, width, height: n, sliceWidth: sw, replies: [mk(0), mk(1)], c_hist: new Array(256).fill(1), cB_hist: new Array(1000).fill(1) },
    { cmap, gain: 6, range: 30, n, ampHeight: 4, minmaxHeight: 8 })
const S = plot.surface
assert.deepStrictEqual(plot.origin, { image: [0, 12], scale: [6, 12] })
assert.strictEqual(S.width, width + 160)
for (let y = 0; y < n; y++) for (let x = 0; x < width; x++) assert.strictEqual(S.data[4 * ((y + 12) * S.width + x)], data[4 * (y * width + x)], 'image pixel')
assert.deepStrictEqual(Array.from(S.data.subarray(0, 4)), [127, 127, 127, 255])                    // amp gauge column 0: grey 255 - 128, full height ~~(255 * 4 / 256) = 3
assert.deepStrictEqual(Array.from(S.data.subarray(4 * 3 * S.width, 4 * 3 * S.width + 4)), [255, 255, 255, 255])   // row 3: past the bar, page white
assert(plot.skipped.some(c => c[0] === 'fillText') && plot.skipped.some(c => c[0] === 'lineTo'))
const ppm = S.toPPM()
assert.strictEqual(ppm.slice(0, 2).toString(), 'P6')
console.log(`raster checks ok (${rects} gauge rectangles of the reference replayed pixel for pixel, ${ticks} ticks)`)
