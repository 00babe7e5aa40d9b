'use strict'
/**
 * js/raster.js (CPU): the software surface against an independent per-pixel model, on the reference's own draw calls:
 *   - the gauge fillRect calls the reference's processData made on recording canvases (tests/golden/caller.json), replayed on a Surface,
 *     against "the last rectangle that covers a pixel wins" evaluated pixel by pixel;
 *   - the colour ramp image and tick rectangles of the reference's drawColorRamp (tests/golden/consumers.json): the ramp pixels are the
 *     recorded image's, integer-aligned ticks are theme.rampFill; text commands come back as skipped;
 *   - the two histogram outlines of the reference's drawHistograms (tests/golden/consumers.json `hist`: its own moveTo / lineTo / fill /
 *     stroke calls): every pixel that lies wholly inside an outline carries that outline's fill, every pixel wholly outside both and
 *     clear of the strokes is untouched - decided by an independent model (the outline is a function x(y), piecewise linear between
 *     rows), not by the scanline code; pixels an edge or a stroke passes through are a browser's anti-aliasing and are not pinned;
 *   - translucent fills blend source-over; composePlot puts strips, gauges, scale and histograms where its layout says.
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

// histogram outlines: the reference's recorded path calls on a Surface, against the per-pixel model
let inside = 0, outside = 0
for (const g of cons) {
    const s = new Surface(g.canvas.width, g.canvas.height, '#fff')
    assert.deepStrictEqual(s.replay(g.hist), [], g.name + ': every outline call is drawn')
    // split the call list into its outlines: [vertices, fill style, stroke style, line width]
    const shapes = []
    let cur = null
    for (const c of g.hist) {
        if (c[0] === 'beginPath') cur = { pts: [] }
        else if (c[0] === 'moveTo' || c[0] === 'lineTo') cur.pts.push([c[1], c[2]])
        else if (c[0] === 'fill') cur.fill = c[1]
        else if (c[0] === 'stroke') { cur.stroke = c[1]; cur.line = c[2]; shapes.push(cur) }
    }
    if (!g.hist.length) continue                                                // histWidth 0: the reference draws none
    assert.strictEqual(shapes.length, 2, g.name + ': two outlines')
    // an outline runs from (left, top) along (left + x_r, top + r), r = 0 .. rows-1, to (left, top + rows) and closes: between two rows its
    // right edge is the segment x_r .. x_r+1, its left edge the line x = left
    const model = shapes.map(sh => {
        const left = sh.pts[0][0], top = sh.pts[0][1], xs = sh.pts.slice(1, -1).map(p => p[0]), rows = xs.length
        assert.strictEqual(sh.pts[sh.pts.length - 1][1], top + rows)
        const xAt = y => {                                                      // right edge at height y (top <= y <= top + rows)
            const r = y - top
            if (r >= rows - 1) return r >= rows ? left : xs[rows - 1] + (left - xs[rows - 1]) * (r - (rows - 1))   // the closing segment to (left, top + rows)
            const i = Math.floor(r)
            return xs[i] + (xs[i + 1] - xs[i]) * (r - i)
        }
        return { left, top, rows, xAt, half: sh.line / 2, fill: parseColor(sh.fill) }
    })
    const blend = (dst, c) => { const a = c[3] / 255; return dst.map((v, k) => k < 3 ? Math.round(c[k] * a + v * (1 - a)) : 255) }
    for (let y = 0; y < s.height; y++) {
        for (let x = 0; x < s.width; x++) {
            let want = [255, 255, 255, 255], decided = true
            for (const m of model) {
                if (y < m.top - 2 || y >= m.top + m.rows + 2) continue                            // clear of this outline, strokes included
                // the right edge over this pixel row and a stroke's reach either side of it (piecewise linear: extremes at row boundaries)
                const margin = m.half + 1
                let lo = Infinity, hi = -Infinity
                for (let yy = Math.floor(y - margin); yy <= Math.ceil(y + 1 + margin); yy++) {
                    const xe = m.xAt(Math.max(m.top, Math.min(m.top + m.rows, yy)))
                    lo = Math.min(lo, xe); hi = Math.max(hi, xe)
                }
                if (y >= m.top + margin && y + 1 <= m.top + m.rows - margin && x >= m.left + margin && x + 1 <= lo - margin) want = blend(want, m.fill)   // wholly inside, clear of the stroke
                else if (x >= hi + margin || x + 1 <= m.left - margin) continue                  // wholly outside
                else decided = false
            }
            if (!decided) continue
            const o = 4 * (y * s.width + x)
            assert.deepStrictEqual(Array.from(s.data.subarray(o, o + 4)), want, `${g.name}: pixel (${x}, ${y})`)
            if (want[0] !== 255 || want[1] !== 255) inside++
            else outside++
        }
    }
}
assert(inside > 5000 && outside > 50000, `histogram pixels checked: ${inside} inside, ${outside} outside`)

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
assert(plot.skipped.length > 0 && plot.skipped.every(c => c[0] === 'fillText'), 'composePlot draws everything but text')
// flat histograms: both outlines span the full histogram width, so a pixel well inside carries both translucent fills over the page white
{
    const px = plot.origin.scale[0] + 55 + 50, py = plot.origin.scale[1] + 10 + 4, o = 4 * (py * S.width + px)
    assert.deepStrictEqual(Array.from(S.data.subarray(o, o + 4)), [223, 194, 223, 255], 'histogram fills in the composed plot')
}
const ppm = S.toPPM()
assert.strictEqual(ppm.slice(0, 2).toString(), 'P6')
console.log(`raster checks ok (${rects} gauge rectangles of the reference replayed pixel for pixel, ${ticks} ticks, ${inside} + ${outside} histogram pixels)`)
