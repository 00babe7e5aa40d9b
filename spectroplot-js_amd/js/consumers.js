'use strict'
/**
 * Headless consumers of a reply's side outputs: the step after the worker in the reference's pipeline.
 * The reference draws these straight into canvases; here each function returns the same drawing as data — an RGBA image
 * for the colour ramp, draw-command lists (`['fillRect', x, y, w, h, fillStyle]`, `['lineTo', x, y]`, …) for markers,
 * histogram outlines and gauges — in the order and with the coordinates the reference hands to its 2-D context, so any
 * rasteriser (node-canvas, a PNG writer, a test) can replay them.
 *
 *   colorRamp / rampMarkers    lib/spectroplot.js:620-684  (drawColorRamp)
 *   histogramOutlines          lib/spectroplot.js:686-757  (drawHistograms)
 *   gaugeColumns               lib/spectroplot.js:1246-1268 (per-reply gauge strips inside processData)
 *
 * colorRamp, rampMarkers and histogramOutlines are pinned call-for-call against the reference's own methods run on a
 * recording context (tests/golden/consumers.json, oracle/ref_harness.mjs); gaugeColumns against the fillRect calls the reference's
 * processData itself makes on recording canvases when run with 1, 2 and 8 workers (tests/golden/caller.json).
 */

const DEFAULT_THEME = { rampFill: '#666', histoLine: 2, histoStroke: '#b0b', histoFill: 'rgba(187,0,187,0.2)', dbfsLine: 2,
    dbfsStroke: '#999', dbfsFill: 'rgba(153,153,153,0.2)' }            // lib/spectroplot.js:390-400 (light theme)
const DEFAULT_OPTS = { dbfsWidth: 60, timeHeight: 20, rampTop: 10, rampWidth: 15, histWidth: 100, histLeft: 55 }   // :266-274

/** The colour ramp image: row y shows LUT entry `len-1 - round(y*(len-1)/(height-1))`   (lib/spectroplot.js:646-660). */
function colorRamp(cmap, rampWidth, rampHeight) {
    const top = cmap.length - 1
    const out = new Uint8ClampedArray(4 * rampWidth * rampHeight)
    for (let row = 0; row < rampHeight; ++row) {
        const rgb = cmap[top - Math.round(row * top / (rampHeight - 1))]
        const base = 4 * rampWidth * row
        for (let col = 0; col < rampWidth; ++col) out.set([rgb[0], rgb[1], rgb[2], 255], base + 4 * col)
    }
    return { data: out, width: rampWidth, height: rampHeight }
}

/**
 * Everything drawColorRamp hands to the dbfs canvas: its size, the ramp image at (35, rampTop) and the tick / label commands
 * (lib/spectroplot.js:620-684).  `o`: {gain, range, height, cmap, histWidth, opts?, theme?}.
 */
function rampMarkers(o) {
    const opts = Object.assign({}, DEFAULT_OPTS, o.opts), theme = Object.assign({}, DEFAULT_THEME, o.theme)
    const histWidth = o.histWidth === undefined ? opts.histWidth : o.histWidth
    const lo = o.gain, span = o.range, hi = lo + span, rows = o.height
    const ramp = colorRamp(o.cmap, opts.rampWidth, rows)
    const calls = [['putImageData', 35, opts.rampTop, ramp.width, ramp.height, ramp]]
    // about one tick per 50 px, in steps that are multiples of 3 dB (at least 1 dB); the last tick lands on the scale's end
    let step = Math.round(hi / (rows / 50) / 3) * 3
    if (step < 1.0) step = 1.0
    const labelDrop = 10 / 2 - 1                                   // half the 10 px font, one pixel up
    for (let level = lo; level < hi; level += step) {
        if (level >= hi - step) level = hi
        const y = opts.rampTop + rows * (level - lo) / span
        calls.push(['fillRect', 30, y, 5, 1, theme.rampFill], ['fillText', (-level).toFixed(0), 11, y + labelDrop])
    }
    return { canvas: { width: opts.dbfsWidth + histWidth, height: rows + opts.timeHeight }, calls }
}

// one filled outline: from (left, top) down the rows, x = left + width * count / peak, closing at (left, top + rows)
function outline(left, top, rows, xOf, fill, stroke, line) {
    const path = [['beginPath'], ['moveTo', left, top]]
    for (let row = 0; row < rows; ++row) path.push(['lineTo', left + xOf(row), top + row])
    path.push(['lineTo', left, top + rows], ['fill', fill], ['stroke', stroke, line])
    return path
}

/**
 * The two filled outlines drawHistograms strokes next to the ramp: colour-index counts and centi-bel counts, one vertex per
 * ramp row (lib/spectroplot.js:686-757).  `o`: {c_hist, cB_hist, cmapLength, height, histWidth, opts?, theme?}.
 * Returns [] when histWidth is 0, as the reference returns early.
 */
function histogramOutlines(o) {
    const opts = Object.assign({}, DEFAULT_OPTS, o.opts), theme = Object.assign({}, DEFAULT_THEME, o.theme)
    const wide = o.histWidth === undefined ? opts.histWidth : o.histWidth
    if (!wide) return []
    const rows = o.height, top = o.cmapLength - 1, last = o.cB_hist.length - 1
    let colourPeak = 0, cbPeak = 0
    for (let i = 0; i <= top; ++i) if (o.c_hist[i] > colourPeak) colourPeak = o.c_hist[i]
    for (let i = 0; i <= last; ++i) if (o.cB_hist[i] > cbPeak) cbPeak = o.cB_hist[i]
    const colour = outline(opts.histLeft, opts.rampTop, rows,
        row => wide * o.c_hist[top - Math.round(row * top / (rows - 1))] / colourPeak, theme.histoFill, theme.histoStroke, theme.histoLine)
    const centibel = outline(opts.histLeft, opts.rampTop, rows,
        row => wide * o.cB_hist[Math.round(row * last / (rows - 1))] / cbPeak, theme.dbfsFill, theme.dbfsStroke, theme.dbfsLine)
    return colour.concat(centibel)
}

/**
 * The gauge strips of one reply: per column a grey level `255 - gauge_max` and the two bars the reference fills
 * (lib/spectroplot.js:1246-1268): min/max bar from `~~(g_min*scale)` of height `~~((g_max-g_min)*scale)`, amplitude bar of
 * height `~~(g_amp*ampScale)`, at x = column + reply.offset.  Pinned by the fillRect calls of the reference's own processData
 * run on recording canvases (tests/golden/caller.json, tests/js/check_caller.js).
 */
function gaugeColumns(reply, sliceWidth, minmaxHeight, ampHeight) {
    const out = { minmax: [], amp: [] }
    for (let col = 0; col < sliceWidth; col++) {
        const lo = reply.gauge_mins[col], hi = reply.gauge_maxs[col]
        const grey = 255 - hi, style = `rgb(${grey},${grey},${grey})`, x = col + reply.offset
        if (minmaxHeight) out.minmax.push(['fillRect', x, ~~(lo * (minmaxHeight / 256)), 1, ~~((hi - lo) * (minmaxHeight / 256)), style])
        if (ampHeight) out.amp.push(['fillRect', x, 0, 1, ~~(reply.gauge_amps[col] * (ampHeight / 256)), style])
    }
    return out
}

module.exports = { colorRamp, rampMarkers, histogramOutlines, gaugeColumns, DEFAULT_THEME, DEFAULT_OPTS }
