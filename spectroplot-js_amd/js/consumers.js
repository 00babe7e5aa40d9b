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
 * recording context (tests/golden/consumers.json, oracle/ref_harness.mjs).  gaugeColumns restates an inline block of
 * processData that cannot be called on its own; it has no reference vector (see its docstring).
 */

const DEFAULT_THEME = { rampFill: '#666', histoLine: 2, histoStroke: '#b0b', histoFill: 'rgba(187,0,187,0.2)', dbfsLine: 2,
    dbfsStroke: '#999', dbfsFill: 'rgba(153,153,153,0.2)' }            // lib/spectroplot.js:390-400 (light theme)
const DEFAULT_OPTS = { dbfsWidth: 60, timeHeight: 20, rampTop: 10, rampWidth: 15, histWidth: 100, histLeft: 55 }   // :266-274

/** The colour ramp image: row y shows LUT entry `len-1 - round(y*(len-1)/(height-1))`   (lib/spectroplot.js:646-660). */
function colorRamp(cmap, rampWidth, rampHeight) {
    const data = new Uint8ClampedArray(4 * rampWidth * rampHeight)
    const color_max = cmap.length - 1
    for (let y = 0; y < rampHeight; ++y) {
        const idx = Math.round(y * color_max / (rampHeight - 1))
        const color = cmap[color_max - idx]
        for (let x = 0; x < rampWidth; ++x) {
            const j = x * 4 + rampWidth * y * 4
            data[j + 0] = color[0]
            data[j + 1] = color[1]
            data[j + 2] = color[2]
            data[j + 3] = 255
        }
    }
    return { data, width: rampWidth, height: rampHeight }
}

/**
 * Everything drawColorRamp hands to the dbfs canvas: its size, the ramp image at (35, rampTop) and the tick / label commands
 * (lib/spectroplot.js:620-684).  `o`: {gain, range, height, cmap, histWidth, opts?, theme?}.
 */
function rampMarkers(o) {
    const opts = Object.assign({}, DEFAULT_OPTS, o.opts), theme = Object.assign({}, DEFAULT_THEME, o.theme)
    const gain = o.gain, dB_range = o.range, height = o.height
    const histWidth = o.histWidth === undefined ? opts.histWidth : o.histWidth
    const calls = []
    const font_y = 10
    const rampLeft = 35, rampTop = opts.rampTop, rampHeight = height
    const ramp = colorRamp(o.cmap, opts.rampWidth, rampHeight)
    calls.push(['putImageData', rampLeft, rampTop, ramp.width, ramp.height, ramp])
    const num_dbfs_markers = height / 50
    let dbfs_markers_step = (gain + dB_range) / num_dbfs_markers
    dbfs_markers_step = Math.round(dbfs_markers_step / 3) * 3
    if (dbfs_markers_step < 1.0) dbfs_markers_step = 1.0
    for (let d = gain; d < gain + dB_range; d += dbfs_markers_step) {
        if (d >= gain + dB_range - dbfs_markers_step) d = gain + dB_range
        const y = rampTop + rampHeight * (d - gain) / dB_range
        calls.push(['fillRect', 30, y, 5, 1, theme.rampFill])
        calls.push(['fillText', (-d).toFixed(0), 11, y + font_y / 2 - 1])
    }
    return { canvas: { width: opts.dbfsWidth + histWidth, height: height + opts.timeHeight }, calls }
}

/**
 * The two filled outlines drawHistograms strokes next to the ramp: colour-index counts and centi-bel counts, one vertex per
 * ramp row (lib/spectroplot.js:686-757).  `o`: {c_hist, cB_hist, cmapLength, height, histWidth, opts?, theme?}.
 * Returns [] when histWidth is 0, as the reference returns early.
 */
function histogramOutlines(o) {
    const opts = Object.assign({}, DEFAULT_OPTS, o.opts), theme = Object.assign({}, DEFAULT_THEME, o.theme)
    const histWidth = o.histWidth === undefined ? opts.histWidth : o.histWidth
    const histLeft = opts.histLeft
    if (!histWidth) return []
    const c_hist = o.c_hist, cB_hist = o.cB_hist
    const color_max = o.cmapLength - 1
    const rampHeight = o.height, rampTop = opts.rampTop
    const calls = []
    let c_hist_max = 0
    calls.push(['beginPath'])
    calls.push(['moveTo', histLeft, rampTop])
    for (let i = 0; i <= color_max; ++i) if (c_hist[i] > c_hist_max) c_hist_max = c_hist[i]
    for (let y = 0; y < rampHeight; ++y) {
        const i = color_max - Math.round(y * color_max / (rampHeight - 1))
        const h = histWidth * c_hist[i] / c_hist_max
        calls.push(['lineTo', histLeft + h, rampTop + y])
    }
    calls.push(['lineTo', histLeft, rampTop + rampHeight])
    calls.push(['fill', theme.histoFill])
    calls.push(['stroke', theme.histoStroke, theme.histoLine])

    const cB_hist_size = cB_hist.length
    let cB_hist_max = 0
    for (let i = 0; i < cB_hist_size; ++i) if (cB_hist[i] > cB_hist_max) cB_hist_max = cB_hist[i]
    calls.push(['beginPath'])
    calls.push(['moveTo', histLeft, rampTop])
    for (let y = 0; y < rampHeight; ++y) {
        const i = Math.round(y * (cB_hist_size - 1) / (rampHeight - 1))
        const h = histWidth * cB_hist[i] / cB_hist_max
        calls.push(['lineTo', histLeft + h, rampTop + y])
    }
    calls.push(['lineTo', histLeft, rampTop + rampHeight])
    calls.push(['fill', theme.dbfsFill])
    calls.push(['stroke', theme.dbfsStroke, theme.dbfsLine])
    return calls
}

/**
 * The gauge strips of one reply: per column a grey level `255 - gauge_max` and the two bars the reference fills
 * (lib/spectroplot.js:1246-1268): min/max bar from `~~(g_min*scale)` of height `~~((g_max-g_min)*scale)`, amplitude bar of
 * height `~~(g_amp*ampScale)`, at x = column + reply.offset.  NOT pinned by a reference vector: the block is inline in
 * processData's promise handler and cannot be run apart from the DOM-bound render; it is restated line by line.
 */
function gaugeColumns(reply, sliceWidth, minmaxHeight, ampHeight) {
    const minmax = [], amp = []
    const ampScale = ampHeight / 256, minmaxScale = minmaxHeight / 256
    for (let x = 0; x < sliceWidth; x++) {
        const g_min = reply.gauge_mins[x], g_max = reply.gauge_maxs[x], g_amp = reply.gauge_amps[x]
        const style = `rgb(${255 - g_max},${255 - g_max},${255 - g_max})`
        if (minmaxHeight) minmax.push(['fillRect', x + reply.offset, ~~(g_min * minmaxScale), 1, ~~((g_max - g_min) * minmaxScale), style])
        if (ampHeight) amp.push(['fillRect', x + reply.offset, 0, 1, ~~(g_amp * ampScale), style])
    }
    return { minmax, amp }
}

module.exports = { colorRamp, rampMarkers, histogramOutlines, gaugeColumns, DEFAULT_THEME, DEFAULT_OPTS }
