'use strict'
/**
 * A software surface for the part of the reference's canvas drawing that is exactly rasterisable, and the composition of a headless
 * "full plot" from a render's outputs (SURVEY 8 f4).
 *
 * The reference draws the side outputs of a reply into canvases (lib/spectroplot.js:620-684 colour ramp and dB ticks, :1246-1268 gauge
 * strips, :1241-1244 the strips themselves); js/consumers.js restates those as draw-command lists, pinned call for call against the
 * reference.  Surface replays the commands whose result does not depend on a browser's rasteriser:
 *   putImageData(image, x, y)          copies pixels, no blending (canvas semantics)
 *   fillRect(x, y, w, h, fillStyle)    integer-aligned rectangles with opaque or translucent CSS colours (#rgb, #rrggbb, rgb(), rgba());
 *                                      source-over blending in 8-bit, as a canvas without colour management does it
 * Rectangles whose edges fall between pixels (a dB tick at a fractional y) are drawn with coverage-weighted source-over, which a
 * browser's anti-aliasing approximates but does not promise: such pixels are not pinned.  Text (`fillText`) and the histogram
 * outlines (`beginPath` / `lineTo` / `fill` / `stroke`: anti-aliased polygons) are skipped by replay() and stay command lists.
 */

function parseColor(style) {
    let m
    if ((m = /^#([0-9a-f]{3})$/i.exec(style))) return [...m[1]].map(h => parseInt(h + h, 16)).concat(255)
    if ((m = /^#([0-9a-f]{6})$/i.exec(style))) return [0, 2, 4].map(i => parseInt(m[1].slice(i, i + 2), 16)).concat(255)
    if ((m = /^rgba?\(\s*([\d.]+)\s*,\s*([\d.]+)\s*,\s*([\d.]+)\s*(?:,\s*([\d.]+)\s*)?\)$/i.exec(style))) {
        const a = m[4] === undefined ? 1 : Math.min(1, Math.max(0, parseFloat(m[4])))
        return [Math.round(+m[1]), Math.round(+m[2]), Math.round(+m[3]), Math.round(a * 255)]
    }
    throw new Error('raster: unsupported fillStyle ' + style)
}

class Surface {
    constructor(width, height, background) {
        this.width = width
        this.height = height
        this.data = new Uint8ClampedArray(4 * width * height)               // transparent black, like a fresh canvas
        if (background) this.fillRect(0, 0, width, height, background)
    }

    /** putImageData: `image` = {data, width, height}; pixels outside the surface are dropped. */
    putImageData(image, x, y) {
        const w = image.width, h = image.height === undefined ? image.data.length / 4 / image.width : image.height
        for (let row = 0; row < h; row++) {
            const ty = y + row
            if (ty < 0 || ty >= this.height) continue
            const x0 = Math.max(0, -x), x1 = Math.min(w, this.width - x)
            if (x1 <= x0) continue
            this.data.set(image.data.subarray(4 * (row * w + x0), 4 * (row * w + x1)), 4 * (ty * this.width + x + x0))
        }
    }

    _blend(px, py, c, coverage) {
        if (px < 0 || py < 0 || px >= this.width || py >= this.height) return
        const a = (c[3] / 255) * coverage
        if (a <= 0) return
        const o = 4 * (py * this.width + px), d = this.data
        const da = d[o + 3] / 255, oa = a + da * (1 - a)
        for (let k = 0; k < 3; k++) d[o + k] = oa ? (c[k] * a + d[o + k] * da * (1 - a)) / oa : 0
        d[o + 3] = oa * 255
    }

    fillRect(x, y, w, h, style) {
        const c = parseColor(style)
        if (w < 0) { x += w; w = -w }
        if (h < 0) { y += h; h = -h }
        const x1 = x + w, y1 = y + h
        for (let py = Math.floor(y); py < Math.ceil(y1); py++) {
            const cy = Math.min(py + 1, y1) - Math.max(py, y)
            if (cy <= 0) continue
            for (let px = Math.floor(x); px < Math.ceil(x1); px++) {
                const cx = Math.min(px + 1, x1) - Math.max(px, x)
                if (cx > 0) this._blend(px, py, c, cx * cy)
            }
        }
    }

    /** Replays a consumers.js command list at an offset; returns the commands it left out (text, paths). */
    replay(calls, dx, dy) {
        dx = dx || 0; dy = dy || 0
        const skipped = []
        for (const c of calls) {
            if (c[0] === 'fillRect') this.fillRect(c[1] + dx, c[2] + dy, c[3], c[4], c[5])
            else if (c[0] === 'putImageData') this.putImageData(c[5], c[1] + dx, c[2] + dy)
            else skipped.push(c)
        }
        return skipped
    }

    /** Binary PPM (alpha composited over `background`, default white - an HTML page's default). */
    toPPM(background) {
        const bg = parseColor(background || '#fff')
        const rgb = Buffer.alloc(3 * this.width * this.height)
        for (let p = 0, q = 0; p < this.data.length; p += 4) {
            const a = this.data[p + 3] / 255
            for (let k = 0; k < 3; k++) rgb[q++] = Math.round(this.data[p + k] * a + bg[k] * (1 - a))
        }
        return Buffer.concat([Buffer.from(`P6\n${this.width} ${this.height}\n255\n`), rgb])
    }
}

/**
 * One image with everything the reference shows around a spectrogram that can be drawn exactly: amplitude gauge strip, min/max gauge
 * strip, the spectrogram, and to its right the dB scale (colour ramp + tick marks).  The layout is this module's (the reference's is
 * CSS): rows = ampHeight | minmaxHeight | image height + timeHeight, columns = image width | dbfsWidth + histWidth.
 * `r` is renderSliced's result, `o` = {cmap (array path: the message's cmap with forced ends), gain, range, n, waterfall?, ampHeight?,
 * minmaxHeight?, histWidth?, opts?, theme?}.  Returns {surface, skipped: the text / outline commands a canvas would still have to draw}.
 */
function composePlot(r, o) {
    const { rampMarkers, histogramOutlines, gaugeColumns } = require('./consumers.js')
    const ampH = o.ampHeight === undefined ? 32 : o.ampHeight, mmH = o.minmaxHeight === undefined ? 32 : o.minmaxHeight
    const scaleRows = o.waterfall ? r.height : o.n
    const ramp = rampMarkers({ gain: o.gain, range: o.range, height: scaleRows, cmap: o.cmap, histWidth: o.histWidth, opts: o.opts, theme: o.theme })
    const gaugeRows = o.waterfall ? 0 : ampH + mmH                     // the gauges run along the time axis, which is horizontal only here
    const s = new Surface(r.width + ramp.canvas.width, gaugeRows + Math.max(r.height, ramp.canvas.height), '#fff')
    let skipped = []
    if (gaugeRows) {
        for (const reply of r.replies) {
            const g = gaugeColumns(reply, r.sliceWidth, mmH, ampH)
            s.replay(g.amp, 0, 0)
            s.replay(g.minmax, 0, ampH)
        }
    }
    s.putImageData({ data: r.data, width: r.width, height: r.height }, 0, gaugeRows)
    skipped = skipped.concat(s.replay(ramp.calls, r.width, gaugeRows))
    skipped = skipped.concat(histogramOutlines({ c_hist: r.c_hist, cB_hist: r.cB_hist, cmapLength: o.cmap.length, height: scaleRows,
        histWidth: o.histWidth, opts: o.opts, theme: o.theme }).map(c => c.slice()))
    return { surface: s, skipped, origin: { image: [0, gaugeRows], scale: [r.width, gaugeRows] } }
}

module.exports = { Surface, parseColor, composePlot }
