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
 *   beginPath / moveTo / lineTo / fill(fillStyle) / stroke(strokeStyle, lineWidth)
 *                                      the path subset of the histogram outlines (lib/spectroplot.js:686-757): polygons filled by
 *                                      scanline under the non-zero winding rule, a pixel belonging to the shape when its centre does
 *                                      (no anti-aliasing); strokes as the set of pixel centres within lineWidth / 2 of a segment
 * Rectangles whose edges fall between pixels (a dB tick at a fractional y) are drawn with coverage-weighted source-over, which a
 * browser's anti-aliasing approximates but does not promise: such pixels are not pinned, and neither are the pixels a polygon's edge
 * or a stroke passes through (a browser shades them by coverage; here they are in or out).  The interior and the exterior are exact.
 * Text (`fillText`) is skipped by replay() and stays a command list: fonts are the caller's.
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

    // ---- paths: the subset drawHistograms uses (one sub-path of straight segments, filled, then stroked) ----
    beginPath() { this.path = [] }
    moveTo(x, y) { (this.path = this.path || []).push([[x, y]]) }
    lineTo(x, y) {
        if (!this.path || !this.path.length) return this.moveTo(x, y)           // canvas: lineTo without a current point acts as moveTo
        this.path[this.path.length - 1].push([x, y])
    }

    /** fill(): every sub-path closed implicitly; non-zero winding; a pixel is inside when its centre is. */
    fill(style) {
        const c = parseColor(style), subs = (this.path || []).filter(p => p.length > 2)
        if (!subs.length) return
        let y0 = Infinity, y1 = -Infinity
        for (const p of subs) for (const [, y] of p) { if (y < y0) y0 = y; if (y > y1) y1 = y }
        for (let py = Math.max(0, Math.floor(y0)); py < Math.min(this.height, Math.ceil(y1)); py++) {
            const yc = py + 0.5, hits = []
            for (const p of subs) {
                for (let i = 0; i < p.length; i++) {
                    const [ax, ay] = p[i], [bx, by] = p[(i + 1) % p.length]
                    if ((ay <= yc) === (by <= yc)) continue                      // the edge does not cross this row of centres
                    hits.push([ax + (yc - ay) * (bx - ax) / (by - ay), by > ay ? 1 : -1])
                }
            }
            hits.sort((u, v) => u[0] - v[0])
            let wind = 0
            for (let k = 0; k + 1 < hits.length; k++) {
                wind += hits[k][1]
                if (!wind) continue
                // centres px + 0.5 in [x_k, x_k+1)
                for (let px = Math.max(0, Math.ceil(hits[k][0] - 0.5)); px < Math.min(this.width, Math.ceil(hits[k + 1][0] - 0.5)); px++) this._blend(px, py, c, 1)
            }
        }
    }

    /** stroke(): every pixel whose centre lies within lineWidth / 2 of a segment of the path, once. */
    stroke(style, lineWidth) {
        const c = parseColor(style), hw = (lineWidth === undefined ? 1 : lineWidth) / 2, seen = new Set()
        for (const p of this.path || []) {
            for (let i = 0; i + 1 < p.length; i++) {
                const [ax, ay] = p[i], [bx, by] = p[i + 1], dx = bx - ax, dy = by - ay, len2 = dx * dx + dy * dy
                for (let py = Math.max(0, Math.floor(Math.min(ay, by) - hw)); py <= Math.min(this.height - 1, Math.ceil(Math.max(ay, by) + hw)); py++) {
                    for (let px = Math.max(0, Math.floor(Math.min(ax, bx) - hw)); px <= Math.min(this.width - 1, Math.ceil(Math.max(ax, bx) + hw)); px++) {
                        const qx = px + 0.5 - ax, qy = py + 0.5 - ay
                        const t = len2 ? Math.min(1, Math.max(0, (qx * dx + qy * dy) / len2)) : 0
                        const ex = qx - t * dx, ey = qy - t * dy
                        if (ex * ex + ey * ey > hw * hw || seen.has(py * this.width + px)) continue
                        seen.add(py * this.width + px)
                        this._blend(px, py, c, 1)
                    }
                }
            }
        }
    }

    /** Replays a consumers.js command list at an offset; returns the commands it left out (text). */
    replay(calls, dx, dy) {
        dx = dx || 0; dy = dy || 0
        const skipped = []
        for (const c of calls) {
            if (c[0] === 'fillRect') this.fillRect(c[1] + dx, c[2] + dy, c[3], c[4], c[5])
            else if (c[0] === 'putImageData') this.putImageData(c[5], c[1] + dx, c[2] + dy)
            else if (c[0] === 'beginPath') this.beginPath()
            else if (c[0] === 'moveTo') this.moveTo(c[1] + dx, c[2] + dy)
            else if (c[0] === 'lineTo') this.lineTo(c[1] + dx, c[2] + dy)
            else if (c[0] === 'fill') this.fill(c[1])
            else if (c[0] === 'stroke') this.stroke(c[1], c[2])
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
 * The two histogram outlines (drawHistograms, lib/spectroplot.js:686-757) are drawn over the scale as the reference draws them.
 * `r` is renderSliced's result, `o` = {cmap (array path: the message's cmap with forced ends), gain, range, n, waterfall?, ampHeight?,
 * minmaxHeight?, histWidth?, opts?, theme?}.  Returns {surface, skipped: the text commands a canvas would still have to draw}.
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
    skipped = skipped.concat(s.replay(histogramOutlines({ c_hist: r.c_hist, cB_hist: r.cB_hist, cmapLength: o.cmap.length, height: scaleRows,
        histWidth: o.histWidth, opts: o.opts, theme: o.theme }), r.width, gaugeRows))
    return { surface: s, skipped, origin: { image: [0, gaugeRows], scale: [r.width, gaugeRows] } }
}

module.exports = { Surface, parseColor, composePlot }
