/**
 * Reference harness — runs the REAL reference worker (lib/worker.js of triq-org/spectroplot-js) under Node.
 *
 * TEST INFRASTRUCTURE (container-only).  This file is copied by oracle/gen_golden.js into a scratch directory
 * OUTSIDE the repo, next to a scratch copy of the reference's lib/ directory, and executed there with
 *   node --experimental-specifier-resolution=node ref_harness.mjs <repo> <cases.json> <outdir>
 * (recipe: SURVEY.md §8c).  Nothing from the reference is ever copied into the repo; only the outputs
 * (vectors) written here are committed under tests/golden/.
 *
 * The reference module is strict-mode and assigns the bare global `onmessage` (lib/worker.js:158), so the
 * globals are created before the dynamic import.
 */
import { createRequire } from 'module'
import fs from 'fs'
import path from 'path'
import crypto from 'crypto'

const require = createRequire(import.meta.url)
const [repo, casesFile, outdir] = process.argv.slice(2)
const siggen = require(path.join(repo, 'oracle/js/siggen.js'))

let captured = null
globalThis.self = globalThis
globalThis.onmessage = null
globalThis.postMessage = (msg) => { captured = msg }

const run = async () => {
    await import('./lib/worker.js')
    const windows = await import('./lib/windows.js')
    const SampleView = (await import('./lib/samples.js')).default
    const FFTNayuki = (await import('./lib/fft_nayuki.js')).default
    const cmapMods = [
        await import('./lib/cube1cmap.js'), await import('./lib/soxcmap.js'), await import('./lib/naivecmap.js'),
        await import('./lib/matplotlibcmaps.js'), await import('./lib/parabolacmap.js'),
    ]
    const cmaps = Object.assign({}, ...cmapMods)

    const sha256 = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex')
    const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.readBigUInt64LE().toString(16).padStart(16, '0') }
    const f64bytes = (arr) => { const b = Buffer.alloc(arr.length * 8); arr.forEach((v, i) => b.writeDoubleLE(v, i * 8)); return b }

    const spec = JSON.parse(fs.readFileSync(casesFile, 'utf8'))

    // ---- colour maps as data (evaluated module exports, before end-forcing) -------------------------------
    {
        const names = Object.keys(cmaps).sort()
        const index = []
        const chunks = []
        let off = 0
        for (const name of names) {
            const lut = cmaps[name]
            const u8 = new Uint8Array(lut.length * 3)
            lut.forEach((c, i) => { u8[3 * i] = c[0]; u8[3 * i + 1] = c[1]; u8[3 * i + 2] = c[2] })
            const allInt = lut.every(c => c.every(v => Number.isInteger(v) && v >= 0 && v <= 255))
            index.push({ name, length: lut.length, offset: off, integer: allInt,
                sha1_12: crypto.createHash('sha1').update(Buffer.from(u8)).digest('hex').slice(0, 12) })
            chunks.push(Buffer.from(u8)); off += u8.length
        }
        fs.writeFileSync(path.join(outdir, 'cmaps.bin'), Buffer.concat(chunks))
        fs.writeFileSync(path.join(outdir, 'cmaps.json'), JSON.stringify(index, null, 1))
    }

    const getCmap = (c) => {
        let lut
        if (c.cmap.startsWith('custom:')) {
            const len = parseInt(c.cmap.split(':')[1], 10)
            lut = []
            for (let i = 0; i < len; i++) lut.push([(i * 7) & 255, (i * 13 + 5) & 255, (255 - i) & 255])
        } else {
            lut = cmaps[c.cmap + '_cmap'].map(e => e.slice())
        }
        if (c.force_ends) { // the caller's in-place edit, lib/spectroplot.js:1129-1130
            lut[0] = [0, 0, 0]
            lut[lut.length - 1] = [255, 255, 255]
        }
        return lut
    }

    const makeInput = (c) => {
        const sw = siggen.SAMPLE_WIDTH[c.gen_format || c.format.toUpperCase()] || 2
        const count = Math.ceil(c.bytes / sw)
        const full = siggen.generate(c.gen_format || c.format, c.gen, count, 0)
        return full.slice(0, c.bytes)
    }

    const renderOne = (c, buffer, width, offset, windowc, block_norm, cmap) => {
        captured = null
        globalThis.onmessage({ data: {
            block_norm, gain: c.gain, range: c.range, cmap, n: c.n, windowc, width, offset,
            buffer, format: c.format, channelMode: !!c.channelMode, waterfall: !!c.waterfall,
        } })
        if (!captured) throw new Error('no reply for ' + c.name)
        return captured
    }

    const digest = (r, keepRgba) => {
        const cB = {}
        r.cB_hist.forEach((v, i) => { if (v) cB[i] = v })
        // properties written at negative "indices" (lib/worker.js:106 quirk) are not part of the array
        const out = {
            rgba_sha256: sha256(r.imageData.data), rgba_len: r.imageData.data.length,
            gauge_mins: Buffer.from(r.gauge_mins).toString('hex'),
            gauge_maxs: Buffer.from(r.gauge_maxs).toString('hex'),
            gauge_amps: Buffer.from(r.gauge_amps).toString('hex'),
            c_hist: Array.from(r.c_hist), cB_hist: cB, cB_len: r.cB_hist.length,
            dBfs_min: f64hex(r.dBfs_min), dBfs_max: f64hex(r.dBfs_max),
            dBfs_min_num: String(r.dBfs_min), dBfs_max_num: String(r.dBfs_max),
            offset: r.offset,
        }
        return out
    }

    // ---- worker cases ----------------------------------------------------------------------------------------
    const results = []
    for (const c of spec.worker_cases) {
        const input = makeInput(c)
        const wf = windows[c.window + 'Window']
        const { window: windowc, weight } = wf(c.n)
        const block_norm = 1.0 / weight // lib/spectroplot.js:1116
        const cmap = getCmap(c)
        const out = { name: c.name, block_norm: f64hex(block_norm), input_sha256: sha256(input) }
        const slices = c.slices || 1
        if (slices === 1 && !c.via_slice) {
            const buf = input.buffer.slice(input.byteOffset, input.byteOffset + input.byteLength)
            let r
            try {
                r = renderOne(c, buf, c.width, c.offset || 0, windowc, block_norm, cmap)
            } catch (e) {
                out.throws = String(e && e.message ? e.constructor.name + ': ' + e.message : e)
                results.push(out); continue
            }
            out.reply = digest(r)
            if (c.keep_rgba) fs.writeFileSync(path.join(outdir, c.name + '.rgba'), Buffer.from(r.imageData.data))
        } else {
            // the caller's slice + merge, lib/spectroplot.js:1206-1244
            const sv = new SampleView(c.format, input.buffer.slice(input.byteOffset, input.byteOffset + input.byteLength))
            const endSample = ~~(input.byteLength / sv.sampleWidth)
            const sliceWidth = ~~(c.width / slices)
            const height = c.n
            const merged = new Uint8Array(4 * c.width * height) // canvas starts transparent black
            const c_hist = new Array(cmap.length).fill(0)
            const cB_hist = new Array(1000).fill(0)
            let dmin = 0.0, dmax = -200.0
            out.slices = []
            for (let i = 0; i < slices; i++) {
                const bufferSlice = sv.slice(i, slices, 0, endSample)
                const r = renderOne(c, bufferSlice, sliceWidth, i * sliceWidth, windowc, block_norm, cmap)
                const d = digest(r)
                d.slice_bytes = bufferSlice.byteLength
                out.slices.push(d)
                if (r.dBfs_min < dmin) dmin = r.dBfs_min
                if (r.dBfs_max > dmax) dmax = r.dBfs_max
                for (let x = 0; x < 1000; x++) cB_hist[x] += r.cB_hist[x]
                for (let x = 0; x < cmap.length; x++) c_hist[x] += r.c_hist[x]
                const img = r.imageData.data
                if (!c.waterfall) {
                    // putImageData(image(w=sliceWidth,h=n), offset, 0) into a canvas of `width` columns
                    for (let y = 0; y < height; y++)
                        for (let x = 0; x < sliceWidth; x++)
                            for (let b = 0; b < 4; b++)
                                merged[4 * (y * c.width + r.offset + x) + b] = img[4 * (y * sliceWidth + x) + b]
                } else {
                    // image(w=n,h=sliceWidth) at (0, width - sliceWidth - offset); canvas is n wide, `width` tall
                    const y0 = c.width - sliceWidth - r.offset
                    for (let y = 0; y < sliceWidth; y++)
                        for (let x = 0; x < height; x++)
                            for (let b = 0; b < 4; b++)
                                merged[4 * ((y0 + y) * height + x) + b] = img[4 * (y * height + x) + b]
                }
            }
            const cB = {}
            cB_hist.forEach((v, i) => { if (v) cB[i] = v })
            out.merged = { rgba_sha256: sha256(merged), c_hist, cB_hist: cB, dBfs_min: f64hex(dmin), dBfs_max: f64hex(dmax),
                slice_width: sliceWidth }
        }
        results.push(out)
    }
    fs.writeFileSync(path.join(outdir, 'worker_expected.json'), JSON.stringify(results, null, 0))

    // ---- window KAT (lib/windows.js) -----------------------------------------------------------------------------
    {
        const idx = []; const chunks = []; let off = 0
        for (const name of spec.window_kat.names) for (const n of spec.window_kat.sizes) {
            const { window, weight } = windows[name + 'Window'](n)
            const b = f64bytes(window)
            idx.push({ name, n, offset: off, weight: f64hex(weight), sha256: crypto.createHash('sha256').update(b).digest('hex') })
            if (n <= spec.window_kat.keep_max_n) { chunks.push(b); off += b.length } else idx[idx.length - 1].offset = -1
        }
        fs.writeFileSync(path.join(outdir, 'windows.bin'), Buffer.concat(chunks))
        fs.writeFileSync(path.join(outdir, 'windows.json'), JSON.stringify(idx, null, 0))
    }

    // ---- twiddle + FFT KAT (lib/fft_nayuki.js) -----------------------------------------------------------------
    {
        const tw = []
        for (const n of spec.fft_kat.twiddle_sizes) {
            const f = new FFTNayuki(n)
            const b = Buffer.concat([f64bytes(f.cosTable), f64bytes(f.sinTable)])
            tw.push({ n, sha256: crypto.createHash('sha256').update(b).digest('hex') })
            if (n === spec.fft_kat.twiddle_keep) fs.writeFileSync(path.join(outdir, 'twiddles_' + n + '.bin'), b)
        }
        const ffts = []
        const chunks = []; let off = 0
        for (const k of spec.fft_kat.cases) {
            const n = k.n
            const re = new Array(n).fill(0), im = new Array(n).fill(0)
            if (k.kind === 'impulse') { re[k.pos] = 1.0; im[k.pos] = -0.5 }
            else if (k.kind === 'dc') { re.fill(0.75); im.fill(-0.25) }
            else { // 'rand': hash-derived values in [-1, 1)
                for (let i = 0; i < n; i++) {
                    re[i] = siggen.hash(k.seed, 2 * i) / 2147483648 - 1.0
                    im[i] = siggen.hash(k.seed, 2 * i + 1) / 2147483648 - 1.0
                }
            }
            const f = new FFTNayuki(n)
            f.transform(re, im)
            if (k.split) f.splitreal(re, im)
            const b = Buffer.concat([f64bytes(re), f64bytes(im)])
            const e = Object.assign({}, k, { sha256: crypto.createHash('sha256').update(b).digest('hex'), offset: -1 })
            if (n <= spec.fft_kat.keep_max_n) { e.offset = off; chunks.push(b); off += b.length }
            ffts.push(e)
        }
        let thrown = null
        try { new FFTNayuki(12) } catch (e) { thrown = { type: typeof e, value: String(e) } }
        fs.writeFileSync(path.join(outdir, 'fft.bin'), Buffer.concat(chunks))
        fs.writeFileSync(path.join(outdir, 'fft.json'), JSON.stringify({ twiddles: tw, cases: ffts, non_pow2_throw: thrown }, null, 0))
    }

    // ---- decode KAT (lib/samples.js) ---------------------------------------------------------------------------
    {
        const out = []
        for (const k of spec.decode_kat) {
            const sw = siggen.SAMPLE_WIDTH[k.gen_format] || 2
            const bytes = k.hex ? new Uint8Array(Buffer.from(k.hex, 'hex'))
                : siggen.generate(k.gen_format, { kind: 'bytes', seed: k.seed }, Math.ceil(k.bytes / sw), 0).slice(0, k.bytes)
            let sv
            const e = { name: k.name, format: k.format, bytes: bytes.length }
            try {
                sv = new SampleView(k.format, bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength))
            } catch (err) {
                e.throws = err.constructor.name + ': ' + err.message
                out.push(e); continue
            }
            e.sampleCount = f64hex(sv.sampleCount); e.sampleWidth = sv.sampleWidth
            e.norm_format = sv.format
            const vals = []
            for (let pos = k.pos_lo; pos < k.pos_hi; pos++) vals.push(f64hex(sv.sampleI(pos)), f64hex(sv.sampleQ(pos)))
            e.pos_lo = k.pos_lo; e.values = vals
            out.push(e)
        }
        fs.writeFileSync(path.join(outdir, 'decode.json'), JSON.stringify(out, null, 0))
    }

    // ---- engine math KAT: Math.log10 / Math.cos / Math.sin of this V8 (the arithmetic the reference runs on) ---
    {
        const N = spec.math_kat.count
        const xs = new Float64Array(N), ls = new Float64Array(N)
        const dv = new DataView(new ArrayBuffer(8))
        for (let i = 0; i < N; i++) {
            // positive doubles with exponents spread over [2^-80, 2^40], plus a sprinkling of specials
            const hi = siggen.hash(spec.math_kat.seed, 2 * i), lo = siggen.hash(spec.math_kat.seed, 2 * i + 1)
            const e = 1023 - 80 + (hi >>> 20) % 121
            dv.setUint32(0, lo, true); dv.setUint32(4, ((e << 20) | (hi & 0xfffff)) >>> 0, true)
            xs[i] = dv.getFloat64(0, true)
        }
        const specials = [0, -0, 1, 10, 100, 1e-300, 5e-324, 2.2250738585072014e-308, Infinity, NaN, -1, 0.1, 0.5, 2, 1e300,
            1.0000000000000002, 0.9999999999999999]
        specials.forEach((v, i) => { xs[i] = v })
        for (let i = 0; i < N; i++) ls[i] = Math.log10(xs[i])
        const M = spec.math_kat.trig_count
        const ts = new Float64Array(M), cs = new Float64Array(M), ss = new Float64Array(M)
        for (let i = 0; i < M; i++) {
            ts[i] = (siggen.hash(spec.math_kat.seed ^ 0x7777, i) / 4294967296) * 6.5 * Math.PI
            cs[i] = Math.cos(ts[i]); ss[i] = Math.sin(ts[i])
        }
        fs.writeFileSync(path.join(outdir, 'math_log10.bin'), Buffer.concat([Buffer.from(xs.buffer), Buffer.from(ls.buffer)]))
        fs.writeFileSync(path.join(outdir, 'math_trig.bin'), Buffer.concat([Buffer.from(ts.buffer), Buffer.from(cs.buffer), Buffer.from(ss.buffer)]))
    }

    // ---- file-name parsing and key lookup KAT (lib/parseFreqRate.js, lib/utils.js lookup) -----------------------------
    {
        const saved = console.log
        console.log = () => {}                      // parseFreqRate logs when it strips a path
        const P = await import('./lib/parseFreqRate.js')
        // utils.js touches `document` / DOMParser only inside functions, lookup itself is pure
        const U = await import('./lib/utils.js')
        const names = spec.parse_kat.names
        const parsed = names.map(nm => ({ name: nm, format: P.parseFormat(nm), fr: P.parseFreqRate(nm) }))
        const table = {}
        for (const k of Object.keys(windows)) table[k] = k
        const lookups = spec.parse_kat.window_keys.map(k => ({ key: k, hit: U.lookup(table, k) }))
        const ctable = {}
        for (const k of Object.keys(cmaps)) ctable[k] = k
        const clookups = spec.parse_kat.cmap_keys.map(k => ({ key: k, hit: U.lookup(ctable, k) }))
        console.log = saved
        fs.writeFileSync(path.join(outdir, 'parse.json'), JSON.stringify({ parsed, lookups, clookups, cmap_key_order: Object.keys(cmaps),
            window_key_order: Object.keys(windows) }, null, 0))
    }


    // ---- side-output consumers KAT (lib/spectroplot.js drawColorRamp :620-684, drawHistograms :686-757) ------------------
    // The methods are run unmodified on a stand-in `this` whose canvas context records the calls it receives.
    {
        const saved = console.log
        console.log = () => {}
        globalThis.window = { File: 1, FileReader: 1, FileList: 1, Blob: 1, addEventListener() {}, dispatchEvent() {} }
        globalThis.alert = () => {}
        globalThis.navigator = { hardwareConcurrency: 2 }
        globalThis.Event = globalThis.Event || function Event() {}
        const mkctx = () => {
            const calls = []
            const ctx = { calls, font: '', fillStyle: '', strokeStyle: '', lineWidth: 0,
                createImageData: (w, h) => ({ data: new Uint8ClampedArray(4 * w * h), width: w, height: h }),
                putImageData(d, x, y) { calls.push(['putImageData', x, y, d.width, d.height, Buffer.from(d.data.buffer).toString('base64')]) },
                fillRect(x, y, w, h) { calls.push(['fillRect', x, y, w, h, this.fillStyle]) },
                fillText(t, x, y) { calls.push(['fillText', t, x, y]) },
                beginPath() { calls.push(['beginPath']) }, moveTo(x, y) { calls.push(['moveTo', x, y]) },
                lineTo(x, y) { calls.push(['lineTo', x, y]) },
                fill() { calls.push(['fill', this.fillStyle]) }, stroke() { calls.push(['stroke', this.strokeStyle, this.lineWidth]) } }
            return ctx
        }
        globalThis.document = { createElement: () => ({ width: 0, height: 0, getContext: () => mkctx() }),
            querySelector: () => null, querySelectorAll: () => [], getElementsByClassName: () => [], addEventListener() {} }
        const SP = await import('./lib/spectroplot.js')
        const proto = SP.Spectroplot.prototype
        const out = []
        for (const c of spec.consumer_kat.cases) {
            const cm = cmaps[c.cmap].map(e => e.slice(0, 3))
            const ctx = mkctx()
            const canvas = { width: 0, height: 0, style: {}, parentNode: { style: {} }, getContext: () => ctx }
            const self = { gain: c.gain, range: c.range, height: c.height, fftN: c.height, histWidth: c.histWidth, cmap: cm,
                opts: { dbfsWidth: 60, timeHeight: 20, rampWidth: c.rampWidth, rampTop: c.rampTop, histLeft: c.histLeft },
                theme: { rampFill: '#666', histoLine: 2, histoStroke: '#b0b', histoFill: 'rgba(187,0,187,0.2)', dbfsLine: 2,
                    dbfsStroke: '#999', dbfsFill: 'rgba(153,153,153,0.2)' },
                parent: { getElementsByClassName: () => [canvas] } }
            proto.drawColorRamp.call(self)
            const ramp = ctx.calls.slice()
            ctx.calls.length = 0
            // deterministic histograms
            const c_hist = new Array(cm.length), cB_hist = new Array(1000)
            for (let i = 0; i < cm.length; i++) c_hist[i] = siggen.hash(c.seed, i) % 5000
            for (let i = 0; i < 1000; i++) cB_hist[i] = siggen.hash(c.seed ^ 0x55, i) % 70000
            proto.drawHistograms.call(self, c_hist, cB_hist)
            out.push({ name: c.name, canvas: { width: canvas.width, height: canvas.height }, ramp, hist: ctx.calls.slice() })
        }
        console.log = saved
        fs.writeFileSync(path.join(outdir, 'consumers.json'), JSON.stringify(out, null, 0))
    }

    // ---- the reference's own caller: startWorkers + processData, unmodified, against a Worker-shaped class ------------------------
    // (lib/spectroplot.js:100-130, :1096-1285).  The module reads navigator.hardwareConcurrency once when it is loaded, so it is
    // imported once per worker count (a query string makes each import a module instance of its own).  The class handed to
    // startWorkers forwards every message to the REAL lib/worker.js loaded above and answers asynchronously, like a worker thread;
    // its instances are Proxies that note which members the reference touches.  processData runs on a stand-in `this` whose
    // canvases record what they are asked to draw.
    {
        const savedLog = console.log, savedTime = console.time, savedTimeEnd = console.timeEnd, savedErr = console.error
        const quiet = () => { console.log = () => {}; console.time = () => {}; console.timeEnd = () => {}; console.error = () => {} }
        const loud = () => { console.log = savedLog; console.time = savedTime; console.timeEnd = savedTimeEnd; console.error = savedErr }
        const out = { runs: [] }
        for (const W of spec.caller_kat.workers) {
            quiet()
            globalThis.navigator = { hardwareConcurrency: W }
            globalThis.window = { File: 1, FileReader: 1, FileList: 1, Blob: 1, Worker: function Worker() {}, innerHeight: 0, addEventListener() {},
                dispatchEvent() {} }
            globalThis.createImageBitmap = () => new Promise(() => {})        // "mostly unsupported": never resolves, nothing is restored
            globalThis.ImageData = class ImageData { constructor(data, width, height) { this.data = data; this.width = width; this.height = height || data.length / 4 / width } }
            const SPW = await import('./lib/spectroplot.js?workers=' + W)
            const touched = { get: new Set(), set: new Set() }
            const messages = []
            class RefWorker {
                constructor() {
                    this.onmessage = null
                    return new Proxy(this, {
                        get(t, k) { if (typeof k === 'string') touched.get.add(k); return t[k] },
                        set(t, k, v) { if (typeof k === 'string') touched.set.add(k); t[k] = v; return true },
                    })
                }
                postMessage(message, transfer) {
                    messages.push({ message, transfer })
                    captured = null
                    globalThis.onmessage({ data: message })                   // the real worker: ignores messages without a buffer
                    const reply = captured
                    if (reply) Promise.resolve().then(() => this.onmessage({ data: reply }))
                }
            }
            SPW.startWorkers(RefWorker)
            const probe = messages.splice(0, messages.length)
            const run = { workers: W, members_read: Array.from(touched.get).sort(), members_written: Array.from(touched.set).sort(),
                probe: probe.map(p => ({ keys: Object.keys(p.message), transfer: p.transfer.length })), cases: [] }
            for (const c of spec.caller_kat.cases) {
                const input = makeInput(c)
                const mkcanvas = (name) => {
                    const calls = []
                    const ctx = { calls, fillStyle: '',
                        getImageData: (x, y, w, h) => ({ data: new Uint8ClampedArray(4 * Math.max(w, 1) * Math.max(h, 1)), width: w, height: h }),
                        drawImage() { calls.push(['drawImage']) },
                        putImageData(d, x, y) { calls.push(['putImageData', x, y, d.width, d.height, sha256(d.data)]) },
                        fillRect(x, y, w, h) { calls.push(['fillRect', x, y, w, h, this.fillStyle]) } }
                    return { name, width: 0, height: 0, style: {}, ctx, getContext: () => ctx }
                }
                const canvases = { fft: mkcanvas('fft'), minmax: mkcanvas('minmax'), amp: mkcanvas('amp') }
                const scroll = { scrollLeft: 0, scrollLeftMax: 0 }
                const hist = {}
                const cmap = cmaps[c.cmap + '_cmap'].map(e => e.slice())
                const self = {
                    buffer: input.buffer.slice(input.byteOffset, input.byteOffset + input.byteLength), inProcess: false, tag: 'kat', turnFlip: !!c.waterfall,
                    opts: { freqWidth: 0, dbfsWidth: 0 }, histWidth: 0, zoom: 1, fftN: c.n, windowF: windows[c.window + 'Window'], gain: c.gain, range: c.range,
                    cmap, channelMode: !!c.channelMode, height: 0, minmaxHeight: c.minmaxHeight, ampHeight: c.ampHeight, scrollElY: { scrollTop: 0 }, zoomPoint: null,
                    parent: { clientWidth: c.width, getElementsByClassName: (cls) => cls === 'scroll' ? [scroll] : [canvases[cls]] },
                    drawColorRamp() {}, drawAxes() {}, buildInfo() {},
                    drawHistograms(c_hist, cB_hist) { hist.c_hist = Array.from(c_hist); hist.cB_hist = {}; cB_hist.forEach((v, i) => { if (v) hist.cB_hist[i] = v }) },
                }
                self.sampleView = new SampleView(c.format, self.buffer)
                globalThis.window.innerHeight = c.width
                await SPW.Spectroplot.prototype.processData.call(self)
                const sent = messages.splice(0, messages.length)
                run.cases.push({ name: c.name, width: self.width,
                    canvas: { fft: [canvases.fft.width, canvases.fft.height], minmax: [canvases.minmax.width, canvases.minmax.height],
                        amp: [canvases.amp.width, canvases.amp.height] },
                    messages: sent.map(p => ({ block_norm: f64hex(p.message.block_norm), gain: p.message.gain, range: p.message.range, n: p.message.n,
                        width: p.message.width, offset: p.message.offset, format: p.message.format, channelMode: p.message.channelMode,
                        waterfall: p.message.waterfall, buffer_bytes: p.message.buffer.byteLength, transfer_is_buffer: p.transfer.length === 1 && p.transfer[0] === p.message.buffer,
                        cmap_len: p.message.cmap.length, cmap_first: p.message.cmap[0], cmap_last: p.message.cmap[p.message.cmap.length - 1],
                        windowc_sha256: crypto.createHash('sha256').update(f64bytes(p.message.windowc)).digest('hex') })),
                    fft_calls: canvases.fft.ctx.calls.filter(k => k[0] === 'putImageData'),
                    minmax_calls: canvases.minmax.ctx.calls, amp_calls: canvases.amp.ctx.calls,
                    c_hist: hist.c_hist, cB_hist: hist.cB_hist, dBfs_min: f64hex(self.dBfs_min), dBfs_max: f64hex(self.dBfs_max), inProcess_after: self.inProcess })
            }
            loud()
            out.runs.push(run)
        }
        fs.writeFileSync(path.join(outdir, 'caller.json'), JSON.stringify(out, null, 0))
    }

    fs.writeFileSync(path.join(outdir, 'provenance.json'), JSON.stringify({
        generated_by: 'oracle/gen_golden.js + oracle/ref_harness.mjs',
        reference: 'triq-org/spectroplot-js lib/worker.js (v1.2.1, /root/reference)',
        node: process.version, v8: process.versions.v8,
    }, null, 1))
}

run().catch(e => { console.error(e); process.exit(1) })
