'use strict'
/**
 * Pins the JavaScript oracle (worker_oracle.js) against the vectors produced by the real reference worker
 * (tests/golden/, see oracle/gen_golden.js).  TEST INFRASTRUCTURE.
 *
 *   node oracle/js/check_golden.js [golden_dir]     exit code 0 = every vector reproduced bit-for-bit
 */
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const O = require('./worker_oracle.js')
const siggen = require('./siggen.js')

const gdir = path.resolve(process.argv[2] || path.join(__dirname, '../../tests/golden'))
const spec = JSON.parse(fs.readFileSync(path.join(gdir, 'cases.json'), 'utf8'))
const expected = JSON.parse(fs.readFileSync(path.join(gdir, 'worker_expected.json'), 'utf8'))
const cmapIndex = JSON.parse(fs.readFileSync(path.join(gdir, 'cmaps.json'), 'utf8'))
const cmapBin = fs.readFileSync(path.join(gdir, 'cmaps.bin'))

const sha256 = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex')
const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.readBigUInt64LE().toString(16).padStart(16, '0') }
const sameF64 = (v, hex) => (Number.isNaN(v) && /^[7f]ff[0-9a-f]*$/.test(hex) && !/^[7f]ff0{13}$/.test(hex)) || f64hex(v) === hex

function getCmap(c) {
    let lut
    if (c.cmap.startsWith('custom:')) {
        const len = parseInt(c.cmap.split(':')[1], 10)
        lut = []
        for (let i = 0; i < len; i++) lut.push([(i * 7) & 255, (i * 13 + 5) & 255, (255 - i) & 255])
    } else {
        const e = cmapIndex.find(x => x.name === c.cmap + '_cmap')
        lut = []
        for (let i = 0; i < e.length; i++) lut.push([cmapBin[e.offset + 3 * i], cmapBin[e.offset + 3 * i + 1], cmapBin[e.offset + 3 * i + 2]])
    }
    if (c.force_ends) { lut[0] = [0, 0, 0]; lut[lut.length - 1] = [255, 255, 255] }
    return lut
}

function makeInput(c) {
    const sw = siggen.SAMPLE_WIDTH[c.gen_format || c.format.toUpperCase()] || 2
    const full = siggen.generate(c.gen_format || c.format, c.gen, Math.ceil(c.bytes / sw), 0)
    const u8 = full.slice(0, c.bytes)
    return u8.buffer.slice(u8.byteOffset, u8.byteOffset + u8.byteLength)
}

let failures = 0
const fail = (name, what) => { failures++; if (failures < 40) console.error(`FAIL ${name}: ${what}`) }

function compareReply(name, r, e) {
    if (sha256(r.imageData.data) !== e.rgba_sha256) fail(name, 'rgba')
    if (Buffer.from(r.gauge_mins).toString('hex') !== e.gauge_mins) fail(name, 'gauge_mins')
    if (Buffer.from(r.gauge_maxs).toString('hex') !== e.gauge_maxs) fail(name, 'gauge_maxs')
    if (Buffer.from(r.gauge_amps).toString('hex') !== e.gauge_amps) fail(name, 'gauge_amps')
    if (JSON.stringify(Array.from(r.c_hist)) !== JSON.stringify(e.c_hist)) fail(name, 'c_hist')
    const cB = {}
    r.cB_hist.forEach((v, i) => { if (v) cB[i] = v })
    if (JSON.stringify(cB) !== JSON.stringify(e.cB_hist)) fail(name, 'cB_hist')
    if (!sameF64(r.dBfs_min, e.dBfs_min)) fail(name, `dBfs_min ${r.dBfs_min} vs ${e.dBfs_min_num}`)
    if (!sameF64(r.dBfs_max, e.dBfs_max)) fail(name, `dBfs_max ${r.dBfs_max} vs ${e.dBfs_max_num}`)
    if (r.offset !== e.offset) fail(name, 'offset')
}

let checked = 0
for (const c of spec.worker_cases) {
    const e = expected.find(x => x.name === c.name)
    const buffer = makeInput(c)
    if (sha256(new Uint8Array(buffer)) !== e.input_sha256) { fail(c.name, 'input generator'); continue }
    const { window: windowc, weight } = O.makeWindow(c.window, c.n)
    const cmap = getCmap(c)
    if (f64hex(1.0 / weight) !== e.block_norm) fail(c.name, 'block_norm')
    const msg = { block_norm: 1.0 / weight, gain: c.gain, range: c.range, cmap, n: c.n, windowc, width: c.width, offset: c.offset || 0,
        buffer, format: c.format, channelMode: !!c.channelMode, waterfall: !!c.waterfall }
    if (e.throws) {
        let threw = null
        try { O.render(msg) } catch (err) { threw = err }
        if (!threw) fail(c.name, 'expected a throw: ' + e.throws)
    } else if (e.reply) {
        compareReply(c.name, O.render(msg), e.reply)
    } else {
        const m = O.renderSliced({ buffer, format: c.format, n: c.n, width: c.width, workers: c.slices, windowc, weight, cmap,
            gain: c.gain, range: c.range, channelMode: !!c.channelMode, waterfall: !!c.waterfall })
        m.replies.forEach((r, i) => compareReply(`${c.name}[${i}]`, r, e.slices[i]))
        if (sha256(m.data) !== e.merged.rgba_sha256) fail(c.name, 'merged rgba')
        if (JSON.stringify(m.c_hist) !== JSON.stringify(e.merged.c_hist)) fail(c.name, 'merged c_hist')
        if (!sameF64(m.dBfs_min, e.merged.dBfs_min) || !sameF64(m.dBfs_max, e.merged.dBfs_max)) fail(c.name, 'merged min/max')
        if (m.sliceWidth !== e.merged.slice_width) fail(c.name, 'slice width')
    }
    checked++
}

// windows
{
    const idx = JSON.parse(fs.readFileSync(path.join(gdir, 'windows.json'), 'utf8'))
    const bin = fs.readFileSync(path.join(gdir, 'windows.bin'))
    for (const e of idx) {
        const { window, weight } = O.makeWindow(e.name, e.n)
        const b = Buffer.alloc(8 * e.n); window.forEach((v, i) => b.writeDoubleLE(v, 8 * i))
        if (crypto.createHash('sha256').update(b).digest('hex') !== e.sha256) fail(`window ${e.name} ${e.n}`, 'values')
        if (!sameF64(weight, e.weight)) fail(`window ${e.name} ${e.n}`, 'weight')
        if (e.offset >= 0 && Buffer.compare(b, bin.slice(e.offset, e.offset + b.length)) !== 0) fail(`window ${e.name} ${e.n}`, 'kept values')
        checked++
    }
}
// twiddles + FFT
{
    const F = JSON.parse(fs.readFileSync(path.join(gdir, 'fft.json'), 'utf8'))
    for (const t of F.twiddles) {
        const p = O.fftPlan(t.n)
        const b = Buffer.concat([Buffer.from(p.cos.buffer), Buffer.from(p.sin.buffer)])
        if (crypto.createHash('sha256').update(b).digest('hex') !== t.sha256) fail(`twiddles ${t.n}`, 'values')
        checked++
    }
    for (const k of F.cases) {
        const n = k.n
        const re = new Float64Array(n), im = new Float64Array(n)
        if (k.kind === 'impulse') { re[k.pos] = 1.0; im[k.pos] = -0.5 } else if (k.kind === 'dc') { re.fill(0.75); im.fill(-0.25) } else {
            for (let i = 0; i < n; i++) {
                re[i] = siggen.hash(k.seed, 2 * i) / 2147483648 - 1.0
                im[i] = siggen.hash(k.seed, 2 * i + 1) / 2147483648 - 1.0
            }
        }
        O.fftInPlace(O.fftPlan(n), re, im)
        if (k.split) O.splitReal(n, re, im)
        const b = Buffer.concat([Buffer.from(re.buffer), Buffer.from(im.buffer)])
        if (crypto.createHash('sha256').update(b).digest('hex') !== k.sha256) fail(`fft n=${n} ${k.kind}`, 'values')
        checked++
    }
    let thrown = null
    try { O.fftPlan(12) } catch (e) { thrown = { type: typeof e, value: String(e) } }
    if (JSON.stringify(thrown) !== JSON.stringify(F.non_pow2_throw)) fail('fft non-pow2', 'throw value')
}
// decode
{
    const D = JSON.parse(fs.readFileSync(path.join(gdir, 'decode.json'), 'utf8'))
    for (const k of spec.decode_kat) {
        const e = D.find(x => x.name === k.name)
        const sw = siggen.SAMPLE_WIDTH[k.gen_format] || 2
        const bytes = k.hex ? new Uint8Array(Buffer.from(k.hex, 'hex'))
            : siggen.generate(k.gen_format, { kind: 'bytes', seed: k.seed }, Math.ceil(k.bytes / sw), 0).slice(0, k.bytes)
        let sv
        try { sv = O.openSamples(k.format, bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength)) } catch (err) {
            if (!e.throws) fail(k.name, 'unexpected throw ' + err)
            checked++; continue
        }
        if (e.throws) { fail(k.name, 'expected throw'); continue }
        if (sv.sampleWidth !== e.sampleWidth || !sameF64(sv.sampleCount, e.sampleCount)) fail(k.name, 'width/count')
        let vi = 0
        for (let pos = k.pos_lo; pos < k.pos_hi; pos++) {
            if (!sameF64(sv.I(pos), e.values[vi++])) fail(k.name, `I(${pos}) = ${sv.I(pos)}`)
            if (!sameF64(sv.Q(pos), e.values[vi++])) fail(k.name, `Q(${pos}) = ${sv.Q(pos)}`)
        }
        checked++
    }
}

if (failures) { console.error(`${failures} mismatches`); process.exit(1) }
console.log(`js oracle reproduces all ${checked} golden vectors bit-for-bit`)
