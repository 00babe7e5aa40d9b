'use strict'
/** Helpers shared by the Node-side tests: golden case loading and reply comparison (mirrors tests/goldenlib.py). */
const fs = require('fs')
const path = require('path')
const crypto = require('crypto')
const siggen = require('../../oracle/js/siggen.js')

const gdir = path.join(__dirname, '..', 'golden')
const spec = JSON.parse(fs.readFileSync(path.join(gdir, 'cases.json'), 'utf8'))
const expected = JSON.parse(fs.readFileSync(path.join(gdir, 'worker_expected.json'), 'utf8'))
const cmapIndex = JSON.parse(fs.readFileSync(path.join(gdir, 'cmaps.json'), 'utf8'))
const cmapBin = fs.readFileSync(path.join(gdir, 'cmaps.bin'))

const sha256 = (u8) => crypto.createHash('sha256').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex')
const f64hex = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b.readBigUInt64LE().toString(16).padStart(16, '0') }
const sameF64 = (v, hex) => (Number.isNaN(v) && /^[7f]ff[0-9a-f]*$/.test(hex) && !/^[7f]ff0{13}$/.test(hex)) || f64hex(v) === hex

function getCmap(c, forceEnds) {
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
    if (forceEnds === undefined ? c.force_ends : forceEnds) { lut[0] = [0, 0, 0]; lut[lut.length - 1] = [255, 255, 255] }
    return lut
}

function makeInput(c) {
    const sw = siggen.SAMPLE_WIDTH[c.gen_format || c.format.toUpperCase()] || 2
    const full = siggen.generate(c.gen_format || c.format, c.gen, Math.ceil(c.bytes / sw), 0)
    const u8 = full.slice(0, c.bytes)
    return u8.buffer.slice(u8.byteOffset, u8.byteOffset + u8.byteLength)
}

function compareReply(r, e) {
    const bad = []
    if (sha256(r.imageData.data) !== e.rgba_sha256) bad.push('rgba')
    for (const k of ['gauge_mins', 'gauge_maxs', 'gauge_amps']) if (Buffer.from(r[k]).toString('hex') !== e[k]) bad.push(k)
    if (JSON.stringify(Array.from(r.c_hist)) !== JSON.stringify(e.c_hist)) bad.push('c_hist')
    const cB = {}
    r.cB_hist.forEach((v, i) => { if (v) cB[i] = v })
    if (JSON.stringify(cB) !== JSON.stringify(e.cB_hist)) bad.push('cB_hist')
    if (!sameF64(r.dBfs_min, e.dBfs_min)) bad.push('dBfs_min')
    if (!sameF64(r.dBfs_max, e.dBfs_max)) bad.push('dBfs_max')
    if (r.offset !== e.offset) bad.push('offset')
    if (!(r.gauge_mins instanceof Uint8ClampedArray) || !(r.imageData.data instanceof Uint8ClampedArray)) bad.push('types')
    return bad
}

module.exports = { spec, expected, getCmap, makeInput, compareReply, sha256, sameF64, gdir }
