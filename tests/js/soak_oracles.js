'use strict'
/**
 * CPU, container-side: the JavaScript oracle (oracle/js/worker_oracle.js, running on the real V8 Math.cos / sin / log10) renders the
 * seeded requests of a JSON case list and prints one digest per case; tests/test_oracle_soak.py renders the same requests with the C
 * oracle (oracle/sp_oracle.c, whose engine math is the fdlibm restatement oracle/v8math.h) and compares the digests.
 *   node tests/js/soak_oracles.js cases.json > digests.txt
 */
const fs = require('fs')
const crypto = require('crypto')
const O = require('../../oracle/js/worker_oracle.js')
const siggen = require('../../oracle/js/siggen.js')

const cases = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'))
const f64bits = (v) => { const b = Buffer.alloc(8); b.writeDoubleLE(v); return b }
const out = []
for (const c of cases) {
    const gen = { kind: c.kind, seed: c.seed, step: 4099, gshift: 9, amp: c.amp, namp: 0.02 }
    const bytes = siggen.generate(c.fmt, gen, c.samples, 0)
    const { window: windowc, weight } = O.makeWindow(c.win, c.n)
    if (weight === 0) { out.push('skip'); continue }
    const cmap = []
    for (let i = 0; i < c.lut_len; i++) cmap.push([(i * 5) & 255, (i * 11 + 3) & 255, (255 - i) & 255])
    const r = O.render({ block_norm: 1.0 / weight, gain: c.gain, range: c.rng, cmap, n: c.n, windowc, width: c.width, offset: 0,
        buffer: bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength), format: c.fmt, channelMode: c.ch, waterfall: c.wf })
    const h = crypto.createHash('sha256')
    h.update(Buffer.from(r.imageData.data.buffer))
    for (const k of ['gauge_mins', 'gauge_maxs', 'gauge_amps']) h.update(Buffer.from(r[k].buffer))
    h.update(Buffer.from(Float64Array.from(r.c_hist).buffer))
    h.update(Buffer.from(Float64Array.from(r.cB_hist).buffer))
    // NaN ranges compare as NaN, whatever the payload
    h.update(Number.isNaN(r.dBfs_min) ? Buffer.from('nan') : f64bits(r.dBfs_min))
    h.update(Number.isNaN(r.dBfs_max) ? Buffer.from('nan') : f64bits(r.dBfs_max))
    out.push(h.digest('hex'))
}
console.log(out.join('\n'))
