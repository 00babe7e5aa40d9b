'use strict'
/**
 * Deterministic synthetic I/Q generator shared by the golden-vector script, the JS oracle tests and
 * the CPU baseline.  TEST/BENCH INFRASTRUCTURE — not part of the product path.
 *
 * Integer-only phase arithmetic + a counter hash, so that the Python (tests/siggen.py) and HIP
 * (csrc/sp_siggen.hip) restatements produce bit-identical bytes:
 *
 *   fmix32(h)          murmur3 finaliser
 *   hash(seed, i)    = fmix32(seed ^ i)
 *   kind 'bytes'     : the buffer is the little-endian u32 words hash(seed, w), w = 0,1,2,...
 *   kind 'trinoise'  : per sample t, component c (0 = I, 1 = Q)
 *        ph   = (t * step) mod 65536,  Q uses (ph - 16384) mod 65536
 *        tri  = (|ph - 32768| - 16384) / 16384            (cos-like triangle wave, exact)
 *        gate = ((t >> gshift) & 1) ? 1 : 0.25
 *        u    = hash(seed, 2t + c) / 2^32 - 0.5
 *        v    = amp * gate * tri + namp * u               (left-to-right f64)
 *      then quantised per format (see the switch in generate()).
 *   kind 'zeros'     : all-zero bytes;  kind 'hexrepeat': gen.hex repeated to fill the buffer.
 */

function fmix32(h) {
    h = h >>> 0
    h ^= h >>> 16
    h = Math.imul(h, 0x85ebca6b) >>> 0
    h ^= h >>> 13
    h = Math.imul(h, 0xc2b2ae35) >>> 0
    h ^= h >>> 16
    return h >>> 0
}

function hash(seed, i) {
    return fmix32((seed ^ i) >>> 0)
}

const SAMPLE_WIDTH = {
    CU4: 1, CS4: 1, CU8: 2, CS8: 2, CU12: 3, CS12: 3, CU16: 4, CS16: 4,
    CU32: 8, CS32: 8, CU64: 16, CS64: 16, CF32: 8, CF64: 16,
}

function value(gen, t, c) {
    const step = gen.step
    let ph = ((t & 0xffff) * step) & 0xffff
    if (c) ph = (ph - 16384) & 0xffff
    const tri = (Math.abs(ph - 32768) - 16384) / 16384
    const gate = ((t >>> gen.gshift) & 1) ? 1 : 0.25
    const u = hash(gen.seed, (2 * t + c) >>> 0) / 4294967296 - 0.5
    return gen.amp * gate * tri + gen.namp * u
}

function clampi(x, lo, hi) { return x < lo ? lo : x > hi ? hi : x }

/** Generates `count` complex samples starting at global sample index t0. Returns a Uint8Array. */
function generate(format, gen, count, t0) {
    format = format.toUpperCase()
    t0 = t0 || 0
    const sw = SAMPLE_WIDTH[format]
    if (!sw) throw new Error('siggen: unknown format ' + format)
    const out = new Uint8Array(count * sw)
    if (gen.kind === 'bytes') {
        // word index is global: byte offset (t0*sw) must be a multiple of 4
        const w0 = (t0 * sw) >>> 2
        const dv = new DataView(out.buffer)
        const words = out.length >>> 2
        for (let w = 0; w < words; w++) dv.setUint32(4 * w, hash(gen.seed, (w0 + w) >>> 0), true)
        for (let b = words * 4; b < out.length; b++) {
            out[b] = (hash(gen.seed, (w0 + words) >>> 0) >>> (8 * (b & 3))) & 0xff
        }
        return out
    }
    if (gen.kind === 'zeros') return out
    if (gen.kind === 'hexrepeat') {
        // the byte pattern gen.hex repeated from buffer offset 0 (t0 must be 0)
        const pat = Buffer.from(gen.hex, 'hex')
        for (let b = 0; b < out.length; b++) out[b] = pat[b % pat.length]
        return out
    }
    if (gen.kind !== 'trinoise') throw new Error('siggen: unknown kind ' + gen.kind)
    const dv = new DataView(out.buffer)
    for (let k = 0; k < count; k++) {
        const t = t0 + k
        const vi = value(gen, t, 0)
        const vq = value(gen, t, 1)
        const o = k * sw
        switch (format) {
        case 'CF32': dv.setFloat32(o, vi, true); dv.setFloat32(o + 4, vq, true); break
        case 'CF64': dv.setFloat64(o, vi, true); dv.setFloat64(o + 8, vq, true); break
        case 'CS16':
            dv.setInt16(o, clampi(Math.floor(vi * 32767 + 0.5), -32768, 32767), true)
            dv.setInt16(o + 2, clampi(Math.floor(vq * 32767 + 0.5), -32768, 32767), true); break
        case 'CU16':
            dv.setUint16(o, clampi(Math.floor(vi * 32767.5 + 32768), 0, 65535), true)
            dv.setUint16(o + 2, clampi(Math.floor(vq * 32767.5 + 32768), 0, 65535), true); break
        case 'CS8':
            dv.setInt8(o, clampi(Math.floor(vi * 127 + 0.5), -128, 127))
            dv.setInt8(o + 1, clampi(Math.floor(vq * 127 + 0.5), -128, 127)); break
        case 'CU8':
            out[o] = clampi(Math.floor(vi * 127.5 + 128), 0, 255)
            out[o + 1] = clampi(Math.floor(vq * 127.5 + 128), 0, 255); break
        case 'CS32':
            dv.setInt32(o, clampi(Math.floor(vi * 2147483647 + 0.5), -2147483648, 2147483647), true)
            dv.setInt32(o + 4, clampi(Math.floor(vq * 2147483647 + 0.5), -2147483648, 2147483647), true); break
        case 'CU32':
            dv.setUint32(o, clampi(Math.floor(vi * 2147483647.5 + 2147483648), 0, 4294967295), true)
            dv.setUint32(o + 4, clampi(Math.floor(vq * 2147483647.5 + 2147483648), 0, 4294967295), true); break
        case 'CS64': case 'CU64': {
            // hi word carries the signal (as CS32 / CU32), lo word is hash noise
            const signed = format === 'CS64'
            const hi = signed
                ? [clampi(Math.floor(vi * 2147483647 + 0.5), -2147483648, 2147483647),
                    clampi(Math.floor(vq * 2147483647 + 0.5), -2147483648, 2147483647)]
                : [clampi(Math.floor(vi * 2147483647.5 + 2147483648), 0, 4294967295),
                    clampi(Math.floor(vq * 2147483647.5 + 2147483648), 0, 4294967295)]
            dv.setUint32(o, hash(gen.seed ^ 0x10101010, (2 * t) >>> 0), true)
            dv.setUint32(o + 4, hi[0] >>> 0, true)
            dv.setUint32(o + 8, hash(gen.seed ^ 0x10101010, (2 * t + 1) >>> 0), true)
            dv.setUint32(o + 12, hi[1] >>> 0, true)
            break
        }
        case 'CS12': case 'CU12': {
            const i12 = (format === 'CS12'
                ? clampi(Math.floor(vi * 2047 + 0.5), -2048, 2047)
                : clampi(Math.floor(vi * 2047.5 + 2048), 0, 4095)) & 0xfff
            const q12 = (format === 'CS12'
                ? clampi(Math.floor(vq * 2047 + 0.5), -2048, 2047)
                : clampi(Math.floor(vq * 2047.5 + 2048), 0, 4095)) & 0xfff
            out[o] = i12 & 0xff
            out[o + 1] = ((i12 >> 8) & 0x0f) | ((q12 & 0x0f) << 4)
            out[o + 2] = (q12 >> 4) & 0xff
            break
        }
        case 'CS4': case 'CU4': {
            const i4 = (format === 'CS4'
                ? clampi(Math.floor(vi * 7 + 0.5), -8, 7)
                : clampi(Math.floor(vi * 7.5 + 8), 0, 15)) & 0xf
            const q4 = (format === 'CS4'
                ? clampi(Math.floor(vq * 7 + 0.5), -8, 7)
                : clampi(Math.floor(vq * 7.5 + 8), 0, 15)) & 0xf
            out[o] = (i4 << 4) | q4
            break
        }
        default: throw new Error('siggen: unhandled format ' + format)
        }
    }
    return out
}

module.exports = { fmix32, hash, generate, SAMPLE_WIDTH }
