'use strict'
/**
 * Parameter helpers on the caller's side of the worker message — the "next" rows of SURVEY.md §8(f):
 *   lookup            lib/utils.js:25-40        exact -> case-insensitive -> prefix key match
 *   windowByName      lib/windows.js:14-88 through the C ABI (sp_window, bit-identical tapers)
 *   computed LUTs     lib/soxcmap.js:12-49, lib/naivecmap.js:13-81 (sox, naive, grayscale, roentgen, phosphor)
 *   parseFormat       lib/parseFreqRate.js:58-70   file name -> FORMAT string
 *   parseFreqRate     lib/parseFreqRate.js:16-55   file name -> {freq, rate}
 * The table-defined colour maps of the reference (viridis, cube1, ...) are data files of that project and are not
 * reproduced here; pass them as arrays, exactly as the worker message carries them.
 */
const path = require('path')

function addon() { return require(path.join(__dirname, '..', 'lib', 'spectroplot_hip.node')) }

/** Returns table[key] for an exact, then case-insensitive, then prefix match; non-strings pass through. */
function lookup(table, arrayOrKey) {
    if (!arrayOrKey || typeof arrayOrKey !== 'string') return arrayOrKey
    if (table[arrayOrKey]) return table[arrayOrKey]
    const want = arrayOrKey.toLowerCase()
    const keys = Object.keys(table)
    for (const k of keys) if (k.toLowerCase() === want) return table[k]
    for (const k of keys) if (k.toLowerCase().startsWith(want)) return table[k]
    return null
}

// Key order matters for prefix matches: the reference looks keys up in an ES module namespace object, whose keys are
// sorted (so 'black' and even 'blackman' resolve to blackmanHarrisWindow, 'b' to bartlettWindow, 'ha' to hammingWindow).
const WINDOW_NAMES = ['bartlett', 'blackmanHarris', 'blackman', 'hamming', 'hann', 'rectangular']
/** The reference's `windows` module as a table of generator functions, evaluated by the native library. */
const windows = {}
for (const name of WINDOW_NAMES) {
    windows[name + 'Window'] = (n) => { const w = addon().window(name, n); return { window: Array.from(w.window), weight: w.weight } }
}
/** `lookup(windows, value) || windows.blackmanHarrisWindow` (lib/spectroplot.js:241) */
function windowByName(value) { return lookup(windows, value) || windows.blackmanHarrisWindow }

function soxCmap(stops) {
    stops = stops || 256
    const out = []
    for (let i = 0; i < stops; ++i) {
        const x = i / (stops - 1.0)
        const r = x < .13 ? 0 : x < .73 ? 1 * Math.sin((x - .13) / .60 * Math.PI / 2) : 1
        const g = x < .60 ? 0 : x < .91 ? 1 * Math.sin((x - .60) / .31 * Math.PI / 2) : 1
        const b = x < .60 ? .5 * Math.sin((x - .00) / .60 * Math.PI) : x < .78 ? 0 : (x - .78) / .22
        out.push([Math.round(255 * r), Math.round(255 * g), Math.round(255 * b)])
    }
    return out
}

function naiveCmap(stops) {
    stops = stops || 256
    const out = []
    for (let i = 0; i < stops; ++i) {
        let r, g, b
        if (i < stops / 4) { b = i * 128 / (stops / 4); g = 0; r = 0 } else if (i < stops / 2) { b = 256 - i / 2; g = 0; r = i - stops / 4 } else if (i < stops * 3 / 4) { b = 0; g = i - stops / 2; r = 255 } else { b = i - stops * 3 / 4; g = 255; r = 255 }
        out.push([~~r, ~~g, ~~b])
    }
    return out
}

function grayscaleCmap(stops) {
    stops = stops || 256
    const out = []
    for (let i = 0; i < stops; ++i) { const c = ~~(i * 255 / stops); out.push([c, c, c]) }
    return out
}

function roentgenCmap(stops) {
    stops = stops || 256
    const out = []
    for (let i = 0; i < stops; ++i) { const c = ~~(255 - (i * 255 / stops)); out.push([c, c, c]) }
    return out
}

function phosphorCmap(stops) {
    stops = stops || 256
    const out = []
    for (let i = 0; i < stops; ++i) {
        let r, g, b
        if (i < stops / 2) { r = 0; g = i * 191 / (stops / 2); b = 0 } else {
            r = (i - stops / 2) * 255 / (stops / 2); g = 191 + (i - stops / 2) * 64 / (stops / 2); b = (i - stops / 2) * 255 / (stops / 2)
        }
        out.push([~~r, ~~g, ~~b])
    }
    return out
}

/** The computed colour maps under the reference's key names (…_cmap); extend with your own tables and use `lookup`. */
const computedCmaps = {
    sox_cmap: soxCmap(), grayscale_cmap: grayscaleCmap(), naive_cmap: naiveCmap(), phosphor_cmap: phosphorCmap(), roentgen_cmap: roentgenCmap(),
}
/** Key order of the reference's merged colour-map table (lib/spectroplot.js:41), for callers that add the table-defined maps. */
const CMAP_KEY_ORDER = ['cube1_cmap', 'sox_cmap', 'grayscale_cmap', 'naive_cmap', 'phosphor_cmap', 'roentgen_cmap', 'afmhot_cmap', 'gist_heat_cmap',
    'hot_cmap', 'inferno_cmap', 'magma_cmap', 'plasma_cmap', 'viridis_cmap', 'parabola_cmap']

/** File name -> upper-cased extension, '?' without one (lib/parseFreqRate.js:58-70). */
function parseFormat(name) {
    if (!name || typeof name !== 'string') return '?'
    const pos = name.lastIndexOf('.')
    return pos !== -1 ? name.substr(pos + 1).toUpperCase() : '?'
}

/** File name -> {freq, rate}: numbers after a separator [-_ .], suffixed M -> centre frequency, k -> sample rate. */
function parseFreqRate(name) {
    if (!name || typeof name !== 'string') return { freq: 0, rate: 0 }
    const slash = name.lastIndexOf('/')
    if (slash !== -1) name = name.substr(slash + 1)
    let freq = 0, rate = 1
    for (let p = 0; p < name.length - 1; ++p) {
        const ch = name[p]
        if (ch !== '_' && ch !== '-' && ch !== ' ' && ch !== '.') continue
        ++p
        const f = parseFloat(name.substr(p))
        if (isNaN(f)) continue
        while (p < name.length && ((name[p] >= '0' && name[p] <= '9') || name[p] === '.')) ++p
        if (name[p] === 'M' || name[p] === 'm') freq = f * 1000000.0
        if (name[p] === 'k' || name[p] === 'K') rate = f * 1000.0
    }
    return { freq, rate }
}

module.exports = { lookup, windows, windowByName, computedCmaps, CMAP_KEY_ORDER, soxCmap, naiveCmap, grayscaleCmap, roentgenCmap, phosphorCmap, parseFormat,
    parseFreqRate }
