'use strict'
/**
 * Parameter helpers on the caller's side of the worker message — the "next" rows of SURVEY.md §8(f):
 *   lookup          key resolution of the reference's option tables (lib/utils.js:25-40)
 *   windows         the reference's `windows` module as a table of generators, evaluated by the native library (sp_window,
 *                   bit-identical to lib/windows.js:14-88)
 *   cmaps           the reference's merged colour-map table (lib/spectroplot.js:41), served by the native library (sp_cmap):
 *                   every map as the reference's modules evaluate it, under the reference's keys and key order
 *   parseFormat     file name -> FORMAT string (lib/parseFreqRate.js:58-70)
 *   parseFreqRate   file name -> {freq, rate} (lib/parseFreqRate.js:16-55)
 */
const path = require('path')

function addon() { return require(path.join(__dirname, '..', 'lib', 'spectroplot_hip.node')) }

// The three match grades of the reference's lookup, strongest first; a key wins by grade, then by table order.
const MATCH_GRADES = [
    (key, want) => key === want,
    (key, want) => key.toLowerCase() === want.toLowerCase(),
    (key, want) => key.toLowerCase().startsWith(want.toLowerCase()),
]

/** table[key] for the best-matching key (exact > same letters > prefix); anything that is not a non-empty string passes through. */
function lookup(table, nameOrValue) {
    if (typeof nameOrValue !== 'string' || nameOrValue === '') return nameOrValue
    const keys = Object.keys(table)
    for (const grade of MATCH_GRADES) {
        const hit = keys.find(key => grade(key, nameOrValue))
        if (hit !== undefined && table[hit]) return table[hit]
    }
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

/** Key order of the reference's merged colour-map table. */
const CMAP_KEY_ORDER = ['cube1_cmap', 'sox_cmap', 'grayscale_cmap', 'naive_cmap', 'phosphor_cmap', 'roentgen_cmap', 'afmhot_cmap', 'gist_heat_cmap',
    'hot_cmap', 'inferno_cmap', 'magma_cmap', 'plasma_cmap', 'viridis_cmap', 'parabola_cmap']

/** One colour map as the worker message carries it: an array of [r, g, b] (ends not forced yet). */
function cmapEntries(key) {
    const flat = addon().cmap(key)
    if (!flat) return null
    const out = new Array(flat.length / 3)
    for (let i = 0; i < out.length; i++) out[i] = [flat[3 * i], flat[3 * i + 1], flat[3 * i + 2]]
    return out
}

/** The merged colour-map table; maps are fetched from the native library on first use. */
const cmaps = {}
for (const key of CMAP_KEY_ORDER) {
    Object.defineProperty(cmaps, key, {
        enumerable: true, configurable: true,
        get() { const v = cmapEntries(key); Object.defineProperty(cmaps, key, { value: v, enumerable: true }); return v },
    })
}
/** `lookup(cmaps, value) || cmaps.cube1_cmap` (lib/spectroplot.js:252-264) */
function cmapByName(value) { return lookup(cmaps, value) || cmaps.cube1_cmap }

/** File name -> the text after its last dot, upper-cased; '?' when there is no dot or no name. */
function parseFormat(name) {
    const m = typeof name === 'string' ? /\.([^.]*)$/.exec(name) : null
    return m ? m[1].toUpperCase() : '?'
}

const SEPARATOR = /[-_ .]/g
const DIGITS_AND_DOTS = /[0-9.]*/y
/**
 * Number tokens of a file name as the reference reads them: a separator, then whatever parseFloat accepts; the character behind the
 * number's digits and dots is its unit.  Characters are consumed left to right: the character behind a separator that starts no
 * number is skipped with it, and so is every unit character (so neither can act as a separator itself), and the last character
 * of the name never starts a token.
 */
function* numberTokens(name) {
    let pos = 0
    while (pos < name.length - 1) {
        SEPARATOR.lastIndex = pos
        const sep = SEPARATOR.exec(name)
        if (!sep || sep.index >= name.length - 1) return
        const start = sep.index + 1
        const value = parseFloat(name.slice(start))
        if (Number.isNaN(value)) { pos = start + 1; continue }
        DIGITS_AND_DOTS.lastIndex = start
        DIGITS_AND_DOTS.exec(name)
        const unitAt = DIGITS_AND_DOTS.lastIndex
        yield { value, unit: name[unitAt] }
        pos = unitAt + 1
    }
}

/** File name -> {freq, rate}: the last number with unit M (MHz) is the centre frequency, the last with unit k (kHz) the sample rate. */
function parseFreqRate(name) {
    if (!name || typeof name !== 'string') return { freq: 0, rate: 0 }
    const base = name.slice(name.lastIndexOf('/') + 1)
    const found = { freq: 0, rate: 1 }
    for (const { value, unit } of numberTokens(base)) {
        if (unit === 'M' || unit === 'm') found.freq = value * 1000000.0
        if (unit === 'k' || unit === 'K') found.rate = value * 1000.0
    }
    return found
}

module.exports = { lookup, windows, windowByName, cmaps, cmapByName, cmapEntries, CMAP_KEY_ORDER, parseFormat, parseFreqRate }
