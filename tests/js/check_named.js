'use strict'
/**
 * GPU: requests by option names (HipWorker.renderNamed -> sp_render_named) and the command-line renderer built on them.
 *   - golden worker cases rendered by names only equal the real reference worker's replies;
 *   - a repeated named request builds no new plan (planCreations);
 *   - js/cli.js on a generated config-1 capture file writes exactly the image of the golden case cfg1_full, sliced over 1 and 2 workers
 *     (2 workers: against renderSliced's own merge of the array path, whose slices the caller goldens pin).
 */
const fs = require('fs')
const os = require('os')
const path = require('path')
const { execFileSync } = require('child_process')
const G = require('./golden_util.js')
const O = require('../../oracle/js/worker_oracle.js')   // makeWindow only (the array path's taper values are inputs of its message)
const { HipWorker, renderSliced } = require('../../spectroplot-js_amd/js')

async function main() {
    const failures = []
    const worker = new HipWorker()
    let ran = 0
    for (const c of G.spec.worker_cases) {
        const e = G.expected.find(x => x.name === c.name)
        if (!e.reply || c.cmap.startsWith('custom:') || !c.force_ends || c.width === 0) continue
        const r = await worker.renderNamed({ buffer: G.makeInput(c), format: c.format, window: c.window + 'Window', cmap: c.cmap + '_cmap', n: c.n, width: c.width,
            gain: c.gain, range: c.range, channelMode: !!c.channelMode, waterfall: !!c.waterfall, offset: c.offset || 0 })
        const bad = G.compareReply(r, e.reply)
        if (bad.length) failures.push(`${c.name} by name: ${bad.join(',')}`)
        ran++
    }
    if (ran < 60) failures.push(`only ${ran} cases ran`)

    const c1 = G.spec.worker_cases.find(x => x.name === 'cfg1_full')
    const before = worker.planCreations()
    for (let k = 0; k < 3; k++) await worker.renderNamed({ buffer: G.makeInput(c1), format: 'cu8', window: 'hann', cmap: 'cube1', n: 512, width: 2048 })
    if (worker.planCreations() !== before + 1) failures.push(`plans built for three identical named requests: ${worker.planCreations() - before}`)
    worker.terminate()

    // the command-line renderer on a capture file
    const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'spcli-'))
    const capture = path.join(dir, 'g001_433.92M_250k.cu8')
    fs.writeFileSync(capture, Buffer.from(G.makeInput(c1)))
    const cli = path.join(__dirname, '..', '..', 'spectroplot-js_amd', 'js', 'cli.js')
    const e1 = G.expected.find(x => x.name === 'cfg1_full')
    const out1 = path.join(dir, 'one.rgba')
    const log = execFileSync('node', [cli, capture, '--n', '512', '--width', '2048', '--window', 'Hann', '--cmap', 'cube', '--workers', '1', '--out', out1]).toString()
    if (G.sha256(new Uint8Array(fs.readFileSync(out1))) !== e1.reply.rgba_sha256) failures.push('cli.js, 1 worker: image differs from cfg1_full')
    if (!/CU8, centre 433920000 Hz, rate 250000 Hz/.test(log)) failures.push('cli.js: file name parsing: ' + log)
    const out2 = path.join(dir, 'two.rgba')
    execFileSync('node', [cli, capture, '--n', '512', '--width', '2048', '--window', 'hann', '--cmap', 'cube1', '--workers', '2', '--out', out2])
    const { window: windowc, weight } = O.makeWindow('hann', 512)
    const m = await renderSliced({ buffer: G.makeInput(c1), format: 'CU8', n: 512, width: 2048, workers: 2, window: { window: windowc, weight },
        cmap: G.getCmap(c1, false), gain: 6, range: 30 })
    if (G.sha256(new Uint8Array(fs.readFileSync(out2))) !== G.sha256(m.data)) failures.push('cli.js, 2 workers: image differs from the array path')
    const ppm = path.join(dir, 'x.ppm')
    execFileSync('node', [cli, capture, '--n', '512', '--width', '2048', '--window', 'hann', '--workers', '1', '--out', ppm])
    const p = fs.readFileSync(ppm)
    const header = 'P6\n2048 512\n255\n'
    if (p.slice(0, header.length).toString() !== header || p.length !== header.length + 3 * 2048 * 512) failures.push('cli.js: PPM header / size')
    const rgba = fs.readFileSync(out1)
    for (const px of [0, 12345, 2048 * 512 - 1]) for (let k = 0; k < 3; k++) if (p[header.length + 3 * px + k] !== rgba[4 * px + k]) { failures.push('cli.js: PPM pixel ' + px); break }
    // --full: gauges above, dB scale to the right; the spectrogram sits at composePlot's image origin, untouched
    const full = path.join(dir, 'full.rgba')
    const flog = execFileSync('node', [cli, capture, '--n', '512', '--width', '2048', '--window', 'hann', '--workers', '1', '--full', '--out', full]).toString()
    const fm = /\((\d+) x (\d+)\)/.exec(flog)
    const FW = +fm[1], FH = +fm[2], fb = fs.readFileSync(full)
    if (FW !== 2048 + 160 || FH !== 64 + 532 || fb.length !== 4 * FW * FH) failures.push(`cli.js --full: size ${FW} x ${FH}`)
    else {
        for (const [x, y] of [[0, 0], [1000, 300], [2047, 511]]) for (let k = 0; k < 4; k++)
            if (fb[4 * ((y + 64) * FW + x) + k] !== rgba[4 * (y * 2048 + x) + k]) { failures.push(`cli.js --full: image pixel ${x},${y}`); break }
        if (fb[4 * ((64 + 10) * FW + 2048 + 35) + 3] !== 255) failures.push('cli.js --full: colour ramp missing')
    }
    fs.rmdirSync(dir, { recursive: true })

    if (failures.length) { console.log(failures.join('\n')); process.exit(1) }
    console.log(`named requests ok: ${ran} golden cases by name, plan kept across repeats, cli.js image bit-for-bit`)
}
// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main().then(() => process.exit(0), e => { console.error(e); process.exit(1) })
