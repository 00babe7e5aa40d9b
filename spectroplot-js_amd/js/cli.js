#!/usr/bin/env node
'use strict'
/**
 * Headless render of an I/Q capture to an image file through the GPUs — the data half of the reference's processData
 * (lib/spectroplot.js:1096-1285) without a browser:
 *
 *   node spectroplot-js_amd/js/cli.js capture_433.92M_250k.cu8 --n 1024 --width 2048 [--format cu8] [--window blackmanHarris]
 *        [--cmap cube1|viridis|plasma|inferno|magma|hot|afmhot|gist_heat|sox|naive|grayscale|roentgen|phosphor|parabola] [--gain 6] [--range 30] [--workers N] [--waterfall] [--lr]
 *        [--full] --out image.ppm
 *
 * The format defaults to the file extension (lib/parseFreqRate.js:58-70), the worker count to the number of visible GPUs.
 * Output: binary PPM (P6, alpha dropped) or, with --out *.rgba, the raw RGBA bytes exactly as the reference's canvas holds them.
 * --full composes the plot the reference shows around the spectrogram (js/raster.js): amplitude and min/max gauge strips above it, to
 * its right the dB scale (colour ramp, tick marks) with the two histogram outlines over it (filled and stroked without anti-aliasing);
 * only the text labels need a canvas.
 */
const fs = require('fs')
const { renderSliced, parseFormat, parseFreqRate, HipWorker, composePlot, cmapByName } = require('./index.js')

function main(argv) {
    const opt = { n: 512, width: 1024, window: 'blackmanHarris', cmap: 'cube1', gain: 6, range: 30, out: 'spectrogram.ppm' }
    let file = null
    for (let i = 0; i < argv.length; i++) {
        const a = argv[i]
        if (a === '--waterfall') opt.waterfall = true
        else if (a === '--full') opt.full = true
        else if (a === '--lr') opt.channelMode = true
        else if (a.startsWith('--')) opt[a.slice(2)] = argv[++i]
        else file = a
    }
    if (!file) { console.error('usage: cli.js <capture> --n N --width W [options] --out image.ppm'); process.exit(2) }
    const bytes = fs.readFileSync(file)
    const buffer = bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength)
    const format = opt.format || parseFormat(file)
    const n = parseInt(opt.n, 10), width = parseInt(opt.width, 10)
    const fr = parseFreqRate(file)
    const t0 = Date.now()
    return renderSliced({ buffer, format, n, width, workers: opt.workers ? parseInt(opt.workers, 10) : HipWorker.deviceCount(),
        // the option names travel as they are: the library resolves them as the reference's caller does (sp_render_named)
        byName: true, window: String(opt.window), cmap: String(opt.cmap), gain: parseFloat(opt.gain), range: parseFloat(opt.range), channelMode: !!opt.channelMode, waterfall: !!opt.waterfall })
        .then(img => {
            if (opt.full) {
                const cmap = cmapByName(String(opt.cmap)).map(c => c.slice())
                cmap[0] = [0, 0, 0]; cmap[cmap.length - 1] = [255, 255, 255]                    // lib/spectroplot.js:1129-1130
                const plot = composePlot(img, { cmap, gain: parseFloat(opt.gain), range: parseFloat(opt.range), n, waterfall: !!opt.waterfall })
                if (opt.out.endsWith('.rgba')) fs.writeFileSync(opt.out, Buffer.from(plot.surface.data.buffer))
                else fs.writeFileSync(opt.out, plot.surface.toPPM())
                img = { width: plot.surface.width, height: plot.surface.height, dBfs_min: img.dBfs_min, dBfs_max: img.dBfs_max }
            } else if (opt.out.endsWith('.rgba')) {
                fs.writeFileSync(opt.out, Buffer.from(img.data.buffer))
            } else {
                const rgb = Buffer.alloc(img.width * img.height * 3)
                for (let p = 0, q = 0; p < img.data.length; p += 4) { rgb[q++] = img.data[p]; rgb[q++] = img.data[p + 1]; rgb[q++] = img.data[p + 2] }
                fs.writeFileSync(opt.out, Buffer.concat([Buffer.from(`P6\n${img.width} ${img.height}\n255\n`), rgb]))
            }
            console.log(`${file}: ${format}, centre ${fr.freq} Hz, rate ${fr.rate} Hz -> ${opt.out} (${img.width} x ${img.height}), ` +
                `dBfs ${img.dBfs_min.toFixed(2)} .. ${img.dBfs_max.toFixed(2)}, ${Date.now() - t0} ms`)
        })
}

// (an explicit exit: Node 12 can crash while it tears its environment down when finalizers of collected reply buffers are
// still queued - after all output, but with status 139; process.exit() does not take that path)
main(process.argv.slice(2)).then(() => process.exit(0), e => { console.error(e.message || e); process.exit(1) })
