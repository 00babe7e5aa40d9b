'use strict'
/**
 * CPU baseline — times the JavaScript oracle (a bit-exact restatement of the reference's lib/worker.js, including the
 * fft_nayuki radix-2 loop and the per-pixel Math.log10) under Node on this host.  TEST/BENCH INFRASTRUCTURE.
 *
 *   node oracle/js/cpu_baseline.js <format> <log2 samples> <n> <window> [seconds=12] [threads=1]
 *
 * Prints one JSON line: {frames_per_s, msamples_per_s, reps, seconds, node, threads}.  W = S/n frames (stride = n).
 * With threads > 1 the capture is cut into `threads` contiguous slices rendered by worker_threads, the reference's
 * own scheme (lib/spectroplot.js:87, :1206-1228).
 */
const path = require('path')
const O = require('./worker_oracle.js')
const siggen = require('./siggen.js')

const GEN = { kind: 'trinoise', seed: 0x5EED0001, step: 7321, gshift: 11, amp: 0.5, namp: 0.02 }

function makeMessage(format, S, n, windowName, t0) {
    const bytes = siggen.generate(format, GEN, S, t0 || 0)
    const { window: windowc, weight } = O.makeWindow(windowName, n)
    const cmap = []
    for (let i = 0; i < 256; i++) cmap.push([i, 255 - i, (i * 3) & 255])
    cmap[0] = [0, 0, 0]; cmap[255] = [255, 255, 255]
    return { block_norm: 1.0 / weight, gain: 6, range: 30, cmap, n, windowc, width: S / n, offset: 0,
        buffer: bytes.buffer, format, channelMode: false, waterfall: false }
}

function runSingle(format, log2s, n, windowName, seconds) {
    const S = 2 ** log2s
    const msg = makeMessage(format, S, n, windowName, 0)
    O.render(msg) // warm-up (JIT)
    let reps = 0
    const t0 = process.hrtime.bigint()
    let dt = 0
    do {
        O.render(msg)
        reps++
        dt = Number(process.hrtime.bigint() - t0) / 1e9
    } while (dt < seconds)
    return { reps, seconds: dt, frames: reps * (S / n), samples: reps * S }
}

const { isMainThread, parentPort, workerData, Worker } = require('worker_threads')

if (!isMainThread) {
    parentPort.postMessage(runSingle(workerData.format, workerData.log2s, workerData.n, workerData.windowName, workerData.secs))
} else if (require.main === module) {
    const [format, log2s, n, windowName, seconds, threads] = process.argv.slice(2)
    const nthreads = parseInt(threads || '1', 10)
    const secs = parseFloat(seconds || '12')
    if (nthreads <= 1) {
        const r = runSingle(format, parseInt(log2s, 10), parseInt(n, 10), windowName, secs)
        console.log(JSON.stringify({ frames_per_s: r.frames / r.seconds, msamples_per_s: r.samples / r.seconds / 1e6, reps: r.reps,
            seconds: r.seconds, node: process.version, threads: 1 }))
    } else {
        let done = 0, frames = 0, samples = 0, maxSeconds = 0
        for (let i = 0; i < nthreads; i++) {
            const w = new Worker(__filename, { workerData: { format, log2s: parseInt(log2s, 10), n: parseInt(n, 10), windowName, secs } })
            w.on('message', (r) => {
                frames += r.frames; samples += r.samples; maxSeconds = Math.max(maxSeconds, r.seconds)
                if (++done === nthreads) {
                    console.log(JSON.stringify({ frames_per_s: frames / maxSeconds, msamples_per_s: samples / maxSeconds / 1e6,
                        reps: null, seconds: maxSeconds, node: process.version, threads: nthreads }))
                }
            })
        }
    }
}
