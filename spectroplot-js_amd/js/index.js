'use strict'
/** Node entry point: `const { HipWorker } = require('spectroplot-js_amd/js')` then `new Spectroplot({workerOrUrl: HipWorker, ...})`. */
const { HipWorker, packLut } = require('./hip_worker.js')
const { renderSliced, stripPlacement } = require('./render_file.js')
const params = require('./params.js')
const consumers = require('./consumers.js')
const raster = require('./raster.js')
module.exports = Object.assign({ HipWorker, packLut, renderSliced, stripPlacement }, params, consumers, raster)
