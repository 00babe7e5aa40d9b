#!/bin/bash
# Profiling ablations of the frame-loop kernel (SP_DEBUG_FLAGS bits: 1 no butterflies, 2 no LDS exchange, 4 no classification,
# 8 no input loads, 16 no centi-bel atomics, 32 no next-frame touch, 64 no output stores).  Results are INVALID as benchmarks;
# they only locate the bottleneck.
CFG=${1:-cfg2}; shift
for f in ${@:-0 32 64 96 104 127}; do
  SP_DEBUG_FLAGS=$f python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rgba --config $CFG 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=$f', d['kernel'], 'kernel_ms=%.4f' % d['roofline']['kernel_ms'])"
done
