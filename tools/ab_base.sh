#!/bin/bash
# Same-box A/B of the built library against lib/variants/base.so (a build of another revision): tools/ab_base.sh cfg3 cfg5 ...
# ms per step and the frame loop's HIP-event time, base / new / base / new per config.
for cfg in "$@"; do
  for v in base new base new; do
    if [ $v = base ]; then export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=base; else unset SP_LIB_VARIANT; fi
    r=$(timeout 300 python3 bench.py --config $cfg --steps 150 --warmup 30 --no-cpu-baseline --no-e2e --no-rocprof 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  kernel(event) %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms_event_pair']))")
    echo "$cfg $v $r"
  done
done
