#!/bin/bash
# A/B on one box: rocprofv3 averages of the frame-loop and finish kernels
#   tools/ab_spec.sh "<configs>" "<kernel[:variant]> ..."     e.g. "cfg2 cfg4" "frames spec spec:s_nocand"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp SP_EXPERIMENT_KNOBS=1
for rep in 1 2; do for c in $1; do for kv in $2; do
  k=${kv%%:*}; v=${kv#*:}; [ "$v" = "$kv" ] && v=""
  if [ -n "$v" ]; then export SP_LIB_VARIANT=$v; else unset SP_LIB_VARIANT; fi
  rm -rf /tmp/ab_x
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_x -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-rocprof --steps 400 --warmup 100 --config $c --kernel $k > /tmp/ab_x.log 2>&1
  f=$(find /tmp/ab_x -name "*kernel_stats.csv" | head -1)
  echo "rep $rep $c $kv: frame-loop avg ns $(grep k_frames $f | head -1 | awk -F, '{print $(NF-4)}') finish $(grep k_finish $f | head -1 | awk -F, '{print $(NF-4)}')"
done; done; done
