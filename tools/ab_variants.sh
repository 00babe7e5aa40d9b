#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh) on one kernel: rocprofv3 kernel-trace average of the frame-loop kernel
#   [REPS=n] tools/ab_variants.sh "<variant> ..." [config] [kernel] [extra bench args]     ("-" = the library in lib/; default 2 repetitions)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${2:-cfg2}; K=${3:-frames}
cd /tmp && export TMPDIR=/tmp SP_EXPERIMENT_KNOBS=1
for rep in $(seq 1 ${REPS:-2}); do
for v in $1; do
  if [ "$v" = "-" ]; then unset SP_LIB_VARIANT; else export SP_LIB_VARIANT=$v; fi
  if [ "$v" != "-" ] && [ ! -f $ROOT/spectroplot-js_amd/lib/variants/$v.so ]; then echo "variant $v: not built, skipped"; continue; fi
  rm -rf /tmp/ab_$v
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --steps 400 --warmup 100 --config $CFG --kernel $K $4 > /tmp/ab_$v.log 2>&1
  f=$(find /tmp/ab_$v -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "rep $rep $CFG $K variant $v $4: no kernel trace (the run failed: $(tail -c 300 /tmp/ab_$v.log | tr '\n' ' '))"; continue; fi
  kk=$(grep "k_frames" "$f" < /dev/null | head -1 | awk -F, '{print $(NF-4), $(NF-2)}')
  echo "rep $rep $CFG $K variant $v $4: frame-loop avg/min ns = $kk  $(tail -1 /tmp/ab_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],5), d['checks'])" 2>/dev/null)"
done
done
