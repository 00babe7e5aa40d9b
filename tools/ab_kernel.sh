#!/bin/bash
# Same-box A/B of library variants: rocprofv3 kernel-trace average of the frame-loop kernel and bench ms_per_step for each
#   tools/ab_kernel.sh "<variant> <variant> ..."     ("-" = the library in lib/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in $1; do
  if [ "$v" = "-" ]; then unset SP_LIB_VARIANT; else export SP_LIB_VARIANT=$v; fi
  rm -rf /tmp/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $ROOT/bench.py --no-cpu-baseline --steps 600 --warmup 200 > /tmp/ab_$v.log 2>&1
  f=$(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1)
  k=$(grep "k_lds_r16" $f | head -1 | awk -F, '{print $(NF-4), $(NF-2)}')
  fin=$(grep "k_finish" $f | head -1 | awk -F, '{print $(NF-4)}')
  ms=$(python3 $ROOT/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "rep $rep variant $v: frame-loop avg/min ns = $k  finish avg ns = $fin  ms_per_step (unprofiled) = $ms"
done
done
