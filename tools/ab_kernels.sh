#!/bin/bash
# Same-box A/B of the device kernels: rocprofv3 kernel-trace average of the frame-loop kernel and bench ms_per_step
#   tools/ab_kernels.sh "<kernel> ..." [config] [extra bench args]      kernels: lds frames auto; SP_LIB_VARIANT honoured
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${2:-cfg2}
cd /tmp && export TMPDIR=/tmp SP_EXPERIMENT_KNOBS=1
for rep in 1 2; do
for k in $1; do
  rm -rf /tmp/ab_$k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$k -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --steps 600 --warmup 200 --config $CFG --kernel $k $3 > /tmp/ab_$k.log 2>&1
  f=$(find /tmp/ab_$k -name "*kernel_stats.csv" | head -1)
  kk=$(grep "k_lds_r16\|k_frames" $f | head -1 | awk -F, '{print $(NF-4), $(NF-2)}')
  fin=$(grep "k_finish" $f | head -1 | awk -F, '{print $(NF-4)}')
  ms=$(python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --config $CFG --kernel $k $3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['checks'])")
  echo "rep $rep $CFG kernel $k: frame-loop avg/min ns = $kk  finish avg ns = $fin  ms_per_step frac checks = $ms"
done
done
