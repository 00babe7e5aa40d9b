cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base new base new; do
  if [ $v = base ]; then export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=base; else unset SP_LIB_VARIANT; fi
  rm -rf /tmp/kt_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python3 $R/bench.py --no-cpu-baseline --no-e2e --no-rocprof --steps 1500 --warmup 200 > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep -h 'k_frames\|k_finish' $f | awk -F, '{printf "%s %s ns | ", substr($1,1,28), $4}') step $(tail -1 /tmp/kt_$v.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")"
done
