for cfg in cfg3 cfg5 cf32_2048 cf32_4096; do
  for v in base new base new; do
    if [ $v = base ]; then export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=base; else unset SP_LIB_VARIANT; fi
    r=$(timeout 300 python3 bench.py --config $cfg --steps 150 --warmup 30 --no-cpu-baseline --no-e2e --no-rocprof 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  kernel(event) %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms_event_pair']))")
    echo "$cfg $v $r"
  done
done
