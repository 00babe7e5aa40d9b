for cfg in ${CFGS:-cfg1 cfg3 cfg4 cfg5}; do
  for v in ${VARS:-old -}; do
    if [ "$v" = "-" ]; then unset SP_LIB_VARIANT; else export SP_LIB_VARIANT=$v; fi
    python3 bench.py --no-cpu-baseline --no-e2e --config $cfg --steps 300 --warmup 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$v', 'frames/s %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], d['kernel'], d['checks'])"
  done
done
