cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T="-p pytest_timeout --timeout 300 --timeout-method thread"
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_node_host.py tests/test_sharding_gloo.py -q -m gpu $T < /dev/null > gpurun_out/t2.log 2>&1; echo "subset rc=$?" >> gpurun_out/t2.log
tail -8 gpurun_out/t2.log
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
    print('value',d['value'],'ms/step',d['ms_per_step'],'cold',d['cold'])
    print('kernel_ms',d['roofline']['kernel_ms'],'frac',d['roofline']['frac'],'rot',d['roofline']['rotating'])
    print('two',d['two_in_flight']); print('e2e',d.get('e2e'))
except Exception as e: print('bench parse failed',e)
PY
timeout 120 node tools/js_small_message.js $GRAFT_REPO_ROOT < /dev/null 2>&1 | tail -2
for c in cfg5 cfg3; do timeout 900 tools/ab_variants.sh "- nofsync" $c < /dev/null > gpurun_out/ab_$c.log 2>&1; cat gpurun_out/ab_$c.log; done
timeout 400 tools/ab_variants.sh "base -" cfg2 < /dev/null > gpurun_out/ab_cfg2.log 2>&1; cat gpurun_out/ab_cfg2.log
export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=stamps
timeout 200 python3 tools/stamps.py cfg2 < /dev/null > gpurun_out/stamps_cfg2.txt 2>&1
grep -E "whole wave|prologue|tail" gpurun_out/stamps_cfg2.txt | cut -c1-200
