cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('ms/step %.5f'%d['ms_per_step'],'value %.1f M'%(d['value']/1e6),'kernel_ms %.5f'%r['kernel_ms'],'frac %.4f'%r['frac'],'frac_rotating',r['frac_rotating'])
print(json.dumps(r['rotating'],indent=0))
PY
timeout 600 python -m pytest tests/test_bench_gpu.py -q -m gpu --timeout 400 --timeout-method thread < /dev/null 2>&1 | tail -3
