cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x --timeout 300 --timeout-method thread < /dev/null > gpurun_out/t1.log 2>&1; echo "parity rc=$?" >> gpurun_out/t1.log; tail -2 gpurun_out/t1.log
for c in cfg2 cfg4 cfg3; do timeout 600 tools/ab_variants.sh "cur -" $c < /dev/null > gpurun_out/ab_$c.log 2>&1; cat gpurun_out/ab_$c.log; done
