cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_node_host.py -q -m gpu --timeout 400 --timeout-method thread < /dev/null > gpurun_out/t2.log 2>&1; echo "node rc=$?" >> gpurun_out/t2.log
tail -4 gpurun_out/t2.log
timeout 300 node tools/js_dropin_bench.js < /dev/null 2>&1 | tail -6
for c in cfg5 cfg3; do timeout 900 tools/ab_variants.sh "- nofsync" $c < /dev/null > gpurun_out/ab_$c.log 2>&1; cat gpurun_out/ab_$c.log; done
