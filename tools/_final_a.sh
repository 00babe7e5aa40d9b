cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r04
timeout 1700 tools/collect_profiles.sh r04 "cfg2 cfg1" < /dev/null > gpurun_out/collect_a.log 2>&1
tail -c 800 gpurun_out/collect_a.log
timeout 120 /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/op_cost.hip -o /tmp/op_cost < /dev/null && timeout 300 /tmp/op_cost < /dev/null > gpurun_out/profiles_r04/r04_op_cost.txt 2>&1
head -8 gpurun_out/profiles_r04/r04_op_cost.txt
export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=stamps
for c in cfg2 cfg3 cfg4 cfg5; do timeout 240 python3 tools/stamps.py $c < /dev/null >> gpurun_out/profiles_r04/r04_stamps.txt 2>&1; done
unset SP_LIB_VARIANT SP_EXPERIMENT_KNOBS
timeout 900 python3 tests/soak_gpu.py 1500 20261004 < /dev/null > gpurun_out/profiles_r04/r04_soak.txt 2>&1; tail -2 gpurun_out/profiles_r04/r04_soak.txt
ls gpurun_out/profiles_r04/
