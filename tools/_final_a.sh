cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profiles_r04; mkdir -p gpurun_out/profiles_r04
timeout 3000 tools/collect_profiles.sh r04 "cfg2 cfg1 cfg3 cfg4 cfg5" < /dev/null > gpurun_out/collect_a.log 2>&1
export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=stamps
for c in cfg2 cfg3 cfg4 cfg5; do timeout 240 python3 tools/stamps.py $c < /dev/null >> gpurun_out/profiles_r04/r04_stamps.txt 2>&1; done
unset SP_LIB_VARIANT SP_EXPERIMENT_KNOBS
ls gpurun_out/profiles_r04/
