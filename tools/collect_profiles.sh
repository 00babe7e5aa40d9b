#!/bin/bash
# Collects, on the GPU box, what profiles/ holds per BASELINE config: rocprofv3 kernel-trace summary, PMC summary (separate passes),
# HBM traffic per launch and the bench line itself.
#   tools/collect_profiles.sh <tag e.g. r02> "<configs e.g. cfg2 cfg1 cfg3 cfg4 cfg5>"
# Results land in gpurun_out/profiles_<tag>/ (copy the files to profiles/ and commit them).
TAG=$1; CFGS=${2:-"cfg2"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
  STEPS="--steps 400 --warmup 100"
  # 1. counters, one pass per counter group (tools/prof_pmc.sh), then the traffic figure bench.py quotes
  $ROOT/tools/prof_pmc.sh ${TAG}_$C --config $C $STEPS > /dev/null 2>&1
  cp $ROOT/gpurun_out/pmc_${TAG}_${C}_summary.txt $OUT/${TAG}_${C}_pmc_summary.txt
  rm -rf $ROOT/gpurun_out/pmc_${TAG}_${C}   # the raw per-dispatch CSVs are large
  python3 - "$OUT/${TAG}_${C}_pmc_summary.txt" "$C" > $OUT/${TAG}_${C}_traffic.json <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
blk = [b for b in txt.split("== ") if "k_frames" in b or "k_scratch" in b]
blk = max(blk, key=lambda b: float(re.search(r"SQ_WAVE_CYCLES\s+n=\s*\d+\s+mean=\s*([\d.]+)", b).group(1)) if "SQ_WAVE_CYCLES" in b else 0)
name = blk.splitlines()[0]
g = lambda k: float(re.search(k + r"\s+n=\s*\d+\s+mean=\s*([\d.]+)", blk).group(1))
fetch, write = g("FETCH_SIZE"), g("WRITE_SIZE")
print(json.dumps({"config": sys.argv[2], "kernel": name,
  "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_pmc.sh), mean over the second half of the launches",
  "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "fetch_correction": 2.0,
  "correction_note": "gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md, HBM): doubled here; WRITE_SIZE is exact for 16-byte streaming stores",
  "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0}, indent=1))
PY
  mkdir -p $ROOT/profiles && cp $OUT/${TAG}_${C}_traffic.json $ROOT/profiles/${TAG}_${C}_traffic.json
  # the instruction mix bench.py's roofline_valu object is computed from (counted wave-instructions per launch)
  python3 - "$OUT/${TAG}_${C}_pmc_summary.txt" "$C" > $OUT/${TAG}_${C}_valu.json <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
blk = [b for b in txt.split("== ") if "k_frames" in b or "k_scratch" in b]
blk = max(blk, key=lambda b: float(re.search(r"SQ_WAVE_CYCLES\s+n=\s*\d+\s+mean=\s*([\d.]+)", b).group(1)) if "SQ_WAVE_CYCLES" in b else 0)
def g(k):
    m = re.search(k + r"\s+n=\s*\d+\s+mean=\s*([\d.]+)", blk)
    return float(m.group(1)) if m else None
keys = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_TRANS_F32",
        "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_SALU",
        "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
print(json.dumps({"config": sys.argv[2], "kernel": blk.splitlines()[0],
  "source": "rocprofv3 --pmc (tools/prof_pmc.sh, separate passes), wave-instructions per launch, mean over the second half of the launches",
  "counters": {k: g(k) for k in keys}}, indent=1))
PY
  cp $OUT/${TAG}_${C}_valu.json $ROOT/profiles/${TAG}_${C}_valu.json
  # 2. kernel durations
  rm -rf /tmp/kt_$C
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$C -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-rocprof --no-extras --config $C < /dev/null > /tmp/kt_$C.log 2>&1
  KS=$(find /tmp/kt_$C -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$KS" ] && cp "$KS" $OUT/${TAG}_${C}_kernel_stats.csv
  # 3. the bench line (un-profiled); the CPU baseline and the Node drop-in leg ride with the headline config
  if [ "$C" = "cfg2" ]; then timeout 900 python3 $ROOT/bench.py --config $C < /dev/null > $OUT/${TAG}_${C}_bench.json 2> /tmp/bench_$C.err
  else timeout 900 python3 $ROOT/bench.py --config $C --no-e2e < /dev/null > $OUT/${TAG}_${C}_bench.json 2> /tmp/bench_$C.err; fi
  tail -c 600 $OUT/${TAG}_${C}_bench.json; echo
done
