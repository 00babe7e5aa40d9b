#!/bin/bash
# Same-box A/B of the finish kernel (and the frame loop) by rocprofv3 --kernel-trace: lib/variants/base.so against the built library,
# base / new / base / new, config 2.      tools/ab_finish.sh        (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in base new base new; do
  if [ $v = base ]; then export SP_EXPERIMENT_KNOBS=1 SP_LIB_VARIANT=base; else unset SP_LIB_VARIANT; fi
  rm -rf /tmp/kt_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python3 $R/bench.py --no-cpu-baseline --no-e2e --no-rocprof --steps 1500 --warmup 200 > /tmp/kt_$v.log 2>&1
  python3 - "$v" "$(find /tmp/kt_$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].split("<")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[2]))}
print("%-5s" % sys.argv[1], "  ".join("%s %.0f ns" % (k.split("::")[-1], v) for k, v in rows.items() if "k_frames" in k or "k_finish" in k))
PY
done
