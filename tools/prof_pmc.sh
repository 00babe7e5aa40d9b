#!/bin/bash
# Collects rocprofv3 PMC counters for the frame-loop kernel in separate passes (gfx950: SQ 8 slots, TCC 4 slots per pass).
#   tools/prof_pmc.sh <tag> [bench args...]      -> gpurun_out/pmc_<tag>/passN/... + gpurun_out/pmc_<tag>_summary.txt
# Run on the GPU box from the repo root. The profiled program is python3 itself (no shell hop behind rocprofv3).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp SP_EXPERIMENT_KNOBS=1
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"
 "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "GRBM_GUI_ACTIVE"
)
# PMC_ONLY="6 7 8": run just these passes (1-based)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi
  timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --no-cpu-baseline --no-e2e --no-rocprof --no-extras "$@" < /dev/null > $OUT/pass$i.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT > $ROOT/gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat $ROOT/gpurun_out/pmc_${TAG}_summary.txt
