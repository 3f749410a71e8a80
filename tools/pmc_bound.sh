#!/bin/bash
# Round 4: which resource bounds rt_render_kernel?  PMC passes by instruction class + LDS pipe, one 8-frame launch each
# (tools/profile_run.py), PMC only (never combined with tracing).  usage: tools/pmc_bound.sh <tag> [scene] [spp]
set -o pipefail
TAG=$1; SCENE=${2:-monkey}; SPP=${3:-1024}
export TMPDIR=/tmp
OUT=gpurun_out/pmcb_$TAG
mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/profile_run.py $SCENE $SPP 1920 1080 8 > $OUT/p$i.log 2> $OUT/p$i.err
  echo "pmc pass $i exit=$? $(tail -1 $OUT/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
t = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    rows = [row for row in csv.DictReader(open(f)) if "rt_render_kernel" in row["Kernel_Name"]]
    last = max(int(row["Dispatch_Id"]) for row in rows) if rows else -1        # (the launch before it is tools/profile_run.py's cost-collecting one)
    for row in rows:
        if int(row["Dispatch_Id"]) == last:
            t[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(t): print("%-28s %.6g" % (k, t[k]))
PY
