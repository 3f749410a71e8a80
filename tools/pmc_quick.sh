#!/bin/bash
# quick PMC pass: tools/pmc_quick.sh <tag> [scene] [spp] ; extra env is inherited
set -o pipefail
TAG=$1; SCENE=${2:-monkey}; SPP=${3:-64}
export TMPDIR=/tmp
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python3 tools/profile_run.py $SCENE $SPP > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/profile_run.py $SCENE $SPP > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
t = collections.defaultdict(float)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    rows = [row for row in csv.DictReader(open(f)) if "rt_render_kernel" in row["Kernel_Name"]]
    last = max(int(row["Dispatch_Id"]) for row in rows) if rows else -1        # (the launch before it is tools/profile_run.py's cost-collecting one)
    for row in rows:
        if int(row["Dispatch_Id"]) == last:
            t[row["Counter_Name"]] += float(row["Counter_Value"])
print("$TAG", open("$OUT/p1.log").read().strip().splitlines()[-1])
print("  waves %d  VALU insts/wave %.3g  SALU/wave %.3g  LDS/wave %.3g" % (t["SQ_WAVES"], t["SQ_INSTS_VALU"]/t["SQ_WAVES"], t["SQ_INSTS_SALU"]/t["SQ_WAVES"], t["SQ_INSTS_LDS"]/t["SQ_WAVES"]))
print("  lane utilisation %.3f" % (t["SQ_THREAD_CYCLES_VALU"]/(t["SQ_ACTIVE_INST_VALU"]*64)))
print("  total VALU insts %.4g  thread-cycles %.4g" % (t["SQ_INSTS_VALU"], t["SQ_THREAD_CYCLES_VALU"]))
print("  of wave cycles: active %.3f wait_any %.3f wait_inst %.3f ; valu active %.3f" % (t["SQ_ACTIVE_INST_ANY"]/t["SQ_WAVE_CYCLES"], t["SQ_WAIT_ANY"]/t["SQ_WAVE_CYCLES"], t["SQ_WAIT_INST_ANY"]/t["SQ_WAVE_CYCLES"], t["SQ_ACTIVE_INST_VALU"]/t["SQ_WAVE_CYCLES"]))
print("  LDS conflict/active %.3f  GUI_ACTIVE %.4g" % (t["SQ_LDS_BANK_CONFLICT"]/max(t["SQ_LDS_IDX_ACTIVE"],1), t["GRBM_GUI_ACTIVE"]))
PY
