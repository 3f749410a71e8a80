#!/bin/bash
# PMC passes for the beyond-LDS path (hybrid kernel: triangles from L2): is it the texture-address / L1 path?  Small launch (2 spp, one
# frame), few counters per pass, each pass under its own time limit (a first version with six TA counters on an 8-frame launch was silent
# for 7 minutes and got killed).  usage: tools/pmc_vmem.sh <tag> [scene] [spp]
set -o pipefail
TAG=$1; SCENE=${2:-sphere50k}; SPP=${3:-2}
export TMPDIR=/tmp
OUT=gpurun_out/pmcv_$TAG
mkdir -p $OUT
i=0
for set in "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/profile_run.py $SCENE $SPP 1920 1080 1 > $OUT/p$i.log 2> $OUT/p$i.err
  rc=$?
  echo "pmc pass $i exit=$rc $(tail -1 $OUT/p$i.log)"
  if [ $rc -ne 0 ]; then echo "stopping after a failed pass"; break; fi
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
for k in sorted(t): print("%-40s %.6g" % (k, t[k]))
PY
