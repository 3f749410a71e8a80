#!/bin/bash
# rocprofv3 passes for the render kernel at the bench configuration (run on the GPU box from the
# repo root).  usage: tools/profile_all.sh <tag>
#   pass 0: --kernel-trace --stats on the default bench.py workload (8 frames per launch; warm-up of the
#           same size and no extra legs, so that every launch of the kernel is of the kind that is timed)
#   pass 1-3: SQ counters, pass 4/5: FETCH_SIZE / WRITE_SIZE, each in its own run (PMC and
#   tracing are never combined), on one launch of the same workload (tools/profile_run.py: the
#   8-frame launch of the default bench).
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-frame-by-frame-leg > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats pass exit=$?"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 tools/profile_run.py monkey 1024 1920 1080 8 > $OUT/pmc$i.log 2> $OUT/pmc$i.err
  echo "pmc pass $i ($set) exit=$?"
done
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
