#!/bin/bash
# rocprofv3 passes for the render kernel at the bench configurations (run on the GPU box from the repo
# root).  usage: tools/profile_all.sh <tag> [scenes...]     (default scenes: monkey three_sphere cube)
#   per scene (monkey, three_sphere, cube = BASELINE configs[3], [1], [2]: 1920x1080, 1024 spp, 8 bounces; also
#   reference_scene0 = the reference's default workload, sphere50k / soup6k = meshes beyond LDS):
#     stats:   --kernel-trace --stats on bench.py in the DRIVER's shape (--steps 20 --warmup 5, no extra legs:
#              every launch of the kernel is then of the kind that is timed)
#     pmc1-3:  SQ counters on one 8-frame launch (tools/profile_run.py; a 1-spp warm-up launch comes first)
#     hbm:     FETCH_SIZE / WRITE_SIZE, each in its own run, on launches of 1, 8 and 20 frames -> traffic.json
#   PMC and tracing are never combined in one run; the program comes directly after `--`.
set -o pipefail
TAG=${1:-r02}; shift
SCENES=${@:-monkey three_sphere cube}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# per scene: bench.py arguments of the stats pass, and "spp W H" of the PMC passes (tools/profile_run.py)
declare -A BENCH=([monkey]="--config 3" [three_sphere]="--config 1" [cube]="--config 2" [reference_scene0]="--config ref0"
                  [sphere50k]="--scene sphere50k --spp 16" [soup6k]="--scene soup6k --spp 64")
declare -A PMC=([monkey]="1024 1920 1080" [three_sphere]="1024 1920 1080" [cube]="1024 1920 1080" [reference_scene0]="100 1000 800"
                [sphere50k]="16 1920 1080" [soup6k]="64 1920 1080")
for s in $SCENES; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$s -- python3 bench.py ${BENCH[$s]} --steps 20 --warmup 5 --no-cpu-baseline --no-frame-by-frame-leg > $OUT/bench_under_rocprof_$s.json 2> $OUT/stats_$s.err
  echo "$s stats pass exit=$?"
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
             "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc${i}_$s -- python3 tools/profile_run.py $s ${PMC[$s]} 8 > $OUT/pmc${i}_$s.log 2> $OUT/pmc${i}_$s.err
    echo "$s pmc pass $i exit=$?"
  done
  # HBM traffic passes: the three BASELINE 1080p scenes only (bench.py's roofline.traffic)
  case $s in monkey|three_sphere|cube) ;; *) continue ;; esac
  for f in 1 8 20; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/hbm_${s}_1920x1080_s1024_f${f}_fetch -- python3 tools/profile_run.py $s 1024 1920 1080 $f > $OUT/hbm_${s}_f${f}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/hbm_${s}_1920x1080_s1024_f${f}_write -- python3 tools/profile_run.py $s 1024 1920 1080 $f > $OUT/hbm_${s}_f${f}_write.log 2>&1
    echo "$s hbm passes f=$f exit=$?"
  done
done
# BASELINE configs[4]: the launch shape of `bench.py --config 4 --steps 2` (two frames of 3840x2160 at 4096 spp)
if [ -z "$NO_CONFIG4" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/hbm_monkey_3840x2160_s4096_f2_fetch -- python3 tools/profile_run.py monkey 4096 3840 2160 2 > $OUT/hbm_config4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/hbm_monkey_3840x2160_s4096_f2_write -- python3 tools/profile_run.py monkey 4096 3840 2160 2 > $OUT/hbm_config4_write.log 2>&1
echo "config4 hbm passes exit=$?"
fi
python3 tools/fit_traffic.py $OUT $TAG > $OUT/fit_traffic.txt 2>&1 && cp profiles/traffic.json $OUT/traffic.json
python3 tools/summarize_profile.py $OUT $SCENES > $OUT/summary.txt
cat $OUT/summary.txt
