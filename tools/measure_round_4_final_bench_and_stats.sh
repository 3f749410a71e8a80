set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O gpurun_out/prof_r04
python tools/frame_hash.py | tail -1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
bash tools/measure_round_2_stats_and_bench.sh
for s in monkey reference_scene0; do
  arg="--config 3"; [ $s = reference_scene0 ] && arg="--config ref0"
  rm -rf gpurun_out/prof_r04/stats_$s
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04/stats_$s -- python3 bench.py $arg --steps 20 --warmup 5 --no-cpu-baseline --no-frame-by-frame-leg > gpurun_out/prof_r04/bench_under_rocprof_$s.json 2> gpurun_out/prof_r04/stats_$s.err
  echo "$s stats pass exit=$?"
done
