set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
NO_CONFIG4= bash tools/profile_all.sh r04 reference_scene0 sphere50k > $O/profile_all2.log 2>&1; echo "profile_all2 exit=$?"; grep "exit=" $O/profile_all2.log | tr '\n' ' '
RT_AMD_LIB=$PWD/ray-tracer_amd/libraytracer_amd_stats.so timeout -k 10 120 python tools/stats_run.py reference_scene0 100 8 > $O/section_stats_reference_scene0_f8.txt 2>&1
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 8 --warmup 2 --spp 128 --check > $O/bench_two_ranks_one_gpu_gloo.json 2> $O/two_ranks.err; echo "two ranks exit=$?"
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --spp 8 > $O/bench_two_ranks_rccl_refused.json 2> $O/rccl_refused.err; echo "rccl strict exit=$? (expected non-zero: two ranks on one device)"
timeout -k 10 300 python bench.py --capi-multi 0,0,0,0 --steps 8 --warmup 2 --spp 128 --no-cpu-baseline --check > $O/bench_capi_multi_4x_one_gpu.json 2>/dev/null; echo "capi exit=$?"
RT_PROBE_PART=both RT_PROBE_DUMP=$O/probe_1024.json timeout -k 10 400 python tools/scaling_probe.py 1024 20 > $O/scaling_1024spp.txt 2>&1; echo "probe1024 exit=$?"
tail -12 $O/scaling_1024spp.txt
