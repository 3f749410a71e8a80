set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit=$?"; tail -2 $O/pytest.log
NO_CONFIG4= bash tools/profile_all.sh r04 monkey three_sphere cube reference_scene0 sphere50k > $O/profile_all.log 2>&1; echo "profile_all exit=$?"; grep "exit=" $O/profile_all.log | tr '\n' ' '
for s in monkey three_sphere cube; do RT_AMD_LIB=/root/repo/ray-tracer_amd/libraytracer_amd_stats.so timeout -k 10 120 python tools/stats_run.py $s 256 8 > $O/section_stats_${s}_f8.txt 2>&1; done; echo stats done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_config3_driver.json 2> $O/bench_config3_driver.err; echo "bench3 exit=$?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2>/dev/null; echo "bench default exit=$?"
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 5 > $O/bench_config1.json 2>/dev/null; echo "bench1 exit=$?"
timeout -k 10 300 python bench.py --config 2 --steps 20 --warmup 5 > $O/bench_config2.json 2>/dev/null; echo "bench2 exit=$?"
timeout -k 10 300 python bench.py --config 4 --steps 2 --warmup 1 > $O/bench_config4.json 2>/dev/null; echo "bench4 exit=$?"
timeout -k 10 300 python bench.py --config ref0 --steps 20 --warmup 5 > $O/bench_ref0.json 2>/dev/null; echo "ref0 exit=$?"
timeout -k 10 300 python bench.py --scene sphere50k --spp 16 --steps 8 --warmup 2 > $O/bench_sphere50k.json 2>/dev/null; echo "sphere50k exit=$?"
timeout -k 10 300 python bench.py --scene soup6k --spp 64 --steps 8 --warmup 2 > $O/bench_soup6k.json 2>/dev/null; echo "soup6k exit=$?"
RT_AMD_SCENE_MODE=0 timeout -k 10 300 python bench.py --scene sphere50k --spp 16 --steps 8 --warmup 2 --no-cpu-baseline --no-frame-by-frame-leg > $O/bench_sphere50k_all_global.json 2>/dev/null
RT_AMD_SCENE_MODE=0 timeout -k 10 300 python bench.py --scene soup6k --spp 64 --steps 8 --warmup 2 --no-cpu-baseline --no-frame-by-frame-leg > $O/bench_soup6k_all_global.json 2>/dev/null; echo "global variants done"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 8 --warmup 2 --spp 128 --check > $O/bench_two_ranks_one_gpu_gloo.json 2> $O/two_ranks.err; echo "two ranks exit=$?"
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --spp 8 > $O/bench_two_ranks_rccl_refused.json 2> $O/rccl_refused.err; echo "rccl strict exit=$? (expected non-zero: two ranks on one device)"
timeout -k 10 300 python bench.py --capi-multi 0,0,0,0 --steps 8 --warmup 2 --spp 128 --no-cpu-baseline --check > $O/bench_capi_multi_4x_one_gpu.json 2>/dev/null; echo "capi exit=$?"
RT_PROBE_PART=both RT_PROBE_DUMP=$O/probe_1024.json timeout -k 10 300 python tools/scaling_probe.py 1024 20 > $O/scaling_1024spp.txt 2>&1; echo "probe1024 exit=$?"
RT_PROBE_PART=both timeout -k 10 300 python tools/scaling_probe.py 256 20 > $O/scaling_256spp.txt 2>&1; echo "probe256 exit=$?"
cat $O/scaling_1024spp.txt
