set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
for s in monkey three_sphere cube; do RT_AMD_LIB=$PWD/ray-tracer_amd/libraytracer_amd_stats.so timeout -k 10 120 python tools/stats_run.py $s 256 8 > $O/section_stats_${s}_f8.txt 2>&1; done; echo stats done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_config3_driver.json 2> $O/bench_config3_driver.err; echo "bench3 exit=$?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2>/dev/null; echo "bench default exit=$?"
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 5 > $O/bench_config1.json 2>/dev/null; echo "bench1 exit=$?"
timeout -k 10 300 python bench.py --config 2 --steps 20 --warmup 5 > $O/bench_config2.json 2>/dev/null; echo "bench2 exit=$?"
timeout -k 10 300 python bench.py --config 4 --steps 2 --warmup 1 > $O/bench_config4.json 2>/dev/null; echo "bench4 exit=$?"
timeout -k 10 300 python bench.py --config ref0 --steps 20 --warmup 5 > $O/bench_ref0.json 2>/dev/null; echo "ref0 exit=$?"
timeout -k 10 300 python bench.py --scene sphere50k --spp 16 --steps 8 --warmup 2 > $O/bench_sphere50k.json 2>/dev/null; echo "sphere50k exit=$?"
timeout -k 10 300 python bench.py --scene soup6k --spp 64 --steps 8 --warmup 2 > $O/bench_soup6k.json 2>/dev/null; echo "soup6k exit=$?"
