mkdir -p gpurun_out/r04c
RT_AMD_LIB=$PWD/ray-tracer_amd/libraytracer_amd_stats.so timeout -k 10 200 python3 tools/stats_run.py monkey 256 8 > gpurun_out/r04c/section_stats_monkey_f8.txt 2>&1
timeout -k 10 1000 python3 tools/sweep_knobs.py monkey 256 8 2 "" "RT_AMD_DESCEND_KEEP=16" "RT_AMD_DESCEND_KEEP=20" "RT_AMD_DESCEND_KEEP=28" "RT_AMD_DESCEND_KEEP=32" "RT_AMD_READY_BREAK=32" "RT_AMD_READY_BREAK=36" "RT_AMD_READY_BREAK=44" "RT_AMD_READY_BREAK=48" "RT_AMD_HIT_BREAK=16" "RT_AMD_HIT_BREAK=20" "RT_AMD_HIT_BREAK=28" "RT_AMD_HIT_BREAK=32" "RT_AMD_WORK_THRESHOLD=2" "RT_AMD_WORK_THRESHOLD=8" "RT_AMD_HIT_LOW=12" "RT_AMD_HIT_LOW=20" "RT_AMD_MIX_BREAK=32" "RT_AMD_MIX_BREAK=48" > gpurun_out/r04c/knobs_monkey.txt 2>&1
echo sweep exit=$?
