mkdir -p gpurun_out/r04e; O=gpurun_out/r04e
for sc in "sphere50k 16" "soup6k 64"; do set -- $sc
  echo "== $1 hybrid (default)"; timeout -k 10 200 python3 tools/profile_run.py $1 $2 1920 1080 8 | tail -1
  echo "== $1 all-global 1024x1"; RT_AMD_SCENE_MODE=0 timeout -k 10 200 python3 tools/profile_run.py $1 $2 1920 1080 8 | tail -1
  echo "== $1 all-global 256x5"; RT_AMD_SCENE_MODE=0 RT_AMD_GLOBAL_THREADS=256 timeout -k 10 200 python3 tools/profile_run.py $1 $2 1920 1080 8 | tail -1
  echo "== $1 all-global 256x6"; RT_AMD_SCENE_MODE=0 RT_AMD_GLOBAL_THREADS=256 RT_AMD_BLOCKS_PER_CU=6 timeout -k 10 200 python3 tools/profile_run.py $1 $2 1920 1080 8 | tail -1
done > $O/big_modes.txt 2>&1
