mkdir -p gpurun_out/r04n
for w in w112 w438 w214 w213; do
  echo "== $w" | tee -a gpurun_out/r04n/weights.txt
  RT_AMD_LIB=$PWD/ray-tracer_amd/libraytracer_amd_$w.so RT_PROBE_N=1,8 RT_PROBE_PART=lists timeout -k 10 200 python tools/scaling_probe.py 1024 20 2>&1 | grep "N=" | tee -a gpurun_out/r04n/weights.txt
done
