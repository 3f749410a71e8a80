export TMPDIR=/tmp
O=gpurun_out/r04m
mkdir -p $O
RT_PROBE_PART=lists timeout -k 10 400 python tools/scaling_probe.py 1024 20 > $O/scaling_1024spp_run2.txt 2>&1; tail -5 $O/scaling_1024spp_run2.txt
RT_PROBE_PART=lists timeout -k 10 400 python tools/scaling_probe.py 1024 20 > $O/scaling_1024spp_run3.txt 2>&1; tail -2 $O/scaling_1024spp_run3.txt
RT_PROBE_PART=both timeout -k 10 300 python tools/scaling_probe.py 256 20 > $O/scaling_256spp.txt 2>&1; tail -3 $O/scaling_256spp.txt
