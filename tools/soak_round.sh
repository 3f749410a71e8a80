#!/bin/bash
# The round's randomised soaks on the final build (GPU box): HIP against oracle, bit for bit; the multi-GPU entry points against the single-context frame.
# Progress goes to files under gpurun_out/ (nothing is piped through tail: a silent run is taken for a hung one).
O=gpurun_out/soak; mkdir -p $O
timeout -k 10 900 python tests/soak/soak_parity.py 500000 6000 random > $O/soak_random.txt 2>&1; tail -1 $O/soak_random.txt
timeout -k 10 400 python tests/soak/soak_parity.py 510000 900 config > $O/soak_config.txt 2>&1; tail -1 $O/soak_config.txt
timeout -k 10 400 python tests/soak/soak_parity.py 520000 600 bigmesh > $O/soak_bigmesh.txt 2>&1; tail -1 $O/soak_bigmesh.txt
RT_SOAK_SIZE=640x360x32 timeout -k 10 300 python tests/soak/soak_parity.py 530000 48 config > $O/soak_large_frames.txt 2>&1; tail -1 $O/soak_large_frames.txt
timeout -k 10 500 python tests/soak/soak_partition.py 540000 4000 > $O/soak_partition.txt 2>&1; tail -1 $O/soak_partition.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python tests/soak/soak_pipeline.py 550000 5000 > $O/soak_pipeline.txt 2>&1; tail -1 $O/soak_pipeline.txt
