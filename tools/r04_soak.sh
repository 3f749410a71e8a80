mkdir -p gpurun_out/r04j; O=gpurun_out/r04j
( timeout -k 10 500 python tests/soak/soak_parity.py 400000 2000 random 2>&1 | grep -v amdgpu | tail -3 ) | tee $O/soak_random.txt
( timeout -k 10 300 python tests/soak/soak_parity.py 410000 400 config 2>&1 | grep -v amdgpu | tail -3 ) | tee $O/soak_config.txt
( timeout -k 10 300 python tests/soak/soak_parity.py 420000 300 bigmesh 2>&1 | grep -v amdgpu | tail -3 ) | tee $O/soak_bigmesh.txt
( timeout -k 10 300 python tests/soak/soak_partition.py 430000 1500 2>&1 | grep -v amdgpu | tail -2 ) | tee $O/soak_partition.txt
