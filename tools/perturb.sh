#!/bin/bash
# Perturbation experiment (round 4): monkey 8 x 256 spp on library builds that add 8 instructions of one class to every node step
# (-DRT_EXP_PERTURB=k: 0 none, 1 8 x v_mul_f32, 2 8 x v_max_f32, 3 4 x ds_read_b128 + wait, 4 8 x s_and_b64, 5 8 x s_nop, 6 16 x v_mul_f32)
for r in 1 2; do for k in "$@"; do
  echo -n "p$k: "; RT_AMD_LIB=$PWD/ray-tracer_amd/libraytracer_amd_p$k.so timeout -k 10 120 python3 tools/profile_run.py monkey 256 1920 1080 8 2>&1 | tail -1
done; done
