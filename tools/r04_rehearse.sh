mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --steps 8 --warmup 2 --spp 128 --check > $O/bench_two_ranks_one_gpu_gloo.json 2> $O/two_ranks.err; echo "two ranks exit=$?"
timeout -k 10 300 python bench.py --capi-multi 0,0,0,0 --steps 8 --warmup 2 --spp 128 --no-cpu-baseline --check > $O/bench_capi_multi_4x_one_gpu.json 2> $O/capi4.err; echo "capi exit=$?"
timeout -k 10 300 python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --spp 8 > $O/bench_two_ranks_rccl_refused.json 2> $O/rccl_refused.err; echo "rccl strict exit=$? (expected 3)"
