set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 120 tools/ubench/valu_tput > $O/valu_tput.txt 2>&1; echo "ubench exit=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err; echo "bench exit=$?"
bash tools/pmc_bound.sh base monkey 1024 > $O/pmc_bound.txt 2>&1; echo "pmc exit=$?"
for shape in "256 4" "256 5" "256 6" "512 2" "1024 1"; do set -- $shape; echo "shape $1 x $2" >> $O/occ_probe.txt; RT_AMD_THREADS=$1 RT_AMD_BLOCKS_PER_CU=$2 timeout -k 10 120 python tools/occupancy_probe.py 150 1024 20 >> $O/occ_probe.txt 2>&1; done; echo occ done
