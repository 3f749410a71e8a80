#!/bin/bash
# Copies what tools/profile_all.sh <tag> left under gpurun_out/prof_<tag> (scratch, merged back by gpurun) into
# profiles/<tag>/ (tracked) under the names profiles/README.md lists.  usage: tools/collect_profiles.sh <tag>
set -e
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/gpurun_out/prof_$TAG
P=$ROOT/profiles/$TAG
mkdir -p $P
cp $G/summary.txt $P/rocprofv3_summary.txt
[ -f $G/fit_traffic.txt ] && cp $G/fit_traffic.txt $P/
[ -f $G/traffic.json ] && cp $G/traffic.json $ROOT/profiles/traffic.json
# (a scratch directory that has seen several runs holds one file set per run: the newest is the one the summary read)
newest() { ls -t $1 2>/dev/null | head -1; }
for s in monkey three_sphere cube reference_scene0 sphere50k soup6k; do
  [ -d $G/stats_$s ] || continue
  cp "$(newest "$G/stats_$s/*/*_kernel_stats.csv")" $P/kernel_stats_$s.csv
  cp "$(newest "$G/stats_$s/*/*_kernel_trace.csv")" $P/kernel_trace_$s.csv
  cp $G/bench_under_rocprof_$s.json $P/
  for i in 1 2 3; do [ -d $G/pmc${i}_$s ] && cp "$(newest "$G/pmc${i}_$s/*/*_counter_collection.csv")" $P/pmc${i}_${s}_counter_collection.csv; done
done
for d in $G/hbm_*_fetch $G/hbm_*_write; do
  [ -d $d ] && cp "$(newest "$d/*/*_counter_collection.csv")" $P/$(basename $d)_counter_collection.csv
done
echo "collected into $P"
