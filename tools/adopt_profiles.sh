#!/bin/bash
# Copy the tables of one tools/gpu_round.sh pass (gpurun_out/<tag>/) into profiles/ under this round's names.   Usage: tools/adopt_profiles.sh <tag> [round, default r06]
# bench.py quotes a table only while its csrc hash matches the sources it times: run the bench lines AFTER this, then copy them with the second form:
#   tools/adopt_profiles.sh --bench <tag>      (gpurun_out/<tag>/bench_driver.json, bench_map.json[, bench_1500.json])
R=${3:-r06}
if [ "$1" = "--bench" ]; then
  T=gpurun_out/$2; R=${3:-r06}
  cp $T/bench_driver.json profiles/${R}_map_bench_driver_args.json
  cp $T/bench_map.json profiles/${R}_map_bench.json
  [ -f $T/bench_1500.json ] && cp $T/bench_1500.json profiles/${R}_map_bench_1500_sweeps.json
  exit 0
fi
T=gpurun_out/$1; R=${2:-r06}
cp $T/kernel_stats_map.txt profiles/${R}_map_kernel_stats.txt
cp $T/kernel_stats_lo.txt profiles/${R}_lo_kernel_stats.txt
cp $T/kernel_stats_batch8.txt profiles/${R}_batch8_kernel_stats.txt
cp $T/hbm_traffic_map.txt profiles/${R}_map_hbm_traffic.txt
for B in 1 8 16; do cp $T/batch${B}_sq_counters.txt profiles/${R}_batch${B}_sq_counters.txt; done
cp $T/batch1_pmc.txt profiles/${R}_batch1_pmc.txt
cp $T/batch8_pmc.txt profiles/${R}_batch8_pmc.txt
cp $T/batch_scaling.txt profiles/${R}_batch_scaling.txt
cp $T/critical_path_map.txt profiles/${R}_map_critical_path_traced.txt
cp $T/timeline_map.txt profiles/${R}_map_stream_timeline.txt
head -3 profiles/${R}_map_kernel_stats.txt | tail -1
