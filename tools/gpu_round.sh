#!/bin/bash
# One GPU-box pass: parity tests, the default bench (configs[2]), rocprofv3 kernel stats + stream timeline + critical path, the multi-session
# scaling table (B distinct sequences), optional PMC traffic and SQ counters.   Usage: tools/gpu_round.sh <tag> [pmc] [notest]
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.   (rocprofv3 --kernel-trace adds ~50 us of host time
# per sweep: the traced single-sequence run is host-bound — kernel durations are the real ones, stream idle times are not; the untraced bench
# lines are the throughput of record.)
TAG=${1:-run}; PMC=${2:-}; NOTEST=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -z "$NOTEST" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-300 $OUT/bench_driver.json
timeout 900 python bench.py > $OUT/bench_map.json 2> $OUT/bench_map.err; cut -c1-300 $OUT/bench_map.json
export TMPDIR=/tmp
for W in map lo; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --synth-procs 1 --no-cpu-baseline --no-kernel-timer --no-extras > $OUT/prof_$W.log 2>&1)
  DB=$(find $OUT/prof_$W -name '*.db' | head -1)
  python tools/rocprof_summary.py $DB $OUT/kernel_stats_$W.txt "bench.py --workload $W ($TAG)" | head -34
  python tools/timeline.py $OUT/prof_$W > $OUT/timeline_$W.txt 2>&1; head -40 $OUT/timeline_$W.txt
  python tools/critical_path.py $OUT/prof_$W > $OUT/critical_path_$W.txt 2>&1; cat $OUT/critical_path_$W.txt
done
# multi-session scaling: B distinct sequences per launch chain (the sweeps are synthesised once and cached for the profiler passes below)
timeout 600 python tools/batch_scaling.py --configs 1x1,1x2,1x4,1x8,1x12,1x16,1x24 --table --cache /tmp/sw24.npy > $OUT/batch_scaling.txt 2>&1; grep "H x B" $OUT/batch_scaling.txt
# host sweeps in: the default (deferred ring on the copy stream) against the inline form (copy on the scan-registration stream), mixed and host-fed-only processes
( echo "# tools/host_input_probe.py (3 passes each; device / pinned / pageable handles alternate), default: deferred ring + copy stream"; timeout 200 python tools/host_input_probe.py --reps 3 --cache /tmp/hi40.npy 2>&1 | tail -3
  echo "# VLOAM_STAGE_INLINE=1 (copy on the scan-registration stream in front of the sweep)"; VLOAM_STAGE_INLINE=1 timeout 200 python tools/host_input_probe.py --reps 3 --cache /tmp/hi40.npy 2>&1 | tail -3
  echo "# --only pinned"; timeout 200 python tools/host_input_probe.py --reps 3 --only pinned --cache /tmp/hi40.npy 2>&1 | tail -1
  echo "# --only pinned, VLOAM_STAGE_INLINE=1"; VLOAM_STAGE_INLINE=1 timeout 200 python tools/host_input_probe.py --reps 3 --only pinned --cache /tmp/hi40.npy 2>&1 | tail -1
  echo "# --only pageable"; timeout 200 python tools/host_input_probe.py --reps 3 --only pageable --cache /tmp/hi40.npy 2>&1 | tail -1 ) > $OUT/host_input.txt 2>&1; cat $OUT/host_input.txt
# batched: kernel stats of B = 8
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_b8 -- python $GRAFT_REPO_ROOT/tools/batch_scaling.py --configs 1x8 --procs 1 --cache /tmp/sw24.npy > $OUT/prof_b8.log 2>&1)
DB=$(find $OUT/prof_b8 -name '*.db' | head -1)
python tools/rocprof_summary.py $DB $OUT/kernel_stats_batch8.txt "tools/batch_scaling.py --configs 1x8, eight distinct sequences ($TAG)" | head -30
if [ -n "$PMC" ] && [ "$PMC" != "-" ]; then
  for CTR in FETCH_SIZE WRITE_SIZE; do
    # counter collection serialises every dispatch: a short run (30-sweep map warm-up, 30 timed sweeps), no worker processes
    (cd /tmp && timeout 420 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc_map_$CTR -- python $GRAFT_REPO_ROOT/bench.py --workload map --map-warmup 30 --steps 30 --warmup 5 --synth-procs 1 --no-cpu-baseline --no-kernel-timer --no-extras > $OUT/pmc_map_$CTR.log 2>&1)
  done
  F=$(find $OUT/pmc_map_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wf=$(find $OUT/pmc_map_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py $F $Wf $OUT/hbm_traffic_map.txt | head -30
  rm -rf $OUT/pmc_map_FETCH_SIZE $OUT/pmc_map_WRITE_SIZE
  # shader counters of a batch (every kernel alone on the chip under the profiler): wave-microseconds, issue / wait shares, instruction mix
  for B in 1 8 16; do
    i=0
    for CTRS in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "VmemLatency" "MeanOccupancyPerCU"; do
      i=$((i+1))
      (cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pmcq_${B}_$i -- python $GRAFT_REPO_ROOT/tools/batch_scaling.py --configs 1x$B --sweeps 40 --warm 14 --steps 6 --procs 1 --cache /tmp/sw24.npy > $OUT/pmcq_${B}_$i.log 2>&1)
    done
    python tools/pmc_table.py $OUT/batch${B}_pmc.txt $(find /tmp/pmcq_${B}_* -name '*counter_collection.csv' | sort) > /dev/null
    S=$(find /tmp/pmcq_${B}_1 -name '*counter_collection.csv' | head -1)
    python tools/sq_summary.py $S $OUT/batch${B}_sq_counters.txt | head -24
    rm -rf /tmp/pmcq_${B}_*
  done
fi
find $OUT -name '*.db' -size +20M -delete
