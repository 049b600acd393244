#!/bin/bash
# One GPU-box pass: parity tests, the default bench (configs[2]), rocprofv3 kernel stats + stream timeline, optional PMC traffic.
# Usage: tools/gpu_round.sh <tag> [pmc] [notest]     Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-run}; PMC=${2:-}; NOTEST=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -z "$NOTEST" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; cut -c1-400 $OUT/bench_driver.json
timeout 600 python bench.py > $OUT/bench_map.json 2> $OUT/bench_map.err; cut -c1-400 $OUT/bench_map.json
export TMPDIR=/tmp
for W in map lo; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --synth-procs 1 --no-cpu-baseline --no-kernel-timer --no-extras > $OUT/prof_$W.log 2>&1)
  DB=$(find $OUT/prof_$W -name '*.db' | head -1)
  python tools/rocprof_summary.py $DB $OUT/kernel_stats_$W.txt "bench.py --workload $W ($TAG)" | head -34
  python tools/timeline.py $OUT/prof_$W > $OUT/timeline_$W.txt 2>&1; head -60 $OUT/timeline_$W.txt
done
# image front-end alone: per-kernel stats of tools/image_probe.py
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_image -- python $GRAFT_REPO_ROOT/tools/image_probe.py --no-table > $OUT/prof_image.log 2>&1)
DB=$(find $OUT/prof_image -name '*.db' | head -1)
python tools/rocprof_summary.py $DB $OUT/kernel_stats_image.txt "tools/image_probe.py --no-table ($TAG)" | head -16
timeout 200 python tools/image_probe.py > $OUT/image_probe.txt 2>&1; grep -v amdgpu.ids $OUT/image_probe.txt | head -24
if [ -n "$PMC" ] && [ "$PMC" != "-" ]; then
  for W in map; do
    for CTR in FETCH_SIZE WRITE_SIZE; do
      # counter collection serialises every dispatch: a short run (30-sweep map warm-up, 30 timed sweeps), no worker processes
      (cd /tmp && timeout 420 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc_${W}_$CTR -- python $GRAFT_REPO_ROOT/bench.py --workload $W --map-warmup 30 --steps 30 --warmup 5 --synth-procs 1 --no-cpu-baseline --no-kernel-timer --no-extras > $OUT/pmc_${W}_$CTR.log 2>&1)
    done
    F=$(find $OUT/pmc_${W}_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wf=$(find $OUT/pmc_${W}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
    python tools/pmc_summary.py $F $Wf $OUT/hbm_traffic_$W.txt | head -30
    rm -rf $OUT/pmc_${W}_FETCH_SIZE $OUT/pmc_${W}_WRITE_SIZE
  done
fi
find $OUT -name '*.db' -size +20M -delete
