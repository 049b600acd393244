#!/bin/bash
# One GPU-box pass: parity tests, both bench workloads, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh <tag> [pmc]
# Everything lands in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-run}; PMC=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
for W in lo map; do
  timeout 600 python bench.py --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err; tail -1 $OUT/bench_$W.json
done
export TMPDIR=/tmp
for W in lo map; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$W -- python $GRAFT_REPO_ROOT/bench.py --workload $W --no-cpu-baseline --no-kernel-timer --vo-frames 0 --sessions 0 > $OUT/prof_$W.log 2>&1)
  DB=$(find $OUT/prof_$W -name '*.db' | head -1)
  python tools/rocprof_summary.py $DB $OUT/kernel_stats_$W.txt "bench.py --workload $W ($TAG)" | head -32
done
if [ -n "$PMC" ]; then
  for W in lo map; do
    for CTR in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 600 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc_${W}_$CTR -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timer --vo-frames 0 --sessions 0 > $OUT/pmc_${W}_$CTR.log 2>&1)
    done
    F=$(find $OUT/pmc_${W}_FETCH_SIZE -name '*counter_collection.csv' | head -1); Wf=$(find $OUT/pmc_${W}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
    python tools/pmc_summary.py $F $Wf $OUT/hbm_traffic_$W.txt | head -12
    rm -rf $OUT/pmc_${W}_FETCH_SIZE $OUT/pmc_${W}_WRITE_SIZE
  done
fi
find $OUT -name '*.db' -size +20M -delete
