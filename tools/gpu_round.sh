#!/bin/bash
# One GPU-box pass: parity tests, bench line, rocprofv3 kernel stats, PMC passes.  Run from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/gpu_round.sh r01'
# Every step writes under gpurun_out/<tag>/; steps are chained with && (no GPU step after a failed one).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-extras --cpu-sample 4096"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
  tail -3 $OUT/pytest_gpu.log
fi
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
cd /tmp &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/stats -o stats --output-format csv -- $BENCH > $R/$OUT/stats.log 2>&1 &&
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $c -d $R/$OUT/pmc/$d -o pmc --output-format csv -- $BENCH > $R/$OUT/pmc_$d.log 2>&1 || { tail -5 $R/$OUT/pmc_$d.log; exit 1; }
done
cd $R
python3 tools/pmc_summarize.py $OUT/pmc $OUT/pmc_summary.json "python3 bench.py --steps 3 --warmup 1 --no-extras --cpu-sample 4096"
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
# keep the merged-back payload small
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +20M -delete
ls -la $OUT
