#!/bin/bash
# One GPU-box pass: parity tests, bench lines of every config, rocprofv3 kernel stats, PMC passes per config.  From the repo root:
#   gpurun --timeout 1100 -- 'bash tools/gpu_round.sh r02 "2 3 4 5"'
# Every step writes under gpurun_out/<tag>/; steps are chained with && / exit-on-failure (no GPU step after a failed one).
set -o pipefail
TAG=${1:-r02}
CONFIGS=${2:-"2"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
  tail -3 $OUT/pytest_gpu.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
  cat $OUT/bench.json
fi
cd /tmp
for cfg in $CONFIGS; do
  # --exact-only: every launch the profiler sees is the path `value` / `roofline` describe (the endomorphism path runs the same kernel)
  BARGS="--config $cfg --steps 2 --warmup 1 --no-extras --cpu-sample 512 --exact-only"
  timeout -k 10 400 python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-extras --cpu-sample 4096 > $R/$OUT/bench_cfg$cfg.json 2> $R/$OUT/bench_cfg$cfg.err || { tail -20 $R/$OUT/bench_cfg$cfg.err; exit 1; }
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/stats_cfg$cfg -o stats --output-format csv -- python3 $R/bench.py $BARGS > $R/$OUT/stats_cfg$cfg.log 2>&1 || { tail -5 $R/$OUT/stats_cfg$cfg.log; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    d=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout -k 10 400 rocprofv3 --pmc $c -d $R/$OUT/pmc_cfg$cfg/$d -o pmc --output-format csv -- python3 $R/bench.py $BARGS > $R/$OUT/pmc_cfg${cfg}_$d.log 2>&1 || { tail -5 $R/$OUT/pmc_cfg${cfg}_$d.log; exit 1; }
  done
  python3 $R/tools/pmc_summarize.py $R/$OUT/pmc_cfg$cfg $R/$OUT/pmc_cfg$cfg.json "python3 bench.py $BARGS"
  find $R/$OUT/stats_cfg$cfg -name '*kernel_stats.csv' -exec cp {} $R/$OUT/kernel_stats_cfg$cfg.csv \;
done
cd $R
# keep the merged-back payload small
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*agent_info.csv' -delete
ls $OUT
