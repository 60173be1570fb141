#!/bin/bash
# One GPU-box pass: parity tests, then PER CONFIG: PMC passes -> summary -> kernel-trace stats -> the bench line (which reads
# the summary just written: PTAU_BENCH_PMC), so a line never carries counters of another config, round or build.  From the
# repo root:
#   gpurun --timeout 1100 -- 'bash tools/gpu_round.sh r03 "2 3 4 5"'
# Every step writes under gpurun_out/<tag>/; steps are chained with exit-on-failure (no GPU step after a failed one).
set -o pipefail
TAG=${1:-r03}
CONFIGS=${2:-"2"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
  tail -3 $OUT/pytest_gpu.log
fi
cd /tmp
if [ "${SKIP_CALIB:-0}" != "1" ] && [ -x $R/tools/ubench_fetch ]; then
  # FETCH_SIZE calibration on known byte counts (16 B/lane streaming vs K1's per-lane chunks and byte reads)
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/fetch_calib -o pmc --output-format csv -- $R/tools/ubench_fetch > $R/$OUT/ubench_fetch.log 2>&1 || { tail -5 $R/$OUT/ubench_fetch.log; exit 1; }
  python3 - $R/$OUT/fetch_calib $R/$OUT/ubench_fetch.txt <<'PY'
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            k = row["Kernel_Name"].split("(")[0]
            acc.setdefault(k, []).append(float(row["Counter_Value"]) * 1024.0)
known = {"stream16": 1 << 30, "lane_chunks": ((1 << 30) // 168) * 168, "lane_bytes": ((1 << 30) // 96) * 96}
with open(sys.argv[2], "w") as o:
    o.write("FETCH_SIZE (KiB x 1024) per launch vs bytes actually read, gfx950, rocprofv3 --pmc FETCH_SIZE -- tools/ubench_fetch\n")
    for k, v in sorted(acc.items()):
        b = next((known[n] for n in known if n in k), None)
        o.write("%-14s launches %d  FETCH_SIZE %.0f B  read %s B  ratio FETCH/read %.3f\n" % (k, len(v), sum(v) / len(v), b, (sum(v) / len(v)) / b if b else float("nan")))
print(open(sys.argv[2]).read())
PY
fi
for cfg in $CONFIGS; do
  # --exact-only: every launch the profiler sees is the path `value` / `roofline` describe (the endomorphism path runs the same kernel)
  BARGS="--config $cfg --steps 2 --warmup 1 --no-extras --cpu-sample 512 --exact-only"
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    d=$(echo $c | tr ' ' '_' | cut -c1-40)
    timeout -k 10 400 rocprofv3 --pmc $c -d $R/$OUT/pmc_cfg$cfg/$d -o pmc --output-format csv -- python3 $R/bench.py $BARGS > $R/$OUT/pmc_cfg${cfg}_$d.log 2>&1 || { tail -5 $R/$OUT/pmc_cfg${cfg}_$d.log; exit 1; }
  done
  python3 $R/tools/pmc_summarize.py $R/$OUT/pmc_cfg$cfg $R/$OUT/pmc_cfg$cfg.json "python3 bench.py $BARGS"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/stats_cfg$cfg -o stats --output-format csv -- python3 $R/bench.py $BARGS > $R/$OUT/stats_cfg$cfg.log 2>&1 || { tail -5 $R/$OUT/stats_cfg$cfg.log; exit 1; }
  find $R/$OUT/stats_cfg$cfg -name '*kernel_stats.csv' -exec cp {} $R/$OUT/kernel_stats_cfg$cfg.csv \;
  # the bench line of this config, carrying the counters collected a minute ago on this box and build
  PTAU_BENCH_PMC=$R/$OUT/pmc_cfg$cfg.json timeout -k 10 500 python3 $R/bench.py --config $cfg --steps 5 --warmup 1 ${BENCH_EXTRA:-} > $R/$OUT/bench_cfg$cfg.json 2> $R/$OUT/bench_cfg$cfg.err || { tail -20 $R/$OUT/bench_cfg$cfg.err; exit 1; }
  cut -c1-300 $R/$OUT/bench_cfg$cfg.json
done
cd $R
# keep the merged-back payload small
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*agent_info.csv' -delete
ls $OUT
