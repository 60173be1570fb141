#!/bin/bash
# A/B of the verify path's device section on one box: default (batched membership through the MSM buckets + MSM tails on a
# second stream), PTAU_BATCHED_SUBGROUP=0 (exact per-element membership), and additionally PTAU_MSM_TAIL_OVERLAP=0.
#   gpurun --timeout 900 -- 'bash tools/ab_verify_tail.sh'
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 700 python3 -m pytest tests/test_gpu_verify.py tests/test_gpu_tail.py -m gpu -x -q > gpurun_out/vt_tests.log 2>&1 || { tail -40 gpurun_out/vt_tests.log; exit 1; }
  tail -3 gpurun_out/vt_tests.log
fi
cd /tmp && export TMPDIR=/tmp
for v in default exact exact_serial; do
  [ $v = exact ] && export PTAU_BATCHED_SUBGROUP=0
  [ $v = exact_serial ] && export PTAU_BATCHED_SUBGROUP=0 PTAU_MSM_TAIL_OVERLAP=0
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/vt_$v -o vt --output-format csv -- python3 $R/tools/verify_timeline.py run > $R/gpurun_out/vt_$v.log 2>&1 || { tail $R/gpurun_out/vt_$v.log; exit 1; }
  python3 $R/tools/verify_timeline.py read $R/gpurun_out/vt_$v --launches > $R/gpurun_out/vt_${v}_timeline.txt
  grep "^1 " $R/gpurun_out/vt_$v.log | cut -c1-400
done
cd $R
find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -delete
