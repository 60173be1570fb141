#!/bin/bash
# A/B of the window-width choice of Auto's probe (PTAU_TEST_COST_SCALE weights a bucket test in msm_plan's cost model)
mkdir -p gpurun_out/ab_probe
for s in 1 0.25 0.5 2 4 8; do
  PTAU_TEST_COST_SCALE=$s timeout -k 10 200 python3 bench.py --config 2 --steps 5 --warmup 1 > gpurun_out/ab_probe/s_$s.json 2> gpurun_out/ab_probe/s_$s.err || exit 1
  python3 - "$s" <<'PY'
import json,sys
s=sys.argv[1]
d=json.loads(open(f"gpurun_out/ab_probe/s_{s}.json").read().strip().split("\n")[-1])
def find(o,k):
    if isinstance(o,dict):
        for kk,v in o.items():
            if kk==k: yield v
            yield from find(v,k)
    elif isinstance(o,list):
        for v in o: yield from find(v,k)
print("scale",s,"value",round(d["value"]/1e6,2),"ms/step",round(d["ms_per_step"],2),"probe_ms",[round(x,2) for x in find(d,"membership_probe_ms")][:3],flush=True)
PY
done
