#!/bin/bash
# A/B of library builds on ONE box (box-to-box clocks differ by a few per cent): ab/<name>.so variants are copied over the
# in-tree library in turn and bench.py reports the G1 kernel time and the whole-chunk device times.
#   [CFG=3] tools/ab_bench.sh A B A B     (run through gpurun from the repo root; results under gpurun_out/ab/)
set -e
mkdir -p gpurun_out/ab
cp snark-setup-operator_amd/libptau_hip.so /tmp/libptau_hip.keep
i=0
for v in "$@"; do
  i=$((i+1))
  cp "ab/$v.so" snark-setup-operator_amd/libptau_hip.so
  timeout -k 10 300 python bench.py --config ${CFG:-2} --steps 5 --warmup 1 > "gpurun_out/ab/$i-$v.json" 2> "gpurun_out/ab/$i-$v.err"
  python - "$v" "gpurun_out/ab/$i-$v.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); e = d.get("extras", {})
print(sys.argv[1], "G1 exact %.2f ms endo %.2f ms" % (d["roofline"]["kernel_ms"], d.get("endomorphism_path", {}).get("kernel_ms", 0.0)),
      " ".join("%s %.1f" % (k.replace("_chunk", ""), v["device_ms"]) for k, v in e.items() if isinstance(v, dict) and "device_ms" in v), flush=True)
PY
done
cp /tmp/libptau_hip.keep snark-setup-operator_amd/libptau_hip.so
