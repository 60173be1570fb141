#!/bin/bash
# The deferred pin: ONE command on a machine with a Rust toolchain and network access.
#
#   tools/pin_reference.sh [path-to-a-snark-setup-operator-checkout]
#
# builds rust-shim/pin (the reference's own phase1-cli / phase1 / setup-utils at the operator's pins), runs it and writes
# tests/golden/arkworks_pins.json; then `python -m pytest tests/test_arkworks_pins.py` (CPU: generators, geometry,
# new_challenge, public keys; `-m gpu`: every response / new challenge / combined / beacon artefact) compares the library
# with it.  With a checkout given, its Cargo.lock is used so that every dependency version is the operator's.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PIN="$ROOT/rust-shim/pin"
command -v cargo >/dev/null || { echo "pin_reference.sh: cargo not found — this step needs a Rust toolchain and network access" >&2; exit 2; }
if [ $# -ge 1 ] && [ -f "$1/Cargo.lock" ]; then
  cp "$1/Cargo.lock" "$PIN/Cargo.lock"
fi
WORK="$(mktemp -d)"
trap 'rm -rf "$WORK"' EXIT
(cd "$PIN" && cargo run --release -- "$WORK") > "$WORK/arkworks_pins.json"
python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('pinned', len(d['cases']), 'ceremonies;', d['generator'])" "$WORK/arkworks_pins.json"
mkdir -p "$ROOT/tests/golden"
cp "$WORK/arkworks_pins.json" "$ROOT/tests/golden/arkworks_pins.json"
# arkworks' G2Affine::generator() of the MNT twists -> constants of the next build (ptau_new_challenge on MNT4-753 / MNT6-753)
python3 - "$WORK/arkworks_pins.json" "$ROOT/snark-setup-operator_amd/csrc/pinned_generators.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["generators"]
json.dump({k: {"g2_uncompressed": d[k]["g2_uncompressed"]} for k in ("mnt4_753", "mnt6_753")}, open(sys.argv[2], "w"), indent=1)
PY
echo "wrote snark-setup-operator_amd/csrc/pinned_generators.json — rebuild (python -c 'import __graft_entry__ as g; g.build()') for the built-in MNT new_challenge"
echo "wrote tests/golden/arkworks_pins.json — now: python -m pytest tests/test_arkworks_pins.py -q   (and -m gpu on an MI355X)"
