#!/bin/bash
# Families object of the bench line, twice (run-to-run spread), after the row-family tests:  bash tools/dev/fam_ab.sh <out dir>
set -u
D=${1:-gpurun_out/fam_ab}; mkdir -p "$D"
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_rowfamilies.py tests/test_gpu_batch.py tests/test_gpu_random_shapes.py -x -q --timeout 120 --timeout-method thread > "$D/tests.log" 2>&1 || { tail -15 "$D/tests.log"; exit 1; }
tail -1 "$D/tests.log"
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-block-calib > "$D/bench$i.log" 2> "$D/bench$i.err" || { tail -5 "$D/bench$i.err"; exit 1; }
  python3 - "$D/bench$i.log" <<'PY'
import json, sys
b = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(b["value"], b["ms_per_step"], " ".join(f"{k}={v['frac_hbm_peak']:.3f}" for k, v in b["families_c3_c4"].items()))
PY
done
